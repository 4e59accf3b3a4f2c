#!/bin/bash
# instruction mix per kernel of the configs[2] training step (two_stage_bench.py), like tools/profile_insts.sh for bench.py
# usage: tools/profile_insts_c2.sh <out-name> [extra two_stage_bench flags]      (through gpurun)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-insts_c2}
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/pmc" -o insts -- python3 "$REPO/tools/two_stage_bench.py" --steps 2 --warmup 2 --json "$@" > /dev/null 2> "$OUT/err.txt"
cd "$REPO"
C=$(find "$OUT/pmc" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_pmc.py "$C" > "$OUT/insts.md"
rm -rf "$OUT/pmc"

#!/bin/bash
# dynamic instruction mix per kernel of the headline step: vector ALU, MFMA, scalar, LDS instructions issued (SQ_INSTS_*)
# usage: tools/profile_insts.sh <out-name> ["extra bench flags"]      (through gpurun; table in gpurun_out/<out-name>/insts.md)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-insts}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/pmc" -o insts -- python3 "$REPO/bench.py" --steps 2 --warmup 2 --no-aux --no-cpu-baseline ${2:-} > /dev/null 2> "$OUT/err.txt"
cd "$REPO"
C=$(find "$OUT/pmc" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_pmc.py "$C" > "$OUT/insts.md"
rm -rf "$OUT/pmc"

#!/bin/bash
# SQ counters (LDS conflicts, MFMA busy, waits) of the configs[4] step's kernels
# usage: tools/profile_c4_sq.sh <out-name> [two_stage_bench flags; default: the configs[4] step]      (through gpurun)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-c4sq}
shift
FLAGS=${*:---two-scales --bf16 --dcn-neck}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/sq" -o sq -- python3 "$REPO/tools/two_stage_bench.py" --steps 1 --warmup 1 --json $FLAGS > "$OUT/step.json" 2> "$OUT/err.txt"
cd "$REPO"
python3 tools/summarize_pmc.py "$(find "$OUT/sq" -name '*counter_collection.csv' | head -1)" > "$OUT/sq_counters.md"
rm -rf "$OUT/sq"

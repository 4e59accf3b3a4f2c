#!/bin/bash
# rocprofv3 kernel table of the configs[4] step (dual pathway, bf16, DeformConv3d neck)
# usage: tools/profile_c4.sh <out-name>      (through gpurun; outputs under gpurun_out/<out-name>/)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-c4}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c4" -o c4 -- python3 "$REPO/tools/two_stage_bench.py" --steps 3 --warmup 2 --json --two-scales --bf16 --dcn-neck > "$OUT/config4_step.json" 2> "$OUT/c4.err"
cd "$REPO"
KS=$(find "$OUT/c4" -name '*kernel_stats.csv' | head -1)
cp "$KS" "$OUT/config4_step_kernel_stats.csv"
python3 tools/summarize_prof.py "$KS" > "$OUT/config4_step_summary.md"
rm -rf "$OUT/c4"
head -45 "$OUT/config4_step_summary.md" | cut -c1-140

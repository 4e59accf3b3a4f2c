#!/bin/bash
# per-kernel times of the DeformConv3d layer (fp32 column-free and bf16 LDS-box kernels) at a pyramid-level shape
# usage: tools/profile_dcn.sh <out-name> [dcn_bench.py flags]      (through gpurun; outputs under gpurun_out/<out-name>/)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-dcn}
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/tr" -o dcn -- python3 "$REPO/tools/dcn_bench.py" "$@" > "$OUT/dcn_bench.txt" 2> "$OUT/err.txt"
cd "$REPO"
KS=$(find "$OUT/tr" -name '*kernel_stats.csv' | head -1)
cp "$KS" "$OUT/dcn_kernel_stats.csv"
python3 tools/summarize_prof.py "$KS" > "$OUT/dcn_summary.md"
rm -rf "$OUT/tr"
cat "$OUT/dcn_bench.txt"
grep -E "dcn_" "$OUT/dcn_summary.md" | cut -c1-150

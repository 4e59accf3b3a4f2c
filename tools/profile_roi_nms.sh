#!/bin/bash
# rocprofv3 evidence for the RoIAlign3D / NMS3D kernels (second half of BASELINE's metric): kernel durations, then SQ counters
# in their own pass (program directly after `--`).  usage: tools/profile_roi_nms.sh <out-name>
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-roinms}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W="$REPO/tools/roi_nms_workload.py"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 "$W" 20 > /dev/null 2> "$OUT/kt.err"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/sq" -o sq -- python3 "$W" 3 > /dev/null 2> "$OUT/sq.err"
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch -- python3 "$W" 3 > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write -- python3 "$W" 3 > /dev/null 2> "$OUT/write.err"
cd "$REPO"
KS=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)
SC=$(find "$OUT/sq" -name '*counter_collection.csv' | head -1)
FC=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
WC=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
cp "$KS" "$OUT/roi_nms_kernel_stats.csv"
python3 tools/summarize_pmc.py "$SC" > "$OUT/roi_nms_sq_counters.md"
python3 tools/summarize_prof.py "$KS" "$FC" "$WC" > "$OUT/roi_nms_summary.md"
rm -rf "$OUT/kt" "$OUT/sq" "$OUT/fetch" "$OUT/write"
ls -la "$OUT"

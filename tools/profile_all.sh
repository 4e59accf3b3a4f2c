#!/bin/bash
# The whole evidence set of a round in one gpurun call: everything profiles/<round>/ holds, under gpurun_out/<name>/ with the
# committed file names (copy the directory to profiles/<round>/ afterwards).  usage: tools/profile_all.sh r05     (~6 GPU-minutes)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-rXX}
OUT=$REPO/gpurun_out/$N
mkdir -p "$OUT"
cd "$REPO"
tools/profile_round.sh ${N}_round > /dev/null 2>&1
tools/profile_insts.sh ${N}_insts > /dev/null 2>&1
tools/profile_timeline.sh ${N}_tl > /dev/null 2>&1
tools/profile_timeline.sh ${N}_tlbf "--dtype" "bf16" > /dev/null 2>&1
tools/profile_c2_timeline.sh ${N}_c2tl > /dev/null 2>&1
tools/profile_steps.sh ${N}_steps > /dev/null 2>&1
tools/profile_roi_nms.sh ${N}_roi > /dev/null 2>&1
python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-aux --no-cpu-baseline > "$OUT/bench_bf16.json" 2>/dev/null
G=$REPO/gpurun_out
cp $G/${N}_round/bench_default.json $G/${N}_round/bench_sustained_2000.json $G/${N}_round/bench_kernel_stats.csv $G/${N}_round/bench_under_rocprof.json \
   $G/${N}_round/summary.md $G/${N}_round/traffic.json "$OUT/" 2>/dev/null
cp $G/${N}_round/sq_counters_table.md "$OUT/sq_counters.md" 2>/dev/null
cp $G/${N}_insts/insts.md "$OUT/insts_per_kernel.md" 2>/dev/null
cp $G/${N}_tl/timeline.txt "$OUT/timeline_headline_step.txt" 2>/dev/null
cp $G/${N}_tl/timeline_dump.txt "$OUT/timeline_headline_step_kernels.txt" 2>/dev/null
cp $G/${N}_tlbf/timeline.txt "$OUT/timeline_bf16_backbone_step.txt" 2>/dev/null
cp $G/${N}_tlbf/timeline_dump.txt "$OUT/timeline_bf16_backbone_step_kernels.txt" 2>/dev/null
cp $G/${N}_c2tl/timeline.txt "$OUT/timeline_config2_step.txt" 2>/dev/null
for f in config2_step.json config2_step_kernel_stats.csv config2_step_summary.md config4_step.json config4_step_kernel_stats.csv \
         config4_step_summary.md bench_bf16_kernel_stats.csv bench_bf16_summary.md bench_bf16_under_rocprof.json bench_bf16_sq_counters.md; do
  cp $G/${N}_steps/$f "$OUT/" 2>/dev/null
done
cp $G/${N}_roi/roi_nms_kernel_stats.csv $G/${N}_roi/roi_nms_sq_counters.md $G/${N}_roi/roi_nms_summary.md "$OUT/" 2>/dev/null
rm -rf $G/${N}_round $G/${N}_insts $G/${N}_tl $G/${N}_tlbf $G/${N}_c2tl $G/${N}_steps $G/${N}_roi
ls -la "$OUT"

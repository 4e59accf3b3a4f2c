#!/bin/bash
# rocprofv3 kernel tables of the steps behind the `aux` numbers of bench.py: the configs[2] single-pathway training step, the
# bf16 backbone (kernel stats + SQ counters) and the configs[4] dual-pathway bf16 + DeformConv3d step.
# usage: tools/profile_steps.sh <out-name>      (run through gpurun; outputs under gpurun_out/<out-name>/)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-steps}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T="$REPO/tools/two_stage_bench.py"
B="$REPO/bench.py"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c2" -o c2 -- python3 "$T" --steps 10 --warmup 3 --json > "$OUT/config2_step.json" 2> "$OUT/c2.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c4" -o c4 -- python3 "$T" --steps 3 --warmup 2 --json --two-scales --bf16 --dcn-neck > "$OUT/config4_step.json" 2> "$OUT/c4.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/bf" -o bf -- python3 "$B" --dtype bf16 --steps 20 --warmup 5 --no-aux --no-cpu-baseline > "$OUT/bench_bf16_under_rocprof.json" 2> "$OUT/bf.err"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/bfsq" -o sq -- python3 "$B" --dtype bf16 --steps 2 --warmup 2 --no-aux --no-cpu-baseline > /dev/null 2> "$OUT/bfsq.err"
cd "$REPO"
for tag in c2 c4 bf; do
  KS=$(find "$OUT/$tag" -name '*kernel_stats.csv' | head -1)
  case $tag in c2) N=config2_step;; c4) N=config4_step;; bf) N=bench_bf16;; esac
  cp "$KS" "$OUT/${N}_kernel_stats.csv"
  python3 tools/summarize_prof.py "$KS" > "$OUT/${N}_summary.md"
done
python3 tools/summarize_pmc.py "$(find "$OUT/bfsq" -name '*counter_collection.csv' | head -1)" > "$OUT/bench_bf16_sq_counters.md"
rm -rf "$OUT/c2" "$OUT/c4" "$OUT/bf" "$OUT/bfsq"
ls -la "$OUT"

#!/bin/bash
# kernel statistics + one or more PMC passes of any python tool of this repo (run on the GPU box through gpurun).
#   usage: tools/pmc_cmd.sh <out-name> "<counters pass 1>[;<counters pass 2>...]" tools/some_bench.py [args...]
# The program sits directly behind `--` (python3 itself: no wrapper may exec after the profiler has initialised the GPU).
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$1; PASSES=$2; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 "$REPO/$1" "${@:2}" > "$OUT/run.log" 2> "$OUT/kt.err"
KS=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/kernel_stats.csv"
i=0
IFS=';' read -ra PP <<< "$PASSES"
for P in "${PP[@]}"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $P --kernel-trace -d "$OUT/pmc$i" -o pmc -- python3 "$REPO/$1" "${@:2}" > /dev/null 2> "$OUT/pmc$i.err"
  C=$(find "$OUT/pmc$i" -name '*counter_collection.csv' | head -1)
  [ -n "$C" ] && python3 "$REPO/tools/summarize_pmc.py" "$C" > "$OUT/pmc$i.md"
  rm -rf "$OUT/pmc$i"
done
rm -rf "$OUT/kt"
head -12 "$OUT/kernel_stats.csv"
for f in "$OUT"/pmc*.md; do head -8 "$f"; done

#!/bin/bash
# kernel statistics + FETCH_SIZE of the default bench command, small: usage tools/quick_prof.sh <out-name>
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-quick}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py"
python3 "$B" --steps 20 --warmup 5 --no-aux --no-cpu-baseline > "$OUT/bench.json" 2>/dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 "$B" --steps 20 --warmup 5 --no-aux --no-cpu-baseline > /dev/null 2> "$OUT/kt.err"
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch -- python3 "$B" --steps 2 --warmup 2 --no-aux --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write -- python3 "$B" --steps 2 --warmup 2 --no-aux --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
cd "$REPO"
KS=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)
FC=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
WC=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_prof.py "$KS" "$FC" "$WC" > "$OUT/summary.md"
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write"

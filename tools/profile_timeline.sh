#!/bin/bash
# timeline of one headline step (queues, idle time, exclusive kernels): rocprofv3 kernel trace of bench.py -> tools/timeline.py
# usage: tools/profile_timeline.sh <out-name>     (through gpurun; outputs under gpurun_out/<out-name>/)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-tl}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d "$OUT/tr" -o tl -- python3 "$REPO/bench.py" --steps 6 --warmup 3 --no-aux --no-cpu-baseline ${2:-} ${3:-} > "$OUT/bench.json" 2> "$OUT/err.txt"
cd "$REPO"
KT=$(find "$OUT/tr" -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$KT" -2 > "$OUT/timeline.txt"
python3 tools/timeline.py "$KT" -2 dump > "$OUT/timeline_dump.txt"
rm -rf "$OUT/tr"

#!/bin/bash
# kernel trace of the configs[2] training step -> timeline of one step (tools/timeline.py), incl. the per-kernel dump
# usage: tools/profile_c2_timeline.sh <out-name> [extra two_stage_bench flags]     (through gpurun)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-c2tl}
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d "$OUT/tr" -o c2 -- python3 "$REPO/tools/two_stage_bench.py" --steps 4 --warmup 3 --json "$@" > "$OUT/step.json" 2> "$OUT/err.txt"
cd "$REPO"
KT=$(find "$OUT/tr" -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$KT" -2 > "$OUT/timeline.txt"
python3 tools/timeline.py "$KT" -2 dump > "$OUT/timeline_dump.txt"
rm -rf "$OUT/tr"

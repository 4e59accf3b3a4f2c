#!/bin/bash
# per-launch durations of the kernels whose name contains $2 over a bench.py run (rocprofv3 kernel trace)
# usage: tools/kernel_durations.sh <out-name> <name-substring> [steps] ["extra bench flags"]      (through gpurun)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-kd}
PAT=${2:-stem_wgrad}
STEPS=${3:-30}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d "$OUT/tr" -o kd -- python3 "$REPO/bench.py" --steps "$STEPS" --warmup 3 --no-aux --no-cpu-baseline ${4:-} > "$OUT/bench.json" 2> "$OUT/err.txt"
cd "$REPO"
KT=$(find "$OUT/tr" -name '*kernel_trace.csv' | head -1)
python3 - "$KT" "$PAT" > "$OUT/durations.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print(len(d), 'launches; us:', ' '.join(f'{v:.0f}' for v in d))
PY
rm -rf "$OUT/tr"
cat "$OUT/durations.txt"

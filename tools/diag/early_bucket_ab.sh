#!/bin/bash
# step time with the early gradient bucket (finishing of layer3.. in mid-backward) against ONE finishing at the end (bench.py --tune 8=1, MR3D_TUNE_LATE_GRAD_FINISH)
REPO=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2 3 4; do for e in 0 1; do
  python3 "$REPO/bench.py" --tune 8=$((1 - e)) --steps 60 --warmup 5 --no-aux --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('early_bucket $e', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms')"
done; done

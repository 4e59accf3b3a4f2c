#!/bin/bash
# step time with the early gradient bucket (finishing of layer3.. in mid-backward) against ONE finishing at the end (MR3D_EARLY_BUCKET=0)
REPO=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2 3 4; do for e in 0 1; do
  MR3D_EARLY_BUCKET=$e python3 "$REPO/bench.py" --steps 60 --warmup 5 --no-aux --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('early_bucket $e', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms')"
done; done

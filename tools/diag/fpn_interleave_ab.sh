#!/bin/bash
# A/B of the forward plan's placement of the P3..P5 output convolutions (bench.py --tune 9=1, MR3D_TUNE_FPN_AFTER_LATERALS: behind all laterals, the round 3-4
# order; 1: each right behind its lateral) on one box.  usage (GPU box): tools/diag/fpn_interleave_ab.sh [rounds] [bench.py args]
R=${1:-3}; shift || true
ARGS=${*:---steps 60 --warmup 5 --no-aux --no-cpu-baseline}
for i in $(seq 1 "$R"); do
  for v in 0 1; do
    python3 bench.py --tune 9=$((1 - v)) $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('interleave $v', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms')"
  done
done

#!/bin/bash
# usage: envab.sh VAR rounds [bench args]: A/B of an environment knob (0 / 1) on one box
V=$1; R=${2:-3}; shift; shift || true
ARGS=${*:---steps 60 --warmup 5 --no-aux --no-cpu-baseline}
for i in $(seq 1 "$R"); do
  for v in 0 1; do
    env $V=$v python3 bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$v', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms')"
  done
done

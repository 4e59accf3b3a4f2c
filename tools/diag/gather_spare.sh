#!/bin/bash
# step time against the number of workgroup slots the stem gather kernel leaves free (MR3D_TUNE_STEM_GATHER_SPARE = key 4)
REPO=${GRAFT_REPO_ROOT:-$PWD}
for sp in ${SPARES:--1 32 64 96 128 -1 64 96}; do
  python3 "$REPO/bench.py" --steps 60 --warmup 5 --no-aux --no-cpu-baseline --tune 4=$sp 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('spare $sp', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms')"
done

#!/bin/bash
# A/B of two builds of libmr3d_hip.so on ONE GPU box (boxes differ by +-1 %, run-to-run noise on a box is +-0.2 %):
#   cp 3d-multi-resolution-rcnn_amd/libmr3d_hip.so tools/ab/A_libmr3d_hip.so     (before the change)
#   ... edit, make ...  then through gpurun:  tools/ab.sh [rounds] [bench.py args]
# alternates A (tools/ab/A_libmr3d_hip.so) and B (the in-tree build) and prints iters/s of each run.
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
R=${1:-3}; shift || true
ARGS=${*:---steps 60 --warmup 5 --no-aux --no-cpu-baseline}
for i in $(seq 1 "$R"); do
  for v in A B; do
    if [ $v = A ]; then export MR3D_LIB=$REPO/tools/ab/A_libmr3d_hip.so; else unset MR3D_LIB; fi
    python3 "$REPO/bench.py" $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms')"
  done
done

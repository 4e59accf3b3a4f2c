#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$PWD}
for f in $REPO/tools/ab/rows_abl_*.so; do
  echo "$(basename $f): $(MR3D_LIB=$f MR3D_LB_ONLY='l1 3^3 16->16' python $REPO/tools/layer_bench.py 2>&1 | grep 'l1 3')"
done

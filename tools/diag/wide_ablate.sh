#!/bin/bash
# timing of the variants built by tools/diag/wide_ablate_build.sh (GPU box)
REPO=${GRAFT_REPO_ROOT:-$PWD}
for f in $REPO/tools/ab/wide_abl_*.so; do
  echo "$(basename $f): $(MR3D_LIB=$f MR3D_LB_ONLY='fpn 3^3 64->64 P2' MR3D_LB_TUNE=5=2 python $REPO/tools/layer_bench.py 2>&1 | grep P2)"
done

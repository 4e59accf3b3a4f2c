#!/bin/bash
# layer_bench lines of two library builds on one box: A = tools/ab/A_libmr3d_hip.so, B = the in-tree build.  usage (GPU box):
#   tools/diag/lb_ab.sh '<substring of the layer names>' [key=value tuning]
for i in 1 2; do
for v in A B; do
  if [ $v = A ]; then export MR3D_LIB=$PWD/tools/ab/A_libmr3d_hip.so; else unset MR3D_LIB; fi
  MR3D_LB_ONLY="${1:-3^3}" MR3D_LB_TUNE="${2:-}" python tools/layer_bench.py 2>&1 | grep -v total | grep ms | sed "s/^/$v  /"
done; done

#!/bin/bash
# timing-only variants of libmr3d_hip.so with compile-time ablations of conv3d_wino2d_rows_kernel into tools/ab/rows_abl_<bits>.so
# (bits: 1 no U loads inside the steps, 2 no window reads, 4 no transform); then tools/diag/rows_ablate.sh on the GPU box
set -eu
cd "$(dirname "$0")/../../3d-multi-resolution-rcnn_amd/csrc"
mkdir -p ../../tools/ab
for b in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DMR3D_ROWS_ABL=$b -c conv3d_wino2d.hip -o /tmp/rows_abl_$b.o
  objs=$(ls *.o | grep -v '^conv3d_wino2d.o$')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/rows_abl_$b.so $objs /tmp/rows_abl_$b.o
done

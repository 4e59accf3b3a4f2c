#!/bin/bash
# builds timing-only variants of libmr3d_hip.so with compile-time ablations of conv3d_wino2d_wide_kernel into tools/ab/wide_abl_<bits>.so
# (bits: 1 no V units, 2 no halo requests, 4 no U loads, 8 no epilogue, 16 no barriers; the argument `stamps` builds tools/ab/wide_stamps.so for
# tools/diag/wide_stamps.py); run here, then tools/diag/wide_ablate.sh on the GPU box
set -eu
cd "$(dirname "$0")/../../3d-multi-resolution-rcnn_amd/csrc"
mkdir -p ../../tools/ab
for b in "$@"; do
  if [ "$b" = stamps ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DMR3D_WIDE_STAMPS=1 -c conv3d_wino2d.hip -o /tmp/wide_stamps.o
    objs=$(ls *.o | grep -v '^conv3d_wino2d.o$')
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/wide_stamps.so $objs /tmp/wide_stamps.o
    continue
  fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DMR3D_WIDE_ABL=$b -c conv3d_wino2d.hip -o /tmp/wide_abl_$b.o
  objs=$(ls *.o | grep -v '^conv3d_wino2d.o$')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/wide_abl_$b.so $objs /tmp/wide_abl_$b.o
done

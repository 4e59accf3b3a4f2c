#!/bin/bash
# builds tools/ab/fabl<n>.so: libmr3d_hip.so with timing-only ablations of conv3d_fwd_kernel's small-tile path (MR3D_FWD_ABL bits:
# 1 no MFMAs, 2 no LDS staging writes, 4 no global fetch, 8 no LDS operand reads); results are wrong by construction.  run here
set -eu
cd "$(dirname "$0")/../../3d-multi-resolution-rcnn_amd/csrc"
mkdir -p ../../tools/ab
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DMR3D_FWD_ABL=$n -c conv3d_fwd.hip -o /tmp/fabl$n.o
  objs=$(ls *.o | grep -v '^conv3d_fwd.o$')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/fabl$n.so $objs /tmp/fabl$n.o
done

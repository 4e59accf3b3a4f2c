#!/bin/bash
# builds tools/ab/rows_stamps.so: libmr3d_hip.so with wall-clock stamps in roi_align3d_fwd_rows (tools/diag/roi_rows_stamps.py); run here
set -eu
cd "$(dirname "$0")/../../3d-multi-resolution-rcnn_amd/csrc"
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DMR3D_ROWS_STAMPS=1 -c roi_align3d.hip -o /tmp/rows_stamps.o
objs=$(ls *.o | grep -v '^roi_align3d.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/rows_stamps.so $objs /tmp/rows_stamps.o

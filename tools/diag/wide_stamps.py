#!/usr/bin/env python3
"""Phase stamps of conv3d_wino2d_wide_kernel on the P2 map: a diagnostic build (tools/diag/wide_ablate_build.sh stamps ->
tools/ab/wide_stamps.so) records wall_clock64 at phase borders of every wave's third tile and of its fourth chunk.
usage (GPU box): MR3D_LIB=tools/ab/wide_stamps.so python tools/diag/wide_stamps.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import mr3d_amd
from mr3d_amd import _hip

dev = torch.device('cuda:0')
x = torch.randn((1, 40, 128, 128, 64), device=dev)
w = torch.randn((27, 64, 64), device=dev) * 0.05
y = torch.empty((1, 40, 128, 128, 64), device=dev)
_hip.lib().mr3d_tuning_set(5, 2)
for _ in range(4):
    _hip.conv3d_forward(x, w, (3, 3, 3), (1, 1, 1), (1, 1, 1), out=y)
torch.cuda.synchronize()
buf = np.zeros(1024 * 8 * 16, dtype=np.uint64)
fn = _hip.lib().mr3d_debug_wide_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
raw = buf.reshape(1024, 8, 16)[:256].astype(np.int64)
us = lambda a: a * 10.0 / 1000.0      # 100 MHz
names = {0: 'kernel start', 1: 'tile 3 start', 2: 'chunk 3 start', 3: 'steps 0-1', 4: 'halo wait', 5: 'barrier 1', 6: 'steps 2-4 + units',
         7: 'barrier 2', 8: 'step 5', 9: 'chunks done', 10: 'epilogue done', 11: 'kernel end'}
print('workgroup duration (us): median %.1f  min %.1f  max %.1f' % tuple(f(us(raw[:, 0, 11] - raw[:, 0, 0])) for f in (np.median, np.min, np.max)))
print('tile 3: chunks %.2f us, epilogue %.2f us (median over workgroups, wave 0)' % (np.median(us(raw[:, 0, 9] - raw[:, 0, 1])), np.median(us(raw[:, 0, 10] - raw[:, 0, 9]))))
for k in range(2, 8):
    print('%-20s' % names[k + 1], '  '.join('w%d %.2f' % (wv, np.median(us(raw[:, wv, k + 1] - raw[:, wv, k]))) for wv in range(8)))

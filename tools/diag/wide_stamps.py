#!/usr/bin/env python3
"""Phase stamps of conv3d_wino2d_wide_kernel (MR3D_TUNE_WINO2D_FORM = 2 + 16 * 32: the kernel writes wall_clock64 stamps over y; diagnostic
builds only).  usage (GPU box): python tools/diag/wide_stamps.py"""
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
for _ in range(3):
    _hip.conv3d_forward(x, w, (3, 3, 3), (1, 1, 1), (1, 1, 1), out=y)
_hip.lib().mr3d_tuning_set(5, 2 + 16 * 32)
_hip.conv3d_forward(x, w, (3, 3, 3), (1, 1, 1), (1, 1, 1), out=y)
torch.cuda.synchronize()
raw = y.view(-1).cpu().numpy().view(np.uint64)[:2560 * 8 * 16].reshape(2560, 8, 16)[:, :, :13].astype(np.int64)
t0 = raw[:, :, 0].min()
rel = (raw - t0) * 10.0 / 1000.0      # us (100 MHz)
names = ['start', 'loop', 'c3 start', 'c3 s0+commit', 'c3 s1+fetch', 'c3 barrier1', 'c3 s2+unit', 'c3 s3,4+units', 'c3 barrier2', 'c3 s5', 'loop end', 'Z', 'end']
d = np.diff(raw, axis=2) * 10.0 / 1000.0
print('workgroup start (us): min %.1f median %.1f max %.1f; end: min %.1f median %.1f max %.1f' % (
    rel[:, 0, 0].min(), np.median(rel[:, 0, 0]), rel[:, 0, 0].max(), rel[:, 0, 12].min(), np.median(rel[:, 0, 12]), rel[:, 0, 12].max()))
print('duration of a workgroup: median %.2f us' % np.median(rel[:, 0, 12] - rel[:, 0, 0]))
for k in range(12):
    for wv in (0, 3, 6, 7):
        print('%-14s -> %-14s wave %d: median %.2f us  (p10 %.2f, p90 %.2f)' % (names[k], names[k + 1], wv, np.median(d[:, wv, k]),
              np.percentile(d[:, wv, k], 10), np.percentile(d[:, wv, k], 90)))

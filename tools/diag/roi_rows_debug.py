#!/usr/bin/env python3
"""Where does the row-streaming RoIAlign3D forward disagree with the oracle?  One single-level call on a small map, error pattern by
(roi, pooled depth, row, column, channel).  usage (GPU box): python tools/diag/roi_rows_debug.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import mr3d_amd  # noqa: F401
from mr3d_amd import _hip
from oracle import binding as orc

dev = torch.device('cuda:0')
rng = np.random.default_rng(3)
N, C, D, H, W = 1, 64, 10, 24, 40
feat = rng.normal(size=(N, C, D, H, W)).astype(np.float32)
rois = np.array([[0, 8, 8, 19, 40, 2, 10],
                 [0, 8, 8, 40, 19, 2, 10],
                 [0, 20, 30, 90, 80, 0, 15],
                 [0, 0, 0, 20, 20, 0, 4],
                 [0, 2, 2, 150, 60, 1, 6],
                 [0, -20, -20, 30, 30, -6, 8],
                 [0, 10, 10, 10, 10, 3, 3]], np.float32)
f = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last_3d)
_hip.lib().mr3d_tuning_set(7, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for P in ((3, 7, 7), (10, 14, 14)):
    exp = orc.roi_align3d_forward(feat, rois, *P, 0.25, 0.5, 2)
    out = torch.zeros(exp.shape, device=dev)
    _hip.roi_align3d_forward(f, torch.from_numpy(rois).to(dev), *P, 0.25, 0.5, 2, out)
    got = out.cpu().numpy()
    err = np.abs(got - exp)
    print('shape', P, 'max err', err.max(), 'max |exp|', np.abs(exp).max())
    for r in range(len(rois)):
        e = err[r]
        bad = e > 1e-4
        print(f' roi {r}: bad {bad.sum()} / {bad.size}', end='')
        if bad.any():
            c, pd, ph, pw = np.nonzero(bad)
            print(f'  channels {np.unique(c)[:8]}.. pd {np.unique(pd)} ph {np.unique(ph)} pw {np.unique(pw)}')
            print('   bad by pw', bad.sum((0, 1, 2)), ' by ph', bad.sum((0, 1, 3)), ' by pd', bad.sum((0, 2, 3)), ' by channel block', bad.reshape(4, 16, -1).sum((1, 2)), ' by ch%16', bad.reshape(4, 16, -1).sum((0, 2)))
            i = (c[0], pd[0], ph[0], pw[0])
            print('   e.g.', i, 'got', got[r][i], 'exp', exp[r][i], ' ratio', got[r][i] / (exp[r][i] + 1e-12))
        else:
            print()

#!/usr/bin/env python3
"""RoIAlign3D backward timing by RoI size class (for rocprofv3 / tuning)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mr3d_amd
from mr3d_amd.mmdet_ops import RoIAlign3D
dev = torch.device('cuda:0')
feat = torch.randn(1, 64, 40, 128, 128, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
rng = np.random.default_rng(0)
def rois(n, lo, hi, zlo, zhi):
    s = np.stack([np.exp(rng.uniform(np.log(lo), np.log(hi), n)), np.exp(rng.uniform(np.log(lo), np.log(hi), n)),
                  np.exp(rng.uniform(np.log(zlo), np.log(zhi), n))], 1)
    c = rng.uniform([0, 0, 0], [512, 512, 80], (n, 3))
    a, b = np.clip(c - s / 2, 0, [511, 511, 79]), np.clip(c + s / 2, 0, [511, 511, 79])
    return torch.tensor(np.stack([np.zeros(n), a[:, 0], a[:, 1], b[:, 0], b[:, 1], a[:, 2], b[:, 2]], 1), dtype=torch.float32, device=dev)
def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters
for tag, n, osz, od in (('bbox', 512, 7, 3), ('mask', 128, 14, 10)):
    layer = RoIAlign3D(osz, od, 0.25, 0.5, sample_num=2)
    for cls, (lo, hi, zlo, zhi) in (('small(<=32px)', (8, 32, 4, 16)), ('big(48-128px)', (48, 128, 24, 40)), ('mixed', (4, 128, 2, 40))):
        r = rois(n, lo, hi, zlo, zhi)
        top = torch.randn(n, 64, od, osz, osz, device=dev)
        out = layer(feat, r)
        def bwd():
            feat.grad = None
            out.backward(top, retain_graph=True)
        print(f'{tag:5s} {cls:14s} bwd {timed(bwd):9.1f} us')

#!/usr/bin/env python3
"""3x3x3 64->64 weight gradient on a P2-sized map: the two-dimensional Winograd kernel (conv3d_wgrad_wino2d.hip,
MR3D_WGRAD_WINO=4) against the F(3,2)-along-W kernel (MR3D_WGRAD_WINO=3; read once per process) -- parity vs an fp64 CPU weight
gradient on a ragged volume (with the bias gradient) and time on 40x128x128.
usage: MR3D_WGRAD_WINO=4 python tools/wgrad_wino2d_check.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd import _hip  # noqa: E402

dev = torch.device('cuda:0')
cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()
g = torch.Generator().manual_seed(0)
for n, dhw in ((1, (9, 37, 43)), (2, (8, 32, 40))):
    x = torch.randn((n, 64) + dhw, generator=g)
    w = torch.zeros((64, 64, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    y = F.conv3d(x.double(), w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    ref = _hip.pack_conv_weight(w.grad.float()).double()
    dw = torch.zeros((27, 64, 64), device=dev)
    db = torch.zeros(64, device=dev)
    _hip.conv3d_wgrad(cl(x).to(dev), cl(dy).to(dev), dw, (3, 3, 3), (1, 1, 1), (1, 1, 1), dbias=db)
    err = (dw.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    l2 = ((dw.cpu().double() - ref).norm() / ref.norm()).item()
    refb = dy.double().sum((0, 2, 3, 4))
    errb = (db.cpu().double() - refb).abs().max().item() / refb.abs().max().item()
    print(f'n={n} {dhw}: dw max err / max |ref| = {err:.3e}   L2 rel = {l2:.3e}   dbias rel = {errb:.3e}')
x = torch.randn(1, 40, 128, 128, 64, device=dev)
dy = torch.randn(1, 40, 128, 128, 64, device=dev)
dw = torch.zeros((27, 64, 64), device=dev)
db = torch.zeros(64, device=dev)
for _ in range(3):
    _hip.conv3d_wgrad(x, dy, dw, (3, 3, 3), (1, 1, 1), (1, 1, 1), dbias=db)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    _hip.conv3d_wgrad(x, dy, dw, (3, 3, 3), (1, 1, 1), (1, 1, 1), dbias=db)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f'40x128x128 64->64 weight gradient: {ms * 1e3:.1f} us = {2 * 27 * 64 * 64 * 40 * 128 * 128 / ms * 1e-9:.1f} TFLOP/s algorithmic '
      f'(MR3D_WGRAD_WINO={os.environ.get("MR3D_WGRAD_WINO", "default")})')

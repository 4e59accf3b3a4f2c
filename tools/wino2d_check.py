#!/usr/bin/env python3
"""3x3x3 64->64 convolution on a P2-sized map: the two-dimensional Winograd kernel (conv3d_wino2d.hip, MR3D_FWD_WINO=4)
against the F(4,3) kernel (MR3D_FWD_WINO=3, read once per process: run twice) -- parity vs an fp64 CPU convolution on a small
volume (with the fused epilogue) and time on 40x128x128.  usage: MR3D_FWD_WINO=4 python tools/wino2d_check.py"""
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
for dhw in ((9, 37, 43), (16, 64, 64)):
    x = torch.randn((1, 64) + dhw, generator=g)
    w = torch.randn((64, 64, 3, 3, 3), generator=g) / np.sqrt(64 * 27)
    shift = torch.randn(64, generator=g)
    add1 = torch.randn((1, 64) + dhw, generator=g)
    ref = torch.relu(F.conv3d(x.double(), w.double(), shift.double(), 1, 1) + add1.double())
    y = _hip.conv3d_forward(cl(x).to(dev), _hip.pack_conv_weight(w).to(dev), (3, 3, 3), (1, 1, 1), (1, 1, 1),
                            shift=shift.to(dev), add1=cl(add1).to(dev), relu=True)
    err = (y.permute(0, 4, 1, 2, 3).cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    l2 = ((y.permute(0, 4, 1, 2, 3).cpu().double() - ref).norm() / ref.norm()).item()
    print(f'{dhw}: max err / max |ref| = {err:.3e}   L2 rel = {l2:.3e}')
x = torch.randn(1, 40, 128, 128, 64, device=dev)
wp = _hip.pack_conv_weight(torch.randn(64, 64, 3, 3, 3) / np.sqrt(64 * 27)).to(dev)
sh = torch.zeros(64, device=dev)
for _ in range(3):
    y = _hip.conv3d_forward(x, wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=sh, relu=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    y = _hip.conv3d_forward(x, wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=sh, relu=True)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print(f'40x128x128 64->64: {ms * 1e3:.1f} us = {2 * 27 * 64 * 64 * 40 * 128 * 128 / ms * 1e-9:.1f} TFLOP/s algorithmic (MR3D_FWD_WINO={os.environ.get("MR3D_FWD_WINO", "3")})')

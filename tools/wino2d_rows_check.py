#!/usr/bin/env python3
"""3x3x3 16->16 (layer1) and 32->32 (layer2) convolutions: parity of the dispatched kernel against an fp64 CPU convolution on
ragged volumes with the fused epilogue, and time on the 512-volume map sizes.  MR3D_FWD_WINO=3 selects the previous kernels
(read once per process).  usage: python tools/wino2d_rows_check.py"""
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
for c, dhw in ((16, (19, 67, 75)), (32, (13, 61, 70)), (16, (40, 64, 64))):
    x = torch.randn((1, c) + dhw, generator=g)
    w = torch.randn((c, c, 3, 3, 3), generator=g) / np.sqrt(c * 27)
    shift = torch.randn(c, generator=g)
    add1 = torch.randn((1, c) + dhw, generator=g)
    mask = torch.randn((1, c) + dhw, generator=g)
    conv = F.conv3d(x.double(), w.double(), shift.double(), 1, 1)
    for tag, ref, kw in (('relu', torch.relu(conv + add1.double()), dict(add1=cl(add1).to(dev), relu=True)),
                         ('mask', (conv + add1.double()) * (mask > 0), dict(add1=cl(add1).to(dev), mask=cl(mask).to(dev)))):
        y = _hip.conv3d_forward(cl(x).to(dev), _hip.pack_conv_weight(w).to(dev), (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift.to(dev), **kw)
        got = y.permute(0, 4, 1, 2, 3).cpu().double()
        print(f'{c}->{c} {dhw} {tag}: max err / max |ref| = {(got - ref).abs().max().item() / ref.abs().max().item():.3e}   '
              f'L2 rel = {((got - ref).norm() / ref.norm()).item():.3e}')
for c, dhw in ((16, (40, 128, 128)), (32, (20, 64, 64))):
    x = torch.randn((1,) + dhw + (c,), device=dev)
    wp = _hip.pack_conv_weight(torch.randn(c, c, 3, 3, 3) / np.sqrt(c * 27)).to(dev)
    sh = torch.zeros(c, device=dev)
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
    fl = 2 * 27 * c * c * dhw[0] * dhw[1] * dhw[2]
    print(f'{dhw} {c}->{c}: {ms * 1e3:.1f} us = {fl / ms * 1e-9:.1f} TFLOP/s algorithmic (MR3D_FWD_WINO={os.environ.get("MR3D_FWD_WINO", "4")})')

#!/usr/bin/env python3
"""time the pieces of ConvTranspose3d(k=2,s=2) 64->64 on 128 rois of 10x14x14 (mask head `upsample`)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd
from mr3d_amd import _hip
dev = torch.device('cuda:0')
N, D, H, W, C = 128, 10, 14, 14, 64
x = torch.randn(N, D, H, W, C, device=dev)
g = torch.randn(N, 2 * D, 2 * H, 2 * W, C, device=dev)
wf = torch.randn(8, C, C, device=dev) * 0.1
y = torch.empty_like(g)
def timed(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
def fwd():
    for t in range(8):
        a, b, c = t >> 2, (t >> 1) & 1, t & 1
        _hip.conv3d_forward(x, wf[t:t + 1], (1, 1, 1), (1, 1, 1), (0, 0, 0), out=y, out_stride=(2, 2, 2),
                            out_offset=((a * 2 * H + b) * 2 * W + c) * C)
dwp = torch.zeros(8, C, C, device=dev)
fl = 2.0 * N * D * H * W * C * C * 8
for name, fn in (('fwd (8 scatter GEMMs)', fwd),
                 ('dgrad (k2s2 conv)', lambda: _hip.conv3d_forward(g, wf, (2, 2, 2), (2, 2, 2), (0, 0, 0))),
                 ('wgrad (k2s2)', lambda: _hip.conv3d_wgrad(g, x, dwp, (2, 2, 2), (2, 2, 2), (0, 0, 0))),
                 ('relu mask mul', lambda: g * (y > 0)),
                 ('bias sum', lambda: g.sum((0, 1, 2, 3)))):
    us = timed(fn)
    print(f'{name:26s} {us:8.1f} us   {fl / us / 1e6:7.1f} TFLOP/s-equiv')

#!/usr/bin/env python3
"""3x3x3 64->64 convolution on roi batches of 10x14x14 volumes (the mask head's conv shape) for 16..128 rois.
usage (GPU box): python tools/roi_conv_bench.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd
from mr3d_amd import _hip
dev = torch.device('cuda:0')
for n in (16, 32, 64, 128):
    x = torch.randn(n, 10, 14, 14, 64, device=dev)
    w = torch.randn(27, 64, 64, device=dev) * 0.02
    f = lambda: _hip.conv3d_forward(x, w, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f'rois {n:4d}: {ms * 1e3:8.1f} us  {2 * 27 * 64 * 64 * n * 1960 / ms / 1e9:7.1f} TF')

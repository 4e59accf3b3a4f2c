#!/usr/bin/env python3
"""bf16 3x3x3 weight gradient on the backbone's shapes: time per launch (HIP events).  usage: python tools/wgrad_bf16_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd import _hip  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for (c, d, h, w) in ((64, 20, 128, 128), (16, 20, 128, 128), (32, 10, 64, 64), (64, 5, 32, 32), (128, 3, 16, 16)):
        x = torch.randn(1, d, h, w, c, device=dev).bfloat16()
        dy = torch.randn(1, d, h, w, c, device=dev).bfloat16()
        dw = torch.zeros(27, c, c, device=dev)
        db = torch.zeros(c, device=dev)
        for _ in range(3):
            _hip.conv3d_wgrad_bf16(x, dy, dw, (3, 3, 3), (1, 1, 1), (1, 1, 1), dbias=db)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            _hip.conv3d_wgrad_bf16(x, dy, dw, (3, 3, 3), (1, 1, 1), (1, 1, 1), dbias=db)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * 27 * c * c * d * h * w
        print(f'{c:4d} -> {c:4d} @ {d}x{h}x{w}: {ms * 1e3:8.1f} us  {fl / (ms * 1e-3) / 1e12:7.1f} TFLOP/s')


if __name__ == '__main__':
    main()

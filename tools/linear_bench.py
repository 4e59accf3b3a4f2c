#!/usr/bin/env python3
"""FC layers of the bbox head: the hand-written GEMM (csrc/linear.hip) against torch.addmm (hipBLASLt), forward + backward.
usage: python tools/linear_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd.detector import layers  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for rows, k, o in ((512, 9408, 1024), (512, 1024, 1024), (512, 1024, 12)):
        x = torch.randn(rows, k, device=dev, requires_grad=True)
        w = (torch.randn(o, k, device=dev) / k ** 0.5).requires_grad_(True)
        b = torch.randn(o, device=dev, requires_grad=True)
        gy = torch.randn(rows, o, device=dev)
        for impl in ('hip', 'blas'):
            for _ in range(3):
                layers.linear(x, w, b, relu=True, impl=impl).backward(gy)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
            n = 20
            for _ in range(n):
                x.grad = w.grad = b.grad = None
                ev[0].record()
                y = layers.linear(x, w, b, relu=True, impl=impl)
                ev[1].record()
                y.backward(gy)
                ev[2].record()
                torch.cuda.synchronize()
                tf += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
            fl = 2.0 * rows * k * o
            print(f'{rows} x {k} -> {o}  {impl:5s} forward {tf / n * 1e3:7.1f} us ({fl / (tf / n * 1e-3) / 1e12:5.1f} TF)   '
                  f'backward {tb / n * 1e3:7.1f} us ({2 * fl / (tb / n * 1e-3) / 1e12:5.1f} TF)')


if __name__ == '__main__':
    main()

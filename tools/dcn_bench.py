#!/usr/bin/env python3
"""DeformConv3d 64 -> 64, 3x3x3 at a pyramid-level shape: forward / backward time of the fused kernels
(csrc/deform_conv3d_fused.hip) and of the column form (MR3D_DCN_FUSED=0), with non-zero offsets.
usage: python tools/dcn_bench.py [--vol 20,128,128] [--offset-std 0.5] [--steps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd.mmdet_ops import DeformConv3d  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vol', default='20,128,128')
    ap.add_argument('--offset-std', type=float, default=0.5)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--skip-columns', action='store_true')
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.vol.split(','))
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    layer = DeformConv3d(64, 64, 3, padding=1).to(dev)
    x = torch.randn(1, 64, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    off = (torch.randn(1, 81, D, H, W, device=dev) * a.offset_std).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    gy = torch.randn(1, 64, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    res = {}
    for mode in (['1'] if a.skip_columns else ['1', '0']):
        os.environ['MR3D_DCN_FUSED'] = mode
        for _ in range(2):
            y = layer(x, off)
            y.backward(gy)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(a.steps):
            x.grad = off.grad = layer.weight.grad = None
            ev[0].record()
            y = layer(x, off)
            ev[1].record()
            y.backward(gy)
            ev[2].record()
            torch.cuda.synchronize()
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
        res[mode] = (y.detach().clone(), x.grad.clone(), off.grad.clone(), layer.weight.grad.clone())
        print(f'{"fused" if mode == "1" else "columns"}: forward {tf / a.steps:8.3f} ms   backward {tb / a.steps:8.3f} ms   '
              f'({D}x{H}x{W}, offset std {a.offset_std})')
    if len(res) == 2:
        for n, u, v in zip(('y', 'dx', 'doffset', 'dw'), res['1'], res['0']):
            print(f'  fused vs columns {n}: max rel {float((u - v).abs().max() / v.abs().max()):.2e}')


if __name__ == '__main__':
    main()

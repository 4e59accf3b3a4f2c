#!/usr/bin/env python3
"""DeformConv3d 64 -> 64, 3x3x3 at a pyramid-level shape with non-zero offsets: forward / backward time of the fp32 column-free
kernels (csrc/deform_conv3d_fused.hip) and of the bf16 LDS-box kernels (csrc/deform_conv3d_bf16.hip).  Per-kernel times: run it
under `rocprofv3 --kernel-trace --stats` (tools/profile_dcn.sh).
usage: python tools/dcn_bench.py [--vol 40,128,128] [--offset-std 0.5] [--steps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd.mmdet_ops import DeformConv3d  # noqa: E402
from mr3d_amd.mmdet_ops.dcn import deform_conv3d_bf16  # noqa: E402


def timed(fwd, bwd, steps):
    for _ in range(2):
        bwd(fwd())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(steps):
        ev[0].record()
        y = fwd()
        ev[1].record()
        bwd(y)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    return tf / steps, tb / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vol', default='40,128,128')
    ap.add_argument('--offset-std', type=float, default=0.5)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--dy-density', type=float, default=1.0, help='fraction of 4x8x8 position blocks with non-zero dy (training: ~0.1)')
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.vol.split(','))
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    layer = DeformConv3d(64, 64, 3, padding=1).to(dev)
    x = torch.randn(1, 64, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    off = (torch.randn(1, 81, D, H, W, device=dev) * a.offset_std).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    gy = torch.randn(1, 64, D, H, W, device=dev)
    if a.dy_density < 1.0:
        keep = (torch.rand(1, 1, -(-D // 4), -(-H // 8), -(-W // 8), device=dev) < a.dy_density).float()
        gy = gy * keep.repeat_interleave(4, 2).repeat_interleave(8, 3).repeat_interleave(8, 4)[:, :, :D, :H, :W]
    gy = gy.contiguous(memory_format=torch.channels_last_3d)
    tf, tb = timed(lambda: layer(x, off), lambda y: y.backward(gy), a.steps)
    print(f'fp32 column-free: forward {tf:8.3f} ms   backward {tb:8.3f} ms   ({D}x{H}x{W}, offset std {a.offset_std})')
    x16 = x.detach().permute(0, 2, 3, 4, 1).contiguous().to(torch.bfloat16).requires_grad_(True)
    off16 = torch.zeros(1, D, H, W, 96, dtype=torch.bfloat16, device=dev)
    off16[..., :81] = off.detach().permute(0, 2, 3, 4, 1)
    off16.requires_grad_(True)
    g16 = gy.permute(0, 2, 3, 4, 1).contiguous().to(torch.bfloat16)
    w = layer.weight.detach().clone().requires_grad_(True)
    tf, tb = timed(lambda: deform_conv3d_bf16(x16, off16, w, 1, 1, 1), lambda y: y.backward(g16), a.steps)
    print(f'bf16 LDS box    : forward {tf:8.3f} ms   backward {tb:8.3f} ms')


if __name__ == '__main__':
    main()

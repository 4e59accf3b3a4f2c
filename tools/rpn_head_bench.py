#!/usr/bin/env python3
"""RPN head alone (rpn_head_3d.py: shared 3x3x3 conv + ReLU + 1x1x1 cls / reg on P2..P6 of a 512x512x80 volume): forward and
backward time of the plan runs, for rocprofv3 / tuning.  usage: python tools/rpn_head_bench.py [--bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd.detector.heads import RPNHead3D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bf16', action='store_true')
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    head = RPNHead3D().to(dev)
    dt = torch.bfloat16 if a.bf16 else torch.float32
    feats = [torch.randn(1, 64, d, h, h, device=dev).contiguous(memory_format=torch.channels_last_3d).to(dt).requires_grad_(True)
             for d, h in ((40, 128), (20, 64), (10, 32), (5, 16), (3, 8))]
    if a.bf16:
        head.bf16 = True
    outs = head(feats)
    gs = [[torch.randn_like(o) for o in lvl] for lvl in outs]
    flat_o = [o for lvl in outs for o in lvl]
    flat_g = [g for lvl in gs for g in lvl]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(a.steps + 2):
        for f in feats:
            f.grad = None
        head.zero_grad()
        ev[0].record()
        outs = head(feats)
        ev[1].record()
        torch.autograd.backward([o for lvl in outs for o in lvl], flat_g)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    print(f'rpn head: forward {tf / a.steps:.3f} ms   backward {tb / a.steps:.3f} ms')


if __name__ == '__main__':
    main()

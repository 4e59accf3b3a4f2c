#!/usr/bin/env python3
"""Per-layer timing of the conv3d kernels on the 512-volume shapes.
usage (GPU box): python tools/layer_bench.py [fwd|wgrad]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mr3d_amd
from mr3d_amd import _hip

dev = torch.device('cuda:0')
# (name, Cin, Cout, k, stride, pad, (D,H,W) input)
LAYERS = [
    ('stem7 4->16', 4, 16, 7, (1, 2, 2), 3, (80, 512, 512)),
    ('l1 1x1 16->16', 16, 16, 1, (1, 1, 1), 0, (40, 128, 128)),
    ('l1 3^3 16->16', 16, 16, 3, (1, 1, 1), 1, (40, 128, 128)),
    ('l1 1x1 16->64', 16, 64, 1, (1, 1, 1), 0, (40, 128, 128)),
    ('l1 1x1 64->16', 64, 16, 1, (1, 1, 1), 0, (40, 128, 128)),
    ('l2 1x1 64->32', 64, 32, 1, (1, 1, 1), 0, (40, 128, 128)),
    ('l2 3^3s2 32->32', 32, 32, 3, (2, 2, 2), 1, (40, 128, 128)),
    ('l2 1x1s2 64->128', 64, 128, 1, (2, 2, 2), 0, (40, 128, 128)),
    ('l2 3^3 32->32', 32, 32, 3, (1, 1, 1), 1, (20, 64, 64)),
    ('l2 1x1 32->128', 32, 128, 1, (1, 1, 1), 0, (20, 64, 64)),
    ('l2 1x1 128->32', 128, 32, 1, (1, 1, 1), 0, (20, 64, 64)),
    ('l3 3^3 64->64', 64, 64, 3, (1, 1, 1), 1, (10, 32, 32)),
    ('l3 1x1 64->256', 64, 256, 1, (1, 1, 1), 0, (10, 32, 32)),
    ('l3 1x1 64->256 big', 64, 256, 1, (1, 1, 1), 0, (15, 48, 48)),          # the 768 x 768 x 120 pathway of the shipped model
    ('l3 1x1 256->64 big', 256, 64, 1, (1, 1, 1), 0, (15, 48, 48)),
    ('l4 1x1 128->256 big', 128, 256, 1, (1, 1, 1), 0, (15, 48, 48)),
    ('l3 3^3 64->64 big', 64, 64, 3, (1, 1, 1), 1, (15, 48, 48)),
    ('l3 1x1 256->64', 256, 64, 1, (1, 1, 1), 0, (10, 32, 32)),
    ('l4 3^3 128->128', 128, 128, 3, (1, 1, 1), 1, (5, 16, 16)),
    ('l4 1x1 128->512', 128, 512, 1, (1, 1, 1), 0, (5, 16, 16)),
    ('fpn lat 64->64', 64, 64, 1, (1, 1, 1), 0, (40, 128, 128)),
    ('fpn 3^3 64->64 P2', 64, 64, 3, (1, 1, 1), 1, (40, 128, 128)),
    ('fpn 3^3 64->64 P3', 64, 64, 3, (1, 1, 1), 1, (20, 64, 64)),
]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
    for kv in filter(None, os.environ.get('MR3D_LB_TUNE', '').split(',')):       # key=value[,...] for mr3d_tuning_set
        _hip.lib().mr3d_tuning_set(*(int(v) for v in kv.split('=')))
    keep = os.environ.get('MR3D_LB_ONLY', '')                                     # substring of the layer names to run
    tot_t = tot_f = 0
    for name, ci, co, k, st, pad, dhw in LAYERS:
        if keep and keep not in name:
            continue
        x = torch.randn((1,) + dhw + (ci,), device=dev)
        w = torch.randn((k ** 3, ci, co), device=dev) * 0.05
        oshape = _hip.conv3d_out_shape(x.shape, co, (k,) * 3, st, (pad,) * 3)
        npos = oshape[1] * oshape[2] * oshape[3]
        flops = 2.0 * npos * co * ci * k ** 3
        if ci == 4:
            flops *= 3 / 4      # algorithmic FLOPs of the 3-channel stem
        byts = 4.0 * (x.numel() + npos * co)
        if only == 'fwd':
            y = torch.empty(oshape, device=dev)
            ms = timeit(lambda: _hip.conv3d_forward(x, w, (k,) * 3, st, (pad,) * 3, out=y))
        else:
            dy = torch.randn(oshape, device=dev)
            dw = torch.zeros_like(w)
            ws = _hip.wgrad_workspace(dev) if os.environ.get('MR3D_LB_WS') == '1' else None     # two-stage reduction A/B
            ms = timeit(lambda: _hip.conv3d_wgrad(x, dy, dw, (k,) * 3, st, (pad,) * 3, workspace=ws))
        tot_t += ms; tot_f += flops
        print(f'{name:22s} {ms:8.3f} ms  {flops / ms / 1e9:8.2f} TFLOP/s  {byts / ms / 1e6:8.1f} GB/s(min traffic)')
    print(f'total {tot_t:.3f} ms, {tot_f / tot_t / 1e9:.2f} TFLOP/s')


if __name__ == '__main__':
    main()

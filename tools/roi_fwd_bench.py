#!/usr/bin/env python3
"""RoIAlign3D forward through SingleRoIExtractor over P2..P5 on the SURVEY 8(d) inputs (bench.py's roi cases): time and
fraction of the HBM roof of the LB bytes.  
usage: python tools/roi_fwd_bench.py [--bf16]"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mr3d_amd  # noqa: E402,F401
from mr3d_amd.detector import SingleRoIExtractor  # noqa: E402

spec = importlib.util.spec_from_file_location('mr3d_bench', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bf16', action='store_true')
    ap.add_argument('--backward', action='store_true', help='also time forward + backward (fp32 features)')
    ap.add_argument('--tune', default='', help='key=value pairs for mr3d_tuning_set, e.g. 4=1 (box forward without the RoI split)')
    ap.add_argument('--nrois', type=int, default=0, help='use only the first n rois of each case')
    ap.add_argument('--sort', default='', help="'desc' / 'asc': hand the rois over ordered by the voxel area of their footprint at their own level")
    ap.add_argument('--sizes', default='', help='xy_lo,xy_hi,z_lo,z_hi in pixels: log-uniform roi sizes instead of the SURVEY 8(d) ranges')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    from mr3d_amd import _hip as _h
    for kv in filter(None, a.tune.split(',')):
        _h.lib().mr3d_tuning_set(*(int(v) for v in kv.split('=')))
    feats = [torch.randn(1, 64, d, h, h, device=dev).contiguous(memory_format=torch.channels_last_3d)
             for d, h in ((40, 128), (20, 64), (10, 32), (5, 16))]
    if a.bf16:
        feats = [f.to(torch.bfloat16) for f in feats]
    rng = np.random.default_rng(0)
    for tag, n, osz, od in (('bbox', 512, 7, 3), ('mask', 128, 14, 10)):
        ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=osz, out_size_depth=od, sample_num=2), 64,
                                 [4, 8, 16, 32], [2, 4, 8, 16])
        r_host = bench.roi_bench_rois(rng, n)
        if a.sizes:
            lo, hi, zlo, zhi = (float(v) for v in a.sizes.split(','))
            sz = np.exp(rng.uniform(np.log([lo, lo, zlo]), np.log([hi, hi, zhi]), (n, 3)))
            c = rng.uniform([0, 0, 0], [512, 512, 80], (n, 3))
            a0, a1 = np.clip(c - sz / 2, 0, [511, 511, 79]), np.clip(c + sz / 2, 0, [511, 511, 79])
            r_host = np.stack([np.zeros(n), a0[:, 0], a0[:, 1], a1[:, 0], a1[:, 1], a0[:, 2], a1[:, 2]], 1).astype(np.float32)
        if a.nrois:
            r_host = r_host[:a.nrois]
            n = len(r_host)
        if a.sort:
            wv, hv, dv = r_host[:, 3] - r_host[:, 1] + 1, r_host[:, 4] - r_host[:, 2] + 1, r_host[:, 6] - r_host[:, 5] + 1
            lvl = np.clip(np.floor(np.log2(np.sqrt(wv * hv * dv) / 56 + 1e-6)), 0, 3)
            area = (wv / (4 * 2 ** lvl) + 2) * (hv / (4 * 2 ** lvl) + 2)
            order = np.argsort(-area if a.sort == 'desc' else area, kind='stable')
            r_host = np.ascontiguousarray(r_host[order])
        r = torch.from_numpy(r_host).to(dev)
        us = timed(lambda: ext(feats, r))
        lb, wr, rd = bench.roialign_lb_bytes(r_host, 64, od, osz, osz)
        if a.backward and not a.bf16:
            fg = [f.clone().requires_grad_(True) for f in feats]
            top = torch.randn(n, 64, od, osz, osz, device=dev)

            def fb():
                for f in fg:
                    f.grad = None
                ext(fg, r).backward(top)
            fbus = timed(fb, iters=10)
            from mr3d_amd import _hip
            grads = [torch.zeros_like(f) for f in feats]
            bus = timed(lambda: _hip.roi_align3d_backward_ml(top, r, grads, od, osz, osz, [1 / 4, 1 / 8, 1 / 16, 1 / 32],
                                                             [1 / 2, 1 / 4, 1 / 8, 1 / 16], 2, 56.0))
            lb_f, wr_f, rd_f = bench.roialign_lb_bytes(r_host, 64, od, osz, osz)
            print(f'{tag}: forward + backward through autograd {fbus:8.1f} us; backward launch alone {bus:8.1f} us = '
                  f'{(wr_f + 2 * rd_f) / bus * 1e-6 / 8.0:5.3f} of the HBM roof of its {(wr_f + 2 * rd_f) / 1e6:.1f} MB')
        print(f'{tag}: {us:8.1f} us   LB {lb / 1e6:6.1f} MB   {lb / us * 1e-6:5.2f} TB/s   {lb / us * 1e-6 / 8.0:5.3f} of the HBM roof'
              f'   checksum {float(ext(feats, r).double().sum()):.6f}')


if __name__ == '__main__':
    main()

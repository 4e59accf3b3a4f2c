#!/usr/bin/env python3
"""A/B of RoIAlign3D kernel variants through the measurement knobs (mr3d_tuning_set) on bench.py's own inputs.
usage: python3 tools/roi_ab.py [key=value,...]  e.g.  tools/roi_ab.py 1=0"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mr3d_amd  # noqa: E402,F401
from mr3d_amd import _hip  # noqa: E402
from mr3d_amd.detector import SingleRoIExtractor  # noqa: E402

spec = importlib.util.spec_from_file_location('mr3d_bench', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def timed(fn, iters=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    settings = [tuple(int(v) for v in a.split('=')) for a in sys.argv[1:]] or [(0, 0)]
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    feats = [torch.randn(1, 64, d, h, h, device=dev).contiguous(memory_format=torch.channels_last_3d)
             for d, h in ((40, 128), (20, 64), (10, 32), (5, 16))]
    rng = np.random.default_rng(0)
    L = _hip.lib()
    scales, scales_d = [1.0 / s for s in (4, 8, 16, 32)], [1.0 / s for s in (2, 4, 8, 16)]
    for tag, n, osz, od in (('bbox', 512, 7, 3), ('mask', 128, 14, 10)):
        ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=osz, out_size_depth=od, sample_num=2), 64, [4, 8, 16, 32], [2, 4, 8, 16])
        r = torch.from_numpy(bench.roi_bench_rois(rng, n)).to(dev)
        top = torch.randn(n, 64, od, osz, osz, device=dev)
        grads = [torch.zeros_like(f) for f in feats]
        ref = None
        for key, val in settings:
            L.mr3d_tuning_set(key, val)
            out = ext(feats, r)
            if ref is None:
                ref = out.clone()
            err = (out - ref).abs().max().item()
            us_f = timed(lambda: ext(feats, r))
            us_b = timed(lambda: _hip.roi_align3d_backward_ml(top, r, grads, od, osz, osz, scales, scales_d, 2, 56.0), iters=20)
            print(f'{tag:5s} knob {key}={val:<4d} fwd {us_f:7.1f} us   bwd {us_b:7.1f} us   max |out - first| {err:.2e}', flush=True)
            L.mr3d_tuning_set(key, 0)


if __name__ == '__main__':
    main()

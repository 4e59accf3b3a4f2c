#!/usr/bin/env python3
"""NMS3D timing on the SURVEY 8(d) inputs: one batched call over the 5 level sets of a pathway, one over the 10 sets of both
pathways (what the dual-pathway detector issues), device-count form (no read-back).  usage: python3 tools/nms_bench.py [iters]"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mr3d_amd  # noqa: E402,F401
from mr3d_amd import _hip  # noqa: E402

spec = importlib.util.spec_from_file_location('mr3d_bench', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    one = [d.to(dev) for d in bench.nms_bench_inputs(g)]
    two = one + [d.to(dev) for d in bench.nms_bench_inputs(g)]
    for tag, sets in (('5 sets (one pathway)', one), ('10 sets (both pathways)', two)):
        caps = [int(d.size(0)) for d in sets]
        cat = torch.cat(sets, 0).contiguous()
        us = timed(lambda: _hip.nms3d_batched_dev(cat, caps, None, 0.7), iters)
        print(f'{tag:26s} {us:8.1f} us per call')
    for d in one:
        us = timed(lambda: _hip.nms3d_batched_dev(d, [int(d.size(0))], None, 0.7), iters)
        print(f'single set n={d.size(0):5d}          {us:8.1f} us per call')


if __name__ == '__main__':
    main()

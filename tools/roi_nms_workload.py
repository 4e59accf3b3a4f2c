#!/usr/bin/env python3
"""The second half of BASELINE's metric as ONE small workload for rocprofv3: NMS3D (one batched call over the 10 level sets of both
pathways, SURVEY 8(d) inputs) and RoIAlign3D forward + backward of the bbox (512 x (3,7,7)) and mask (128 x (10,14,14))
extractors over P2..P5 of a 512x512x80 volume -- bench.py's own inputs.  usage: python3 tools/roi_nms_workload.py [iters]"""
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


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    g = torch.Generator(device='cpu').manual_seed(0)
    level_dets = [d.to(dev) for d in bench.nms_bench_inputs(g)] + [d.to(dev) for d in bench.nms_bench_inputs(g)]   # both pathways
    caps = [int(d.size(0)) for d in level_dets]
    cat = torch.cat(level_dets, 0).contiguous()
    feats = [torch.randn(1, 64, d, h, h, device=dev).contiguous(memory_format=torch.channels_last_3d)
             for d, h in ((40, 128), (20, 64), (10, 32), (5, 16))]
    rng = np.random.default_rng(0)
    cases = []
    for tag, n, osz, od in (('bbox', 512, 7, 3), ('mask', 128, 14, 10)):
        ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=osz, out_size_depth=od, sample_num=2), 64,
                                 [4, 8, 16, 32], [2, 4, 8, 16])
        r = torch.from_numpy(bench.roi_bench_rois(rng, n)).to(dev)
        top = torch.randn(n, 64, od, osz, osz, device=dev)
        grads = [torch.zeros_like(f) for f in feats]
        cases.append((ext, r, top, grads, od, osz))
    scales, scales_d = [1.0 / s for s in (4, 8, 16, 32)], [1.0 / s for s in (2, 4, 8, 16)]
    for _ in range(iters):
        _hip.nms3d_batched_dev(cat, caps, None, 0.7)          # the product path: ten sets, one launch sequence, no read-back
        for ext, r, top, grads, od, osz in cases:
            ext(feats, r)
            _hip.roi_align3d_backward_ml(top, r, grads, od, osz, osz, scales, scales_d, 2, 56.0)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()

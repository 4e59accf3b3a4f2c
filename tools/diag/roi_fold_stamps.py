#!/usr/bin/env python3
"""Phase stamps of roi_align3d_fwd_fold on the bench's bbox / mask extractor inputs (diagnostic build tools/diag/roi_rows_build.sh ->
tools/ab/rows_stamps.so; wall_clock64 = 100 MHz).  usage (GPU box): MR3D_LIB=tools/ab/rows_stamps.so python tools/diag/roi_fold_stamps.py"""
import ctypes
import importlib.util
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mr3d_amd  # noqa: F401
from mr3d_amd import _hip
from mr3d_amd.detector import SingleRoIExtractor

spec = importlib.util.spec_from_file_location('mr3d_bench', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
_hip.lib().mr3d_tuning_set(7, 8)
dev = torch.device('cuda:0')
torch.manual_seed(0)
feats = [torch.randn(1, 64, d, h, h, device=dev).contiguous(memory_format=torch.channels_last_3d) for d, h in ((40, 128), (20, 64), (10, 32), (5, 16))]
rng = np.random.default_rng(0)
fn = _hip.lib().mr3d_debug_rows_stamps
fn.argtypes = [ctypes.c_void_p]
us = lambda a: a * 0.01
for tag, n, osz, od in (('bbox', 512, 7, 3), ('mask', 128, 14, 10)):
    ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=osz, out_size_depth=od, sample_num=2), 64, [4, 8, 16, 32], [2, 4, 8, 16])
    r = torch.from_numpy(bench.roi_bench_rois(rng, n)).to(dev)
    for _ in range(5):
        ext(feats, r)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    raw = buf.reshape(8192, 16).astype(np.int64)[:n * od]
    t0 = raw[:, 0].min()
    print(f'== {tag}: {len(raw)} workgroups; launch span {us(raw[:, 9].max() - t0):.1f} us; starts: median {us(np.median(raw[:, 0]) - t0):.1f} 90% {us(np.percentile(raw[:, 0], 90) - t0):.1f} last {us(raw[:, 0].max() - t0):.1f}')
    print(f'   footprint: nx median {np.median(raw[:, 10]):.0f} max {raw[:, 10].max()}  ny median {np.median(raw[:, 11]):.0f} max {raw[:, 11].max()}  planes median {np.median(raw[:, 12]):.0f}')
    names = ['roi + planes', 'samples', 'x table', 'phase A (last pass)', 'barrier', 'phase B', 'barrier + tile', 'stores issued', 'stores drained']
    for k, nm in enumerate(names):
        ok = (raw[:, k + 1] > 0) & (raw[:, k] > 0)
        d = us(raw[ok, k + 1] - raw[ok, k])
        if len(d):
            print(f'  {nm:20s} median {np.median(d):6.2f}  mean {d.mean():6.2f}  90% {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us')
    tot = us(raw[:, 9] - raw[:, 0])
    print(f'  lifetime             median {np.median(tot):6.2f}  mean {tot.mean():6.2f}  90% {np.percentile(tot, 90):6.2f}  max {tot.max():6.2f} us')

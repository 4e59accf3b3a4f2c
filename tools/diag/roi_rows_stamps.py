#!/usr/bin/env python3
"""Phase stamps of roi_align3d_fwd_rows on the bench's bbox / mask extractor inputs: a diagnostic build (tools/diag/roi_rows_build.sh ->
tools/ab/rows_stamps.so) records wall_clock64 (100 MHz) at a unit's phase borders.
usage (GPU box): MR3D_LIB=tools/ab/rows_stamps.so python tools/diag/roi_rows_stamps.py [form]"""
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

form = int(sys.argv[1]) if len(sys.argv) > 1 else 0
abl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_hip.lib().mr3d_tuning_set(7, form)
dev = torch.device('cuda:0')
torch.manual_seed(0)
feats = [torch.randn(1, 64, d, h, h, device=dev).contiguous(memory_format=torch.channels_last_3d) for d, h in ((40, 128), (20, 64), (10, 32), (5, 16))]
rng = np.random.default_rng(0)
fn = _hip.lib().mr3d_debug_rows_stamps
assert _hip.lib().mr3d_debug_rows_ablate(abl) == 0
print('ablation bits', abl)
fn.argtypes = [ctypes.c_void_p]
us = lambda a: a * 0.01
for tag, n, osz, od in (('bbox', 512, 7, 3), ('mask', 128, 14, 10)):
    ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=osz, out_size_depth=od, sample_num=2), 64, [4, 8, 16, 32], [2, 4, 8, 16])
    r = torch.from_numpy(bench.roi_bench_rois(rng, n)).to(dev)
    for _ in range(5):
        ext(feats, r)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    zero = np.zeros(8192 * 16, dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    raw = buf.reshape(8192, 16).astype(np.int64)
    ex = (raw[:, 7] > 0) & (raw[:, 6] == 0)
    if ex.any():
        e = raw[ex]
        print(f'  {ex.sum()} workgroups left at the split test: lifetime median {us(np.median(e[:, 7] - e[:, 0])):.2f} max {us((e[:, 7] - e[:, 0]).max()):.2f} us; their starts: median {us(np.median(e[:, 0]) - raw[raw[:, 0] > 0][:, 0].min()):.1f} last {us(e[:, 0].max() - raw[raw[:, 0] > 0][:, 0].min()):.1f} us')
    live = raw[:, 6] > 0
    s = raw[live]
    t0 = s[:, 0].min()
    st = np.sort(us(s[:, 0] - t0))
    print('  starts of the staying workgroups (us) at 10/25/50/75/90/100 %:', ' '.join(f'{st[int(q * (len(st) - 1))]:.1f}' for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0)))
    print(f'== {tag} form {form}: {live.sum()} workgroups; launch span (first start -> last end) {us(s[:, 6].max() - t0):.1f} us; '
          f'starts spread {us(s[:, 0].max() - t0):.1f} us; ends: median {us(np.median(s[:, 6]) - t0):.1f}  90% {us(np.percentile(s[:, 6], 90) - t0):.1f}')
    names = ['roi + tables', 'x taps', 'first batch', 'rest of rows', 'flush issue', 'store drain']
    for k, nm in enumerate(names):
        d = us(s[:, k + 1] - s[:, k])
        d = d[(s[:, k + 1] > 0) & (s[:, k] > 0)]
        if len(d):
            print(f'  {nm:14s} median {np.median(d):6.2f}  mean {d.mean():6.2f}  90% {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us')
    tot = us(s[:, 6] - s[:, 0])
    print(f'  unit total     median {np.median(tot):6.2f}  mean {tot.mean():6.2f}  90% {np.percentile(tot, 90):6.2f}  max {tot.max():6.2f} us')
    bat, pcs, rpb = s[:, 8], s[:, 9], s[:, 10]
    print(f'  batches/workgroup median {np.median(bat):.0f} mean {bat.mean():.1f} max {bat.max()}; pieces per row median {np.median(pcs):.0f} max {pcs.max()}; rows per batch median {np.median(rpb):.0f}')
    ok = bat > 0
    if ok.sum() > 10:
        st = us(s[ok, 4] - s[ok, 3])
        print(f'  rest-of-rows per batch: {np.median(st / np.maximum(bat[ok], 1)):.2f} us median')

"""RoI extractor forward time by roi size class (development aid: which kernel a class of rois lands in shows as its time)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mr3d_amd
from mr3d_amd import _hip
from mr3d_amd.detector import SingleRoIExtractor
dev = torch.device('cuda:0')
feats = [torch.randn(1, 64, d, h, h, device=dev).contiguous(memory_format=torch.channels_last_3d) for d, h in ((40, 128), (20, 64), (10, 32), (5, 16))]
rng = np.random.default_rng(0)
def rois(n, lo, hi, zlo, zhi):
    s = np.stack([np.exp(rng.uniform(np.log(lo), np.log(hi), n)), np.exp(rng.uniform(np.log(lo), np.log(hi), n)), np.exp(rng.uniform(np.log(zlo), np.log(zhi), n))], 1)
    c = rng.uniform([0, 0, 0], [512, 512, 80], (n, 3))
    a, b = np.clip(c - s / 2, 0, [511, 511, 79]), np.clip(c + s / 2, 0, [511, 511, 79])
    return torch.tensor(np.stack([np.zeros(n), a[:, 0], a[:, 1], b[:, 0], b[:, 1], a[:, 2], b[:, 2]], 1), dtype=torch.float32, device=dev)
def timed(fn, iters=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters
for tag, n, osz, od in (('bbox', 512, 7, 3), ('mask', 128, 14, 10)):
    ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=osz, out_size_depth=od, sample_num=2), 64, [4, 8, 16, 32], [2, 4, 8, 16])
    for cls, (lo, hi, zlo, zhi) in (('tiny 8-16px', (8, 16, 4, 8)), ('small 8-32px', (8, 32, 4, 16)), ('flat 64-128 x 2-4', (64, 128, 2, 4)), ('big 48-128px', (48, 128, 24, 40)), ('mixed', (4, 128, 2, 40))):
        r = rois(n, lo, hi, zlo, zhi)
        print(f'{tag:5s} {cls:20s} {timed(lambda: ext(feats, r)):7.1f} us', flush=True)

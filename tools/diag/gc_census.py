#!/usr/bin/env python3
"""What does CPython's cyclic collector track once the two-stage detector and its plans exist?  (dist_utils.freeze_gc_after_warmup
exists because a generation-2 pass over those objects took tens of ms.)  usage (GPU box): python tools/diag/gc_census.py"""
import collections
import gc
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import torch

gc.collect()
base = len(gc.get_objects())
import mr3d_amd  # noqa: F401
from mr3d_amd.detector import TwoStageDetector3D
from mr3d_amd.dist_utils import ArenaSGD, DetectorArena, train_step
from two_stage_bench import synthetic_gt
dev = torch.device('cuda:0')
gc.collect()
after_import = len(gc.get_objects())
det = TwoStageDetector3D(device=dev)
det.extract.init_weights()
arena = DetectorArena(det)
opt = ArenaSGD(arena, lr=0.001, momentum=0.9, weight_decay=1e-4)
D, H, W = 80, 512, 512
rng = np.random.default_rng(0)
img = torch.randn(1, 3, D, H, W, device=dev)
gt, masks = synthetic_gt(rng, 12, D, H, W, dev)
gl = torch.ones(12, dtype=torch.long, device=dev)
meta = [dict(img_shape=(H, W, 3, D), pad_shape=(H, W, 3, D), ori_shape=(H, W, 3))]
batch = ((img, meta, [gt], [gl]), dict(gt_masks=[masks]))
for _ in range(3):
    train_step(det, arena, opt, batch, 1)
torch.cuda.synchronize()
gc.collect()
objs = gc.get_objects()
print('tracked objects: interpreter start', base, ' after imports', after_import, ' after model + plans + 3 steps', len(objs))
hist = collections.Counter(type(o).__name__ for o in objs)
print(hist.most_common(25))
# where do the tuples / lists / dicts hang?  sample owners by module of the referrer
t0 = time.perf_counter(); gc.collect(); print('full collection: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
big = collections.Counter()
for o in objs:
    if isinstance(o, (list, tuple)) and len(o) > 64:
        big[(type(o).__name__, len(o))] += 1
print('long containers:', big.most_common(10))

"""which source lines issue the small torch ops of one configs[2] training step: a TorchDispatchMode logs every aten op that launches
a kernel (copy_, fill_, add, cat, ...) with its output size and the innermost frame inside this repo; autograd-thread ops show up
under the Function.backward frame that issued them.  usage: python tools/diag/glue_ops.py"""
import os, sys, collections, traceback
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import mr3d_amd  # noqa
from mr3d_amd.detector import TwoStageDetector3D
from two_stage_bench import synthetic_gt
dev = torch.device('cuda:0'); torch.manual_seed(0)
det = TwoStageDetector3D(device=dev); det.extract.init_weights()
rng = np.random.default_rng(0)
D, H, W = 80, 512, 512
img = torch.randn(1, 3, D, H, W, device=dev)
gt, masks = synthetic_gt(rng, 12, D, H, W, dev)
gl = torch.ones(12, dtype=torch.long, device=dev)
meta = [dict(img_shape=(H, W, 3, D), pad_shape=(H, W, 3, D), ori_shape=(H, W, 3))]
def step():
    det.extract.zero_grad()
    return det.forward_backward(img, meta, [gt], [gl], gt_masks=[masks])
for _ in range(3): step()
torch.cuda.synchronize()
VIEW = ('view', 'reshape', 'permute', 'detach', 'alias', 'as_strided', 'slice', 'select', 'expand', 'transpose', 't.', 'unsqueeze', 'squeeze',
        'size', 'stride', 'is_', '_unsafe_view', 'empty', 'unbind', 'split', 'narrow', 'lift_fresh', 'record_stream', 'set_')
agg = collections.defaultdict(lambda: [0, 0])
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(v in name for v in VIEW):
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda:
                fr = [f for f in traceback.extract_stack() if '3d-multi-resolution-rcnn_amd' in f.filename]
                where = f'{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno}' if fr else '?'
                a = agg[(name.replace('aten.', ''), where)]
                a[0] += 1; a[1] += t.numel() * t.element_size()
        return out
with Log():
    step()
torch.cuda.synchronize()
rows = sorted(agg.items(), key=lambda kv: (-kv[1][0], -kv[1][1]))
print('ops', sum(v[0] for v in agg.values()))
for (name, where), (n, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{by / 1e6:9.2f} MB  n={n:3d}  {name:26s} {where}')
print('--- by count')
for (name, where), (n, by) in rows[:40]:
    print(f'n={n:3d} {by / 1e6:9.3f} MB  {name:26s} {where}')

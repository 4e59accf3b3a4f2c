import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import mr3d_amd
from mr3d_amd import _hip
dev = torch.device('cuda:0')
d, h, w = 80, 512, 512
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((1, 3, d, h, w), device=dev, generator=g)
gs = torch.randn((1, d, h // 2, w // 2, 16), device=dev, generator=g).bfloat16()
dw = torch.zeros(343, 4, 16, device=dev); db = torch.zeros(16, device=dev)
L = _hip.lib()
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for tag, m in (('all', 0), ('no mma', 1), ('no commit', 2), ('no fetch', 4), ('no mma, commit', 3), ('no fetch, commit', 6), ('only mma', 6), ('nothing', 7)):
    L.mr3d_tuning_set(0, m)
    print(f'{tag:20s} {timed(lambda: _hip.stem_conv_wgrad_bf16(x, gs, dw, db)):8.1f} us', flush=True)
L.mr3d_tuning_set(0, 0)

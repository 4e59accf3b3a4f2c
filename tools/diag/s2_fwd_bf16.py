"""bf16 stride-2 3x3x3 forward convolutions of layer2.0 / 3.0 / 4.0 (512 volume): time per launch (development aid)."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import mr3d_amd
from mr3d_amd import _hip
dev = torch.device('cuda:0'); torch.manual_seed(0)
def timed(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for c, d, h in ((32, 40, 128), (64, 20, 64), (128, 10, 32)):
    x = torch.randn(1, d, h, h, c, device=dev).bfloat16()
    w16 = _hip.pack_conv_weight_bf16(torch.randn(c, c, 3, 3, 3) / (c * 27) ** 0.5).to(dev)
    sh = torch.randn(c, device=dev)
    t = timed(lambda: _hip.conv3d_forward_bf16(x, w16, c, (3, 3, 3), (2, 2, 2), (1, 1, 1), shift=sh, relu=True))
    print(f'{c:4d} -> {c:4d} stride 2 on {d}x{h}x{h}: {t:7.1f} us', flush=True)

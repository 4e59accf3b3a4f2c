"""bf16 3x3x3 stride-1 forward on the backbone's shapes (the 16 / 32 / 64-channel kernels): time per launch (development aid)."""
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
for c, d, h in ((16, 40, 128), (32, 20, 64), (64, 40, 128), (64, 20, 64), (64, 10, 32), (128, 5, 16)):
    x = torch.randn(1, d, h, h, c, device=dev).bfloat16()
    w16 = _hip.pack_conv_weight_bf16(torch.randn(c, c, 3, 3, 3) / (c * 27) ** 0.5).to(dev)
    sh = torch.randn(c, device=dev)
    t = timed(lambda: _hip.conv3d_forward_bf16(x, w16, c, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=sh, relu=True))
    mb = 2 * x.numel() * 2 / 1e6
    fl = 2.0 * 27 * c * c * d * h * h
    print(f'{c:4d} -> {c:4d} on {d}x{h}x{h}: {t:7.1f} us   {mb / t:6.2f} TB/s of in + out   {fl / t / 1e6:7.1f} TFLOP/s', flush=True)

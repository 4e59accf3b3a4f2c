"""The 64-channel bf16 kernel under repetition: 200 launches on a many-tiles-per-workgroup map must give one bit pattern (its halo double
buffering has no atomics and no data-dependent order), with other kernels running beside it on a second stream."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import mr3d_amd
from mr3d_amd import _hip
dev = torch.device('cuda:0'); torch.manual_seed(0)
x = torch.randn(1, 40, 128, 128, 64, device=dev).bfloat16()
w16 = _hip.pack_conv_weight_bf16(torch.randn(64, 64, 3, 3, 3) / (64 * 27) ** 0.5).to(dev)
sh = torch.randn(64, device=dev)
ref = _hip.conv3d_forward_bf16(x, w16, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=sh, relu=True).clone()
side = torch.cuda.Stream()
a = torch.randn(64 * 1024 * 1024, device=dev)
bad = 0
for it in range(200):
    with torch.cuda.stream(side):
        a.mul_(1.0001)                      # bandwidth-heavy neighbour
    y = _hip.conv3d_forward_bf16(x, w16, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=sh, relu=True)
    if not torch.equal(y, ref):
        bad += 1
torch.cuda.synchronize()
print('mismatching launches:', bad, 'of 200')

#!/usr/bin/env python3
"""conv3d_wino2d wide form: N identical launches on the P2 map give bit-identical results (hand-counted vmcnt waits, LDS written by
`buffer_load ... lds`: a race would show as a differing output).  usage (GPU box): python tools/diag/wide_soak.py [launches]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mr3d_amd
from mr3d_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(3)
x = torch.randn((1, 40, 128, 128, 64), generator=g).to(dev)
w = (torch.randn((27, 64, 64), generator=g) * 0.05).to(dev)
other = torch.randn((64, 1024, 1024), device=dev)
_hip.lib().mr3d_tuning_set(5, 2)
ref = _hip.conv3d_forward(x, w, (3,) * 3, (1,) * 3, (1,) * 3).clone()
y = torch.empty_like(ref)
bad = 0
side = torch.cuda.Stream()
for i in range(n):
    if i % 3 == 0:                      # a bandwidth-heavy neighbour on another stream
        with torch.cuda.stream(side):
            other.mul_(1.0001)
    _hip.conv3d_forward(x, w, (3,) * 3, (1,) * 3, (1,) * 3, out=y)
    if i % 50 == 49 or i == n - 1:
        if not torch.equal(y, ref):
            bad += 1
            print('launch', i, 'differs: max abs', (y - ref).abs().max().item())
torch.cuda.synchronize()
print(f'{n} launches, {bad} differing checks')
sys.exit(1 if bad else 0)

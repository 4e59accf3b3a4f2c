#!/usr/bin/env python3
"""Run only the stem kernels (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mr3d_amd
from mr3d_amd import _hip
dev = torch.device('cuda:0')
x = torch.randn(1, 3, 80, 512, 512, device=dev)
wp = torch.randn(343, 4, 16, device=dev) * 0.03
wp[:, 3] = 0
sh = torch.randn(16, device=dev)
dy = torch.randn(1, 80, 256, 256, 16, device=dev)
dw = torch.zeros(343, 4, 16, device=dev); db = torch.zeros(16, device=dev)
for _ in range(3):
    y = _hip.stem_conv_forward(x, wp, sh, relu=True)
    _hip.stem_conv_wgrad(x, dy, dw, db)
torch.cuda.synchronize()

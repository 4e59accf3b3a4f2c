#!/usr/bin/env python3
"""Runs the backbone's forward plan a few times (argument 1: 0 = the plan executor's side queues off, 1 = on) -- to be traced:
rocprofv3 --kernel-trace -- python3 tools/diag/fwd_gaps.py 0, then tools/timeline.py on the trace shows the gaps between the main
queue's kernels."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mr3d_amd
from mr3d_amd import _hip
from mr3d_amd.detector.extract_feat import ExtractFeat3D

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda:0')
eng = ExtractFeat3D(device=dev, dtype=torch.float32)
x = torch.randn((1, 3, 80, 512, 512), device=dev)
eng.forward_features(x)
plan = eng._plan_for(x.shape)
_hip.lib().mr3d_plan_set_side_stream(side)
for _ in range(4):
    eng.bind_input(plan, x)
    plan.fwd.run(dev)
    torch.cuda.synchronize()

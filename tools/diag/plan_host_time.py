#!/usr/bin/env python3
"""Host time of the backbone plan's two native calls (mr3d_plan_run enqueues a whole pass; the GPU runs behind it) against the GPU
time of the step: is the main queue ever empty because the host has not enqueued the next kernel yet?
usage (GPU box): python tools/diag/plan_host_time.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mr3d_amd
from mr3d_amd.detector.extract_feat import ExtractFeat3D

dev = torch.device('cuda:0')
eng = ExtractFeat3D(device=dev, dtype=torch.float32)
x = torch.randn((1, 3, 80, 512, 512), device=dev)
outs = eng.forward_features(x)
cots = [torch.randn(o.shape, device=dev).contiguous(memory_format=torch.channels_last_3d) for o in outs]
plan = eng._plan_for(x.shape)
for _ in range(5):
    eng.zero_grad(); eng.bind_input(plan, x); plan.fwd.run(dev); eng.bind_grads(plan, cots); plan.bwd.run(dev)
torch.cuda.synchronize()
tf = tb = tt = 0.0
n = 30
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.zero_grad(); eng.bind_input(plan, x)
    t1 = time.perf_counter()
    plan.fwd.run(dev)
    t2 = time.perf_counter()
    eng.bind_grads(plan, cots)
    t3 = time.perf_counter()
    plan.bwd.run(dev)
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    tf += t2 - t1; tb += t4 - t3; tt += t5 - t0
print(f'forward plan: {plan.fwd.n} records, host {tf / n * 1e3:.3f} ms; backward plan: {plan.bwd.n} records, host {tb / n * 1e3:.3f} ms; '
      f'step (host enqueue + GPU, one at a time) {tt / n * 1e3:.3f} ms')

#!/usr/bin/env python3
"""A/B of plan-level placement knobs of the headline step (ExtractFeat3D attributes read when a plan is built).
usage: python3 tools/placement_ab.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mr3d_amd  # noqa: E402,F401
from mr3d_amd.detector import ExtractFeat3D  # noqa: E402


def measure(eng, x, cots, steps):
    plan = eng._plan_for(x.shape)
    dev = x.device

    def step():
        eng.zero_grad()
        eng.bind_input(plan, x)
        plan.fwd.run(dev)
        eng.bind_grads(plan, cots)
        plan.bwd.run(dev)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    x = torch.randn(1, 3, 80, 512, 512, device=dev)
    cots = None
    for tag, attrs in (('default', {}), ('wgrads of layer3+layer4 at stage end', dict(wgrad_defer=(2, 3))),
                       ('wgrads of layer4 at stage end', dict(wgrad_defer=(3,))), ('wgrads of layer2..4 at stage end', dict(wgrad_defer=(1, 2, 3))),
                       ('default (again)', {})):
        eng = ExtractFeat3D(device=dev)
        for k, v in attrs.items():
            setattr(eng, k, v)
        if cots is None:
            cots = [torch.randn(o.shape, device=dev).contiguous(memory_format=torch.channels_last_3d) for o in eng.forward_features(x)]
        print(f'{tag:45s} {measure(eng, x, cots, steps):7.3f} ms', flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

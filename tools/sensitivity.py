#!/usr/bin/env python3
"""What would the headline step cost WITHOUT one class of plan records?  Upper bounds for kernel work, measured by leaving the
records out (MR3D_TUNE_PLAN_SKIP; the results of such a step are wrong -- timing only).  usage: python3 tools/sensitivity.py [steps] [bf16]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mr3d_amd  # noqa: E402,F401
from mr3d_amd import _hip  # noqa: E402
from mr3d_amd.detector import ExtractFeat3D  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    bf16 = 'bf16' in sys.argv[2:]
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    eng = ExtractFeat3D(device=dev, dtype=torch.bfloat16 if bf16 else torch.float32)
    x = torch.randn(1, 3, 80, 512, 512, device=dev)
    outs = eng.forward_features(x)
    cots = [torch.randn(o.shape, device=dev).contiguous(memory_format=torch.channels_last_3d) for o in outs]
    plan = eng._plan_for(x.shape)
    L = _hip.lib()

    def step():
        eng.zero_grad()
        eng.bind_input(plan, x)
        plan.fwd.run(dev)
        eng.bind_grads(plan, cots)
        plan.bwd.run(dev)

    def measure():
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    cases = [('everything', 0), ('- wgrad 3x3x3 16->16 (layer1 conv2)', 1), ('- wgrad 3x3x3 64->64 (P2 / P3 / layer3)', 2), ('- wgrad 1x1x1', 64),
             ('- every conv wgrad', 4), ('- max-pool forward', 8), ('- gradient finishing', 16), ('- stem wgrad', 32),
             ('- small-map 3x3x3 convs / dgrads (layer3/4)', 128),
             ('- stem forward', 256), ('- zero-stuff / upsample adjoint / max-pool backward', 512), ('- stem forward and wgrad', 256 + 32), ('everything (again)', 0)]
    base = None
    for tag, mask in cases:
        L.mr3d_tuning_set(1, mask)
        ms = measure()
        base = base or ms
        print(f'{tag:50s} {ms:7.3f} ms   ({ms - base:+.3f})', flush=True)
    L.mr3d_tuning_set(1, 0)


if __name__ == '__main__':
    main()

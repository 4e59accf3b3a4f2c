#!/usr/bin/env python3
"""bf16 stem kernels alone on the 512x512x80 volume: forward, dense-dy weight gradient, pooled-gradient weight gradient.
usage: python3 tools/stem_bf16_bench.py [reps]      (through gpurun; also the workload of rocprofv3 counter passes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd import _hip  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((1, 3, 80, 512, 512), device=dev, generator=g)
    wp = torch.randn((343, 4, 16), device=dev, generator=g) * 0.03
    wp[:, 3] = 0
    sh = torch.randn(16, device=dev, generator=g)
    dy = torch.randn((1, 80, 256, 256, 16), device=dev, generator=g).bfloat16()
    dw = torch.zeros(343, 4, 16, device=dev)
    db = torch.zeros(16, device=dev)
    w16 = torch.empty(49 * 4 * 16 * 8, dtype=torch.bfloat16, device=dev)
    y = torch.empty((1, 80, 256, 256, 16), dtype=torch.bfloat16, device=dev)
    L = _hip.lib()
    sp = _hip.stream_ptr(dev)
    _hip.check(L.mr3d_stem_pack_weights_bf16(_hip.ptr(wp), _hip.ptr(w16), sp), 'pack')

    def fwd():
        _hip.check(L.mr3d_stem_conv_forward_bf16(_hip.ptr(x), _hip.ptr(w16), _hip.ptr(sh), _hip.ptr(y), 1, 80, 512, 512, 1, sp), 'fwd')

    print(f'stem_fwd_bf16   {timed(fwd, reps):7.1f} us   (172 GFLOP useful)')
    print(f'stem_wgrad_bf16 {timed(lambda: _hip.stem_conv_wgrad_bf16(x, dy, dw, db), reps):7.1f} us')


if __name__ == '__main__':
    main()

"""stem weight gradient on the 512x512x80 volume: max-pool backward + dense-dy kernel against the pooled-gradient kernel
(mr3d_stem_conv_wgrad_pooled).  usage: python tools/stem_wgrad_bench.py [--density 1.0]      (through gpurun)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd import _hip  # noqa: E402


_flush = None


def timed_cold(fn, reps=6):
    """as timed(), but a 1 GB buffer is rewritten in front of every call: the operands come from HBM as they do inside a
    training step, not from the 256 MB Infinity Cache a back-to-back loop leaves them in"""
    global _flush
    if _flush is None:
        _flush = torch.empty(256 << 20, dtype=torch.float32, device='cuda')
    tot = 0.0
    for i in range(reps + 1):
        _flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        if i:
            tot += a.elapsed_time(b)
    return tot / reps * 1e3


def timed(fn, reps=10):
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--density', type=float, default=1.0, help='fraction of pooled positions with a non-zero gradient')
    ap.add_argument('--shape', type=int, nargs=3, default=(80, 512, 512))
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32', help='storage type of the activations / gradients')
    ap.add_argument('--tune', default='', help='key=value[,...] for mr3d_tuning_set')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    for kv in filter(None, a.tune.split(',')):
        _hip.lib().mr3d_tuning_set(*(int(v) for v in kv.split('=')))
    d, h, w = a.shape
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((1, 3, d, h, w), device=dev, generator=g)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    bf = a.dtype == 'bf16'
    dt = torch.bfloat16 if bf else torch.float32

    def op(kind, ptrs, ints):
        o = _hip.make_op(kind, ptrs, ints)
        if bf:
            o.i[25] = _hip.DTYPE_BF16
        return o

    y = torch.randn((1, d, ho, wo, 16), device=dev, generator=g).relu_().to(dt)
    pooled = torch.empty((1, (d - 1) // 2 + 1, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1, 16), device=dev, dtype=dt)
    am = torch.empty(pooled.shape[:4] + (4,), dtype=torch.int32, device=dev)
    _hip.Plan([op(_hip.OP_MAXPOOL_FWD, [y, pooled, am], [1, d, ho, wo, 16])]).run(dev)
    gp = torch.randn(pooled.shape, device=dev, generator=g)
    if a.density < 1.0:
        gp = gp * (torch.rand(gp.shape[:4] + (1,), device=dev, generator=g) < a.density)
    gp = gp.to(dt)
    gs = torch.empty_like(y)
    dw = torch.zeros((343, 4, 16), device=dev)
    db = torch.zeros(16, device=dev)
    bwd = _hip.Plan([op(_hip.OP_MAXPOOL_BWD, [gp, am, pooled, gs], [1, d, ho, wo, 16])])
    dense = (lambda w_, b_: _hip.stem_conv_wgrad_bf16(x, gs, w_, b_)) if bf else (lambda w_, b_: _hip.stem_conv_wgrad(x, gs, w_, b_))
    fused = ((lambda w_, b_: _hip.stem_conv_wgrad_bf16_pooled(x, gp, am, pooled, w_, b_)) if bf else
             (lambda w_, b_: _hip.stem_conv_wgrad_pooled(x, gp, am, pooled, w_, b_)))
    t_pool = timed(lambda: bwd.run(dev))
    t_dense = timed(lambda: dense(dw, db))
    dw1 = torch.zeros_like(dw); dense(dw1, None)
    _hip.lib().mr3d_tuning_set(3, 1)          # MR3D_TUNE_STEM_WGRAD_DENSE: the MFMA kernel with the scatter in its dy staging
    t_fused = timed(lambda: fused(dw, db))
    dw2 = torch.zeros_like(dw); fused(dw2, None)
    err = (dw1 - dw2).abs().max().item() / dw1.abs().max().item()
    _hip.lib().mr3d_tuning_set(3, 0)          # the gather over the pooled elements (stem_wgrad_gather.hip)
    ws = _hip.wgrad_workspace(dev)
    t_gather = timed(lambda: fused(dw, db))
    dw3 = torch.zeros_like(dw); db3 = torch.zeros_like(db); fused(dw3, db3)
    err3 = (dw1 - dw3).abs().max().item() / dw1.abs().max().item()
    print(f'{a.dtype} density {a.density}: maxpool_bwd {t_pool:.0f} us + stem_wgrad {t_dense:.0f} us = {t_pool + t_dense:.0f} us; '
          f'pooled-gradient MFMA kernel {t_fused:.0f} us; rel diff {err:.1e}; gather kernel {t_gather:.0f} us; rel diff {err3:.1e}')
    if not bf:
        t_ws = timed(lambda: _hip.stem_conv_wgrad_pooled(x, gp, am, pooled, dw, db, workspace=ws))
        t_cold = timed_cold(lambda: _hip.stem_conv_wgrad_pooled(x, gp, am, pooled, dw, db, workspace=ws))
        _hip.lib().mr3d_tuning_set(3, 1)
        t_cold_mfma = timed_cold(lambda: _hip.stem_conv_wgrad_pooled(x, gp, am, pooled, dw, db, workspace=ws))
        _hip.lib().mr3d_tuning_set(3, 0)
        print(f'gather kernel with the partial-sum workspace {t_ws:.0f} us; operands from HBM: gather {t_cold:.0f} us, MFMA kernel {t_cold_mfma:.0f} us')


if __name__ == '__main__':
    main()

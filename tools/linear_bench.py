#!/usr/bin/env python3
"""The bbox head's FC layers (csrc/linear.hip) alone: forward, data gradient and weight gradient of fc1 (512 x 9408 -> 1024) and
fc2 (1024 -> 1024) through the C ABI.  usage: python3 tools/linear_bench.py     (through gpurun)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd import _hip  # noqa: E402


def timed(fn, reps=20):
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
    dev = torch.device('cuda:0')
    L = _hip.lib()
    L.mr3d_linear_workspace_bytes.restype = ctypes.c_size_t
    st = _hip.stream_ptr(dev)
    for R, K, O in ((512, 9408, 1024), (512, 1024, 1024)):
        x = torch.randn(R, K, device=dev)
        w = torch.randn(O, K, device=dev) / K ** 0.5
        b = torch.randn(O, device=dev)
        g = torch.randn(R, O, device=dev)
        y, gx, gw = torch.empty(R, O, device=dev), torch.empty(R, K, device=dev), torch.empty(O, K, device=dev)
        need = L.mr3d_linear_workspace_bytes(R, K, O)
        ws = torch.empty(max(need // 4, 1), device=dev)
        wsp, wsl = ctypes.c_void_p(ws.data_ptr()), ctypes.c_longlong(ws.numel() * 4)
        fl = 2.0 * R * K * O
        t = timed(lambda: _hip.check(L.mr3d_linear_forward(_hip.ptr(x), _hip.ptr(w), _hip.ptr(b), _hip.ptr(y), R, K, O, 1, wsp, wsl, st), 'f'))
        print(f'{R} x {K} -> {O}  forward {t:7.1f} us  {fl / t / 1e6:6.1f} TFLOP/s', flush=True)
        t = timed(lambda: _hip.check(L.mr3d_linear_dgrad(_hip.ptr(g), _hip.ptr(w), _hip.ptr(gx), R, K, O, wsp, wsl, st), 'd'))
        print(f'{R} x {K} -> {O}  dgrad   {t:7.1f} us  {fl / t / 1e6:6.1f} TFLOP/s', flush=True)
        t = timed(lambda: _hip.check(L.mr3d_linear_wgrad(_hip.ptr(g), _hip.ptr(x), _hip.ptr(gw), R, K, O, wsp, wsl, st), 'w'))
        print(f'{R} x {K} -> {O}  wgrad   {t:7.1f} us  {fl / t / 1e6:6.1f} TFLOP/s', flush=True)
        ref = torch.relu(x.double() @ w.double().t() + b.double())
        print('   forward max rel err', ((y.double() - ref).abs().max() / ref.abs().max()).item(),
              ' dgrad', ((gx.double() - g.double() @ w.double()).abs().max() / (g.double() @ w.double()).abs().max()).item(),
              ' wgrad', ((gw.double() - g.double().t() @ x.double()).abs().max() / (g.double().t() @ x.double()).abs().max()).item())


if __name__ == '__main__':
    main()

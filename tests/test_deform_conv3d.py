"""DeformConv3d.  CPU: the torch restatement of the 3D extension reduces to the reference's 2D
operator on a kd = 1 slice.  GPU: the HIP operator (forward + all gradients) against the restatement."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import deform_ref as dr


def test_restatement_2d_slice_equals_reference_2d_operator():
    g = torch.Generator().manual_seed(0)
    N, C, H, W, Co, kh, kw, dg = 2, 4, 7, 6, 3, 3, 3, 2
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, C, kh, kw, generator=g, dtype=torch.float64)
    stride, pad, dil = (1, 1), (1, 1), (1, 1)
    off2 = torch.randn(N, dg * 2 * kh * kw, 7, 6, generator=g, dtype=torch.float64) * 1.5     # some samples leave the image
    ref2d = dr.deform_conv2d_reference_forward(x.numpy(), off2.numpy(), w.numpy(), stride, pad, dil, dg)
    # 3D call on a depth-1 volume: kd = 1, depth offsets 0, (dh, dw) copied per tap
    off3 = torch.zeros(N, dg * 3 * kh * kw, 1, 7, 6, dtype=torch.float64)
    for gi in range(dg):
        for t in range(kh * kw):
            off3[:, (gi * kh * kw + t) * 3 + 1, 0] = off2[:, gi * 2 * kh * kw + 2 * t]
            off3[:, (gi * kh * kw + t) * 3 + 2, 0] = off2[:, gi * 2 * kh * kw + 2 * t + 1]
    out3 = dr.deform_conv3d_torch(x.unsqueeze(2), off3, w.unsqueeze(2), stride=(1, 1, 1), pad=(0, 1, 1), dil=(1, 1, 1), dg=dg)
    np.testing.assert_allclose(out3[:, :, 0].numpy(), ref2d, rtol=1e-12, atol=1e-12)
    # zero offsets: plain convolution
    plain = dr.deform_conv3d_torch(x.unsqueeze(2), off3 * 0, w.unsqueeze(2), pad=(0, 1, 1), dg=dg)
    np.testing.assert_allclose(plain[:, :, 0].numpy(), F.conv2d(x, w, padding=1).numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('ci,co,k,stride,pad,dil,dg,modulated', [
    (16, 16, 3, 1, 1, 1, 1, False),
    (32, 24, 3, 1, 1, 1, 2, True),
    (16, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), 1, 1, False),
    (16, 16, 3, 1, 2, 2, 4, True),
])
def test_deform_conv3d_gpu_forward_backward(ci, co, k, stride, pad, dil, dg, modulated):
    from mr3d_amd.mmdet_ops import DeformConv3d, ModulatedDeformConv3d
    from torch.nn.modules.utils import _triple
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(ci + co)
    k3, s3, p3, d3 = _triple(k), _triple(stride), _triple(pad), _triple(dil)
    K = k3[0] * k3[1] * k3[2]
    N, D, H, W = 2, 5, 9, 8
    x = torch.randn(N, ci, D, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    osz = tuple((s_ + 2 * p_ - (d_ * (k_ - 1) + 1)) // st_ + 1 for s_, p_, d_, k_, st_ in zip((D, H, W), p3, d3, k3, s3))
    off = (torch.randn(N, dg * 3 * K, *osz, generator=g, dtype=torch.float64) * 1.2).requires_grad_(True)
    mask = torch.rand(N, dg * K, *osz, generator=g, dtype=torch.float64).requires_grad_(True) if modulated else None
    cls = ModulatedDeformConv3d if modulated else DeformConv3d
    layer = cls(ci, co, k, stride=stride, padding=pad, dilation=dil, deformable_groups=dg).to(dev)
    w64 = layer.weight.detach().cpu().double().requires_grad_(True)
    b64 = None
    if modulated:
        with torch.no_grad():
            layer.bias.copy_(torch.randn(co, generator=g))
        b64 = layer.bias.detach().cpu().double().requires_grad_(True)
    ref = dr.deform_conv3d_torch(x, off, w64, mask, b64, s3, p3, d3, dg)
    cot = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(cot)
    xd = x.detach().float().to(dev).requires_grad_(True)
    od = off.detach().float().to(dev).requires_grad_(True)
    args = [xd, od]
    if modulated:
        md = mask.detach().float().to(dev).requires_grad_(True)
        args.append(md)
    out = layer(*args)
    assert out.shape == ref.shape

    def rel(a, b):
        return float((a.cpu().double() - b).abs().max() / b.abs().max())
    assert rel(out.detach(), ref.detach()) <= 1e-4
    out.backward(cot.float().to(dev))
    assert rel(xd.grad, x.grad) <= 1e-4
    assert rel(layer.weight.grad, w64.grad) <= 1e-4
    # offset gradients are discontinuous where a sample sits within rounding of an integer coordinate or of the
    # (-1, dim) validity box: compare in L2 and bound the max error loosely
    l2 = float((od.grad.cpu().double() - off.grad).norm() / off.grad.norm())
    assert l2 <= 1e-4 and rel(od.grad, off.grad) <= 1e-3
    if modulated:
        assert rel(md.grad, mask.grad) <= 1e-4 and rel(layer.bias.grad, b64.grad) <= 1e-4


@pytest.mark.gpu
def test_deform_conv3d_zero_offset_equals_conv3d_and_errors():
    from mr3d_amd.mmdet_ops import DeformConv3d
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    layer = DeformConv3d(64, 64, 3, padding=1).to(dev)
    x = torch.randn(1, 64, 6, 12, 10, device=dev)
    off = torch.zeros(1, 81, 6, 12, 10, device=dev)
    ref = F.conv3d(x.cpu(), layer.weight.detach().cpu(), padding=1)
    out = layer(x, off)
    assert float((out.cpu() - ref).abs().max() / ref.abs().max()) <= 1e-4
    with pytest.raises(ValueError):
        layer(x[0], off)                                   # the reference raises on non-(N+2)-D input (deform_conv.py:21-24)
    with pytest.raises(NotImplementedError):
        layer(x.cpu(), off.cpu())

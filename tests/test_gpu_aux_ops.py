"""GPU parity tests of the HBM-bound helper kernels (csrc/aux_ops.hip) against torch CPU ops."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from mr3d_amd import _hip
    assert torch.cuda.is_available()
    return _hip


def _cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous().cuda()


def _nc(y):
    return y.permute(0, 4, 1, 2, 3).cpu()


def _run(hip, op):
    hip.Plan([op]).run(torch.device('cuda:0'))
    torch.cuda.synchronize()


@pytest.mark.parametrize('shape', [(1, 16, 8, 16, 16), (2, 16, 5, 9, 11), (1, 8, 7, 6, 10)])
def test_maxpool_forward_backward(hip, shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).requires_grad_(True)
    y, idx = F.max_pool3d(x, 3, 2, 1, return_indices=True)
    dy = torch.randn(y.shape, generator=g)
    xr = torch.relu(x.detach()).requires_grad_(True)          # backward with the fused ReLU mask of the stem
    yr = F.max_pool3d(xr, 3, 2, 1)
    n, c, d, h, w = shape
    xc = _cl(x.detach())
    yo = torch.empty((n,) + tuple(y.shape[2:]) + (c,), device='cuda')
    am = torch.empty(yo.shape[:4] + (c // 4,), dtype=torch.int32, device='cuda')
    _run(hip, hip.make_op(hip.OP_MAXPOOL_FWD, [xc, yo, am], [n, d, h, w, c]))
    np.testing.assert_array_equal(_nc(yo).numpy(), y.detach().numpy())
    y.backward(dy)
    dx = torch.empty_like(xc)
    _run(hip, hip.make_op(hip.OP_MAXPOOL_BWD, [_cl(dy), am, None, dx], [n, d, h, w, c]))
    np.testing.assert_allclose(_nc(dx).numpy(), x.grad.numpy(), rtol=1e-6, atol=1e-6)
    # with the pooled map: pool relu(x), route, and apply the ReLU mask through the argmax (y[o] > 0 <=> x[argmax(o)] > 0)
    xrc = _cl(xr.detach())
    _run(hip, hip.make_op(hip.OP_MAXPOOL_FWD, [xrc, yo, am], [n, d, h, w, c]))
    yr.backward(dy)
    _run(hip, hip.make_op(hip.OP_MAXPOOL_BWD, [_cl(dy), am, yo, dx], [n, d, h, w, c]))
    expect = xr.grad * (xr.detach() > 0)
    # windows of relu(x) that are entirely 0 tie at 0: torch and this kernel both take the first tap, and the mask zeroes it
    np.testing.assert_allclose(_nc(dx).numpy(), expect.numpy(), rtol=1e-6, atol=1e-6)


def test_zero_stuff_upsample_adjoint_subsample_volume_pack(hip):
    g = torch.Generator().manual_seed(5)
    # zero-stuffing: dst[2o] = src[o]
    src = torch.randn((1, 8, 3, 4, 5), generator=g)
    dst = torch.full((1, 5, 8, 9, 8), 7.0, device='cuda')
    _run(hip, hip.make_op(hip.OP_ZERO_STUFF2, [_cl(src), dst], [1, 3, 4, 5, 5, 8, 9, 8]))
    ref = torch.zeros(1, 8, 5, 8, 9); ref[:, :, ::2, ::2, ::2] = src
    np.testing.assert_array_equal(_nc(dst).numpy(), ref.numpy())
    # nearest-upsample adjoint, exact 2x and odd ratios (sizes as F.interpolate(size=...) sees them)
    for cs, fs in (((2, 4, 4), (4, 8, 8)), ((3, 5, 2), (5, 9, 4)), ((1, 3, 3), (2, 5, 7))):
        coarse = torch.zeros((1, 4) + cs, requires_grad=True)
        up = F.interpolate(coarse, size=fs, mode='nearest')
        gf = torch.randn(up.shape, generator=g)
        up.backward(gf)
        base = torch.randn((1, 4) + cs, generator=g)
        gc = _cl(base)
        _run(hip, hip.make_op(hip.OP_UPSAMPLE_ADJ, [_cl(gf), gc], [1, *fs, *cs, 4]))
        np.testing.assert_allclose(_nc(gc).numpy(), (base + coarse.grad).numpy(), rtol=1e-6, atol=1e-6)
    # stride-2 subsample (= max_pool3d(k=1, s=2)) and its adjoint
    x = torch.randn((2, 8, 5, 7, 6), generator=g, requires_grad=True)
    y = F.max_pool3d(x, 1, stride=2)
    yo = torch.empty((2,) + tuple(y.shape[2:]) + (8,), device='cuda')
    _run(hip, hip.make_op(hip.OP_SUBSAMPLE_FWD, [_cl(x.detach()), yo], [2, 5, 7, 6, 8]))
    np.testing.assert_array_equal(_nc(yo).numpy(), y.detach().numpy())
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    base = torch.randn(x.shape, generator=g)
    gx = _cl(base)
    _run(hip, hip.make_op(hip.OP_SUBSAMPLE_BWD, [_cl(dy), gx], [2, 5, 7, 6, 8]))
    np.testing.assert_allclose(_nc(gx).numpy(), (base + x.grad).numpy(), rtol=1e-6, atol=1e-6)
    # volume repack
    v = torch.randn((2, 3, 4, 6, 5), generator=g).cuda()
    out = torch.empty((2, 4, 6, 5, 4), device='cuda')
    _run(hip, hip.make_op(hip.OP_VOLUME_PACK, [v, out], [2, 4, 6, 5]))
    np.testing.assert_array_equal(out[..., :3].cpu().numpy(), v.permute(0, 2, 3, 4, 1).cpu().numpy())
    assert float(out[..., 3].abs().max()) == 0.0

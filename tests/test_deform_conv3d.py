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
@pytest.mark.parametrize('k,stride,pad,dil,scale', [
    (3, 1, 1, 1, 1.2),                       # the neck's shape (tiled dgrad, some corners beyond the LDS box); samples leave the volume
    (3, 1, 1, 1, 0.3),                       # every corner inside the LDS box
    (3, (1, 2, 2), 1, 1, 0.4),               # reach of a position block > 640 voxels: direct-atomic dgrad
    ((1, 3, 3), 1, (0, 2, 2), 2, 2.5),
])
def test_deform_conv3d_fused_path(k, stride, pad, dil, scale):
    """Cin = Cout = 64, one deformable group: the column-free kernels (csrc/deform_conv3d_fused.hip) -- forward and all three
    gradients against the float64 restatement, and the library reports the shape as fused"""
    import importlib
    from mr3d_amd.mmdet_ops import DeformConv3d
    mod = importlib.import_module('mr3d_amd.mmdet_ops.dcn.deform_conv3d')
    from torch.nn.modules.utils import _triple
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    k3, s3, p3, d3 = _triple(k), _triple(stride), _triple(pad), _triple(dil)
    K = k3[0] * k3[1] * k3[2]
    N, D, H, W, C = 2, 5, 9, 8, 64
    assert mod._fused_desc((N, D, H, W, C), k3, s3, p3, d3, 1, 64, False, False) is not None
    assert mod._fused_desc((N, D, H, W, 32), k3, s3, p3, d3, 1, 64, False, False) is None
    x = torch.randn(N, C, D, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    osz = tuple((s_ + 2 * p_ - (d_ * (k_ - 1) + 1)) // st_ + 1 for s_, p_, d_, k_, st_ in zip((D, H, W), p3, d3, k3, s3))
    off = (torch.randn(N, 3 * K, *osz, generator=g, dtype=torch.float64) * scale).requires_grad_(True)
    layer = DeformConv3d(C, C, k, stride=stride, padding=pad, dilation=dil).to(dev)
    w64 = layer.weight.detach().cpu().double().requires_grad_(True)
    ref = dr.deform_conv3d_torch(x, off, w64, None, None, s3, p3, d3, 1)
    cot = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(cot)
    xd = x.detach().float().to(dev).requires_grad_(True)
    od = off.detach().float().to(dev).requires_grad_(True)
    out = layer(xd, od)

    def rel(a, b):
        return float((a.cpu().double() - b).abs().max() / b.abs().max())
    assert out.shape == ref.shape and rel(out.detach(), ref.detach()) <= 1e-4
    out.backward(cot.float().to(dev))
    assert rel(xd.grad, x.grad) <= 1e-4
    assert rel(layer.weight.grad, w64.grad) <= 1e-4
    l2 = float((od.grad.cpu().double() - off.grad).norm() / off.grad.norm())
    assert l2 <= 1e-4 and rel(od.grad, off.grad) <= 1e-3
    # samples outside the validity box read nothing: a non-finite voxel 0 must not leak into them
    with torch.no_grad():
        x2 = xd.detach().clone()
        far = torch.full_like(od, 1e4)
        x2[0, :, 0, 0, 0] = float('inf')
        assert torch.equal(layer(x2, far), torch.zeros_like(out))


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


@pytest.mark.gpu
def test_dcn_neck_forward_backward_vs_restatement():
    """detector/dcn_neck.DCNNeck3D (BASELINE configs[4] "DeformConv3d neck"): offset conv + DeformConv3d per level and the
    stride-2 P6, forward and every gradient, against float64 torch (F.conv3d for the offsets + the DCN restatement); a
    zero-initialised neck equals the plain 3x3x3 FPN conv it replaces (fpn3d.py:56-64,111-123)"""
    from mr3d_amd.detector.dcn_neck import DCNNeck3D
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    neck = DCNNeck3D(channels=16, num_levels=2).to(dev)
    lats = [torch.randn(1, 16, 6, 10, 8, generator=g), torch.randn(1, 16, 3, 5, 4, generator=g)]
    # (a) fresh neck == plain conv with the DCN weights
    with torch.no_grad():
        outs = neck([l.to(dev).contiguous(memory_format=torch.channels_last_3d) for l in lats])
    assert len(outs) == 3
    for o, l, conv in zip(outs, lats, neck.conv):
        ref = F.conv3d(l.double(), conv.weight.detach().cpu().double(), None, 1, 1)
        assert float((o.cpu().double() - ref).abs().max() / ref.abs().max()) <= 1e-4
    assert torch.equal(outs[2], outs[1][:, :, ::2, ::2, ::2])
    # (b) trained-looking offsets: every gradient
    with torch.no_grad():
        for off in neck.conv_offset:
            off.weight.copy_(torch.randn(off.weight.shape, generator=g) * 0.05)
            off.bias.copy_(torch.randn(off.bias.shape, generator=g) * 0.3)
    xs = [l.to(dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True) for l in lats]
    outs = neck(xs)
    cots = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * c.to(dev)).sum() for o, c in zip(outs, cots)).backward()
    x64 = [l.double().requires_grad_(True) for l in lats]
    p64 = {n: p.detach().cpu().double().requires_grad_(True) for n, p in neck.named_parameters()}
    outs64 = []
    for i, x in enumerate(x64):
        offset = F.conv3d(x, p64[f'conv_offset.{i}.weight'], p64[f'conv_offset.{i}.bias'], 1, 1)
        outs64.append(dr.deform_conv3d_torch(x, offset, p64[f'conv.{i}.weight'], pad=(1, 1, 1)))
    outs64.append(outs64[-1][:, :, ::2, ::2, ::2])
    sum((o * c.double()).sum() for o, c in zip(outs64, cots)).backward()
    for o, r in zip(outs, outs64):
        assert float((o.detach().cpu().double() - r.detach()).abs().max() / r.detach().abs().max()) <= 1e-4
    for x, r in zip(xs, x64):
        assert float((x.grad.cpu().double() - r.grad).abs().max() / r.grad.abs().max()) <= 2e-4
    for n, p in neck.named_parameters():
        e = float((p.grad.cpu().double() - p64[n].grad).abs().max() / p64[n].grad.abs().max().clamp(min=1e-30))
        assert e <= 2e-4, (n, e)


@pytest.mark.gpu
@pytest.mark.parametrize('bf16', [False, True])
def test_detector_with_dcn_neck_trains(bf16):
    """BASELINE configs[4] switches on the single-pathway detector: DeformConv3d neck (+ bf16 backbone/FPN storage) -- the
    loss dict is finite, every parameter incl. the neck's offset convs and the backbone receives a gradient, the state
    dict uses the reference's `neck.fpn_convs.<i>.{conv,conv_offset}` names and round-trips"""
    from mr3d_amd.detector import TwoStageDetector3D
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    det = TwoStageDetector3D(device=dev, dtype=torch.bfloat16 if bf16 else torch.float32, neck_dcn=True)
    det.extract.zero_init_residual = False
    det.extract.init_weights()
    with torch.no_grad():
        for off in det.dcn_neck.conv_offset:
            off.weight.normal_(0, 0.01)
    img = torch.randn(1, 3, 32, 128, 128, device=dev)
    gt = torch.tensor([[20, 30, 60, 75, 4, 20], [70, 20, 110, 64, 10, 28]], dtype=torch.float32, device=dev)
    masks = torch.zeros((2, 32, 128, 128), dtype=torch.uint8, device=dev)
    for k, b in enumerate(gt.int().tolist()):
        masks[k, b[4]:b[5] + 1, b[1]:b[3] + 1, b[0]:b[2] + 1] = 1
    meta = [dict(img_shape=(128, 128, 3, 32), pad_shape=(128, 128, 3, 32))]
    det.extract.zero_grad()
    losses = det.forward_train(img, meta, [gt], [torch.ones(2, dtype=torch.long, device=dev)], gt_masks=[masks])
    total = det.total_loss(losses)
    assert torch.isfinite(total)
    total.backward()
    for n, p in det.dcn_neck.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        if '.0.' in n or '.1.' in n:          # P2 / P3 always carry sampled anchors and rois (the 2x4x4 P5 of this crop may not)
            assert p.grad.abs().sum() > 0, n
    g = det.extract.flat_grads
    assert torch.isfinite(g).all() and g.abs().sum() > 0
    assert det.extract.get_parameter('backbone.conv1.weight').grad.abs().sum() > 0
    sd = det.state_dict()
    assert 'neck.fpn_convs.0.conv.weight' in sd and 'neck.fpn_convs.3.conv_offset.bias' in sd and 'neck.lateral_convs.0.conv.weight' in sd
    assert not any(k.startswith(('extract.', 'dcn_neck.')) for k in sd)
    det2 = TwoStageDetector3D(device=dev, dtype=torch.bfloat16 if bf16 else torch.float32, neck_dcn=True)
    missing, unexpected = det2.load_state_dict(sd, strict=True)
    assert torch.equal(det2.dcn_neck.conv[2].weight, det.dcn_neck.conv[2].weight)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,scale', [
    ((2, 5, 9, 8), 0.4),          # every sample inside the LDS box; tiles cut by the volume on every axis
    ((1, 9, 13, 19), 1.3),        # offsets beyond +-1: samples outside the box (global-memory corners) and outside the volume
    ((1, 4, 8, 16), 0.0),         # zero offsets: the plain convolution
])
def test_deform_conv3d_bf16_box_kernels(shape, scale):
    """The bf16 form of the neck's DeformConv3d (csrc/deform_conv3d_bf16.hip: samples from a voxel box in LDS, bf16 MFMA) --
    forward and the three gradients against the float64 restatement evaluated on the SAME bf16-rounded inputs, offsets and
    weights.  What is left is the kernels' own rounding: columns / dcol to bf16 (2^-9 relative per entry, averaged over 1728
    resp. 64 products) and the bf16 results."""
    from mr3d_amd.mmdet_ops.dcn import deform_conv3d_bf16
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(7)
    N, D, H, W = shape
    C, K, S = 64, 27, 96
    rb = lambda t: t.to(torch.bfloat16).double()
    x = rb(torch.randn(N, C, D, H, W, generator=g)).requires_grad_(True)
    off = rb(torch.randn(N, 3 * K, D, H, W, generator=g) * scale).requires_grad_(True)
    w = rb(torch.randn(C, C, 3, 3, 3, generator=g) * 0.05).requires_grad_(True)
    ref = dr.deform_conv3d_torch(x, off, w, None, None, (1, 1, 1), (1, 1, 1), (1, 1, 1), 1)
    cot = rb(torch.randn(ref.shape, generator=g))
    ref.backward(cot)
    x16 = x.detach().permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16).requires_grad_(True)
    off16 = torch.zeros(N, D, H, W, S, dtype=torch.bfloat16, device=dev)
    off16[..., :3 * K] = off.detach().permute(0, 2, 3, 4, 1).to(dev, torch.bfloat16)
    off16.requires_grad_(True)
    wd = w.detach().float().to(dev).requires_grad_(True)
    y16 = deform_conv3d_bf16(x16, off16, wd, 1, 1, 1)
    assert y16.dtype == torch.bfloat16 and tuple(y16.shape) == (N, D, H, W, C)

    def l2(a, b):
        return float((a.cpu().double() - b).norm() / b.norm())

    def mx(a, b):
        return float((a.cpu().double() - b).abs().max() / b.abs().max())
    y_ref = ref.detach().permute(0, 2, 3, 4, 1)
    assert l2(y16, y_ref) <= 4e-3 and mx(y16, y_ref) <= 1.5e-2, (l2(y16, y_ref), mx(y16, y_ref))
    y16.backward(cot.permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16))
    gx_ref = x.grad.permute(0, 2, 3, 4, 1)
    assert l2(x16.grad, gx_ref) <= 4e-3 and mx(x16.grad, gx_ref) <= 1.5e-2, (l2(x16.grad, gx_ref), mx(x16.grad, gx_ref))
    assert l2(wd.grad, w.grad) <= 4e-3 and mx(wd.grad, w.grad) <= 1.5e-2, (l2(wd.grad, w.grad), mx(wd.grad, w.grad))
    go = off16.grad
    assert go.dtype == torch.bfloat16 and float(go[..., 3 * K:].abs().max()) == 0.0          # the padding stays zero
    if scale > 0:
        go_ref = off.grad.permute(0, 2, 3, 4, 1)
        # offset gradients are discontinuous where a sample sits within rounding of an integer coordinate or of the validity box
        assert l2(go[..., :3 * K], go_ref) <= 1e-2, l2(go[..., :3 * K], go_ref)

"""GPU parity tests of the MFMA implicit-GEMM conv3d kernels against torch's CPU fp32 conv
(the arithmetic the reference delegates to torch.nn.Conv3d)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
RTOL = 1e-4     # north-star tolerance for conv floating point


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _cl(x):      # NCDHW cpu tensor -> NDHWC device tensor
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _check(got_ndhwc, ref_ncdhw, what):
    """1e-4 of the tensor's scale in max-norm (an output is a sum of up to 1728 signed products: an element-wise relative bound
    on its near-zero entries would be meaningless) AND 1e-4 in L2"""
    got = got_ndhwc.permute(0, 4, 1, 2, 3).cpu()
    scale = ref_ncdhw.abs().max().item()
    err = (got - ref_ncdhw).abs().max().item()
    assert err <= RTOL * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e}'
    l2 = ((got - ref_ncdhw).norm() / ref_ncdhw.norm().clamp(min=1e-30)).item()
    assert l2 <= RTOL, f'{what}: L2 rel err {l2:.3e}'


CASES = [
    # (Cin, Cout, k, stride, pad, (D,H,W))
    (16, 16, 1, (1, 1, 1), 0, (5, 9, 11)),
    (16, 64, 1, (1, 1, 1), 0, (4, 8, 8)),
    (64, 16, 1, (1, 1, 1), 0, (6, 10, 7)),
    (64, 32, 1, (1, 1, 1), 0, (4, 9, 8)),
    (32, 128, 1, (1, 1, 1), 0, (3, 8, 9)),
    (128, 64, 1, (1, 1, 1), 0, (3, 5, 8)),
    (512, 128, 1, (1, 1, 1), 0, (2, 4, 5)),
    (256, 64, 1, (1, 1, 1), 0, (5, 16, 15)),
    (128, 512, 1, (1, 1, 1), 0, (3, 8, 9)),
    (256, 256, 1, (1, 1, 1), 0, (2, 5, 7)),
    (64, 128, 1, (2, 2, 2), 0, (6, 10, 12)),
    (16, 16, 3, (1, 1, 1), 1, (6, 10, 12)),
    (32, 32, 3, (1, 1, 1), 1, (5, 9, 9)),
    (64, 64, 3, (1, 1, 1), 1, (8, 16, 17)),
    (128, 128, 3, (1, 1, 1), 1, (3, 6, 8)),
    (32, 32, 3, (2, 2, 2), 1, (8, 16, 14)),
    (64, 64, 3, (2, 2, 2), 1, (7, 11, 9)),
    (4, 16, 7, (1, 2, 2), 3, (6, 20, 24)),
]


@pytest.mark.parametrize('cin,cout,k,stride,pad,dhw', CASES)
def test_conv3d_forward_plain(dev, cin, cout, k, stride, pad, dhw):
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(cin * 1000 + cout + k)
    x = torch.randn((2, cin) + dhw, generator=g)
    w = torch.randn((cout, cin, k, k, k), generator=g) / np.sqrt(cin * k ** 3)
    ref = F.conv3d(x, w, None, stride, pad)
    y = _hip.conv3d_forward(_cl(x).to(dev), _hip.pack_conv_weight(w).to(dev), (k,) * 3, stride, (pad,) * 3)
    assert tuple(y.shape) == (2,) + tuple(ref.shape[2:]) + (cout,)
    _check(y, ref, 'plain')


def test_conv3d_forward_fused_epilogue(dev):
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(1)
    x = torch.randn((1, 64, 6, 10, 12), generator=g)
    w = torch.randn((64, 64, 3, 3, 3), generator=g) / np.sqrt(64 * 27)
    shift = torch.randn(64, generator=g)
    add1 = torch.randn((1, 64, 6, 10, 12), generator=g)
    add2 = torch.randn((1, 64, 6, 10, 12), generator=g)
    up = torch.randn((1, 64, 3, 5, 6), generator=g)
    mask = torch.randn((1, 64, 6, 10, 12), generator=g)
    conv = F.conv3d(x, w, shift, 1, 1)
    ref = torch.relu(conv + add1 + add2 + F.interpolate(up, size=(6, 10, 12), mode='nearest'))
    y = _hip.conv3d_forward(_cl(x).to(dev), _hip.pack_conv_weight(w).to(dev), (3, 3, 3), (1, 1, 1), (1, 1, 1),
                            shift=shift.to(dev), add1=_cl(add1).to(dev), add2=_cl(add2).to(dev), up=_cl(up).to(dev),
                            relu=True)
    _check(y, ref, 'fused relu')
    ref2 = (conv + add1) * (mask > 0)
    y2 = _hip.conv3d_forward(_cl(x).to(dev), _hip.pack_conv_weight(w).to(dev), (3, 3, 3), (1, 1, 1), (1, 1, 1),
                             shift=shift.to(dev), add1=_cl(add1).to(dev), mask=_cl(mask).to(dev))
    _check(y2, ref2, 'fused mask')
    # odd upsample ratio (5 -> 3 style sizes as F.interpolate(size=...) sees them in FPN3D)
    up3 = torch.randn((1, 64, 2, 3, 5), generator=g)
    ref3 = conv + F.interpolate(up3, size=(6, 10, 12), mode='nearest')
    y3 = _hip.conv3d_forward(_cl(x).to(dev), _hip.pack_conv_weight(w).to(dev), (3, 3, 3), (1, 1, 1), (1, 1, 1),
                             shift=shift.to(dev), up=_cl(up3).to(dev))
    _check(y3, ref3, 'fused upsample odd')


@pytest.fixture(params=[1, 2], ids=['tile2x8x8', 'wide4x8x8'])
def wino2d_form(request):
    """MR3D_TUNE_WINO2D_FORM: both kernels of csrc/conv3d_wino2d.hip on every case (left alone the library picks per map)"""
    from mr3d_amd import _hip
    _hip.lib().mr3d_tuning_set(5, request.param)
    yield request.param
    _hip.lib().mr3d_tuning_set(5, 0)


def test_conv3d_forward_winograd_2d_large_map(dev, wino2d_form):
    """3x3x3 64 -> 64 on a map with >= 256 tiles of 2x8x8 takes the F(4,3) x F(2,3) kernels (csrc/conv3d_wino2d.hip): odd
    extents (partial tiles on every axis), the whole fused epilogue, and the data-gradient use (ReLU mask) against an fp64 conv"""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(5)
    dhw = (17, 45, 51)
    x = torch.randn((1, 64) + dhw, generator=g)
    w = torch.randn((64, 64, 3, 3, 3), generator=g) / np.sqrt(64 * 27)
    shift = torch.randn(64, generator=g)
    add1 = torch.randn((1, 64) + dhw, generator=g)
    up = torch.randn((1, 64, 9, 23, 26), generator=g)
    mask = torch.randn((1, 64) + dhw, generator=g)
    conv = F.conv3d(x.double(), w.double(), shift.double(), 1, 1)
    ref = torch.relu(conv + add1.double() + F.interpolate(up.double(), size=dhw, mode='nearest'))
    wp = _hip.pack_conv_weight(w).to(dev)
    y = _hip.conv3d_forward(_cl(x).to(dev), wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift.to(dev), add1=_cl(add1).to(dev),
                            up=_cl(up).to(dev), relu=True)
    _check(y, ref.float(), 'winograd 2d fused relu')
    ref2 = (conv + add1.double()) * (mask > 0)
    y2 = _hip.conv3d_forward(_cl(x).to(dev), wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift.to(dev), add1=_cl(add1).to(dev),
                             mask=_cl(mask).to(dev))
    _check(y2, ref2.float(), 'winograd 2d fused mask')
    l2 = ((y2.permute(0, 4, 1, 2, 3).cpu().double() - ref2).norm() / ref2.norm()).item()
    assert l2 <= 1e-5, f'L2-relative error {l2:.3e}'


@pytest.mark.parametrize('cin,cout', [(64, 128), (128, 64), (32, 64)])
def test_conv3d_forward_winograd_2d_other_channel_counts(dev, cin, cout, wino2d_form):
    """the 64-wide two-dimensional Winograd kernels with two Cout tiles (blockIdx.y), with 16 Cin chunks of 8 and with 4: against an
    fp64 conv on a ragged map, fused shift + ReLU"""
    from mr3d_amd import _hip
    import ctypes
    g = torch.Generator().manual_seed(cin * 3 + cout)
    dhw = (17, 45, 51)                                  # 378 tiles of 2x8x8: the map takes these kernels
    d = _hip.Conv3dDesc()
    d.n, d.d, d.h, d.w_in, d.cin, d.cout = 1, dhw[0], dhw[1], dhw[2], cin, cout
    d.kd = d.kh = d.kw = 3
    d.sd = d.sh = d.sw = d.pd = d.ph = d.pw = 1
    d.workspace_bytes = -1
    v = _hip.Conv3dVariant()
    assert _hip.lib().mr3d_conv3d_forward_variant(ctypes.byref(d), ctypes.byref(v)) == 0
    assert v.winograd == 8 and v.tile_d == (4 if wino2d_form == 2 else 2)
    x = torch.randn((1, cin) + dhw, generator=g)
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(cin * 27)
    shift = torch.randn(cout, generator=g)
    ref = torch.relu(F.conv3d(x.double(), w.double(), shift.double(), 1, 1))
    y = _hip.conv3d_forward(_cl(x).to(dev), _hip.pack_conv_weight(w).to(dev), (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift.to(dev), relu=True)
    _check(y, ref.float(), f'{cin}->{cout}')


def test_conv3d_forward_winograd_2d_strided_output_grid(dev, wino2d_form):
    """mr3d_conv3d_desc.ys_* / y_*: the result of a 64 -> 64 3x3x3 convolution scattered onto the even positions of a zeroed grid of
    twice the extent (how a coarse-grid result lands in a fine map), from both two-dimensional Winograd kernels: the even positions
    equal the plain call, everything else stays zero."""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(17)
    dhw = (9, 42, 50)
    x = _cl(torch.randn((1, 64) + dhw, generator=g)).to(dev)
    wp = _hip.pack_conv_weight(torch.randn((64, 64, 3, 3, 3), generator=g) / np.sqrt(64 * 27)).to(dev)
    shift = torch.randn(64, generator=g).to(dev)
    plain = _hip.conv3d_forward(x, wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift, relu=True)
    big = torch.zeros((1, 2 * dhw[0], 2 * dhw[1], 2 * dhw[2], 64), device=dev)
    _hip.conv3d_forward(x, wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift, relu=True, out=big, out_stride=(2, 2, 2))
    assert torch.equal(big[:, ::2, ::2, ::2], plain)
    big[:, ::2, ::2, ::2] = 0
    assert not bool(big.any())


def test_conv3d_forward_winograd_2d_wide_form_walks_tiles(dev):
    """The wide kernel's workgroups are persistent: 256 of them walk the tiles of a launch, an XCD's 32 side by side over one eighth
    of the tile range, with the chunk pipeline running across tile borders (the next tile's halo and first V are requested inside the
    current tile's last chunks).  Two images, ragged extents (1080 tiles of 4x8x8: 135 per XCD, 4.2 per workgroup, so ranges end
    mid-row and some workgroups take one tile more than others), the fused tail -- against the 2x8x8-tile kernel (itself checked
    against fp64 above) and, on a window, against an fp64 conv."""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(11)
    dhw = (21, 70, 75)
    x = torch.randn((2, 64) + dhw, generator=g)
    w = torch.randn((64, 64, 3, 3, 3), generator=g) / np.sqrt(64 * 27)
    shift = torch.randn(64, generator=g)
    add1 = torch.randn((2, 64) + dhw, generator=g)
    xd, wp, sd, ad = _cl(x).to(dev), _hip.pack_conv_weight(w).to(dev), shift.to(dev), _cl(add1).to(dev)
    ys = {}
    for form in (1, 2):
        _hip.lib().mr3d_tuning_set(5, form)
        try:
            ys[form] = _hip.conv3d_forward(xd, wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=sd, add1=ad, relu=True)
        finally:
            _hip.lib().mr3d_tuning_set(5, 0)
    scale = ys[1].abs().max().item()
    assert (ys[1] - ys[2]).abs().max().item() <= 2e-5 * scale
    # a window that spans tile borders on every axis and the seam between the two images' tile ranges, in fp64
    win = (slice(15, 21), slice(58, 70), slice(60, 75))
    xin = x[:, :, 13:21, 56:70, 58:75].double()                 # the window's halo (the volume's far faces pad with zeros)
    ref = F.conv3d(F.pad(xin, (0, 1, 0, 1, 0, 1)), w.double(), shift.double())[:, :, 1:, 1:, 1:]
    ref = torch.relu(ref + add1[(slice(None), slice(None)) + win].double())
    got = ys[2].permute(0, 4, 1, 2, 3)[(slice(None), slice(None)) + win].cpu().double()
    assert (got - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize('c,dhw', [(16, (19, 67, 75)), (32, (13, 61, 70))])
def test_conv3d_forward_winograd_2d_narrow_cout(dev, c, dhw):
    """3x3x3 16 -> 16 / 32 -> 32 on maps with >= 256 tiles take the row-split F(4,3) x F(2,3) kernels (conv3d_wino2d_rows_kernel):
    ragged extents, fused epilogue in both uses (forward: ReLU; data gradient: ReLU mask), against an fp64 conv"""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(c)
    x = torch.randn((1, c) + dhw, generator=g)
    w = torch.randn((c, c, 3, 3, 3), generator=g) / np.sqrt(c * 27)
    shift = torch.randn(c, generator=g)
    add1 = torch.randn((1, c) + dhw, generator=g)
    mask = torch.randn((1, c) + dhw, generator=g)
    conv = F.conv3d(x.double(), w.double(), shift.double(), 1, 1)
    wp = _hip.pack_conv_weight(w).to(dev)
    y = _hip.conv3d_forward(_cl(x).to(dev), wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift.to(dev), add1=_cl(add1).to(dev), relu=True)
    _check(y, torch.relu(conv + add1.double()).float(), 'winograd 2d rows relu')
    y2 = _hip.conv3d_forward(_cl(x).to(dev), wp, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift.to(dev), add1=_cl(add1).to(dev),
                             mask=_cl(mask).to(dev))
    _check(y2, ((conv + add1.double()) * (mask > 0)).float(), 'winograd 2d rows mask')


def test_conv3d_winograd_2d_other_channel_counts(dev):
    """the two-dimensional Winograd kernels with two Cin chunks and two 64-wide Cout tiles (32 -> 128): forward and weight gradient"""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(21)
    dhw = (9, 41, 50)                                    # 5 x 6 x 7 = 210 tiles of 2x8x8 per image, two images
    x = torch.randn((2, 32) + dhw, generator=g)
    w = (torch.randn((128, 32, 3, 3, 3), generator=g) / np.sqrt(32 * 27)).double().requires_grad_(True)
    shift = torch.randn(128, generator=g)
    y = F.conv3d(x.double(), w, shift.double(), 1, 1)
    got = _hip.conv3d_forward(_cl(x).to(dev), _hip.pack_conv_weight(w.detach().float()).to(dev), (3, 3, 3), (1, 1, 1), (1, 1, 1),
                              shift=shift.to(dev))
    _check(got, y.detach().float(), 'winograd 2d 32 -> 128')
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    ref = _hip.pack_conv_weight(w.grad.float()).double()
    dw = torch.zeros((27, 32, 128), device=dev)
    db = torch.zeros(128, device=dev)
    _hip.conv3d_wgrad(_cl(x).to(dev), _cl(dy).to(dev), dw, (3, 3, 3), (1, 1, 1), (1, 1, 1), dbias=db)
    err = (dw.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= RTOL, f'wgrad 2d 32 -> 128: max err {err:.3e}'
    refb = dy.double().sum((0, 2, 3, 4))
    assert (db.cpu().double() - refb).abs().max().item() <= 2e-5 * refb.abs().max().item()


def test_conv3d_unsupported_is_loud(dev):
    from mr3d_amd import _hip
    x = torch.zeros((1, 2, 4, 4, 12), device=dev)      # Cin = 12
    w = torch.zeros((1, 12, 16), device=dev)
    with pytest.raises(_hip.Mr3dError):
        _hip.conv3d_forward(x, w, (1, 1, 1), (1, 1, 1), (0, 0, 0))


@pytest.mark.parametrize('cin,cout,k,stride,pad,dhw', CASES)
def test_conv3d_wgrad(dev, cin, cout, k, stride, pad, dhw):
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(cin * 1000 + cout + k + 7)
    x = torch.randn((2, cin) + dhw, generator=g)
    w = torch.zeros((cout, cin, k, k, k), requires_grad=True)
    y = F.conv3d(x, w, None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ref = _hip.pack_conv_weight(w.grad)                     # [taps, Cin, Cout]
    dw = torch.zeros(ref.shape, device=dev)
    _hip.conv3d_wgrad(_cl(x).to(dev), _cl(dy).to(dev), dw, (k,) * 3, stride, (pad,) * 3)
    err = (dw.cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= RTOL * scale, f'wgrad max err {err:.3e} vs scale {scale:.3e}'
    # two-stage reduction (private partial blocks + a summing kernel) instead of atomics: same result, accumulating
    # into dw, and bit-identical from run to run
    ws = _hip.wgrad_workspace(dev)
    dw2 = dw.clone()
    _hip.conv3d_wgrad(_cl(x).to(dev), _cl(dy).to(dev), dw2, (k,) * 3, stride, (pad,) * 3, workspace=ws)
    assert (dw2.cpu() - 2 * ref).abs().max().item() <= 2 * RTOL * scale
    dw3, dw4 = torch.zeros_like(dw), torch.zeros_like(dw)
    _hip.conv3d_wgrad(_cl(x).to(dev), _cl(dy).to(dev), dw3, (k,) * 3, stride, (pad,) * 3, workspace=ws)
    _hip.conv3d_wgrad(_cl(x).to(dev), _cl(dy).to(dev), dw4, (k,) * 3, stride, (pad,) * 3, workspace=ws)
    if not (k == 7 and cin == 4):              # the kw-merged 7x7x7 fallback keeps atomics
        assert torch.equal(dw3, dw4)


def test_conv3d_wgrad_winograd_2d_large_map(dev):
    """3x3x3 64 -> 64 weight gradient on a map with >= 256 tiles of 2x8x8 takes the F(3,2) x F(3,4) kernel
    (csrc/conv3d_wgrad_wino2d.hip): ragged extents, two images, bias gradient, accumulation into a non-zero dw; vs an fp64 conv"""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(9)
    dhw = (11, 45, 51)
    x = torch.randn((2, 64) + dhw, generator=g)
    w = torch.zeros((64, 64, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    y = F.conv3d(x.double(), w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    ref = _hip.pack_conv_weight(w.grad.float()).double()
    dw = torch.zeros((27, 64, 64), device=dev)
    db = torch.zeros(64, device=dev)
    for rep in (1, 2):                                   # the second call accumulates on top of the first
        _hip.conv3d_wgrad(_cl(x).to(dev), _cl(dy).to(dev), dw, (3, 3, 3), (1, 1, 1), (1, 1, 1), dbias=db)
        err = (dw.cpu().double() - rep * ref).abs().max().item() / ref.abs().max().item()
        l2 = ((dw.cpu().double() - rep * ref).norm() / ref.norm()).item()
        assert err <= rep * RTOL and l2 <= rep * 1e-5, f'wgrad 2d: max err {err:.3e}, L2 {l2:.3e}'
    refb = dy.double().sum((0, 2, 3, 4))
    assert (db.cpu().double() - 2 * refb).abs().max().item() <= 2e-5 * refb.abs().max().item()


@pytest.mark.parametrize('n,dhw', [(1, (8, 32, 32)), (1, (16, 64, 64)), (2, (10, 40, 36)), (1, (5, 70, 130)), (1, (9, 17, 23))])
def test_stem_kernels_forward_and_wgrad(dev, n, dhw):
    """the specialised 7x7x7 / stride (1,2,2) stem kernels against an fp64 torch conv"""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(sum(dhw) + n)
    x = torch.randn((n, 3) + dhw, generator=g)
    w = (torch.randn((16, 3, 7, 7, 7), generator=g) / np.sqrt(1029)).requires_grad_(True)
    shift = torch.randn(16, generator=g)
    ref = F.conv3d(x.double(), w.double(), shift.double(), (1, 2, 2), 3)
    w4 = torch.zeros(16, 4, 7, 7, 7); w4[:, :3] = w.detach()
    wp = _hip.pack_conv_weight(w4).to(dev)
    y = _hip.stem_conv_forward(x.to(dev), wp, shift.to(dev), relu=False)
    err = (y.permute(0, 4, 1, 2, 3).cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-5, f'stem fwd rel err {err:.3e}'
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    dw = torch.zeros((343, 4, 16), device=dev); db = torch.zeros(16, device=dev)
    _hip.stem_conv_wgrad(x.to(dev), _cl(dy).to(dev), dw, db)
    dw_ws = torch.zeros_like(dw)
    _hip.stem_conv_wgrad(x.to(dev), _cl(dy).to(dev), dw_ws, None, workspace=_hip.wgrad_workspace(dev))
    assert (dw_ws - dw).abs().max().item() <= 2e-5 * dw.abs().max().item()
    refp = _hip.pack_conv_weight(torch.cat([w.grad, torch.zeros(16, 1, 7, 7, 7)], 1))
    errw = (dw.cpu() - refp).abs().max().item() / refp.abs().max().item()
    errb = (db.cpu() - dy.sum((0, 2, 3, 4))).abs().max().item() / dy.sum((0, 2, 3, 4)).abs().max().item()
    assert errw <= 2e-5 and errb <= 1e-4, f'stem wgrad rel err {errw:.3e}, dbias {errb:.3e}'


@pytest.mark.parametrize('n,dhw,relu', [(1, (8, 32, 32), True), (2, (10, 40, 36), True), (1, (5, 70, 130), False), (1, (9, 17, 23), True),
                                        (1, (16, 64, 64), True), (1, (7, 38, 90), True), (2, (3, 6, 10), False), (1, (1, 2, 2), True), (1, (2, 4, 6), False),
                                        (3, (4, 8, 10), True), (1, (33, 34, 66), True)])
def test_stem_wgrad_from_the_pooled_gradient(dev, n, dhw, relu):
    """mr3d_stem_conv_wgrad_pooled: conv -> (ReLU) -> MaxPool3d(3, 2, 1) differentiated by torch in fp64 (resnet3d.py:467-471)
    against the kernel that scatters the pooled gradient to its arg-max positions inside its dy staging; the pooled
    gradient is sparse in one case (most windows 0) like the detector's"""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(sum(dhw) + 3 * n)
    x = torch.randn((n, 3) + dhw, generator=g)
    w = (torch.randn((16, 3, 7, 7, 7), generator=g) / np.sqrt(1029)).requires_grad_(True)
    shift = (0.3 * torch.randn(16, generator=g)).requires_grad_(True)
    w4 = torch.zeros(16, 4, 7, 7, 7); w4[:, :3] = w.detach()
    wp = _hip.pack_conv_weight(w4).to(dev)
    y = _hip.stem_conv_forward(x.to(dev), wp, shift.detach().to(dev), relu=relu)            # NDHWC fp32
    nn, d, ho, wo, _ = y.shape
    pooled = torch.empty((nn, (d - 1) // 2 + 1, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1, 16), device=dev)
    am = torch.empty(pooled.shape[:4] + (4,), dtype=torch.int32, device=dev)
    _hip.Plan([_hip.make_op(_hip.OP_MAXPOOL_FWD, [y, pooled, am], [nn, d, ho, wo, 16])]).run(dev)
    gp = torch.randn(pooled.shape, generator=g)
    if dhw[0] == 10:
        gp = gp * (torch.rand(gp.shape[:4] + (1,), generator=g) < 0.1)
    # torch reference on the device's own activation (ties / arg-max then agree): pool the kernel's y, route gp through
    # the pool, mask by y > 0, differentiate the conv in fp64
    yk = y.permute(0, 4, 1, 2, 3).cpu().double().requires_grad_(True)
    F.max_pool3d(yk, 3, 2, 1).backward(gp.permute(0, 4, 1, 2, 3).double())
    dy = yk.grad * (yk.detach() > 0) if relu else yk.grad
    F.conv3d(x.double(), w.double(), shift.double(), (1, 2, 2), 3).backward(dy)
    # both forms: the gather over the pooled elements (default; even W) and the dense MFMA kernel (MR3D_TUNE_STEM_WGRAD_DENSE)
    for dense, ws in ((0, None), (0, _hip.wgrad_workspace(dev)), (1, None), (1, _hip.wgrad_workspace(dev))):
        _hip.lib().mr3d_tuning_set(3, dense)
        try:
            dw = torch.zeros((343, 4, 16), device=dev); db = torch.zeros(16, device=dev)
            _hip.stem_conv_wgrad_pooled(x.to(dev), gp.to(dev), am, pooled if relu else None, dw, db, workspace=ws)
        finally:
            _hip.lib().mr3d_tuning_set(3, 0)
        refp = _hip.pack_conv_weight(torch.cat([w.grad, torch.zeros(16, 1, 7, 7, 7)], 1))
        errw = (dw.cpu() - refp).abs().max().item() / refp.abs().max().item()
        errb = (db.cpu() - shift.grad).abs().max().item() / shift.grad.abs().max().item()
        assert errw <= 2e-5 and errb <= 1e-4, f'stem wgrad (pooled, dense={dense}) rel err {errw:.3e}, dbias {errb:.3e}'
    # and against the two-launch form it replaces
    gs = torch.empty_like(y)
    _hip.Plan([_hip.make_op(_hip.OP_MAXPOOL_BWD, [gp.to(dev), am, pooled if relu else None, gs], [nn, d, ho, wo, 16])]).run(dev)
    dw2 = torch.zeros_like(dw)
    _hip.stem_conv_wgrad(x.to(dev), gs, dw2, None)
    assert (dw2 - dw).abs().max().item() <= 2e-5 * dw2.abs().max().item()


# the streaming pointwise kernels (csrc/pointwise.hip) only engage above 32768 positions
PW_CASES = [(16, 16), (16, 64), (64, 16), (64, 32), (32, 64), (64, 64), (32, 128), (128, 32), (64, 128), (128, 64)]


@pytest.mark.parametrize('cin,cout', PW_CASES)
def test_pointwise_streaming_forward_epilogues_and_wgrad(dev, cin, cout):
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(cin * 7 + cout)
    dhw = (9, 64, 66)                                   # 38016 positions: not a multiple of the 64/32-row tiles
    x = torch.randn((1, cin) + dhw, generator=g)
    w = torch.randn((cout, cin, 1, 1, 1), generator=g) / np.sqrt(cin)
    shift = torch.randn(cout, generator=g)
    add1, add2, mask = (torch.randn((1, cout) + dhw, generator=g) for _ in range(3))
    up = torch.randn((1, cout, 5, 32, 33), generator=g)
    conv = F.conv3d(x.double(), w.double(), shift.double())
    xc, wp = _cl(x).to(dev), _hip.pack_conv_weight(w).to(dev)
    k1, s1, p0 = (1, 1, 1), (1, 1, 1), (0, 0, 0)
    y = _hip.conv3d_forward(xc, wp, k1, s1, p0, shift=shift.to(dev), add1=_cl(add1).to(dev), relu=True)
    _check(y, torch.relu(conv + add1).float(), 'pw add1+relu')
    y = _hip.conv3d_forward(xc, wp, k1, s1, p0, add1=_cl(add1).to(dev), add2=_cl(add2).to(dev), mask=_cl(mask).to(dev))
    _check(y, ((conv - shift.double().view(1, -1, 1, 1, 1) + add1 + add2) * (mask > 0)).float(), 'pw adds+mask')
    y = _hip.conv3d_forward(xc, wp, k1, s1, p0, shift=shift.to(dev), up=_cl(up).to(dev))
    _check(y, (conv + F.interpolate(up.double(), size=dhw, mode='nearest')).float(), 'pw upsample-add')
    wz = torch.zeros_like(w, requires_grad=True)
    yz = F.conv3d(x, wz)
    dy = torch.randn(yz.shape, generator=g)
    yz.backward(dy)
    dw, db = torch.zeros((1, cin, cout), device=dev), torch.zeros(cout, device=dev)
    _hip.conv3d_wgrad(xc, _cl(dy).to(dev), dw, k1, s1, p0, dbias=db)
    ref = _hip.pack_conv_weight(wz.grad)
    assert (dw.cpu() - ref).abs().max().item() <= RTOL * ref.abs().max().item()
    dw_ws = torch.zeros_like(dw)
    _hip.conv3d_wgrad(xc, _cl(dy).to(dev), dw_ws, k1, s1, p0, workspace=_hip.wgrad_workspace(dev))
    assert (dw_ws.cpu() - ref).abs().max().item() <= RTOL * ref.abs().max().item()
    sb = dy.sum((0, 2, 3, 4))
    assert (db.cpu() - sb).abs().max().item() <= 1e-4 * dy.abs().sum((0, 2, 3, 4)).max().item()


@pytest.mark.parametrize('cin', [64, 128])
def test_pointwise_streaming_column_blocks_to_256(dev, cin):
    """64 -> 256 and 128 -> 256 on 15 x 48 x 48 (layer3's conv3 and layer4.0 conv1's data gradient on the 768 x 768 x 120 pathway of
    the shipped model: 34,560 positions, whole 64-row tiles) run as column blocks of the streaming 64 -> 64 / 128 -> 32 instances
    (csrc/pointwise.hip, PwCfg<..., BLK>): the plain call, shift + residual + ReLU, two adds + mask, against an fp64 conv; a ragged
    position count and an upsample add keep the generic kernel (same results)."""
    import ctypes
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(256 + cin)
    k1, s1, p0 = (1, 1, 1), (1, 1, 1), (0, 0, 0)
    w = torch.randn((256, cin, 1, 1, 1), generator=g) / np.sqrt(cin)
    wp = _hip.pack_conv_weight(w).to(dev)
    shift = torch.randn(256, generator=g)
    for dhw, blocked in (((15, 48, 48), True), ((15, 48, 47), False)):
        x = torch.randn((1, cin) + dhw, generator=g)
        add1, add2, mask = (torch.randn((1, 256) + dhw, generator=g) for _ in range(3))
        conv = F.conv3d(x.double(), w.double(), shift.double())
        xc = _cl(x).to(dev)
        d = _hip.Conv3dDesc()
        d.n, d.d, d.h, d.w_in, d.cin, d.cout = 1, dhw[0], dhw[1], dhw[2], cin, 256
        d.kd = d.kh = d.kw = d.sd = d.sh = d.sw = 1
        d.ys_d = d.ys_h = d.ys_w = 1
        d.y_d, d.y_h, d.y_w = dhw
        v = _hip.Conv3dVariant()
        assert _hip.lib().mr3d_conv3d_forward_variant(ctypes.byref(d), ctypes.byref(v)) == 0 and bool(v.streaming_pointwise) == blocked
        y = _hip.conv3d_forward(xc, wp, k1, s1, p0)
        _check(y, (conv - shift.double().view(1, -1, 1, 1, 1)).float(), 'plain')
        y = _hip.conv3d_forward(xc, wp, k1, s1, p0, shift=shift.to(dev), add1=_cl(add1).to(dev), relu=True)
        _check(y, torch.relu(conv + add1).float(), 'shift + add1 + relu')
        y = _hip.conv3d_forward(xc, wp, k1, s1, p0, add1=_cl(add1).to(dev), add2=_cl(add2).to(dev), mask=_cl(mask).to(dev))
        _check(y, ((conv - shift.double().view(1, -1, 1, 1, 1) + add1 + add2) * (mask > 0)).float(), 'adds + mask')
    up = torch.randn((1, 256, 8, 24, 24), generator=g)
    dhw = (15, 48, 48)
    x = torch.randn((1, cin) + dhw, generator=g)
    conv = F.conv3d(x.double(), w.double())
    y = _hip.conv3d_forward(_cl(x).to(dev), wp, k1, s1, p0, up=_cl(up).to(dev))
    _check(y, (conv + F.interpolate(up.double(), size=dhw, mode='nearest')).float(), 'upsample add: generic kernel')


@pytest.mark.parametrize('planes,dhw', [(32, (8, 16, 14)), (64, (7, 11, 9)), (128, (5, 16, 16)), (16, (9, 33, 20)),
                                        (64, (20, 64, 64))])
def test_conv3d_dgrad_s2k3_parity_decomposed(dev, planes, dhw):
    """mr3d_conv3d_dgrad_s2k3 == torch's conv_transpose3d data gradient of the strided conv2 (resnet3d.py:139),
    with the producing layer's ReLU mask and two residual adds fused; odd and even extents, every Cout tile."""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(planes + dhw[0])
    n = 1 if dhw[1] >= 64 else 2
    w = torch.randn((planes, planes, 3, 3, 3), generator=g) / np.sqrt(planes * 27)
    do = tuple((v - 1) // 2 + 1 for v in dhw)
    dy = torch.randn((n, planes) + do, generator=g)
    a1 = torch.randn((n, planes) + dhw, generator=g)          # the conv's input (post-ReLU activation): mask = a1 > 0
    add = torch.randn((n, planes) + dhw, generator=g)
    ref = torch.nn.grad.conv3d_input((n, planes) + dhw, w, dy, stride=2, padding=1)
    # wb[26 - tap][co][ci] = W[co][ci][tap]   (mr3d_prepare_weights layout)
    wb = w.permute(2, 3, 4, 0, 1).reshape(27, planes, planes).flip(0).contiguous().to(dev)
    dx = _hip.conv3d_dgrad_s2k3(_cl(dy).to(dev), wb, (n,) + dhw + (planes,))
    _check(dx, ref, 'plain')
    dx = _hip.conv3d_dgrad_s2k3(_cl(dy).to(dev), wb, (n,) + dhw + (planes,), add1=_cl(add).to(dev), mask=_cl(a1).to(dev))
    _check(dx, torch.where(a1 > 0, ref + add, torch.zeros_like(ref)), 'add+mask')


def test_conv3d_dgrad_s2k3_rejects_bad_grid(dev):
    from mr3d_amd import _hip
    dy = torch.zeros((1, 4, 4, 4, 32), device=dev)
    wb = torch.zeros((27, 32, 32), device=dev)
    with pytest.raises(_hip.Mr3dError):
        _hip.conv3d_dgrad_s2k3(dy, wb, (1, 10, 8, 8, 32))      # 10 -> coarse 5, not 4


@pytest.mark.parametrize('cin,cout,n,dhw', [(64, 64, 72, (10, 14, 14)), (16, 32, 80, (10, 14, 14)), (32, 16, 70, (5, 7, 13))])
def test_conv3d_roi_volumes_5x7x7_tiles(dev, cin, cout, n, dhw):
    """mask-head shapes (fcn_mask_head_3d.py:45-70 on [R,64,10,14,14] roi features): the 245-position 5x7x7 tile path
    (forward, data gradient = forward on flipped weights, weight gradient + bias gradient)."""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(cin + cout + n)
    x = torch.randn((n, cin) + dhw, generator=g)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(cin * 27)).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv3d(x, w, b, 1, 1)
    yd = _hip.conv3d_forward(_cl(x).to(dev), _hip.pack_conv_weight(w.detach()).to(dev), (3,) * 3, (1,) * 3, (1,) * 3,
                             shift=b.to(dev), relu=True)
    _check(yd, y.detach().clamp(min=0), 'forward+bias+relu')
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw = torch.zeros((27, cin, cout), device=dev)
    db = torch.zeros(cout, device=dev)
    _hip.conv3d_wgrad(_cl(x).to(dev), _cl(dy).to(dev), dw, (3,) * 3, (1,) * 3, (1,) * 3, dbias=db)
    ref = _hip.pack_conv_weight(w.grad)
    assert (dw.cpu() - ref).abs().max().item() <= RTOL * ref.abs().max().item()
    refb = dy.sum((0, 2, 3, 4))
    assert (db.cpu() - refb).abs().max().item() <= RTOL * refb.abs().max().item()


def test_dominant_kernel_full_size_vs_miopen(dev):
    """BASELINE configs[1] size: the P2 3x3x3 64->64 conv ([1,64,40,128,128], the roofline's dominant kernel), forward,
    data gradient and weight gradient, against torch's own GPU convolution (MIOpen) -- an implementation that shares
    nothing with this one -- plus linearity of the weight gradient in dy (size-independent property)."""
    from mr3d_amd import _hip
    g = torch.Generator(device='cpu').manual_seed(99)
    x = torch.randn((1, 64, 40, 128, 128), generator=g).to(dev)
    w = (torch.randn((64, 64, 3, 3, 3), generator=g) / np.sqrt(64 * 27)).to(dev)
    dy = torch.randn((1, 64, 40, 128, 128), generator=g).to(dev)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    try:
        ref = F.conv3d(xr, wr, None, 1, 1)
        ref.backward(dy)
    except RuntimeError as exc:          # the checker, not the product: a MIOpen failure must not mask the other tests
        pytest.skip(f'torch/MIOpen convolution unavailable: {exc}')
    xc = x.permute(0, 2, 3, 4, 1).contiguous()
    dyc = dy.permute(0, 2, 3, 4, 1).contiguous()
    y = _hip.conv3d_forward(xc, _hip.pack_conv_weight(w), (3,) * 3, (1,) * 3, (1,) * 3)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    assert rel(y.permute(0, 4, 1, 2, 3), ref.detach()) <= RTOL
    wb = w.reshape(64, 64, 27).flip(2).permute(2, 0, 1).contiguous()
    dx = _hip.conv3d_forward(dyc, wb, (3,) * 3, (1,) * 3, (1,) * 3)
    assert rel(dx.permute(0, 4, 1, 2, 3), xr.grad) <= RTOL
    dw = torch.zeros((27, 64, 64), device=dev)
    _hip.conv3d_wgrad(xc, dyc, dw, (3,) * 3, (1,) * 3, (1,) * 3)
    assert rel(dw, _hip.pack_conv_weight(wr.grad)) <= RTOL
    dw2 = torch.zeros_like(dw)
    _hip.conv3d_wgrad(xc, (dyc * 2.0).contiguous(), dw2, (3,) * 3, (1,) * 3, (1,) * 3)
    assert rel(dw2, 2.0 * dw) <= 1e-5


@pytest.mark.parametrize('n,c,dhw,form', [(2, 64, (16, 61, 63), 'F(4,3), 4x8x8 tiles'), (1, 64, (9, 61, 63), 'F(2,3), 2x8x8 tiles'),
                                          (1, 32, (16, 61, 63), 'F(2,3), 4x8x8 tiles, 32-wide Cout tile')])
def test_conv3d_winograd_w_path_matches_direct_and_reference(dev, n, c, dhw, form):
    """3x3x3 / stride 1 / 64-wide Cout tile takes a Winograd-along-W form of the kernel (FwdCfg<..., WINO>): F(4,3) on
    4x8x8 tiles when there are >= 512 of them, else F(2,3) on 2x8x8 tiles (>= 256).  Same results as an fp64 torch
    convolution within the kernel tolerance, on volumes with an ODD width (the last output group is partly valid) and
    with every fused epilogue stream; the measured error must also leave a margin under the tolerance."""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(4242 + n + c)
    co = c                                              # 64 -> 64 (FPN / RPN / mask head), 32 -> 32 (layer2)
    x = torch.randn((n, c) + dhw, generator=g)
    w = torch.randn((co, c, 3, 3, 3), generator=g) / np.sqrt(27 * c)
    shift = torch.randn(co, generator=g)
    add1, mask = torch.randn((n, co) + dhw, generator=g), torch.randn((n, co) + dhw, generator=g)
    ref = F.conv3d(x.double(), w.double(), shift.double(), 1, 1)
    xc, wp = _cl(x).to(dev), _hip.pack_conv_weight(w).to(dev)
    y = _hip.conv3d_forward(xc, wp, (3,) * 3, (1,) * 3, (1,) * 3, shift=shift.to(dev))
    _check(y, ref.float(), f'winograd {form} plain')
    err = ((y.permute(0, 4, 1, 2, 3).cpu().double() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 3e-5, f'{form}: {err:.2e} leaves no margin under the 1e-4 tolerance'
    y = _hip.conv3d_forward(xc, wp, (3,) * 3, (1,) * 3, (1,) * 3, shift=shift.to(dev), add1=_cl(add1).to(dev), relu=True)
    _check(y, torch.relu(ref + add1).float(), f'winograd {form} add+relu')
    y = _hip.conv3d_forward(xc, wp, (3,) * 3, (1,) * 3, (1,) * 3, add1=_cl(add1).to(dev), mask=_cl(mask).to(dev))
    _check(y, ((ref - shift.double().view(1, -1, 1, 1, 1) + add1) * (mask > 0)).float(), f'winograd {form} add+mask')
    if c != 64:
        return
    # 128 output channels: two Cout tiles per position tile
    w2 = torch.randn((128, c, 3, 3, 3), generator=g) / np.sqrt(27 * c)
    y = _hip.conv3d_forward(xc, _hip.pack_conv_weight(w2).to(dev), (3,) * 3, (1,) * 3, (1,) * 3)
    _check(y, F.conv3d(x.double(), w2.double(), None, 1, 1).float(), f'winograd {form} 128 cout')


def test_conv3d_winograd_nonfinite_input_stays_local(dev):
    """the Winograd-along-W forms are not bit-compatible with the direct form for NON-FINITE input.  The
    reference's direct convolution turns one inf voxel into inf / nan in the 27 outputs whose window contains it; F(4,3)
    transforms 6 inputs x[4q .. 4q+5] into every product of an output GROUP of 4 (F(2,3): 4 inputs, groups of 2), so the
    whole group of any output that reads the voxel is poisoned -- a superset along W, bounded here: non-finite outputs stay
    inside the 3x3 (d, h) neighbourhood and inside the W-groups that overlap the reference's window, every other output is
    unchanged, and MR3D_FWD_WINO=0 semantics (the direct kernel: exactly the reference's set) remain selectable."""
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(5)
    dhw = (16, 61, 64)                                     # >= 512 tiles of 4x8x8: the F(4,3) instance
    x = torch.randn((1, 64) + dhw, generator=g)
    w = torch.randn((64, 64, 3, 3, 3), generator=g) / np.sqrt(27 * 64)
    xc, wp = _cl(x).to(dev), _hip.pack_conv_weight(w).to(dev)
    clean = _hip.conv3d_forward(xc, wp, (3,) * 3, (1,) * 3, (1,) * 3)
    pz, py, px = 7, 30, 33
    xc2 = xc.clone()
    xc2[0, pz, py, px, 5] = float('inf')
    y = _hip.conv3d_forward(xc2, wp, (3,) * 3, (1,) * 3, (1,) * 3)
    bad = ~torch.isfinite(y).all(dim=4)[0]                  # [D,H,W] positions with a non-finite channel
    idx = bad.nonzero()
    assert len(idx) >= 27                                   # at least the reference's 3x3x3 window
    assert idx[:, 0].min() >= pz - 1 and idx[:, 0].max() <= pz + 1 and idx[:, 1].min() >= py - 1 and idx[:, 1].max() <= py + 1
    lo, hi = ((px - 1) // 4) * 4, ((px + 1) // 4) * 4 + 3   # the groups of 4 outputs that overlap [px-1, px+1]
    assert idx[:, 2].min() >= lo and idx[:, 2].max() <= hi, (idx[:, 2].min().item(), idx[:, 2].max().item(), lo, hi)
    ref_window = torch.zeros_like(bad)
    ref_window[pz - 1:pz + 2, py - 1:py + 2, px - 1:px + 2] = True
    assert bad[ref_window].all()                             # a superset of the reference's poisoned outputs, never less
    good = ~bad
    assert torch.equal(y[0][good], clean[0][good])           # everything else is bit-identical to the clean run


def test_conv3d_wgrad_winograd_path_ragged_volume(dev):
    """3x3x3 / stride 1 weight gradient on >= 2048 tiles takes the Winograd F(3,2)-along-W form (WgCfg<..., WINO>):
    checked on a volume whose extents are NOT multiples of the 4x8x8 tile (half-valid output pairs, tiles hanging over
    every face) against torch's own GPU convolution backward, plus the bias gradient and accumulation into dw"""
    from mr3d_amd import _hip
    g = torch.Generator(device='cpu').manual_seed(77)
    x = torch.randn((1, 64, 41, 127, 125), generator=g).to(dev)
    dy = torch.randn((1, 64, 41, 127, 125), generator=g).to(dev)
    wr = torch.zeros((64, 64, 3, 3, 3), device=dev, requires_grad=True)
    try:
        F.conv3d(x, wr, None, 1, 1).backward(dy)
    except RuntimeError as exc:
        pytest.skip(f'torch/MIOpen convolution unavailable: {exc}')
    ref = _hip.pack_conv_weight(wr.grad)
    xc, dyc = x.permute(0, 2, 3, 4, 1).contiguous(), dy.permute(0, 2, 3, 4, 1).contiguous()
    dw, db = torch.zeros((27, 64, 64), device=dev), torch.zeros(64, device=dev)
    _hip.conv3d_wgrad(xc, dyc, dw, (3,) * 3, (1,) * 3, (1,) * 3, dbias=db)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    assert rel(dw, ref) <= RTOL
    assert rel(db, dy.sum((0, 2, 3, 4))) <= 1e-4
    _hip.conv3d_wgrad(xc, dyc, dw, (3,) * 3, (1,) * 3, (1,) * 3)            # accumulates
    assert rel(dw, 2 * ref) <= RTOL


def test_conv3d_forward_liveness_bitmap_writes_dead_tiles_from_the_epilogue(dev, wino2d_form):
    """mr3d_voxel_liveness + mr3d_conv3d_desc.x_live (data gradients of sparse maps in detector training): the bitmap holds exactly
    the positions with a non-zero channel, and the large-map 3x3x3 kernel given the bitmap returns what it returns without it -- on a
    map with a few non-zero boxes and single voxels (on tile corners, on the volume's faces, a lone -0.0 that does not count), with
    shift / add / ReLU in the epilogue (dead tiles are NOT zero then), for 64 fp32 / 16 fp32 / 64 bf16 channels, and on a dense map"""
    from mr3d_amd import _hip
    g = torch.Generator(device='cpu').manual_seed(21)
    shape = (1, 40, 128, 125, 64)                   # W not a multiple of 8 or 32: rows straddle bitmap words
    dense = torch.randn(shape, generator=g).to(dev)
    sparse = torch.zeros_like(dense)
    boxes = ((3, 9, 10, 31, 7, 22), (20, 21, 64, 65, 64, 65), (38, 40, 120, 128, 118, 125), (15, 17, 7, 9, 90, 100),
             (0, 1, 0, 1, 0, 1), (39, 40, 127, 128, 124, 125), (8, 9, 8, 9, 8, 9), (7, 8, 7, 8, 7, 8))
    for (d0, d1, h0, h1, w0, w1) in boxes:
        sparse[:, d0:d1, h0:h1, w0:w1] = dense[:, d0:d1, h0:h1, w0:w1]
    sparse[0, 30, 30, 30, 5] = -0.0
    sparse[0, 12, 100, 3, 63] = 1e-30               # one tiny channel is enough

    def unpack(bits, npos):
        b = bits[: (npos + 31) // 32].view(torch.int32).cpu().numpy().astype(np.uint32)
        return torch.from_numpy(((b[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1)[:npos].astype(bool))
    npos = sparse.numel() // 64
    for x in (sparse, sparse[..., :16].contiguous(), sparse.to(torch.bfloat16)):
        got = unpack(_hip.voxel_liveness(x), npos)
        assert torch.equal(got, (x.float() != 0).any(dim=4).reshape(-1).cpu())
    assert bool(unpack(_hip.voxel_liveness(dense), npos).all())
    w = (torch.randn((27, 64, 64), generator=g) * 0.05).to(dev)
    shift = torch.randn(64, generator=g).to(dev)
    add = torch.randn(shape, generator=g).to(dev)
    for x in (sparse, dense):
        live = _hip.voxel_liveness(x)
        for kw in (dict(), dict(shift=shift, add1=add, relu=True)):
            y0 = _hip.conv3d_forward(x, w, (3,) * 3, (1,) * 3, (1,) * 3, **kw)
            y1 = _hip.conv3d_forward(x, w, (3,) * 3, (1,) * 3, (1,) * 3, x_live=live, **kw)
            assert torch.equal(y0, y1)
    # timing: the sparse map's launch is several times shorter than the same launch without the bitmap
    m = _hip.voxel_liveness(sparse)

    def timed(**kw):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        _hip.conv3d_forward(sparse, w, (3,) * 3, (1,) * 3, (1,) * 3, **kw)
        ev[0].record()
        for _ in range(5):
            _hip.conv3d_forward(sparse, w, (3,) * 3, (1,) * 3, (1,) * 3, **kw)
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / 5
    t_dense, t_map = timed(), timed(x_live=m)
    print(f'3x3x3 64->64 on 40x128x125, a few live boxes: {t_dense:.3f} ms without the bitmap, {t_map:.3f} ms with it')
    assert t_map < 0.6 * t_dense
    # and through a plan: the VOXEL_LIVENESS record arms the CONV_FWD record behind it
    y2 = torch.empty_like(sparse)
    lbits = torch.empty_like(m)
    ops = [_hip.make_op(_hip.OP_VOXEL_LIVENESS, [sparse, lbits], [1, 40, 128, 125, 64, 1]),
           _hip.make_op(_hip.OP_CONV_FWD, [sparse, w, y2, None, None, None, None, None],
                        [1, 40, 128, 125, 64, 64, 3, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0])]
    _hip.Plan(ops).run(dev)
    assert torch.equal(lbits, m) and torch.equal(y2, _hip.conv3d_forward(sparse, w, (3,) * 3, (1,) * 3, (1,) * 3))


def test_conv3d_wgrad_sparse_dy_flag_skips_zero_tiles_only(dev):
    """MR3D_WGRAD_SPARSE_DY (detector training: the gradient at the RPN conv / FPN output convs is zero on most position tiles):
    with a dy that is non-zero in a few scattered boxes -- some cut by tile borders, one single voxel -- the tile-skipping
    instance returns the dense instance's sums (and bias gradient); with a dense dy likewise"""
    from mr3d_amd import _hip
    g = torch.Generator(device='cpu').manual_seed(5)
    shape = (1, 41, 127, 125, 64)
    xc = torch.randn(shape, generator=g).to(dev)
    dense = torch.randn(shape, generator=g).to(dev)
    sparse = torch.zeros_like(dense)
    for (d0, d1, h0, h1, w0, w1) in ((3, 9, 10, 31, 7, 22), (20, 21, 64, 65, 64, 65), (38, 41, 120, 127, 118, 125), (15, 17, 7, 9, 90, 100)):
        sparse[:, d0:d1, h0:h1, w0:w1] = dense[:, d0:d1, h0:h1, w0:w1]
    for dy in (sparse, dense):
        out = []
        for flag in (False, True):
            dw, db = torch.zeros((27, 64, 64), device=dev), torch.zeros(64, device=dev)
            _hip.conv3d_wgrad(xc, dy, dw, (3,) * 3, (1,) * 3, (1,) * 3, dbias=db, sparse_dy=flag)
            out.append((dw, db))
        assert out[0][0].abs().max() > 0
        assert ((out[1][0] - out[0][0]).abs().max() / out[0][0].abs().max()).item() <= 1e-5          # atomics: order only
        assert ((out[1][1] - out[0][1]).abs().max() / out[0][1].abs().max()).item() <= 1e-5


@pytest.mark.parametrize('c0,c1,c2,P', [(64, 256, 64, 10240), (64, 256, 64, 77), (64, 256, 64, 32), (16, 64, 16, 65536), (16, 64, 16, 32768 + 32 * 7)])
@pytest.mark.parametrize('mode', ['forward', 'backward'])
def test_pointwise_pair_vs_fp64(dev, c0, c1, c2, P, mode):
    """mr3d_pointwise_pair (csrc/pw_pair.hip): conv3 + identity + ReLU -> next conv1 + ReLU (forward) and conv1^T + residual
    gradient + mask -> conv3^T + mask (backward) in one launch, against float64; ragged position counts included"""
    import ctypes
    from mr3d_amd import _hip
    g = torch.Generator().manual_seed(c0 + P)
    x = torch.randn(P, c0, generator=g)
    w1 = torch.randn(c0, c1, generator=g) / c0 ** 0.5
    w2 = torch.randn(c1, c2, generator=g) / c1 ** 0.5
    add1 = torch.randn(P, c1, generator=g)
    if mode == 'forward':
        sh1, sh2 = torch.randn(c1, generator=g), torch.randn(c2, generator=g)
        y1 = torch.relu(x.double() @ w1.double() + sh1.double() + add1.double())
        y2 = torch.relu(y1 @ w2.double() + sh2.double())
        m1 = m2 = None
    else:
        sh1 = sh2 = None
        m1, m2 = torch.randn(P, c1, generator=g), torch.randn(P, c2, generator=g)
        y1 = (x.double() @ w1.double() + add1.double()) * (m1 > 0)
        y2 = (y1 @ w2.double()) * (m2 > 0)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    xd, w1d, w2d, ad, s1, s2, m1d, m2d = d(x), d(w1), d(w2), d(add1), d(sh1), d(sh2), d(m1), d(m2)
    o1 = torch.full((P, c1), 7.0, device=dev)
    o2 = torch.full((P, c2), 7.0, device=dev)
    # small maps: the workgroup-tile kernel (64 / 256 / 64); >= 32k positions in whole 32-row tiles: the streaming kernel (16 / 64 / 16)
    assert _hip.pw_pair_supported(c0, c1, c2, P) and not _hip.pw_pair_supported(64, 256, 64, 40000) and not _hip.pw_pair_supported(16, 64, 16, 1024)
    assert not _hip.pw_pair_supported(16, 64, 16, 32768 + 5)
    op = _hip.make_pw_pair_op(xd, w1d, s1, ad, m1d, o1, w2d, s2, m2d, o2, c0, c1, c2, mode == 'forward', mode == 'forward', P)
    _hip.Plan([op]).run(dev)
    torch.cuda.synchronize()
    for got, exp in ((o1, y1), (o2, y2)):
        err = (got.double().cpu() - exp).abs().max().item()
        assert err <= 1e-4 * exp.abs().max().item(), (mode, err)

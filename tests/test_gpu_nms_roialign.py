"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the
C ABI, against the oracle and the committed golden vectors."""
import numpy as np
import pytest
import torch

from oracle import binding as orc
import ref_impls as ri

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    import mr3d_amd
    assert torch.cuda.is_available()
    assert mr3d_amd._hip.lib().mr3d_device_ok() == 1
    return torch.device('cuda:0')


# ------------------------------------------------------------------ NMS3D: bit-exact indices
def test_nms3d_golden(dev, golden_dir):
    from mr3d_amd.mmdet_ops import nms
    g = np.load(f'{golden_dir}/nms3d.npz')
    for k in range(int(g['num_cases'])):
        dets = torch.from_numpy(g[f'dets_{k}']).to(dev)
        kept, inds = nms(dets, float(g[f'thr_{k}']))
        assert inds.dtype == torch.long and inds.device == dets.device
        np.testing.assert_array_equal(inds.cpu().numpy(), g[f'keep_{k}'], err_msg=f'case {k}')
        np.testing.assert_array_equal(kept.cpu().numpy(), g[f'dets_{k}'][g[f'keep_{k}']])


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 127, 128, 129, 777, 2000, 4096, 4097, 6000])
def test_nms3d_vs_oracle_random(dev, n):
    from mr3d_amd.mmdet_ops.nms import nms_cuda
    rng = np.random.default_rng(n)
    nc = max(1, n // 6)
    ctr = rng.uniform([10, 10, 4], [500, 500, 76], size=(nc, 3))[rng.integers(0, nc, n)] + rng.normal(0, 1.5, (n, 3))
    half = rng.uniform([3, 3, 1.5], [14, 14, 6], size=(nc, 3))[rng.integers(0, nc, n)] * rng.uniform(0.8, 1.2, (n, 3))
    boxes = np.stack([ctr[:, 0] - half[:, 0], ctr[:, 1] - half[:, 1], ctr[:, 0] + half[:, 0], ctr[:, 1] + half[:, 1],
                      ctr[:, 2] - half[:, 2], ctr[:, 2] + half[:, 2]], 1)
    scores = rng.uniform(0, 1, n)
    dets = np.concatenate([boxes, scores[:, None]], 1).astype(np.float32)
    for thr in (0.3, 0.7):
        inds = nms_cuda.nms_3d(torch.from_numpy(dets).to(dev), thr)
        np.testing.assert_array_equal(inds.cpu().numpy(), orc.nms3d(dets, thr))


def test_nms3d_ties_duplicates_and_strided_input(dev):
    from mr3d_amd.mmdet_ops.nms import nms_cuda
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 100, size=(40, 3))
    boxes = np.concatenate([base, base + rng.uniform(5, 20, size=(40, 3))], 1)[:, [0, 1, 3, 4, 2, 5]]
    dets = np.concatenate([np.tile(boxes, (5, 1)), np.zeros((200, 1))], 1).astype(np.float32)
    dets[:, 6] = np.repeat([0.5, 0.25, 0.5, 0.75, 0.5], 40)          # many equal scores, exact duplicates
    exp = orc.nms3d(dets, 0.5)
    got = nms_cuda.nms_3d(torch.from_numpy(dets).to(dev), 0.5)
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
    wide = torch.zeros((200, 11), device=dev)
    wide[:, 2:9] = torch.from_numpy(dets).to(dev)
    got2 = nms_cuda.nms_3d(wide[:, 2:9], 0.5)                         # row stride 11, not contiguous
    np.testing.assert_array_equal(got2.cpu().numpy(), exp)


def test_nms3d_empty_and_strict_threshold(dev):
    from mr3d_amd.mmdet_ops import nms
    kept, inds = nms(torch.zeros((0, 7), device=dev), 0.5)
    assert inds.numel() == 0 and inds.dtype == torch.long
    a = torch.tensor([[0, 0, 3, 3, 0, 0, 0.9], [0, 0, 3, 1, 0, 0, 0.8]], device=dev)   # IoU == 0.5 exactly
    assert nms(a, 0.5)[1].tolist() == [0, 1]
    assert nms(a, 0.4999)[1].tolist() == [0]
    # numpy in -> numpy out with device_id (nms_wrapper.py:29-31,50-51)
    k, i = nms(a.cpu().numpy(), 0.4999, device_id=0)
    assert isinstance(i, np.ndarray) and i.tolist() == [0] and k.shape == (1, 7)


def test_nms3d_all_suppressed_and_none_suppressed_full_size(dev):
    """size-independent properties at the benchmark size (N = 2000)."""
    from mr3d_amd.mmdet_ops.nms import nms_cuda
    n = 2000
    one = np.array([[10, 10, 50, 50, 4, 20, 0]], np.float32)
    dets = np.tile(one, (n, 1)); dets[:, 6] = np.linspace(0.1, 0.9, n)
    got = nms_cuda.nms_3d(torch.from_numpy(dets).to(dev), 0.7)
    assert got.tolist() == [n - 1]                                      # only the best survives
    g = np.arange(n, dtype=np.float32)
    far = np.stack([g * 100, g * 0, g * 100 + 10, g * 0 + 10, g * 0, g * 0 + 4, np.linspace(0.9, 0.1, n)], 1).astype(np.float32)
    got = nms_cuda.nms_3d(torch.from_numpy(far).to(dev), 0.7)
    assert got.tolist() == list(range(n))                               # disjoint boxes: all kept, ascending
    # idempotence: NMS of the kept set keeps everything
    rng = np.random.default_rng(1)
    d = np.concatenate([rng.uniform(0, 60, (n, 3)), rng.uniform(0, 1, (n, 4))], 1).astype(np.float32)
    d = np.stack([d[:, 0], d[:, 1], d[:, 0] + 20 * d[:, 3] + 4, d[:, 1] + 20 * d[:, 4] + 4, d[:, 2], d[:, 2] + 8 * d[:, 5] + 2, d[:, 6]], 1)
    t = torch.from_numpy(d.astype(np.float32)).to(dev)
    k1 = nms_cuda.nms_3d(t, 0.5)
    k2 = nms_cuda.nms_3d(t[k1], 0.5)
    assert k2.tolist() == list(range(k1.numel()))
    np.testing.assert_array_equal(k1.cpu().numpy(), orc.nms3d(d.astype(np.float32), 0.5))


# ------------------------------------------------------------------ RoIAlign3D: <= 1e-4 rel
RTOL, ATOL = 1e-4, 1e-5      # north-star tolerance for floating point (BASELINE.json)


def _feat(rng, shape, fmt, dev):
    f = torch.from_numpy(rng.normal(size=shape).astype(np.float32)).to(dev)
    return f.contiguous(memory_format=fmt)


@pytest.mark.parametrize('fmt', [torch.contiguous_format, torch.channels_last_3d])
@pytest.mark.parametrize('C,P,sample_num', [(64, (3, 7, 7), 2), (64, (10, 14, 14), 2), (5, (2, 3, 4), 0),
                                            (70, (3, 7, 7), 1), (16, (1, 1, 1), 3)])
def test_roi_align3d_forward_backward_vs_oracle(dev, fmt, C, P, sample_num):
    from mr3d_amd.mmdet_ops import RoIAlign3D
    rng = np.random.default_rng(C * 100 + P[0])
    N, D, H, W = 2, 10, 24, 20
    feat = _feat(rng, (N, C, D, H, W), fmt, dev)
    rois = ri.rand_rois(rng, 33, N, (W * 4, H * 4, D * 2), max_xy=90, max_z=20)
    rois[0, 1:] = [-20, -20, 30, 30, -6, 8]
    rois[1, 1:] = [70, 80, 200, 200, 15, 60]
    rois[2, 1:] = [10, 10, 10, 10, 3, 3]
    rois[3, 1:] = [30, 30, 20, 20, 5, 2]          # malformed: end < start -> extent clamped (:241-243)
    fnp = feat.cpu().numpy()                       # logical NCDHW view
    for bug in (True, False):
        out_size = P[1] if P[1] == P[2] else (P[1], P[2])
        layer = RoIAlign3D(out_size, P[0], 0.25, 0.5, sample_num=sample_num, bug_compat=bug)
        f = feat.clone().requires_grad_(True)
        out = layer(f, torch.from_numpy(rois).to(dev))
        exp = orc.roi_align3d_forward(fnp, rois, *P, 0.25, 0.5, sample_num)
        if sample_num == 0:
            ok = np.isfinite(exp)                  # zero-extent roi with adaptive sampling is 0/0 = NaN (Appendix A.1)
            assert np.array_equal(np.isfinite(out.detach().cpu().numpy()), ok)
        else:
            ok = np.ones_like(exp, bool)
        np.testing.assert_allclose(out.detach().cpu().numpy()[ok], exp[ok], rtol=RTOL, atol=ATOL)
        top = rng.normal(size=exp.shape).astype(np.float32)
        out.backward(torch.from_numpy(top).to(dev))
        expb = orc.roi_align3d_backward(top, rois, fnp.shape, 0.25, 0.5, sample_num, bug_compat=bug)
        assert f.grad.is_contiguous(memory_format=fmt)
        # a voxel gradient is a sum of many +- atomics (order differs from the serial oracle): tolerance is
        # 1e-4 relative to the gradient's scale, not to each (possibly cancelling) element
        np.testing.assert_allclose(f.grad.cpu().numpy(), expb, rtol=RTOL, atol=RTOL * np.abs(expb).max())


@pytest.mark.parametrize('fmt', [torch.contiguous_format, torch.channels_last_3d])
def test_roi_align3d_double_and_half_dispatch(dev, fmt):
    """The reference dispatches float, double and half (roi_align_kernel.cu:323,674): features, rois and output share the type.
    double: against the float64 restatement (forward) and its autograd gradient (the correct backward); half: against the
    fp32 oracle on the half-rounded inputs, at half resolution."""
    from mr3d_amd.mmdet_ops import RoIAlign3D
    rng = np.random.default_rng(11)
    N, C, D, H, W = 2, 6, 8, 20, 18
    P = (2, 5, 4)
    feat32 = rng.normal(size=(N, C, D, H, W)).astype(np.float32)
    rois = ri.rand_rois(rng, 21, N, (W * 4, H * 4, D * 2), max_xy=70, max_z=14)
    rois[0, 1:] = [-20, -20, 30, 30, -6, 8]
    rois[1, 1:] = [10, 10, 10, 10, 3, 3]
    top = rng.normal(size=(21, C) + P).astype(np.float32)
    # ---- double
    f = torch.from_numpy(feat32.astype(np.float64)).to(dev).contiguous(memory_format=fmt).requires_grad_(True)
    layer = RoIAlign3D((P[1], P[2]), P[0], 0.25, 0.5, sample_num=2, bug_compat=False)
    out = layer(f, torch.from_numpy(rois.astype(np.float64)).to(dev))
    assert out.dtype == torch.float64
    exp = ri.roi_align3d_f64(feat32, rois, *P, 0.25, 0.5, 2)
    np.testing.assert_allclose(out.detach().cpu().numpy(), exp, rtol=1e-12, atol=1e-12)
    out.backward(torch.from_numpy(top.astype(np.float64)).to(dev))
    ft = torch.from_numpy(feat32.astype(np.float64)).requires_grad_(True)
    ri.roi_align3d_torch(ft, torch.from_numpy(rois), *P, 0.25, 0.5, 2).backward(torch.from_numpy(top.astype(np.float64)))
    assert f.grad.dtype == torch.float64 and f.grad.is_contiguous(memory_format=fmt)
    np.testing.assert_allclose(f.grad.cpu().numpy(), ft.grad.numpy(), rtol=1e-11, atol=1e-11)
    # ---- half (roi coordinates are kept to values a half holds exactly: integers below 2048)
    rois_h = np.round(rois).astype(np.float32)
    feat_h = feat32.astype(np.float16)
    fh = torch.from_numpy(feat_h).to(dev).contiguous(memory_format=fmt).requires_grad_(True)
    for bug in (True, False):
        layer = RoIAlign3D((P[1], P[2]), P[0], 0.25, 0.5, sample_num=2, bug_compat=bug)
        fh.grad = None
        outh = layer(fh, torch.from_numpy(rois_h.astype(np.float16)).to(dev))
        assert outh.dtype == torch.float16
        exph = orc.roi_align3d_forward(feat_h.astype(np.float32), rois_h, *P, 0.25, 0.5, 2)
        np.testing.assert_allclose(outh.detach().float().cpu().numpy(), exph, rtol=2e-3, atol=2e-3)
        top_h = top.astype(np.float16)
        outh.backward(torch.from_numpy(top_h).to(dev))
        expb = orc.roi_align3d_backward(top_h.astype(np.float32), rois_h, feat_h.shape, 0.25, 0.5, 2, bug_compat=bug)
        # every accumulated contribution is rounded to half: error ~ (number of contributions) x half ulp of the sum
        np.testing.assert_allclose(fh.grad.float().cpu().numpy(), expb, rtol=2e-2, atol=2e-2 * np.abs(expb).max())
    with pytest.raises(RuntimeError):            # mixed types: the reference reads everything as the features' scalar_t
        layer(fh, torch.from_numpy(rois_h).to(dev))


def test_roi_align3d_benchmark_shapes_properties(dev):
    """Full BASELINE sizes (512 rois x 64 ch x (3,7,7) on a P2-sized map): properties that need no oracle run:
    a constant feature map pools to the same constant wherever all taps are in range, and the
    two layouts agree with each other."""
    from mr3d_amd.mmdet_ops import RoIAlign3D
    rng = np.random.default_rng(2)
    N, C, D, H, W = 1, 64, 40, 128, 128
    rois = ri.rand_rois(rng, 512, N, (512, 512, 80))
    r = torch.from_numpy(rois).to(dev)
    layer = RoIAlign3D(7, 3, 0.25, 0.5, sample_num=2)
    const = torch.full((N, C, D, H, W), 1.5, device=dev)
    np.testing.assert_allclose(layer(const, r).cpu().numpy(), 1.5, rtol=1e-6)
    f = torch.randn((N, C, D, H, W), device=dev)
    a = layer(f, r)
    b = layer(f.contiguous(memory_format=torch.channels_last_3d), r)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=RTOL, atol=ATOL)
    # linearity in the features
    f2 = torch.randn_like(f)
    c = layer(2 * f + f2, r)
    np.testing.assert_allclose(c.cpu().numpy(), (2 * a + layer(f2, r)).cpu().numpy(), rtol=1e-4, atol=1e-4)
    # sampled check against the oracle on a subset of rois
    sub = rois[:16]
    exp = orc.roi_align3d_forward(f.cpu().numpy(), sub, 3, 7, 7, 0.25, 0.5, 2)
    np.testing.assert_allclose(a[:16].cpu().numpy(), exp, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('P', [(3, 7, 7), (10, 14, 14)])
@pytest.mark.parametrize("form", [0, 1, 2, 4, 5, 6, 8])
def test_roi_align3d_row_streaming_forward_paths(dev, P, form):
    """the round-6 forward kernels (row streaming roi_align3d_fwd_rows; planes folded first roi_align3d_fwd_fold: form 0 takes it for
    (3,7,7), form 8 for both shapes, form 2 = row streaming for both) against the oracle on the row shapes that take their different
    paths: one-piece rows (<= 4 voxels), unrolled k-step counts (<= 16 voxels), long rows alone in a double-size ring buffer
    (17 .. 32 / 64 voxels), rows wider than the ring (bin-by-bin fallback), footprints with untouched rows (bins of > 2 voxels),
    rois hanging over every border, a degenerate roi, two images, 128 channels; every (chunk, ring) instance behind
    MR3D_TUNE_ROIALIGN_FORM and the round-5 kernels (form 1) give the same answer"""
    from mr3d_amd import _hip
    rng = np.random.default_rng(100 + P[0])
    N, C, D, H, W = 2, 128, 12, 40, 300
    feat = _feat(rng, (N, C, D, H, W), torch.channels_last_3d, dev)
    rois = np.array([[0, 8, 8, 19, 40, 2, 10],          # 4 voxels wide: one piece per row
                     [1, 8, 8, 60, 150, 1, 20],         # 14 voxels wide
                     [0, 100, 20, 200, 60, 4, 9],       # 26 voxels: long row (LPB 4) / unrolled (LPB 8)
                     [1, 300, 10, 520, 30, 0, 6],       # 56 voxels: long row in both ring sizes
                     [0, 100, 40, 700, 70, 3, 8],       # 151 voxels: wider than the ring
                     [1, 40, 2, 60, 158, 0, 23],        # 39 rows, bins of 5.6 voxels: untouched rows in between
                     [0, -30, -20, 50, 30, -6, 8],      # over the low borders
                     [1, 1100, 120, 1250, 170, 15, 40],  # over the high borders
                     [0, 10, 10, 10, 10, 3, 3],         # one pixel
                     [1, 30, 30, 20, 20, 5, 2]], np.float32)   # malformed: end < start
    exp = orc.roi_align3d_forward(feat.cpu().numpy(), rois, *P, 0.25, 0.5, 2)
    out = torch.zeros(exp.shape, device=dev)
    _hip.lib().mr3d_tuning_set(7, form)
    try:
        _hip.roi_align3d_forward(feat, torch.from_numpy(rois).to(dev), *P, 0.25, 0.5, 2, out)
    finally:
        _hip.lib().mr3d_tuning_set(7, 0)
    # 1e-4 relative, the absolute floor tied to the map's scale as in the full-size test: sample coordinates reach x = 175 here, where one
    # fp32 ulp of a coordinate (1.5e-5) is 3e-5 of a bin built from N(0, 1) voxels -- every kernel of the file, the round-5 ones
    # included, sits within 5e-5 of the oracle on the 56- and 151-voxel rows and within 3e-7 of each other
    np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=RTOL, atol=2e-5 * np.abs(exp).max())


@pytest.mark.parametrize('P', [(3, 7, 7), (10, 14, 14), (16, 14, 14), (17, 14, 14)])     # (pooled depth 17: beyond the kernel, the round-5 one takes it)
@pytest.mark.parametrize('form', [0, 1])
def test_roi_align3d_plane_owner_backward_paths(dev, P, form):
    """the plane-owner backward (round 6, roi_align3d_bwd_planes) against the oracle on the footprints that take its different paths:
    a few voxels (fewer planes than workgroups), bins of several voxels (planes no pooled depth reaches, rows no pooled row reaches),
    footprints beyond its dense weight tables (> 128 rows or columns: weights formed on the fly), rois over every border, a
    degenerate and a malformed roi, two images, 128 channels, the reference's top-gradient index bug on and off, the maximum
    pooled depth of the kernel; form 1 = the round-5 kernels on the same inputs"""
    from mr3d_amd import _hip
    rng = np.random.default_rng(200 + P[0])
    N, C, D, H, W = 2, 128, 12, 140, 160
    rois = np.array([[0, 8, 8, 19, 40, 2, 10],
                     [1, 8, 8, 60, 150, 1, 20],
                     [0, 2, 3, 630, 550, 0, 22],         # 158 x 138 x 12 voxels: beyond both weight tables
                     [1, 100, 20, 700, 60, 4, 9],        # 151 columns, 11 rows
                     [0, 40, 2, 60, 300, 0, 47],         # 75 rows, bins of 5 voxels along y
                     [1, -30, -20, 50, 30, -6, 8],
                     [0, 600, 500, 700, 580, 18, 30],
                     [1, 10, 10, 10, 10, 3, 3],
                     [0, 30, 30, 20, 20, 5, 2]], np.float32)
    top = rng.normal(size=(len(rois), C) + P).astype(np.float32)
    for bug in (True, False):
        if bug and (P[0] - 1) * P[0] * P[2] + P[1] * P[2] > P[0] * P[1] * P[2]:
            continue                                     # (the bug's offsets leave the roi's block: mmdet_ops routes that case elsewhere)
        exp = orc.roi_align3d_backward(top, rois, (N, C, D, H, W), 0.25, 0.5, 2, bug_compat=bug)
        grad = torch.zeros((N, C, D, H, W), device=dev).contiguous(memory_format=torch.channels_last_3d)
        _hip.lib().mr3d_tuning_set(7, form)
        try:
            _hip.roi_align3d_backward(torch.from_numpy(top).to(dev), torch.from_numpy(rois).to(dev), *P, 0.25, 0.5, 2, grad, bug_compat=bug)
        finally:
            _hip.lib().mr3d_tuning_set(7, 0)
        np.testing.assert_allclose(grad.cpu().numpy(), exp, rtol=RTOL, atol=RTOL * np.abs(exp).max())


def test_roi_align3d_error_conventions(dev):
    from mr3d_amd.mmdet_ops.roi_align import roi_align_cuda
    f = torch.zeros((1, 4, 4, 8, 8), device=dev)
    out = torch.zeros((2, 4, 3, 7, 7), device=dev)
    with pytest.raises(RuntimeError):
        roi_align_cuda.forward3d(f, torch.zeros((2, 5), device=dev), 3, 7, 7, 0.25, 0.5, 2, out)    # wrong roi size
    with pytest.raises(RuntimeError):
        roi_align_cuda.forward3d(f.cpu(), torch.zeros((2, 7), device=dev), 3, 7, 7, 0.25, 0.5, 2, out)
    with pytest.raises(RuntimeError):
        roi_align_cuda.forward3d(f.permute(0, 1, 3, 2, 4), torch.zeros((2, 7), device=dev), 3, 7, 7, 0.25, 0.5, 2, out)
    assert roi_align_cuda.forward3d(f, torch.zeros((0, 7), device=dev), 3, 7, 7, 0.25, 0.5, 2, out[:0]) == 1


def test_single_roi_extractor_levels_golden_and_forward(dev, golden_dir):
    """SingleRoIExtractor: level map against the reference's map_roi_levels (golden), forward against per-level oracle calls"""
    from mr3d_amd.detector import SingleRoIExtractor
    g = np.load(f'{golden_dir}/roi_levels.npz')
    ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=7, out_size_depth=3, sample_num=2), 8, [4, 8, 16, 32], [2, 4, 8, 16])
    lv = ext.map_roi_levels(torch.from_numpy(g['rois']).to(dev), 4)
    np.testing.assert_array_equal(lv.cpu().numpy(), g['levels'])
    rng = np.random.default_rng(8)
    feats_np = [rng.normal(size=(1, 8, d, h, h)).astype(np.float32) for d, h in ((20, 48), (10, 24), (5, 12), (3, 6))]
    rois = g['rois'][:60].copy()
    rois[:, [1, 3]] = np.clip(rois[:, [1, 3]], 0, 191); rois[:, [2, 4]] = np.clip(rois[:, [2, 4]], 0, 191); rois[:, [5, 6]] = np.clip(rois[:, [5, 6]], 0, 39)
    for fmt in (torch.contiguous_format, torch.channels_last_3d):
        feats = [torch.from_numpy(f).to(dev).contiguous(memory_format=fmt) for f in feats_np]
        out = ext(feats, torch.from_numpy(rois).to(dev)).cpu().numpy()
        lvl = ext.map_roi_levels(torch.from_numpy(rois), 4).numpy()
        for i in range(4):
            sel = np.nonzero(lvl == i)[0]
            if len(sel):
                exp = orc.roi_align3d_forward(feats_np[i], rois[sel], 3, 7, 7, 1 / [4, 8, 16, 32][i], 1 / [2, 4, 8, 16][i], 2)
                np.testing.assert_allclose(out[sel], exp, rtol=RTOL, atol=ATOL)


def test_nms3d_batched_equals_per_set_calls():
    """mr3d_nms3d_batched: the 5 pyramid levels of one image (rpn_head_3d.py:81-135) in one launch sequence; every
    segment must give exactly the index list of its own mr3d_nms3d call (and of the oracle), empty and 1-box sets
    included"""
    import mr3d_amd  # noqa: F401
    from mr3d_amd import _hip
    from oracle import binding as ob
    rng = np.random.default_rng(21)
    dev = torch.device('cuda:0')
    sets = []
    for n in (2000, 0, 1, 777, 64, 1280, 192):
        b = rng.uniform(0, 200, (n, 3))
        sz = rng.uniform(4, 40, (n, 3))
        d = np.concatenate([b[:, :2], b[:, :2] + sz[:, :2], b[:, 2:], b[:, 2:] + sz[:, 2:], rng.random((n, 1))], 1).astype(np.float32)
        sets.append(d)
    dl = [torch.from_numpy(d).to(dev) for d in sets]
    got = _hip.nms3d_batched(dl, 0.3)
    for d, t, g in zip(sets, dl, got):
        single = _hip.nms3d(t, 0.3).cpu().numpy()
        assert np.array_equal(g.cpu().numpy(), single)
        if len(d):
            assert np.array_equal(single, ob.nms3d(d, 0.3))


def test_multilevel_roialign_level_rule_and_backward(dev, golden_dir):
    """the one-launch multi-level extractor: (1) the in-kernel level rule picks the level the reference's map_roi_levels
    picks, for every golden roi (features = constant level id make the choice visible); (2) forward and backward equal
    the per-level path (NCDHW features take it) on mixed-size rois"""
    from mr3d_amd.detector import SingleRoIExtractor
    g = np.load(f'{golden_dir}/roi_levels.npz')
    ext = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=7, out_size_depth=3, sample_num=2), 64, [4, 8, 16, 32], [2, 4, 8, 16])
    rois = torch.from_numpy(g['rois']).to(dev)
    const_feats = [torch.full((1, 64, d, h, h), float(i + 1), device=dev).contiguous(memory_format=torch.channels_last_3d)
                   for i, (d, h) in enumerate(((40, 128), (20, 64), (10, 32), (5, 16)))]
    out = ext(const_feats, rois)
    # a roi entirely outside its level's map yields 0; inside, the level id: take the max over the bins
    seen = out.amax(dim=(1, 2, 3, 4)).cpu().numpy()
    inside = seen > 0
    assert inside.mean() > 0.5
    assert np.array_equal(seen[inside].astype(np.int64) - 1, g['levels'][inside])

    rng = np.random.default_rng(5)
    r = g['rois'][:200].copy()
    r[:, [1, 3]] = np.clip(r[:, [1, 3]], 0, 511); r[:, [2, 4]] = np.clip(r[:, [2, 4]], 0, 511); r[:, [5, 6]] = np.clip(r[:, [5, 6]], 0, 79)
    rt = torch.from_numpy(r).to(dev)
    feats_np = [rng.normal(size=(1, 64, d, h, h)).astype(np.float32) for d, h in ((40, 128), (20, 64), (10, 32), (5, 16))]
    top = torch.from_numpy(rng.normal(size=(200, 64, 3, 7, 7)).astype(np.float32)).to(dev)
    res = {}
    for name, fmt in (('ml', torch.channels_last_3d), ('per_level', torch.contiguous_format)):
        feats = [torch.from_numpy(f).to(dev).contiguous(memory_format=fmt).requires_grad_(True) for f in feats_np]
        o = ext(feats, rt)
        o.backward(top)
        res[name] = (o.detach().cpu().numpy(), [f.grad.cpu().numpy() for f in feats])
    np.testing.assert_allclose(res['ml'][0], res['per_level'][0], rtol=RTOL, atol=ATOL)
    for a, b in zip(res['ml'][1], res['per_level'][1]):
        assert np.abs(a - b).max() <= 1e-4 * max(np.abs(b).max(), 1e-6)


def test_compiled_extension_and_ctypes_forms_agree(dev):
    """the PYBIND11 modules (at::Tensor arguments, TORCH_CHECKs) and the ctypes forms reach the same C ABI: identical NMS3D
    index lists, identical RoIAlign3D outputs and gradients in both layouts and for the half / double dispatch"""
    from mr3d_amd.mmdet_ops.roi_align import roi_align_cuda, roi_align_ctypes
    from mr3d_amd.mmdet_ops.nms import nms_cuda, nms_ctypes
    assert roi_align_cuda.__file__.endswith('.so') and nms_cuda.__file__.endswith('.so')
    rng = np.random.default_rng(21)
    n = 700
    ctr = rng.uniform([10, 10, 4], [200, 200, 40], size=(n, 3))
    half = rng.uniform([4, 4, 2], [24, 24, 8], size=(n, 3))
    dets = np.concatenate([ctr[:, :1] - half[:, :1], ctr[:, 1:2] - half[:, 1:2], ctr[:, :1] + half[:, :1], ctr[:, 1:2] + half[:, 1:2],
                           ctr[:, 2:] - half[:, 2:], ctr[:, 2:] + half[:, 2:], rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    t = torch.from_numpy(dets).to(dev)
    a, b = nms_cuda.nms_3d(t, 0.5), nms_ctypes.nms_3d(t, 0.5)
    assert a.dtype == torch.long and a.device == t.device and torch.equal(a, b)
    assert np.array_equal(a.cpu().numpy(), orc.nms3d(dets, 0.5))
    wide = torch.zeros((n, 11), device=dev)
    wide[:, 2:9] = t
    assert torch.equal(nms_cuda.nms_3d(wide[:, 2:9], 0.5), a)                  # row stride 11
    rois = torch.tensor([[0, 4, 6, 60, 70, 2, 12], [1, 30, 20, 90, 80, 0, 15], [0, 0, 0, 20, 20, 0, 4]], dtype=torch.float32, device=dev)
    for dtype in (torch.float32, torch.float64, torch.float16):
        for fmt in (torch.contiguous_format, torch.channels_last_3d):
            f = torch.randn(2, 64, 8, 24, 24, device=dev).to(dtype).contiguous(memory_format=fmt)
            r = rois.to(dtype)
            o1 = torch.zeros(3, 64, 3, 7, 7, device=dev, dtype=dtype)
            o2 = torch.zeros_like(o1)
            assert roi_align_cuda.forward3d(f, r, 3, 7, 7, 0.25, 0.5, 2, o1) == 1
            assert roi_align_ctypes.forward3d(f, r, 3, 7, 7, 0.25, 0.5, 2, o2) == 1
            assert torch.equal(o1, o2)
            top = torch.randn_like(o1)
            for bug in (True, False):
                g1, g2 = torch.zeros_like(f), torch.zeros_like(f)
                roi_align_cuda.backward3d(top, r, 3, 7, 7, 0.25, 0.5, 2, g1, bug_compat=bug)
                roi_align_ctypes.backward3d(top, r, 3, 7, 7, 0.25, 0.5, 2, g2, bug_compat=bug)
                # same kernels, same inputs; only the order of the atomic adds differs between two launches
                tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.float16: 4e-3}[dtype]
                assert (g1.double() - g2.double()).abs().max().item() <= tol * max(1.0, g2.double().abs().max().item())
    with pytest.raises(RuntimeError):
        roi_align_cuda.forward3d(f, torch.zeros((2, 5), device=dev, dtype=f.dtype), 3, 7, 7, 0.25, 0.5, 2, o1)       # wrong roi size
    with pytest.raises(RuntimeError):
        roi_align_cuda.forward3d(f.permute(0, 1, 3, 2, 4), r, 3, 7, 7, 0.25, 0.5, 2, o1)                             # not contiguous


def test_nms3d_long_dependency_chains_and_negative_threshold(dev):
    """The chain sweep resolves block after block through published kept words: a set in which every box's fate depends on its
    predecessor's (a row of boxes, each overlapping only its neighbours above the threshold) exercises every hand-over, and a
    negative threshold (every pair with IoU >= 0 'overlaps', nms_kernel.cu:122 is a plain `>`) the no-skip instance."""
    from mr3d_amd.mmdet_ops.nms import nms_cuda
    for n in (64, 65, 1000, 2048):
        x0 = np.arange(n, dtype=np.float32) * 2.0
        dets = np.stack([x0, np.zeros(n), x0 + 9, np.full(n, 9.0), np.zeros(n), np.full(n, 9.0), np.linspace(0.9, 0.1, n)], 1).astype(np.float32)
        for thr in (0.5, 0.6, 0.25):
            exp = orc.nms3d(dets, thr)
            got = nms_cuda.nms_3d(torch.from_numpy(dets).to(dev), thr)
            np.testing.assert_array_equal(got.cpu().numpy(), exp)
        # the same row in reverse score order: the chain runs from the last box to the first
        dets[:, 6] = np.linspace(0.1, 0.9, n)
        np.testing.assert_array_equal(nms_cuda.nms_3d(torch.from_numpy(dets).to(dev), 0.5).cpu().numpy(), orc.nms3d(dets, 0.5))
    rng = np.random.default_rng(3)
    d = np.concatenate([rng.uniform(0, 100, (300, 3)), rng.uniform(0, 1, (300, 4))], 1)
    d = np.stack([d[:, 0], d[:, 1], d[:, 0] + 10 * d[:, 3] + 2, d[:, 1] + 10 * d[:, 4] + 2, d[:, 2], d[:, 2] + 5 * d[:, 5] + 1, d[:, 6]], 1).astype(np.float32)
    got = nms_cuda.nms_3d(torch.from_numpy(d).to(dev), -0.5)
    np.testing.assert_array_equal(got.cpu().numpy(), orc.nms3d(d, -0.5))
    assert got.numel() == 1                                              # IoU >= 0 > -0.5 for every pair: only the best survives

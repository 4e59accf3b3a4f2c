"""GPU parity tests of the proposal path: box codec, IoU matrix, anchors against the
reference-generated golden vectors; RPN proposal generation and multiclass NMS against a restatement that uses
the C oracle for NMS3D."""
import numpy as np
import pytest
import torch

from oracle import binding as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def test_delta2bbox3d_bbox2delta3d_golden(dev, golden_dir):
    from mr3d_amd.core import delta2bbox3D, bbox2delta3d, bbox2roi3D
    g = np.load(f'{golden_dir}/codec.npz')
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ms = tuple(int(v) for v in g['max_shape'])
    rois = T(g['rois'])
    for key, means, stds, shape, deltas in (
            ('dec_rpn', [0.] * 6, [1.] * 6, ms, g['deltas']),
            ('dec_rcnn', g['means'].tolist(), g['stds'].tolist(), ms, g['deltas']),
            ('dec_noclip', g['means'].tolist(), g['stds'].tolist(), None, g['deltas']),
            ('dec12', g['means'].tolist(), g['stds'].tolist(), ms, g['deltas12'])):
        out = delta2bbox3D(rois, T(deltas), means, stds, shape)
        # exp/log on the device differ from torch's CPU vectorised versions by an ulp or two (<= 1e-6 rel)
        np.testing.assert_allclose(out.cpu().numpy(), g[key], rtol=2e-6, atol=2e-4, err_msg=key)
        np.testing.assert_allclose(out.cpu().numpy(), orc.delta2bbox3d(g['rois'], deltas, means, stds, shape), rtol=2e-6, atol=2e-4)
    enc = bbox2delta3d(rois, T(g['gt']), g['means'].tolist(), g['stds'].tolist())
    np.testing.assert_allclose(enc.cpu().numpy(), g['enc'], rtol=2e-5, atol=2e-5)
    r7 = bbox2roi3D([rois[:100], rois[100:100], rois[100:]])
    np.testing.assert_array_equal(r7.cpu().numpy(), g['rois7'])


def test_bbox_overlaps_golden(dev, golden_dir):
    from mr3d_amd.core import bbox_overlaps
    g = np.load(f'{golden_dir}/iou3d.npz')
    T = lambda a: torch.from_numpy(a).to(dev)
    iou = bbox_overlaps(T(g['a']), T(g['b']))
    np.testing.assert_array_equal(iou.cpu().numpy(), g['iou'])                  # same fp32 op order: bit-equal
    ka = bbox_overlaps(T(g['ka1']), T(g['ka2'])).diagonal().cpu().numpy()
    assert [round(float(v), 4) for v in ka] == [1.0, 0.798, 0.7899, 0.6125]     # geometry.py:81-101
    assert bbox_overlaps(T(g['a'][:0]), T(g['b'])).shape == (0, 53)
    big = torch.rand(3, 6, device=dev)
    big[:, [2, 3, 5]] += 1.0
    anchors = torch.rand(70000, 6, device=dev) * 50
    anchors[:, [2, 3, 5]] = anchors[:, [0, 1, 4]] + torch.rand(70000, 3, device=dev) * 20
    got = bbox_overlaps(big * 30, anchors).cpu().numpy()
    exp = orc.iou3d_matrix((big * 30).cpu().numpy(), anchors[:200].cpu().numpy())
    np.testing.assert_array_equal(got[:, :200], exp)


def test_anchor_generator_golden(dev, golden_dir):
    from mr3d_amd.core import AnchorGenerator3D
    g = np.load(f'{golden_dir}/anchors.npz')
    strides, dstrides = [4, 8, 16, 32, 64], [2, 4, 8, 16, 32]
    for tag, scale in (('s2', 2), ('s3', 3)):
        for lv, (s, ds) in enumerate(zip(strides, dstrides)):
            gen = AnchorGenerator3D(s, [scale], [scale], [1.0], ds)
            np.testing.assert_array_equal(gen.base_anchors.numpy(), g[f'base_{tag}_{lv}'])
    gen = AnchorGenerator3D(4, [2], [2], [1.0], 2)
    fs = tuple(int(v) for v in g['grid_fs'])
    np.testing.assert_array_equal(gen.grid_anchors(fs, 4, 2, device=dev).cpu().numpy(), g['grid'])
    np.testing.assert_array_equal(gen.valid_flags(fs, tuple(int(v) for v in g['valid_size']), device=dev).cpu().numpy(), g['valid'])
    # multi-anchor generator keeps the (h, w, d, a) order
    gen2 = AnchorGenerator3D(8, [2, 4], [2, 4], [1.0], 4)
    got = gen2.grid_anchors((2, 3, 4), 8, 4, device=dev).cpu()
    base = gen2.base_anchors
    k = 0
    for h in range(3):
        for w in range(4):
            for d in range(2):
                for a in range(base.size(0)):
                    sh = torch.tensor([w * 8, h * 8, w * 8, h * 8, d * 4, d * 4], dtype=torch.float32)
                    assert torch.equal(got[k], base[a] + sh)
                    k += 1


def _ref_get_bboxes_single(cls_scores, bbox_preds, gens, strides, dstrides, img_shape, cfg, pos_indices=None):
    """line-by-line torch/CPU restatement of rpn_head_3d.py:72-149 with the C oracle as `nms`"""
    mlvl = []
    for lvl, (cls, reg) in enumerate(zip(cls_scores, bbox_preds)):
        A, D, H, W = cls.shape
        anchors = gens[lvl].grid_anchors((D, H, W), strides[lvl], dstrides[lvl], device=cls.device).cpu()
        scores = cls.cpu().permute(2, 3, 1, 0).reshape(-1).sigmoid()
        pred = reg.cpu().permute(2, 3, 1, 0).reshape(-1, 6)
        if cfg['nms_pre'] > 0 and scores.shape[0] > cfg['nms_pre']:
            if pos_indices is not None and pos_indices[lvl].shape == scores.shape:       # :97-101
                m = pos_indices[lvl].cpu().bool()
                scores, pred, anchors = scores[m], pred[m], anchors[m]
            if scores.shape[0] > cfg['nms_pre']:
                scores, top = scores.topk(cfg['nms_pre'])
                pred, anchors = pred[top], anchors[top]
        props = torch.from_numpy(orc.delta2bbox3d(anchors.numpy(), pred.numpy(), [0.] * 6, [1.] * 6, img_shape))
        dets = torch.cat([props, scores[:, None]], 1)
        keep = orc.nms3d(dets.numpy(), cfg['nms_thr'])
        mlvl.append(dets[torch.from_numpy(keep)][:cfg['nms_post']])
    props = torch.cat(mlvl, 0)
    num = min(cfg['max_num'], props.shape[0])
    _, top = props[:, 6].topk(num)
    return props[top]


@pytest.mark.parametrize('nms_post', [700, 1500])       # < nms_pre: the path with host-side counts; == nms_pre: the sync-free device path
@pytest.mark.parametrize('channels_last', [False, True, 'flags', 'few_flags'])
def test_rpn_get_bboxes_single_vs_restatement(dev, channels_last, nms_post):
    from mr3d_amd.core import AnchorGenerator3D
    from mr3d_amd.detector import get_bboxes_single
    torch.manual_seed(11)
    strides, dstrides = [4, 8, 16], [2, 4, 8]
    gens = [AnchorGenerator3D(s, [2], [2], [1.0], ds) for s, ds in zip(strides, dstrides)]
    sizes = [(10, 32, 32), (5, 16, 16), (3, 8, 8)]
    img_shape = (128, 128, 3, 20)
    cls, reg = [], []
    for (d, h, w) in sizes:
        c = torch.randn(1, d, h, w, device=dev) * 2
        r = torch.randn(6, d, h, w, device=dev) * 0.4
        if channels_last is True:    # what the conv engine hands over: [C, D, H, W] views of an NDHWC buffer
            r = r.permute(1, 2, 3, 0).contiguous().permute(3, 0, 1, 2)
        cls.append(c); reg.append(r)
    cfg = dict(nms_across_levels=False, nms_pre=1500, nms_post=nms_post, max_num=1000, nms_thr=0.7, min_bbox_size=0)
    flags = None
    if channels_last == 'flags':     # the reference's cached inside-flags (rpn_head_3d.py:97-101): random 70 % here
        flags = [(torch.rand(d * h * w, device=dev) < 0.7).to(torch.uint8) for (d, h, w) in sizes]
    if channels_last == 'few_flags':
        # fewer than nms_pre anchors of the large level are flagged in (10240 * 0.1 < 1500): the reference takes ALL of them,
        # unsorted, and nothing else (rpn_head_3d.py:95-112); the masked surplus picks must not reach the output
        flags = [(torch.rand(d * h * w, device=dev) < 0.1).to(torch.uint8) for (d, h, w) in sizes]
        assert int(flags[0].sum()) < 1500
    got = get_bboxes_single(cls, reg, gens, strides, dstrides, img_shape, cfg, valid_flags=flags)
    exp = _ref_get_bboxes_single(cls, reg, gens, strides, dstrides, img_shape, cfg, pos_indices=flags)
    assert got.shape == exp.shape
    if nms_post >= cfg['nms_pre']:
        # the device path itself: a fixed [max_num, 7] block, zero rows behind the device-side count
        from types import SimpleNamespace
        from mr3d_amd.detector.rpn_proposals import get_bboxes_single_device, _device_path_ok
        assert _device_path_ok(cls, SimpleNamespace(**cfg))
        blk, cnt = get_bboxes_single_device(cls, reg, gens, strides, dstrides, img_shape, cfg, valid_flags=flags)
        assert tuple(blk.shape) == (cfg['max_num'], 7) and cnt.dtype == torch.int32 and int(cnt) == got.size(0)
        assert torch.equal(blk[:int(cnt)], got) and (blk[int(cnt):] == 0).all()
    # same proposals, score-descending; sigmoid saturates, so equal fp32 scores occur and torch.topk orders ties
    # differently on host and device: compare in a canonical order (score desc, then coordinates).  Coordinates
    # differ by device-vs-host expf ulps only.
    def canon(a):
        key = np.lexsort((np.round(a[:, 1], 2), np.round(a[:, 0], 2), -a[:, 6]))
        return a[key]
    g, e = got.cpu().numpy(), exp.numpy()
    assert np.all(np.diff(g[:, 6]) <= 0)
    np.testing.assert_allclose(canon(g), canon(e), rtol=2e-6, atol=2e-4)


def test_pair_proposals_one_nms_call_equals_two_separate_calls(dev):
    """rpn_proposals.get_bboxes_pair_device: both pathways of the dual-pathway detector (two_stage_3d_2scales.py:170-176) through
    ONE NMS3D launch sequence over their 2 x L level sets -- every row and both counts equal the per-pathway calls exactly."""
    from mr3d_amd.core import AnchorGenerator3D
    from mr3d_amd.detector.rpn_proposals import get_bboxes_pair_device, get_bboxes_single_device
    torch.manual_seed(5)
    strides, dstrides = [4, 8, 16], [2, 4, 8]
    gens = [AnchorGenerator3D(s, [2], [2], [1.0], ds) for s, ds in zip(strides, dstrides)]
    cfg = dict(nms_across_levels=False, nms_pre=1500, nms_post=1500, max_num=1000, nms_thr=0.7, min_bbox_size=0)
    means, stds = (0.,) * 6, (1.,) * 6

    def pathway(sizes, img_shape, flag_p):
        cls = [torch.randn(1, d, h, w, device=dev) * 2 for d, h, w in sizes]
        reg = [(torch.randn(6, d, h, w, device=dev) * 0.4).permute(1, 2, 3, 0).contiguous().permute(3, 0, 1, 2) for d, h, w in sizes]
        flags = None if flag_p is None else [(torch.rand(d * h * w, device=dev) < flag_p).to(torch.uint8) for d, h, w in sizes]
        return (cls, reg, gens, strides, dstrides, img_shape, means, stds), flags

    a, fa = pathway([(10, 32, 32), (5, 16, 16), (3, 8, 8)], (128, 128, 3, 20), 0.7)
    b, fb = pathway([(15, 48, 48), (8, 24, 24), (4, 12, 12)], (192, 192, 3, 30), None)      # the 1.5x pathway
    (pa, ca), (pb, cb) = get_bboxes_pair_device(a, b, cfg, fa, fb)
    for args, flags, p, c in ((a, fa, pa, ca), (b, fb, pb, cb)):
        ep, ec = get_bboxes_single_device(*args[:6], cfg, args[6], args[7], valid_flags=flags)
        assert int(c) == int(ec) and int(c) > 0
        assert torch.equal(p, ep)


def test_multiclass_nms_3d_vs_restatement(dev):
    from mr3d_amd.core import multiclass_nms_3d
    rng = np.random.default_rng(4)
    n = 900
    ctr = rng.uniform([10, 10, 4], [300, 300, 60], (40, 3))[rng.integers(0, 40, n)] + rng.normal(0, 2, (n, 3))
    half = rng.uniform([4, 4, 2], [14, 14, 6], (n, 3))
    b = np.stack([ctr[:, 0] - half[:, 0], ctr[:, 1] - half[:, 1], ctr[:, 0] + half[:, 0], ctr[:, 1] + half[:, 1],
                  ctr[:, 2] - half[:, 2], ctr[:, 2] + half[:, 2]], 1).astype(np.float32)
    boxes12 = np.concatenate([b + 1000, b], 1)                          # class-0 slot is never read
    scores = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    got_b, got_l = multiclass_nms_3d(torch.from_numpy(boxes12).to(dev), torch.from_numpy(scores).to(dev), 0.2,
                                     dict(type='nms', iou_thr=0.5), 200)
    sel = scores[:, 1] > 0.2
    dets = np.concatenate([b[sel], scores[sel, 1:2]], 1)
    keep = orc.nms3d(dets, 0.5)
    kept = dets[keep]
    order = np.argsort(-kept[:, 6], kind='stable')[:200]
    np.testing.assert_array_equal(got_b.cpu().numpy(), kept[order])
    assert got_l.dtype == torch.long and int(got_l.abs().sum()) == 0 and got_l.shape[0] == len(order)
    # shared 6-column boxes + nothing above threshold -> empty (k,7) / (0,) outputs (bbox_nms.py:102-104)
    e_b, e_l = multiclass_nms_3d(torch.from_numpy(b).to(dev), torch.from_numpy(scores * 0).to(dev), 0.2,
                                 dict(type='nms', iou_thr=0.5), 200)
    assert e_b.shape == (0, 7) and e_l.shape == (0,)


def _ref_multiclass(b_cols, scores, thr, iou_thr, max_num):
    """bbox_nms.py:57-106 restated with the NMS3D oracle: per class filter -> nms -> concat -> optional top-k"""
    n, C = scores.shape
    dets_all, labels_all = [], []
    for c in range(1, C):
        sel = scores[:, c] > thr
        if not sel.any():
            continue
        bx = b_cols[sel, :6] if b_cols.shape[1] == 6 else b_cols[sel, 6 * c:6 * c + 6]
        dets = np.concatenate([bx, scores[sel, c:c + 1]], 1).astype(np.float32)
        kept = dets[orc.nms3d(dets, iou_thr)]
        dets_all.append(kept); labels_all.append(np.full(len(kept), c - 1, np.int64))
    if not dets_all:
        return np.zeros((0, 7), np.float32), np.zeros((0,), np.int64)
    d, l = np.concatenate(dets_all), np.concatenate(labels_all)
    if d.shape[0] > max_num:
        order = np.argsort(-d[:, 6], kind='stable')
        order = order[:max_num]                      # max_num = -1: drops the last row, like the reference's slice
        d, l = d[order], l[order]
    return d, l


@pytest.mark.parametrize('max_num', [10000, 150, -1])
@pytest.mark.parametrize('shared_boxes', [False, True])
def test_multiclass_nms_3d_several_classes(dev, max_num, shared_boxes):
    from mr3d_amd.core import multiclass_nms_3d
    rng = np.random.default_rng(9)
    n, C = 700, 4
    ctr = rng.uniform([10, 10, 4], [200, 200, 50], (30, 3))[rng.integers(0, 30, n)] + rng.normal(0, 2, (n, 3))
    half = rng.uniform([4, 4, 2], [14, 14, 6], (n, 3))
    b = np.stack([ctr[:, 0] - half[:, 0], ctr[:, 1] - half[:, 1], ctr[:, 0] + half[:, 0], ctr[:, 1] + half[:, 1],
                  ctr[:, 2] - half[:, 2], ctr[:, 2] + half[:, 2]], 1).astype(np.float32)
    cols = b if shared_boxes else np.concatenate([b + 500 * c + rng.normal(0, 1, b.shape).astype(np.float32) * (c > 0)
                                                  for c in range(C)], 1).astype(np.float32)
    scores = rng.permutation(n * C).reshape(n, C).astype(np.float32) / (n * C)          # distinct scores
    scores[:, 3] *= 0.1                                                                  # class 3: nothing above threshold
    got_b, got_l = multiclass_nms_3d(torch.from_numpy(cols).to(dev), torch.from_numpy(scores).to(dev), 0.3,
                                     dict(type='nms', iou_thr=0.5), max_num)
    exp_b, exp_l = _ref_multiclass(cols, scores, 0.3, 0.5, max_num)
    np.testing.assert_array_equal(got_b.cpu().numpy(), exp_b)
    np.testing.assert_array_equal(got_l.cpu().numpy(), exp_l)
    with pytest.raises(NotImplementedError):
        multiclass_nms_3d(torch.from_numpy(cols).to(dev), torch.from_numpy(scores).to(dev), 0.3, dict(type='soft_nms', iou_thr=0.5), 10)


def test_multiclass_nms_3d_beyond_the_fused_ops_limits(dev):
    """more rows than the one-launch op ranks in LDS (4096): the class-by-class walk over the device NMS3D gives the restated
    reference flow's result too (e.g. a test_cfg whose two pathways bring 2 x 2500 rows)"""
    from mr3d_amd.core import multiclass_nms_3d
    from mr3d_amd import _hip
    rng = np.random.default_rng(4)
    n, C = 5000, 3
    assert not _hip.multiclass_nms3d_fits(n, C) and _hip.multiclass_nms3d_fits(4000, 2)
    ctr = rng.uniform([10, 10, 4], [400, 400, 70], (200, 3))[rng.integers(0, 200, n)] + rng.normal(0, 2, (n, 3))
    half = rng.uniform([4, 4, 2], [14, 14, 6], (n, 3))
    b = np.stack([ctr[:, 0] - half[:, 0], ctr[:, 1] - half[:, 1], ctr[:, 0] + half[:, 0], ctr[:, 1] + half[:, 1],
                  ctr[:, 2] - half[:, 2], ctr[:, 2] + half[:, 2]], 1).astype(np.float32)
    scores = rng.permutation(n * C).reshape(n, C).astype(np.float32) / (n * C)
    for max_num in (10000, 300):
        got_b, got_l = multiclass_nms_3d(torch.from_numpy(b).to(dev), torch.from_numpy(scores).to(dev), 0.5,
                                         dict(type='nms', iou_thr=0.5), max_num)
        exp_b, exp_l = _ref_multiclass(b, scores, 0.5, 0.5, max_num)
        np.testing.assert_array_equal(got_b.cpu().numpy(), exp_b)
        np.testing.assert_array_equal(got_l.cpu().numpy(), exp_l)


@pytest.mark.parametrize('n,k,frac', [(655360, 2000, 0.7), (81920, 2000, 1.0), (10240, 2000, 0.05), (5000, 2000, 0.0), (8192, 4096, 0.9)])
def test_rpn_select_topk_is_an_exact_topk(dev, n, k, frac):
    """mr3d_rpn_select_topk (radix select on the logits): the selected set equals torch.topk's on the masked scores wherever
    the k-th score is not tied, comes out in ascending anchor order, carries sigmoid(logit), and reports
    min(k, #valid) rows; strided (channels-last) logit maps give the same selection."""
    from mr3d_amd import _hip
    g = torch.Generator(device='cpu').manual_seed(n + k)
    A, D = 1, 8
    W = 16
    H = n // (A * D * W)
    n = A * D * H * W
    logits = (torch.randn(A, D, H, W, generator=g) * 3).to(dev)
    flags = (torch.rand(n, generator=g) < frac).to(dev) if frac < 1.0 else None
    small = torch.randn(1, 2, 4, 4, generator=g).to(dev)                  # a level with fewer anchors than k: all 32 taken
    for maps in ([logits, small], [logits.permute(1, 2, 3, 0).contiguous().permute(3, 0, 1, 2), small]):
        caps, flat, scores, counts = _hip.rpn_select_topk(maps, [flags, None], k)
        assert caps == [min(n, k), 32]
        c = counts.cpu().numpy()
        s_flat = maps[0].permute(2, 3, 1, 0).reshape(-1).sigmoid()           # the reference's flat order
        lg_flat = maps[0].permute(2, 3, 1, 0).reshape(-1)
        nvalid = n if flags is None else int(flags.sum())
        assert c[0] == min(k, nvalid) and c[1] == 32
        sel = flat[:c[0]]
        assert (sel[1:] > sel[:-1]).all()                                      # ascending anchor order, no duplicates
        if flags is not None:
            assert flags[sel].all()
        np.testing.assert_allclose(scores[:c[0]].cpu().numpy(), s_flat[sel].cpu().numpy(), rtol=2e-6, atol=1e-7)
        if c[0] == k:
            masked = lg_flat if flags is None else torch.where(flags, lg_flat, lg_flat.new_full((), -1e30))
            ref_v, ref_i = masked.topk(k)
            kth = ref_v[-1]
            assert (lg_flat[sel] >= kth).all()                                # nothing below the k-th logit
            strict = set(ref_i[ref_v > kth].tolist())
            assert strict <= set(sel.tolist())                                # everything strictly above it
        assert torch.equal(flat[caps[0]:caps[0] + 32], torch.arange(32, device=dev))
        np.testing.assert_allclose(scores[caps[0]:].cpu().numpy(), small.permute(2, 3, 1, 0).reshape(-1).sigmoid().cpu().numpy(), rtol=2e-6)

"""GPU parity of the training-target kernels (csrc/assign_ops.hip, loss_ops.hip) through the C ABI against the
reference's own outputs (tests/golden/targets.npz, losses.npz) and the numpy oracle on larger seeded inputs.
Index / label / weight outputs: bit-exact.  IoU: bit-exact fp32.  Deltas (logf): <= 2e-6 abs.  Losses: rtol 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import targets_ref as T

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def tg():
    return np.load(os.path.join(HERE, 'golden', 'targets.npz'))


@pytest.fixture(scope='module')
def lg():
    return np.load(os.path.join(HERE, 'golden', 'losses.npz'))


def draws_of(g, tag, num_pos_expected, pos_candidates):
    ds = [g[f'{tag}_draw{k}'] for k in range(int(g[f'{tag}_num_draws']))]
    dp = dn = np.zeros(0, np.int64)
    if pos_candidates > num_pos_expected:
        dp, ds = ds[0], ds[1:]
    if ds:
        dn = ds[0]
    return dp, dn


def unpad(lst, cnt):
    lst, cnt = lst.cpu().numpy(), int(cnt)
    assert (lst[cnt:] == -1).all()
    return lst[:cnt]


@pytest.mark.parametrize('tag', ['rpn_shipped', 'rpn_manypos', 'rpn_border'])
def test_anchor_targets_match_reference(dev, tg, tag):
    from mr3d_amd import _hip
    pos_thr, neg_thr, min_pos, num, frac, pos_weight, border = tg[f'{tag}_cfg']
    anchors_np, gt_np = tg[f'{tag}_anchors'], tg[f'{tag}_gt']
    inside_np = tg[f'{tag}_inside_flags'].astype(bool)
    anchors, gt = torch.from_numpy(anchors_np).to(dev), torch.from_numpy(gt_np).to(dev)
    inside = torch.from_numpy(inside_np).to(dev)
    assigned, max_ov, _ = _hip.max_iou_assign(anchors, gt, pos_thr, neg_thr, min_pos, inside=inside)
    a = assigned.cpu().numpy()
    assert np.array_equal(a[inside_np], tg[f'{tag}_assigned_gt_inds'])            # the reference's compacted view
    assert (a[~inside_np] == -1).all()
    assert np.array_equal(max_ov.cpu().numpy()[inside_np], tg[f'{tag}_max_overlaps'])   # bit-exact fp32
    dp, dn = draws_of(tg, tag, int(num * frac), int((a > 0).sum()))
    pos, neg, counts, sel = _hip.random_sample(assigned, int(num), float(frac), -1, dp, dn, want_sel=True)
    c = counts.cpu().numpy()
    # the reference indexes the INSIDE subset; map its lists to full-set indices
    full = np.nonzero(inside_np)[0]
    assert np.array_equal(unpad(pos, c[0]), full[tg[f'{tag}_pos_inds']])
    assert np.array_equal(unpad(neg, c[1]), full[tg[f'{tag}_neg_inds']])
    assert c[2] == (a > 0).sum() and c[3] == (a == 0).sum()
    labels, lw, bt, bw = _hip.anchor_target(anchors, assigned, sel, gt, [0.] * 6, [1.] * 6, float(pos_weight))
    assert np.array_equal(labels.cpu().numpy(), tg[f'{tag}_labels'])
    assert np.array_equal(lw.cpu().numpy(), tg[f'{tag}_label_weights'])
    assert np.array_equal(bw.cpu().numpy(), tg[f'{tag}_bbox_weights'])
    np.testing.assert_allclose(bt.cpu().numpy(), tg[f'{tag}_bbox_targets'], rtol=0, atol=2e-6)


@pytest.mark.parametrize('tag', ['rcnn_shipped', 'rcnn_fewpos', 'rcnn_small'])
def test_rcnn_sampling_and_targets_match_reference(dev, tg, tag):
    from mr3d_amd import _hip
    pos_thr, neg_thr, min_pos, num, frac, pos_weight = tg[f'{tag}_cfg']
    props = torch.from_numpy(tg[f'{tag}_proposals']).to(dev)               # [P,7] with score: stride 7
    gt, gl = torch.from_numpy(tg[f'{tag}_gt']).to(dev), torch.from_numpy(tg[f'{tag}_gt_labels']).to(dev)
    assigned, max_ov, labels = _hip.max_iou_assign(props, gt, pos_thr, neg_thr, min_pos, gt_labels=gl)
    assert np.array_equal(assigned.cpu().numpy(), tg[f'{tag}_assigned_gt_inds'])
    assert np.array_equal(max_ov.cpu().numpy(), tg[f'{tag}_max_overlaps'])
    assert np.array_equal(labels.cpu().numpy(), tg[f'{tag}_assigned_labels'])
    G = gt.size(0)
    allv = _hip.concat_gt_assign(assigned, G)
    dp, dn = draws_of(tg, tag, int(num * frac), int((allv > 0).sum()))
    pos, neg, counts, _ = _hip.random_sample(allv, int(num), float(frac), -1, dp, dn)
    c = counts.cpu().numpy()
    assert np.array_equal(unpad(pos, c[0]), tg[f'{tag}_pos_inds'])
    assert np.array_equal(unpad(neg, c[1]), tg[f'{tag}_neg_inds'])
    cap = int(num)
    rois, lab, lw, bt, bw, pgi = _hip.bbox_target3d(props, gt, gl, allv, pos, neg, counts, cap, 0, [0.] * 6,
                                                    tg[f'{tag}_target_stds'], float(pos_weight))
    r = int(c[0] + c[1])
    assert np.array_equal(rois.cpu().numpy()[:r], tg[f'{tag}_rois'])
    assert np.array_equal(lab.cpu().numpy()[:r], tg[f'{tag}_labels']) and (lab.cpu().numpy()[r:] == 0).all()
    assert np.array_equal(lw.cpu().numpy()[:r], tg[f'{tag}_label_weights']) and (lw.cpu().numpy()[r:] == 0).all()
    assert np.array_equal(bw.cpu().numpy()[:r], tg[f'{tag}_bbox_weights']) and (bw.cpu().numpy()[r:] == 0).all()
    np.testing.assert_allclose(bt.cpu().numpy()[:r], tg[f'{tag}_bbox_targets'], rtol=0, atol=2e-5)
    assert np.array_equal(pgi.cpu().numpy()[:c[0]], tg[f'{tag}_pos_assigned_gt_inds']) and (pgi.cpu().numpy()[c[0]:] == -1).all()
    # the list-in / tensor-out form of the reference (bbox_target.py:118-145), a call into the same kernel: identical rows
    from mr3d_amd.core.bbox import bbox_target_single_3d
    npos = int(c[0])
    gi = pgi[:npos].clamp(min=0)
    lab2, lw2, bt2, bw2 = bbox_target_single_3d(rois[:npos, 1:], rois[npos:r, 1:], gt[gi], gl[gi], dict(pos_weight=float(pos_weight)),
                                                target_stds=tg[f'{tag}_target_stds'])
    assert torch.equal(lab2, lab[:r]) and torch.equal(lw2, lw[:r]) and torch.equal(bt2, bt[:r]) and torch.equal(bw2, bw[:r])
    e = rois[:0, 1:]
    lab0, lw0, bt0, bw0 = bbox_target_single_3d(e, rois[npos:r, 1:], e, gl[:0], dict(pos_weight=float(pos_weight)))
    assert lab0.numel() == r - npos and (lab0 == 0).all() and (lw0 == 1).all() and (bw0 == 0).all()


def test_assign_and_sample_full_size_vs_oracle(dev):
    """BASELINE configs[1] scale: 748,992 anchors, 12 gts; oracle on the same seeded inputs, and the size-independent
    properties: positives/negatives disjoint, sorted, unique, within the candidate sets, counts consistent."""
    from mr3d_amd import _hip
    from mr3d_amd.core.anchor import AnchorGenerator3D
    rng = np.random.default_rng(3)
    lv = []
    for st, ds, fs in zip([4, 8, 16, 32, 64], [2, 4, 8, 16, 32], [(40, 128, 128), (20, 64, 64), (10, 32, 32), (5, 16, 16), (3, 8, 8)]):
        lv.append(AnchorGenerator3D(st, [2], [2], [1.0], ds).grid_anchors(fs, st, ds, device=dev))
    anchors = torch.cat(lv)
    assert anchors.size(0) == 748992
    an = anchors.cpu().numpy()
    pick = rng.integers(0, len(an), 12)
    gt_np = (an[pick] + rng.normal(0, 1.0, (12, 6))).astype(np.float32)
    gt_np[0] = an[pick[0]]                                   # an exact hit
    gt = torch.from_numpy(gt_np).to(dev)
    inside_np = T.anchor_inside_flags(an, np.ones(len(an), np.uint8), (512, 512, 3, 80), 0)
    assigned, max_ov, _ = _hip.max_iou_assign(anchors, gt, 0.7, 0.3, 0.3, inside=torch.from_numpy(inside_np).to(dev))
    ra, rm, _ = T.max_iou_assign(an[inside_np], gt_np, 0.7, 0.3, 0.3)
    a = assigned.cpu().numpy()
    assert np.array_equal(a[inside_np], ra) and np.array_equal(max_ov.cpu().numpy()[inside_np], rm)
    dp = rng.integers(0, 2 ** 31 - 1, 128)
    dn = rng.integers(0, 2 ** 31 - 1, 256)
    pos, neg, counts, sel = _hip.random_sample(assigned, 256, 0.5, -1, dp, dn, want_sel=True)
    c = counts.cpu().numpy()
    rp, rn = T.random_sample(a, 256, 0.5, -1, dp, dn)
    p, n = unpad(pos, c[0]), unpad(neg, c[1])
    assert np.array_equal(p, rp) and np.array_equal(n, rn)
    assert (np.diff(p) > 0).all() and (np.diff(n) > 0).all() and (a[p] > 0).all() and (a[n] == 0).all()
    s = sel.cpu().numpy()
    assert (s == 1).sum() == c[0] and (s == 2).sum() == c[1] and c[0] + c[1] <= 256


def test_assigner_rejects_empty_and_unsupported(dev):
    from mr3d_amd import _hip
    boxes = torch.zeros((4, 6), device=dev)
    with pytest.raises(ValueError):
        _hip.max_iou_assign(boxes, torch.zeros((0, 6), device=dev), 0.5, 0.5, 0.5)
    with pytest.raises(_hip.Mr3dError):
        _hip.max_iou_assign(boxes, torch.ones((1, 6), device=dev), 0.5, 0.5, 0.5, gt_max_assign_all=False)


def test_losses_match_reference(dev, lg):
    from mr3d_amd import _hip
    t = lambda k: torch.from_numpy(lg[k]).to(dev)
    close = lambda a, b, tol=1e-5: np.testing.assert_allclose(a.cpu().numpy().reshape(-1), np.asarray(b).reshape(-1), rtol=tol, atol=1e-7)
    l, g = _hip.weighted_bce_logits(t('bce_score'), t('bce_labels'), t('bce_weights'), float(lg['bce_avg']))
    close(l, lg['bce_loss']); close(g, lg['bce_grad'])
    avg_dev = torch.tensor([float(lg['bce_avg'])], device=dev)                       # normaliser resident on the device
    l2, g2 = _hip.weighted_bce_logits(t('bce_score'), t('bce_labels'), t('bce_weights'), avg_dev)
    assert torch.equal(l, l2) and torch.equal(g, g2)                                 # and deterministic
    l, g = _hip.weighted_smoothl1(t('sl1_pred'), t('sl1_target'), t('sl1_weights'), float(lg['sl1_beta']), float(lg['sl1_avg']))
    close(l, lg['sl1_loss']); close(g, lg['sl1_grad'])
    l, acc, g = _hip.weighted_softmax_ce(t('ce_score'), t('ce_labels'), t('ce_weights'))
    close(l, lg['ce_loss']); close(g, lg['ce_grad']); close(acc, lg['ce_acc'])
    # bbox head regression loss: dense form with zero weights on the negatives == the reference's pos-row gather
    R = lg['reg_pred'].shape[0]
    l, g = _hip.weighted_smoothl1(t('reg_pred'), t('reg_target'), t('reg_weights'), 1.0, float(R))
    close(l, lg['reg_loss']); close(g, lg['reg_grad'])
    l, g = _hip.mask_bce(t('mask_pred'), t('mask_target'), t('mask_labels'))
    close(l, lg['mask_loss']); close(g, lg['mask_grad'])


def test_losses_with_padding_rows(dev, lg):
    """static-shape form: rows beyond *valid_rows are padding and must not change value, gradient or accuracy"""
    from mr3d_amd import _hip
    t = lambda k: torch.from_numpy(lg[k]).to(dev)
    pad = 37
    score = torch.cat([t('ce_score'), torch.randn(pad, 2, device=dev)])
    labels = torch.cat([t('ce_labels'), torch.zeros(pad, dtype=torch.long, device=dev)])
    weights = torch.cat([t('ce_weights'), torch.zeros(pad, device=dev)])
    valid = torch.tensor([lg['ce_score'].shape[0]], dtype=torch.int32, device=dev)
    l, acc, g = _hip.weighted_softmax_ce(score, labels, weights, valid_rows=valid)
    np.testing.assert_allclose(l.item(), lg['ce_loss'][0], rtol=1e-5)
    np.testing.assert_allclose(acc.item(), lg['ce_acc'][0], rtol=1e-6)
    assert (g[-pad:] == 0).all()
    P = lg['mask_pred'].shape[0]
    mp = torch.cat([t('mask_pred'), torch.randn((3,) + lg['mask_pred'].shape[1:], device=dev)])
    mt = torch.cat([t('mask_target'), torch.zeros((3,) + lg['mask_target'].shape[1:], device=dev)])
    ml = torch.cat([t('mask_labels'), torch.ones(3, dtype=torch.long, device=dev)])
    l, g = _hip.mask_bce(mp, mt, ml, valid_rows=torch.tensor([P], dtype=torch.int32, device=dev))
    np.testing.assert_allclose(l.item(), lg['mask_loss'][0], rtol=1e-5)
    assert (g[P:] == 0).all()

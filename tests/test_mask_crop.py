"""Mask-crop: numpy restatement invariants (CPU) and GPU parity with it."""
import numpy as np
import pytest
import torch

from oracle import mask_ref


def _scene(rng, G=3, D=24, H=48, W=40):
    gt = np.zeros((G, D, H, W), np.uint8)
    for g in range(G):
        c = rng.uniform([6, 10, 10], [D - 6, H - 10, W - 10])
        r = rng.uniform(2.5, 6.0, 3)
        z, y, x = np.ogrid[:D, :H, :W]
        gt[g] = (((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2 <= 1).astype(np.uint8)
    P = 40
    gi = rng.integers(0, G, P)
    boxes = np.zeros((P, 6), np.float32)
    for p in range(P):
        zz, yy, xx = np.nonzero(gt[gi[p]])
        j = rng.normal(0, 1.5, 6)
        boxes[p] = [xx.min() + j[0], yy.min() + j[1], xx.max() + j[2], yy.max() + j[3], zz.min() + j[4], zz.max() + j[5]]
    boxes[0] = [-5.3, -2.1, 12.7, 9.2, -1.5, 4.4]          # leaves the volume on the low side
    boxes[1] = [30.2, 40.9, 60.0, 70.0, 20.1, 40.0]        # leaves it on the high side
    boxes[2] = [10.9, 10.2, 10.1, 10.3, 5.5, 5.6]          # degenerate -> 1 voxel
    return gt, boxes, gi.astype(np.int64)


def test_restatement_invariants():
    rng = np.random.default_rng(0)
    gt, boxes, gi = _scene(rng)
    t = mask_ref.mask_target(gt, boxes, gi, 20, 28, 28)
    assert t.shape == (40, 20, 28, 28) and set(np.unique(t)) <= {0.0, 1.0}
    ones = np.ones_like(gt)
    assert mask_ref.mask_target(ones, boxes, gi, 20, 28, 28).min() == 1.0          # all-ones stays all-ones
    assert mask_ref.mask_target(ones * 0, boxes, gi, 20, 28, 28).max() == 0.0      # all-zeros stays all-zeros
    big = mask_ref.mask_target(gt, boxes, gi, 40, 56, 56)                           # up-scaled targets: no Gaussian widening
    assert big.mean() >= t.mean() - 0.02


def test_target_rule_contains_the_literal_scipy_evaluation():
    """The reference's `255 * resize(crop)` -> astype(uint8) -> > 0 (mask_target.py:41-44), evaluated literally with the
    scipy.ndimage routines scikit-image 0.18 calls, is a SUBSET of the restated rule (every voxel it sets needs all stencil
    voxels set), and equals it whenever scipy's float64 sums of an all-ones stencil come out as exactly 1/255 -- they land
    one ulp below for some interpolation weights, and the truncation then drops the voxel (for some crops all of them)."""
    shapes = [(16, 40, 36), (8, 20, 24), (12, 28, 28), (25, 50, 60), (5, 9, 7)]
    exact = 0
    for shape in shapes:
        zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing='ij')
        crop = ((zz ** 2 + yy ** 2 + xx ** 2) < 0.8).astype(np.uint8)
        lit = mask_ref.mask_target_literal_scipy(crop, (10, 14, 14))
        D, H, W = shape
        box = np.array([[0, 0, W - 1, H - 1, 0, D - 1]], np.float32)
        rule = mask_ref.mask_target(crop[None], box, np.zeros(1, np.int64), 10, 14, 14)[0] > 0
        assert not (lit & ~rule).any(), shape                      # literal evaluation never exceeds the rule
        exact += int(np.array_equal(lit, rule))
    assert exact >= 2                                               # the strongly shrunk crops agree voxel for voxel


@pytest.mark.gpu
def test_mask_target_and_paste_gpu_vs_restatement():
    from mr3d_amd.core import mask_target_single, mask_target, paste_masks_3d
    rng = np.random.default_rng(1)
    gt, boxes, gi = _scene(rng)
    dev = torch.device('cuda:0')
    cfg = dict(mask_size=28, mask_size_depth=20)
    boxes7 = np.concatenate([boxes, rng.uniform(0, 1, (len(boxes), 1)).astype(np.float32)], 1)   # dets carry a score column
    got = mask_target_single(torch.from_numpy(boxes7).to(dev), torch.from_numpy(gi).to(dev), torch.from_numpy(gt), cfg)
    exp = mask_ref.mask_target(gt, boxes, gi, 20, 28, 28)
    mism = (got.cpu().numpy() != exp).mean()
    assert mism == 0, mism                         # an AND over integer stencils: exact
    both = mask_target([torch.from_numpy(boxes[:7]).to(dev), torch.from_numpy(boxes[7:]).to(dev)],
                       [torch.from_numpy(gi[:7]).to(dev), torch.from_numpy(gi[7:]).to(dev)],
                       [torch.from_numpy(gt), torch.from_numpy(gt)], cfg)
    assert torch.equal(both, got)
    assert mask_target_single(torch.zeros((0, 6), device=dev), torch.zeros(0, dtype=torch.long), torch.from_numpy(gt), cfg).shape == (0, 20, 28, 28)
    prob = torch.sigmoid(torch.randn(len(boxes), 20, 28, 28, device=dev) * 2)
    D, H, W = gt.shape[1:]
    pasted = paste_masks_3d(prob, torch.from_numpy(boxes7).to(dev), (D, H, W), 0.25)
    expp = mask_ref.mask_paste(prob.cpu().numpy(), boxes, 0.25, D, H, W)
    assert pasted.dtype == torch.uint8 and pasted.shape == (len(boxes), D, H, W)
    # the kernel evaluates exp() and the double sums on the device: a voxel whose value sits within float32 rounding of the
    # threshold can flip
    assert (pasted.cpu().numpy() != expp).mean() < 1e-5


def _microbleeds(rng, D=32, H=128, W=128):
    """cerebral-microbleed-sized lesions: spheres of 2-5 voxels radius in-plane (2-10 mm at SWI's 0.5-1 mm pixels), 1-3 slices
    of radius in depth (2 mm slices) -- the GTs the reference trains on -- with proposals the rcnn sampler would call
    positive (the GT box itself, as add_gt_as_proposals does, and jittered copies with IoU >= 0.5)"""
    G = 6
    gt = np.zeros((G, D, H, W), np.uint8)
    boxes, gi = [], []
    z, y, x = np.ogrid[:D, :H, :W]
    for g in range(G):
        r_xy, r_z = rng.uniform(2.0, 5.0), rng.uniform(1.0, 3.0)
        c = rng.uniform([r_z + 1, r_xy + 1, r_xy + 1], [D - r_z - 2, H - r_xy - 2, W - r_xy - 2])
        gt[g] = (((z - c[0]) / r_z) ** 2 + ((y - c[1]) / r_xy) ** 2 + ((x - c[2]) / r_xy) ** 2 <= 1).astype(np.uint8)
        zz, yy, xx = np.nonzero(gt[g])
        b = np.array([xx.min(), yy.min(), xx.max(), yy.max(), zz.min(), zz.max()], np.float32)
        boxes.append(b); gi.append(g)
        for _ in range(3):                               # jitter by up to ~10 % of the size per face: IoU stays >= 0.5
            size = np.array([b[2] - b[0] + 1, b[3] - b[1] + 1, b[2] - b[0] + 1, b[3] - b[1] + 1, b[5] - b[4] + 1, b[5] - b[4] + 1])
            boxes.append(b + rng.uniform(-0.1, 0.1, 6).astype(np.float32) * size); gi.append(g)
    return gt, np.stack(boxes).astype(np.float32), np.array(gi, np.int64)


def test_microbleed_sized_gt_yields_positive_mask_target_voxels():
    """The target rule is an AND over the voxels under both stencils (an eroded mask when a crop SHRINKS).  Lesions of the size
    the reference trains on are far smaller than the 20x28x28 target, so every axis is up-sampled and the rule keeps a positive
    core: (1) the GT box itself -- always among the sampled positives (add_gt_as_proposals, configs/...py:113) -- gives a
    non-empty target for EVERY lesion, down to 2 slices x 4 x 4 pixels; (2) so does every IoU >= 0.5 proposal of a lesion
    that spans >= 3 slices.  What the rule cannot rescue (and neither can the reference expression, a subset of it): a proposal
    whose truncated corner (mask_target.py:37-40 `astype(np.int32)`) cuts one of the only two slices of a 2-slice lesion --
    the order-1 taps then always straddle an empty slice.  That case is counted, not hidden."""
    gt, boxes, gi = _microbleeds(np.random.default_rng(7))
    t = mask_ref.mask_target(gt, boxes, gi, 20, 28, 28)
    frac = t.reshape(len(boxes), -1).mean(1)
    exact = frac[::4]                                    # proposals that ARE the GT box
    assert (exact > 0.05).all(), exact
    depth = np.array([np.ptp(np.nonzero(gt[g])[0]) + 1 for g in gi])
    assert depth.min() == 2 and depth.max() >= 4
    assert (frac[depth >= 3] > 0.1).all(), frac          # an ellipsoid fills pi/6 of its box; the AND rule erodes its rim
    assert frac.max() < 0.75
    assert (frac[depth < 3] > 0).mean() >= 0.75          # 2-slice lesions: most jittered proposals still carry signal


@pytest.mark.gpu
def test_microbleed_sized_gt_positive_voxels_gpu():
    from mr3d_amd.core import mask_target_single
    gt, boxes, gi = _microbleeds(np.random.default_rng(7))
    dev = torch.device('cuda:0')
    got = mask_target_single(torch.from_numpy(boxes).to(dev), torch.from_numpy(gi).to(dev), torch.from_numpy(gt),
                             dict(mask_size=28, mask_size_depth=20))
    exp = mask_ref.mask_target(gt, boxes, gi, 20, 28, 28)
    assert np.array_equal(got.cpu().numpy(), exp)
    assert (got.flatten(1).sum(1)[::4] >= 1).all()           # the GT boxes themselves: never an empty target

"""CPU tests: the C oracle against the reference-generated golden vectors and against
independent float64 re-derivations.  """
import numpy as np
import pytest
import torch

from oracle import binding as orc
import ref_impls as ri


def _load(golden_dir, name):
    return np.load(f'{golden_dir}/{name}', allow_pickle=False)


def test_nms3d_oracle_matches_reference_python(golden_dir):
    g = _load(golden_dir, 'nms3d.npz')
    for k in range(int(g['num_cases'])):
        keep = orc.nms3d(g[f'dets_{k}'], float(g[f'thr_{k}']))
        assert keep.dtype == np.int64
        np.testing.assert_array_equal(keep, g[f'keep_{k}'], err_msg=f'case {k}')


def test_nms3d_numpy_restatement_matches_reference_python(golden_dir):
    """oracle/nms_ref.py (the CPU baseline timed by bench.py) against the same reference-made fixtures"""
    from oracle.nms_ref import nms3d_numpy
    g = _load(golden_dir, 'nms3d.npz')
    for k in range(int(g['num_cases'])):
        picked = nms3d_numpy(g[f'dets_{k}'], float(g[f'thr_{k}']))
        np.testing.assert_array_equal(np.sort(picked), g[f'keep_{k}'], err_msg=f'case {k}')
    assert nms3d_numpy(np.zeros((0, 7)), 0.5).shape == (0,)


def test_nms3d_edge_cases():
    assert orc.nms3d(np.zeros((0, 7), np.float32), 0.5).shape == (0,)
    one = np.array([[0, 0, 4, 4, 0, 4, 0.3]], np.float32)
    np.testing.assert_array_equal(orc.nms3d(one, 0.5), [0])
    # identical boxes: IoU == 1 > thr -> only the best score survives; ties -> lower index first
    same = np.tile(one, (5, 1)); same[:, 6] = [0.1, 0.9, 0.9, 0.2, 0.3]
    np.testing.assert_array_equal(orc.nms3d(same, 0.5), [1])
    # strict '>' : IoU exactly equal to thr is NOT suppressed (nms_kernel.cu:122)
    a = np.array([[0, 0, 3, 3, 0, 0, 0.9], [0, 0, 3, 1, 0, 0, 0.8]], np.float32)   # IoU = 8/16 = 0.5
    np.testing.assert_array_equal(orc.nms3d(a, 0.5), [0, 1])
    np.testing.assert_array_equal(orc.nms3d(a, 0.4999), [0])


def test_nms3d_mask_is_consistent_with_sweep(golden_dir):
    g = _load(golden_dir, 'nms3d.npz')
    dets, thr = g['dets_8'], float(g['thr_8'])
    order = np.argsort(-dets[:, 6], kind='stable')
    sd = dets[order]
    mask = orc.nms3d_mask(sd, thr)
    n, cb = mask.shape
    remv = np.zeros(cb, np.uint64)
    keep = []
    for i in range(n):
        if not (int(remv[i // 64]) >> (i % 64)) & 1:
            keep.append(order[i])
            remv |= mask[i]
    np.testing.assert_array_equal(np.sort(keep), g['keep_8'])


def test_iou3d_known_answers_and_matrix(golden_dir):
    g = _load(golden_dir, 'iou3d.npz')
    got = np.array([orc.iou3d(a, b) for a, b in zip(g['ka1'], g['ka2'])])
    assert [round(float(v), 4) for v in got] == [1.0, 0.798, 0.7899, 0.6125]   # geometry.py:81-101
    np.testing.assert_array_equal(got.astype(np.float32), g['ka'])
    m = orc.iou3d_matrix(g['a'], g['b'])
    np.testing.assert_allclose(m, g['iou'], rtol=0, atol=0)


def test_delta2bbox3d_matches_reference(golden_dir):
    g = _load(golden_dir, 'codec.npz')
    ms = tuple(int(v) for v in g['max_shape'])
    for key, means, stds, shape, deltas in (
            ('dec_rpn', [0.] * 6, [1.] * 6, ms, g['deltas']),
            ('dec_rcnn', g['means'], g['stds'], ms, g['deltas']),
            ('dec_noclip', g['means'], g['stds'], None, g['deltas']),
            ('dec12', g['means'], g['stds'], ms, g['deltas12'])):
        out = orc.delta2bbox3d(g['rois'], deltas, means, stds, shape)
        np.testing.assert_allclose(out, g[key], rtol=2e-6, atol=2e-4, err_msg=key)


@pytest.mark.parametrize('P,sample_num', [((3, 7, 7), 2), ((10, 14, 14), 2), ((2, 3, 4), 0), ((1, 1, 1), 3)])
def test_roi_align3d_forward_oracle_vs_f64(P, sample_num):
    rng = np.random.default_rng(3)
    N, C, D, H, W = 2, 5, 10, 24, 20
    feat = rng.normal(size=(N, C, D, H, W)).astype(np.float32)
    rois = ri.rand_rois(rng, 24, N, (W * 4, H * 4, D * 2), max_xy=90, max_z=20)
    rois[0, 1:] = [-20, -20, 30, 30, -6, 8]          # partly outside (taps < -1 dropped, clamp at 0)
    rois[1, 1:] = [70, 80, 200, 200, 15, 60]         # beyond the far edges
    rois[2, 1:] = [10, 10, 10, 10, 3, 3]             # single voxel
    got = orc.roi_align3d_forward(feat, rois, *P, 0.25, 0.5, sample_num)
    exp = ri.roi_align3d_f64(feat, rois, *P, 0.25, 0.5, sample_num)
    np.testing.assert_allclose(got, exp, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('bug', [False, True])
def test_roi_align3d_backward_oracle_vs_autograd(bug):
    rng = np.random.default_rng(4)
    N, C, D, H, W = 2, 3, 6, 12, 10
    P = (3, 7, 7)
    rois = ri.rand_rois(rng, 9, N, (W * 4, H * 4, D * 2), max_xy=40, max_z=12)
    top = rng.normal(size=(9, C, *P)).astype(np.float32)
    got = orc.roi_align3d_backward(top, rois, (N, C, D, H, W), 0.25, 0.5, 2, bug_compat=bug)
    feat = torch.zeros((N, C, D, H, W), dtype=torch.float64, requires_grad=True)
    out = ri.roi_align3d_torch(feat, torch.from_numpy(rois), *P, 0.25, 0.5, 2)
    t = ri.bug_permute_top(top) if bug else top
    out.backward(torch.from_numpy(t.astype(np.float64)))
    np.testing.assert_allclose(got, feat.grad.numpy(), rtol=1e-4, atol=2e-5)
    if bug:   # the two modes must really differ for PD < PH
        fixed = orc.roi_align3d_backward(top, rois, (N, C, D, H, W), 0.25, 0.5, 2, bug_compat=False)
        assert np.abs(fixed - got).max() > 1e-3


def test_conv3d_oracle_vs_torch():
    rng = np.random.default_rng(9)
    x = rng.normal(size=(1, 3, 5, 9, 8)).astype(np.float32)
    w = rng.normal(size=(4, 3, 3, 3, 3)).astype(np.float32)
    b = rng.normal(size=(4,)).astype(np.float32)
    for stride, pad in (((1, 1, 1), (1, 1, 1)), ((1, 2, 2), (1, 1, 1)), ((2, 2, 2), (0, 1, 1))):
        y = orc.conv3d(x, w, b, stride, pad)
        ref = torch.nn.functional.conv3d(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                                         torch.from_numpy(b).double(), stride, pad).numpy()
        np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)

"""ctypes binding of oracle/libmr3d_oracle.so (see mr3d_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never from the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libmr3d_oracle.so')
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, 'mr3d_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_iou3d.restype = ctypes.c_float
        _lib.oracle_nms3d.restype = ctypes.c_int
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def iou3d(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().oracle_iou3d(_p(a), _p(b)))


def iou3d_matrix(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    L = lib()
    for i in range(a.shape[0]):
        for j in range(b.shape[0]):
            out[i, j] = L.oracle_iou3d(_p(a[i]), _p(b[j]))
    return out


def nms3d(dets, thr):
    dets = _f32(dets)
    n = dets.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    assert dets.shape[1] == 7
    keep = np.empty((n,), np.int64)
    k = lib().oracle_nms3d(_p(dets), ctypes.c_int(n), ctypes.c_int(7), ctypes.c_float(thr), _p(keep, _i64p))
    return keep[:k].copy()


def nms3d_mask(sorted_dets, thr):
    d = _f32(sorted_dets)
    n = d.shape[0]
    cb = (n + 63) // 64
    mask = np.zeros((n, cb), np.uint64)
    lib().oracle_nms3d_mask(_p(d), ctypes.c_int(n), ctypes.c_int(d.shape[1]), ctypes.c_float(thr), _p(mask, _u64p))
    return mask


def roi_align3d_forward(feat, rois, out_d, out_h, out_w, scale, scale_d, sample_num):
    feat, rois = _f32(feat), _f32(rois)
    N, C, D, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, out_d, out_h, out_w), np.float32)
    if R:
        lib().oracle_roi_align3d_forward(_p(feat), _p(rois), R, C, D, H, W, out_d, out_h, out_w,
                                         ctypes.c_float(scale), ctypes.c_float(scale_d), int(sample_num), _p(out))
    return out


def roi_align3d_backward(top, rois, feat_shape, scale, scale_d, sample_num, bug_compat=True):
    top, rois = _f32(top), _f32(rois)
    N, C, D, H, W = feat_shape
    R, C2, PD, PH, PW = top.shape
    assert C2 == C
    bottom = np.zeros((N, C, D, H, W), np.float32)
    if R:
        lib().oracle_roi_align3d_backward(_p(top), _p(rois), R, C, D, H, W, PD, PH, PW, ctypes.c_float(scale),
                                          ctypes.c_float(scale_d), int(sample_num), int(bool(bug_compat)), _p(bottom))
    return bottom


def delta2bbox3d(rois, deltas, means, stds, max_shape=None, wh_ratio_clip=16 / 1000):
    """deltas may be [n, 6k]; mirrors mmdet/core/bbox/transforms.py:105-160."""
    rois, deltas = _f32(rois), _f32(deltas)
    n, k6 = deltas.shape
    k = k6 // 6
    means, stds = _f32(means), _f32(stds)
    out = np.empty((n, k6), np.float32)
    for c in range(k):
        dl = np.ascontiguousarray(deltas[:, 6 * c:6 * c + 6])
        o = np.empty((n, 6), np.float32)
        if max_shape is None:
            lib().oracle_delta2bbox3d(_p(rois), _p(dl), n, _p(means), _p(stds), 0, 0, 0, 0,
                                      ctypes.c_double(wh_ratio_clip), _p(o))
        else:
            lib().oracle_delta2bbox3d(_p(rois), _p(dl), n, _p(means), _p(stds), 1, int(max_shape[0]), int(max_shape[1]),
                                      int(max_shape[3]), ctypes.c_double(wh_ratio_clip), _p(o))
        out[:, 6 * c:6 * c + 6] = o
    return out


def conv3d(x, w, bias, stride, padding):
    x, w = _f32(x), _f32(w)
    N, Ci, D, H, W = x.shape
    Co, _, kd, kh, kw = w.shape
    sd, sh, sw = stride
    pd, ph, pw = padding
    Do, Ho, Wo = (D + 2 * pd - kd) // sd + 1, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    y = np.empty((N, Co, Do, Ho, Wo), np.float32)
    b = None if bias is None else _f32(bias)
    lib().oracle_conv3d(_p(x), _p(w), None if b is None else _p(b), N, Ci, D, H, W, Co, kd, kh, kw, sd, sh, sw,
                        pd, ph, pw, _p(y))
    return y


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))

"""`nms(dets, iou_thr, device_id=None) -> (kept_dets, inds)`: the dispatcher the reference's
heads call (mmdet/ops/nms/nms_wrapper.py:8-52; callers rpn_head_3d.py:134, bbox_nms.py:76-89).

Behaviour kept: tensor or numpy in -> same type out; empty input -> empty long indices;
7 columns -> NMS3D; kept rows come back in INPUT order (inds ascending), not score order.
Deviations (documented in INTEGRATION.md): 5-column (2D) dets and CPU tensors raise
NotImplementedError instead of running the reference's 2D CPU kernel (which, fed 7 columns,
would silently treat z1 as the score: nms_wrapper.py:48, nms_cpu.cpp:12-16).
"""
import numpy as np
import torch

from . import nms_cuda


def nms(dets, iou_thr, device_id=None):
    if isinstance(dets, torch.Tensor):
        as_numpy, dets_t = False, dets
    elif isinstance(dets, np.ndarray):
        as_numpy = True
        if device_id is None:
            raise NotImplementedError('CPU NMS is outside the MI355X hot path: pass device_id or a device tensor')
        dets_t = torch.from_numpy(dets).to(f'cuda:{device_id}')
    else:
        raise TypeError(f'dets must be either a Tensor or numpy array, but got {type(dets)}')

    if dets_t.shape[0] == 0:
        inds = dets_t.new_zeros(0, dtype=torch.long)
    elif not dets_t.is_cuda:
        raise NotImplementedError('CPU NMS is outside the MI355X hot path (the reference has no CPU NMS3D)')
    elif dets_t.shape[1] == 7:
        inds = nms_cuda.nms_3d(dets_t, iou_thr)
    elif dets_t.shape[1] == 5:
        inds = nms_cuda.nms(dets_t, iou_thr)
    else:
        raise ValueError(f'dets must have 7 (3D) columns, got {dets_t.shape[1]}')

    if as_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    # dead in the reference as well: its first statement is breakpoint() (nms_wrapper.py:56)
    raise NotImplementedError('soft_nms (2D, Cython) is outside the 3D hot path')

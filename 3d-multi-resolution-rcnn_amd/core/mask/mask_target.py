"""mask_target / mask_target_single (mmdet/core/mask/mask_target.py:8-50) and the paste half of
FCNMaskHead3D.get_seg_masks (mmdet/models/mask_heads/fcn_mask_head_3d.py:163-189), on the device: no D2H of
boxes, no per-RoI Python loop, no H2D of the result.  Resampling semantics: csrc/mask_ops.hip (parity unpinned:
the reference delegates to skimage.transform.resize, which is not in its tree)."""
import ctypes
from types import SimpleNamespace

import torch

from ... import _hip


def _cfg(cfg):
    return SimpleNamespace(**cfg) if isinstance(cfg, dict) else cfg


def mask_target_single(pos_proposals, pos_assigned_gt_inds, gt_masks, cfg):
    """pos_proposals [P, >=6] fp32 device, pos_assigned_gt_inds [P] int64, gt_masks [G, D, H, W] uint8
    -> [P, mask_size_depth, mask_size, mask_size] fp32 in {0, 1}"""
    cfg = _cfg(cfg)
    ms, msd = int(cfg.mask_size), int(cfg.mask_size_depth)
    P = pos_proposals.size(0)
    if P == 0:
        return pos_proposals.new_zeros((0, msd, ms, ms))
    if not pos_proposals.is_cuda:
        raise NotImplementedError('mask targets are built on the MI355X (no CPU fallback)')
    gt = gt_masks.to(device=pos_proposals.device, dtype=torch.uint8).contiguous()
    boxes = pos_proposals.float().contiguous()
    inds = pos_assigned_gt_inds.to(device=boxes.device, dtype=torch.long).contiguous()
    out = torch.empty((P, msd, ms, ms), dtype=torch.float32, device=boxes.device)
    G, D, H, W = gt.shape
    with torch.cuda.device(boxes.device):
        st = _hip.lib().mr3d_mask_target3d(_hip.ptr(gt), D, H, W, _hip.ptr(boxes), boxes.size(1), _hip.ptr(inds), P, msd, ms, ms,
                                           _hip.ptr(out), _hip.stream_ptr(boxes.device))
    _hip.check(st, 'mr3d_mask_target3d')
    return out


def mask_target(pos_proposals_list, pos_assigned_gt_inds_list, gt_masks_list, cfg):
    return torch.cat([mask_target_single(p, i, g, cfg)
                      for p, i, g in zip(pos_proposals_list, pos_assigned_gt_inds_list, gt_masks_list)])


def paste_masks_3d(mask_prob, det_bboxes, vol_shape_dhw, mask_thr_binary):
    """mask_prob [P, md, mh, mw] (sigmoid of the det's class channel), det_bboxes [P, >=6] -> uint8 [P, D, H, W]"""
    P = mask_prob.size(0)
    D, H, W = (int(v) for v in vol_shape_dhw)
    out = torch.zeros((P, D, H, W), dtype=torch.uint8, device=mask_prob.device)
    if P == 0:
        return out
    prob = mask_prob.float().contiguous()
    boxes = det_bboxes.float().contiguous()
    with torch.cuda.device(prob.device):
        st = _hip.lib().mr3d_mask_paste3d(_hip.ptr(prob), prob.size(1), prob.size(2), prob.size(3), _hip.ptr(boxes),
                                          boxes.size(1), P, ctypes.c_float(mask_thr_binary), D, H, W, _hip.ptr(out),
                                          _hip.stream_ptr(prob.device))
    _hip.check(st, 'mr3d_mask_paste3d')
    return out

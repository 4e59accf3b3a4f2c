"""3D box codec + roi format (mmdet/core/bbox/transforms.py:34-65,105-160,220-239)."""
import torch

from ... import _hip


def delta2bbox3D(rois, deltas, means=[0, 0, 0, 0, 0, 0], stds=[1, 1, 1, 1, 1, 1], max_shape=None,
                 wh_ratio_clip=16 / 1000, d_ratio_clip=64 / 1000):
    """rois [n,6], deltas [n,6k] -> boxes [n,6k].  `max_shape` is the reference's (H, W, 3, D) img_shape tuple.
    `d_ratio_clip` is accepted and ignored, exactly like the reference (transforms.py:123-124)."""
    return _hip.delta2bbox3d(rois, deltas, means, stds, max_shape, wh_ratio_clip)


def bbox2delta3d(proposals, gt, means=[0, 0, 0, 0, 0, 0], stds=[1, 1, 1, 1, 1, 1]):
    assert proposals.size() == gt.size()
    return _hip.bbox2delta3d(proposals.float(), gt.float(), means, stds)


def bbox2roi3D(bbox_list):
    """list of per-image [k_i, >=6] boxes -> [sum k_i, 7] rows [batch_ind, x1, y1, x2, y2, z1, z2]
    (pure indexing: stays in PyTorch, transforms.py:220-239)"""
    rois_list = []
    for img_id, bboxes in enumerate(bbox_list):
        if bboxes.size(0) > 0:
            img_inds = bboxes.new_full((bboxes.size(0), 1), img_id)
            rois = torch.cat([img_inds, bboxes[:, :6]], dim=-1)
        else:
            rois = bboxes.new_zeros((0, 7))
        rois_list.append(rois)
    return torch.cat(rois_list, 0)

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
    """per-image boxes [k_i, >=6] -> rois [sum k_i, 7] = (image index, x1, y1, x2, y2, z1, z2) (transforms.py:220-239);
    one output allocation filled block by block"""
    counts = [int(b.size(0)) for b in bbox_list]
    rois = bbox_list[0].new_empty((sum(counts), 7))
    row = 0
    for img_id, (boxes, k) in enumerate(zip(bbox_list, counts)):
        block = rois[row:row + k]
        block[:, 0] = img_id
        block[:, 1:] = boxes[:, :6]
        row += k
    return rois

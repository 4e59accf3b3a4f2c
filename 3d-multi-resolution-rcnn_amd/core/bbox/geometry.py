"""3D IoU matrix (mmdet/core/bbox/geometry.py:4-76, the 6-column branch :49-60)."""
from ... import _hip


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False):
    assert mode in ['iou', 'iof']
    if bboxes1.shape[1] != 6 or mode != 'iou' or is_aligned:
        # the reference only ever reaches the 6-column IoU branch on the 3D path (its aligned branch is a breakpoint())
        raise NotImplementedError('only the 3D (6-column) IoU matrix is on the hot path')
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if rows * cols == 0:
        return bboxes1.new(rows, cols)
    return _hip.bbox_overlaps3d(bboxes1.float(), bboxes2.float())

"""Post-hoc evaluation of 3D detections (host side, numpy)."""
from .coco3d import COCOeval3D, det2json3D, apply_nms_3d, nms_3d_records, xyxyzz2xywhzd, iou3d_xywhzd

__all__ = ['COCOeval3D', 'det2json3D', 'apply_nms_3d', 'nms_3d_records', 'xyxyzz2xywhzd', 'iou3d_xywhzd']

from .bbox_nms import multiclass_nms_3d

__all__ = ['multiclass_nms_3d']

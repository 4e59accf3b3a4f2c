"""Host-side mirror of the pieces of `mmdet.core` that sit on the proposal path
(mmdet/core/{bbox,anchor,post_processing}); every function keeps the reference's name and signature and
runs its arithmetic in one HIP kernel instead of a chain of elementwise ATen ops."""
from .bbox import delta2bbox3D, bbox2delta3d, bbox2roi3D, bbox_overlaps
from .anchor import AnchorGenerator3D
from .post_processing import multiclass_nms_3d
from .mask import mask_target, mask_target_single, paste_masks_3d

__all__ = ['delta2bbox3D', 'bbox2delta3d', 'bbox2roi3D', 'bbox_overlaps', 'AnchorGenerator3D', 'multiclass_nms_3d',
           'mask_target', 'mask_target_single', 'paste_masks_3d']

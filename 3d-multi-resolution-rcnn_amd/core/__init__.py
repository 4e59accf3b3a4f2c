"""Host-side mirror of the pieces of `mmdet.core` that sit on the proposal path
(mmdet/core/{bbox,anchor,post_processing}); every function keeps the reference's name and signature and
runs its arithmetic in one HIP kernel instead of a chain of elementwise ATen ops."""
from .bbox import (delta2bbox3D, bbox2delta3d, bbox2roi3D, bbox_overlaps, MaxIoUAssigner, AssignResult, RandomSampler,
                   SamplingResult, bbox_target_3d, build_assigner, build_sampler, assign_and_sample)
from .anchor import AnchorGenerator3D, anchor_target, anchor_inside_flags
from .loss import (weighted_binary_cross_entropy, weighted_smoothl1, weighted_cross_entropy, smooth_l1_loss, accuracy,
                   mask_cross_entropy)
from .post_processing import multiclass_nms_3d
from .mask import mask_target, mask_target_single, paste_masks_3d

__all__ = ['delta2bbox3D', 'bbox2delta3d', 'bbox2roi3D', 'bbox_overlaps', 'AnchorGenerator3D', 'multiclass_nms_3d',
           'mask_target', 'mask_target_single', 'paste_masks_3d', 'MaxIoUAssigner', 'AssignResult', 'RandomSampler',
           'SamplingResult', 'bbox_target_3d', 'build_assigner', 'build_sampler', 'assign_and_sample', 'anchor_target',
           'anchor_inside_flags', 'weighted_binary_cross_entropy', 'weighted_smoothl1', 'weighted_cross_entropy',
           'smooth_l1_loss', 'accuracy', 'mask_cross_entropy']

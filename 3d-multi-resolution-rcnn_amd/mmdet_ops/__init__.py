"""Mirror of the reference's `mmdet.ops` plugin surface (mmdet/ops/__init__.py:1-16).

In scope (HIP kernels behind the C ABI): nms (7-column dets -> NMS3D), RoIAlign3D /
RoIAlignFunction3D / roi_align_3d.  The 2D operators of upstream mmdetection that the shipped
3D config never instantiates are exported as names that raise NotImplementedError when used,
so `from mmdet.ops import ...` statements of the reference keep resolving.
"""
from .nms import nms, soft_nms
from .roi_align import RoIAlign3D, RoIAlignFunction3D, roi_align_3d, RoIAlign, roi_align
from .dcn import DeformConv3d, ModulatedDeformConv3d, DeformConvFunction3D, deform_conv3d
from ._unsupported import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack, DeformRoIPooling,
                           DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack, deform_conv, modulated_deform_conv,
                           deform_roi_pooling, RoIPool, roi_pool, SigmoidFocalLoss, sigmoid_focal_loss)

__all__ = [
    'nms', 'soft_nms', 'RoIAlign', 'roi_align', 'RoIPool', 'roi_pool', 'DeformConv', 'DeformConvPack',
    'DeformRoIPooling', 'DeformRoIPoolingPack', 'ModulatedDeformRoIPoolingPack', 'ModulatedDeformConv',
    'ModulatedDeformConvPack', 'deform_conv', 'modulated_deform_conv', 'deform_roi_pooling', 'SigmoidFocalLoss',
    'sigmoid_focal_loss', 'RoIAlign3D', 'RoIAlignFunction3D', 'roi_align_3d', 'DeformConv3d', 'ModulatedDeformConv3d',
    'DeformConvFunction3D', 'deform_conv3d'
]

from .roi_align_3d import RoIAlign3D, RoIAlignFunction3D, roi_align_3d
from .._unsupported import _module, _raiser

# 2D variants (mmdet/ops/roi_align/functions/roi_align.py): unused by the shipped 3D config
RoIAlign = _module('RoIAlign')
roi_align = _raiser('roi_align')

__all__ = ['RoIAlign3D', 'RoIAlignFunction3D', 'roi_align_3d', 'RoIAlign', 'roi_align']

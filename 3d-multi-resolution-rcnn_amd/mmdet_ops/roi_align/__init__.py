import sys

# `roi_align_cuda`: the compiled PyTorch-ROCm extension module (PYBIND11_MODULE in csrc/torch_ext/roi_align_cuda.cpp, at::Tensor
# arguments, built in-tree by `make -C csrc ext`) -- the module name and entry points the reference's RoIAlignFunction3D
# imports (mmdet/ops/roi_align/functions/roi_align_3d.py:4).  Not built: the same entry points over ctypes.  Either way the
# work is done by libmr3d_hip.so; there is no CPU path.
try:
    from . import roi_align_cuda
except ImportError:
    from . import roi_align_ctypes as roi_align_cuda
    sys.modules[__name__ + '.roi_align_cuda'] = roi_align_cuda

from .roi_align_3d import RoIAlign3D, RoIAlignFunction3D, roi_align_3d
from .._unsupported import _module, _raiser

# 2D variants (mmdet/ops/roi_align/functions/roi_align.py): unused by the shipped 3D config
RoIAlign = _module('RoIAlign')
roi_align = _raiser('roi_align')

__all__ = ['RoIAlign3D', 'RoIAlignFunction3D', 'roi_align_3d', 'RoIAlign', 'roi_align', 'roi_align_cuda']

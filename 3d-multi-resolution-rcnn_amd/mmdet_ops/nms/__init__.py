import sys

# `nms_cuda`: the compiled PyTorch-ROCm extension module (csrc/torch_ext/nms_cuda.cpp; the reference binds it at
# mmdet/ops/nms/src/nms_cuda.cpp:23-26 and imports it in nms_wrapper.py:4).  Not built: the same entry points over ctypes.
try:
    from . import nms_cuda
except ImportError:
    from . import nms_ctypes as nms_cuda
    sys.modules[__name__ + '.nms_cuda'] = nms_cuda

from .nms_wrapper import nms, soft_nms

__all__ = ['nms', 'soft_nms', 'nms_cuda']

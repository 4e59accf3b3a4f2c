"""ctypes form of the `nms_cuda` extension module (mmdet/ops/nms/src/nms_cuda.cpp:23-26: `nms`, `nms_3d`): same entry points
as the compiled PYBIND11 module csrc/torch_ext/nms_cuda.cpp, used when that module has not been built."""
from ... import _hip


def nms_3d(dets, threshold):
    """[n,7] fp32 device tensor -> kept original indices, ascending, int64, on dets' device
    (mmdet/ops/nms/src/nms_cuda.cpp:16-21; nms_kernel.cu:196-257)."""
    return _hip.nms3d(dets, float(threshold))


def nms(dets, threshold):
    raise NotImplementedError('2D nms_cuda.nms is outside the 3D hot path (5-column dets)')

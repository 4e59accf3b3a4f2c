"""ctypes form of the `roi_align_cuda` extension module (mmdet/ops/roi_align/src/roi_align_cuda.cpp:152-157: forward,
forward3d, backward, backward3d): same entry points as the compiled PYBIND11 module csrc/torch_ext/roi_align_cuda.cpp, used
when that module has not been built.  Same positional signatures; outputs are written in place; returns 1 like the reference."""
import torch

from ... import _hip


_DTYPES = (torch.float32, torch.float64, torch.float16)      # AT_DISPATCH_FLOATING_TYPES_AND_HALF (roi_align_kernel.cu:323,674)


def _check(t, name, like=None):
    # CHECK_INPUT = CHECK_CUDA + CHECK_CONTIGUOUS (roi_align_cuda.cpp:34-39) -> RuntimeError
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDAtensor ')
    if t.dtype not in _DTYPES:
        raise RuntimeError(f'{name} must be float32, float64 or float16')
    if like is not None and t.dtype != like.dtype:
        # the reference reads all three through data<scalar_t>() of the features' type
        raise RuntimeError(f'{name} must have the dtype of the features ({like.dtype})')


def forward3d(features, rois, pooled_depth, pooled_height, pooled_width, spatial_scale, spatial_scale_depth,
              sample_num, output):
    _check(features, 'features'); _check(rois, 'rois', features); _check(output, 'output', features)
    if not (rois.is_contiguous() and output.is_contiguous()):
        raise RuntimeError('rois/output must be contiguous ')
    if rois.dim() != 2 or rois.size(1) != 7:
        # the reference prints "wrong roi size" and returns 0, which its Python caller ignores
        # (roi_align_cuda.cpp:80-83); raising is strictly safer
        raise RuntimeError('wrong roi size: rois must be [num_rois, 7]')
    _hip.roi_align3d_forward(features, rois, int(pooled_depth), int(pooled_height), int(pooled_width),
                             float(spatial_scale), float(spatial_scale_depth), int(sample_num), output)
    return 1


def backward3d(top_grad, rois, pooled_depth, pooled_height, pooled_width, spatial_scale, spatial_scale_depth,
               sample_num, bottom_grad, bug_compat=True):
    _check(top_grad, 'top_grad'); _check(rois, 'rois', top_grad); _check(bottom_grad, 'bottom_grad', top_grad)
    if not (rois.is_contiguous() and top_grad.is_contiguous()):
        raise RuntimeError('rois/top_grad must be contiguous ')
    if rois.dim() != 2 or rois.size(1) != 7:
        raise RuntimeError('wrong roi size: rois must be [num_rois, 7]')
    _hip.roi_align3d_backward(top_grad, rois, int(pooled_depth), int(pooled_height), int(pooled_width),
                              float(spatial_scale), float(spatial_scale_depth), int(sample_num), bottom_grad,
                              bug_compat=bug_compat)
    return 1


def forward(*args, **kwargs):
    raise NotImplementedError('2D roi_align_cuda.forward is outside the 3D hot path')


def backward(*args, **kwargs):
    raise NotImplementedError('2D roi_align_cuda.backward is outside the 3D hot path')

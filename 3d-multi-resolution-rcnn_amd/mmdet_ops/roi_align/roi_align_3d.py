"""RoIAlign3D as a torch.autograd.Function + nn.Module with the reference's call signatures
(mmdet/ops/roi_align/functions/roi_align_3d.py:6-92, modules/roi_align_3d.py:5-18; looked up by
name through `getattr(ops, 'RoIAlign3D')` in models/roi_extractors/single_level.py:45-56).

Extensions over the reference, all opt-in or layout-driven:
  * features may be channels_last_3d (NDHWC strides): the coalesced gfx950 kernel is used and the
    input gradient comes back in the same memory format;
  * `RoIAlign3D(..., bug_compat=False)` selects the mathematically correct backward; the default
    reproduces the reference's top_grad indexing (roi_align_kernel.cu:554-555) for gradient parity.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import roi_align_cuda


class RoIAlignFunction3D(Function):

    @staticmethod
    def forward(ctx, features, rois, out_size, out_size_depth, spatial_scale, spatial_scale_depth, sample_num=0,
                bug_compat=True):
        if isinstance(out_size, int):
            out_h = out_w = out_size
            out_d = out_size_depth
        elif isinstance(out_size, tuple):
            # the reference stops in breakpoint() here (functions/roi_align_3d.py:14-15); (h, w) is accepted instead
            if len(out_size) != 2 or not all(isinstance(v, int) for v in out_size):
                raise TypeError('"out_size" must be an integer or tuple of integers')
            out_h, out_w = out_size
            out_d = out_size_depth
        else:
            raise TypeError('"out_size" must be an integer or tuple of integers')
        if not features.is_cuda:
            raise NotImplementedError      # functions/roi_align_3d.py:36-37
        ctx.spatial_scale = spatial_scale
        ctx.spatial_scale_depth = spatial_scale_depth
        ctx.sample_num = sample_num
        ctx.bug_compat = bug_compat
        ctx.feature_size = features.size()
        ctx.channels_last = (not features.is_contiguous()) and features.is_contiguous(
            memory_format=torch.channels_last_3d)
        ctx.save_for_backward(rois)
        output = features.new_zeros(rois.size(0), features.size(1), out_d, out_h, out_w)
        roi_align_cuda.forward3d(features, rois, out_d, out_h, out_w, spatial_scale, spatial_scale_depth, sample_num,
                                 output)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois = ctx.saved_tensors[0]
        assert ctx.feature_size is not None and grad_output.is_cuda
        grad_input = None
        if ctx.needs_input_grad[0]:
            fmt = torch.channels_last_3d if ctx.channels_last else torch.contiguous_format
            grad_input = torch.empty(ctx.feature_size, dtype=rois.dtype, device=rois.device, memory_format=fmt).zero_()
            out_d, out_h, out_w = grad_output.shape[2:]
            roi_align_cuda.backward3d(grad_output.contiguous(), rois, out_d, out_h, out_w, ctx.spatial_scale,
                                      ctx.spatial_scale_depth, ctx.sample_num, grad_input, bug_compat=ctx.bug_compat)
        return grad_input, None, None, None, None, None, None, None


roi_align_3d = RoIAlignFunction3D.apply


class RoIAlign3D(nn.Module):

    def __init__(self, out_size, out_size_depth, spatial_scale, spatial_scale_depth, sample_num=0, bug_compat=True):
        super().__init__()
        self.out_size = out_size
        self.out_size_depth = out_size_depth
        self.spatial_scale = float(spatial_scale)
        self.spatial_scale_depth = float(spatial_scale_depth)
        self.sample_num = int(sample_num)
        self.bug_compat = bool(bug_compat)

    def forward(self, features, rois):
        return RoIAlignFunction3D.apply(features, rois, self.out_size, self.out_size_depth, self.spatial_scale,
                                        self.spatial_scale_depth, self.sample_num, self.bug_compat)

// `roi_align_cuda` -- the compiled extension module the reference's RoIAlignFunction3D imports
// (mmdet/ops/roi_align/functions/roi_align_3d.py:4, bound at mmdet/ops/roi_align/src/roi_align_cuda.cpp:152-157):
// forward3d / backward3d with the reference's positional signatures, forwarding to mr3d_roi_align3d_{forward,backward}[_dtype].
#include "common.h"

namespace {

int dtype_code(const at::Tensor &t) {
  // AT_DISPATCH_FLOATING_TYPES_AND_HALF (roi_align_kernel.cu:323,674)
  switch (t.scalar_type()) {
    case at::kFloat: return MR3D_DTYPE_F32;
    case at::kHalf: return MR3D_DTYPE_F16;
    case at::kDouble: return MR3D_DTYPE_F64;
    default: TORCH_CHECK(false, "roi_align 3d: float32, float64 or float16 tensors only, got ", t.scalar_type());
  }
  return -1;
}

// the feature map / its gradient: NCDHW contiguous as in the reference, or channels_last_3d (what the conv engine hands out)
int layout_of(const at::Tensor &t, const char *name) {
  TORCH_CHECK(t.dim() == 5, name, " must be [N, C, D, H, W]");
  if (t.is_contiguous()) return MR3D_LAYOUT_NCDHW;
  if (t.is_contiguous(at::MemoryFormat::ChannelsLast3d)) return MR3D_LAYOUT_NDHWC;
  TORCH_CHECK(false, name, " must be contiguous ");
  return -1;
}

int roi_align_forward_3d(at::Tensor features, at::Tensor rois, int pooled_depth, int pooled_height, int pooled_width,
                         float spatial_scale, float spatial_scale_depth, int sample_num, at::Tensor output) {
  MR3D_CHECK_CUDA(features); MR3D_CHECK_CUDA(rois); MR3D_CHECK_CUDA(output);
  MR3D_CHECK_CONTIGUOUS(rois); MR3D_CHECK_CONTIGUOUS(output);
  const int layout = layout_of(features, "features");
  const int dt = dtype_code(features);
  TORCH_CHECK(rois.scalar_type() == features.scalar_type() && output.scalar_type() == features.scalar_type(),
              "rois / output must have the dtype of the features");      // the reference reads all three as scalar_t
  // the reference prints "wrong roi size" and returns 0, which its Python caller ignores (:80-83); raising is safer
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 7, "wrong roi size: rois must be [num_rois, 7]");
  const int R = rois.size(0), N = features.size(0), C = features.size(1), D = features.size(2), H = features.size(3),
            W = features.size(4);
  TORCH_CHECK(output.dim() == 5 && output.size(0) == R && output.size(1) == C && output.size(2) == pooled_depth &&
              output.size(3) == pooled_height && output.size(4) == pooled_width, "output must be [num_rois, C, pd, ph, pw]");
  Mr3dDeviceGuard guard(features.device());
  int st;
  if (dt == MR3D_DTYPE_F32)
    st = mr3d_roi_align3d_forward(features.data_ptr<float>(), rois.data_ptr<float>(), R, N, C, D, H, W, pooled_depth, pooled_height,
                                  pooled_width, spatial_scale, spatial_scale_depth, sample_num, layout, output.data_ptr<float>(),
                                  mr3d_current_stream(features));
  else
    st = mr3d_roi_align3d_forward_dtype(features.data_ptr(), rois.data_ptr(), R, N, C, D, H, W, pooled_depth, pooled_height,
                                        pooled_width, spatial_scale, spatial_scale_depth, sample_num, layout, dt, output.data_ptr(),
                                        mr3d_current_stream(features));
  mr3d_check_status(st, "mr3d_roi_align3d_forward");
  return 1;
}

// bug_compat: reproduce the reference's bottom-gradient index arithmetic (roi_align_kernel.cu:554-555) -- the default,
// as everywhere in this package; false = the mathematically consistent gradient
int roi_align_backward_3d(at::Tensor top_grad, at::Tensor rois, int pooled_depth, int pooled_height, int pooled_width,
                          float spatial_scale, float spatial_scale_depth, int sample_num, at::Tensor bottom_grad,
                          bool bug_compat) {
  MR3D_CHECK_CUDA(top_grad); MR3D_CHECK_CUDA(rois); MR3D_CHECK_CUDA(bottom_grad);
  MR3D_CHECK_CONTIGUOUS(top_grad); MR3D_CHECK_CONTIGUOUS(rois);
  const int layout = layout_of(bottom_grad, "bottom_grad");
  const int dt = dtype_code(top_grad);
  TORCH_CHECK(rois.scalar_type() == top_grad.scalar_type() && bottom_grad.scalar_type() == top_grad.scalar_type(),
              "rois / bottom_grad must have the dtype of top_grad");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 7, "wrong roi size: rois must be [num_rois, 7]");
  const int R = rois.size(0), N = bottom_grad.size(0), C = bottom_grad.size(1), D = bottom_grad.size(2), H = bottom_grad.size(3),
            W = bottom_grad.size(4);
  TORCH_CHECK(top_grad.dim() == 5 && top_grad.size(0) == R && top_grad.size(1) == C && top_grad.size(2) == pooled_depth &&
              top_grad.size(3) == pooled_height && top_grad.size(4) == pooled_width, "top_grad must be [num_rois, C, pd, ph, pw]");
  Mr3dDeviceGuard guard(bottom_grad.device());
  const int flags = bug_compat ? MR3D_ROIALIGN_BUG_COMPAT : 0;
  int st;
  if (dt == MR3D_DTYPE_F32)
    st = mr3d_roi_align3d_backward(top_grad.data_ptr<float>(), rois.data_ptr<float>(), R, N, C, D, H, W, pooled_depth, pooled_height,
                                   pooled_width, spatial_scale, spatial_scale_depth, sample_num, layout, flags,
                                   bottom_grad.data_ptr<float>(), mr3d_current_stream(bottom_grad));
  else
    st = mr3d_roi_align3d_backward_dtype(top_grad.data_ptr(), rois.data_ptr(), R, N, C, D, H, W, pooled_depth, pooled_height,
                                         pooled_width, spatial_scale, spatial_scale_depth, sample_num, layout, flags, dt,
                                         bottom_grad.data_ptr(), mr3d_current_stream(bottom_grad));
  mr3d_check_status(st, "mr3d_roi_align3d_backward");
  return 1;
}

int unsupported_2d(py::args, py::kwargs) {
  PyErr_SetString(PyExc_NotImplementedError, "2D roi_align_cuda.forward / backward are outside the 3D hot path");
  throw py::error_already_set();
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("forward3d", &roi_align_forward_3d, "Roi_Align forward 3D (MI355X)");
  m.def("backward3d", &roi_align_backward_3d, "Roi_Align backward 3D (MI355X)", py::arg("top_grad"), py::arg("rois"),
        py::arg("pooled_depth"), py::arg("pooled_height"), py::arg("pooled_width"), py::arg("spatial_scale"),
        py::arg("spatial_scale_depth"), py::arg("sample_num"), py::arg("bottom_grad"), py::arg("bug_compat") = true);
  m.def("forward", &unsupported_2d);
  m.def("backward", &unsupported_2d);
}

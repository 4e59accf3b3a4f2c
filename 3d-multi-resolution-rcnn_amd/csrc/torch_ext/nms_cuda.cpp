// `nms_cuda` -- the compiled extension module the reference's nms_wrapper imports (mmdet/ops/nms/nms_wrapper.py:4, bound at
// mmdet/ops/nms/src/nms_cuda.cpp:23-26): nms_3d(dets [n,7] float32 device tensor, threshold) -> kept ORIGINAL indices,
// ascending, int64 (nms_kernel.cu:196-257), forwarding to mr3d_nms3d.
#include "common.h"

namespace {

at::Tensor nms_3d(const at::Tensor &dets, const float threshold) {
  MR3D_CHECK_CUDA(dets);                                                       // nms_cuda.cpp:4,17
  if (dets.numel() == 0) return at::empty({0}, dets.options().dtype(at::kLong).device(at::kCPU));      // :18-19
  TORCH_CHECK(dets.dim() == 2 && dets.size(1) == 7, "nms_3d expects dets of shape [n, 7]");
  TORCH_CHECK(dets.scalar_type() == at::kFloat, "nms_3d expects float32 dets");     // nms_kernel.cu:197,207 (data<float>())
  const at::Tensor d = dets.stride(1) == 1 ? dets : dets.contiguous();
  const int n = d.size(0);
  Mr3dDeviceGuard guard(d.device());
  const size_t ws_bytes = mr3d_nms3d_workspace_bytes(n);
  at::Tensor ws = at::empty({(int64_t)ws_bytes + 256}, d.options().dtype(at::kByte));
  char *base = (char *)ws.data_ptr();
  base += (256 - ((uintptr_t)base & 255)) & 255;
  at::Tensor keep = at::empty({n}, d.options().dtype(at::kLong));
  at::Tensor num = at::empty({1}, d.options().dtype(at::kInt));
  const int st = mr3d_nms3d(d.data_ptr<float>(), n, (int)d.stride(0), threshold, keep.data_ptr<int64_t>(), num.data_ptr<int32_t>(), base,
                            ws_bytes, mr3d_current_stream(d));
  mr3d_check_status(st, "mr3d_nms3d");
  const int k = num.item<int32_t>();           // the only host sync: 4 bytes (the reference copies the whole bit-matrix, :229-233)
  return keep.narrow(0, 0, k);
}

at::Tensor nms_2d(const at::Tensor &, const float) {
  PyErr_SetString(PyExc_NotImplementedError, "2D nms_cuda.nms is outside the 3D hot path (5-column dets)");
  throw py::error_already_set();
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("nms_3d", &nms_3d, "non-maximum suppression 3D (MI355X)");
  m.def("nms", &nms_2d, "non-maximum suppression (2D: not on the 3D hot path)");
}

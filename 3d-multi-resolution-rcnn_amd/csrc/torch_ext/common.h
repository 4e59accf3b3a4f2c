// Shared by the two PyTorch-ROCm extension modules (roi_align_cuda, nms_cuda): at::Tensor in, C ABI (include/mr3d.h) out.
// Host-only C++ (g++): the device code lives in libmr3d_hip.so, which these modules link against.
#pragma once
#include <torch/extension.h>
// PyTorch-ROCm presents its devices with the device type "cuda": the stream / guard types that carry that type
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>

#include "../../../include/mr3d.h"

// CHECK_CUDA / CHECK_CONTIGUOUS of the reference (mmdet/ops/roi_align/src/roi_align_cuda.cpp:34-39), same messages
#define MR3D_CHECK_CUDA(x) TORCH_CHECK((x).is_cuda(), #x, " must be a CUDAtensor ")
#define MR3D_CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x, " must be contiguous ")

inline mr3d_stream_t mr3d_current_stream(const at::Tensor &t) {
  return (mr3d_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

inline void mr3d_check_status(int st, const char *what) {
  TORCH_CHECK(st == MR3D_OK, what, ": ", mr3d_status_string(st), " (status ", st, ")");
}

using Mr3dDeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;

// Shared helpers for the gfx950 kernels behind include/mr3d.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/mr3d.h"

#define MR3D_WAVE 64

extern int g_mr3d_tuning[MR3D_TUNE_NUM_KEYS];          // mr3d_tuning_set (mr3d_api.hip)

static inline hipStream_t mr3d_stream(mr3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipGetLastError() also clears the sticky error, so a failed call does not poison the next one
static inline int mr3d_launch_status() {
  return hipGetLastError() == hipSuccess ? MR3D_OK : MR3D_ERR_LAUNCH;
}

static inline size_t mr3d_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline int mr3d_next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

__device__ __forceinline__ uint64_t mr3d_readlane64(uint64_t v, int lane) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t mr3d_uniform64(uint64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// 16 bytes through a raw buffer descriptor: address = descriptor base + vector byte offset + SCALAR byte offset.  What differs
// between the loads of a loop is folded into the scalar operand, so a load costs no vector ALU instruction (flat loads rebuild a
// 64-bit pointer per load, and vector ALU work does not hide under fp32 MFMAs on this chip).
typedef float buf_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const float *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 buffer_load4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  const buf_f32x4 v = __builtin_bit_cast(buf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, (int)soff_bytes, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
typedef float buf_f32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ buf_f32x3 buffer_load3(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(buf_f32x3, __builtin_amdgcn_raw_buffer_load_b96(r, (int)voff_bytes, (int)soff_bytes, 0));
}
__device__ __forceinline__ float buffer_load1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff_bytes, (int)soff_bytes, 0));
}

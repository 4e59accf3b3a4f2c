// Shared helpers for the gfx950 kernels behind include/mr3d.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/mr3d.h"

#define MR3D_WAVE 64

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

// One-workgroup bitonic sort of 64-bit keys held in registers (shared by nms3d.hip and proposals.hip).
#pragma once
#include "mr3d_common.h"

// A thread keeps E consecutive keys in registers.  A compare-exchange step with partner distance j is then
//   j < E        inside the thread,
//   j < 64 * E   inside the wave: the partner key comes through two ds_bpermute (no LDS memory, no barrier),
//   otherwise    across waves: one round trip through LDS with two barriers.
// For 2048 keys (E = 2) only 10 of the 66 steps of the network touch LDS; the all-LDS form spent 66 barriers (32 us).
namespace {

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int lane_mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, lane_mask);
  const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), lane_mask);
  return ((uint64_t)hi << 32) | lo;
}

template <int E>
__device__ __forceinline__ void bitonic_sort_regs(uint64_t (&v)[E], int tid, int np2, uint64_t *keys) {
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64 * E) {
#pragma unroll
        for (int e = 0; e < E; ++e) keys[tid * E + e] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = tid * E + e;
          const uint64_t o = keys[i ^ j];
          const bool lower = (i & j) == 0, up = (i & k) == 0;
          v[e] = (lower == up) ? (v[e] < o ? v[e] : o) : (v[e] > o ? v[e] : o);
        }
        __syncthreads();
      } else if (j >= E) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = tid * E + e;
          const uint64_t o = shfl_xor64(v[e], j / E);
          const bool lower = (i & j) == 0, up = (i & k) == 0;
          v[e] = (lower == up) ? (v[e] < o ? v[e] : o) : (v[e] > o ? v[e] : o);
        }
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
          if ((e & j) == 0) {
            const int i = tid * E + e;
            const uint64_t x = v[e], y = v[e ^ j];
            const bool up = (i & k) == 0;
            const bool sw = (x > y) == up;
            v[e] = sw ? y : x;
            v[e ^ j] = sw ? x : y;
          }
      }
    }
  }
}


}  // namespace

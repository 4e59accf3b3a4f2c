// Sustained rate of the fp32 matrix instructions with every CU busy, and the shader clock under that load:
// s_memtime (shader cycles) against wall_clock64 (constant 100 MHz) around a register-only MFMA loop.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_clock.hip -o gpurun_out/mfma_clock && gpurun_out/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, long long *ticks, int iters) {
  f32x4 a4[NACC];
  f32x16 a16[NACC];
  for (int i = 0; i < NACC; ++i) { a4[i] = f32x4{0, 0, 0, 0}; for (int r = 0; r < 16; ++r) a16[i][r] = 0.f; }
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  const long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) a4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a4[i], 0, 0, 0);
      else a16[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a16[i], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += KIND == 0 ? a4[i][0] : a16[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}

// the shape of stem_wgrad_kernel's inner loop without its LDS reads: 3 accumulators, the A / B operands of every k-step formed by
// a few vector and scalar instructions, a barrier every `per_barrier` k-steps (0 = none)
__global__ __launch_bounds__(512) void stem_like_loop(float *out, long long *ticks, int iters, int per_barrier) {
  f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  const int lk = threadIdx.x >> 4 & 3;
  const long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    for (int ks = 0; ks < 128; ks += 2) {
      const float a0 = __int_as_float(ks + lk), a1 = __int_as_float(ks + 1 + lk);
      float b0[3], b1[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) { b0[q] = __int_as_float(ks * 3 + q + it); b1[q] = __int_as_float(ks * 3 + 3 + q + it); }
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[q], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[q], acc[q], 0, 0, 0);
    }
    if (per_barrier && (it % per_barrier) == per_barrier - 1) __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][0] + acc[2][0];
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}

void run_stem_like(int cus, int per_barrier) {
  const int grid = 2 * cus, iters = 60;                // 2 workgroups of 8 waves per CU = 4 waves per SIMD, 384 MFMAs per iteration
  float *out; long long *ticks;
  hipMalloc(&out, grid * 512 * 4); hipMalloc(&ticks, grid * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  stem_like_loop<<<grid, 512>>>(out, ticks, 2, per_barrier);
  hipEventRecord(e0);
  stem_like_loop<<<grid, 512>>>(out, ticks, iters, per_barrier);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas_per_simd = 384.0 * iters * 4;
  printf("stem-like loop, 4 waves/SIMD, barrier every %d x 384 MFMAs: %.3f ms, ns per MFMA and SIMD %.2f\n", per_barrier, ms,
         ms * 1e6 / mfmas_per_simd);
  hipFree(out); hipFree(ticks);
}

template <int KIND, int NACC>
void run(const char *name, int wgs_per_cu, int cus) {
  const int grid = cus * wgs_per_cu, iters = 20000;
  float *out; long long *ticks;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&ticks, grid * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<KIND, NACC><<<grid, 256>>>(out, ticks, 100);
  hipEventRecord(e0);
  mfma_loop<KIND, NACC><<<grid, 256>>>(out, ticks, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(2 * grid);
  hipMemcpy(h.data(), ticks, grid * 16, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  cyc /= grid; wall /= grid;
  const double mfmas_per_simd = (double)iters * NACC * wgs_per_cu;          // 4 waves of a workgroup = one per SIMD
  const double flop = (KIND == 0 ? 2048.0 : 4096.0) * iters * NACC * 4.0 * grid;
  printf("%-28s %d wg/CU: %.3f ms, %.1f TFLOP/s, s_memtime cycles per MFMA and SIMD %.1f, s_memtime rate %.0f MHz (wall_clock64 100 MHz), "
         "ns per MFMA and SIMD %.2f\n", name, wgs_per_cu, ms, flop / ms * 1e-9, cyc / mfmas_per_simd, cyc / wall * 100.0, ms * 1e6 / mfmas_per_simd);
  hipFree(out); hipFree(ticks);
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  run<0, 3>("16x16x4 f32, 3 accumulators", 1, cus);
  run<0, 3>("16x16x4 f32, 3 accumulators", 2, cus);
  run<0, 3>("16x16x4 f32, 3 accumulators", 4, cus);
  run<0, 8>("16x16x4 f32, 8 accumulators", 1, cus);
  run<1, 2>("32x32x2 f32, 2 accumulators", 1, cus);
  run<1, 2>("32x32x2 f32, 2 accumulators", 2, cus);
  run<0, 3>("16x16x4 f32, 3 acc, 32 CUs", 1, 32);
  run_stem_like(cus, 0);
  run_stem_like(cus, 1);
  return 0;
}

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
  return 0;
}

// Issue cost of fp32 vector instructions with every SIMD busy: v_fma_f32 against v_pk_fma_f32 (and v_pk_add_f32), at 1, 2, 4
// and 8 waves per SIMD, s_memtime cycles per instruction and SIMD.  Answers: does a packed instruction do two lanes' worth of
// work in the time of one (it is how the 157 TFLOP/s vector peak is quoted), or does it cost two issue slots?
//   hipcc -O3 -Wno-unused-result --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate; gpurun -- tools/micro/valu_rate
// Measured (MI355X, 4-8 waves per SIMD, every SIMD busy; WALL time per instruction and SIMD relative to v_fma_f32 = 1, about 1.9 ns):
// v_pk_fma_f32 1.15-1.2 (two multiply-adds per lane: 1.7-1.8x the work per unit of time), v_add_f32 0.65, v_pk_add_f32 1.1 (1.15x the
// work): packing pays for multiply-adds, not for the add / subtract chains of a Winograd transform (the printed cycle counts are
// s_memtime deltas of a workgroup, the printed clock is only their ratio to the launch's wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void valu_loop(float *out, long long *ticks, int iters) {
  f32x2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f32x2{threadIdx.x * 1e-3f + i, 1.f + i};
  const f32x2 m = f32x2{1.0001f, 0.9999f}, c = f32x2{1e-6f, -1e-6f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
        else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int cus, int wg_per_cu) {
  const int grid = wg_per_cu * cus, iters = 2000;
  float *out; long long *ticks;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&ticks, grid * 8);
  valu_loop<KIND><<<grid, 256>>>(out, ticks, 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  valu_loop<KIND><<<grid, 256>>>(out, ticks, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid);
  hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : h) cyc += v; cyc /= grid;
  const double insts = (double)iters * 32;                 // per wave
  // a workgroup = 4 waves = one per SIMD; wg_per_cu workgroups per CU -> wg_per_cu waves per SIMD
  printf("%-14s %d waves/SIMD: %.2f shader cycles per instruction and wave, %.2f per instruction and SIMD; %.3f ms, %.2f GHz\n", name,
         wg_per_cu, cyc / insts, cyc / insts / wg_per_cu, ms, cyc / (ms * 1e6));
  hipFree(out); hipFree(ticks);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", cus, w);
    run<1>("v_pk_fma_f32", cus, w);
    run<3>("v_add_f32", cus, w);
    run<2>("v_pk_add_f32", cus, w);
  }
  return 0;
}

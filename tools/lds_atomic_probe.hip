// micro-probe: rate of LDS fp32 atomic adds vs plain LDS read-modify-write vs integer atomics on gfx950
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_probe.hip -o gpurun_out/scratch/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE, int STRIDE>
__global__ __launch_bounds__(256) void probe(float *out, int iters) {
  __shared__ float box[16384];
  for (int e = threadIdx.x; e < 16384; e += 256) box[e] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int a = wave * 4096 + lane * STRIDE;
  float v = 1.0f + lane;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = (a + u * 64 * STRIDE) & 16383;
      if (MODE == 0) atomicAdd(box + idx, v);
      else if (MODE == 1) box[idx] += v;
      else if (MODE == 2) atomicAdd(reinterpret_cast<int *>(box) + idx, (int)v);
      else if (MODE == 3) { float r = atomicAdd(box + idx, v); v += r * 1e-30f; }
      else if (MODE == 4) {
        int *ip = reinterpret_cast<int *>(box) + idx;
        const int seen = *ip;
        const int got = atomicCAS(ip, seen, __float_as_int(__int_as_float(seen) + v));
        if (got != seen) v += 1e-30f;
      } else if (MODE == 5) {
        atomicAdd(reinterpret_cast<unsigned long long *>(box) + (idx >> 1), (unsigned long long)(long long)v);
      } else if (MODE == 7) {
        atomicAdd(reinterpret_cast<double *>(box) + (idx >> 1), (double)v);
      } else if (MODE == 8) {
        __hip_atomic_fetch_max(box + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 6) {
        int r = atomicAdd(reinterpret_cast<int *>(box) + idx, (int)v); v += r * 1e-30f;
      }
    }
    a += 1;
  }
  __syncthreads();
  float s = 0.f;
  for (int e = threadIdx.x; e < 16384; e += 256) s += box[e];
  out[blockIdx.x * 256 + threadIdx.x] = s + v;
}
template <int MODE, int STRIDE>
void run(const char *name) {
  float *out;
  hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, STRIDE><<<512, 256>>>(out, 10);
  hipEventRecord(e0);
  probe<MODE, STRIDE><<<512, 256>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = 512.0 * 4 * iters * 8;          // wave-level LDS instructions
  // 2 workgroups per CU resident (64 KB each): per-CU rate
  printf("%-28s stride %2d: %8.3f ms  %.1f cycles per wave-instruction per CU (2.4 GHz, 256 CUs)\n", name, STRIDE, ms,
         ms * 1e-3 * 2.4e9 / (instr / 256.0));
  hipFree(out);
}
int main() {
  run<0, 1>("ds_add_f32");
  run<0, 2>("ds_add_f32");
  run<0, 16>("ds_add_f32");
  run<3, 1>("ds_add_rtn_f32");
  run<1, 1>("read + add + write");
  run<2, 1>("ds_add_u32");
  run<6, 1>("ds_add_rtn_u32");
  run<4, 1>("read + ds_cmpst_rtn_b32");
  run<5, 1>("ds_add_u64");
  run<7, 1>("ds_add_f64");
  run<8, 1>("ds_max_f32");
  return 0;
}

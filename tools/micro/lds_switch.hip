// Do back-to-back dependent kernels of one stream start later when consecutive kernels differ in how much LDS they ask for
// (<= 64 KB vs the opt-in sizes above it)?  The step timeline shows 0 us between some main-queue kernels and 6-7 us between others.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/lds_switch.hip -o tools/micro/lds_switch ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *p, int n) {
  extern __shared__ float s[];
  s[threadIdx.x] = p[blockIdx.x * 256 + threadIdx.x];
  __syncthreads();
  float a = s[(threadIdx.x + 1) & 255];
  for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
  p[blockIdx.x * 256 + threadIdx.x] = a;
}
static float run(float *d, int la, int lb, int wgs, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) { k<<<wgs, 256, la>>>(d, 2000); k<<<wgs, 256, lb>>>(d, 2000); }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) { k<<<wgs, 256, la>>>(d, 2000); k<<<wgs, 256, lb>>>(d, 2000); }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / (2 * iters);
}
int main() {
  float *d; hipMalloc(&d, 4096 * 256 * 4); hipMemset(d, 0, 4096 * 256 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int sizes[] = {1024, 32 * 1024, 60 * 1024, 66 * 1024, 82 * 1024, 140 * 1024};
  for (int wgs : {256, 2048}) {
    printf("workgroups %d: us per kernel (rows: LDS of kernel A, columns: LDS of kernel B, alternating A B A B ...)\n        ", wgs);
    for (int b : sizes) printf("%7dK", b / 1024);
    printf("\n");
    for (int a : sizes) {
      printf("%6dK ", a / 1024);
      for (int b : sizes) printf("%8.2f", run(d, a, b, wgs, 200));
      printf("\n");
    }
  }
  return 0;
}

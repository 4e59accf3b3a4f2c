// Gaps between dependent kernels of one stream as the kernel trace shows them (the step timeline has 0 us between some pairs and 6-7 us
// between others): does it depend on the grid (one round of workgroups or many), the block size, the amount of LDS?
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/launch_gap.hip -o tools/micro/launch_gap
// run (GPU box): rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -o gap -- ./tools/micro/launch_gap ; python tools/timeline.py ...
#include <hip/hip_runtime.h>
#include <cstdio>
template <int TAG>
__global__ void k(float *p, int n) {
  extern __shared__ float s[];
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff;
  s[threadIdx.x] = p[i];
  __syncthreads();
  float a = s[(threadIdx.x + 1) % blockDim.x];
  for (int j = 0; j < n; ++j) a = a * 1.0001f + 0.5f;
  p[i] = a;
}
int main() {
  float *d; hipMalloc(&d, (1 << 20) * 4); hipMemset(d, 0, (1 << 20) * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 6; ++rep) {
    // k<0>: one round, small LDS; k<1>: many rounds (8192 workgroups); k<2>: one round, 82 KB LDS; k<3>: 1024-thread blocks
    k<0><<<256, 256, 4096>>>(d, 6000);
    k<0><<<256, 256, 4096>>>(d, 6000);
    k<1><<<8192, 256, 4096>>>(d, 300);
    k<1><<<8192, 256, 4096>>>(d, 300);
    k<0><<<256, 256, 4096>>>(d, 6000);
    k<2><<<256, 256, 82 * 1024>>>(d, 6000);
    k<2><<<256, 256, 82 * 1024>>>(d, 6000);
    k<0><<<256, 256, 4096>>>(d, 6000);
    k<3><<<256, 1024, 4096>>>(d, 6000);
    k<3><<<256, 1024, 4096>>>(d, 6000);
    k<1><<<8192, 256, 4096>>>(d, 300);
    k<0><<<256, 256, 4096>>>(d, 6000);
  }
  hipDeviceSynchronize();
  return 0;
}

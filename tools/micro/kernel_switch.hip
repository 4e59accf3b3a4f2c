// Back-to-back launches in one stream: the same kernel against two alternating kernels, with tiny and with padded code.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/kernel_switch.hip -o tools/micro/kernel_switch ; run on the GPU box
// Measured (round 4): 2.8 us per launch of a tiny kernel whether the stream alternates kernels, code sizes or dynamic LDS sizes or
// not -- switching kernels costs nothing by itself; the 5-7 us in front of different-kernel launches in the traced plan runs
// (profiles/r04/timeline_bf16_backbone_step_kernels.txt) are not an instruction-cache or dispatcher-state effect.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ID>
__global__ void tiny(float *p, int work) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < work; ++i) v = v * 1.0001f + (float)ID;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

// a kernel with a long straight-line body (~12 KB of code), executed once per wave
template <int ID>
__global__ void padded(float *p, int work) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < work; ++i) {
#pragma unroll
    for (int j = 0; j < 1500; ++j) v = v * 1.0001f + (float)(ID + j);
  }
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

template <class F>
float timed(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f(i);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) f(i);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  float *p;
  hipMalloc(&p, 256 * 1024 * sizeof(float));
  hipMemset(p, 0, 256 * 1024 * sizeof(float));
  const int reps = 2000;
  for (int grid : {256, 1024}) {
    for (int work : {1, 2000}) {
      float same = timed([&](int) { tiny<0><<<grid, 256>>>(p, work); }, reps);
      float alt = timed([&](int i) { if (i & 1) tiny<1><<<grid, 256>>>(p, work); else tiny<2><<<grid, 256>>>(p, work); }, reps);
      printf("tiny   grid %4d work %4d: same kernel %.2f us/launch, alternating %.2f\n", grid, work, same, alt);
    }
    float same = timed([&](int) { padded<0><<<grid, 256>>>(p, 1); }, reps);
    float alt = timed([&](int i) { if (i & 1) padded<1><<<grid, 256>>>(p, 1); else padded<2><<<grid, 256>>>(p, 1); }, reps);
    float alt3 = timed([&](int i) { if (i % 3 == 0) padded<3><<<grid, 256>>>(p, 1); else if (i % 3 == 1) padded<4><<<grid, 256>>>(p, 1); else tiny<5><<<grid, 256>>>(p, 1); }, reps);
    printf("padded grid %4d: same kernel %.2f us/launch, alternating two %.2f, three-way %.2f\n", grid, same, alt, alt3);
  }
  // the same kernel with alternating dynamic LDS sizes / a data dependency through memory
  hipFuncSetAttribute(reinterpret_cast<const void *>(tiny<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void *>(tiny<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  {
    float a = timed([&](int) { tiny<7><<<256, 256, 8 * 1024>>>(p, 1); }, reps);
    float b = timed([&](int i) { tiny<7><<<256, 256, (i & 1) ? 8 * 1024 : 100 * 1024>>>(p, 1); }, reps);
    float c = timed([&](int i) { if (i & 1) tiny<7><<<256, 256, 8 * 1024>>>(p, 1); else tiny<8><<<256, 256, 100 * 1024>>>(p, 1); }, reps);
    float d = timed([&](int) { tiny<7><<<256, 256, 100 * 1024>>>(p, 1); }, reps);
    printf("dynamic LDS: 8 KB every launch %.2f us, 100 KB every launch %.2f, alternating 8 / 100 KB %.2f, alternating kernels and sizes %.2f\n", a, d, b, c);
  }
  return 0;
}

// What does a fork point cost the queue it is recorded in?  A chain of dependent kernels on one stream with, between every two of them,
// (a) nothing, (b) hipEventRecord of a timing-less event (+ a second stream waiting on it), (c) the same with
// hipEventDisableSystemFence, (d) hipStreamWriteValue32 (+ hipStreamWaitValue32 on the second stream).
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/fork_cost.hip -o tools/micro/fork_cost ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float a = p[i];
  for (int j = 0; j < n; ++j) a = a * 1.0001f + 0.5f;
  p[i] = a;
}
int main() {
  float *d, *d2; hipMalloc(&d, 65536 * 4); hipMalloc(&d2, 65536 * 4);
  hipMemset(d, 0, 65536 * 4); hipMemset(d2, 0, 65536 * 4);
  unsigned *flag; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipEvent_t ev[64], evn[64], t0, t1;
  for (int i = 0; i < 64; ++i) { hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); hipEventCreateWithFlags(&evn[i], hipEventDisableTiming | hipEventDisableSystemFence); }
  hipEventCreate(&t0); hipEventCreate(&t1);
  const int N = 200;
  const char *names[] = {"plain chain", "hipEventRecord between (+ waiter)", "same, DisableSystemFence", "hipStreamWriteValue32 between (+ WaitValue32 waiter)", "event record, nobody waits"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(t0, a);
      for (int i = 0; i < N; ++i) {
        k<<<256, 256, 0, a>>>(d, 2000);
        if (mode == 1) { hipEventRecord(ev[i & 63], a); hipStreamWaitEvent(b, ev[i & 63], 0); k<<<16, 256, 0, b>>>(d2, 10); }
        if (mode == 2) { hipEventRecord(evn[i & 63], a); hipStreamWaitEvent(b, evn[i & 63], 0); k<<<16, 256, 0, b>>>(d2, 10); }
        if (mode == 3) { hipStreamWriteValue32(a, flag, (unsigned)(rep * N + i + 1), 0); hipStreamWaitValue32(b, flag, (unsigned)(rep * N + i + 1), hipStreamWaitValueGte, 0xffffffffu); k<<<16, 256, 0, b>>>(d2, 10); }
        if (mode == 4) { hipEventRecord(evn[i & 63], a); }
      }
      hipEventRecord(t1, a);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, t0, t1);
      if (rep == 1) printf("%-56s %7.2f us per kernel\n", names[mode], ms * 1000.f / N);
    }
  }
  return 0;
}

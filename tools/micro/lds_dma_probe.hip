// Probe of `buffer_load_dwordx4 ... lds` on gfx950 (global -> LDS without a register stop):
//   * where do the 64 lanes' 16 bytes land (M0 base + lane * 16?), does the instruction offset move the LDS side too,
//   * what does a lane whose offset lies beyond the descriptor's range write (zeros, or nothing),
//   * is the write visible behind s_waitcnt vmcnt(0) + s_barrier.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/lds_dma_probe.hip -o tools/micro/lds_dma_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr;
__global__ void probe(const float *src, float *out, int n_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -7.f;        // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n_floats * 4, 0x00020000);
  const int lane = threadIdx.x;
  // lanes 0..39: in range; lanes 40..47: beyond the range (should write zeros?); lanes 48..63: offset 0xfffffff0
  unsigned off = lane < 40 ? lane * 16u : (lane < 48 ? (unsigned)n_floats * 4u + lane * 16u : 0xfffffff0u);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + 64), 16, (int)off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 1024 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  probe<<<1, 64, 4096>>>(d, o, n);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
  printf("lds[60..72]:"); for (int i = 60; i < 72; ++i) printf(" %g", r[i]); printf("\n");
  printf("lane 39 slot (lds[64+156..64+160]):"); for (int i = 64 + 156; i < 64 + 160; ++i) printf(" %g", r[i]); printf("\n");
  printf("lane 40 slot (out of range by n):"); for (int i = 64 + 160; i < 64 + 164; ++i) printf(" %g", r[i]); printf("\n");
  printf("lane 50 slot (offset 0xfffffff0):"); for (int i = 64 + 200; i < 64 + 204; ++i) printf(" %g", r[i]); printf("\n");
  printf("lds[64+256..64+260] (behind lane 63):"); for (int i = 64 + 256; i < 64 + 260; ++i) printf(" %g", r[i]); printf("\n");
  return 0;
}

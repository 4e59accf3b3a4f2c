// HBM-bound helper kernels of the backbone/neck path (all activations NDHWC fp32, C % 4 == 0,
// 16-byte vector accesses along C):
//   maxpool 3x3x3 / stride 2 / pad 1 forward + backward   (resnet3d.py:409 nn.MaxPool3d, :471)
//   zero-stuffing for stride-2 data gradients, nearest-upsample adjoint (fpn3d.py:106-108 backward),
//   stride-2 subsample forward/backward (fpn3d.py:121-123: F.max_pool3d(x, 1, stride=2)),
//   NCDHW(3 ch) -> NDHWC(4 ch) volume repack for the stem,
//   table-driven weight preparation (BN-eval folding + repack) and gradient finishing.
#include "mr3d_common.h"

namespace {

constexpr int kT = 256;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// bf16 storage (configs[4]): the same kernels on 4 x bf16 = 8-byte accesses, arithmetic in fp32
using bf16_t = uint16_t;
__device__ __forceinline__ float4 ld4(const bf16_t *p) {
  const uint2 q = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                     __uint_as_float(q.y & 0xffff0000u));
}
__device__ __forceinline__ uint32_t bf16_rne(float f) {     // round to nearest even, NaN stays NaN (v_cvt_pk_bf16_f32)
  return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ void st4(bf16_t *p, float4 v) {
  uint2 q;
  q.x = bf16_rne(v.x) | (bf16_rne(v.y) << 16);
  q.y = bf16_rne(v.z) | (bf16_rne(v.w) << 16);
  *reinterpret_cast<uint2 *>(p) = q;
}

// ---------------------------------------------------------------- maxpool k3 s2 p1
// scan order (kd, kh, kw); the FIRST maximum wins (torch: `val > maxval`), its tap id is stored.
// A workgroup owns one pooled row (n, od, oh): the (<= 3 x 3) input rows it reads are scalars, a thread walks the row's
// (ow, channel quad) items with 32-bit indices (the element-per-thread form with 64-bit div / mod chains per element took
// 137 us for the stem's 335 MB map).
template <typename T>
__global__ __launch_bounds__(2 * kT) void maxpool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                        uint32_t *__restrict__ idx, int N, int D, int H, int W, int C,
                                                        int Do, int Ho, int Wo) {
  const int C4 = C / 4;
  // Output rows are dealt to the XCDs in contiguous runs (workgroup b runs on XCD b & 7): rows (od, oh) and (od, oh + 1) share a
  // third of their nine input rows, and dealt round-robin they sat on different XCDs -- every input row was fetched by up to four
  // L2s (694 MiB of HBM reads per launch for a 320 MiB map, profiles/r03/summary.md).
  const int nrows = N * Do * Ho, per_xcd = (int)gridDim.x >> 3;              // grid = 8 * ceil(nrows / 8)
  int row = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);       // (n, od, oh)
  if (row >= nrows) return;
  const int oh = row % Ho; row /= Ho;
  const int od = row % Do;
  const int n = row / Do;
  const size_t orow = ((size_t)(n * Do + od) * Ho + oh) * (size_t)Wo * C4;
  for (int e = threadIdx.x; e < Wo * C4; e += (int)blockDim.x) {          // (512 threads when a row has more than 256 items)
    const int ow = e / C4, c4 = e - ow * C4;
    // All 27 taps are requested before the first comparison (clamped addresses; a tap outside the map is discarded afterwards):
    // with the loads inside the bounds tests a thread went through nine dependent round trips per output (3 TB/s on a kernel that
    // only moves bytes).
    float4 v[27];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int id = min(max(od * 2 - 1 + kd, 0), D - 1);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = min(max(oh * 2 - 1 + kh, 0), H - 1);
        const T *src = x + ((size_t)(n * D + id) * H + ih) * (size_t)W * C + c4 * 4;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = min(max(ow * 2 - 1 + kw, 0), W - 1);
          v[(kd * 3 + kh) * 3 + kw] = ld4(src + iw * C);
        }
      }
    }
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uint32_t bi0 = 0, bi1 = 0, bi2 = 0, bi3 = 0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const bool okd = (unsigned)(od * 2 - 1 + kd) < (unsigned)D;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const bool okh = okd && (unsigned)(oh * 2 - 1 + kh) < (unsigned)H;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const bool ok = okh && (unsigned)(ow * 2 - 1 + kw) < (unsigned)W;
          const uint32_t tap = (kd * 3 + kh) * 3 + kw;
          const float4 t = v[tap];
          if (ok && t.x > best.x) { best.x = t.x; bi0 = tap; }
          if (ok && t.y > best.y) { best.y = t.y; bi1 = tap; }
          if (ok && t.z > best.z) { best.z = t.z; bi2 = tap; }
          if (ok && t.w > best.w) { best.w = t.w; bi3 = tap; }
        }
      }
    }
    st4(y + (orow + e) * 4, best);
    idx[orow + e] = bi0 | (bi1 << 8) | (bi2 << 16) | (bi3 << 24);
  }
}

// dx[i] = sum over the (<= 8) windows o that cover i and whose argmax is i, of relu'(y[o]) * dy[o].  The ReLU mask of the
// layer in front of the pool comes from the POOLED map: y[o] is the value at the argmax, so y[o] > 0 <=> x[argmax] > 0 --
// 1/8 of the bytes of reading the un-pooled activation (335 MB at the stem of a 512x512x80 volume).
// A workgroup owns one input row (n, id, ih): the (<= 2 x 2) candidate (od, oh) pairs are scalars, a thread walks the row's
// (iw, channel quad) items with 32-bit indices -- the element-per-thread form with 64-bit div / mod chains per element ran at
// 1.7 TB/s (0.29 ms for the stem's 335 MB gradient, on the critical path in front of the stem weight gradient).
template <typename T>
__global__ __launch_bounds__(kT) void maxpool_bwd_kernel(const T *__restrict__ dy, const uint32_t *__restrict__ idx,
                                                        const T *__restrict__ pooled, T *__restrict__ dx, int N,
                                                        int D, int H, int W, int C, int Do, int Ho, int Wo) {
  const int C4 = C / 4;
  int row = blockIdx.x;                                  // (n, id, ih)
  const int ih = row % H; row /= H;
  const int id = row % D;
  const int n = row / D;
  // windows along one axis: i = 2o - 1 + k, k in {0,1,2}  ->  o = (i+1)/2 (k = i+1-2o) and, for odd i, (i-1)/2
  const int od_hi = (id + 1) / 2, nod = (id & 1) ? 2 : 1;
  const int oh_hi = (ih + 1) / 2, noh = (ih & 1) ? 2 : 1;
  T *out = dx + ((size_t)(n * D + id) * H + ih) * (size_t)W * C;
  for (int e = threadIdx.x; e < W * C4; e += kT) {
    const int iw = e / C4, c4 = e - iw * C4;
    const int ow_hi = (iw + 1) / 2, now = (iw & 1) ? 2 : 1;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < nod; ++a) {
      const int od = od_hi - a;
      if ((unsigned)od >= (unsigned)Do) continue;
      const int kd = id + 1 - 2 * od;
      for (int bq = 0; bq < noh; ++bq) {
        const int oh = oh_hi - bq;
        if ((unsigned)oh >= (unsigned)Ho) continue;
        const int kh = ih + 1 - 2 * oh;
        const size_t orow = ((size_t)(n * Do + od) * Ho + oh) * (size_t)Wo;
        for (int cq = 0; cq < now; ++cq) {
          const int ow = ow_hi - cq;
          if ((unsigned)ow >= (unsigned)Wo) continue;
          const int kw = iw + 1 - 2 * ow;
          const uint32_t tap = (kd * 3 + kh) * 3 + kw;
          const size_t o = (orow + ow) * C4 + c4;
          const uint32_t ix = idx[o];
          float4 v = ld4(dy + o * 4);
          if (pooled) {
            const float4 pa = ld4(pooled + o * 4);
            v.x = pa.x > 0.f ? v.x : 0.f; v.y = pa.y > 0.f ? v.y : 0.f; v.z = pa.z > 0.f ? v.z : 0.f; v.w = pa.w > 0.f ? v.w : 0.f;
          }
          if ((ix & 255u) == tap) g.x += v.x;
          if (((ix >> 8) & 255u) == tap) g.y += v.y;
          if (((ix >> 16) & 255u) == tap) g.z += v.z;
          if ((ix >> 24) == tap) g.w += v.w;
        }
      }
    }
    st4(out + (size_t)e * 4, g);
  }
}

// ---------------------------------------------------------------- zero-stuffing: dst[2o] = src[o], else 0
template <typename T>
__global__ __launch_bounds__(kT) void zero_stuff_kernel(const T *__restrict__ src, T *__restrict__ dst, int N, int Do,
                                                       int Ho, int Wo, int D, int H, int W, int C) {
  const int C4 = C / 4;
  const long long total = (long long)N * D * H * W * C4;
  for (long long t = (long long)blockIdx.x * kT + threadIdx.x; t < total; t += (long long)gridDim.x * kT) {
    const int c4 = (int)(t % C4);
    long long r = t / C4;
    const int iw = (int)(r % W); r /= W;
    const int ih = (int)(r % H); r /= H;
    const int id = (int)(r % D);
    const int n = (int)(r / D);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!((id | ih | iw) & 1) && id / 2 < Do && ih / 2 < Ho && iw / 2 < Wo)
      v = ld4(src + ((((size_t)n * Do + id / 2) * Ho + ih / 2) * Wo + iw / 2) * C + c4 * 4);
    st4(dst + (size_t)t * 4, v);
  }
}

// ---------------------------------------------------------------- nearest-upsample adjoint
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  int s = (int)floorf((float)dst * ((float)in / (float)out));
  return s < in - 1 ? s : in - 1;
}
// [lo, hi) = fine indices whose nearest source is u (nearest_src is monotone in dst)
__device__ __forceinline__ void child_range(int u, int in, int out, int &lo, int &hi) {
  int f = (int)(((long long)u * out) / in);
  if (f > out - 1) f = out - 1;
  while (f > 0 && nearest_src(f - 1, in, out) >= u) --f;
  while (f < out && nearest_src(f, in, out) < u) ++f;
  lo = f;
  while (f < out && nearest_src(f, in, out) == u) ++f;
  hi = f;
}

template <typename T>
__global__ __launch_bounds__(kT) void upsample_adjoint_add_kernel(const T *__restrict__ gf, T *__restrict__ gc, int N,
                                                                 int Df, int Hf, int Wf, int Dc, int Hc, int Wc, int C) {
  const int C4 = C / 4;
  const long long total = (long long)N * Dc * Hc * Wc * C4;
  for (long long t = (long long)blockIdx.x * kT + threadIdx.x; t < total; t += (long long)gridDim.x * kT) {
    const int c4 = (int)(t % C4);
    long long r = t / C4;
    const int uw = (int)(r % Wc); r /= Wc;
    const int uh = (int)(r % Hc); r /= Hc;
    const int ud = (int)(r % Dc);
    const int n = (int)(r / Dc);
    float4 g = ld4(gc + (size_t)t * 4);
    if (Df == 2 * Dc && Hf == 2 * Hc && Wf == 2 * Wc) {
      // exact doubling (every level of the pyramid): the eight children are requested together -- inside the run-time-bounded loops
      // below each load waits for the previous one (140 us for the P2 -> P3 step: 1.5 TB/s on a kernel that only moves bytes)
      const T *base = gf + ((((size_t)n * Df + 2 * ud) * Hf + 2 * uh) * Wf + 2 * uw) * C + c4 * 4;
      const size_t sw = C, sh = (size_t)Wf * C, sd = (size_t)Hf * Wf * C;
      const float4 v0 = ld4(base), v1 = ld4(base + sw), v2 = ld4(base + sh), v3 = ld4(base + sh + sw);
      const float4 v4 = ld4(base + sd), v5 = ld4(base + sd + sw), v6 = ld4(base + sd + sh), v7 = ld4(base + sd + sh + sw);
      // (same order of additions as the loops: d, then h, then w)
      g.x += v0.x; g.y += v0.y; g.z += v0.z; g.w += v0.w;
      g.x += v1.x; g.y += v1.y; g.z += v1.z; g.w += v1.w;
      g.x += v2.x; g.y += v2.y; g.z += v2.z; g.w += v2.w;
      g.x += v3.x; g.y += v3.y; g.z += v3.z; g.w += v3.w;
      g.x += v4.x; g.y += v4.y; g.z += v4.z; g.w += v4.w;
      g.x += v5.x; g.y += v5.y; g.z += v5.z; g.w += v5.w;
      g.x += v6.x; g.y += v6.y; g.z += v6.z; g.w += v6.w;
      g.x += v7.x; g.y += v7.y; g.z += v7.z; g.w += v7.w;
      st4(gc + (size_t)t * 4, g);
      continue;
    }
    int d0, d1, h0, h1, w0, w1;
    child_range(ud, Dc, Df, d0, d1);
    child_range(uh, Hc, Hf, h0, h1);
    child_range(uw, Wc, Wf, w0, w1);
    for (int d = d0; d < d1; ++d)
      for (int h = h0; h < h1; ++h)
        for (int w = w0; w < w1; ++w) {
          const float4 v = ld4(gf + ((((size_t)n * Df + d) * Hf + h) * Wf + w) * C + c4 * 4);
          g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
        }
    st4(gc + (size_t)t * 4, g);
  }
}

// ---------------------------------------------------------------- stride-2 subsample (max_pool3d k=1 s=2)
template <bool kBackward, typename T>
__global__ __launch_bounds__(kT) void subsample2_kernel(const T *__restrict__ src, T *__restrict__ dst, int N, int D,
                                                       int H, int W, int C, int Do, int Ho, int Wo) {
  // forward: dst[o] = src[2o] (dst is the coarse grid); backward: dst[2o] += src[o] (dst is the fine grid)
  const int C4 = C / 4;
  const long long total = (long long)N * Do * Ho * Wo * C4;
  for (long long t = (long long)blockIdx.x * kT + threadIdx.x; t < total; t += (long long)gridDim.x * kT) {
    const int c4 = (int)(t % C4);
    long long r = t / C4;
    const int ow = (int)(r % Wo); r /= Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int od = (int)(r % Do);
    const int n = (int)(r / Do);
    const size_t fine = ((((size_t)n * D + od * 2) * H + oh * 2) * W + ow * 2) * C + c4 * 4;
    if (!kBackward) {
      st4(dst + (size_t)t * 4, ld4(src + fine));
    } else {
      float4 a = ld4(dst + fine);
      const float4 v = ld4(src + (size_t)t * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      st4(dst + fine, a);
    }
  }
}

// ---------------------------------------------------------------- volume repack [N,3,D,H,W] -> [N,D,H,W,4]
__global__ __launch_bounds__(kT) void ncdhw3_to_ndhwc4_kernel(const float *__restrict__ src, float *__restrict__ dst, int N,
                                                             long long vox) {
  const long long total = (long long)N * vox;
  for (long long t = (long long)blockIdx.x * kT + threadIdx.x; t < total; t += (long long)gridDim.x * kT) {
    const long long n = t / vox, v = t - n * vox;
    const float *s = src + n * 3 * vox + v;
    st4(dst + (size_t)t * 4, make_float4(s[0], s[vox], s[2 * vox], 0.f));
  }
}

__global__ __launch_bounds__(kT) void axpy_kernel(float *__restrict__ dst, const float *__restrict__ src, float alpha,
                                                 long long n4) {
  for (long long t = (long long)blockIdx.x * kT + threadIdx.x; t < n4; t += (long long)gridDim.x * kT) {
    float4 a = ld4(dst + t * 4);
    const float4 b = ld4(src + t * 4);
    a.x += alpha * b.x; a.y += alpha * b.y; a.z += alpha * b.z; a.w += alpha * b.w;
    st4(dst + t * 4, a);
  }
}

// ---------------------------------------------------------------- table-driven weight prep / grad finish
__device__ __forceinline__ float layer_scale(const mr3d_layer_desc &L, const float *params, int co) {
  if (L.gamma_off < 0) return 1.f;
  return params[L.gamma_off + co] * (1.f / sqrtf(params[L.var_off + co] + L.eps));   // BN eval: gamma * invstd
}

// packed fwd weights wf[tap][ci_pad][co] = s[co] * W[co][ci][tap]; dgrad weights wb[taps-1-tap][co][ci_pad] = same;
// shift[co] = beta - mean*s (BN) or bias or 0; scale[co] = s.
__global__ __launch_bounds__(kT) void prep_weights_kernel(const mr3d_layer_desc *__restrict__ table,
                                                         const float *__restrict__ params, float *__restrict__ packed) {
  const mr3d_layer_desc L = table[blockIdx.y];
  const int total = L.taps * L.ci_pad * L.co;
  for (int e = blockIdx.x * kT + threadIdx.x; e < total; e += gridDim.x * kT) {
    const int co = e % L.co, ci = (e / L.co) % L.ci_pad, tap = e / (L.co * L.ci_pad);
    float v = 0.f;
    if (ci < L.ci) v = layer_scale(L, params, co) * params[L.w_off + ((long long)co * L.ci + ci) * L.taps + tap];
    packed[L.wf_off + e] = v;
    if (L.wb_off >= 0) packed[L.wb_off + ((long long)(L.taps - 1 - tap) * L.co + co) * L.ci_pad + ci] = v;
  }
  if (blockIdx.x == 0)
    for (int co = threadIdx.x; co < L.co; co += kT) {
      const float s = layer_scale(L, params, co);
      float sh = 0.f;
      if (L.gamma_off >= 0) sh = params[L.beta_off + co] - params[L.mean_off + co] * s;
      else if (L.bias_off >= 0) sh = params[L.bias_off + co];
      packed[L.shift_off + co] = sh;
      packed[L.scale_off + co] = s;
    }
}

// grads[w] += s[co] * dwraw[tap][ci][co]   (dwraw = gradient w.r.t. the FOLDED weights)
// Both finishing kernels are latency chains, not bandwidth (3.5 M parameters in all): a thread's gathers from the
// [tap][ci][co] scratch depend on nothing but its index, so they are issued four at a time, and the grid is wide enough that
// the largest layer (128 x 128 x 27) takes 7 rounds instead of 27 (64 blocks per layer: 100 us beside the stem weight gradient).
__global__ __launch_bounds__(kT) void finish_wgrads_kernel(const mr3d_layer_desc *__restrict__ table,
                                                          const float *__restrict__ params,
                                                          const float *__restrict__ scratch, float *__restrict__ grads) {
  const mr3d_layer_desc L = table[blockIdx.y];
  const int total = L.co * L.ci * L.taps;
  const int stride = gridDim.x * kT;
  for (int e0 = blockIdx.x * kT + threadIdx.x; e0 < total; e0 += 4 * stride) {
    float v[4], sc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * stride;
      v[u] = 0.f; sc[u] = 0.f;
      if (e < total) {
        const int tap = e % L.taps, ci = (e / L.taps) % L.ci, co = e / (L.taps * L.ci);
        v[u] = scratch[L.dwraw_off + ((long long)tap * L.ci_pad + ci) * L.co + co];
        sc[u] = layer_scale(L, params, co);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * stride;
      if (e < total) grads[L.w_off + e] += sc[u] * v[u];
    }
  }
}

// per output channel c (one workgroup each):  dot = sum_k W[c][k] * dwraw[k][c]  ( = sum_p dy*z )
//   BN:   dgamma += invstd*(dot - mean*sumdy),  dbeta += sumdy        conv bias: dbias += sumdy
__global__ __launch_bounds__(kT) void finish_affine_kernel(const mr3d_layer_desc *__restrict__ table,
                                                          const float *__restrict__ params,
                                                          const float *__restrict__ scratch, float *__restrict__ grads) {
  const mr3d_layer_desc L = table[blockIdx.y];
  const int c = blockIdx.x;
  if (c >= L.co) return;
  const float sumdy = scratch[L.dbias_off + c];
  if (L.gamma_off < 0) {
    if (L.bias_off >= 0 && threadIdx.x == 0) grads[L.bias_off + c] += sumdy;
    return;
  }
  float dot = 0.f;
  const int K = L.ci * L.taps;
  for (int k0 = threadIdx.x; k0 < K; k0 += 4 * kT) {
    float w[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * kT;
      w[u] = 0.f; g[u] = 0.f;
      if (k < K) {
        const int tap = k % L.taps, ci = k / L.taps;
        w[u] = params[L.w_off + (long long)c * K + k];
        g[u] = scratch[L.dwraw_off + ((long long)tap * L.ci_pad + ci) * L.co + c];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) dot += w[u] * g[u];
  }
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  __shared__ float part[kT / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) {
    dot = (part[0] + part[1]) + (part[2] + part[3]);
    const float invstd = 1.f / sqrtf(params[L.var_off + c] + L.eps);
    grads[L.gamma_off + c] += invstd * (dot - params[L.mean_off + c] * sumdy);
    grads[L.beta_off + c] += sumdy;
  }
}

// bf16 operand packing for conv3d_bf16.hip / conv3d_wgrad_bf16.hip: a B fragment of v_mfma_f32_32x32x16_bf16 is 8
// consecutive K values of one N column, so K-groups of 8 are the innermost run:
//   forward  wf16[tap][ci16/8][co32][8]           = s[co] * W[co][ci][tap]   (K = input channel, N = output channel)
//   dgrad    wb16[taps-1-tap][co16/8][ci32][8]    = same value               (K = output channel, N = input channel)
// ci16 / co16 = channel counts rounded up to 16, co32 / ci32 to 32 (zero filled).  wf_off / wb_off of the table are
// offsets in bf16 ELEMENTS into `packed16`; shift / scale stay in the fp32 packed arena (mr3d_prepare_weights).
__global__ __launch_bounds__(kT) void prep_weights_bf16_kernel(const mr3d_layer_desc *__restrict__ table,
                                                              const float *__restrict__ params, bf16_t *__restrict__ packed) {
  const mr3d_layer_desc L = table[blockIdx.y];
  const int ci16 = (L.ci_pad + 15) / 16 * 16, co32 = (L.co + 31) / 32 * 32;
  const int co16 = (L.co + 15) / 16 * 16, ci32 = (L.ci_pad + 31) / 32 * 32;
  const int total_f = L.taps * ci16 * co32;
  for (int e = blockIdx.x * kT + threadIdx.x; e < total_f; e += gridDim.x * kT) {
    const int j = e & 7, co = (e >> 3) % co32, g = (e >> 3) / co32 % (ci16 / 8), tap = e / (ci16 * co32);
    const int ci = g * 8 + j;
    float v = 0.f;
    if (ci < L.ci && co < L.co) v = layer_scale(L, params, co) * params[L.w_off + ((long long)co * L.ci + ci) * L.taps + tap];
    packed[L.wf_off + e] = (bf16_t)bf16_rne(v);
  }
  if (L.wb_off < 0) return;
  const int total_b = L.taps * co16 * ci32;
  for (int e = blockIdx.x * kT + threadIdx.x; e < total_b; e += gridDim.x * kT) {
    const int j = e & 7, ci = (e >> 3) % ci32, g = (e >> 3) / ci32 % (co16 / 8), tapb = e / (co16 * ci32);
    const int co = g * 8 + j, tap = L.taps - 1 - tapb;
    float v = 0.f;
    if (ci < L.ci && co < L.co) v = layer_scale(L, params, co) * params[L.w_off + ((long long)co * L.ci + ci) * L.taps + tap];
    packed[L.wb_off + e] = (bf16_t)bf16_rne(v);
  }
}

inline int grid_for(long long total) {
  long long b = (total + kT - 1) / kT;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int mr3d_maxpool3d_forward(const float *x, float *y, uint32_t *argmax, int n, int d, int h, int w, int c,
                                      mr3d_stream_t s) {
  if (!x || !y || !argmax || c % 4) return MR3D_ERR_INVALID_ARG;
  const int Do = (d - 1) / 2 + 1, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
  maxpool_fwd_kernel<<<(unsigned)((n * Do * Ho + 7) / 8 * 8), Wo * (c / 4) > kT ? 2 * kT : kT, 0, mr3d_stream(s)>>>(x, y, argmax, n, d, h, w, c, Do, Ho, Wo);
  return mr3d_launch_status();
}

extern "C" int mr3d_maxpool3d_backward(const float *dy, const uint32_t *argmax, const float *pooled, float *dx, int n, int d,
                                       int h, int w, int c, mr3d_stream_t s) {
  if (!dy || !dx || !argmax || c % 4) return MR3D_ERR_INVALID_ARG;
  const int Do = (d - 1) / 2 + 1, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
  if ((long long)n * d * h > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  maxpool_bwd_kernel<<<(unsigned)(n * d * h), kT, 0, mr3d_stream(s)>>>(dy, argmax, pooled, dx, n, d, h, w, c, Do, Ho, Wo);
  return mr3d_launch_status();
}

extern "C" int mr3d_zero_stuff2(const float *src, float *dst, int n, int do_, int ho, int wo, int d, int h, int w, int c,
                                mr3d_stream_t s) {
  if (!src || !dst || c % 4) return MR3D_ERR_INVALID_ARG;
  zero_stuff_kernel<<<grid_for((long long)n * d * h * w * c / 4), kT, 0, mr3d_stream(s)>>>(src, dst, n, do_, ho, wo, d, h, w, c);
  return mr3d_launch_status();
}

extern "C" int mr3d_upsample_nearest_adjoint_add(const float *grad_fine, float *grad_coarse, int n, int df, int hf, int wf,
                                                 int dc, int hc, int wc, int c, mr3d_stream_t s) {
  if (!grad_fine || !grad_coarse || c % 4) return MR3D_ERR_INVALID_ARG;
  upsample_adjoint_add_kernel<<<grid_for((long long)n * dc * hc * wc * c / 4), kT, 0, mr3d_stream(s)>>>(grad_fine, grad_coarse, n,
                                                                                                     df, hf, wf, dc, hc, wc, c);
  return mr3d_launch_status();
}

extern "C" int mr3d_subsample2_forward(const float *x, float *y, int n, int d, int h, int w, int c, mr3d_stream_t s) {
  if (!x || !y || c % 4) return MR3D_ERR_INVALID_ARG;
  const int Do = (d - 1) / 2 + 1, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
  subsample2_kernel<false><<<grid_for((long long)n * Do * Ho * Wo * c / 4), kT, 0, mr3d_stream(s)>>>(x, y, n, d, h, w, c, Do, Ho, Wo);
  return mr3d_launch_status();
}

extern "C" int mr3d_subsample2_backward_add(const float *dy, float *dx, int n, int d, int h, int w, int c, mr3d_stream_t s) {
  if (!dy || !dx || c % 4) return MR3D_ERR_INVALID_ARG;
  const int Do = (d - 1) / 2 + 1, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
  subsample2_kernel<true><<<grid_for((long long)n * Do * Ho * Wo * c / 4), kT, 0, mr3d_stream(s)>>>(dy, dx, n, d, h, w, c, Do, Ho, Wo);
  return mr3d_launch_status();
}

extern "C" int mr3d_volume_to_ndhwc4(const float *ncdhw3, float *ndhwc4, int n, int d, int h, int w, mr3d_stream_t s) {
  if (!ncdhw3 || !ndhwc4) return MR3D_ERR_INVALID_ARG;
  const long long vox = (long long)d * h * w;
  ncdhw3_to_ndhwc4_kernel<<<grid_for((long long)n * vox), kT, 0, mr3d_stream(s)>>>(ncdhw3, ndhwc4, n, vox);
  return mr3d_launch_status();
}

// out = act > 0 ? g : 0 (the ReLU backward of a layer whose forward output `act` was kept), optionally with the
// per-channel column sums of `out` (bias gradient) from the same pass.  Rows are [.., C] with C % 4 == 0 and
// 256 % (C/4) == 0 or (C/4) % 256 == 0 is NOT required: a thread's channel quad changes along its grid-stride
// walk unless the stride is a multiple of C/4, so the launcher rounds the grid to make it one.
__global__ __launch_bounds__(kT) void relu_mask_kernel(const float4 *__restrict__ g, const float4 *__restrict__ act,
                                                       float4 *__restrict__ out, long long n4, int c4, float *__restrict__ dbias) {
  __shared__ float4 red[kT];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long stride = (long long)gridDim.x * kT;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n4; i += stride) {
    const float4 a = act[i];
    float4 v = g[i];
    v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
    out[i] = v;
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (!dbias) return;
  // stride % c4 == 0 and kT % c4 == 0 or c4 % kT == 0: thread t always sees channel quad (blockIdx.x*kT + t) % c4
  red[threadIdx.x] = acc;
  __syncthreads();
  const int q0 = (int)(((long long)blockIdx.x * kT) % c4);
  if (c4 <= kT) {
    // threads t, t + c4, t + 2*c4, ... share a quad (kT % c4 == 0)
    if (threadIdx.x < c4) {
      float4 s = red[threadIdx.x];
      for (int t = threadIdx.x + c4; t < kT; t += c4) { s.x += red[t].x; s.y += red[t].y; s.z += red[t].z; s.w += red[t].w; }
      const int q = (q0 + threadIdx.x) % c4;
      atomicAdd(dbias + 4 * q + 0, s.x); atomicAdd(dbias + 4 * q + 1, s.y);
      atomicAdd(dbias + 4 * q + 2, s.z); atomicAdd(dbias + 4 * q + 3, s.w);
    }
  } else {
    const int q = (q0 + threadIdx.x) % c4;
    atomicAdd(dbias + 4 * q + 0, acc.x); atomicAdd(dbias + 4 * q + 1, acc.y);
    atomicAdd(dbias + 4 * q + 2, acc.z); atomicAdd(dbias + 4 * q + 3, acc.w);
  }
}

extern "C" int mr3d_relu_mask(const float *g, const float *act, float *out, long long rows, int c, float *dbias,
                              mr3d_stream_t s) {
  if (!g || !act || !out || rows <= 0 || c <= 0 || c % 4) return MR3D_ERR_INVALID_ARG;
  const int c4 = c / 4;
  if (dbias && !(kT % c4 == 0 || c4 % kT == 0)) return MR3D_ERR_UNSUPPORTED;
  const long long n4 = rows * c4;
  long long blocks = (n4 + kT - 1) / kT;
  if (blocks > 4096) blocks = 4096;
  if (dbias && c4 > kT) {                 // grid stride must be a multiple of c4
    const long long m = c4 / kT;
    blocks = blocks / m * m;
    if (blocks < m) blocks = m;
  }
  relu_mask_kernel<<<(unsigned)blocks, kT, 0, mr3d_stream(s)>>>(reinterpret_cast<const float4 *>(g),
                                                               reinterpret_cast<const float4 *>(act),
                                                               reinterpret_cast<float4 *>(out), n4, c4, dbias);
  return mr3d_launch_status();
}

extern "C" int mr3d_axpy(float *dst, const float *src, float alpha, long long count, mr3d_stream_t s) {
  if (!dst || !src || count % 4) return MR3D_ERR_INVALID_ARG;
  axpy_kernel<<<grid_for(count / 4), kT, 0, mr3d_stream(s)>>>(dst, src, alpha, count / 4);
  return mr3d_launch_status();
}

extern "C" int mr3d_prepare_weights(const mr3d_layer_desc *table, int num_layers, const float *params, float *packed,
                                    mr3d_stream_t s) {
  if (!table || !params || !packed || num_layers <= 0) return MR3D_ERR_INVALID_ARG;
  prep_weights_kernel<<<dim3(64, num_layers), kT, 0, mr3d_stream(s)>>>(table, params, packed);
  return mr3d_launch_status();
}

extern "C" int mr3d_finish_grads(const mr3d_layer_desc *table, int num_layers, int max_cout, const float *params,
                                 const float *scratch, float *grads, mr3d_stream_t s) {
  if (!table || !params || !scratch || !grads || num_layers <= 0 || max_cout <= 0) return MR3D_ERR_INVALID_ARG;
  finish_wgrads_kernel<<<dim3(256, num_layers), kT, 0, mr3d_stream(s)>>>(table, params, scratch, grads);
  finish_affine_kernel<<<dim3(max_cout, num_layers), kT, 0, mr3d_stream(s)>>>(table, params, scratch, grads);
  return mr3d_launch_status();
}


// ------------------------------------------------------------------------------------------------ bf16 storage variants
extern "C" int mr3d_maxpool3d_forward_bf16(const void *x, void *y, uint32_t *argmax, int n, int d, int h, int w, int c,
                                           mr3d_stream_t s) {
  if (!x || !y || !argmax || c % 4) return MR3D_ERR_INVALID_ARG;
  const int Do = (d - 1) / 2 + 1, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
  maxpool_fwd_kernel<<<(unsigned)((n * Do * Ho + 7) / 8 * 8), Wo * (c / 4) > kT ? 2 * kT : kT, 0, mr3d_stream(s)>>>((const bf16_t *)x, (bf16_t *)y, argmax, n, d,
                                                                                                                          h, w, c, Do, Ho, Wo);
  return mr3d_launch_status();
}

extern "C" int mr3d_maxpool3d_backward_bf16(const void *dy, const uint32_t *argmax, const void *pooled, void *dx, int n, int d,
                                            int h, int w, int c, mr3d_stream_t s) {
  if (!dy || !dx || !argmax || c % 4) return MR3D_ERR_INVALID_ARG;
  const int Do = (d - 1) / 2 + 1, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
  if ((long long)n * d * h > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  maxpool_bwd_kernel<<<(unsigned)(n * d * h), kT, 0, mr3d_stream(s)>>>((const bf16_t *)dy, argmax, (const bf16_t *)pooled, (bf16_t *)dx, n, d,
                                                                       h, w, c, Do, Ho, Wo);
  return mr3d_launch_status();
}

extern "C" int mr3d_zero_stuff2_bf16(const void *src, void *dst, int n, int do_, int ho, int wo, int d, int h, int w, int c,
                                     mr3d_stream_t s) {
  if (!src || !dst || c % 4) return MR3D_ERR_INVALID_ARG;
  zero_stuff_kernel<<<grid_for((long long)n * d * h * w * c / 4), kT, 0, mr3d_stream(s)>>>((const bf16_t *)src, (bf16_t *)dst, n, do_,
                                                                                        ho, wo, d, h, w, c);
  return mr3d_launch_status();
}

extern "C" int mr3d_upsample_nearest_adjoint_add_bf16(const void *grad_fine, void *grad_coarse, int n, int df, int hf, int wf,
                                                      int dc, int hc, int wc, int c, mr3d_stream_t s) {
  if (!grad_fine || !grad_coarse || c % 4) return MR3D_ERR_INVALID_ARG;
  upsample_adjoint_add_kernel<<<grid_for((long long)n * dc * hc * wc * c / 4), kT, 0, mr3d_stream(s)>>>(
      (const bf16_t *)grad_fine, (bf16_t *)grad_coarse, n, df, hf, wf, dc, hc, wc, c);
  return mr3d_launch_status();
}

extern "C" int mr3d_subsample2_forward_bf16(const void *x, void *y, int n, int d, int h, int w, int c, mr3d_stream_t s) {
  if (!x || !y || c % 4) return MR3D_ERR_INVALID_ARG;
  const int Do = (d - 1) / 2 + 1, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
  subsample2_kernel<false><<<grid_for((long long)n * Do * Ho * Wo * c / 4), kT, 0, mr3d_stream(s)>>>((const bf16_t *)x, (bf16_t *)y, n,
                                                                                                  d, h, w, c, Do, Ho, Wo);
  return mr3d_launch_status();
}

extern "C" int mr3d_subsample2_backward_add_bf16(const void *dy, void *dx, int n, int d, int h, int w, int c, mr3d_stream_t s) {
  if (!dy || !dx || c % 4) return MR3D_ERR_INVALID_ARG;
  const int Do = (d - 1) / 2 + 1, Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
  subsample2_kernel<true><<<grid_for((long long)n * Do * Ho * Wo * c / 4), kT, 0, mr3d_stream(s)>>>((const bf16_t *)dy, (bf16_t *)dx, n,
                                                                                                 d, h, w, c, Do, Ho, Wo);
  return mr3d_launch_status();
}

extern "C" int mr3d_prepare_weights_bf16(const mr3d_layer_desc *table, int num_layers, const float *params, void *packed16,
                                         mr3d_stream_t s) {
  if (!table || !params || !packed16 || num_layers <= 0) return MR3D_ERR_INVALID_ARG;
  prep_weights_bf16_kernel<<<dim3(64, num_layers), kT, 0, mr3d_stream(s)>>>(table, params, (bf16_t *)packed16);
  return mr3d_launch_status();
}

// ------------------------------------------------------------------------------------------
// Voxel liveness of an NDHWC map: ONE BIT per position (bit p & 31 of word p >> 5, p = ((n*D + d)*H + h)*W + w), set iff one of the
// position's channels is non-zero (NaN counts, -0 does not).  The large-map data-gradient convolution reads it
// (mr3d_conv3d_desc.x_live): the gradient that reaches the RPN conv / the FPN output convs in detector training is zero outside the
// sampled anchors' / rois' neighbourhoods, and a tile whose halo holds no live voxel is its epilogue.  A wave owns 64 positions: the
// 16-byte pieces of 64 / P positions per load instruction (P = pieces per position: fully coalesced), a ballot per instruction.
// HBM-bound: one pass over the map.
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void voxel_liveness_kernel(const unsigned char *__restrict__ x, long long total, int row_bytes, int P,
                                                            unsigned keep, unsigned *__restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long base = wave * 64;
  if (base >= total) return;
  const int PP = 64 / P;                            // positions per load instruction
  const int g = lane / P, piece = lane - g * P;
  const unsigned long long gmask = P == 64 ? ~0ull : ((1ull << P) - 1ull);
  unsigned long long acc = 0;
  for (int it = 0; it < P; ++it) {
    const long long pos = base + (long long)it * PP + g;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (pos < total) v = *reinterpret_cast<const uint4 *>(x + pos * row_bytes + piece * 16);
    const unsigned long long m = __ballot(((v.x | v.y | v.z | v.w) & keep) != 0);
    for (int q = 0; q < PP; ++q)
      if ((m >> (q * P)) & gmask) acc |= 1ull << (it * PP + q);
  }
  if (lane == 0) {
    bits[wave * 2] = (unsigned)acc;
    bits[wave * 2 + 1] = (unsigned)(acc >> 32);
  }
}
}  // namespace

extern "C" long long mr3d_voxel_liveness_words(long long positions) { return positions <= 0 ? 0 : ((positions + 63) / 64) * 2 + 2; }

extern "C" int mr3d_voxel_liveness(const void *x, int bf16, int n, int d, int h, int w, int c, unsigned *bits, mr3d_stream_t s) {
  if (!x || !bits || n <= 0 || d <= 0 || h <= 0 || w <= 0 || c <= 0 || ((uintptr_t)x & 15)) return MR3D_ERR_INVALID_ARG;
  const int row_bytes = c * (bf16 ? 2 : 4);
  const int P = row_bytes / 16;
  if (row_bytes % 16 != 0 || P < 1 || P > 64 || (64 % P) != 0) return MR3D_ERR_UNSUPPORTED;
  const long long total = (long long)n * d * h * w;
  const long long waves = (total + 63) / 64;
  if ((waves + 3) / 4 > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  if (hipMemsetAsync(bits + waves * 2, 0, 8, mr3d_stream(s)) != hipSuccess) return MR3D_ERR_LAUNCH;      // the two words of padding
  voxel_liveness_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, mr3d_stream(s)>>>((const unsigned char *)x, total, row_bytes, P,
                                                                                 bf16 ? 0x7fff7fffu : 0x7fffffffu, bits);
  return mr3d_launch_status();
}

// ResNet3D stem on the bf16 matrix cores (BASELINE configs[4]): Conv3d(3, 16, kernel 7, stride (1,2,2), padding 3) +
// folded BN + ReLU forward, and its weight gradient (mmdet/models/backbones/resnet3d.py:397-409,467-469).
// Input: the reference's planar fp32 volume [N,3,D,H,W] (converted to bf16 while it is staged); output / dy: bf16
// NDHWC [N,D,H/2,W/2,16]; fp32 accumulation (v_mfma_f32_16x16x32_bf16: N = 16 output channels is exactly one tile).
//
// Forward.  GEMM K is walked as 49 (kd,kh) steps of 32 = 4 channel slots x 8 kw slots (3 x 7 real: 66 % of the issued
// MFMA work is useful, still ~4x the fp32 kernel's rate).  A lane's A fragment -- 8 consecutive kw taps of one channel for
// one output position -- is 8 CONSECUTIVE input columns starting at 2*ow - 3: one 16-byte LDS read from the planar bf16
// slab [c][d][h][w] at a 4-byte-aligned address (gfx950's DS unit takes unaligned b128 accesses).  Weights are pre-packed
// [kd*7+kh][c4][cout][kw8] bf16 (50 KB, L2-resident); a lane's B fragment is one 16-byte global load.
//
// Weight gradient.  GEMM M = the 1176 (kd,kh,c,kw8) rows in 74 tiles of 16, N = 16 output channels, K = positions, 32
// consecutive ow per step.  A fragment = 8 consecutive positions of one (c, tap): with the slab's even and odd input
// columns stored in separate planes (iw = 2*ow + kw - 3 has the parity of kw) these are again 8 consecutive bf16 -- one
// unaligned 16-byte read.  B = dy [position][16] read with the transposing LDS read (ds_read_b64_tr_b16).  8 waves split the
// 74 row tiles (+ one tile of ones: sum_p dy); persistent workgroups, fp32 atomics at the end into the generic packed
// layout [343][4][16] that mr3d_finish_grads consumes.
#include "mr3d_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
using bf16_t = uint16_t;
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

__device__ __forceinline__ bf16_t f2bf(float f) {          // round to nearest even, NaN stays NaN: v_cvt_pk_bf16_f32
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}

struct __attribute__((packed, aligned(2))) U16x8 { uint16_t v[8]; };       // 16 bytes at 2-byte alignment

__device__ __forceinline__ bf16x8 lds_frag_unaligned(const unsigned char *p) {
  const U16x8 raw = *reinterpret_cast<const U16x8 *>(p);
  return __builtin_bit_cast(bf16x8, raw);
}

// ------------------------------------------------------------------------------------------------ weight packing
// wf: generic packing [343][4][16] fp32 (BN scale folded, channel 3 = 0) -> [49][4][16][8] bf16 (kw 7 = 0)
__global__ __launch_bounds__(256) void stem_pack_bf16_kernel(const float *__restrict__ wf, bf16_t *__restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 49 * 4 * 16 * 8) return;
  const int kw = e & 7, co = (e >> 3) & 15, c = (e >> 7) & 3, kdkh = e >> 9;
  float v = 0.f;
  if (kw < 7 && c < 3) v = wf[((kdkh * 7 + kw) * 4 + c) * 16 + co];
  out[e] = f2bf(v);
}

// ------------------------------------------------------------------------------------------------ forward
struct FwdGeo {
  static constexpr int TD = 2, TH = 8, TW = 16;
  static constexpr int HD = TD + 6, HH = 2 * (TH - 1) + 7, HW = 40;          // 38 columns used (+1 read by the zero kw slot)
  static constexpr int PLANE = HD * HH * HW;                                  // bf16 elements per channel
  static constexpr int LDS_BYTES = 3 * PLANE * 2 + 64;
};

struct StemBfArgs {
  const float *x; const bf16_t *w16; const float *shift; bf16_t *y;
  int N, D, H, W, Ho, Wo, relu;
};

__global__ __launch_bounds__(256, 2) void stem_fwd_bf16_kernel(StemBfArgs p, int tiles_d, int tiles_h, int tiles_w) {
  using G = FwdGeo;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  bf16_t *slab = reinterpret_cast<bf16_t *>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  int b = blockIdx.x;
  const int tw_i = b % tiles_w; b /= tiles_w;
  const int th_i = b % tiles_h; b /= tiles_h;
  const int td_i = b % tiles_d; b /= tiles_d;
  const int n = b;
  const int d0 = td_i * G::TD, h0 = th_i * G::TH, w0 = tw_i * G::TW;
  const int id0 = d0 - 3, ih0 = 2 * h0 - 3, iw0 = 2 * w0 - 3;
  const size_t vox = (size_t)p.D * p.H * p.W;

  // ---- stage the slab: threads take (row, column pair); global pairs start at an even column (one 8-byte load when W is
  // even).  Loads are issued in batches of 8 before any of them is consumed: a load -> convert -> store loop would keep ONE
  // load in flight per thread and cost a full memory latency per pair (measured: 22 us per workgroup that way).
  constexpr int ROWS = 3 * G::HD * G::HH, PAIRS = 20, TOTAL = ROWS * PAIRS, NIT = (TOTAL + 255) / 256;
  const bool even_w = (p.W & 1) == 0;
  for (int base = 0; base < NIT; base += 8) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int it = tid + (base + u) * 256;
      v[u] = make_float2(0.f, 0.f);
      if (it < TOTAL) {
        const int row = it / PAIRS, pr = it - row * PAIRS;
        const int hh = row % G::HH, hd = (row / G::HH) % G::HD, c = row / (G::HH * G::HD);
        const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + 2 * pr - 1;
        if ((unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H) {
          const float *src = p.x + ((size_t)n * 3 + c) * vox + ((size_t)id * p.H + ih) * p.W;
          if (even_w && iw >= 0 && iw + 1 < p.W) {
            v[u] = *reinterpret_cast<const float2 *>(src + iw);
          } else {
            if ((unsigned)iw < (unsigned)p.W) v[u].x = src[iw];
            if ((unsigned)(iw + 1) < (unsigned)p.W) v[u].y = src[iw + 1];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int it = tid + (base + u) * 256;
      if (it < TOTAL) {
        const int row = it / PAIRS, pr = it - row * PAIRS;
        bf16_t *dst = slab + (size_t)row * G::HW;          // slab columns 2*pr - 1, 2*pr
        if (pr >= 1) dst[2 * pr - 1] = f2bf(v[u].x);
        dst[2 * pr] = f2bf(v[u].y);
      }
    }
  }
  __syncthreads();

  f32x4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // this wave's four (td, th) rows of 16 outputs; lane (r, q): output column r, channel slot q (slot 3: zeros)
  int abase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mt = wave * 4 + i, td = mt >> 3, th = mt & 7;
    abase[i] = ((((q < 3 ? q : 0) * G::HD + td) * G::HH + 2 * th) * G::HW + 2 * r) * 2;
  }
  const bf16x8 *wl = reinterpret_cast<const bf16x8 *>(p.w16) + q * 16 + r;
  bf16x8 zero;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.0f;
  bf16x8 bcur = wl[0];
#pragma unroll 1
  for (int kd = 0; kd < 7; ++kd) {
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
      const int ks = kd * 7 + kh;
      const bf16x8 bnext = wl[(size_t)((ks + 1 < 49) ? ks + 1 : 48) * 64];
      const int off = ((kd * G::HH + kh) * G::HW) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x8 a = lds_frag_unaligned(lds + abase[i] + off);
        if (q == 3) a = zero;
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bcur, acc[i], 0, 0, 0);
      }
      bcur = bnext;
    }
  }
  // ---- epilogue: C/D layout of 16x16: column (cout) = lane & 15, row (output column) = 4 * (lane >> 4) + reg
  const float sh = p.shift ? p.shift[r] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mt = wave * 4 + i, td = mt >> 3, th = mt & 7;
    const int od = d0 + td, oh = h0 + th;
    if (od >= p.D || oh >= p.Ho) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ow = w0 + 4 * q + e;
      if (ow < p.Wo) {
        float v = acc[i][e] + sh;
        if (p.relu) v = fmaxf(v, 0.f);
        p.y[((((size_t)n * p.D + od) * p.Ho + oh) * p.Wo + ow) * 16 + r] = f2bf(v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgGeo {
  static constexpr int TD = 4, TH = 4, TW = 32;
  static constexpr int HD = TD + 6, HH = 2 * (TH - 1) + 7;                    // 10 x 13 rows
  static constexpr int HALF = 40;                                             // columns per parity plane (35 used + slack)
  static constexpr int ROW = 2 * HALF;                                        // even plane | odd plane
  static constexpr int PLANE = HD * HH * ROW;
  static constexpr int DYROW = 24;                                            // bf16 per dy row: 16 + 8 of padding (48 B)
  static constexpr int M = TD * TH * TW, KS = TD * TH;
  static constexpr int SLAB_BYTES = 3 * PLANE * 2, DY_BYTES = M * DYROW * 2;
  static constexpr int LDS_BYTES = SLAB_BYTES + DY_BYTES + 64;
  static constexpr int ROWS = 49 * 24, MT = (ROWS + 15) / 16;                 // 1176 GEMM rows -> 74 tiles (+ 1 of ones)
  static constexpr int MTW = 10;                                              // row tiles per wave (8 waves x 10 >= 75)
};

struct StemBfWgArgs {
  const float *x; const bf16_t *dy; float *dw; float *dbias;
  int N, D, H, W, Ho, Wo;
};

__global__ __launch_bounds__(512, 2) void stem_wgrad_bf16_kernel(StemBfWgArgs p, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  using G = WgGeo;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  bf16_t *slab = reinterpret_cast<bf16_t *>(lds);
  unsigned char *dyt = lds + G::SLAB_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const size_t vox = (size_t)p.D * p.H * p.W;

  // per-lane slab offset (bytes) of GEMM row R = 16*m + r for every row tile of this wave, k = 8q (+ j) folded in
  int aoff[G::MTW];
  bool ones[G::MTW], live[G::MTW];
#pragma unroll
  for (int i = 0; i < G::MTW; ++i) {
    const int m = wave + 8 * i;                          // wave-uniform tile index
    const int R = 16 * m + r;
    const int kdkh = R / 24, rem = R - kdkh * 24, c = rem >> 3, kw = rem & 7;
    const int kd = kdkh / 7, kh = kdkh - kd * 7;
    ones[i] = (m == G::MT);                              // the extra tile: a fragment of ones -> sum_p dy
    live[i] = (m < G::MT) && R < G::ROWS && kw < 7;
    // input column 2*ow + kw (relative to the slab origin): parity plane kw & 1, index ow + (kw >> 1)
    const int cc = live[i] ? c : 0, kk = live[i] ? kw : 0, dd = live[i] ? kd : 0, hh_ = live[i] ? kh : 0;
    aoff[i] = (((cc * G::HD + dd) * G::HH + hh_) * G::ROW + (kk & 1) * G::HALF + (kk >> 1) + 8 * q) * 2;
  }
  // transposing read of dy [position][16]: group q reads positions 8q..8q+7; inside the group lane 4a+b gives row a, columns 4b..4b+3
  const int i16 = lane & 15, ra = i16 >> 2, cb = i16 & 3;
  const int b_lane = (8 * q + ra) * (G::DYROW * 2) + cb * 8;

  f32x4_t acc[G::MTW];
#pragma unroll
  for (int i = 0; i < G::MTW; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8 one8, zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { one8[e] = (__bf16)1.0f; zero8[e] = (__bf16)0.0f; }

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b = tile;
    const int tw_i = b % tiles_w; b /= tiles_w;
    const int th_i = b % tiles_h; b /= tiles_h;
    const int td_i = b % tiles_d; b /= tiles_d;
    const int n = b;
    const int d0 = td_i * G::TD, h0 = th_i * G::TH, w0 = tw_i * G::TW;
    const int id0 = d0 - 3, ih0 = 2 * h0 - 3, iw0 = 2 * w0 - 3;
    __syncthreads();
    // slab: column s = iw - iw0; even s -> plane 0 index s/2, odd s -> plane 1 index (s-1)/2.  Threads take aligned global pairs
    // (iw0 + s0 even for s0 = 2*pr - 1): s0 odd -> plane 1 index pr - 1, s0 + 1 even -> plane 0 index pr.  Loads in batches
    // of 8 (see the forward kernel).
    constexpr int ROWS = 3 * G::HD * G::HH, PAIRS = G::HALF, TOTAL = ROWS * PAIRS, NIT = (TOTAL + 511) / 512;
    const bool even_w = (p.W & 1) == 0;
    for (int base = 0; base < NIT; base += 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int it = tid + (base + u) * 512;
        v[u] = make_float2(0.f, 0.f);
        if (it < TOTAL) {
          const int row = it / PAIRS, pr = it - row * PAIRS;
          const int hh = row % G::HH, hd = (row / G::HH) % G::HD, c = row / (G::HH * G::HD);
          const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + 2 * pr - 1;
          if ((unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H) {
            const float *src = p.x + ((size_t)n * 3 + c) * vox + ((size_t)id * p.H + ih) * p.W;
            if (even_w && iw >= 0 && iw + 1 < p.W) {
              v[u] = *reinterpret_cast<const float2 *>(src + iw);
            } else {
              if ((unsigned)iw < (unsigned)p.W) v[u].x = src[iw];
              if ((unsigned)(iw + 1) < (unsigned)p.W) v[u].y = src[iw + 1];
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int it = tid + (base + u) * 512;
        if (it < TOTAL) {
          const int row = it / PAIRS, pr = it - row * PAIRS;
          bf16_t *dst = slab + (size_t)row * G::ROW;
          if (pr >= 1) dst[G::HALF + pr - 1] = f2bf(v[u].x);
          dst[pr] = f2bf(v[u].y);
        }
      }
    }
    // dy tile [m][16] (m = (td*TH + th)*32 + tw), zeros outside the volume
    for (int it = tid; it < G::M * 2; it += 512) {
      const int m = it >> 1, half = it & 1;
      const int tw = m & 31, th = (m >> 5) % G::TH, td = m / (32 * G::TH);
      const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (od < p.D && oh < p.Ho && ow < p.Wo)
        v = *reinterpret_cast<const uint4 *>(p.dy + ((((size_t)n * p.D + od) * p.Ho + oh) * p.Wo + ow) * 16 + half * 8);
      *reinterpret_cast<uint4 *>(dyt + m * (G::DYROW * 2) + half * 16) = v;
    }
    __syncthreads();
    for (int ks = 0; ks < G::KS; ++ks) {
      const int th = ks % G::TH, td = ks / G::TH;
      const unsigned char *yb = dyt + ks * 32 * (G::DYROW * 2) + b_lane;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(yb));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(yb + 4 * (G::DYROW * 2)));
      const bf16x8 bf = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      const int rowoff = ((td * G::HH + 2 * th) * G::ROW) * 2;
#pragma unroll
      for (int i = 0; i < G::MTW; ++i) {
        if (wave + 8 * i > G::MT) continue;               // wave-uniform
        bf16x8 a = lds_frag_unaligned(lds + aoff[i] + rowoff);
        if (!live[i]) a = ones[i] ? one8 : zero8;
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf, acc[i], 0, 0, 0);
      }
    }
  }
  // ---- write-out: column (cout) = lane & 15, GEMM row = 16*m + 4*(lane>>4) + reg
#pragma unroll
  for (int i = 0; i < G::MTW; ++i) {
    const int m = wave + 8 * i;
    if (m > G::MT) continue;
    if (m == G::MT) {
      if (p.dbias && q == 0) atomicAdd(p.dbias + r, acc[i][0]);
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int R = 16 * m + 4 * q + e;
      const int kdkh = R / 24, rem = R - kdkh * 24, c = rem >> 3, kw = rem & 7;
      if (R < G::ROWS && kw < 7) atomicAdd(p.dw + ((size_t)(kdkh * 7 + kw) * 4 + c) * 16 + r, acc[i][e]);
    }
  }
}

}  // namespace

extern "C" int mr3d_stem_pack_weights_bf16(const float *w_packed, void *w16, mr3d_stream_t stream_) {
  if (!w_packed || !w16) return MR3D_ERR_INVALID_ARG;
  stem_pack_bf16_kernel<<<(49 * 4 * 16 * 8 + 255) / 256, 256, 0, mr3d_stream(stream_)>>>(w_packed, (bf16_t *)w16);
  return mr3d_launch_status();
}

extern "C" int mr3d_stem_conv_forward_bf16(const float *x_ncdhw, const void *w16, const float *shift, void *y_bf16, int n, int d, int h,
                                           int w, int relu, mr3d_stream_t stream_) {
  if (!x_ncdhw || !w16 || !y_bf16 || n <= 0 || d <= 0 || h <= 0 || w <= 0) return MR3D_ERR_INVALID_ARG;
  StemBfArgs a;
  a.x = x_ncdhw; a.w16 = (const bf16_t *)w16; a.shift = shift; a.y = (bf16_t *)y_bf16;
  a.N = n; a.D = d; a.H = h; a.W = w; a.Ho = (h - 1) / 2 + 1; a.Wo = (w - 1) / 2 + 1; a.relu = relu;
  using G = FwdGeo;
  const int td = (d + G::TD - 1) / G::TD, th = (a.Ho + G::TH - 1) / G::TH, tw = (a.Wo + G::TW - 1) / G::TW;
  stem_fwd_bf16_kernel<<<(unsigned)(n * td * th * tw), 256, G::LDS_BYTES, mr3d_stream(stream_)>>>(a, td, th, tw);
  return mr3d_launch_status();
}

extern "C" int mr3d_stem_conv_wgrad_bf16(const float *x_ncdhw, const void *dy_bf16, float *dw_packed, float *dbias, int n, int d, int h,
                                         int w, mr3d_stream_t stream_) {
  if (!x_ncdhw || !dy_bf16 || !dw_packed || n <= 0 || d <= 0 || h <= 0 || w <= 0) return MR3D_ERR_INVALID_ARG;
  StemBfWgArgs a;
  a.x = x_ncdhw; a.dy = (const bf16_t *)dy_bf16; a.dw = dw_packed; a.dbias = dbias;
  a.N = n; a.D = d; a.H = h; a.W = w; a.Ho = (h - 1) / 2 + 1; a.Wo = (w - 1) / 2 + 1;
  using G = WgGeo;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(stem_wgrad_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr = true;
  }
  const int td = (d + G::TD - 1) / G::TD, th = (a.Ho + G::TH - 1) / G::TH, tw = (a.Wo + G::TW - 1) / G::TW;
  const long long ntiles = (long long)n * td * th * tw;
  if (ntiles > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  const int grid = ntiles < 512 ? (int)ntiles : 512;
  stem_wgrad_bf16_kernel<<<(unsigned)grid, 512, G::LDS_BYTES, mr3d_stream(stream_)>>>(a, td, th, tw, (int)ntiles);
  return mr3d_launch_status();
}

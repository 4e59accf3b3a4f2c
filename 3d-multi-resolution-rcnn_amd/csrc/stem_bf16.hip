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
#include <atomic>
#include "mr3d_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
using bf16_t = uint16_t;
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

__device__ __forceinline__ bf16_t f2bf(float f) {          // round to nearest even, NaN stays NaN: v_cvt_pk_bf16_f32
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}

// 16 bytes at 4-byte alignment: the compiler emits dword-granular LDS reads (ds_read2_b32 pairs).  A 16-byte read whose address
// is only 2-byte aligned does execute on gfx950, but the DS unit then stays busy ~8x longer (SQ_LDS_IDX_ACTIVE: 79 % of the
// weight-gradient kernel's run time) -- every fragment address in this file is an even bf16 element.
struct __attribute__((packed, aligned(4))) U16x8 { uint16_t v[8]; };

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
// Persistent workgroups (one per CU, 8 waves) walk the 4x8x16 output tiles.  The packed weights (50 KB) are staged in LDS
// once; the input slab of the NEXT tile is fetched into registers (all loads in flight at once) while the current tile is
// multiplied, then converted to bf16 and written to LDS -- the global latency never sits between two MFMA phases.
struct FwdGeo {
  static constexpr int TD = 4, TH = 8, TW = 16, NT = 512;
  static constexpr int HD = TD + 6, HH = 2 * (TH - 1) + 7, HW = 40;          // 38 columns used (+1 read by the zero kw slot)
  static constexpr int PLANE = HD * HH * HW;                                  // bf16 elements per channel
  static constexpr int W_BYTES = 49 * 4 * 16 * 8 * 2;
  static constexpr int LDS_BYTES = W_BYTES + 3 * PLANE * 2 + 64;
  static constexpr int PAIRS = 20;                                           // 21 rows x 20 pairs = 420 of the 512 threads stage a plane
  static_assert(HH * PAIRS <= NT, "one (channel, depth) plane per staging step");
};

struct StemBfArgs {
  const float *x; const bf16_t *w16; const float *shift; bf16_t *y;
  int N, D, H, W, Ho, Wo, relu;
};


// Tile sequence of the persistent stem kernels: depth-fastest inside a (n, h, w) column; a workgroup walks a contiguous
// run, and the runs of one XCD's workgroups (blockIdx & 7 = XCD) are adjacent -- the input planes shared by depth
// neighbours (10 of 10 + 4 ...) then come from that XCD's L2 (see csrc/stem.hip).
struct BfTile { int n, td, th, tw; };
__device__ __forceinline__ BfTile bf_tile_of(int seq, int tiles_d, int tiles_h, int tiles_w) {
  BfTile t;
  t.td = seq % tiles_d; seq /= tiles_d;
  t.tw = seq % tiles_w; seq /= tiles_w;
  t.th = seq % tiles_h;
  t.n = seq / tiles_h;
  return t;
}
__device__ __forceinline__ void bf_tile_run(int bid, int nwg, int ntiles, int &begin, int &end) {
  if ((nwg & 7) == 0) {
    const int per_xcd = (ntiles + 7) >> 3, wgs = nwg >> 3;
    const int per_wg = (per_xcd + wgs - 1) / wgs;
    const int x0 = (bid & 7) * per_xcd;
    begin = x0 + (bid >> 3) * per_wg;
    end = min(min(begin + per_wg, x0 + per_xcd), ntiles);
  } else {
    const int per_wg = (ntiles + nwg - 1) / nwg;
    begin = bid * per_wg;
    end = min(begin + per_wg, ntiles);
  }
  if (begin > end) begin = end;
}

__global__ __launch_bounds__(512, 1) void stem_fwd_bf16_kernel(StemBfArgs p, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  using G = FwdGeo;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *wl = lds;
  unsigned char *slab_b = lds + G::W_BYTES;
  bf16_t *slab = reinterpret_cast<bf16_t *>(slab_b);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const size_t vox = (size_t)p.D * p.H * p.W;
  const bool even_w = (p.W & 1) == 0;

  for (int i = tid; i < G::W_BYTES / 16; i += G::NT)
    reinterpret_cast<uint4 *>(wl)[i] = reinterpret_cast<const uint4 *>(p.w16)[i];

  // staging map: iteration u = one (channel, depth) plane of the slab (compile-time), thread = (row hh, column pair pr) of it:
  // the per-thread part of every address is ONE register, the plane part is wave-uniform
  constexpr int NPL = 3 * G::HD;
  const int s_hh = tid / G::PAIRS, s_pr = tid - s_hh * G::PAIRS;
  const bool s_act = s_hh < G::HH;
  float2 st[NPL];
  auto fetch = [&](int tile) {
    const BfTile bt = bf_tile_of(tile < ntiles ? tile : 0, tiles_d, tiles_h, tiles_w);
    const int tw_i = bt.tw, th_i = bt.th, td_i = bt.td, n = bt.n;
    const int id0 = td_i * G::TD - 3, ih = 2 * th_i * G::TH - 3 + s_hh, iw = 2 * tw_i * G::TW - 3 + 2 * s_pr - 1;   // iw is even
    const bool row_ok = s_act && tile < ntiles && (unsigned)ih < (unsigned)p.H;
    const bool pair_ok = even_w && iw >= 0 && iw + 1 < p.W;
    const long long off = (long long)ih * p.W + iw;
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      const int c = u / G::HD, hd = u % G::HD;
      const int id = id0 + hd;                                        // wave-uniform
      st[u] = make_float2(0.f, 0.f);
      if (row_ok && (unsigned)id < (unsigned)p.D) {
        const float *src = p.x + ((size_t)n * 3 + c) * vox + (size_t)id * p.H * p.W + off;
        if (pair_ok) {
          st[u] = *reinterpret_cast<const float2 *>(src);
        } else {
          if ((unsigned)iw < (unsigned)p.W) st[u].x = src[0];
          if ((unsigned)(iw + 1) < (unsigned)p.W) st[u].y = src[1];
        }
      }
    }
  };
  auto commit = [&]() {
    if (!s_act) return;
    bf16_t *dst = slab + s_hh * G::HW;                                // slab columns 2*pr - 1, 2*pr
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      if (s_pr >= 1) dst[u * G::HH * G::HW + 2 * s_pr - 1] = f2bf(st[u].x);
      dst[u * G::HH * G::HW + 2 * s_pr] = f2bf(st[u].y);
    }
  };

  // this wave's four (td, th) rows of 16 outputs; lane (r, q): output column r, channel slot q (slot 3: zeros)
  int abase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mt = wave * 4 + i, td = mt >> 3, th = mt & 7;
    abase[i] = ((((q < 3 ? q : 0) * G::HD + td) * G::HH + 2 * th) * G::HW + 2 * r) * 2;
  }
  const unsigned char *wlane = wl + (q * 16 + r) * 16;
  bf16x8 zero;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.0f;
  const float sh = p.shift ? p.shift[r] : 0.f;

  int t_begin, t_end;
  bf_tile_run((int)blockIdx.x, (int)gridDim.x, ntiles, t_begin, t_end);
  fetch(t_begin < t_end ? t_begin : ntiles);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                       // the previous tile's fragments have been read
    commit();
    __syncthreads();
    fetch(tile + 1 < t_end ? tile + 1 : ntiles);      // in flight during the MFMA phase
    f32x4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kd = 0; kd < 7; ++kd) {
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) {
        const bf16x8 bw = *reinterpret_cast<const bf16x8 *>(wlane + (kd * 7 + kh) * 1024);
        const int off = ((kd * G::HH + kh) * G::HW) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16x8 a = lds_frag_unaligned(slab_b + abase[i] + off);
          if (q == 3) a = zero;
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw, acc[i], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: C/D layout of 16x16: column (cout) = lane & 15, row (output column) = 4 * (lane >> 4) + reg
    const BfTile bt = bf_tile_of(tile < ntiles ? tile : 0, tiles_d, tiles_h, tiles_w);
    const int tw_i = bt.tw, th_i = bt.th, td_i = bt.td, n = bt.n;
    const int d0 = td_i * G::TD, h0 = th_i * G::TH, w0 = tw_i * G::TW;
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
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgGeo {
  static constexpr int TD = 4, TH = 4, TW = 32, NT = 512;
  static constexpr int HD = TD + 6, HH = 2 * (TH - 1) + 7;                    // 10 x 13 rows
  // The slab holds the even and the odd input columns as two separate images of [channel][depth][row][40] bf16 (36 used: column
  // 2*ow + kw of the slab = image kw & 1, index ow + (kw >> 1)).  A lane serves ONE (channel, kd, kh) group per row tile: it reads
  // the aligned window of 12 (even image) / 10 (odd image) elements that holds the fragments of ALL the group's kw taps -- 24 + 20
  // bytes for 7 fragments instead of 7 x 16 -- and shifts the odd starts out with v_alignbit.  Strides in dwords: a row is 20, the
  // pads make (group g + 1) - (group g) == 20 (mod 32) across the kh, kd and channel wraps, so the 16-byte reads of any 8
  // consecutive lanes cover the 32 banks once.
  static constexpr int ROWD = 20, DSTR = HH * ROWD + 8, PLN = HD * DSTR + 28;
  static_assert((DSTR - 6 * ROWD) % 32 == ROWD && (PLN - 6 * DSTR - 6 * ROWD) % 32 == ROWD, "regular bank stride over the group wraps");
  static constexpr int IMG_BYTES = 3 * PLN * 4;
  static_assert(IMG_BYTES % 16 == 0 && (DSTR % 4) == 0 && (PLN % 4) == 0, "16-byte aligned windows");
  static constexpr int DYROW = 24;                                            // bf16 per dy row: 16 + 8 of padding (48 B)
  static constexpr int M = TD * TH * TW, KS = TD * TH;
  static constexpr int SLAB_BYTES = 2 * IMG_BYTES, DY_BYTES = M * DYROW * 2;
  static constexpr int LDS_BYTES = SLAB_BYTES + DY_BYTES + 64;
  static constexpr int SCATTER_BYTES = M * 16 * 4;                           // POOLED: fp32 [position][16] the windows add into
  static constexpr int WINDOWS = 3 * 3 * 17;                                  // pooled windows that reach a 4 x 4 x 32 tile
  static constexpr int GROUPS = 3 * 49;                                       // 147 (channel, kd, kh) groups -> 10 row tiles x 7 kw
  static constexpr int PAIRS = 36;                                            // columns -1 .. 70; 13 rows x 36 pairs = 468 threads per plane
  static_assert(HH * PAIRS <= NT, "one (channel, depth) plane per staging step");
};

struct StemBfWgArgs {
  const float *x; const bf16_t *dy; float *dw; float *dbias;
  int N, D, H, W, Ho, Wo;
  // POOLED (mr3d_stem_conv_wgrad_bf16_pooled): dy is the gradient of the max-pooled map [N, Dp, Hp, Wp, 16]
  const uint32_t *pool_idx; const bf16_t *pooled;
  int Dp, Hp, Wp;
  int queue_slot;                      // g_stem_bf16_queues slot of this launch
  int dbg;
};

// tile queues of the weight-gradient kernel, as in csrc/stem.hip: per slot 8 queue heads + 1 finished-workgroup count,
// all 0 between launches
__device__ unsigned g_stem_bf16_queues[64 * 16];

template <int V> struct IntC { static constexpr int value = V; };

// Work split of a k-step over the 8 waves: a unit = (row tile of 16 groups, column parity) = 4 (even kw) or 3 (odd kw) MFMAs on one
// window.  Waves 0-3: two even units (wave 0 also the tile of ones: sum_p dy); waves 4-5: three odd units; waves 6-7: one even +
// two odd -- 8..10 MFMAs per wave, 17-18 per SIMD (waves w and w + 4).
//
// POOLED: the max-pool backward (3x3x3 / stride 2 / pad 1) and the stem's ReLU mask are folded into the dy staging: the
// windows that reach the tile add relu'(pooled) * g at their arg-max into an fp32 tile in LDS, which is rounded to the bf16 dy
// rows once (the un-pooled bf16 gradient -- 168 MB on the 512 volume -- is neither written nor read).
template <bool POOLED>
__global__ __launch_bounds__(512, 1) void stem_wgrad_bf16_kernel(StemBfWgArgs p, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  using G = WgGeo;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *dyt = lds + G::SLAB_BYTES;
  float *scat = reinterpret_cast<float *>(lds + G::LDS_BYTES);       // POOLED only
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const size_t vox = (size_t)p.D * p.H * p.W;
  const bool even_w = (p.W & 1) == 0;

  // units of this wave (wave-uniform row tiles) and the lane's window offsets (bytes; k = 8q folded in)
  int egt[2] = {0, 0}, ogt[3] = {0, 0, 0};
  if (wave < 4) { egt[0] = 2 * wave; egt[1] = 2 * wave + 1; }
  else if (wave < 6) { ogt[0] = 3 * (wave - 4); ogt[1] = ogt[0] + 1; ogt[2] = ogt[0] + 2; }
  else { egt[0] = 8 + (wave - 6); ogt[0] = 6 + 2 * (wave - 6); ogt[1] = ogt[0] + 1; }
  auto window_off = [&](int gt) {                        // a lane past the last group reads the last group's window and drops the result
    const int g = min(16 * gt + r, G::GROUPS - 1);
    const int c = g / 49, kdkh = g - c * 49, kd = kdkh / 7, kh = kdkh - kd * 7;
    return (c * G::PLN + kd * G::DSTR + kh * G::ROWD + 4 * q) * 4;
  };
  const int eoff[2] = {window_off(egt[0]), window_off(egt[1])};
  const int ooff[3] = {window_off(ogt[0]) + G::IMG_BYTES, window_off(ogt[1]) + G::IMG_BYTES, window_off(ogt[2]) + G::IMG_BYTES};
  // transposing read of dy [position][16]: group q reads positions 8q..8q+7; inside the group lane 4a+b gives row a, columns 4b..4b+3
  const int i16 = lane & 15, ra = i16 >> 2, cb = i16 & 3;
  const int b_lane = (8 * q + ra) * (G::DYROW * 2) + cb * 8;

  f32x4_t acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // next tile's operands in registers: x as aligned float2 pairs, dy as two 16-byte halves of one position row.
  // Staging map as in the forward kernel: iteration u = one (channel, depth) plane, thread = (row hh, column pair pr).
  constexpr int NPL = 3 * G::HD;
  const int s_hh = tid / G::PAIRS, s_pr = tid - s_hh * G::PAIRS;
  const bool s_act = s_hh < G::HH;
  float2 st[NPL];
  uint4 sdy[2];
  auto fetch = [&](int tile) {
    const BfTile bt = bf_tile_of(tile < ntiles ? tile : 0, tiles_d, tiles_h, tiles_w);
    const int tw_i = bt.tw, th_i = bt.th, td_i = bt.td, n = bt.n;
    const int d0 = td_i * G::TD, h0 = th_i * G::TH, w0 = tw_i * G::TW;
    const int id0 = d0 - 3, ih = 2 * h0 - 3 + s_hh, iw = 2 * w0 - 3 + 2 * s_pr - 1;
    const bool tile_ok = tile < ntiles;
    const bool row_ok = s_act && tile_ok && (unsigned)ih < (unsigned)p.H;
    const bool pair_ok = even_w && iw >= 0 && iw + 1 < p.W;
    const long long off = (long long)ih * p.W + iw;
    if (even_w) {
      // even W: iw is even, a pair is inside the row or outside it as a whole.  Buffer loads: a lane outside its row gets an offset
      // past the end of the image and the hardware returns zeros -- no bounds test around a load, the 30 of them are in flight together
      // (a load inside a per-lane bounds test is a branch per load and the wave walks them one after the other)
      const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x + (size_t)n * 3 * vox), 0,
                                                                          (unsigned)(3 * vox * 4), 0x00020000);
      const unsigned voff = (row_ok && pair_ok) ? (unsigned)off * 4u : 0x80000000u;
#pragma unroll
      for (int u = 0; u < NPL; ++u) {
        const int c = u / G::HD, hd = u % G::HD;
        const int id = id0 + hd;                         // wave-uniform
        st[u] = make_float2(0.f, 0.f);
        if ((unsigned)id < (unsigned)p.D)
          st[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(xr, (int)voff, (int)(((unsigned)c * (unsigned)vox + (unsigned)id * (unsigned)(p.H * p.W)) * 4u), 0));
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPL; ++u) {
        const int c = u / G::HD, hd = u % G::HD;
        const int id = id0 + hd;
        st[u] = make_float2(0.f, 0.f);
        if (row_ok && (unsigned)id < (unsigned)p.D) {
          const float *src = p.x + ((size_t)n * 3 + c) * vox + (size_t)id * p.H * p.W + off;
          if ((unsigned)iw < (unsigned)p.W) st[u].x = src[0];
          if ((unsigned)(iw + 1) < (unsigned)p.W) st[u].y = src[1];
        }
      }
    }
    if constexpr (!POOLED) {
      const int m = tid;                                 // 512 positions, one per thread
      const int tw = m & 31, th = (m >> 5) % G::TH, td = m / (32 * G::TH);
      const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
      sdy[0] = sdy[1] = make_uint4(0u, 0u, 0u, 0u);
      if (tile_ok && od < p.D && oh < p.Ho && ow < p.Wo) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.dy + ((((size_t)n * p.D + od) * p.Ho + oh) * p.Wo + ow) * 16);
        sdy[0] = src[0]; sdy[1] = src[1];
      }
    } else {
      // (window, channel quad) items: 3 x 3 x 17 windows x 4 = 612 for 512 threads; sdy[j] = {tap ids, g (2 dwords), ReLU bits}
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int it = tid + j * G::NT, q4 = it & 3, pp = it >> 2;
        const int pw = pp % 17, ph = (pp / 17) % 3, pd = pp / 51;
        const int od = (d0 >> 1) + pd, oh = (h0 >> 1) + ph, ow = (w0 >> 1) + pw;
        sdy[j] = make_uint4(0u, 0u, 0u, 0u);
        if (tile_ok && it < G::WINDOWS * 4 && od < p.Dp && oh < p.Hp && ow < p.Wp) {
          const size_t o = ((((size_t)n * p.Dp + od) * p.Hp + oh) * p.Wp + ow) * 4 + q4;
          const uint2 g = *reinterpret_cast<const uint2 *>(p.dy + o * 4);
          uint32_t pos = 0xfu;                           // bit r: pooled value of channel r is > 0
          if (p.pooled) {
            const uint2 a = *reinterpret_cast<const uint2 *>(p.pooled + o * 4);
            auto gt0 = [](uint32_t h) { return (h & 0x7fffu) != 0u && (h & 0x8000u) == 0u && (h & 0x7fffu) <= 0x7f80u; };
            pos = (gt0(a.x & 0xffffu) ? 1u : 0u) | (gt0(a.x >> 16) ? 2u : 0u) | (gt0(a.y & 0xffffu) ? 4u : 0u) | (gt0(a.y >> 16) ? 8u : 0u);
          }
          sdy[j] = make_uint4(p.pool_idx[o], g.x, g.y, pos);
        }
      }
    }
  };
  auto commit = [&]() {
    // slab column s = iw - iw0: the pair (2*pr - 1, 2*pr) lands at odd-image index pr - 1 and even-image index pr
    if (s_act) {
      bf16_t *dst = reinterpret_cast<bf16_t *>(lds) + s_hh * (2 * G::ROWD);
#pragma unroll
      for (int u = 0; u < NPL; ++u) {
        bf16_t *d = dst + ((u / G::HD) * G::PLN + (u % G::HD) * G::DSTR) * 2;
        d[s_pr] = f2bf(st[u].y);
        if (s_pr >= 1) d[G::IMG_BYTES / 2 + s_pr - 1] = f2bf(st[u].x);
      }
    }
    if constexpr (!POOLED) {
      *reinterpret_cast<uint4 *>(dyt + tid * (G::DYROW * 2)) = sdy[0];
      *reinterpret_cast<uint4 *>(dyt + tid * (G::DYROW * 2) + 16) = sdy[1];
    }
  };
  // POOLED: every window adds its gradient at its arg-max position of the tile
  auto scatter = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int it = tid + j * G::NT, q4 = it & 3, pp = it >> 2;
      const int pw = pp % 17, ph = (pp / 17) % 3, pd = pp / 51;
      const uint32_t gq[2] = {sdy[j].y, sdy[j].z};
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const uint32_t gb = (rr & 1) ? (gq[rr >> 1] & 0xffff0000u) : (gq[rr >> 1] << 16);
        const int tap = (sdy[j].x >> (8 * rr)) & 255;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        // window (pd, ph, pw) of the tile's window block covers tile positions 2 p - 1 + k
        const int td = 2 * pd - 1 + kd, th = 2 * ph - 1 + kh, tw = 2 * pw - 1 + kw;
        if (((sdy[j].w >> rr) & 1u) && (gb & 0x7fffffffu) != 0u && (unsigned)td < (unsigned)G::TD && (unsigned)th < (unsigned)G::TH &&
            (unsigned)tw < (unsigned)G::TW)
          atomicAdd(scat + ((td * G::TH + th) * G::TW + tw) * 16 + q4 * 4 + rr, __uint_as_float(gb));
      }
    }
  };
  // after the barrier behind scatter(): a thread rounds its position's 16 sums into its bf16 dy row and clears them
  auto round_rows = [&]() {
    float4 *src = reinterpret_cast<float4 *>(scat + tid * 16);
    auto pk = [](const float4 &a, const float4 &b) {
      return make_uint4((uint32_t)f2bf(a.x) | ((uint32_t)f2bf(a.y) << 16), (uint32_t)f2bf(a.z) | ((uint32_t)f2bf(a.w) << 16),
                        (uint32_t)f2bf(b.x) | ((uint32_t)f2bf(b.y) << 16), (uint32_t)f2bf(b.z) | ((uint32_t)f2bf(b.w) << 16));
    };
    const float4 a = src[0], b = src[1], c = src[2], d = src[3];
    src[0] = src[1] = src[2] = src[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<uint4 *>(dyt + tid * (G::DYROW * 2)) = pk(a, b);
    *reinterpret_cast<uint4 *>(dyt + tid * (G::DYROW * 2) + 16) = pk(c, d);
  };

  // ---- the 16 k-steps of a tile for a wave with NE even and NO odd units (+ the tile of ones)
  auto mma_phase = [&](auto ne_, auto no_, auto ones_) {
    constexpr int NE = decltype(ne_)::value, NO = decltype(no_)::value;
    constexpr bool ONES = decltype(ones_)::value != 0;
    struct Ops { u32x4 ea[NE ? NE : 1]; u32x2 eb[NE ? NE : 1]; u32x4 oa[NO ? NO : 1]; uint32_t ob[NO ? NO : 1]; bf16x8 b; };
    auto load_step = [&](int ks, Ops &o) {
      const int th = ks % G::TH, td = ks / G::TH;
      unsigned char *yb = dyt + ks * 32 * (G::DYROW * 2) + b_lane;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(yb));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(yb + 4 * (G::DYROW * 2)));
      o.b = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      const int rowoff = (td * G::DSTR + 2 * th * G::ROWD) * 4;
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        o.ea[u] = *reinterpret_cast<const u32x4 *>(lds + eoff[u] + rowoff);
        o.eb[u] = *reinterpret_cast<const u32x2 *>(lds + eoff[u] + rowoff + 16);
      }
#pragma unroll
      for (int u = 0; u < NO; ++u) {
        o.oa[u] = *reinterpret_cast<const u32x4 *>(lds + ooff[u] + rowoff);
        o.ob[u] = *reinterpret_cast<const uint32_t *>(lds + ooff[u] + rowoff + 16);
      }
    };
    auto mfma = [&](f32x4_t &c, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, const bf16x8 &b) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, u32x4{a0, a1, a2, a3}), b, c, 0, 0, 0);
    };
    auto mma_step = [&](const Ops &o) {
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const uint32_t d0 = o.ea[u][0], d1 = o.ea[u][1], d2 = o.ea[u][2], d3 = o.ea[u][3], d4 = o.eb[u][0], d5 = o.eb[u][1];
        const uint32_t s0 = __builtin_amdgcn_alignbit(d1, d0, 16), s1 = __builtin_amdgcn_alignbit(d2, d1, 16),
                       s2 = __builtin_amdgcn_alignbit(d3, d2, 16), s3 = __builtin_amdgcn_alignbit(d4, d3, 16),
                       s4 = __builtin_amdgcn_alignbit(d5, d4, 16);
        mfma(acc[4 * u + 0], d0, d1, d2, d3, o.b);          // kw 0: elements 0..7
        mfma(acc[4 * u + 1], s0, s1, s2, s3, o.b);          // kw 2: 1..8
        mfma(acc[4 * u + 2], d1, d2, d3, d4, o.b);          // kw 4: 2..9
        mfma(acc[4 * u + 3], s1, s2, s3, s4, o.b);          // kw 6: 3..10
      }
#pragma unroll
      for (int u = 0; u < NO; ++u) {
        const uint32_t d0 = o.oa[u][0], d1 = o.oa[u][1], d2 = o.oa[u][2], d3 = o.oa[u][3], d4 = o.ob[u];
        const uint32_t s0 = __builtin_amdgcn_alignbit(d1, d0, 16), s1 = __builtin_amdgcn_alignbit(d2, d1, 16),
                       s2 = __builtin_amdgcn_alignbit(d3, d2, 16), s3 = __builtin_amdgcn_alignbit(d4, d3, 16);
        mfma(acc[4 * NE + 3 * u + 0], d0, d1, d2, d3, o.b);   // kw 1
        mfma(acc[4 * NE + 3 * u + 1], s0, s1, s2, s3, o.b);   // kw 3
        mfma(acc[4 * NE + 3 * u + 2], d1, d2, d3, d4, o.b);   // kw 5
      }
      if constexpr (ONES) mfma(acc[4 * NE + 3 * NO], 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, o.b);
    };
    // the windows of k-step ks + 1 are requested before the MFMAs of k-step ks
    Ops o0, o1;
    load_step(0, o0);
#pragma unroll 1
    for (int ks = 0; ks < G::KS; ks += 2) {
      load_step(ks + 1, o1);
      mma_step(o0);
      if (ks + 2 < G::KS) load_step(ks + 2, o0);
      mma_step(o1);
    }
  };
  // ---- write-out of one accumulator: column (cout) = lane & 15, group = 16*gt + 4*(lane>>4) + reg
  auto write_acc = [&](const f32x4_t &a, int gt, int kw) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = 16 * gt + 4 * q + e;
      const int c = g / 49, kdkh = g - c * 49;
      if (g < G::GROUPS) atomicAdd(p.dw + ((size_t)(kdkh * 7 + kw) * 4 + c) * 16 + r, a[e]);
    }
  };

  // Tiles are pulled from eight queues (one contiguous range of the depth-fastest order per XCD, the workgroup's own XCD first):
  // with a fixed run per workgroup the launch ended with its latest workgroup, and in the training step the other layers'
  // last weight gradients and their finishing run beside this kernel (see stem_wgrad_kernel in csrc/stem.hip).
  __shared__ int s_tile;
  unsigned *queue = g_stem_bf16_queues + 16 * p.queue_slot;
  const int per_xcd = (ntiles + 7) >> 3, home = blockIdx.x & 7;
  auto range = [&](int x, int &lo) { lo = x * per_xcd; return max(min(lo + per_xcd, ntiles) - lo, 0); };
  auto take = [&](unsigned ticket) {                     // thread 0
    int lo, cnt = range(home, lo);
    if (ticket < (unsigned)cnt) return lo + (int)ticket;
    for (int a = 1; a < 8; ++a) {
      cnt = range((home + a) & 7, lo);
      if (cnt > 0) {
        const unsigned v = atomicAdd(queue + ((home + a) & 7), 1u);
        if (v < (unsigned)cnt) return lo + (int)v;
      }
    }
    return -1;
  };
  if constexpr (POOLED) {
    for (int e = tid; e < G::M * 4; e += G::NT) reinterpret_cast<float4 *>(scat)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  unsigned ticket = 0;
  if (tid == 0) s_tile = take(atomicAdd(queue + home, 1u));
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane(s_tile);      // wave-uniform: the tile arithmetic below stays on the scalar unit
  fetch(tile >= 0 ? tile : ntiles);
  if (tid == 0) ticket = atomicAdd(queue + home, 1u);
  while (tile >= 0) {
    __syncthreads();
    if (!(p.dbg & 2)) commit();
    if constexpr (POOLED) { if (!(p.dbg & 8)) scatter(); }
    if (tid == 0) s_tile = take(ticket);                 // issued a tile ago
    __syncthreads();
    if constexpr (POOLED) {
      round_rows();
      __syncthreads();
    }
    const int next = __builtin_amdgcn_readfirstlane(s_tile);
    if (tid == 0 && next >= 0) ticket = atomicAdd(queue + home, 1u);
    if (!(p.dbg & 4)) fetch(next >= 0 ? next : ntiles);                    // in flight during the MFMA phase
    if (p.dbg & 1) {} else
    if (wave == 0) mma_phase(IntC<2>{}, IntC<0>{}, IntC<1>{});
    else if (wave < 4) mma_phase(IntC<2>{}, IntC<0>{}, IntC<0>{});
    else if (wave < 6) mma_phase(IntC<0>{}, IntC<3>{}, IntC<0>{});
    else mma_phase(IntC<1>{}, IntC<2>{}, IntC<0>{});
    tile = next;
  }
  if (tid == 0) {                                        // the last workgroup to run dry re-arms the queues
    const unsigned done = atomicAdd(queue + 8, 1u);
    if (done == gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 9; ++x) atomicExch(queue + x, 0u);
    }
  }
  if (wave < 4) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) write_acc(acc[4 * u + k], egt[u], 2 * k);
    if (wave == 0 && p.dbias && q == 0) atomicAdd(p.dbias + r, acc[8][0]);
  } else if (wave < 6) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) write_acc(acc[3 * u + k], ogt[u], 2 * k + 1);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) write_acc(acc[k], egt[0], 2 * k);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) write_acc(acc[4 + 3 * u + k], ogt[u], 2 * k + 1);
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
  static bool attr = false;
  static int cus = 0;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(stem_fwd_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    attr = true;
  }
  const int td = (d + G::TD - 1) / G::TD, th = (a.Ho + G::TH - 1) / G::TH, tw = (a.Wo + G::TW - 1) / G::TW;
  const long long ntiles = (long long)n * td * th * tw;
  if (ntiles > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  int grid = ntiles < cus ? (int)ntiles : cus;               // one persistent 8-wave workgroup per CU (100 KB of LDS)
  if (grid >= 8) grid &= ~7;                                 // XCD-contiguous tile runs (bf_tile_run)
  stem_fwd_bf16_kernel<<<(unsigned)grid, G::NT, G::LDS_BYTES, mr3d_stream(stream_)>>>(a, td, th, tw, (int)ntiles);
  return mr3d_launch_status();
}

static int stem_wgrad_bf16_impl(const float *x_ncdhw, const void *dy_bf16, const uint32_t *pool_idx, const void *pooled, float *dw_packed,
                                float *dbias, int n, int d, int h, int w, mr3d_stream_t stream_) {
  if (!x_ncdhw || !dy_bf16 || !dw_packed || n <= 0 || d <= 0 || h <= 0 || w <= 0) return MR3D_ERR_INVALID_ARG;
  StemBfWgArgs a;
  a.x = x_ncdhw; a.dy = (const bf16_t *)dy_bf16; a.dw = dw_packed; a.dbias = dbias;
  a.N = n; a.D = d; a.H = h; a.W = w; a.Ho = (h - 1) / 2 + 1; a.Wo = (w - 1) / 2 + 1;
  a.pool_idx = pool_idx; a.pooled = (const bf16_t *)pooled;
  a.Dp = (d - 1) / 2 + 1; a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;
  using G = WgGeo;
  static bool attr = false;
  static int cus = 0;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(stem_wgrad_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(stem_wgrad_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS_BYTES + G::SCATTER_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    attr = true;
  }
  const int td = (d + G::TD - 1) / G::TD, th = (a.Ho + G::TH - 1) / G::TH, tw = (a.Wo + G::TW - 1) / G::TW;
  const long long ntiles = (long long)n * td * th * tw;
  if (ntiles > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  int grid = ntiles < cus ? (int)ntiles : cus;               // one persistent workgroup per CU (140 / 156 KB of LDS)
  if (grid >= 8) grid &= ~7;
  static std::atomic<unsigned> next_slot{0};                 // ring of queue slots: see stem_wgrad_impl in csrc/stem.hip
  a.queue_slot = (int)(next_slot.fetch_add(1) & 63u);
  a.dbg = g_mr3d_tuning[0];
  if (pool_idx)
    stem_wgrad_bf16_kernel<true><<<(unsigned)grid, 512, G::LDS_BYTES + G::SCATTER_BYTES, mr3d_stream(stream_)>>>(a, td, th, tw, (int)ntiles);
  else
    stem_wgrad_bf16_kernel<false><<<(unsigned)grid, 512, G::LDS_BYTES, mr3d_stream(stream_)>>>(a, td, th, tw, (int)ntiles);
  return mr3d_launch_status();
}

extern "C" int mr3d_stem_conv_wgrad_bf16(const float *x_ncdhw, const void *dy_bf16, float *dw_packed, float *dbias, int n, int d, int h,
                                         int w, mr3d_stream_t stream_) {
  return stem_wgrad_bf16_impl(x_ncdhw, dy_bf16, nullptr, nullptr, dw_packed, dbias, n, d, h, w, stream_);
}

extern "C" int mr3d_stem_conv_wgrad_bf16_pooled(const float *x_ncdhw, const void *g_pooled_bf16, const uint32_t *argmax,
                                                const void *pooled_bf16, float *dw_packed, float *dbias, int n, int d, int h, int w,
                                                mr3d_stream_t stream_) {
  if (!argmax) return MR3D_ERR_INVALID_ARG;
  return stem_wgrad_bf16_impl(x_ncdhw, g_pooled_bf16, argmax, pooled_bf16, dw_packed, dbias, n, d, h, w, stream_);
}

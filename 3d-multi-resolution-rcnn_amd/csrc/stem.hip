// The ResNet3D stem: Conv3d(3, 16, kernel 7, stride (1,2,2), padding 3) (+ folded BN + ReLU), forward and
// weight gradient (mmdet/models/backbones/resnet3d.py:397-409,467-469).  40 % of the backbone's forward
// FLOPs sit in this one layer, and it is the worst shape for a generic implicit GEMM (Cin = 3, Cout = 16),
// so it gets its own kernels that read the volume in the reference's own planar NCDHW layout.
//
// GEMM mapping (v_mfma_f32_16x16x4_f32):
//   forward  D[i = 16 consecutive output x][j = 16 Cout] += A[i][k] * B[k][j],  K = (ci, kd, kh, kw)
//     kw is padded 7 -> 8 and split as kw = 2*k + s - 1 (k = MFMA k index 0..3, s = step 0/1; kw = -1 has
//     weight 0), so the A operands of BOTH steps of a (ci,kd,kh) row are one aligned ds_read_b64 per lane:
//        A_s[i][k] = slab[plane d+kd][row 2h+kh][2*x_i + 2*k + s]      (slab column 0 = input column 2*w0 - 4)
//     consecutive lanes read consecutive 8-byte words: conflict-free without padding tricks.
//   wgrad    D[i = Cout][j = tap] += A[i][k = 4 output x] * B[k][j]   with the 343 taps flattened into 24 groups
//        of 16 columns: A = dy[pos][co] (LDS tile), B = slab[halo origin of position k + per-lane tap offset];
//     the 3 x 24 (ci, tap group) accumulator tiles are split over the 8 waves (3 groups each) and live in
//     registers across a persistent loop over tiles; one atomic pass at the end.
// One input channel of the halo block ((TD+6) x (2*TH+5) x 72 floats) is resident in LDS at a time.
#include <atomic>
#include "conv3d.h"
#include <stdlib.h>

namespace {

constexpr int kStemThreads = 512;      // 8 waves: 2 per SIMD

// Tile order of the stem kernels.  The 7-tap depth halo makes a 4-plane tile read 10 input planes: dealt out in
// blockIdx order (w fastest, round-robin over the 8 XCDs) a tile's depth neighbours ran 64 slots apart on their XCD
// and the shared planes had left its 4 MB L2 by then -- 2.0 GB of HBM reads per launch for 0.59 GB of input.  Here
// tiles are ordered depth-fastest inside a (n, h, w) column, and each XCD (= blockIdx & 7) owns a contiguous range of
// columns: the tiles in flight on one XCD are whole columns, whose planes are fetched once.
struct StemTile { int n, td, th, tw; bool valid; };

__device__ __forceinline__ StemTile stem_tile_of(int seq, int tiles_d, int tiles_h, int tiles_w) {      // seq: depth-fastest order
  StemTile t;
  t.valid = true;
  t.td = seq % tiles_d; seq /= tiles_d;
  t.tw = seq % tiles_w; seq /= tiles_w;
  t.th = seq % tiles_h;
  t.n = seq / tiles_h;
  return t;
}

// one-tile-per-workgroup kernels: grid = 8 * cols_per_xcd * tiles_d
__device__ __forceinline__ StemTile stem_tile_xcd(int bid, int ncols, int tiles_d, int tiles_h, int tiles_w) {
  const int cols_per = (ncols + 7) >> 3;
  const int xcd = bid & 7, s = bid >> 3;
  const int col = xcd * cols_per + s / tiles_d;
  StemTile t;
  if (s >= cols_per * tiles_d || col >= ncols) { t.valid = false; t.n = t.td = t.th = t.tw = 0; return t; }
  t = stem_tile_of(col * tiles_d + s % tiles_d, tiles_d, tiles_h, tiles_w);
  return t;
}

// ------------------------------------------------------------------------------------------------ forward
struct StemFwdArgs {
  const float *x;      // [N,3,D,H,W] planar
  const float *wf;     // packed [tap = (kd*7+kh)*7+kw][4][16] (generic packing, BN scale folded, ci 3 = 0)
  const float *shift;  // [16]
  float *y;            // [N, D, Ho, Wo, 16]
  int N, D, H, W, Ho, Wo;
  int relu;
  int y_bf16;          // configs[4]: y is a bf16 buffer (round to nearest even); the arithmetic stays fp32
};

__device__ __forceinline__ uint16_t stem_f2bf(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);          // round to nearest even, NaN stays NaN
}
__device__ __forceinline__ void stem_store(const StemFwdArgs &p, size_t idx, float v) {
  if (p.y_bf16) reinterpret_cast<uint16_t *>(p.y)[idx] = stem_f2bf(v);
  else p.y[idx] = v;
}

// ROW = LDS row stride in floats of the FORWARD slab (multiple of 8 keeps the ds_read_b64 words aligned); the weight
// gradient has its own geometry (StemWgGeo)
template <int TD, int TH, int TW, int ROW>
struct StemGeo {
  static constexpr int PD = TD + 6, PH = 2 * (TH - 1) + 7, PW = 2 * (TW - 1) + 7 + 1;   // +1: slab starts one column early
  static constexpr int SW_ROW = ROW;
  static constexpr int SLAB = PD * PH * ROW;             // floats of one input channel
  static_assert(PW <= ROW, "row stride");
};

// Slab column c holds input column 2*w0 - 4 + c (one column left of the halo origin 2*w0 - 3): the slab then
// starts on a 16-byte boundary of the volume row, so a row is staged with aligned float4 loads (W % 4 == 0)
// instead of 37 scalar ones -- staging VALU work was ~20 % of the MFMA time with scalar loads.
template <int NTHREADS, class G>
__device__ __forceinline__ void stem_stage_channel(float *slab, const float *__restrict__ xc, int D, int H, int W, int d0,
                                                   int h0, int w0, int tid) {
  constexpr int SW_ROW = G::SW_ROW;
  static_assert(SW_ROW % 4 == 0, "row stride in whole float4");
  if ((W & 3) == 0) {
    // A thread owns one float4 column group (r, q) of EVERY plane: its row / column tests and its 32-bit offset inside a plane are
    // formed once, a plane's pointer and its depth test are scalars, the LDS address is the thread's base plus an immediate -- no
    // vector ALU work per element.  (One item per loop trip with div / mod chains, three bounds tests and a 64-bit multiply
    // chain each was 0.9 vector instructions per MFMA of this kernel, and those do not hide under fp32 MFMAs.)
    constexpr int Q = SW_ROW / 4;
    static_assert(G::PH * Q <= NTHREADS, "one plane per pass");
    const int r = tid / Q, q = tid - r * Q;
    const int ih = 2 * h0 - 3 + r, iw = 2 * w0 - 4 + 4 * q;
    const bool mine = tid < G::PH * Q && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    const unsigned voff = mine ? (unsigned)(ih * W + iw) : 0u;
    const int plane = H * W;                             // a channel volume is < 2^28 elements (checked by the launcher)
    const __amdgpu_buffer_rsrc_t xr = buffer_of(xc);
    float4 v[G::PD];
#pragma unroll
    for (int pl = 0; pl < G::PD; ++pl) {
      const int id = d0 - 3 + pl;                                   // (scalar)
      v[pl] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)id < (unsigned)D && mine) v[pl] = buffer_load4(xr, voff * 4u, (unsigned)(id * plane) * 4u);       // scalar plane offset
    }
    if (tid < G::PH * Q) {
      float *dst = slab + r * SW_ROW + 4 * q;
#pragma unroll
      for (int pl = 0; pl < G::PD; ++pl) *reinterpret_cast<float4 *>(dst + pl * G::PH * SW_ROW) = v[pl];
    }
  } else {
    for (int e = tid; e < G::PD * G::PH * SW_ROW; e += NTHREADS) {
      const int c = e % SW_ROW, r = (e / SW_ROW) % G::PH, pl = e / (SW_ROW * G::PH);
      const int id = d0 - 3 + pl, ih = 2 * h0 - 3 + r, iw = 2 * w0 - 4 + c;
      float v = 0.f;
      if ((unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
        v = xc[((size_t)id * H + ih) * W + iw];
      slab[e] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward, Winograd along D
// The stem has stride 1 along depth, so its seven depth taps split as 3 + 3 + 1 and each 3-tap group takes the
// F(2,3) form over a PAIR of output depths (d, d+1):  with slab planes P_j = plane(d + j),
//   group kd 0..2: (P0-P2, P1+P2, P2-P1, P1-P3) x G(g0,g1,g2),   group kd 3..5: (P3-P5, P4+P5, P5-P4, P4-P6) x G(g3,g4,g5),
//   tap kd 6: P6*g6 belongs to y[d] only and P7*g6 to y[d+1] only, i.e. to M0 and (negated) M3 of
//   y[d] = M0+M1+M2, y[d+1] = M1-M2-M3 -- no extra accumulators.
// 10 MFMA groups per (kh, kw step) and depth pair instead of 14.  A wave owns a depth pair and four of the tile's
// eight output rows.  Operands: the planar input slab of one channel at a time in LDS (stem_stage_channel), the kw taps split
// as kw = 2k + s - 1 so that a stride-2 output row reads two contiguous LDS words (ds_read_b64), B fragments from L2.
// (The direct form this replaced -- one wave per output depth, 14 MFMA groups -- ran at 1.79 ms on the 512 volume.)
template <int TW>
__global__ __launch_bounds__(256, 3) void stem_fwd_wino_kernel(StemFwdArgs p, int tiles_d, int tiles_h, int tiles_w) {
  constexpr int TD = 4, TH = 8, THW = 4;                       // THW: output rows per wave
  static_assert(TW == 16, "one 16-column MFMA tile per row");
  using G = StemGeo<TD, TH, TW, 40>;
  constexpr int SW_ROW = G::SW_ROW;
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dp = wave >> 1, hh = wave & 1;                      // depth pair, row half
  const int li = lane & 15, lk = lane >> 4;
  const StemTile st = stem_tile_xcd(blockIdx.x, p.N * tiles_h * tiles_w, tiles_d, tiles_h, tiles_w);
  if (!st.valid) return;
  const int n = st.n;
  const int d0 = st.td * TD, h0 = st.th * TH, w0 = st.tw * TW;

  f32x4 acc[4][THW];                                            // [Winograd product f][row]
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int t = 0; t < THW; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane_off = 2 * li + 2 * lk;
  const size_t vox = (size_t)p.D * p.H * p.W;
  // weight taps of this lane: kw = 2 lk - 1 (none for lk = 0: the product is masked) and kw = 2 lk
  const unsigned goff0 = (unsigned)(((lk > 0 ? 2 * lk - 1 : 0) * 64 + li) * 4), goff1 = (unsigned)((2 * lk * 64 + li) * 4);
  const float gmask0 = lk > 0 ? 1.f : 0.f;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wf), 0, 343 * 4 * 16 * 4, 0x00020000);

  for (int ci = 0; ci < 3; ++ci) {
    __syncthreads();
    stem_stage_channel<256, G>(slab, p.x + ((size_t)n * 3 + ci) * vox, p.D, p.H, p.W, d0, h0, w0, tid);
    __syncthreads();
    // raw weights of one kh row: g[kd][step]; step s of MFMA k index lk is tap kw = 2*lk + s - 1 (kw = -1: weight 0)
    // buffer loads: the (kd, kh, ci) row is a SCALAR byte offset, the lane's two taps are constant vector offsets -- no vector
    // address arithmetic per load (flat loads rebuilt a 64-bit pointer per kd: ~4 vector instructions per tap pair)
    auto load_g = [&](int kh, float (&g)[7][2]) {
#pragma unroll
      for (int kd = 0; kd < 7; ++kd) {
        const int srow = (((kd * 7 + kh) * 7) * 64 + ci * 16) * 4;
        g[kd][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, goff0, srow, 0)) * gmask0;
        g[kd][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, goff1, srow, 0));
      }
    };
    float g0[7][2], g1[7][2];
    load_g(0, g0);
    auto row = [&](int kh, const float (&g)[7][2]) {
      // B values: [group][f][step]
      float u1[4][2], u2[4][2], u6[2], u6n[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const float a0 = g[0][s_], a1 = g[1][s_], a2 = g[2][s_], b0 = g[3][s_], b1 = g[4][s_], b2 = g[5][s_];
        u1[0][s_] = a0; u1[1][s_] = 0.5f * ((a0 + a2) + a1); u1[2][s_] = 0.5f * ((a0 + a2) - a1); u1[3][s_] = a2;
        u2[0][s_] = b0; u2[1][s_] = 0.5f * ((b0 + b2) + b1); u2[2][s_] = 0.5f * ((b0 + b2) - b1); u2[3][s_] = b2;
        u6[s_] = g[6][s_]; u6n[s_] = -g[6][s_];
      }
#pragma unroll
      for (int t = 0; t < THW; ++t) {
        const float *base = slab + ((2 * dp) * G::PH + kh + 2 * (hh * THW + t)) * SW_ROW + lane_off;
        float2 P[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) P[j] = *reinterpret_cast<const float2 *>(base + j * G::PH * SW_ROW);
        // The ten transformed operand pairs of the row are formed FIRST, in one batch, and the 20 MFMAs follow behind a
        // scheduling barrier: issued pairwise (v_sub -> v_mfma) every MFMA waited two `s_nop` wait states for its freshly written
        // operand register (62 s_nop per 80 MFMAs), and vector ALU instructions do not hide under fp32 MFMAs on this chip anyway.
        auto sub = [](float2 a, float2 b_) { return make_float2(a.x - b_.x, a.y - b_.y); };
        auto add = [](float2 a, float2 b_) { return make_float2(a.x + b_.x, a.y + b_.y); };
        float2 T[10];
        T[0] = sub(P[0], P[2]); T[1] = add(P[1], P[2]); T[2] = sub(P[2], P[1]); T[3] = sub(P[1], P[3]);
        T[4] = sub(P[3], P[5]); T[5] = add(P[4], P[5]); T[6] = sub(P[5], P[4]); T[7] = sub(P[4], P[6]);
        T[8] = P[6]; T[9] = P[7];
        __builtin_amdgcn_sched_barrier(0);
        auto mm = [&](f32x4 &c, float2 a, const float (&bb)[2]) {
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb[1], c, 0, 0, 0);
        };
        mm(acc[0][t], T[0], u1[0]);
        mm(acc[1][t], T[1], u1[1]);
        mm(acc[2][t], T[2], u1[2]);
        mm(acc[3][t], T[3], u1[3]);
        mm(acc[0][t], T[4], u2[0]);
        mm(acc[1][t], T[5], u2[1]);
        mm(acc[2][t], T[6], u2[2]);
        mm(acc[3][t], T[7], u2[3]);
        mm(acc[0][t], T[8], u6);
        mm(acc[3][t], T[9], u6n);                      // m3 -= P7 * g6: the sign sits on the weight
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // kh rows two at a time, the next row's weights requested before the current row's MFMAs
    for (int kh = 0; kh < 6; kh += 2) {
      load_g(kh + 1, g1);
      row(kh, g0);
      load_g(kh + 2, g0);
      row(kh + 1, g1);
    }
    row(6, g0);
  }

  // Stores through a buffer descriptor: the output row (n, od, oh) is a SCALAR byte offset (the depth pair and row half of a wave
  // are wave-uniform), a lane's four columns are constant offsets -- no 64-bit index arithmetic per element (32 stores per lane).
  const float sh = p.shift ? p.shift[li] : 0.f;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0x7fffffff, 0x00020000);
  const int esz = p.y_bf16 ? 2 : 4;                                  // (the launcher bounds the output below 2^31 bytes)
  const int lane_b = ((w0 + lk * 4) * 16 + li) * esz;
#pragma unroll
  for (int t = 0; t < THW; ++t) {
    const int oh = h0 + hh * THW + t;
    if (oh >= p.Ho) continue;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int od = d0 + 2 * dp + e;
      if (od >= p.D) continue;
      const int row_b = (((n * p.D + od) * p.Ho + oh) * p.Wo) * 16 * esz;          // (scalar)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ow = w0 + lk * 4 + r;
        if (ow < p.Wo) {
          const float m0 = acc[0][t][r], m1 = acc[1][t][r], m2 = acc[2][t][r], m3 = acc[3][t][r];
          float v = (e == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3) + sh;
          if (p.relu) v = fmaxf(v, 0.f);
          if (p.y_bf16) __builtin_amdgcn_raw_buffer_store_b16(stem_f2bf(v), yr, lane_b + r * 16 * esz, row_b, 0);
          else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, lane_b + r * 16 * esz, row_b, 0);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ wgrad
struct StemWgradArgs {
  const float *x;      // [N,3,D,H,W] planar
  const float *dy;     // [N, D, Ho, Wo, 16]
  float *dw;           // packed [343][4][16], pre-zeroed (ci 3 stays 0)
  float *dbias;        // [16] or null
  int N, D, H, W, Ho, Wo;
  float *part;         // per-workgroup partial blocks [343][3][16] (two-stage reduction) or null (atomics)
  int dy_bf16;         // configs[4]: dy is a bf16 buffer
  // max-pool backward folded into the dy staging (mr3d_stem_conv_wgrad_pooled): dy is then the gradient of the POOLED map
  // [N, Dp, Hp, Wp, 16], scattered to its arg-max positions inside the tile
  const uint32_t *pool_idx;   // [N, Dp, Hp, Wp, 4] tap ids of the 3x3x3 / stride 2 / pad 1 pool, or null
  const float *pooled;        // pooled activation (its sign is the ReLU mask) or null
  int Dp, Hp, Wp;
  int queue_slot;             // which g_stem_wgrad_queues slot this launch pulls its tiles from
};

// LDS geometry of the weight-gradient slab.  Row stride 39 (== 7 mod 32) and plane stride 849 (== 17 == 49 - 32 mod 32)
// make the bank of tap t's halo offset equal to (t + const) mod 32: the 16 consecutive taps of a column group hit 16
// consecutive banks.  The four GEMM-k positions of a step are output columns (x, x+8, x+1, x+9): lanes of one 32-lane
// half differ by 8 columns = 16 floats = 16 banks, so every B read is conflict-free (with the 40-float rows and four
// consecutive columns 12 of 16 lanes collided pairwise and each read took two LDS passes).
struct StemWgGeo {
  static constexpr int TD = 4, TH = 8, TW = 16;
  static constexpr int PD = TD + 6, PH = 2 * (TH - 1) + 7, PW = 2 * (TW - 1) + 7 + 1;     // 10 x 21 x 38 (one spare column on the left)
  static constexpr int ROW = 39, PLANE = PH * ROW + 30, SLAB = PD * PLANE;
  static_assert(PW <= ROW && ROW % 32 == 7 && PLANE % 32 == 17, "bank layout");
};

__device__ __forceinline__ void stem_wgrad_stage_channel(float *slab, const float *__restrict__ xc, int D, int H, int W, int d0,
                                                         int h0, int w0, int tid) {
  using G = StemWgGeo;
  constexpr int Q = 10;                                  // float4 per row: slab columns 0..39, column 39 is not stored
  if ((W & 3) == 0) {
    // as in stem_stage_channel: a thread owns the column group (r, q) of every second plane (two planes per pass of the 512
    // threads); per element there is one load with a scalar plane pointer and four LDS stores at immediate offsets
    constexpr int PER = G::PH * Q;                       // 210 column groups per plane
    static_assert(2 * PER <= kStemThreads && G::PD % 2 == 0, "two planes per pass");
    const int half = tid >= PER ? 1 : 0, t = tid - half * PER;
    const int r = t / Q, q = t - r * Q;
    const int ih = 2 * h0 - 3 + r, iw = 2 * w0 - 4 + 4 * q;
    const bool lane_ok = tid < 2 * PER;
    const bool mine = lane_ok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    const unsigned voff = mine ? (unsigned)(ih * W + iw) : 0u;
    const int plane = H * W;
    const __amdgpu_buffer_rsrc_t xr = buffer_of(xc);
    float4 v[G::PD / 2];
#pragma unroll
    for (int k = 0; k < G::PD / 2; ++k) {
      const int id = d0 - 3 + 2 * k + half;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)id < (unsigned)D && mine) v[k] = buffer_load4(xr, (voff + (unsigned)(id * plane)) * 4u, 0u);
    }
    if (lane_ok) {
      float *d = slab + half * G::PLANE + r * G::ROW + 4 * q;
#pragma unroll
      for (int k = 0; k < G::PD / 2; ++k) {
        float *dd = d + 2 * k * G::PLANE;
        dd[0] = v[k].x; dd[1] = v[k].y; dd[2] = v[k].z;
        if (q < Q - 1) dd[3] = v[k].w;
      }
    }
    return;
  }
  for (int e = tid; e < G::PD * G::PH * Q; e += kStemThreads) {
    const int q = e % Q, r = (e / Q) % G::PH, pl = e / (Q * G::PH);
    const int id = d0 - 3 + pl, ih = 2 * h0 - 3 + r, iw = 2 * w0 - 4 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H) {
      const float *src = xc + ((size_t)id * H + ih) * W;
      if ((unsigned)iw < (unsigned)W) v.x = src[iw];
      if ((unsigned)(iw + 1) < (unsigned)W) v.y = src[iw + 1];
      if ((unsigned)(iw + 2) < (unsigned)W) v.z = src[iw + 2];
      if ((unsigned)(iw + 3) < (unsigned)W) v.w = src[iw + 3];
    }
    float *d = slab + pl * G::PLANE + r * G::ROW + 4 * q;
    d[0] = v.x; d[1] = v.y; d[2] = v.z;
    if (q < Q - 1) d[3] = v.w;
  }
}

// per slot: 8 tile-queue heads + 1 finished-workgroup count, all 0 between launches
__device__ unsigned g_stem_wgrad_queues[64 * 16];

__global__ __launch_bounds__(kStemThreads) void stem_wgrad_kernel(StemWgradArgs p, int tiles_d, int tiles_h, int tiles_w) {
  using G = StemWgGeo;
  constexpr int TD = G::TD, TH = G::TH, TW = G::TW;
  constexpr int M = TD * TH * TW;          // 512 positions per tile
  constexpr int DYROW = 16;                // lanes (co, k) read word 16k + co: conflict-free
  constexpr int NQ = 3;                    // accumulator groups per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *dyt = lds;                        // [M][16], rows in GEMM-k order (see ld)
  float *slab = lds + M * DYROW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const size_t vox = (size_t)p.D * p.H * p.W;
  const int num_tiles = p.N * tiles_d * tiles_h * tiles_w;

  // GEMM columns = the 343 taps FLATTENED: group g = q*8 + wave (q = 0..2, 24 groups of 16 columns), column j = lane
  // li holds tap t = 16 g + j = (kd*7 + kh)*7 + kw.  B[k][j] is then simply slab[position k + offset(t)] with a
  // per-lane constant offset, and 343 of 384 issued columns are real (the (kh-pair, kw<8) layout used 49 of 64).
  f32x4 acc[3][NQ];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
    for (int q = 0; q < NQ; ++q) acc[ci][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  int goff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int t = 16 * (q * 8 + wave) + li;
    if (t > 342) t = 342;                  // surplus column: recomputes the last tap, result dropped
    const int kw = t % 7, kh = (t / 7) % 7, kd = t / 49;
    goff[q] = kd * G::PLANE + kh * G::ROW + kw + 1;      // +1: slab starts one column early
  }
  // GEMM-k order: step ks = (td*TH + th)*4 + j covers output columns 2j + (lk >> 1) + 8*(lk & 1) of row (td, th)
  const int lane_col = (lk >> 1) + 8 * (lk & 1);
  float bsum = 0.f;

  // Tiles are PULLED, one at a time, from eight queues -- the depth-fastest tile order cut into one contiguous range per XCD; a
  // workgroup drains its own XCD's range first (neighbouring tiles share halo planes in that XCD's L2), then the others'.
  // A fixed run of tiles per workgroup made the whole launch as late as its latest workgroup: in the training step the
  // last weight gradients and the gradient finishing of the other layers run beside this kernel, and the workgroups that
  // shared their CU with those finished 0.3 ms after the rest (1.83 ms alone, 2.14 ms in the step).
  __shared__ int s_tile;
  unsigned *queue = g_stem_wgrad_queues + 16 * p.queue_slot;
  const int per_xcd = (num_tiles + 7) >> 3;
  const int home = blockIdx.x & 7;
  auto range = [&](int x, int &lo) { lo = x * per_xcd; return max(min(lo + per_xcd, num_tiles) - lo, 0); };
  auto take = [&](unsigned ticket) {          // thread 0: the home queue's ticket, else the other queues in turn
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
  unsigned ticket = 0;
  if (tid == 0) s_tile = take(atomicAdd(queue + home, 1u));
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane(s_tile);      // wave-uniform: the tile arithmetic below stays on the scalar unit
  while (tile >= 0) {
    const StemTile st = stem_tile_of(tile, tiles_d, tiles_h, tiles_w);
    const int n = st.n;
    const int d0 = st.td * TD, h0 = st.th * TH, w0 = st.tw * TW;
    if (tid == 0) ticket = atomicAdd(queue + home, 1u);      // the next tile's ticket: looked at after this tile's MFMAs
    if (p.pool_idx) {
      // dy = the max-pool backward of the pooled gradient, built in place: the tile's 4 x 8 x 16 positions are reached by
      // 3 x 5 x 9 windows; each window adds relu'(pooled) * g at its arg-max when that lies inside the tile (a window on the
      // tile border belongs to up to eight tiles, each keeps its own positions) -- the 335 MB un-pooled gradient of the
      // 512 volume is neither written nor read
      for (int it = tid; it < M * 4; it += kStemThreads) reinterpret_cast<float4 *>(dyt)[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
      for (int it = tid; it < 3 * 5 * 9 * 4; it += kStemThreads) {
        const int q4 = it & 3, pp = it >> 2;
        const int pw = pp % 9, ph = (pp / 9) % 5, pd = pp / 45;
        const int od = (d0 >> 1) + pd, oh = (h0 >> 1) + ph, ow = (w0 >> 1) + pw;
        if (od < p.Dp && oh < p.Hp && ow < p.Wp) {
          const size_t o = ((((size_t)n * p.Dp + od) * p.Hp + oh) * p.Wp + ow) * 4 + q4;
          const uint32_t ix = p.pool_idx[o];
          const float4 g = *reinterpret_cast<const float4 *>(p.dy + o * 4);
          float4 pa = make_float4(1.f, 1.f, 1.f, 1.f);
          if (p.pooled) pa = *reinterpret_cast<const float4 *>(p.pooled + o * 4);
          const float gv[4] = {g.x, g.y, g.z, g.w}, pv[4] = {pa.x, pa.y, pa.z, pa.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tap = (ix >> (8 * r)) & 255;
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int td = 2 * od - 1 + kd - d0, th = 2 * oh - 1 + kh - h0, tw = 2 * ow - 1 + kw - w0;
            if (pv[r] > 0.f && gv[r] != 0.f && (unsigned)td < (unsigned)TD && (unsigned)th < (unsigned)TH && (unsigned)tw < (unsigned)TW) {
              const int m = (td << 7) | (th << 4) | (((tw & 7) >> 1) << 2) | ((tw & 1) << 1) | (tw >> 3);
              atomicAdd(dyt + m * DYROW + q4 * 4 + r, gv[r]);
            }
          }
        }
      }
    } else {
      // dy tile [m'][16], m' = ks*4 + k in GEMM-k order, zeros outside
      for (int it = tid; it < M * 4; it += kStemThreads) {
        const int m = it >> 2, q4 = it & 3;
        const int k = m & 3, j = (m >> 2) & 3, th = (m >> 4) & 7, td = m >> 7;
        const int tw = 2 * j + (k >> 1) + 8 * (k & 1);
        const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (od < p.D && oh < p.Ho && ow < p.Wo) {
          const size_t at = ((((size_t)n * p.D + od) * p.Ho + oh) * p.Wo + ow) * 16 + q4 * 4;
          if (p.dy_bf16) {
            const uint2 q = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(p.dy) + at);
            v = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                            __uint_as_float(q.y & 0xffff0000u));
          } else {
            v = *reinterpret_cast<const float4 *>(p.dy + at);
          }
        }
        *reinterpret_cast<float4 *>(dyt + m * DYROW + q4 * 4) = v;
      }
    }
    // The k-steps of a channel are FULLY unrolled: every LDS address is then a per-lane base register (one for the dy column, one
    // per tap-column group) plus an immediate offset, and the loop carries no vector instruction besides its MFMAs -- on this
    // chip a vector ALU instruction does not hide under an fp32 MFMA (tools/micro/mfma_clock.hip: 14.7 ns per MFMA and SIMD bare,
    // 17.4 with ten integer adds per six MFMAs), and the four address adds per k-step cost this kernel 0.15 ms.  The three
    // channels share the one unrolled body: it always accumulates into acc[0], and the accumulator groups rotate after it.
    const float *abase = dyt + lk * DYROW + li;
    const float *bbase[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) bbase[q] = slab + goff[q] + 2 * lane_col;
#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci) {
      if (ci) __syncthreads();
      stem_wgrad_stage_channel(slab, p.x + ((size_t)n * 3 + ci) * vox, p.D, p.H, p.W, d0, h0, w0, tid);
      __syncthreads();
      if (ci == 0 && wave == 0) {                       // bias gradient: this wave's column sums of the dy tile
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int ks = 0; ks < M / 4; ks += 2) { s0 += abase[ks * 4 * DYROW]; s1 += abase[(ks + 1) * 4 * DYROW]; }
        bsum += s0 + s1;
      }
      auto ld = [&](int ks, float &a, float (&bv)[NQ]) {
        const int j = ks & 3, th = (ks >> 2) & 7, td = ks >> 5;
        a = abase[ks * 4 * DYROW];                                      // A[i = co][k]
        const int pos = td * G::PLANE + 2 * th * G::ROW + 4 * j;        // halo origin of that k-step's positions (immediate)
#pragma unroll
        for (int q = 0; q < NQ; ++q) bv[q] = bbase[q][pos];
      };
      float a0, a1, b0[NQ], b1[NQ];
      ld(0, a0, b0);
#pragma unroll
      for (int ks = 0; ks < M / 4; ks += 2) {
        ld(ks + 1, a1, b1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[q], acc[0][q], 0, 0, 0);
        if (ks + 2 < M / 4) ld(ks + 2, a0, b0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[q], acc[0][q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {                    // rotate: after the third channel the groups are back in place
        const f32x4 t = acc[0][q];
        acc[0][q] = acc[1][q]; acc[1][q] = acc[2][q]; acc[2][q] = t;
      }
    }
    if (tid == 0) s_tile = take(ticket);
    __syncthreads();                       // also: every wave is done with this tile's dy rows and slab
    tile = __builtin_amdgcn_readfirstlane(s_tile);
  }
  if (tid == 0) {                          // the last workgroup to run dry re-arms the queues for the next launch
    const unsigned done = atomicAdd(queue + 8, 1u);
    if (done == gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 9; ++x) atomicExch(queue + x, 0u);
    }
  }

  // D[i = co][j]: lane holds column j = li (tap t), rows co = lk*4 + r
  float *part = p.part ? p.part + (size_t)blockIdx.x * (343 * 3 * 16) : nullptr;
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int t = 16 * (q * 8 + wave) + li;
      if (t < 343) {
        if (part) {
          *reinterpret_cast<float4 *>(part + ((size_t)t * 3 + ci) * 16 + lk * 4) =
              make_float4(acc[ci][q][0], acc[ci][q][1], acc[ci][q][2], acc[ci][q][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(p.dw + ((size_t)t * 4 + ci) * 16 + lk * 4 + r, acc[ci][q][r]);
        }
      }
    }
  if (p.dbias && wave == 0) {
    float s = bsum;
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lk == 0) atomicAdd(p.dbias + li, s);
  }
}

}  // namespace

static int stem_forward_impl(const float *x_ncdhw, const float *w_packed, const float *shift, float *y, int y_bf16, int n, int d,
                             int h, int w, int relu, mr3d_stream_t stream_) {
  if (!x_ncdhw || !w_packed || !y || n <= 0 || d <= 0 || h <= 0 || w <= 0) return MR3D_ERR_INVALID_ARG;
  if ((long long)d * h * w >= (1LL << 28)) return MR3D_ERR_UNSUPPORTED;       // byte offsets inside a channel volume below 2^30
  if ((long long)n * d * ((h - 1) / 2 + 1) * ((w - 1) / 2 + 1) * 64 >= (1LL << 31)) return MR3D_ERR_UNSUPPORTED;   // ... and in the output
  StemFwdArgs a;
  a.x = x_ncdhw; a.wf = w_packed; a.shift = shift; a.y = y; a.y_bf16 = y_bf16;
  a.N = n; a.D = d; a.H = h; a.W = w;
  a.Ho = (h + 6 - 7) / 2 + 1; a.Wo = (w + 6 - 7) / 2 + 1;
  a.relu = relu;
  // 4x8x16 output tile: 33 KiB slab; direct form: four 4-wave workgroups per CU = 4 waves per SIMD, so the matrix
  // pipe always has a ready wave while others sit at a barrier / wait for LDS (measured: 2 waves per SIMD left it idle
  // 40 % of the time); the Winograd form needs 166 registers: three workgroups per CU
  constexpr int TD = 4, TW = 16;
  using G = StemGeo<TD, 8, TW, 40>;
  constexpr int lds_bytes = G::SLAB * 4;
  const int td = (d + TD - 1) / TD, th = (a.Ho + 7) / 8, tw = (a.Wo + TW - 1) / TW;
  // Winograd along depth (1.79 -> 1.43 ms on the 512 volume)
  const unsigned grid = 8u * (unsigned)((n * th * tw + 7) / 8) * (unsigned)td;           // stem_tile_xcd
  stem_fwd_wino_kernel<TW><<<grid, 256, lds_bytes, mr3d_stream(stream_)>>>(a, td, th, tw);
  return mr3d_launch_status();
}

extern "C" int mr3d_stem_conv_forward(const float *x_ncdhw, const float *w_packed, const float *shift, float *y, int n, int d,
                                      int h, int w, int relu, mr3d_stream_t stream_) {
  return stem_forward_impl(x_ncdhw, w_packed, shift, y, 0, n, d, h, w, relu, stream_);
}

extern "C" int mr3d_stem_conv_forward_bf16out(const float *x_ncdhw, const float *w_packed, const float *shift, void *y_bf16, int n,
                                              int d, int h, int w, int relu, mr3d_stream_t stream_) {
  return stem_forward_impl(x_ncdhw, w_packed, shift, (float *)y_bf16, 1, n, d, h, w, relu, stream_);
}

static int stem_wgrad_impl(const float *x_ncdhw, const float *dy, int dy_bf16, float *dw_packed, float *dbias, int n, int d, int h,
                           int w, float *workspace, long long workspace_bytes, mr3d_stream_t stream_,
                           const uint32_t *pool_idx = nullptr, const float *pooled = nullptr);

extern "C" int mr3d_stem_conv_wgrad_pooled(const float *x_ncdhw, const float *g_pooled, const uint32_t *argmax, const float *pooled,
                                           float *dw_packed, float *dbias, int n, int d, int h, int w, float *workspace,
                                           long long workspace_bytes, mr3d_stream_t stream_) {
  if (!argmax) return MR3D_ERR_INVALID_ARG;
  if (x_ncdhw && g_pooled && dw_packed && n > 0 && d > 0 && h > 0 && w > 0 && mr3d_stem_wgrad_gather_ok(d, h, w))
    return mr3d_stem_wgrad_gather(x_ncdhw, g_pooled, argmax, pooled, 0, dw_packed, dbias, n, d, h, w, workspace, workspace_bytes,
                                  mr3d_stream(stream_));
  return stem_wgrad_impl(x_ncdhw, g_pooled, 0, dw_packed, dbias, n, d, h, w, workspace, workspace_bytes, stream_, argmax, pooled);
}

extern "C" int mr3d_stem_conv_wgrad_bf16dy(const float *x_ncdhw, const void *dy_bf16, float *dw_packed, float *dbias, int n, int d,
                                           int h, int w, mr3d_stream_t stream_) {
  return stem_wgrad_impl(x_ncdhw, (const float *)dy_bf16, 1, dw_packed, dbias, n, d, h, w, nullptr, 0, stream_);
}

extern "C" int mr3d_stem_conv_wgrad(const float *x_ncdhw, const float *dy, float *dw_packed, float *dbias, int n, int d, int h,
                                    int w, mr3d_stream_t stream_) {
  return mr3d_stem_conv_wgrad_ws(x_ncdhw, dy, dw_packed, dbias, n, d, h, w, nullptr, 0, stream_);
}

extern "C" int mr3d_stem_conv_wgrad_ws(const float *x_ncdhw, const float *dy, float *dw_packed, float *dbias, int n, int d, int h,
                                       int w, float *workspace, long long workspace_bytes, mr3d_stream_t stream_) {
  return stem_wgrad_impl(x_ncdhw, dy, 0, dw_packed, dbias, n, d, h, w, workspace, workspace_bytes, stream_);
}

static int stem_wgrad_impl(const float *x_ncdhw, const float *dy, int dy_bf16, float *dw_packed, float *dbias, int n, int d, int h,
                           int w, float *workspace, long long workspace_bytes, mr3d_stream_t stream_, const uint32_t *pool_idx,
                           const float *pooled) {
  if (!x_ncdhw || !dy || !dw_packed || n <= 0 || d <= 0 || h <= 0 || w <= 0) return MR3D_ERR_INVALID_ARG;
  if ((long long)d * h * w >= (1LL << 28)) return MR3D_ERR_UNSUPPORTED;       // byte offsets inside a channel volume below 2^30
  StemWgradArgs a;
  a.x = x_ncdhw; a.dy = dy; a.dw = dw_packed; a.dbias = dbias; a.dy_bf16 = dy_bf16;
  a.N = n; a.D = d; a.H = h; a.W = w;
  a.Ho = (h + 6 - 7) / 2 + 1; a.Wo = (w + 6 - 7) / 2 + 1;
  a.pool_idx = pool_idx; a.pooled = pooled;
  a.Dp = (d - 1) / 2 + 1; a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;
  constexpr int lds_bytes = (StemWgGeo::SLAB + 4 * 8 * 16 * 16) * 4;       // 66.0 KiB: two workgroups per CU
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(stem_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds_bytes) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr = true;
  }
  const int td = (d + 3) / 4, th = (a.Ho + 7) / 8, tw = (a.Wo + 15) / 16;
  int tiles = n * td * th * tw;
  static int cus = 0;        // queried once: hipGetDeviceProperties is slow and must not run inside a stream capture
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  int grid = tiles < 2 * cus ? tiles : 2 * cus;
  if (grid >= 8) grid &= ~7;
  // tile queues: a ring of 64 slots in a zero-initialised device global; a launch leaves its slot zeroed again, so only
  // launches that are in flight at the same time need different slots
  static std::atomic<unsigned> next_slot{0};
  a.queue_slot = (int)(next_slot.fetch_add(1) & 63u);
  // two-stage reduction when the caller lends a workspace: 512 workgroups x 16,464 same-address atomics are ~0.13 ms
  a.part = (workspace && workspace_bytes >= (long long)grid * 343 * 3 * 16 * 4) ? workspace : nullptr;
  stem_wgrad_kernel<<<(unsigned)grid, kStemThreads, lds_bytes, mr3d_stream(stream_)>>>(a, td, th, tw);
  const int st = mr3d_launch_status();
  if (st != MR3D_OK || !a.part) return st;
  return mr3d_reduce_partials(a.part, grid, 343, 3, 16, 1, 1, 0, 0, 4, 16, dw_packed, mr3d_stream(stream_));
}

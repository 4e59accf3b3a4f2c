// 3x3x3 / stride-1 convolution (forward and, with flipped transposed weights, data gradient) as a TWO-dimensional Winograd
// form on the fp32 matrix cores: F(4,3) along W x F(2,3) along H.  Same contract as conv3d_fwd.hip's kernels (Conv3dFwdArgs:
// NDHWC fp32 activations, [tap][Cin][Cout] weights, the fused epilogue: shift / residual adds / nearest-upsample add / ReLU /
// ReLU mask / strided output grid); replaces cuDNN's conv3d of the FPN output convs and the RPN conv on the large maps
// (mmdet/models/necks/fpn3d.py:56-64,111-113, anchor_heads/rpn_head_3d.py:19-24) where the one-dimensional F(4,3) kernel of
// conv3d_fwd.hip issues 1/2 of the algorithmic MFMA work: this form issues 1/3.
//
// A GEMM row is a 2 (h) x 4 (w) patch of outputs.  Per kd tap and input channel the patch needs the 4 x 6 input window d;
//   V = Bh^T d Bw        (24 values; Bh^T = F(2,3)'s (d0-d2, d1+d2, d2-d1, d1-d3), Bw^T = Lavin & Gray's F(4,3) matrix)
//   U = Gh g Gw^T        (24 values per (kd, ci, co), from the 3 x 3 (kh, kw) taps g: computed once per launch -- once per plan
//                         stretch for all such convolutions, mr3d_conv3d_wino2d_prepare -- into a scratch buffer, [kd][Cin][6][Cout][4]: product f = fh*6+fw at [f/4][co][f%4])
//   M[fh][fw] += V[fh][fw] * U[fh][fw]   summed over kd and ci on the matrix cores: 24 accumulator sets,
//   Y = Ah^T M Aw        (8 outputs) in the epilogue -- 72 MFMA products per 8 outputs and input channel instead of 216 (direct)
//                         or 108 (F(4,3) along W only).
// v_mfma_f32_16x16x4_f32: a wave owns 16 GEMM rows x 16 output channels (24 x 4 accumulator registers); the four waves of a
// workgroup share the 16 rows (2 x 8 x 8 outputs) and split a 64-wide Cout tile.  Per Cin chunk (16 channels) the workgroup
// stages the halo ([4][10][10] positions, odd row pitch) in LDS, forms V ONCE for its 4 planes x 8 patches x 16 channels
// (two items per thread, 88 VALU operations each; the windows are pulled into registers first and V is written OVER the halo:
// 49 KB of LDS) in a layout whose fragment reads are conflict-free, and then runs 12 (kd, k-step) steps of 24 MFMAs whose A
// fragments are plain LDS reads (half a step ahead) and whose U fragments come
// straight from L2, two steps ahead (an L2 round trip is longer than one step).  The halo of the next chunk is fetched at the
// chunk boundary (the second workgroup of the CU covers it; fetching it piecewise under the MFMA steps was measured equal).
// Measured on the P2 FPN / RPN conv (40 x 128 x 128, 64 -> 64): 640-660 us against 834 us for the F(4,3) kernel; with every
// operand load removed from the steps 570 us, with the staging and V phases of chunks 1-3 removed as well 492 us (the bare MFMA
// stream: 0.62 of the 2.4 GHz matrix peak).  Tried and not kept: every wave forming its own V under its MFMAs (720 us: four
// times the VALU work, U one step ahead for lack of registers), two MFMA tiles per wave at one wave per SIMD (925), a
// producer / consumer split of a 512-thread workgroup (770: one MFMA wave per SIMD does not fill the pipe), staggered starts.
// The narrow-Cout kernels further down (conv3d_wino2d_rows_kernel) keep the per-wave V: their waves own different rows.
#include "conv3d.h"
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace {

constexpr int kThreads = 256;
#ifndef MR3D_ROWS_ABL
#define MR3D_ROWS_ABL 0
#endif
constexpr int RABL = MR3D_ROWS_ABL;                    // timing-only ablations of conv3d_wino2d_rows_kernel (tools/diag); 0 in the library

constexpr int TD = 2, TH = 8, TW = 8;                  // output tile: 16 GEMM rows (patches of 2 x 4 outputs)
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;   // halo
constexpr int HHP = HH + 1;                            // one spare row per halo plane: 2-way instead of 3-way LDS bank conflicts in the V phase
constexpr int CC = 16, ROW = CC + 1;                   // Cin chunk, LDS row pitch in floats
constexpr int HP = HD * HHP * HW;
constexpr int NFC = 24;
constexpr int VPITCH = NFC * CC + 2;                    // floats per (plane, patch) of V: 2 * (plane * 8 + patch) + k spreads a read over all 32 banks
constexpr int VFLOATS = HD * 8 * VPITCH;
constexpr int LDS_BYTES = (VFLOATS > HP * ROW ? VFLOATS : HP * ROW) * 4;      // V overlays the halo
constexpr int NF = 24;                                 // products per patch and kd tap
// conv3d_wino2d_rows_kernel reads its 4 x 6 windows straight from the halo: a lane's address is (plane, row pair, column quad, k)
// of its patch, and with the plain geometry (row pitch 10, planes of 1700 floats) the plane and the column quad both moved a lane by
// 4 banks: 3-way conflicts, 65 % of the kernel's LDS-active cycles.  A row pitch of 12 positions and 2 floats between planes make the
// 32 lanes of a read hit 32 banks (searched over row / plane pitches with the kernel's own address map).
constexpr int HWR = 12, RPAD = 2;
constexpr int rows_plane(int hh) { return hh * HWR * ROW + RPAD; }

// U(kd, f = fh*6+fw, ci, co) = sum_kh sum_kw Gh[fh][kh] Gw[fw][kw] w[(kd*3+kh)*3+kw][ci][co]
// Uw (may be null): the same values as [kd][ci][fh][co][6] for conv3d_wino2d_wide_kernel
__global__ __launch_bounds__(256) void wino2d_weights_kernel(const float *__restrict__ w, float *__restrict__ U, float *__restrict__ Uw, int cin, int cout) {
  const int e = blockIdx.x * 256 + threadIdx.x;          // (kd, ci, co)
  if (e >= 3 * cin * cout) return;
  const int kd = e / (cin * cout), cc = e - kd * (cin * cout);
  float g[3][3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) g[kh][kw] = w[(size_t)((kd * 3 + kh) * 3 + kw) * cin * cout + cc];
  float t[3][6];                                         // along W first: t[kh][fw]
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const float g0 = g[kh][0], g1 = g[kh][1], g2 = g[kh][2];
    const float s02 = g0 + g2, h = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), q = g1 * (1.f / 12.f);
    t[kh][0] = 0.25f * g0; t[kh][1] = (s02 + g1) * (-1.f / 6.f); t[kh][2] = (s02 - g1) * (-1.f / 6.f);
    t[kh][3] = h + q; t[kh][4] = h - q; t[kh][5] = g2;
  }
#pragma unroll
  for (int fw = 0; fw < 6; ++fw) {
    const float a = t[0][fw], b = t[1][fw], c = t[2][fw];
    const float u[4] = {a, 0.5f * ((a + c) + b), 0.5f * ((a + c) - b), c};
#pragma unroll
    for (int fh = 0; fh < 4; ++fh) {                       // [kd][ci][f / 4][co][f % 4]: a lane's four consecutive products in 16 bytes
      const int f = fh * 6 + fw, ci = cc / cout, co = cc - ci * cout;
      U[((((size_t)kd * cin + ci) * (NF / 4) + f / 4) * cout + co) * 4 + (f & 3)] = u[fh];
      if (Uw) Uw[((((size_t)kd * cin + ci) * 4 + fh) * cout + co) * 6 + fw] = u[fh];
    }
  }
}

// LIVE: the caller lent a voxel liveness bitmap of x (mr3d_voxel_liveness): a tile whose halo holds no live voxel is written from its
// epilogue alone.  The plain instance carries no test (the kernel sits at 251 registers; the test costs the dense case 1.4 %).
template <bool LIVE>
__global__ __launch_bounds__(kThreads, 2) void conv3d_wino2d_kernel(Conv3dFwdArgs p, const float *__restrict__ U, int tiles_d,
                                                                   int tiles_h, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *vl = lds;                                       // V[(plane * 8 + patch)][f][ci], pitch VPITCH per (plane, patch): OVER the halo (see below)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int tw_i = b % tiles_w; b /= tiles_w;
  const int th_i = b % tiles_h; b /= tiles_h;
  const int td_i = b % tiles_d; b /= tiles_d;
  const int n = b;
  const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
  const int id0 = od0 - p.pd, ih0 = oh0 - p.ph, iw0 = ow0 - p.pw;
  // halo origin (may lie in front of x: only addresses inside the volume are dereferenced)
  const float *xtile = p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * p.Cin;
  const int li = lane & 15, lk = lane >> 4;
  const int co = blockIdx.y * 64 + wave * 16 + li;
  // A operand: this lane's GEMM row li = (output plane li >> 3, patch li & 7) and its k index inside the k-step (lk)
  const float *vlane = vl + li * VPITCH + lk;
  // U fragments: scalar 64-bit base of the (kd, k-step) + this lane's 32-bit offset (vector ALU instructions do not hide under
  // fp32 MFMAs on this chip: no 64-bit address arithmetic per load)
  const unsigned ul_off = (unsigned)((lk * (NF / 4) * p.Cout + co) * 4 * 4);             // bytes
  const __amdgpu_buffer_rsrc_t ursrc = buffer_of(U), xrsrc = buffer_of(xtile);

  f32x4 acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float bq[3][NF], aq[2][NF / 2];
  auto load_b = [&](float (&dst)[NF], int c0, int kd, int kk) {
    const unsigned src = (unsigned)(((kd * p.Cin + c0 + kk * 4) * (NF / 4) * p.Cout * 4) * 4);      // (scalar byte offset: U is < 2 GiB)
#pragma unroll
    for (int f4 = 0; f4 < NF / 4; ++f4) {
      const float4 v = buffer_load4(ursrc, ul_off, src + (unsigned)(f4 * p.Cout * 4 * 4));
      dst[f4 * 4] = v.x; dst[f4 * 4 + 1] = v.y; dst[f4 * 4 + 2] = v.z; dst[f4 * 4 + 3] = v.w;
    }
  };
  // half h of step s: products f = 12 h .. 12 h + 11 of V(plane (li >> 3) + kd, this lane's patch, channel kk * 4 + lk)
  auto load_a = [&](float (&dst)[NF / 2], int s, int h) {
    const float *src = vlane + (s / 4) * 8 * VPITCH + (s % 4) * 4 + h * (NF / 2) * CC;
#pragma unroll
    for (int f = 0; f < NF / 2; ++f) dst[f] = src[f * CC];
  };
  constexpr int Q = CC / 4;
  // Halo staging.  A thread owns ONE (column, channel quad) of the halo rows rg, rg + 6, rg + 12, ... (40 lanes per row of
  // 10 columns x 4 quads, 6 row groups: 240 of the 256 threads, 7 pieces): the validity bits of its seven pieces are formed
  // ONCE, in front of the chunk loop; a piece's offset is three 24-bit multiply-adds from scalars.  The piece-per-thread map this replaced recomputed three divisions, three bounds
  // tests and a product chain per piece and chunk, twice (fetch and commit): 540 vector instructions, a fifth of them 16-cycle
  // integer multiplies, per 288 MFMAs -- and vector ALU work does not hide under fp32 MFMAs on this chip.
  constexpr int NIT = (HD * HH + 5) / 6;                 // 7
  unsigned s_ok = 0;                                     // bit j: piece j of this thread lies inside the volume
  {
    const int lane40 = tid % (HW * Q), rg = tid / (HW * Q), hw = lane40 / Q;
    const bool col_ok = rg < 6 && (unsigned)(iw0 + hw) < (unsigned)p.W;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int row = rg + 6 * j, hd = row / HH, hh = row - hd * HH;
      const bool ok = col_ok && row < HD * HH && (unsigned)(id0 + hd) < (unsigned)p.D && (unsigned)(ih0 + hh) < (unsigned)p.H;
      s_ok |= ok ? (1u << j) : 0u;
    }
  }
  const int s_plane = p.H * p.W * p.Cin, s_rowp = p.W * p.Cin;               // (scalars; the halo spans < 2^24 elements)
  // the thread's own coordinates are re-derived per chunk from an opaque copy of tid: kept across the MFMA steps they cost
  // registers this kernel does not have (it spilled 64 bytes per lane with them hoisted)
  struct Own { int rg, hw, q; };
  auto own = [&]() {
    int tv = tid;
    asm volatile("" : "+v"(tv));
    Own o;
    o.rg = tv / (HW * Q);
    const int lane40 = tv - o.rg * (HW * Q);
    o.hw = lane40 / Q; o.q = lane40 - o.hw * Q;
    return o;
  };
  auto fetch = [&](const Own &o, int j, int c0) {
    // a piece outside the volume reads beyond the descriptor's range: zeros, no memory touched, and no branch around the load (a load
    // under a lane mask makes the compiler drain the vector-memory queue -- the next chunk's U fragments -- at the join)
    const int row = o.rg + 6 * j;
    const int hd = (row >= HH) + (row >= 2 * HH) + (row >= 3 * HH), hh = row - hd * HH;
    const int rel = (int)__umul24(hd, s_plane) + (int)__umul24(hh, s_rowp) + (int)__umul24(o.hw, p.Cin) + o.q * 4;   // full-rate multiplies
    return buffer_load4(xrsrc, ((s_ok >> j) & 1u) ? (unsigned)rel * 4u : 0xfffffff0u, (unsigned)c0 * 4u);
  };
  auto commit = [&](const Own &o, int j, const float4 &v) {
    const int row = o.rg + 6 * j;                        // (hd * HHP + hh) = row + hd: one spare row per plane
    if (o.rg < 6 && row < HD * HH) {
      const int hd = (row >= HH) + (row >= 2 * HH) + (row >= 3 * HH);
      float *dd = lds + ((row + hd) * HW + o.hw) * ROW + o.q * 4;
      dd[0] = v.x; dd[1] = v.y; dd[2] = v.z; dd[3] = v.w;
    }
  };
  // one (kd, k-step) s of a chunk in two halves of 12 MFMAs on aq[half & 1] and bq[s % 3]: the A fragments of the next half
  // (LDS) are requested before a half's MFMAs, the U fragments of step s + 2 (L2) before the step's
  auto step = [&](int s, int c0) {
    const int s2 = s + 2;
    if (s2 < 12) load_b(bq[s2 % 3], c0, s2 / 4, s2 % 4);
    else if (c0 + CC < p.Cin) load_b(bq[s2 % 3], c0 + CC, (s2 - 12) / 4, (s2 - 12) % 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 0) load_a(aq[1], s, 1);
      else if (s + 1 < 12) load_a(aq[0], s + 1, 0);
      __builtin_amdgcn_sched_barrier(0);                 // the requests stay up here: the scheduler would sink them next to their uses
#pragma unroll
      for (int f = 0; f < NF / 2; ++f)
        acc[h * (NF / 2) + f] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[h][f], bq[s % 3][h * (NF / 2) + f], acc[h * (NF / 2) + f], 0, 0, 0);
    }
  };
  // sparse input (the data gradient of the RPN conv / an FPN output conv in detector training): one bit per voxel of x; lane r < 40
  // tests the 10 voxels of halo row (plane r / 10, row r % 10); all clear = nothing to multiply (wave-uniform)
  bool dead = false;
  if constexpr (LIVE) {
    const int hd = lane / HH, hh = lane - hd * HH;
    const int id = id0 + hd, ih = ih0 + hh;
    unsigned bits = 0;
    if (lane < HD * HH && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H) {
      const int w_lo = max(iw0, 0), w_hi = min(iw0 + HW, p.W);                  // [w_lo, w_hi) of this row
      const long long b0 = (((long long)n * p.D + id) * p.H + ih) * p.W + w_lo;
      const unsigned *wd = reinterpret_cast<const unsigned *>(p.x_live) + (b0 >> 5);
      const int sh = (int)(b0 & 31), cnt = w_hi - w_lo;                          // cnt <= 10: at most two words
      const unsigned long long two = (unsigned long long)wd[0] | ((unsigned long long)(sh + cnt > 32 ? wd[1] : 0u) << 32);
      bits = (unsigned)(two >> sh) & ((1u << cnt) - 1u);
    }
    dead = __ballot(bits != 0) == 0;
  }
  if (!dead) {
  load_b(bq[0], 0, 0, 0);
  load_b(bq[1], 0, 0, 1);
  }
  for (int c0 = 0; c0 < (dead ? 0 : p.Cin); c0 += CC) {
    __syncthreads();                                     // the previous chunk's steps are done with V (and its V phase with the halo)
    {
      float4 stage[NIT];
      const Own o = own();
#pragma unroll
      for (int j = 0; j < NIT; ++j) stage[j] = fetch(o, j, c0);
#pragma unroll
      for (int j = 0; j < NIT; ++j) commit(o, j, stage[j]);
    }
    __syncthreads();
    // ---- V = Bh^T d Bw of every (plane, patch, channel) of the chunk: 4 * 8 * 16 = 512 items, two per thread.  V is written
    // over the halo (49 KB of LDS instead of 76): every thread first pulls its two 4 x 6 windows into registers.
    constexpr int REP = HD * 8 * CC / kThreads;
    float d[REP][4][6];
#pragma unroll
    for (int rep = 0; rep < REP; ++rep) {
      const int item = tid + rep * kThreads;
      const int ci = item & 15, patch = (item >> 4) & 7, plane = item >> 7;
      const float *src = lds + (((plane * HHP) + (patch >> 1) * 2) * HW + (patch & 1) * 4) * ROW + ci;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[rep][i][j] = src[(i * HW + j) * ROW];
    }
    __syncthreads();                                     // every window is in registers: the halo may be overwritten
#pragma unroll
    for (int rep = 0; rep < REP; ++rep) {
      const int item = tid + rep * kThreads;
      const int ci = item & 15, patch = (item >> 4) & 7, plane = item >> 7;
      float *dst = vl + (plane * 8 + patch) * VPITCH + ci;
#pragma unroll
      for (int fh = 0; fh < 4; ++fh) {
        float t[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
          t[j] = fh == 0 ? d[rep][0][j] - d[rep][2][j] : (fh == 1 ? d[rep][1][j] + d[rep][2][j] : (fh == 2 ? d[rep][2][j] - d[rep][1][j] : d[rep][1][j] - d[rep][3][j]));
        const float e = t[4] - t[2], o = t[3] - t[1];
        dst[(fh * 6 + 0) * CC] = 4.f * t[0] - 5.f * t[2] + t[4];
        dst[(fh * 6 + 1) * CC] = (t[3] + t[4]) - 4.f * (t[1] + t[2]);
        dst[(fh * 6 + 2) * CC] = (t[4] - t[3]) + 4.f * (t[1] - t[2]);
        dst[(fh * 6 + 3) * CC] = e + 2.f * o;
        dst[(fh * 6 + 4) * CC] = e - 2.f * o;
        dst[(fh * 6 + 5) * CC] = 4.f * t[1] - 5.f * t[3] + t[5];
      }
    }
    __syncthreads();
    load_a(aq[0], 0, 0);
#pragma unroll
    for (int s = 0; s < 12; ++s) step(s, c0);             // the U buffers rotate with period 3, the A buffers with period 2
  }

  // ---- epilogue: Y = Ah^T M Aw per accumulator element, then shift / adds / upsample add / ReLU / mask; NDHWC store
  const float sh = p.shift ? p.shift[co] : 0.f;
  // Output addressing: a SCALAR 64-bit index of the tile's first output + 32-bit lane offsets (one per accumulator row, the row's
  // eight outputs are scalar steps away).  (One size_t product chain per output was ~1,600 vector instructions per tile, 300 of them 16-cycle multiplies.)
  const size_t ybase = ((((size_t)n * p.yD + (size_t)od0 * p.ysd) * p.yH + (size_t)oh0 * p.ysh) * p.yW + (size_t)ow0 * p.ysw) * p.Cout;
  const float *add1b = p.add1 ? p.add1 + ybase : nullptr, *add2b = p.add2 ? p.add2 + ybase : nullptr, *maskb = p.mask ? p.mask + ybase : nullptr;
  float *yb = p.y + ybase;
  const int xstep = p.ysw * p.Cout * 4, rowstep = p.ysh * p.yW * p.Cout * 4, planestep = p.ysd * p.yH * p.yW * p.Cout * 4;   // bytes
  auto emit = [&](unsigned rel, int od, int oh, int ow, float val) {
    if (od < p.Do && oh < p.Ho && ow < p.Wo) {
      float v = val + sh;
      if (add1b) v += *reinterpret_cast<const float *>(reinterpret_cast<const char *>(add1b) + rel);
      if (add2b) v += *reinterpret_cast<const float *>(reinterpret_cast<const char *>(add2b) + rel);
      if (p.up) {
        auto nearest = [](int dst, int in, int out) {          // torch nearest: min(floor(dst * (in/out)), in-1)
          const int s_ = (int)floorf((float)dst * ((float)in / (float)out));
          return s_ < in - 1 ? s_ : in - 1;
        };
        const int ud = nearest(od, p.Du, p.Do), uh = nearest(oh, p.Hu, p.Ho), uw = nearest(ow, p.Wu, p.Wo);
        v += p.up[((((size_t)n * p.Du + ud) * p.Hu + uh) * p.Wu + uw) * p.Cout + co];
      }
      if (p.relu) v = fmaxf(v, 0.f);
      if (maskb) v = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(maskb) + rel) > 0.f ? v : 0.f;
      *reinterpret_cast<float *>(reinterpret_cast<char *>(yb) + rel) = v;
    }
  };
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lk * 4 + r;
    const int dz = row >> 3, dh = ((row >> 1) & 3) * 2, dw = (row & 1) * 4;
    const int od = od0 + dz, oh = oh0 + dh, ow = ow0 + dw;
    float z[4][4];                                       // along W: z[fh][x]
#pragma unroll
    for (int fh = 0; fh < 4; ++fh) {
      const float m0 = acc[fh * 6][r], m1 = acc[fh * 6 + 1][r], m2 = acc[fh * 6 + 2][r], m3 = acc[fh * 6 + 3][r],
                  m4 = acc[fh * 6 + 4][r], m5 = acc[fh * 6 + 5][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      z[fh][0] = (m0 + s12) + s34; z[fh][1] = d12 + 2.f * d34; z[fh][2] = s12 + 4.f * s34; z[fh][3] = (d12 + 8.f * d34) + m5;
    }
    const unsigned rel0 = (unsigned)(dz * planestep + dh * rowstep + dw * xstep + co * 4);        // byte offset inside the tile
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      emit(rel0 + x * xstep, od, oh, ow + x, (z[0][x] + z[1][x]) + z[2][x]);
      emit(rel0 + x * xstep + rowstep, od, oh + 1, ow + x, (z[1][x] - z[2][x]) - z[3][x]);
    }
  }
}

// Narrow Cout tiles (16 or 32 output channels: the 3x3x3 convs of layer1 / layer2 on the large maps).  WN waves split the Cout
// tile, the other 4 / WN waves split the GEMM rows: a wave owns 16 rows = 2 output planes x 8 patches of its own, so no two
// waves need the same V and every wave forms it itself, under the MFMAs of the previous step (88 VALU operations per 24 MFMAs).
// Workgroup tile: (8 / WN) x 8 x 8 outputs; U (3 x 24 x Cin x Cout floats) stays L1 / L2 resident.
template <int WN>
__global__ __launch_bounds__(kThreads, 2) void conv3d_wino2d_rows_kernel(Conv3dFwdArgs p, const float *__restrict__ U, int tiles_d,
                                                                        int tiles_h, int tiles_w) {
  constexpr int WM = 4 / WN, TDr = 2 * WM, HDr = TDr + 2, HHr = HH;   // no spare halo row here: 68 KB at WN = 1
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int tw_i = b % tiles_w; b /= tiles_w;
  const int th_i = b % tiles_h; b /= tiles_h;
  const int td_i = b % tiles_d; b /= tiles_d;
  const int n = b;
  const int od0 = td_i * TDr, oh0 = th_i * TH, ow0 = tw_i * TW;
  const int id0 = od0 - p.pd, ih0 = oh0 - p.ph, iw0 = ow0 - p.pw;
  const float *xtile = p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * p.Cin;       // see conv3d_wino2d_kernel
  const int li = lane & 15, lk = lane >> 4;
  const int wave_m = wave / WN, wave_n = wave % WN;
  const int co = blockIdx.y * (16 * WN) + wave_n * 16 + li;
  // A operand: this lane's GEMM row (li) = patch (td, th2, tw4) and its k index inside the k-step (lk)
  constexpr int PL = rows_plane(HHr);
  const int abase = (2 * wave_m + (li >> 3)) * PL + (((li >> 1) & 3) * 2 * HWR + (li & 1) * 4) * ROW + lk;
  const unsigned ul_off = (unsigned)((lk * (NF / 4) * p.Cout + co) * 4 * 4);             // bytes
  const __amdgpu_buffer_rsrc_t ursrc = buffer_of(U), xrsrc = buffer_of(xtile);

  constexpr int MT = 1;
  f32x4 acc[MT][NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[0][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float b0[NF], b1[NF], a0[MT][NF], a1[MT][NF], d[MT][4][6];
  auto load_b = [&](float (&dst)[NF], int c0, int kd, int kk) {
    const unsigned src = (unsigned)(((kd * p.Cin + c0 + kk * 4) * (NF / 4) * p.Cout * 4) * 4);      // (scalar byte offset)
#pragma unroll
    for (int f4 = 0; f4 < NF / 4; ++f4) {
      const float4 v = buffer_load4(ursrc, ul_off, src + (unsigned)(f4 * p.Cout * 4 * 4));
      dst[f4 * 4] = v.x; dst[f4 * 4 + 1] = v.y; dst[f4 * 4 + 2] = v.z; dst[f4 * 4 + 3] = v.w;
    }
  };
  auto read_d = [&](int kd, int kk) {                    // the 4 x 6 input window of this lane's patch and channel
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float *src = lds + abase + (kd + 2 * mt) * PL + kk * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[mt][i][j] = src[(i * HWR + j) * ROW];
    }
  };
  auto xform = [&](int mt, int fh, float (&dst)[NF]) {   // row fh of V = Bh^T d Bw
    float t[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
      t[j] = fh == 0 ? d[mt][0][j] - d[mt][2][j] : (fh == 1 ? d[mt][1][j] + d[mt][2][j] : (fh == 2 ? d[mt][2][j] - d[mt][1][j] : d[mt][1][j] - d[mt][3][j]));
    const float e = t[4] - t[2], o = t[3] - t[1];
    dst[fh * 6 + 0] = 4.f * t[0] - 5.f * t[2] + t[4];
    dst[fh * 6 + 1] = (t[3] + t[4]) - 4.f * (t[1] + t[2]);
    dst[fh * 6 + 2] = (t[4] - t[3]) + 4.f * (t[1] - t[2]);
    dst[fh * 6 + 3] = e + 2.f * o;
    dst[fh * 6 + 4] = e - 2.f * o;
    dst[fh * 6 + 5] = 4.f * t[1] - 5.f * t[3] + t[5];
  };
  auto mma = [&](int mt, int fh, const float (&av)[NF], const float (&bw)[NF]) {
#pragma unroll
    for (int fw = 0; fw < 6; ++fw)
      acc[mt][fh * 6 + fw] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fh * 6 + fw], bw[fh * 6 + fw], acc[mt][fh * 6 + fw], 0, 0, 0);
  };
  // one (kd, k-step): the 24 MFMAs of step s on (av, bw), interleaved with the transform of step s + 1's window into an
  // (its LDS reads and the U fragments of step s + 1 are requested first) -- the wave's own VALU work rides in the shadow
  // of its MFMAs instead of waiting for the other wave of the SIMD to be out of phase
  auto step = [&](int s, int c0, const float (&av)[MT][NF], const float (&bw)[NF], float (&an)[MT][NF], float (&bnx)[NF]) {
    const bool more = s + 1 < 12;
    if (RABL & 1) {}
    else if (more) load_b(bnx, c0, (s + 1) / 4, (s + 1) % 4);
    else if (c0 + CC < p.Cin) load_b(bnx, c0 + CC, 0, 0);
    if (more && !(RABL & 2)) read_d((s + 1) / 4, (s + 1) % 4);
    __builtin_amdgcn_sched_barrier(0);                   // the requests stay up here: the scheduler would sink them next to their uses
#pragma unroll
    for (int fh = 0; fh < 4; ++fh)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        mma(mt, fh, av[mt], bw);
        if (more && !(RABL & 4)) xform(mt, fh, an[mt]);
      }
  };

  constexpr int Q = CC / 4;
  // Halo staging by ownership, as in conv3d_wino2d_kernel: a thread owns one (column, channel quad) of the halo rows rg, rg + 6, ...
  // (HDr * HH rows of HW * Q = 40 lanes; 240 of the 256 threads).  The validity bits of its pieces are formed once per tile; a
  // piece's offsets are 24-bit multiply-adds from scalars.  (A flat piece index with three divisions, three bounds tests and two
  // product chains per piece, once for the load and once for the LDS store, was ~880 of this kernel's ~3,000 vector
  // instructions per tile and wave.)
  constexpr int ROWS_H = HDr * HH, NIT = (ROWS_H + 5) / 6;
  static_assert(NIT <= 32, "one validity bit per piece");
  unsigned s_ok = 0;
  {
    const int lane40 = tid % (HW * Q), rg = tid / (HW * Q), hw = lane40 / Q;
    const bool col_ok = rg < 6 && (unsigned)(iw0 + hw) < (unsigned)p.W;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int row = rg + 6 * j, hd = row / HH, hh = row - hd * HH;
      const bool ok = col_ok && row < ROWS_H && (unsigned)(id0 + hd) < (unsigned)p.D && (unsigned)(ih0 + hh) < (unsigned)p.H;
      s_ok |= ok ? (1u << j) : 0u;
    }
  }
  const int s_plane = p.H * p.W * p.Cin, s_rowp = p.W * p.Cin;               // (scalars; the halo spans < 2^24 elements)
  load_b(b0, 0, 0, 0);
  for (int c0 = 0; c0 < p.Cin; c0 += CC) {
    __syncthreads();
    {
      // four pieces at a time (16 in flight would spill); the thread's coordinates are re-derived from an opaque copy of tid (kept
      // across the MFMA steps they would cost registers this kernel does not have)
      int tv = tid;
      asm volatile("" : "+v"(tv));
      const int rg = tv / (HW * Q), lane40 = tv - rg * (HW * Q), hw = lane40 / Q, q = lane40 - hw * Q;
      const unsigned col = __umul24(hw, p.Cin) + q * 4;
      float *lcol = lds + hw * ROW + q * 4;
#pragma unroll 1
      for (int j0 = 0; j0 < NIT; j0 += 4) {
        float4 stage[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int row = rg + 6 * (j0 + jj);
          const int hd = (int)(__umul24(row, 205) >> 11), hh = row - hd * HH;          // row / 10 for row < 1029
          // (outside the volume: an offset beyond the descriptor's range reads zeros -- no branch around the load)
          stage[jj] = buffer_load4(xrsrc, ((s_ok >> (j0 + jj)) & 1u) ? (__umul24(hd, s_plane) + __umul24(hh, s_rowp) + col) * 4u : 0xfffffff0u, (unsigned)c0 * 4u);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int row = rg + 6 * (j0 + jj);
          if (rg < 6 && row < ROWS_H) {
            const int hd = (int)(__umul24(row, 205) >> 11), hh = row - hd * HH;
            float *dd = lcol + hd * PL + hh * (HWR * ROW);
            dd[0] = stage[jj].x; dd[1] = stage[jj].y; dd[2] = stage[jj].z; dd[3] = stage[jj].w;
          }
        }
      }
    }
    __syncthreads();
    read_d(0, 0);                                         // the chunk's first window: exposed once per chunk
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int fh = 0; fh < 4; ++fh) xform(mt, fh, a0[mt]);
    // 12 (kd, k-step) steps per chunk; after an even number of steps the live buffers are (a0, b0) again
#pragma unroll
    for (int s = 0; s < 12; s += 2) {
      step(s, c0, a0, b0, a1, b1);
      step(s + 1, c0, a1, b1, a0, b0);
    }
  }

  // ---- epilogue: Y = Ah^T M Aw per accumulator element, then shift / adds / upsample add / ReLU / mask; NDHWC store
  const float sh = p.shift ? p.shift[co] : 0.f;
  // Output addressing: a SCALAR 64-bit index of the tile's first output + 32-bit lane offsets (one per accumulator row, the row's
  // eight outputs are scalar steps away).  (One size_t product chain per output was ~1,600 vector instructions per tile, 300 of them 16-cycle multiplies.)
  const size_t ybase = ((((size_t)n * p.yD + (size_t)od0 * p.ysd) * p.yH + (size_t)oh0 * p.ysh) * p.yW + (size_t)ow0 * p.ysw) * p.Cout;
  const float *add1b = p.add1 ? p.add1 + ybase : nullptr, *add2b = p.add2 ? p.add2 + ybase : nullptr, *maskb = p.mask ? p.mask + ybase : nullptr;
  float *yb = p.y + ybase;
  const int xstep = p.ysw * p.Cout * 4, rowstep = p.ysh * p.yW * p.Cout * 4, planestep = p.ysd * p.yH * p.yW * p.Cout * 4;   // bytes
  auto emit = [&](unsigned rel, int od, int oh, int ow, float val) {
    if (od < p.Do && oh < p.Ho && ow < p.Wo) {
      float v = val + sh;
      if (add1b) v += *reinterpret_cast<const float *>(reinterpret_cast<const char *>(add1b) + rel);
      if (add2b) v += *reinterpret_cast<const float *>(reinterpret_cast<const char *>(add2b) + rel);
      if (p.up) {
        auto nearest = [](int dst, int in, int out) {          // torch nearest: min(floor(dst * (in/out)), in-1)
          const int s_ = (int)floorf((float)dst * ((float)in / (float)out));
          return s_ < in - 1 ? s_ : in - 1;
        };
        const int ud = nearest(od, p.Du, p.Do), uh = nearest(oh, p.Hu, p.Ho), uw = nearest(ow, p.Wu, p.Wo);
        v += p.up[((((size_t)n * p.Du + ud) * p.Hu + uh) * p.Wu + uw) * p.Cout + co];
      }
      if (p.relu) v = fmaxf(v, 0.f);
      if (maskb) v = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(maskb) + rel) > 0.f ? v : 0.f;
      *reinterpret_cast<float *>(reinterpret_cast<char *>(yb) + rel) = v;
    }
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lk * 4 + r;
    const int dz = 2 * wave_m + (row >> 3), dh = ((row >> 1) & 3) * 2, dw = (row & 1) * 4;
    const int od = od0 + dz, oh = oh0 + dh, ow = ow0 + dw;
    float z[4][4];                                       // along W: z[fh][x]
#pragma unroll
    for (int fh = 0; fh < 4; ++fh) {
      const float m0 = acc[mt][fh * 6][r], m1 = acc[mt][fh * 6 + 1][r], m2 = acc[mt][fh * 6 + 2][r], m3 = acc[mt][fh * 6 + 3][r],
                  m4 = acc[mt][fh * 6 + 4][r], m5 = acc[mt][fh * 6 + 5][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      z[fh][0] = (m0 + s12) + s34; z[fh][1] = d12 + 2.f * d34; z[fh][2] = s12 + 4.f * s34; z[fh][3] = (d12 + 8.f * d34) + m5;
    }
    const unsigned rel0 = (unsigned)(dz * planestep + dh * rowstep + dw * xstep + co * 4);        // byte offset inside the tile
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      emit(rel0 + x * xstep, od, oh, ow + x, (z[0][x] + z[1][x]) + z[2][x]);
      emit(rel0 + x * xstep + rowstep, od, oh + 1, ow + x, (z[1][x] - z[2][x]) - z[3][x]);
    }
  }
}

// ---- wide form of the 64-wide kernel (round 5).  conv3d_wino2d_kernel streams the whole transformed-weight tensor from L2 for every
// 16 GEMM rows (6 GB per P2 launch, ~10.8 TB/s) and reads one A fragment from LDS per MFMA: a wave owns ALL 24 products of its 16 x 16
// tile, so nothing it loads is used twice.  Here a wave owns ONE fh row of the products (6 of the 24) for 2 row tiles x 2 Cout tiles
// (the same 96 accumulator registers): every A fragment feeds two MFMAs and every U fragment feeds two -- half the L2 stream and half
// the LDS reads per MFMA.  Eight waves = 4 fh rows x 2 Cout halves share the V of 32 GEMM rows (a 4 x 8 x 8 output tile: 6 halo planes
// for 4 output planes instead of 4 for 2).  The fh rows are summed in the epilogue (Y = Ah^T (M Aw): each wave applies Aw to its own
// row, the four rows meet in LDS).  One workgroup per CU, so the phases are software-pipelined INSIDE the workgroup, across chunks
// and across tiles: V and the halo are double-buffered (Cin chunks of 8), the halo of chunk c + 2 travels global -> LDS while the V
// of chunk c + 1 is formed between the MFMA steps of chunk c; a workgroup walks a sequence of tiles, and what lies behind a tile's last chunk is the next
// tile's first.  Two barriers per chunk, both between MFMA steps whose operands are already in registers.
// U layout for this form: [kd][ci][fh][co][6] (a lane's six fw products of its fh row in 24 bytes: two 12-byte loads per Cout tile).
constexpr int kThreadsW = 512;
#ifndef MR3D_WIDE_ABL
#define MR3D_WIDE_ABL 0
#endif
constexpr int WABL = MR3D_WIDE_ABL;                    // timing-only ablations (tools/diag/wide_ablate_build.sh); 0 in the library
#ifdef MR3D_WIDE_STAMPS                                // diagnostic builds: wall-clock stamps of workgroup phases (tools/diag/wide_stamps.py)
__device__ unsigned long long g_wide_stamps[1024 * 8 * 16];
#define WIDE_MARK(k) do { if (stamp_tile) stamp[k] = wall_clock64(); } while (0)
#else
#define WIDE_MARK(k) do { } while (0)
#endif
constexpr int WTD = 4, WHD = WTD + 2;                  // output planes of a tile, halo planes
constexpr int WC = 8;                                  // Cin chunk
// halo: 30 row pairs (6 planes x 10 rows) of 2 x 10 positions x 8 channels, one 264-float slot each, two buffers.  A slot is what ONE
// `buffer_load_dwordx4 ... lds` writes (global -> LDS without a register stop: lane l's 16 bytes land at M0 + 16 l; lanes 0..39 carry the
// 40 pieces of the row pair, the others -- and every piece outside the volume -- an offset beyond the descriptor's range, which writes
// zeros: tools/micro/lds_dma_probe.hip).  264 = 8 mod 64 puts the 64 lanes of a V-unit read (8 patches x 8 channels: patch row pair
// -> one slot = 8 banks, patch column quad -> 32 floats, channel -> 1) on 64 different banks.
constexpr int WRP = 264;
constexpr int WHBUF = 30 * WRP;                        // 7920 floats per halo buffer
// V [(plane, patch)][fh][fw pair][ci][2] + 4: a lane's A fragments of two adjacent products are ONE ds_read_b64, and 196 = 4 mod 64
// puts the 32 lanes of a read pass (16 rows x 2 k) on 64 different banks
constexpr int WVP = 4 * 3 * WC * 2 + 4;                // 196
constexpr int WV = WHD * 8 * WVP;                      // 9408 floats
// epilogue exchange of ONE row tile, Z [fh][row 0..15][x 0..3][co 0..63], row pitch 260 floats (4 rows further = 16 banks further): it
// overlays V1 and halo buffer 0, which are free behind a tile's last chunk, while V0 already holds the next tile's first chunk and
// halo buffer 1 receives its second
constexpr int WZP = 4 * 64 + 4;
constexpr int WZ = 4 * 16 * WZP;                       // 16,640 floats
static_assert(WZ <= WV + WHBUF, "one half of the epilogue exchange fits V1 | halo 0");
constexpr int WLDS_BYTES = (2 * WV + 2 * WHBUF) * 4;   // 138,624

template <bool LIVE>
__global__ __launch_bounds__(kThreadsW, 2) void conv3d_wino2d_wide_kernel(Conv3dFwdArgs p, const float *__restrict__ U, int tiles_d,
                                                                         int tiles_h, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *halo = lds + 2 * WV;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fh = wave & 3, ch = wave >> 2;               // this wave's row of products and its half of the 64-wide Cout tile
  // ---- the tiles of this workgroup.  Workgroup b runs on XCD b & 7 (dispatch order): an XCD takes one contiguous eighth of the
  // tiles and its workgroups walk it side by side, so that tiles with common halo planes meet in one L2.
  const int T = p.N * tiles_d * tiles_h * tiles_w;
  int t_cur, t_step, t_end;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, per = (T + 7) >> 3;
    t_cur = xcd * per + (int)(blockIdx.x >> 3); t_step = (int)(gridDim.x >> 3); t_end = min(T, (xcd + 1) * per);
  } else {
    t_cur = blockIdx.x; t_step = gridDim.x; t_end = T;
  }
  if (t_cur >= t_end) return;
#ifdef MR3D_WIDE_STAMPS
  const unsigned long long stamp0 = wall_clock64();
#endif
  struct Tile { int n, od0, oh0, ow0; };
  auto tile_at = [&](int t) {
    Tile r;
    r.ow0 = (t % tiles_w) * TW; t /= tiles_w;
    r.oh0 = (t % tiles_h) * TH; t /= tiles_h;
    r.od0 = (t % tiles_d) * WTD; r.n = t / tiles_d;
    return r;
  };
  const int li = lane & 15, lk = lane >> 4;
  const int cobase = blockIdx.y * 64;
  // A operand: GEMM row li of row tile rt = (output plane 2 rt + (li >> 3), patch li & 7); k index lk of the k-step
  const int vlane = li * WVP + fh * (3 * WC * 2) + lk * 2;                                 // floats, inside a V buffer
  const unsigned ul_off = (unsigned)((lk * 4 * p.Cout + cobase + ch * 32 + li) * 6 * 4);   // bytes: (ci = lk, fh = 0, this lane's co)
  const __amdgpu_buffer_rsrc_t ursrc = buffer_of(U);

#ifdef MR3D_WIDE_STAMPS
  unsigned long long stamp[16] = {};
  bool stamp_tile = false;
  int tiles_done = 0;
  stamp[0] = stamp0;
#endif
  f32x4 acc[24];                                         // [fw][rt][ct]
#pragma unroll
  for (int f = 0; f < 24; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float bq[4][12], aq[2][6];                             // U fragments [ct][fw] (four steps in flight), A fragments [rt][fw] (half a step ahead)
  auto load_b = [&](float (&dst)[12], int c0, int kd, int kk) {
    const unsigned src = (unsigned)((((kd * p.Cin + c0 + kk * 4) * 4 + fh) * p.Cout) * 6 * 4);       // scalar byte offset
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const buf_f32x3 v = buffer_load3(ursrc, ul_off + (unsigned)(ct * 16 * 24 + hf * 12), src);
        dst[ct * 6 + hf * 3] = v.x; dst[ct * 6 + hf * 3 + 1] = v.y; dst[ct * 6 + hf * 3 + 2] = v.z;
      }
  };
  auto load_a = [&](float (&dst)[6], const float *vbuf, int rt, int kd, int kk) {
#pragma unroll
    for (int fp = 0; fp < 3; ++fp) {
      const float2 v = *reinterpret_cast<const float2 *>(vbuf + vlane + (2 * rt + kd) * 8 * WVP + fp * (WC * 2) + kk * 8);
      dst[fp * 2] = v.x; dst[fp * 2 + 1] = v.y;
    }
  };
  // halo staging, global -> LDS: wave w requests the row pairs w, w + 8, w + 16, w + 24 (< 30) of a chunk, one instruction each.  A
  // piece's byte offset in x is formed once per tile; outside the volume (and in lanes 40..63) it lies beyond the descriptor's range.
  // The requests are inline assembly: the compiler's wait-count pass makes every later LDS read wait for an LDS-writing load it knows
  // of, and these take an HBM round trip.  It does not count them, so its own vmcnt waits (for the U fragments) are at worst four
  // entries stricter than needed; the wait for the halo itself is counted by hand (below).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 xdesc = {(unsigned)(uintptr_t)p.x, (unsigned)((uintptr_t)p.x >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
  const unsigned lds_base = (unsigned)(uintptr_t)lds;    // LDS byte address of the dynamic array (the low half of its flat address)
  auto halo_offsets = [&](const Tile &t, unsigned (&rel)[4]) {
    const int id0 = t.od0 - p.pd, ih0 = t.oh0 - p.ph, iw0 = t.ow0 - p.pw;
    const int origin = ((t.n * p.D + id0) * p.H + ih0) * p.W + iw0;                        // position index of the halo's corner (may be negative)
    int lv = lane;                                       // (an opaque copy: the decomposition below is redone per call -- three times per tile --
    asm volatile("" : "+v"(lv));                         //  instead of being kept in registers over the chunk loop, which has none to spare)
    const int r2 = lv >= 20 ? 1 : 0, piece = lv - 20 * r2, hw = piece >> 1, q = piece & 1;      // lanes 0..39: row of the pair, piece of the row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 2 * (wave + 8 * i) + r2;           // (wave + 8 i < 30: the caller skips the others)
      const int hd = (row * 205) >> 11, hh = row - hd * HH;                                // row / 10 for row < 64
      const bool ok = lv < 40 && (unsigned)(id0 + hd) < (unsigned)p.D && (unsigned)(ih0 + hh) < (unsigned)p.H && (unsigned)(iw0 + hw) < (unsigned)p.W;
      rel[i] = ok ? (unsigned)((origin + (hd * p.H + hh) * p.W + hw) * p.Cin + q * 4) * 4u : 0xfffffff0u;
    }
  };
  auto halo_request = [&](const unsigned (&rel)[4], int c0, int buf) {          // chunk c0 .. c0 + 7 of a tile into halo buffer buf
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < 3 || wave < 6) {                           // (wave-uniform; row pairs 30, 31 do not exist)
        const unsigned m0 = lds_base + (unsigned)((2 * WV + buf * WHBUF + (wave + 8 * i) * WRP) * 4);
        // (m0 is written without being declared: hipcc rejects it as a clobber ("reserved register").  Nothing else in this kernel
        //  reads or writes m0 -- no ds_*_addtid, no s_movrel, no readlane / sendmsg through it -- which is what makes that safe.)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     : : "s"(m0), "v"(rel[i]), "s"(xdesc), "s"((unsigned)c0 * 4u) : "memory");
      }
  };
  // V = Bh^T d Bw of a chunk: 6 planes x 8 patches x 8 channels = 384 windows.  A wave only ever multiplies the fh row of V it owns,
  // and that row needs two rows of a window (d0 - d2, d1 + d2, d2 - d1, d1 - d3): wave (fh, ch) forms row fh of the planes 3 ch ..
  // 3 ch + 2, one plane = one unit of 12 LDS reads, 21 vector instructions and 3 LDS writes per lane (lane = patch x channel) -- three
  // equal units per wave and chunk.  A unit's reads are requested in FRONT of an MFMA step, its arithmetic and writes follow the
  // step (both waves of a SIMD run the same phase at the same time: a read waited for behind the MFMAs is latency nobody covers).
  // (The first version formed whole windows on six of the eight waves: the two SIMDs with two such waves were the critical path.)
  const int vf_ra = fh == 0 ? 0 : (fh == 2 ? 2 : 1), vf_rb = fh == 3 ? 3 : (fh == 2 ? 1 : 2);
  const float vf_sg = fh == 1 ? 1.f : -1.f;
  // halo row r = 10 plane + hh sits in slot r >> 1 at (r & 1) * 80 floats; a patch's window starts on an even row
  const int vf_src = (3 * ch * 5 + ((lane >> 3) >> 1)) * WRP + ((lane >> 3) & 1) * 32 + (lane & 7);
  const int vf_dst = (3 * ch * 8 + (lane >> 3)) * WVP + fh * (3 * WC * 2) + (lane & 7) * 2;
  float vd[12];
  auto vunit_read = [&](const float *hbuf, int k) {
    const float *sa = hbuf + vf_src + (k * 5 + (vf_ra >> 1)) * WRP + (vf_ra & 1) * 80, *sb = hbuf + vf_src + (k * 5 + (vf_rb >> 1)) * WRP + (vf_rb & 1) * 80;
#pragma unroll
    for (int j = 0; j < 6; ++j) { vd[j] = sa[j * 8]; vd[6 + j] = sb[j * 8]; }
  };
  auto vunit_finish = [&](float *vdst, int k) {
    float t[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) t[j] = fmaf(vf_sg, vd[6 + j], vd[j]);
    const float e = t[4] - t[2], o = t[3] - t[1];
    float2 *dd = reinterpret_cast<float2 *>(vdst + vf_dst + k * 8 * WVP);
    dd[0] = make_float2(4.f * t[0] - 5.f * t[2] + t[4], (t[3] + t[4]) - 4.f * (t[1] + t[2]));
    dd[WC] = make_float2((t[4] - t[3]) + 4.f * (t[1] - t[2]), e + 2.f * o);
    dd[2 * WC] = make_float2(e - 2.f * o, 4.f * t[1] - 5.f * t[3] + t[5]);
  };
  // one (kd, k-step) s = 0..5 of a chunk (par: its parity -- chunks are taken in pairs, 12 steps, so that the four U buffers rotate
  // with compile-time indices): 24 MFMAs on aq[s & 1] and bq[(6 par + s) % 4].  In front of them the requests: the A fragments of
  // step s + 1 (LDS; behind step 5: step 0 of the next chunk) and the U fragments of step s + 3 (L2; behind the last chunk: chunk 0
  // again -- the next tile's).  Three steps, not two: the older wave of a SIMD wins every MFMA issue slot, so the two waves run their
  // steps one after the other rather than interleaved, and the one that runs alone covers two steps in 0.7 us -- less than an L2
  // round trip under this load.
  auto step = [&](int par, int s, int c0, const float *vcur, const float *vnext, int unit = -1) {
    const int s12 = par * 6 + s, s3 = s + 3;
    if (unit >= 0 && !(WABL & 1)) vunit_read(halo + (1 - par) * WHBUF, unit);      // the next chunk's halo
    if (WABL & 4) {}
    else if (s3 < 6) load_b(bq[(s12 + 3) % 4], c0, s3 / 2, s3 % 2);
    else load_b(bq[(s12 + 3) % 4], c0 + WC < p.Cin ? c0 + WC : 0, (s3 - 6) / 2, (s3 - 6) % 2);
    // A fragments half a step ahead: row tile 1's (this step) while row tile 0's MFMAs run, then row tile 0's of the next step
    if (!(WABL & 32)) load_a(aq[1], vcur, 1, s / 2, s % 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int fw = 0; fw < 6; ++fw)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        acc[(fw * 2 + 0) * 2 + ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[0][fw], bq[s12 % 4][ct * 6 + fw], acc[(fw * 2 + 0) * 2 + ct], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (WABL & 32) {}
    else if (s + 1 < 6) load_a(aq[0], vcur, 0, (s + 1) / 2, (s + 1) % 2);
    else load_a(aq[0], vnext, 0, 0, 0);                  // step 0 of the next chunk (its V is complete: barrier 2 sits in front of step 5)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int fw = 0; fw < 6; ++fw)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        acc[(fw * 2 + 1) * 2 + ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[1][fw], bq[s12 % 4][ct * 6 + fw], acc[(fw * 2 + 1) * 2 + ct], 0, 0, 0);
  };

  // ---- epilogue of one tile, a row tile (2 output planes) at a time: z = (M Aw) of this wave's fh row into Z, then per output
  // Y = Ah^T z over the four rows + the fused tail, 16-byte stores.  Ends behind a barrier (the halo region is free again).
  const int xstep = p.ysw * p.Cout * 4, rowstep = p.ysh * p.yW * p.Cout * 4, planestep = p.ysd * p.yH * p.yW * p.Cout * 4;   // bytes
  auto epilogue = [&](const Tile &t, bool zero) {
    float *Z = lds + WV;
    int tv = tid;                                        // (an opaque copy: what is derived from it stays inside the epilogue instead of
    asm volatile("" : "+v"(tv));                         //  being hoisted over the chunk loop, which has no registers to spare)
    const int q = tv & 15, row = (tv >> 4) & 15, xp = tv >> 8;         // thread: GEMM row of the row tile x four output channels x two of the four columns
    const int co = cobase + q * 4;
    const float4 sh4 = p.shift ? *reinterpret_cast<const float4 *>(p.shift + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int dh = ((row >> 1) & 3) * 2, dw = (row & 1) * 4 + xp * 2;
    const size_t ybase = ((((size_t)t.n * p.yD + (size_t)t.od0 * p.ysd) * p.yH + (size_t)t.oh0 * p.ysh) * p.yW + (size_t)t.ow0 * p.ysw) * p.Cout;
    const char *add1b = p.add1 ? reinterpret_cast<const char *>(p.add1 + ybase) : nullptr;
    const char *add2b = p.add2 ? reinterpret_cast<const char *>(p.add2 + ybase) : nullptr;
    const char *maskb = p.mask ? reinterpret_cast<const char *>(p.mask + ybase) : nullptr;
    char *yb = reinterpret_cast<char *>(p.y + ybase);
    // Z overlays V1: step 5 of the tile's last chunk still reads its A fragments out of V1 (load_a(aq[1], vcur = V1) behind barrier 2),
    // and nothing but the ~48 MFMAs in between kept a fast wave's first Z store behind a slow wave's read.  One barrier per tile.
    if (!zero) __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      if (!zero) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float m0 = acc[(0 * 2 + rt) * 2 + ct][r], m1 = acc[(1 * 2 + rt) * 2 + ct][r], m2 = acc[(2 * 2 + rt) * 2 + ct][r],
                        m3 = acc[(3 * 2 + rt) * 2 + ct][r], m4 = acc[(4 * 2 + rt) * 2 + ct][r], m5 = acc[(5 * 2 + rt) * 2 + ct][r];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            float *zz = Z + (fh * 16 + lk * 4 + r) * WZP + ch * 32 + ct * 16 + li;
            zz[0] = (m0 + s12) + s34; zz[64] = d12 + 2.f * d34; zz[128] = s12 + 4.f * s34; zz[192] = (d12 + 8.f * d34) + m5;
          }
        __syncthreads();
      }
      const int od = t.od0 + 2 * rt + (row >> 3);
      auto emit = [&](unsigned rel, int oh, int ow, float4 v) {
        if (od < p.Do && oh < p.Ho && ow < p.Wo) {
          v.x += sh4.x; v.y += sh4.y; v.z += sh4.z; v.w += sh4.w;
          if (add1b) { const float4 a = *reinterpret_cast<const float4 *>(add1b + rel); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
          if (add2b) { const float4 a = *reinterpret_cast<const float4 *>(add2b + rel); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
          if (p.up) {
            auto nearest = [](int dst, int in, int out) {          // torch nearest: min(floor(dst * (in/out)), in-1)
              const int s_ = (int)floorf((float)dst * ((float)in / (float)out));
              return s_ < in - 1 ? s_ : in - 1;
            };
            const int ud = nearest(od, p.Du, p.Do), uh = nearest(oh, p.Hu, p.Ho), uw = nearest(ow, p.Wu, p.Wo);
            const float4 a = *reinterpret_cast<const float4 *>(p.up + ((((size_t)t.n * p.Du + ud) * p.Hu + uh) * p.Wu + uw) * p.Cout + co);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (maskb) {
            const float4 m = *reinterpret_cast<const float4 *>(maskb + rel);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
          }
          *reinterpret_cast<float4 *>(yb + rel) = v;
        }
      };
      const unsigned rel0 = (unsigned)((2 * rt + (row >> 3)) * planestep + dh * rowstep + dw * xstep + co * 4);
#pragma unroll
      for (int xi = 0; xi < 2; ++xi) {
        float4 z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          z[g] = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(Z + (g * 16 + row) * WZP + (xp * 2 + xi) * 64 + q * 4);
        const float4 y0 = make_float4((z[0].x + z[1].x) + z[2].x, (z[0].y + z[1].y) + z[2].y, (z[0].z + z[1].z) + z[2].z, (z[0].w + z[1].w) + z[2].w);
        const float4 y1 = make_float4((z[1].x - z[2].x) - z[3].x, (z[1].y - z[2].y) - z[3].y, (z[1].z - z[2].z) - z[3].z, (z[1].w - z[2].w) - z[3].w);
        emit(rel0 + xi * xstep, t.oh0 + dh, t.ow0 + dw + xi, y0);
        emit(rel0 + xi * xstep + rowstep, t.oh0 + dh + 1, t.ow0 + dw + xi, y1);
      }
      if (!zero) __syncthreads();
    }
  };

  Tile tile = tile_at(t_cur);
  if constexpr (LIVE) {                                  // one tile per workgroup (mr3d_conv3d_wino2d launches this instance so)
    // lane r < 60 tests the 10 voxels of halo row (plane r / 10, row r % 10); all clear = nothing to multiply
    const int id0 = tile.od0 - p.pd, ih0 = tile.oh0 - p.ph, iw0 = tile.ow0 - p.pw;
    const int hd = lane / HH, hh = lane - hd * HH;
    const int id = id0 + hd, ih = ih0 + hh;
    unsigned bits = 0;
    if (lane < WHD * HH && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H) {
      const int w_lo = max(iw0, 0), w_hi = min(iw0 + HW, p.W);
      const long long b0 = (((long long)tile.n * p.D + id) * p.H + ih) * p.W + w_lo;
      const unsigned *wd = reinterpret_cast<const unsigned *>(p.x_live) + (b0 >> 5);
      const int sh = (int)(b0 & 31), cnt = w_hi - w_lo;
      const unsigned long long two = (unsigned long long)wd[0] | ((unsigned long long)(sh + cnt > 32 ? wd[1] : 0u) << 32);
      bits = (unsigned)(two >> sh) & ((1u << cnt) - 1u);
    }
    if (__ballot(bits != 0) == 0) {                      // (every wave tests the same rows: uniform over the workgroup)
      epilogue(tile, true);
      return;
    }
  }
  const int nc = p.Cin / WC;                             // even (Cin is a multiple of 16): a tile starts on V0 and ends on V1
  unsigned rel_c[4];                                     // halo piece offsets of this tile (the next tile's are formed where they are used)
  halo_offsets(tile, rel_c);
  int t_next = t_cur + t_step < t_end ? t_cur + t_step : -1;
  Tile tnext = tile_at(t_next < 0 ? t_cur : t_next);
  // pipeline fill: chunk 0 of the first tile
  load_b(bq[0], 0, 0, 0);
  load_b(bq[1], 0, 0, 1);
  load_b(bq[2], 0, 1, 0);
  halo_request(rel_c, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) { vunit_read(halo, k); vunit_finish(lds, k); }
  halo_request(rel_c, WC, 1);
  __syncthreads();
  load_a(aq[0], lds, 0, 0, 0);
  bool filling = true;                                   // the very first chunk: fewer requests lie between its halo and barrier 1
  // The loop body carries no branch around a compiler-visible memory request: behind the last tile's last chunk the same tile is requested again and dropped.
  auto chunk = [&](int par, int c) {                     // V[par] -> MFMAs; V[1 - par] <- halo[1 - par] (the next chunk); halo[par] <- the chunk after it
    const float *vcur = lds + par * WV;
    float *vnext = lds + (1 - par) * WV;
    const int c0 = c * WC;
    if (c == 3) WIDE_MARK(2);
    step(par, 0, c0, vcur, vnext);
    step(par, 1, c0, vcur, vnext);
    if (c == 3) WIDE_MARK(3);
    // the next chunk's halo was requested one chunk ago; the requests behind it are the U fragments of six steps (24; sched_barriers
    // pin each step's four in front of its MFMAs) -- two steps in the first chunk
    // (the counts are vector-memory INSTRUCTIONS: kLoadsPerStep buffer loads per load_b -- 2 column tiles x 2 halves of 12 bytes -- times
    //  the steps requested behind the halo; a load_b that issues another number of loads needs other constants here)
    constexpr int kLoadsPerStep = 4;
    static_assert(2 * kLoadsPerStep == 8 && 6 * kLoadsPerStep == 24, "hand-counted vmcnt waits below");
    if (filling) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    filling = false;
    if (c == 3) WIDE_MARK(4);
    if (!(WABL & 16)) __syncthreads();                   // 1: the next chunk's halo is complete; halo[par] (read one chunk ago) is free
    if (c == 3) WIDE_MARK(5);
    if (!(WABL & 2)) {                                   // the chunk after the next: of this tile, or chunk 0 / 1 of the next tile
      const bool wrap = c + 2 >= nc;
      unsigned r[4] = {rel_c[0], rel_c[1], rel_c[2], rel_c[3]};
      if (wrap) halo_offsets(tnext, r);
      halo_request(r, (wrap ? c + 2 - nc : c + 2) * WC, par);
    }
    step(par, 2, c0, vcur, vnext, 0);
    if (!(WABL & 1)) vunit_finish(vnext, 0);
    step(par, 3, c0, vcur, vnext, 1);
    if (!(WABL & 1)) vunit_finish(vnext, 1);
    step(par, 4, c0, vcur, vnext, 2);
    if (!(WABL & 1)) vunit_finish(vnext, 2);
    if (c == 3) WIDE_MARK(6);
    if (!(WABL & 16)) __syncthreads();                   // 2: V of the next chunk is complete; V of this chunk (behind step 4's reads) is free
    if (c == 3) WIDE_MARK(7);
    step(par, 5, c0, vcur, vnext);
    if (c == 3) WIDE_MARK(8);
  };
  for (;;) {
#ifdef MR3D_WIDE_STAMPS
    stamp_tile = tiles_done == 2;                        // the third tile of the walk
#endif
    WIDE_MARK(1);
    for (int c = 0; c < nc; c += 2) {
      chunk(0, c);
      chunk(1, c + 1);
    }
    WIDE_MARK(9);
    if (!(WABL & 8)) epilogue(tile, false);              // Z overlays V1 | halo 0; V0 holds the next tile's first chunk, halo 1 receives its second
    WIDE_MARK(10);
#ifdef MR3D_WIDE_STAMPS
    ++tiles_done;
#endif
#pragma unroll
    for (int f = 0; f < 24; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t_next < 0) break;
    t_cur = t_next; tile = tnext;
    halo_offsets(tile, rel_c);
    t_next = t_cur + t_step < t_end ? t_cur + t_step : -1;
    tnext = tile_at(t_next < 0 ? t_cur : t_next);
  }
#ifdef MR3D_WIDE_STAMPS
  stamp[11] = wall_clock64();
  if (lane == 0 && blockIdx.x < 1024) {
#pragma unroll
    for (int k = 0; k < 16; ++k) g_wide_stamps[((size_t)blockIdx.x * 8 + wave) * 16 + k] = stamp[k];
  }
#endif
}

// ---- batched form of wino2d_weights_kernel for a plan run: one launch transforms the weights of every convolution of the
// plan that will take these kernels (16 launches of ~33 us each sat between the convolutions of a step: 0.5 ms of 11.8)
constexpr int kPrepMax = 32;
struct PrepTable { const float *w[kPrepMax]; float *U[kPrepMax], *Uw[kPrepMax]; int cin[kPrepMax], cout[kPrepMax]; };
__global__ __launch_bounds__(256) void wino2d_weights_batched_kernel(PrepTable t) {
  const int k = blockIdx.y, cin = t.cin[k], cout = t.cout[k];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 3 * cin * cout) return;
  const float *w = t.w[k];
  float *U = t.U[k], *Uw = t.Uw[k];
  const int kd = e / (cin * cout), cc = e - kd * (cin * cout);
  float g[3][3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) g[kh][kw] = w[(size_t)((kd * 3 + kh) * 3 + kw) * cin * cout + cc];
  float tt[3][6];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const float g0 = g[kh][0], g1 = g[kh][1], g2 = g[kh][2];
    const float s02 = g0 + g2, h = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), q = g1 * (1.f / 12.f);
    tt[kh][0] = 0.25f * g0; tt[kh][1] = (s02 + g1) * (-1.f / 6.f); tt[kh][2] = (s02 - g1) * (-1.f / 6.f);
    tt[kh][3] = h + q; tt[kh][4] = h - q; tt[kh][5] = g2;
  }
#pragma unroll
  for (int fw = 0; fw < 6; ++fw) {
    const float a = tt[0][fw], b = tt[1][fw], c = tt[2][fw];
    const float u[4] = {a, 0.5f * ((a + c) + b), 0.5f * ((a + c) - b), c};
#pragma unroll
    for (int fh = 0; fh < 4; ++fh) {
      const int f = fh * 6 + fw, ci = cc / cout, co = cc - ci * cout;
      U[((((size_t)kd * cin + ci) * (NF / 4) + f / 4) * cout + co) * 4 + (f & 3)] = u[fh];
      if (Uw) Uw[((((size_t)kd * cin + ci) * 4 + fh) * cout + co) * 6 + fw] = u[fh];
    }
  }
}

// weights already transformed for the plan run in progress on this thread: packed-weight pointer -> U
thread_local std::map<const float *, const float *> g_prepared;

// one layout of U; a 64-wide Cout keeps both (the second for conv3d_wino2d_wide_kernel: which kernel a launch takes depends on its map,
// and the RPN conv's weights serve every level)
size_t u_layout_floats(int cin, int cout) { return ((size_t)3 * NF * cin * cout + 63) / 64 * 64; }
bool u_has_wide(int cout) { return cout % 64 == 0; }
size_t u_floats(int cin, int cout) { return u_layout_floats(cin, cout) * (u_has_wide(cout) ? 2 : 1); }

// Where the transformed weights of a launch come from: the plan run in progress prepared them (WINO2D_PREP), or the
// caller lent a scratch buffer with the call; nothing is allocated here.
bool u_available(const Conv3dFwdArgs &a) {
  if (a.ws_bytes == -1) return true;                                   // variant query: assume the scratch
  if (g_prepared.find(a.w) != g_prepared.end()) return true;
  return a.ws && a.ws_bytes >= (long long)(u_floats(a.Cin, a.Cout) * sizeof(float));
}

}  // namespace

// narrow Cout tiles: Cout 16 or 32 exactly (a wider Cout takes the 64-wide kernel or the one-dimensional forms)
bool mr3d_conv3d_wino2d_rows_ok(const Conv3dFwdArgs &a) {
  // (the even-position add of MR3D_CONV_UP_EVEN exists in the generic and the streaming pointwise kernels only)
  return !a.up_even && a.Cin % CC == 0 && (a.Cout == 16 || a.Cout == 32) && a.Do == a.D + 2 * a.pd - 2 && a.Ho == a.H + 2 * a.ph - 2 &&
         a.Wo == a.W + 2 * a.pw - 2 && u_available(a);
}

template <int WN>
static int launch_rows(const Conv3dFwdArgs &a, const float *U, hipStream_t stream) {
  constexpr int TDr = 2 * (4 / WN), lds = (TDr + 2) * rows_plane(HH) * 4;
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_wino2d_rows_kernel<WN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    if (!ok) (void)hipGetLastError();
  });
  if (!ok) return MR3D_ERR_LAUNCH;
  const int tiles_d = (a.Do + TDr - 1) / TDr, tiles_h = (a.Ho + TH - 1) / TH, tiles_w = (a.Wo + TW - 1) / TW;
  dim3 grid((unsigned)(a.N * tiles_d * tiles_h * tiles_w), (unsigned)(a.Cout / (16 * WN)));
  conv3d_wino2d_rows_kernel<WN><<<grid, kThreads, lds, stream>>>(a, U, tiles_d, tiles_h, tiles_w);
  return mr3d_launch_status();
}

// U of a launch: prepared by the plan run in progress, else transformed now into the scratch the caller lent
static const float *u_for(const Conv3dFwdArgs &a, hipStream_t stream) {
  const auto it = g_prepared.find(a.w);
  if (it != g_prepared.end()) return it->second;
  if (!a.ws || a.ws_bytes < (long long)(u_floats(a.Cin, a.Cout) * sizeof(float))) return nullptr;
  float *U = a.ws;
  const int nw = 3 * a.Cin * a.Cout;
  wino2d_weights_kernel<<<(nw + 255) / 256, 256, 0, stream>>>(a.w, U, u_has_wide(a.Cout) ? U + u_layout_floats(a.Cin, a.Cout) : nullptr, a.Cin, a.Cout);
  return U;
}

int mr3d_conv3d_wino2d_rows(const Conv3dFwdArgs &a, hipStream_t stream) {
  const float *U = u_for(a, stream);
  if (!U) return MR3D_ERR_LAUNCH;
  return a.Cout == 16 ? launch_rows<1>(a, U, stream) : launch_rows<2>(a, U, stream);
}

// plan pre-pass: transform the weights of n convolutions (packed [27][cin][cout] each) in ONE launch; until
// mr3d_conv3d_wino2d_prepare_end() the launches of this thread that use one of these weight buffers skip their own transform
int mr3d_conv3d_wino2d_prepare(const float *const *w, const int *cin, const int *cout, int n, float *ws, long long ws_bytes,
                               hipStream_t stream) {
  if (n <= 0) return MR3D_OK;
  if (n > kPrepMax) n = kPrepMax;                          // the rest take the kernels that need no transform
  size_t total = 0;
  for (int k = 0; k < n; ++k) total += u_floats(cin[k], cout[k]);
  if (!ws || ((uintptr_t)ws & 255) || ws_bytes < (long long)(total * sizeof(float))) return MR3D_ERR_WORKSPACE;
  float *base = ws;
  PrepTable t{};
  size_t off = 0;
  int maxw = 0;
  for (int k = 0; k < n; ++k) {
    t.w[k] = w[k]; t.U[k] = base + off; t.cin[k] = cin[k]; t.cout[k] = cout[k];
    t.Uw[k] = u_has_wide(cout[k]) ? base + off + u_layout_floats(cin[k], cout[k]) : nullptr;
    off += u_floats(cin[k], cout[k]);
    maxw = maxw > 3 * cin[k] * cout[k] ? maxw : 3 * cin[k] * cout[k];
  }
  wino2d_weights_batched_kernel<<<dim3((maxw + 255) / 256, n), 256, 0, stream>>>(t);
  const int st = mr3d_launch_status();
  if (st != MR3D_OK) return st;
  for (int k = 0; k < n; ++k) g_prepared[w[k]] = t.U[k];
  return MR3D_OK;
}

void mr3d_conv3d_wino2d_prepare_end() { g_prepared.clear(); }

#ifdef MR3D_WIDE_STAMPS
extern "C" int mr3d_debug_wide_stamps(unsigned long long *dst) {     // 1024 x 8 x 16 stamps of the last wide launch
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wide_stamps), sizeof(unsigned long long) * 1024 * 8 * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t mr3d_conv3d_wino2d_workspace_bytes(int cin, int cout) {
  if (cin <= 0 || cout <= 0) return 0;
  return u_floats(cin, cout) * sizeof(float);
}

bool mr3d_conv3d_wino2d_ok(const Conv3dFwdArgs &a) {
  return !a.up_even && a.Cin % CC == 0 && a.Cout % 64 == 0 && a.Do == a.D + 2 * a.pd - 2 && a.Ho == a.H + 2 * a.ph - 2 &&
         a.Wo == a.W + 2 * a.pw - 2 && u_available(a) && (long long)a.H * a.W * a.Cin < (1LL << 24);      // 24-bit offset products in the halo staging
}

// Which of the two 64-wide kernels a launch takes: the wide form (4 x 8 x 8 tiles, one persistent 512-thread workgroup per CU) on
// maps with at least two rounds of such tiles whose x fits 32-bit byte offsets; MR3D_TUNE_WINO2D_FORM forces either.
static int g_wino2d_cus = 0;
bool mr3d_conv3d_wino2d_wide(const Conv3dFwdArgs &a) {
  if (g_wino2d_cus == 0) {
    int dev = 0, n = 0;
    g_wino2d_cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  const int form = g_mr3d_tuning[MR3D_TUNE_WINO2D_FORM];
  const long long tiles = (long long)a.N * ((a.Do + WTD - 1) / WTD) * ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW);
  const bool fits = (long long)a.N * a.D * a.H * a.W * a.Cin * 4 < (1LL << 31) && tiles < (1LL << 30);
  // with a liveness bitmap (sparse data gradients in detector training) most tiles leave through the epilogue at once: that is a
  // dispatch-rate question, and the 2 x 8 x 8 kernel's 256-thread workgroups (two or three per CU) retire dead tiles faster than one
  // 512-thread workgroup per CU -- the wide form takes such launches only when forced
  if (a.x_live && form != 2) return false;
  return fits && (form == 2 || (form == 0 && tiles >= 2LL * g_wino2d_cus));
}

int mr3d_conv3d_wino2d(const Conv3dFwdArgs &a, hipStream_t stream) {
  const float *U = u_for(a, stream);
  if (!U) return MR3D_ERR_LAUNCH;
  const int tiles_h = (a.Ho + TH - 1) / TH, tiles_w = (a.Wo + TW - 1) / TW;
  static std::once_flag attr_once;                       // LDS above 64 KiB needs the opt-in attribute (one CU has 160 KiB)
  static bool attr_ok = false;
  std::call_once(attr_once, [] {
    auto lds_of = [](const void *f, int bytes) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess; };
    attr_ok = lds_of(reinterpret_cast<const void *>(conv3d_wino2d_kernel<false>), LDS_BYTES) &&
              lds_of(reinterpret_cast<const void *>(conv3d_wino2d_kernel<true>), LDS_BYTES) &&
              lds_of(reinterpret_cast<const void *>(conv3d_wino2d_wide_kernel<false>), WLDS_BYTES) &&
              lds_of(reinterpret_cast<const void *>(conv3d_wino2d_wide_kernel<true>), WLDS_BYTES);
    if (!attr_ok) (void)hipGetLastError();
  });
  if (!attr_ok) return MR3D_ERR_LAUNCH;
  if (mr3d_conv3d_wino2d_wide(a)) {
    const int tiles_dw = (a.Do + WTD - 1) / WTD;
    const long long wide_tiles = (long long)a.N * tiles_dw * tiles_h * tiles_w;
    const float *Uw = U + u_layout_floats(a.Cin, a.Cout);
    if (a.x_live) {                                      // one tile per workgroup: an all-dead halo leaves through the epilogue at once
      conv3d_wino2d_wide_kernel<true><<<dim3((unsigned)wide_tiles, (unsigned)(a.Cout / 64)), kThreadsW, WLDS_BYTES, stream>>>(a, Uw, tiles_dw, tiles_h, tiles_w);
    } else {
      const unsigned wgs = (unsigned)(wide_tiles < g_wino2d_cus ? wide_tiles : g_wino2d_cus);
      conv3d_wino2d_wide_kernel<false><<<dim3(wgs, (unsigned)(a.Cout / 64)), kThreadsW, WLDS_BYTES, stream>>>(a, Uw, tiles_dw, tiles_h, tiles_w);
    }
    return mr3d_launch_status();
  }
  const int tiles_d = (a.Do + TD - 1) / TD;
  dim3 grid((unsigned)(a.N * tiles_d * tiles_h * tiles_w), (unsigned)(a.Cout / 64));
  if (a.x_live) conv3d_wino2d_kernel<true><<<grid, kThreads, LDS_BYTES, stream>>>(a, U, tiles_d, tiles_h, tiles_w);
  else conv3d_wino2d_kernel<false><<<grid, kThreads, LDS_BYTES, stream>>>(a, U, tiles_d, tiles_h, tiles_w);
  return mr3d_launch_status();
}

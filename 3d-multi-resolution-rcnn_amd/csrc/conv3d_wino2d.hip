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
__global__ __launch_bounds__(256) void wino2d_weights_kernel(const float *__restrict__ w, float *__restrict__ U, int cin, int cout) {
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
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((s_ok >> j) & 1u) {
      const int row = o.rg + 6 * j;
      const int hd = (row >= HH) + (row >= 2 * HH) + (row >= 3 * HH), hh = row - hd * HH;
      const int rel = (int)__umul24(hd, s_plane) + (int)__umul24(hh, s_rowp) + (int)__umul24(o.hw, p.Cin) + o.q * 4;   // full-rate multiplies
      v = buffer_load4(xrsrc, (unsigned)rel * 4u, (unsigned)c0 * 4u);
    }
    return v;
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
    if (more) load_b(bnx, c0, (s + 1) / 4, (s + 1) % 4);
    else if (c0 + CC < p.Cin) load_b(bnx, c0 + CC, 0, 0);
    if (more) read_d((s + 1) / 4, (s + 1) % 4);
    __builtin_amdgcn_sched_barrier(0);                   // the requests stay up here: the scheduler would sink them next to their uses
#pragma unroll
    for (int fh = 0; fh < 4; ++fh)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        mma(mt, fh, av[mt], bw);
        if (more) xform(mt, fh, an[mt]);
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
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((s_ok >> (j0 + jj)) & 1u) v = buffer_load4(xrsrc, (__umul24(hd, s_plane) + __umul24(hh, s_rowp) + col) * 4u, (unsigned)c0 * 4u);
          stage[jj] = v;
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

// ---- batched form of wino2d_weights_kernel for a plan run: one launch transforms the weights of every convolution of the
// plan that will take these kernels (16 launches of ~33 us each sat between the convolutions of a step: 0.5 ms of 11.8)
constexpr int kPrepMax = 32;
struct PrepTable { const float *w[kPrepMax]; float *U[kPrepMax]; int cin[kPrepMax], cout[kPrepMax]; };
__global__ __launch_bounds__(256) void wino2d_weights_batched_kernel(PrepTable t) {
  const int k = blockIdx.y, cin = t.cin[k], cout = t.cout[k];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 3 * cin * cout) return;
  const float *w = t.w[k];
  float *U = t.U[k];
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
    }
  }
}

// weights already transformed for the plan run in progress on this thread: packed-weight pointer -> U
thread_local std::map<const float *, const float *> g_prepared;

size_t u_floats(int cin, int cout) { return ((size_t)3 * NF * cin * cout + 63) / 64 * 64; }

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
  wino2d_weights_kernel<<<(nw + 255) / 256, 256, 0, stream>>>(a.w, U, a.Cin, a.Cout);
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

extern "C" size_t mr3d_conv3d_wino2d_workspace_bytes(int cin, int cout) {
  if (cin <= 0 || cout <= 0) return 0;
  return u_floats(cin, cout) * sizeof(float);
}

bool mr3d_conv3d_wino2d_ok(const Conv3dFwdArgs &a) {
  return !a.up_even && a.Cin % CC == 0 && a.Cout % 64 == 0 && a.Do == a.D + 2 * a.pd - 2 && a.Ho == a.H + 2 * a.ph - 2 &&
         a.Wo == a.W + 2 * a.pw - 2 && u_available(a) && (long long)a.H * a.W * a.Cin < (1LL << 24);      // 24-bit offset products in the halo staging
}

int mr3d_conv3d_wino2d(const Conv3dFwdArgs &a, hipStream_t stream) {
  const float *U = u_for(a, stream);
  if (!U) return MR3D_ERR_LAUNCH;
  const int tiles_d = (a.Do + TD - 1) / TD, tiles_h = (a.Ho + TH - 1) / TH, tiles_w = (a.Wo + TW - 1) / TW;
  static std::once_flag attr_once;                       // LDS above 64 KiB needs the opt-in attribute (one CU has 160 KiB)
  static bool attr_ok = false;
  std::call_once(attr_once, [] {
    attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_wino2d_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES) == hipSuccess &&
              hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_wino2d_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES) == hipSuccess;
    if (!attr_ok) (void)hipGetLastError();
  });
  if (!attr_ok) return MR3D_ERR_LAUNCH;
  dim3 grid((unsigned)(a.N * tiles_d * tiles_h * tiles_w), (unsigned)(a.Cout / 64));
  if (a.x_live) conv3d_wino2d_kernel<true><<<grid, kThreads, LDS_BYTES, stream>>>(a, U, tiles_d, tiles_h, tiles_w);
  else conv3d_wino2d_kernel<false><<<grid, kThreads, LDS_BYTES, stream>>>(a, U, tiles_d, tiles_h, tiles_w);
  return mr3d_launch_status();
}

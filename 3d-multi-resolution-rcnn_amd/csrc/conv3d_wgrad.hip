// conv3d weight gradient on the fp32 matrix cores of gfx950.
//
//   dw[tap][ci][co] = sum over (n, output position p) of x[n, p*stride + tap - pad][ci] * dy[n, p][co]
//
// the wgrad half of the autograd backward of torch.nn.Conv3d at the reference's call sites
// (mmdet/models/backbones/resnet3d.py:126,139,173,251,398; utils/conv_module3d.py:34).
//
// GEMM view per tap: M = Cin (a chunk of CC channels), N = Cout, K = output positions -- the long
// dimension, so every workgroup walks many 256-position tiles and keeps its accumulators in
// registers the whole time.  At the end the partial sums go either straight into dw with fp32 atomics or --
// when the caller supplies a workspace -- to a private block per workgroup with plain stores, and a second kernel
// adds the blocks: all workgroups of a (chunk, Cout tile) add into the SAME taps*CC*NB addresses, device-scope
// atomics are executed at the memory side one after another per address, and 512 workgroups x 27,648 elements =
// 14 M of them cost more than 0.2 ms per launch whatever the size of the map (on P3..P5 more than the MFMA work).
//   * per tile the x halo block [halo position][CC] and the dy tile [position][NB] are staged in LDS
//     (the same halo block serves all taps: a tap is a constant LDS offset);
//   * v_mfma_f32_16x16x4_f32: A[i = ci][k = 4 consecutive positions], B[k][j = co];
//   * the 4 waves split the TAPS (each wave owns every 4th tap with its own accumulators and streams
//     all 256 positions); a pointwise conv (1 tap) splits the POSITIONS across waves instead;
//   * the 7x7x7 stem (Cin = 3 padded to 4) merges (kw, ci) into the GEMM row: with channels innermost
//     and no row padding, row i = kw*4 + ci of tap group (kd, kh) is simply halo_float[base + i],
//     so a 28(32)-row M dimension exists even though Cin is 4.
#include "conv3d.h"
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;

// WINO_: Winograd F(3,2) along W for 3x3x3 / stride 1.  A GEMM-K row is a PAIR of adjacent outputs; per tap row (kd,kh)
// the three kw gradients come from four products M_f = sum_pairs (B^T d)_f (G g)_f with d = x[2p .. 2p+3] and
// g = dy[2p], dy[2p+1]:  B^T d = (d0-d2, d1+d2, d2-d1, d3-d1),  G g = (g0, (g0+g1)/2, (g0-g1)/2, g1),
//   dw[kw=0] = M0+M1+M2, dw[1] = M1-M2, dw[2] = M1+M2+M3  -- 4 MFMAs per pair instead of 6.
// Wave f owns product f for all nine (kd,kh) (9 accumulator groups per wave, an even split where the direct form
// has 7/7/7/6); both transforms happen on the fly from the same LDS tiles; the final combination rides on the atomics.
template <int KD_, int KH_, int KW_, int SD_, int SH_, int SW_, int CC_, int NB_, bool MERGEW_, int TD_ = 4, int TH_ = 8, int TW_ = 8,
          bool WINO_ = false>
struct WgCfg {
  static constexpr int KD = KD_, KH = KH_, KW = KW_, SD = SD_, SH = SH_, SW = SW_, CC = CC_, NB = NB_;
  static constexpr bool MERGEW = MERGEW_, WINO = WINO_;
  // MV real positions per tile, run as M GEMM-K rows (a 5x7x7 tile: 245 of 256; the padding rows carry dy = 0)
  static constexpr int TD = TD_, TH = TH_, TW = TW_, MV = TD * TH * TW, M = (MV + 63) / 64 * 64;
  static constexpr bool PW1 = (KD == 1 && KH == 1 && KW == 1);
  static constexpr int LSD = PW1 ? 1 : SD, LSH = PW1 ? 1 : SH, LSW = PW1 ? 1 : SW;
  static constexpr int GSD = PW1 ? SD : 1, GSH = PW1 ? SH : 1, GSW = PW1 ? SW : 1;
  static constexpr int HD = (TD - 1) * LSD + KD, HH = (TH - 1) * LSH + KH, HW = (TW - 1) * LSW + KW;
  static constexpr int HP = HD * HH * HW;
  // WINO: a k-step's lane groups read output pairs, i.e. rows 2*ROW (2*DROW) apart: 24 / 72 floats put them 16 banks
  // apart (conflict-free with the 16 channel lanes), and keep the rows 16-byte aligned for vector staging stores
  // (kept compact: a workgroup of <= 107 KB leaves room for a streaming pointwise wave of the other stream on the CU)
  static constexpr bool WALIGN = false;      // measured: the conflict-free 24 / 72-float rows buy nothing, and cost 24 KB of LDS
  static constexpr int ROW = WALIGN ? CC + 8 : (MERGEW ? CC : CC + 1);
  static constexpr int DROW = WALIGN ? NB + 8 : NB + 1;
  static constexpr int XF = HP * ROW + (MERGEW ? 32 : 0);   // merged rows of the last halo position read past it
  // WINO: the epilogue combines the four waves' products in LDS (4 x 9 x CC x NB floats) before the atomics
  // (in two passes over the tap rows, so that it fits the tile storage)
  static constexpr int COMB_PASSES = 2, COMB_G = (KD * KH + COMB_PASSES - 1) / COMB_PASSES;
  static constexpr int LDS_TILES = (XF + M * DROW) * 4, LDS_COMB = WINO_ ? 4 * COMB_G * CC * NB * 4 : 0;
  static constexpr int LDS_BYTES = LDS_TILES > LDS_COMB ? LDS_TILES : LDS_COMB;
  static constexpr int MROWS = MERGEW ? 32 : CC;
  static constexpr int MT = MROWS / 16, NT = NB / 16;
  static constexpr int TAPG = WINO_ ? KD * KH * 4 : (MERGEW ? KD * KH : KD * KH * KW);   // accumulator groups
  static constexpr int TPW = PW1 ? 1 : (TAPG + 3) / 4;           // tap groups per wave
  // two workgroups per CU (one hides the other's staging) where LDS allows it and the register budget nearly does
  // partial-sum block of one workgroup: [tap group][row][NB]; a pointwise conv has one block per WAVE (waves split positions)
  static constexpr int PER_BLOCK = (PW1 ? 1 : TAPG) * MROWS * NB;
  static constexpr int BLOCKS_PER_WG = PW1 ? 4 : 1;
  static constexpr int MINWG = (KD == 3 && SD == 1 && NB == 32 && LDS_BYTES * 2 <= 160 * 1024) ? 2 : 1;
  // one workgroup per CU: the next tile is fetched into registers before the MFMA loop (nothing else hides the global
  // latency).  Two per CU hide each other's staging, and holding 18 float4 across the MFMA loop would not fit the
  // halved register budget (it spilled 132 B/lane, and a kernel that uses scratch stalls the queue ~19 us at dispatch)
  static constexpr bool PREFETCH = (MINWG == 1);
  static_assert(!MERGEW || CC == 4, "kw-merge needs 4 channels, unpadded rows");
  static_assert(!WINO_ || (KD_ == 3 && KH_ == 3 && KW_ == 3 && SD_ == 1 && SH_ == 1 && SW_ == 1 && TW_ % 2 == 0 && !MERGEW_ &&
                           TD_ * TH_ * TW_ % 64 == 0), "WINO: 3x3x3 stride 1, whole tiles");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <class C>
__global__ __launch_bounds__(kThreads, C::MINWG) void conv3d_wgrad_kernel(Conv3dWgradArgs p, int tiles_d, int tiles_h, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *xh = lds;
  float *dyt = lds + C::XF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int c0 = blockIdx.y * C::CC;
  const int n0 = blockIdx.z * C::NB;
  const int num_tiles = p.N * tiles_d * tiles_h * tiles_w;

  f32x4 acc[C::TPW][C::MT][C::NT];
#pragma unroll
  for (int t = 0; t < C::TPW; ++t)
    for (int mt = 0; mt < C::MT; ++mt)
      for (int nt = 0; nt < C::NT; ++nt) acc[t][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this thread's share of sum_p dy[p][co] (only the first Cin chunk reduces it): taken from the staging registers,
  // where a thread always holds the same four channels (kThreads % QN == 0) -- 64 adds per tile instead of 64
  // dependent LDS reads
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool want_bias = p.dbias && blockIdx.y == 0;
  // LDS offset of each of this wave's tap groups (wave-uniform)
  int tapoff[C::TPW];
#pragma unroll
  for (int t = 0; t < C::TPW; ++t) {
    int g = t * 4 + wave;
    if (g >= C::TAPG) g = C::TAPG - 1;          // surplus slot: recomputes the last group, result dropped
    if (C::WINO) {
      tapoff[t] = (((t / C::KH) * C::HH + t % C::KH) * C::HW) * C::ROW;      // group t = (kd, kh); this wave's product f = wave
    } else if (C::MERGEW) {
      const int kd = g / C::KH, kh = g % C::KH;
      tapoff[t] = ((kd * C::HH + kh) * C::HW) * C::ROW;
    } else {
      const int kw = g % C::KW, kh = (g / C::KW) % C::KH, kd = g / (C::KW * C::KH);
      tapoff[t] = ((kd * C::HH + kh) * C::HW + kw) * C::ROW;
    }
  }

  // ---- tile staging, software-pipelined: the NEXT tile's x halo block and dy tile are fetched into registers before
  // the MFMA loop of the current tile and written to LDS after it (with ~107 KB of LDS per workgroup only one
  // workgroup fits a CU, so nothing else would hide the global latency).  Everything about a staging slot that does not
  // depend on the tile is tabulated once per thread: its element offset from the tile's corner (32-bit) and its packed
  // halo coordinates for the bounds test; per tile only a wave-uniform 64-bit base remains.
  constexpr int Q = C::CC / 4, QN = C::NB / 4;
  static_assert(kThreads % QN == 0, "a thread keeps the same dy channels in every staging slot");
  constexpr int NX = (C::HP * Q + kThreads - 1) / kThreads, NDY = (C::M * QN + kThreads - 1) / kThreads;
  constexpr unsigned kNoSlot = 0xffffffffu;
  float4 xs[NX], dys[NDY];
  int xoff[NX], doff[NDY];
  unsigned xc[NX], dc[NDY];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int it = tid + j * kThreads;
    const int pos = it / Q, q = it - pos * Q;
    const int hw = (pos % C::HW) * C::GSW, hh = ((pos / C::HW) % C::HH) * C::GSH, hd = (pos / (C::HW * C::HH)) * C::GSD;
    xoff[j] = ((hd * p.H + hh) * p.W + hw) * p.Cin + q * 4;
    xc[j] = it < C::HP * Q ? (unsigned)(hd | (hh << 8) | (hw << 16)) : kNoSlot;
  }
#pragma unroll
  for (int j = 0; j < NDY; ++j) {
    const int it = tid + j * kThreads;
    const int m = it / QN, q = it - m * QN;
    const int tw = m % C::TW, th = (m / C::TW) % C::TH, td = m / (C::TW * C::TH);
    doff[j] = ((td * p.Ho + th) * p.Wo + tw) * p.Cout + q * 4;
    dc[j] = (it < C::M * QN && m < C::MV) ? (unsigned)(td | (th << 8) | (tw << 16)) : kNoSlot;
  }
  auto fetch = [&](int tile) {
    int b = tile;
    const int tw_i = b % tiles_w; b /= tiles_w;
    const int th_i = b % tiles_h; b /= tiles_h;
    const int td_i = b % tiles_d; b /= tiles_d;
    const int n = b;
    const int od0 = td_i * C::TD, oh0 = th_i * C::TH, ow0 = tw_i * C::TW;
    const int id0 = od0 * C::SD - p.pd, ih0 = oh0 * C::SH - p.ph, iw0 = ow0 * C::SW - p.pw;
    // corner of the halo block / of the dy tile (may lie outside the tensor: only in-range slots are dereferenced)
    const float *xb = p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * (long long)p.Cin + c0;
    const float *db = p.dy + ((((long long)n * p.Do + od0) * p.Ho + oh0) * p.Wo + ow0) * (long long)p.Cout + n0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const unsigned c = xc[j];
      const bool ok = c != kNoSlot && (unsigned)(id0 + (int)(c & 255u)) < (unsigned)p.D &&
                      (unsigned)(ih0 + (int)((c >> 8) & 255u)) < (unsigned)p.H && (unsigned)(iw0 + (int)(c >> 16)) < (unsigned)p.W;
      xs[j] = ok ? *reinterpret_cast<const float4 *>(xb + xoff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const unsigned c = dc[j];
      const bool ok = c != kNoSlot && od0 + (int)(c & 255u) < p.Do && oh0 + (int)((c >> 8) & 255u) < p.Ho &&
                      ow0 + (int)(c >> 16) < p.Wo;
      dys[j] = ok ? *reinterpret_cast<const float4 *>(db + doff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (C::PREFETCH && (int)blockIdx.x < num_tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    __syncthreads();
    if (!C::PREFETCH) fetch(tile);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int it = tid + j * kThreads;
      if (it < C::HP * Q) {
        const int pos = it / Q, q = it - pos * Q;
        float *d = xh + pos * C::ROW + q * 4;
        if constexpr (C::WALIGN) *reinterpret_cast<float4 *>(d) = xs[j];
        else { d[0] = xs[j].x; d[1] = xs[j].y; d[2] = xs[j].z; d[3] = xs[j].w; }
      }
    }
    if (C::MERGEW)
      for (int it = tid; it < 32; it += kThreads) xh[C::HP * C::ROW + it] = 0.f;
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int it = tid + j * kThreads;
      if (it < C::M * QN) {
        const int m = it / QN, q = it - m * QN;
        float *d = dyt + m * C::DROW + q * 4;
        if constexpr (C::WALIGN) *reinterpret_cast<float4 *>(d) = dys[j];
        else { d[0] = dys[j].x; d[1] = dys[j].y; d[2] = dys[j].z; d[3] = dys[j].w; }
      }
      if (want_bias) { bsum.x += dys[j].x; bsum.y += dys[j].y; bsum.z += dys[j].z; bsum.w += dys[j].w; }
    }
    __syncthreads();
    if (C::PREFETCH && tile + (int)gridDim.x < num_tiles) fetch(tile + gridDim.x);
    // ---- K loop: 4 positions per MFMA
    constexpr int KSTEPS = C::M / 4 / (C::WINO ? 2 : 1);    // WINO: a k row is an output pair
    constexpr int KPW = C::PW1 ? KSTEPS / 4 : KSTEPS;       // pointwise: waves split the positions
    // WINO, wave-uniform: a = x[oA] + sB * x[oB] and b = cg0 * dy[2p] + cg1 * dy[2p+1] of this wave's product
    const int woA = wave * C::ROW, woB = (wave <= 1 ? 2 : 1) * C::ROW;
    const float wsB = wave == 1 ? 1.f : -1.f;
    const float cg0 = wave == 0 ? 1.f : (wave == 3 ? 0.f : 0.5f), cg1 = wave == 0 ? 0.f : (wave == 3 ? 1.f : (wave == 1 ? 0.5f : -0.5f));
    const int kk0 = C::PW1 ? wave * KPW : 0;
    // explicit two-stage software pipeline: the LDS operands of k-step q+1 are requested before the MFMAs
    // of k-step q issue (the compiler does not do this on its own: it waits on every ds_read before its MFMA)
    auto ld = [&](int kq, float (&a)[C::TPW][C::MT], float (&bf)[C::NT]) {
      if constexpr (C::WINO) {
        const int mp = kq * 4 + lk;                              // pair index inside the tile
        constexpr int PW_ = C::TW / 2;
        const int tp = mp % PW_, th = (mp / PW_) % C::TH, td = mp / (PW_ * C::TH);
        const int hp = ((td * C::HH + th) * C::HW + 2 * tp) * C::ROW + li;
        const float *dp = dyt + (2 * mp) * C::DROW + li;
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) bf[nt] = cg0 * dp[nt * 16] + cg1 * dp[C::DROW + nt * 16];
#pragma unroll
        for (int t = 0; t < C::TPW; ++t)
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            const float *xp = xh + hp + tapoff[t] + mt * 16;
            a[t][mt] = xp[woA] + wsB * xp[woB];
          }
        return;
      }
      const int m = (kk0 + kq) * 4 + lk;
      const int mx = (C::MV < C::M && m >= C::MV) ? 0 : m;      // padding row: dy is 0 there, any in-range x address
      const int tw = mx % C::TW, th = (mx / C::TW) % C::TH, td = mx / (C::TW * C::TH);
      const int hp = (((td * C::LSD) * C::HH + th * C::LSH) * C::HW + tw * C::LSW) * C::ROW + li;
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) bf[nt] = dyt[m * C::DROW + nt * 16 + li];
#pragma unroll
      for (int t = 0; t < C::TPW; ++t)
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) a[t][mt] = xh[hp + tapoff[t] + mt * 16];
    };
    auto mm = [&](const float (&a)[C::TPW][C::MT], const float (&bf)[C::NT]) {
#pragma unroll
      for (int t = 0; t < C::TPW; ++t)
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
            acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][mt], bf[nt], acc[t][mt][nt], 0, 0, 0);
    };
    static_assert(KPW % 2 == 0, "k-steps are processed in pairs");
    float a0[C::TPW][C::MT], a1[C::TPW][C::MT], b0[C::NT], b1[C::NT];
    ld(0, a0, b0);
    for (int kq = 0; kq < KPW; kq += 2) {
      ld(kq + 1, a1, b1);
      mm(a0, b0);
      if (kq + 2 < KPW) ld(kq + 2, a0, b0);
      mm(a1, b1);
    }
  }

  if (want_bias) {
    // reduce the per-thread partials of each channel inside the workgroup first: ONE atomic per channel per
    // workgroup (256 lanes x 512 workgroups hammering 16 addresses cost more than the MFMA work itself).
    // Thread t holds channels 4*(t % QN) .. +3; slots that were out of range added zeros.
    __syncthreads();
    lds[tid * 4 + 0] = bsum.x; lds[tid * 4 + 1] = bsum.y; lds[tid * 4 + 2] = bsum.z; lds[tid * 4 + 3] = bsum.w;
    __syncthreads();
    if (tid < C::NB) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < kThreads / QN; ++r) s += lds[(r * QN + (tid >> 2)) * 4 + (tid & 3)];
      atomicAdd(p.dbias + n0 + tid, s);
    }
  }
  // ---- partial sums: private block (two-stage reduction) or atomics into dw[tap][ci][co]
  if (p.part && !C::MERGEW && !C::WINO) {
    const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float *dst = p.part + (wg * C::BLOCKS_PER_WG + (C::PW1 ? wave : 0)) * C::PER_BLOCK;
#pragma unroll
    for (int t = 0; t < C::TPW; ++t) {
      const int g = t * 4 + wave;
      if (!C::PW1 && g >= C::TAPG) continue;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dst[((C::PW1 ? 0 : g) * C::MROWS + mt * 16 + lk * 4 + r) * C::NB + nt * 16 + li] = acc[t][mt][nt][r];
    }
    return;
  }
  if constexpr (C::WINO) {
    // dw[kd][kh][0] = M0+M1+M2, [1] = M1-M2, [2] = M1+M2+M3: the four waves' products meet in LDS, then one atomic per
    // element per workgroup (as many as the direct form issues; adding coefficient * M_f from every wave would be 2.7x)
    constexpr int BLK = C::CC * C::NB, NG = C::COMB_G;
#pragma unroll
    for (int pass = 0; pass < C::COMB_PASSES; ++pass) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < C::TPW; ++t) {
        if (t / NG != pass) continue;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              lds[(wave * NG + t % NG) * BLK + (mt * 16 + lk * 4 + r) * C::NB + nt * 16 + li] = acc[t][mt][nt][r];
      }
      __syncthreads();
      const int ng = (pass + 1) * NG <= C::TPW ? NG : C::TPW - pass * NG;       // tap rows in this pass
      for (int e = tid; e < ng * BLK; e += kThreads) {
        const int t = pass * NG + e / BLK, rc = e % BLK;
        const float m0 = lds[e], m1 = lds[NG * BLK + e], m2 = lds[2 * NG * BLK + e], m3 = lds[3 * NG * BLK + e];
        float *dst = p.dw + ((size_t)(t * 3) * p.Cin + c0 + rc / C::NB) * p.Cout + n0 + rc % C::NB;
        const size_t tap = (size_t)p.Cin * p.Cout;
        atomicAdd(dst, (m0 + m1) + m2);
        atomicAdd(dst + tap, m1 - m2);
        atomicAdd(dst + 2 * tap, (m1 + m2) + m3);
      }
    }
    return;
  }
  if constexpr (C::PW1) {
    // pointwise: the four waves hold partials of the SAME [CC x NB] block (they split the positions): sum them in LDS
    // and issue one atomic per element per workgroup (four waves adding to the same addresses at the same moment
    // serialise at the memory side)
    static_assert(4 * C::CC * C::NB * 4 <= C::LDS_BYTES, "partials fit the tile storage");
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(wave * C::CC + mt * 16 + lk * 4 + r) * C::NB + nt * 16 + li] = acc[0][mt][nt][r];
    __syncthreads();
    for (int e = tid; e < C::CC * C::NB; e += kThreads) {
      const float s = (lds[e] + lds[C::CC * C::NB + e]) + (lds[2 * C::CC * C::NB + e] + lds[3 * C::CC * C::NB + e]);
      atomicAdd(p.dw + (size_t)(c0 + e / C::NB) * p.Cout + n0 + e % C::NB, s);
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < C::TPW; ++t) {
    const int g = t * 4 + wave;
    if (!C::PW1 && g >= C::TAPG) continue;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mt * 16 + lk * 4 + r;
          const int co = n0 + nt * 16 + li;
          size_t idx;
          if (C::MERGEW) {
            const int kw = row >> 2, ci = row & 3;
            if (kw >= C::KW) continue;
            idx = ((size_t)(g * C::KW + kw) * p.Cin + ci) * p.Cout + co;
          } else {
            idx = ((size_t)(C::PW1 ? 0 : g) * p.Cin + c0 + row) * p.Cout + co;
          }
          atomicAdd(p.dw + idx, acc[t][mt][nt][r]);
        }
  }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ part, int nparts, int per, int R, int NBc,
                                                              int cstep, int nstep, int Cin, int Cout, float *__restrict__ dw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= per) return;
  const int y = blockIdx.y, z = blockIdx.z;
  const float *src = part + ((size_t)z * gridDim.y + y) * nparts * per + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 3 < nparts; b += 4) {
    s0 += src[(size_t)b * per]; s1 += src[(size_t)(b + 1) * per]; s2 += src[(size_t)(b + 2) * per]; s3 += src[(size_t)(b + 3) * per];
  }
  for (; b < nparts; ++b) s0 += src[(size_t)b * per];
  const int col = e % NBc, row = (e / NBc) % R, g = e / (NBc * R);
  dw[((size_t)g * Cin + y * cstep + row) * Cout + z * nstep + col] += (s0 + s1) + (s2 + s3);
}

}  // namespace

int mr3d_reduce_partials(const float *part, int nparts, int G, int R, int NBc, int chunks, int ntiles, int cstep, int nstep,
                         int Cin, int Cout, float *dw, hipStream_t stream) {
  const int per = G * R * NBc;
  dim3 grid((unsigned)((per + 255) / 256), (unsigned)chunks, (unsigned)ntiles);
  reduce_partials_kernel<<<grid, 256, 0, stream>>>(part, nparts, per, R, NBc, cstep, nstep, Cin, Cout, dw);
  return mr3d_launch_status();
}

namespace {

template <class C>
int launch_wgrad(const Conv3dWgradArgs &a, hipStream_t stream) {
  const int tiles_d = (a.Do + C::TD - 1) / C::TD, tiles_h = (a.Ho + C::TH - 1) / C::TH, tiles_w = (a.Wo + C::TW - 1) / C::TW;
  const int num_tiles = a.N * tiles_d * tiles_h * tiles_w;
  static bool attr_done = false;
  if (!attr_done && C::LDS_BYTES > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_wgrad_kernel<C>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int chunks = a.Cin / C::CC, ntiles = a.Cout / C::NB;
  constexpr int total_wgs = 512;   // tuning knob
  int gx = total_wgs / (chunks * ntiles);      // ~2 workgroups per CU in total
  if (gx < 1) gx = 1;
  if (gx > num_tiles) gx = num_tiles;
  dim3 grid((unsigned)gx, (unsigned)chunks, (unsigned)ntiles);
  Conv3dWgradArgs k = a;
  const long long need = (long long)gx * chunks * ntiles * C::BLOCKS_PER_WG * C::PER_BLOCK * 4;
  if (C::MERGEW || C::WINO || !k.part || k.part_bytes < need) k.part = nullptr;
  conv3d_wgrad_kernel<C><<<grid, kThreads, C::LDS_BYTES, stream>>>(k, tiles_d, tiles_h, tiles_w);
  const int st = mr3d_launch_status();
  if (st != MR3D_OK || !k.part) return st;
  return mr3d_reduce_partials(k.part, gx * C::BLOCKS_PER_WG, C::PW1 ? 1 : C::TAPG, C::MROWS, C::NB, chunks, ntiles, C::CC, C::NB,
                              a.Cin, a.Cout, a.dw, stream);
}

template <int K, int SD, int SH, int SW, int CC, int TD = 4>
int dispatch_wg(const Conv3dWgradArgs &a, int nb, hipStream_t stream) {
  switch (nb) {
    case 16: return launch_wgrad<WgCfg<K, K, K, SD, SH, SW, CC, 16, false, TD>>(a, stream);
    case 32: return launch_wgrad<WgCfg<K, K, K, SD, SH, SW, CC, 32, false, TD>>(a, stream);
    case 64: return launch_wgrad<WgCfg<K, K, K, SD, SH, SW, CC, 64, false, TD>>(a, stream);
  }
  return MR3D_ERR_UNSUPPORTED;
}

}  // namespace

// which form the 3x3x3 / stride-1 weight gradient takes (one decision, shared by the launch and by the measurement harness's
// mr3d_conv3d_wgrad_issue_factor)
namespace {
enum WgForm { WG_DIRECT_577, WG_WINO2D, WG_WINO_W64, WG_WINO_W32, WG_WINO_W16, WG_DIRECT };
WgForm wgrad_k3s1_form(const Conv3dWgradArgs &a, int nb, int *nb3_out) {
  const long long vol = (long long)a.Do * a.Ho * a.Wo;
  const long long t577 = (long long)((a.Do + 4) / 5) * ((a.Ho + 6) / 7) * ((a.Wo + 6) / 7);
  const long long t488 = (long long)((a.Do + 3) / 4) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8);
  if ((double)vol / t577 > 1.15 * (double)vol / t488 && a.N * t577 >= 512) return WG_DIRECT_577;
  // 3x3x3, Cout % 64 == 0: the 32-wide Cout tile (two workgroups per CU, twice the workgroups) wins on maps with few
  // tiles (P3: 100 vs 79 TFLOP/s, layer4: 31 vs 25), the 64-wide one on the big maps (P2: 111 vs 106)
  const int nb3 = a.N * t488 < 2048 ? 32 : 64;
  *nb3_out = nb3;
  // Winograd forms (a caller that lends a workspace wants the bit-reproducible two-stage reduction: direct form)
  constexpr int wino_env = 4;    // 0: direct form only; 1/2/3: F(3,2) along W on 64- / +32- / +16-wide tiles; 4: + the two-dimensional form
  const int wino = a.part ? 0 : wino_env;
  // 4: the two-dimensional form (conv3d_wgrad_wino2d.hip) on maps with >= min2d tiles of 2x8x8: P2 64->64 1096 -> 857 us
  // standalone, backbone step 11.68 -> 11.48 ms, configs[2] step 19.7 -> 19.3 ms
  constexpr int min2d = 256;
  if (wino >= 4 && nb == 64 && (long long)a.N * ((a.Do + 1) / 2) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8) >= min2d &&
      mr3d_conv3d_wgrad_wino2d_ok(a))
    return WG_WINO2D;
  if (wino && nb == 64 && nb3 == 64) return WG_WINO_W64;
  if (wino >= 2 && nb >= 32 && nb3 == 32) return WG_WINO_W32;
  if (wino >= 3 && nb == 16) return WG_WINO_W16;
  return WG_DIRECT;
}
void wgrad_args_from_desc(const mr3d_conv3d_wgrad_desc *d, Conv3dWgradArgs &a) {
  a.x = d->x; a.dy = d->dy; a.dw = d->dw; a.dbias = d->dbias;
  a.part = d->workspace; a.part_bytes = d->workspace ? d->workspace_bytes : 0;
  a.sparse_dy = d->flags & MR3D_WGRAD_SPARSE_DY;
  a.N = d->n; a.D = d->d; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Cout = d->cout;
  a.pd = d->pd; a.ph = d->ph; a.pw = d->pw;
  a.Do = (d->d + 2 * d->pd - d->kd) / d->sd + 1;
  a.Ho = (d->h + 2 * d->ph - d->kh) / d->sh + 1;
  a.Wo = (d->w_in + 2 * d->pw - d->kw) / d->sw + 1;
}
}  // namespace

// MFMA FLOPs the launch of this descriptor ISSUES / its algorithmic FLOPs (2 * MAC): 1 for the direct forms, 2/3 for F(3,2) along
// W (48 products per patch and kd where the direct form has 72), 1/3 for the two-dimensional form (24); nothing is launched
extern "C" float mr3d_conv3d_wgrad_issue_factor(const mr3d_conv3d_wgrad_desc *d) {
  if (!d || d->kd != d->kh || d->kh != d->kw) return 1.f;
  Conv3dWgradArgs a;
  wgrad_args_from_desc(d, a);
  const int nb = a.Cout % 64 == 0 ? 64 : (a.Cout % 32 == 0 ? 32 : (a.Cout % 16 == 0 ? 16 : 0));
  if (!nb || d->kd != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1 || a.Cin % 16 != 0 || a.Do <= 0 || a.Ho <= 0 || a.Wo <= 0) return 1.f;
  int nb3 = 64;
  switch (wgrad_k3s1_form(a, nb, &nb3)) {
    case WG_WINO2D: return 1.f / 3.f;
    case WG_WINO_W64: case WG_WINO_W32: case WG_WINO_W16: return 2.f / 3.f;
    default: return 1.f;
  }
}

extern "C" int mr3d_conv3d_wgrad(const mr3d_conv3d_wgrad_desc *d, mr3d_stream_t stream_) {
  if (!d || !d->x || !d->dy || !d->dw) return MR3D_ERR_INVALID_ARG;
  if (d->kd != d->kh || d->kh != d->kw) return MR3D_ERR_UNSUPPORTED;
  Conv3dWgradArgs a;
  wgrad_args_from_desc(d, a);
  if (a.N <= 0 || a.Do <= 0 || a.Ho <= 0 || a.Wo <= 0) return MR3D_ERR_INVALID_ARG;
  const int nb = a.Cout % 64 == 0 ? 64 : (a.Cout % 32 == 0 ? 32 : (a.Cout % 16 == 0 ? 16 : 0));
  if (!nb) return MR3D_ERR_UNSUPPORTED;
  hipStream_t stream = mr3d_stream(stream_);
  const int k = d->kd, s = d->sd * 100 + d->sh * 10 + d->sw;
  if (k == 1 && s == 111) {
    if (a.pd == 0 && a.ph == 0 && a.pw == 0) {
      const int st = mr3d_pointwise_wgrad_try(a, stream);
      if (st != MR3D_ERR_UNSUPPORTED) return st;
    }
    if (a.Cin % 64 == 0) return dispatch_wg<1, 1, 1, 1, 64>(a, nb, stream);
    if (a.Cin % 32 == 0) return dispatch_wg<1, 1, 1, 1, 32>(a, nb, stream);
    if (a.Cin % 16 == 0) return dispatch_wg<1, 1, 1, 1, 16>(a, nb, stream);
  } else if (k == 1 && s == 222) {
    if (a.Cin % 64 == 0) return dispatch_wg<1, 2, 2, 2, 64>(a, nb, stream);
    if (a.Cin % 16 == 0) return dispatch_wg<1, 2, 2, 2, 16>(a, nb, stream);
  } else if (k == 3 && s == 111) {
    if (a.Cin % 16 == 0) {
      int nb3 = 64;
      switch (wgrad_k3s1_form(a, nb, &nb3)) {
        case WG_DIRECT_577:                                                            // 10x14x14 roi volumes (mask head)
          switch (nb) {
            case 16: return launch_wgrad<WgCfg<3, 3, 3, 1, 1, 1, 16, 16, false, 5, 7, 7>>(a, stream);
            case 32: return launch_wgrad<WgCfg<3, 3, 3, 1, 1, 1, 16, 32, false, 5, 7, 7>>(a, stream);
            case 64: return launch_wgrad<WgCfg<3, 3, 3, 1, 1, 1, 16, 64, false, 5, 7, 7>>(a, stream);
          }
          break;
        case WG_WINO2D: return mr3d_conv3d_wgrad_wino2d(a, stream);
        case WG_WINO_W64: return launch_wgrad<WgCfg<3, 3, 3, 1, 1, 1, 16, 64, false, 4, 8, 8, true>>(a, stream);
        case WG_WINO_W32: return launch_wgrad<WgCfg<3, 3, 3, 1, 1, 1, 16, 32, false, 4, 8, 8, true>>(a, stream);
        case WG_WINO_W16: return launch_wgrad<WgCfg<3, 3, 3, 1, 1, 1, 16, 16, false, 4, 8, 8, true>>(a, stream);
        case WG_DIRECT: return dispatch_wg<3, 1, 1, 1, 16>(a, nb < nb3 ? nb : nb3, stream);
      }
    }
  } else if (k == 3 && s == 222) {
    if (a.Cin % 16 == 0) return dispatch_wg<3, 2, 2, 2, 16, 1>(a, nb, stream);   // 3x17x17 halo: one output plane per tile
  } else if (k == 7 && s == 122) {
    if (a.Cin == 4 && nb == 16) return launch_wgrad<WgCfg<7, 7, 7, 1, 2, 2, 4, 16, true>>(a, stream);
  } else if (k == 2 && s == 222 && a.pd == 0 && a.ph == 0 && a.pw == 0) {
    // kernel 2 / stride 2: the weight gradient of ConvTranspose3d(k=2, s=2) (roles swapped by the caller)
    if (a.Cin % 16 == 0 && nb == 64) {
      const long long t277 = (long long)((a.Do + 1) / 2) * ((a.Ho + 6) / 7) * ((a.Wo + 6) / 7);
      const long long t288 = (long long)((a.Do + 1) / 2) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8);
      if (t277 * 98 <= t288 * 128 && a.Ho % 7 == 0 && a.Wo % 7 == 0)
        return launch_wgrad<WgCfg<2, 2, 2, 2, 2, 2, 16, 64, false, 2, 7, 7>>(a, stream);
      return launch_wgrad<WgCfg<2, 2, 2, 2, 2, 2, 16, 64, false, 2, 8, 8>>(a, stream);
    }
  }
  return MR3D_ERR_UNSUPPORTED;
}

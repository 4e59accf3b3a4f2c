// 3D convolution forward as an implicit GEMM on the fp32 matrix cores of gfx950.
//
// Replaces the torch.nn.Conv3d (cuDNN) calls of the reference's backbone / neck / heads
// (mmdet/models/backbones/resnet3d.py:126,139,173,251,398; utils/conv_module3d.py:34) with the
// norm / activation / residual steps that follow them (resnet3d.py:198-226, fpn3d.py:99-108)
// fused into the epilogue.  The same kernel computes data gradients (a stride-1 convolution with
// flipped, transposed weights) with the ReLU mask of the producing layer in its epilogue.
//
// GEMM view: M = output positions of one (TD x TH x TW) tile, N = Cout, K = taps x Cin.
//   * A (activations): the tile's input halo block for CC input channels is staged ONCE in LDS
//     as [halo position][CC+1] (odd row stride -> conflict-light ds_read_b32) and re-used by every
//     tap: a tap is just a constant LDS offset.
//   * B (weights): [tap][Cin][Cout] in global memory; every lane loads its own fragment element
//     (rows of Cout are contiguous: 128-B runs per half-wave), prefetched one tap ahead into
//     VGPRs.  Weights never touch LDS, so the only barriers are at Cin-chunk boundaries.
//   * MFMA: v_mfma_f32_32x32x2_f32 (Cout tile 32/64) or v_mfma_f32_16x16x4_f32 (Cout 16): exact
//     fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD -- the fp32 matrix peak (157 TFLOP/s).
//     One wave per SIMD with >= 4 independent accumulator tiles keeps the pipe issuing.
//   * 4 waves per workgroup, M = 256 positions; waves tile the block as 2(M) x 2(N) for a
//     64-wide Cout tile and 4(M) x 1(N) otherwise.
#include "conv3d.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;

// PAR_: the parity-decomposed data gradient of a stride-2 3x3x3 convolution (see conv3d_dgrad_s2k3 below); the
// kernel extents are then the per-axis MAXIMUM (2) and the taps actually walked are chosen per parity class.
// ONE_: Cin == CC, a single chunk -- no next-chunk prefetch, so the staging registers are dead during the MFMA loop.
// WINO_: Winograd F(2,3) along W for 3x3x3 / stride 1: a GEMM row is a PAIR of adjacent outputs (x = 2p, 2p+1); per tap
// row (kd,kh) the three kw taps become four products m_f = (B^T d)_f * (G g)_f over the inputs d = x[2p .. 2p+3]:
//   B^T d = (d0-d2, d1+d2, d2-d1, d1-d3),  G g = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2),
//   y[2p] = m0+m1+m2, y[2p+1] = m1-m2-m3  -- 4 MFMAs per two outputs instead of 6.  Both transforms are done on the fly
// from the same LDS halo and the same weight-row registers as the direct form (one ds_read + one v_sub/v_add per MFMA).
template <int KD, int KH, int KW, int SD_, int SH_, int SW_, int TD_, int TH_, int TW_, int CC_, int NB_, bool PAR_ = false, bool ONE_ = false,
          int WINO_ = 0>
struct FwdCfg {
  static constexpr bool PAR = PAR_, ONE = ONE_;
  // WINO: 0 = direct, 2 = F(2,3) (above), 4 = F(4,3): a GEMM row is FOUR adjacent outputs, six products per tap row
  // instead of twelve (Lavin & Gray's matrices; fp32 error ~5e-6 of the output scale at K = 1728, direct 3e-7)
  static constexpr int WINO = WINO_;
  static constexpr int kD = KD, kH = KH, kW = KW;
  static constexpr int SD = SD_, SH = SH_, SW = SW_, TD = TD_, TH = TH_, TW = TW_, CC = CC_, NB = NB_;
  // halo block: (T-1)*S + K input positions per axis; a pointwise (K == 1) strided conv only
  // touches every S-th position, so it stages exactly the T positions it reads (LS = LDS step per output)
  static constexpr bool PW1 = (KD == 1 && KH == 1 && KW == 1);
  static constexpr int LSD = PW1 ? 1 : SD, LSH = PW1 ? 1 : SH, LSW = PW1 ? 1 : SW;   // LDS step per output position
  static constexpr int GSD = PW1 ? SD : 1, GSH = PW1 ? SH : 1, GSW = PW1 ? SW : 1;   // global step per halo position
  // WPAD: one spare column per halo row in the F(4,3) form.  Their GEMM rows step 2 or 4 positions along W, and with the
  // natural 10-position rows the 32 lanes of a ds_read_b32 group fall 3-way onto the 32 banks; 11-position rows make it 2-way
  // (the best any row pitch reaches: SQ_LDS_BANK_CONFLICT was 66 % of the LDS-active cycles of the F(4,3) kernel).
  static constexpr int WPAD = (WINO_ == 4) ? 1 : 0;        // F(4,3): 0.480 -> 0.477 ms on P2; the F(2,3) 4x8x8 instance would spill
  static constexpr int HD = (TD - 1) * LSD + KD, HH = (TH - 1) * LSH + KH, HW = (TW - 1) * LSW + KW + WPAD;
  static constexpr int HP = HD * HH * HW;       // halo positions
  static constexpr int ROW = CC + 1;            // LDS row stride in floats
  static constexpr int OPR = WINO_ ? WINO_ : 1; // outputs per GEMM row
  static constexpr int MV = TD * TH * TW / OPR;  // GEMM rows of the tile
  static constexpr int NF = WINO_ == 2 ? 4 : (WINO_ == 4 ? 6 : 1);      // accumulator sets per MFMA tile (Winograd products)
  static_assert(!WINO_ || (KD == 3 && KH == 3 && KW == 3 && SD_ == 1 && SH_ == 1 && SW_ == 1 && TW_ % (WINO_ ? WINO_ : 1) == 0 && !PAR_), "WINO: 3x3x3 stride 1");
  // GEMM rows per workgroup: MV rounded up to whole MFMA tiles per wave (a 5x7x7 tile -- the divisor-friendly shape
  // for 10x14x14 roi volumes -- has 245 positions and runs as 256 rows, the last 11 idle)
  static constexpr int MQ = (NB_ == 32) ? 128 : 64;
  static constexpr int M = (MV + MQ - 1) / MQ * MQ;
  static constexpr int TAPS = KD * KH * KW;
  static constexpr bool k16 = (NB == 16);       // 16x16x4 MFMA, else 32x32x2
  static constexpr int TM = k16 ? 16 : 32;      // MFMA tile edge
  static constexpr int KSTEP = k16 ? 4 : 2;     // K consumed per MFMA
  static constexpr int WN = (NB == 64) ? 2 : 1; // waves along N
  static constexpr int WM = 4 / WN;             // waves along M
  static constexpr int WROWS = M / WM;          // rows per wave
  static constexpr int MT = WROWS / TM;         // MFMA tiles per wave along M
  static constexpr int KK = CC / KSTEP;         // MFMA k-steps per tap per chunk
  // Small tiles (64 positions: layer3/4, P4/P5 after the tile narrowing) have ONE MFMA tile per wave, so a
  // tap row is 12 MFMAs = 384 cycles -- less than an L2 round trip: fetching weight fragments a row ahead cannot hide
  // the latency and the kernel crawls at ~1 us per row.  There the chunk's whole weight tile [taps][CC][NB] (if <= 32 KB)
  // is staged in LDS with the same register pipeline as the halo, one Cin chunk ahead.
  static constexpr int WTAPS = PAR_ ? 27 : TAPS;
  static constexpr bool BLDS = (!PAR_ && WINO_ == 0 && M == 64 && WTAPS * CC_ * NB_ * 4 <= 32 * 1024);   // PAR walks <= 8 of its 27 taps
  static constexpr int WLDS = BLDS ? WTAPS * CC_ * NB_ : 0;       // floats
  static constexpr int LDS_BYTES = (HP * ROW + WLDS) * 4;
  // single-chunk kernels (layer1's 3x3x3 16->16: load -> compute -> store with nothing to overlap inside a workgroup)
  // want a third workgroup per CU; without the prefetch registers they fit the 170-register budget without spilling
  // (a kernel that touches scratch stalls the queue for ~19 us at dispatch)
  // NOPF: no register prefetch of the next chunk (two workgroups per CU hide each other's staging instead)
  static constexpr bool NOPF = WINO_ != 0 && TD_ == 4;
  static constexpr int MINWG = ONE_ ? 3 : (NOPF ? 2 : 1);
  static_assert(WROWS >= TM && WROWS % TM == 0, "each wave needs at least one whole MFMA tile of rows");
  static_assert(CC % KSTEP == 0 && CC % 4 == 0, "channel chunk");
  static_assert(LDS_BYTES <= 160 * 1024, "halo block exceeds the CU's LDS");
};

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  // torch nearest: min(floor(dst * (in/out)), in-1)   (F.interpolate(mode='nearest'), fpn3d.py:106-108)
  int s = (int)floorf((float)dst * ((float)in / (float)out));
  return s < in - 1 ? s : in - 1;
}

#ifndef MR3D_FWD_ABL
#define MR3D_FWD_ABL 0                                       // timing-only ablations of the small-tile path (tools/diag/fwd_abl.sh); 0 in the library
#endif
template <class C>
__global__ __launch_bounds__(kThreads, C::MINWG) void conv3d_fwd_kernel(Conv3dFwdArgs p, int tiles_d, int tiles_h, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / C::WN, wave_n = wave % C::WN;

  // block -> (n, tile_d, tile_h, tile_w); blockIdx.y -> Cout tile
  int b = blockIdx.x;
  const int tw_i = b % tiles_w; b /= tiles_w;
  const int th_i = b % tiles_h; b /= tiles_h;
  const int td_i = b % tiles_d; b /= tiles_d;
  const int n = b;
  const int od0 = td_i * C::TD, oh0 = th_i * C::TH, ow0 = tw_i * C::TW;
  const int id0 = od0 * C::SD - p.pd, ih0 = oh0 * C::SH - p.ph, iw0 = ow0 * C::SW - p.pw;
  const int n0 = blockIdx.y * C::NB;
  // parity class of the outputs this workgroup produces (PAR only); the 8-tap class is scheduled first
  const int cls = C::PAR ? 7 - (int)blockIdx.z : 0;
  const int ad = (cls >> 2) & 1, ah = (cls >> 1) & 1, aw = cls & 1;
  const int nkd = C::PAR ? 1 + ad : C::kD, nkh = C::PAR ? 1 + ah : C::kH, nkw = C::PAR ? 1 + aw : C::kW;
  // tap (kd,kh,kw) of the walk -> index into the weight array (PAR: taps {1} or {0,2} of the flipped 3x3x3 kernel)
  auto wtap = [&](int kd_, int kh_, int kw_) {
    if constexpr (C::PAR) return (((ad ? 2 * kd_ : 1) * 3 + (ah ? 2 * kh_ : 1)) * 3 + (aw ? 2 * kw_ : 1));
    else return (kd_ * C::kH + kh_) * C::kW + kw_;
  };

  const int li = lane % C::TM;     // row (A) / column (B) inside the MFMA tile
  const int lk = lane / C::TM;     // k index inside the MFMA k-step

  int abase[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    int m = wave_m * C::WROWS + mt * C::TM + li;
    if (C::MV < C::M && m >= C::MV) m = 0;        // idle padding row: any in-range address, result never stored
    constexpr int RW = C::TW / C::OPR;       // GEMM rows per tile row
    const int tw = (m % RW) * C::OPR, th = (m / RW) % C::TH, td = m / (RW * C::TH);
    abase[mt] = (((td * C::LSD) * C::HH + th * C::LSH) * C::HW + tw * C::LSW) * C::ROW + lk;
  }
  const float *wlane = p.w + (size_t)lk * p.Cout + n0 + wave_n * C::TM + li;
  // halo origin of the tile (may lie in front of x: only addresses inside the volume are dereferenced); the halo spans < 2^24
  // elements, the packed weights < 2^31 bytes
  const __amdgpu_buffer_rsrc_t xrsrc = buffer_of(p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * p.Cin),
                               wrsrc = buffer_of(p.w);

  using acc_t = typename std::conditional<C::k16, f32x4, f32x16>::type;
  acc_t acc[C::NF * C::MT];                    // WINO: [f][mt]
#pragma unroll
  for (int mt = 0; mt < C::NF * C::MT; ++mt)
    for (int r = 0; r < (C::k16 ? 4 : 16); ++r) acc[mt][r] = 0.f;

  // ---- halo staging, software-pipelined over the Cin chunks: the NEXT chunk's halo block is fetched into registers
  // before the MFMA loop of the current one and only written to LDS after it (global latency hidden behind the math)
  constexpr int Q = C::CC / 4;
  constexpr int NIT = (C::HP * Q + kThreads - 1) / kThreads;
  float4 stage[NIT];
  float *wl = lds + C::HP * C::ROW;                                // weight tile of the chunk (BLDS only)
  constexpr int WQ = C::BLDS ? C::WTAPS * C::CC * C::NB / 4 : 0;   // float4 per chunk
  constexpr int NWV = C::BLDS ? (WQ + kThreads - 1) / kThreads : 1;
  float4 wstage[NWV];
  auto fetch = [&](int c0) {
    if (MR3D_FWD_ABL & 4) return;
    if constexpr (C::BLDS) {
#pragma unroll
      for (int j = 0; j < NWV; ++j) {
        const int e = tid + j * kThreads;
        constexpr int PER_TAP = C::CC * C::NB / 4, PER_ROW = C::NB / 4;
        const int tap = e / PER_TAP, rem = e - tap * PER_TAP;
        const int ci = rem / PER_ROW, q = rem - ci * PER_ROW;
        // buffer load: 32-bit lane offset (24-bit multiplies) + the chunk as a SCALAR offset -- a size_t product chain per load is
        // 4-5 quarter-rate integer multiplies, and on the small maps this staging sits in the kernel's latency chain
        wstage[j] = (e < WQ) ? buffer_load4(wrsrc, (__umul24(__umul24(tap, p.Cin) + ci, p.Cout) + (unsigned)(n0 + q * 4)) * 4u,
                                            (unsigned)(c0 * p.Cout) * 4u)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int it = tid + j * kThreads;
      const int pos = it / Q, q = it - pos * Q;
      const int hw = pos % C::HW, hh = (pos / C::HW) % C::HH, hd = pos / (C::HW * C::HH);
      const int id = id0 + hd * C::GSD, ih = ih0 + hh * C::GSH, iw = iw0 + hw * C::GSW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < C::HP * Q && hw < C::HW - C::WPAD && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
        v = buffer_load4(xrsrc, (__umul24(__umul24(__umul24(hd * C::GSD, p.H) + hh * C::GSH, p.W) + hw * C::GSW, p.Cin) + (unsigned)(q * 4)) * 4u,
                         (unsigned)c0 * 4u);                     // relative to the tile's halo origin (scalar 64-bit, in the descriptor)
      stage[j] = v;
    }
  };
  // ---- weight fragments (B): straight from global/L2 into VGPRs, one ROW of taps (kw = 0..KW-1) at a time, fetched
  // a whole row ahead of its use -- across Cin-chunk boundaries too -- so that the L2 latency stays hidden even when
  // a tap is only a handful of MFMAs (small tiles, pointwise convs)
  constexpr int RT = C::PAR ? 2 : C::kW;
  const int nrt = C::PAR ? nkw : C::kW;
  const int nrows = nkd * nkh;
  float bc[RT][C::KK], bn[RT][C::KK];
  auto load_row = [&](float (&dst)[RT][C::KK], int c0, int kd_, int kh_) {
    const float *wc = wlane + (size_t)c0 * p.Cout;
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (t < nrt) {
        const float *wt = wc + (size_t)wtap(kd_, kh_, t) * p.Cin * p.Cout;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) dst[t][kk] = wt[(size_t)kk * C::KSTEP * p.Cout];
      }
  };
  auto mma_row = [&](const float (&bw)[RT][C::KK], int kd_, int kh_) {
    if constexpr (C::WINO == 4) {
      const int rowoff = (kd_ * C::HH + kh_) * C::HW * C::ROW;
#pragma unroll
      for (int kk = 0; kk < C::KK; ++kk) {
        const float g0 = bw[0][kk], g1 = bw[1][kk], g2 = bw[2][kk];
        const float s02 = g0 + g2, h = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), q = g1 * (1.f / 12.f);
        const float u[6] = {0.25f * g0, (s02 + g1) * (-1.f / 6.f), (s02 - g1) * (-1.f / 6.f), h + q, h - q, g2};
        float a[C::MT][6];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          const float *src = lds + abase[mt] + rowoff + kk * C::KSTEP;
          const float d0 = src[0], d1 = src[C::ROW], d2 = src[2 * C::ROW], d3 = src[3 * C::ROW], d4 = src[4 * C::ROW], d5 = src[5 * C::ROW];
          const float e = d4 - d2, f4 = 4.f * d2, o = d3 - d1;       // shared terms
          a[mt][0] = 4.f * d0 - 5.f * d2 + d4;
          a[mt][1] = (d3 + d4) - 4.f * (d1 + d2);
          a[mt][2] = (d4 - d3) + 4.f * (d1 - d2);
          a[mt][3] = e + 2.f * o;
          a[mt][4] = e - 2.f * o;
          a[mt][5] = 4.f * d1 - 5.f * d3 + d5;
          (void)f4;
        }
#pragma unroll
        for (int f = 0; f < 6; ++f)
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            if constexpr (C::k16)
              acc[f * C::MT + mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][f], u[f], acc[f * C::MT + mt], 0, 0, 0);
            else
              acc[f * C::MT + mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][f], u[f], acc[f * C::MT + mt], 0, 0, 0);
          }
      }
      return;
    }
    if constexpr (C::WINO == 2) {
      const int rowoff = (kd_ * C::HH + kh_) * C::HW * C::ROW;
#pragma unroll
      for (int kk = 0; kk < C::KK; ++kk) {
        const float g0 = bw[0][kk], g1 = bw[1][kk], g2 = bw[2][kk];
        const float u[4] = {g0, 0.5f * ((g0 + g2) + g1), 0.5f * ((g0 + g2) - g1), g2};
        float a[C::MT][4];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          const float *src = lds + abase[mt] + rowoff + kk * C::KSTEP;
          const float d0 = src[0], d1 = src[C::ROW], d2 = src[2 * C::ROW], d3 = src[3 * C::ROW];
          a[mt][0] = d0 - d2; a[mt][1] = d1 + d2; a[mt][2] = d2 - d1; a[mt][3] = d1 - d3;
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            if constexpr (C::k16)
              acc[f * C::MT + mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][f], u[f], acc[f * C::MT + mt], 0, 0, 0);
            else
              acc[f * C::MT + mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][f], u[f], acc[f * C::MT + mt], 0, 0, 0);
          }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (t < nrt) {
        const int tapoff = ((kd_ * C::HH + kh_) * C::HW + t) * C::ROW;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
          float a[C::MT];
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) a[mt] = lds[abase[mt] + tapoff + kk * C::KSTEP];
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            if constexpr (C::k16)
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bw[t][kk], acc[mt], 0, 0, 0);
            else
              acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bw[t][kk], acc[mt], 0, 0, 0);
          }
        }
      }
  };
  // one row: fetch the row after it (maybe the first row of the next chunk) into `nxt`, then multiply with `cur`
  auto step = [&](const float (&cur)[RT][C::KK], float (&nxt)[RT][C::KK], int c0, int &kd_, int &kh_) {
    int kh2 = kh_ + 1, kd2 = kd_, c2 = c0;
    if (kh2 == nkh) { kh2 = 0; if (++kd2 == nkd) { kd2 = 0; c2 = c0 + C::CC; } }
    if (c2 < p.Cin) load_row(nxt, c2, kd2, kh2);
    mma_row(cur, kd_, kh_);
    kd_ = kd2; kh_ = kh2;
  };

  if constexpr (!C::BLDS) load_row(bc, 0, 0, 0);
  if constexpr (!C::NOPF) fetch(0);
  for (int c0 = 0; c0 < p.Cin; c0 += C::CC) {
    __syncthreads();
    if constexpr (C::NOPF) fetch(c0);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int it = tid + j * kThreads;
      if (it < C::HP * Q && !(MR3D_FWD_ABL & 2)) {
        const int pos = it / Q, q = it - pos * Q;
        float *d = lds + pos * C::ROW + q * 4;
        d[0] = stage[j].x; d[1] = stage[j].y; d[2] = stage[j].z; d[3] = stage[j].w;
      }
    }
    if constexpr (C::BLDS) {
#pragma unroll
      for (int j = 0; j < NWV; ++j) {
        const int e = tid + j * kThreads;
        if (e < WQ && !(MR3D_FWD_ABL & 2)) *reinterpret_cast<float4 *>(wl + e * 4) = wstage[j];
      }
    }
    __syncthreads();
    if constexpr (!C::ONE && !C::NOPF) {
      if (c0 + C::CC < p.Cin) fetch(c0 + C::CC);
    }

    if constexpr (C::BLDS) {
      // ---- K loop with both operands in LDS.  The operand pairs of a tap row (kW taps x KK k-steps) are read ONE ROW AHEAD into a second
      // register set (round 6): written as `a = lds[..]; mfma(a, b)` the compiler kept every A fragment in the same register pair, so
      // each pair of MFMAs waited for its own LDS read -- ~80 cycles per 32-cycle MFMA with the one or two waves per SIMD these
      // launches have (layer3 / layer4 / P4 / P5: 64-position tiles).
      static_assert(C::MT == 1, "64-row tiles: one MFMA tile per wave");
      float a0[RT][C::KK], b0[RT][C::KK], a1[RT][C::KK], b1[RT][C::KK];
      auto ld = [&](float (&av)[RT][C::KK], float (&bv)[RT][C::KK], int kd, int kh) {
        if (MR3D_FWD_ABL & 8) {
#pragma unroll
          for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) { av[t][kk] = (float)(kd + t); bv[t][kk] = (float)(kh + kk); }
          return;
        }
#pragma unroll
        for (int t = 0; t < RT; ++t)
          if (t < nrt) {
            const int tapoff = ((kd * C::HH + kh) * C::HW + t) * C::ROW;
            const float *wt = wl + (wtap(kd, kh, t) * C::CC + lk) * C::NB + wave_n * C::TM + li;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) {
              bv[t][kk] = wt[kk * C::KSTEP * C::NB];
              av[t][kk] = lds[abase[0] + tapoff + kk * C::KSTEP];
            }
          }
      };
      auto mm = [&](const float (&av)[RT][C::KK], const float (&bv)[RT][C::KK]) {
        if (MR3D_FWD_ABL & 1) { acc[0][0] += av[0][0] * bv[0][0]; return; }
#pragma unroll
        for (int t = 0; t < RT; ++t)
          if (t < nrt) {
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) {
              if constexpr (C::k16)
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][kk], bv[t][kk], acc[0], 0, 0, 0);
              else
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][kk], bv[t][kk], acc[0], 0, 0, 0);
            }
          }
      };
      int kd = 0, kh = 0;
      auto next = [&]() { if (++kh == nkh) { kh = 0; ++kd; } };
      ld(a0, b0, 0, 0);
      int row = 0;
      for (; row + 1 < nrows; row += 2) {
        next();
        ld(a1, b1, kd, kh);
        __builtin_amdgcn_sched_barrier(0);                     // the next row's reads are in flight under this row's MFMAs
        mm(a0, b0);
        next();
        if (row + 2 < nrows) ld(a0, b0, kd, kh);
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1);
      }
      if (row < nrows) mm(a0, b0);
      continue;
    }

    // ---- K loop over tap rows, ping-ponging the two fragment buffers
    int kd = 0, kh = 0;
    int row = 0;
    for (; row + 1 < nrows; row += 2) {
      step(bc, bn, c0, kd, kh);
      step(bn, bc, c0, kd, kh);
    }
    if (row < nrows) {        // odd row count: the chunk ends on `bn`; move it back
      step(bc, bn, c0, kd, kh);
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) bc[t][kk] = bn[t][kk];
    }
  }

  // ---- epilogue: shift, residual adds, nearest-upsample add, ReLU / ReLU-mask; NDHWC store
  const int co = n0 + wave_n * C::TM + li;
  const float sh = p.shift ? p.shift[co] : 0.f;
  // Output addressing: a SCALAR 64-bit index of the tile's first output + a 32-bit byte offset per element (three 32-bit products of
  // tile-local coordinates with scalar steps).  One size_t product chain per element was 25 vector instructions, five of them
  // quarter-rate multiplies -- 3,000 per wave in the strided / parity instances, and vector ALU work does not hide under fp32 MFMAs.
  const int sd_ = C::PAR ? 2 : p.ysd, sh_ = C::PAR ? 2 : p.ysh, sw_ = C::PAR ? 2 : p.ysw;
  const size_t ybase = ((((size_t)n * p.yD + (size_t)(od0 * sd_ + (C::PAR ? ad : 0))) * p.yH + (size_t)(oh0 * sh_ + (C::PAR ? ah : 0))) * p.yW +
                        (size_t)(ow0 * sw_ + (C::PAR ? aw : 0))) * p.Cout;
  const int xstep = sw_ * p.Cout * 4, rstep = sh_ * p.yW * p.Cout * 4, pstep = sd_ * p.yH * p.yW * p.Cout * 4;       // bytes (a tile spans < 2^31)
  const char *add1b = p.add1 ? reinterpret_cast<const char *>(p.add1 + ybase) : nullptr;
  const char *add2b = p.add2 ? reinterpret_cast<const char *>(p.add2 + ybase) : nullptr;
  const char *maskb = p.mask ? reinterpret_cast<const char *>(p.mask + ybase) : nullptr;
  char *yb = reinterpret_cast<char *>(p.y + ybase);
  // rel_of: byte offset of tile-local output (td, th, tw).  Accumulator elements come in groups of four consecutive GEMM rows that
  // share (td, th) whenever a tile row holds a multiple of four GEMM rows: their offsets are one product set + scalar steps.
  auto rel_of = [&](int td, int th, int tw) { return (unsigned)(td * pstep + th * rstep + tw * xstep + co * 4); };
  auto emit = [&](unsigned rel, int od, int oh, int ow, bool inside, float val) {
    if constexpr (C::PAR) inside = inside && 2 * od + ad < p.yD && 2 * oh + ah < p.yH && 2 * ow + aw < p.yW;
    if (inside) {
      float v = val + sh;
      if (add1b) v += *reinterpret_cast<const float *>(add1b + rel);
      if (add2b) v += *reinterpret_cast<const float *>(add2b + rel);
      if (p.up) {
        if (p.up_even) {
          if (!((od | oh | ow) & 1)) v += p.up[((((size_t)n * p.Du + (od >> 1)) * p.Hu + (oh >> 1)) * p.Wu + (ow >> 1)) * p.Cout + co];
        } else {
          const int ud = nearest_src(od, p.Du, p.Do), uh = nearest_src(oh, p.Hu, p.Ho), uw = nearest_src(ow, p.Wu, p.Wo);
          v += p.up[((((size_t)n * p.Du + ud) * p.Hu + uh) * p.Wu + uw) * p.Cout + co];
        }
      }
      if (p.relu) v = fmaxf(v, 0.f);
      if (maskb) v = *reinterpret_cast<const float *>(maskb + rel) > 0.f ? v : 0.f;
      *reinterpret_cast<float *>(yb + rel) = v;
    }
  };
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    unsigned rel_g = 0;
#pragma unroll
    for (int r = 0; r < (C::k16 ? 4 : 16); ++r) {
      const int row = C::k16 ? (lk * 4 + r) : ((r & 3) + 8 * (r >> 2) + 4 * lk);
      const int m = wave_m * C::WROWS + mt * C::TM + row;
      constexpr int RW = C::TW / C::OPR;
      const int tw = (m % RW) * C::OPR, th = (m / RW) % C::TH, td = m / (RW * C::TH);
      const int od = od0 + td, oh = oh0 + th, ow = ow0 + tw;
      const bool inside = od < p.Do && oh < p.Ho && ow < p.Wo && (C::MV == C::M || m < C::MV);
      unsigned rel;
      if constexpr (RW % 4 == 0) {
        if ((r & 3) == 0) rel_g = rel_of(td, th, tw);
        rel = rel_g + (unsigned)((r & 3) * C::OPR * xstep);
      } else {
        rel = rel_of(td, th, tw);
      }
      if constexpr (C::WINO == 4) {
        const float m0 = acc[mt][r], m1 = acc[C::MT + mt][r], m2 = acc[2 * C::MT + mt][r], m3 = acc[3 * C::MT + mt][r],
                    m4 = acc[4 * C::MT + mt][r], m5 = acc[5 * C::MT + mt][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        emit(rel, od, oh, ow, inside, (m0 + s12) + s34);
        emit(rel + xstep, od, oh, ow + 1, inside && ow + 1 < p.Wo, d12 + 2.f * d34);
        emit(rel + 2 * xstep, od, oh, ow + 2, inside && ow + 2 < p.Wo, s12 + 4.f * s34);
        emit(rel + 3 * xstep, od, oh, ow + 3, inside && ow + 3 < p.Wo, (d12 + 8.f * d34) + m5);
      } else if constexpr (C::WINO == 2) {
        const float m0 = acc[mt][r], m1 = acc[C::MT + mt][r], m2 = acc[2 * C::MT + mt][r], m3 = acc[3 * C::MT + mt][r];
        emit(rel, od, oh, ow, inside, (m0 + m1) + m2);
        emit(rel + xstep, od, oh, ow + 1, inside && ow + 1 < p.Wo, (m1 - m2) - m3);
      } else {
        emit(rel, od, oh, ow, inside, acc[mt][r]);
      }
    }
  }
}


// mr3d_conv3d_forward_variant: the dispatch below runs unchanged but records the instance it would launch
thread_local mr3d_conv3d_variant *g_variant_probe = nullptr;

template <class C>
int launch_fwd(const Conv3dFwdArgs &a, hipStream_t stream) {
  if (g_variant_probe) {
    mr3d_conv3d_variant &v = *g_variant_probe;
    v.streaming_pointwise = 0;
    v.winograd = C::WINO; v.tile_d = C::TD; v.tile_h = C::TH; v.tile_w = C::TW; v.cin_chunk = C::CC; v.cout_tile = C::NB;
    v.parity_decomposed = C::PAR ? 1 : 0;
    // MFMA work issued per algorithmic FLOP: F(2,3) 4 products per 2 outputs x 3 taps, F(4,3) 6 per 4 x 3
    v.mfma_issue_factor = C::WINO == 4 ? 0.5f : (C::WINO == 2 ? 2.f / 3.f : 1.f);
    return MR3D_OK;
  }
  const int tiles_d = (a.Do + C::TD - 1) / C::TD, tiles_h = (a.Ho + C::TH - 1) / C::TH, tiles_w = (a.Wo + C::TW - 1) / C::TW;
  static bool attr_done = false;   // LDS above 64 KiB needs the opt-in attribute (one CU has 160 KiB)
  if (!attr_done && C::LDS_BYTES > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_fwd_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  dim3 grid((unsigned)(a.N * tiles_d * tiles_h * tiles_w), (unsigned)(a.Cout / C::NB), C::PAR ? 8u : 1u);
  conv3d_fwd_kernel<C><<<grid, kThreads, C::LDS_BYTES, stream>>>(a, tiles_d, tiles_h, tiles_w);
  return mr3d_launch_status();
}

// Output tile choice.  256 positions (4x8x8) amortise the halo best; small feature maps (layer3/4, P4/P5) would
// then leave most of the 256 CUs idle, so they take 128- or 64-position tiles: the largest tile that still
// yields >= 2 workgroups per CU, else the smallest one.
template <int K, int SD, int SH, int SW, int CC, int NB>
int dispatch_tile(const Conv3dFwdArgs &a, hipStream_t stream) {
  auto wgs = [&](int td, int th, int tw) {
    return (long long)a.N * ((a.Do + td - 1) / td) * ((a.Ho + th - 1) / th) * ((a.Wo + tw - 1) / tw) * (a.Cout / NB);
  };
  using T0 = FwdCfg<K, K, K, SD, SH, SW, 4, 8, 8, CC, NB>;
  using T1 = FwdCfg<K, K, K, SD, SH, SW, 2, 8, 8, CC, NB>;
  if constexpr (K == 3 && SD == 1 && SH == 1 && SW == 1) {
    // volumes whose extents are multiples of 5/7/7 (10x14x14 mask-head rois): a 5x7x7 tile wastes 4 % of the GEMM
    // rows where the power-of-two tiles waste 23-36 %
    const long long vol = (long long)a.Do * a.Ho * a.Wo;
    const long long t577 = (long long)((a.Do + 4) / 5) * ((a.Ho + 6) / 7) * ((a.Wo + 6) / 7);
    const long long t488 = (long long)((a.Do + 3) / 4) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8);
    const long long t288 = (long long)((a.Do + 1) / 2) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8);
    const double c577 = (double)vol / (t577 * 256.0), c488 = (double)vol / (t488 * 256.0), c288 = (double)vol / (t288 * 128.0);
    // (the F(4,3) form on padded 4x8x8 tiles was measured equal on these volumes: 463 vs 463 us for 128 rois)
    // large roi batches (mask heads on >= ~52 rois): the two-dimensional Winograd kernel (2x8x8 tiles on a 10x14x14 volume use
    // 77 % of their rows, at 1/3 of the MFMA work) instead of the direct 5x7x7 tiles: 128-row step 26.3 -> 24.9 ms; on 16-row
    // batches the two are equal
    constexpr int roi2d_min = 1024;
    constexpr int wino_roi = 4;
    const bool skip577 = wino_roi >= 4 && NB == 64 && CC == 16 && a.N * t288 >= roi2d_min && mr3d_conv3d_wino2d_ok(a);
    if (!skip577 && c577 > 1.15 * (c488 > c288 ? c488 : c288) && a.N * t577 * (a.Cout / NB) >= 512)
      return launch_fwd<FwdCfg<3, 3, 3, 1, 1, 1, 5, 7, 7, CC, NB>>(a, stream);
  }
  if constexpr (K == 3 && SD == 1 && SH == 1 && SW == 1 && CC == 16 && (NB == 16 || NB == 32)) {
    // narrow Cout (layer1 16 -> 16, layer2 32 -> 32 on the large maps): F(4,3) x F(2,3) with the waves split over GEMM rows
    constexpr int wino_r = 4;
    constexpr int min2d_r = 256;   // backbone step 11.83 -> 11.69 (layer1) -> 11.64 ms (+ layer2)
    if (wino_r >= 4 && a.Cout == NB && wgs(NB == 16 ? 8 : 4, 8, 8) >= min2d_r && mr3d_conv3d_wino2d_rows_ok(a)) {
      if (g_variant_probe) {
        mr3d_conv3d_variant &v = *g_variant_probe;
        v.streaming_pointwise = 0; v.winograd = 8; v.tile_d = NB == 16 ? 8 : 4; v.tile_h = 8; v.tile_w = 8; v.cin_chunk = 16; v.cout_tile = NB;
        v.parity_decomposed = 0; v.mfma_issue_factor = 1.f / 3.f;
        return MR3D_OK;
      }
      return mr3d_conv3d_wino2d_rows(a, stream);
    }
  }
  if constexpr (K == 3 && SD == 1 && SH == 1 && SW == 1 && NB == 16 && CC == 16) {
    if (a.Cin == CC && wgs(4, 8, 8) >= 512) return launch_fwd<FwdCfg<3, 3, 3, 1, 1, 1, 4, 8, 8, CC, NB, false, true>>(a, stream);
  }
  if constexpr (K == 3 && SD == 1 && SH == 1 && SW == 1 && CC == 16 && NB == 64) {
    // Winograd F(2,3) along W (see FwdCfg): 4 MFMAs per output pair and tap row instead of 6.  64-wide tile: 2x8x8
    // outputs = 64 pair rows keep the accumulators at 64 registers and two workgroups per CU (P2 3x3x3 64->64:
    // 1.19 -> 1.02 ms; the 4x8x8 form needs 128 accumulator registers, one workgroup per CU: 1.11 ms).
    // The 16-wide single-chunk kernel (layer1) gains nothing from it (96 vs 96 TFLOP/s: not MFMA-bound).
    constexpr int wino = 4;     // 0: direct form only; 1: F(2,3) 2x8x8 only; 2: + F(2,3) 4x8x8; 3: F(4,3) 4x8x8; 4: + F(4,3) x F(2,3) on large maps
    // 4x8x8 outputs = 128 pair rows: 128 accumulator registers; without the next-chunk register prefetch it still fits
    // two workgroups per CU, which hide each other's staging: P2 0.91 ms (159 TFLOP/s of algorithmic work)
    constexpr int min43 = 256;   // P3 (320 tiles): 0.164 -> 0.157 ms
    // 4: F(4,3) along W x F(2,3) along H (conv3d_wino2d.hip) on maps with >= min2d tiles of 2x8x8: P2 64->64 834 -> 640-660 us,
    // backbone step 12.52 -> 11.8 ms (P3, 640 tiles, included)
    constexpr int min2d = 256;
    if (wino >= 4 && wgs(2, 8, 8) >= min2d && mr3d_conv3d_wino2d_ok(a)) {
      if (g_variant_probe) {
        mr3d_conv3d_variant &v = *g_variant_probe;
        const bool wide = mr3d_conv3d_wino2d_wide(a);
        v.streaming_pointwise = 0; v.winograd = 8; v.tile_d = wide ? 4 : 2; v.tile_h = 8; v.tile_w = 8; v.cin_chunk = wide ? 8 : 16; v.cout_tile = 64;
        v.parity_decomposed = 0; v.mfma_issue_factor = 1.f / 3.f;
        return MR3D_OK;
      }
      return mr3d_conv3d_wino2d(a, stream);
    }
    if (wino >= 3 && wgs(4, 8, 8) >= min43) return launch_fwd<FwdCfg<3, 3, 3, 1, 1, 1, 4, 8, 8, CC, NB, false, false, 4>>(a, stream);
    if (wino >= 2 && wgs(4, 8, 8) >= 512) return launch_fwd<FwdCfg<3, 3, 3, 1, 1, 1, 4, 8, 8, CC, NB, false, false, 2>>(a, stream);
    constexpr int wino_min = 256;   // 16-roi mask batches (320 tiles): 133 -> 109 us
    if (wino && wgs(2, 8, 8) >= wino_min) return launch_fwd<FwdCfg<3, 3, 3, 1, 1, 1, 2, 8, 8, CC, NB, false, false, 2>>(a, stream);
  }
  if constexpr (K == 3 && SD == 1 && SH == 1 && SW == 1 && CC == 16 && NB == 32) {
    // 32-wide Cout tile (layer2 32->32 on 20x64x64): F(2,3) on 4x8x8 tiles, 128 pair rows: 67 -> 57 us
    constexpr int wino32 = 3;
    if (wino32 && wgs(4, 8, 8) >= 256) return launch_fwd<FwdCfg<3, 3, 3, 1, 1, 1, 4, 8, 8, CC, NB, false, false, 2>>(a, stream);
  }
  if (wgs(4, 8, 8) >= 512) return launch_fwd<T0>(a, stream);
  if constexpr (NB == 32) {
    return launch_fwd<T1>(a, stream);
  } else {
    using T2 = FwdCfg<K, K, K, SD, SH, SW, 2, 4, 8, CC, NB>;
    if (wgs(2, 8, 8) >= 512) return launch_fwd<T1>(a, stream);
    // Round 6, what the 3x3x3 launches of layer3 (10x32x32, 64 -> 64: 640 workgroups) and layer4 (5x16x16, 128 -> 128: 160) wait for,
    // by timing-only ablations (tools/diag/fwd_abl.sh; back-to-back launches through ctypes cost 13 us each with an empty kernel):
    // 34 / 27 us as built; without the MFMAs 14 / 13; without the LDS staging writes and the global fetch 28 / 21; without the LDS
    // operand reads 29 / 22.  A wave's 432 / 864 MFMAs are 14.4 / 11.5 us of matrix-pipe time per SIMD (2.5 / 1 waves per SIMD): these
    // launches run at 72-82 % of the direct form's MFMA time plus the launch skeleton.  32 input channels per chunk (87 KB of LDS, half
    // the chunk passes) was measured too: layer3 33 -> 47 us (one workgroup per CU), layer4 30 -> 30.  Fewer multiplies would need
    // a Winograd form on 16-wide Cout tiles, whose input transform is then formed four times (9.6 vector instructions per MFMA in
    // conv3d_wino2d_rows_kernel<1>): not built.
    return launch_fwd<T2>(a, stream);
  }
}

// Small feature maps (layer3/4, P4/P5): with the widest Cout tile even the smallest position tile yields fewer
// workgroups than the chip has CUs and every one of them walks the whole K extent alone.  Narrow the Cout tile
// (64 -> 32 -> 16) until there are at least as many workgroups as CUs; the halo is re-staged per Cout tile, which is
// cheap exactly when the map is small.
inline int narrow_nb(const Conv3dFwdArgs &a, int nb) {
  auto wgs = [&](int nb_) {
    const int th = nb_ == 32 ? 8 : 4;      // smallest position tile of that Cout tile: (2,8,8) for 32, else (2,4,8)
    return (long long)a.N * ((a.Do + 1) / 2) * ((a.Ho + th - 1) / th) * ((a.Wo + 7) / 8) * (a.Cout / nb_);
  };
  constexpr int min_wgs = 256;
  while (nb > 16 && wgs(nb) < min_wgs) nb /= 2;
  return nb;
}

template <int K, int SD, int SH, int SW, int CC>
int dispatch_nb(const Conv3dFwdArgs &a, int nb, hipStream_t stream) {
  nb = narrow_nb(a, nb);
  switch (nb) {
    case 16: return dispatch_tile<K, SD, SH, SW, CC, 16>(a, stream);
    case 32: return dispatch_tile<K, SD, SH, SW, CC, 32>(a, stream);
    case 64: return dispatch_tile<K, SD, SH, SW, CC, 64>(a, stream);
  }
  return MR3D_ERR_UNSUPPORTED;
}

// stride-2 3x3x3: the 4x8x8 tile's 9x17x17 halo only fits LDS with 8 channels per chunk; the smaller tiles take 16
template <int NB>
int dispatch_s2(const Conv3dFwdArgs &a, hipStream_t stream) {
  auto wgs = [&](int td, int th, int tw) {
    return (long long)a.N * ((a.Do + td - 1) / td) * ((a.Ho + th - 1) / th) * ((a.Wo + tw - 1) / tw) * (a.Cout / NB);
  };
  if (a.Cin % 16 != 0 || wgs(4, 8, 8) >= 1024) return launch_fwd<FwdCfg<3, 3, 3, 2, 2, 2, 4, 8, 8, 8, NB>>(a, stream);
  if constexpr (NB == 32) {
    return launch_fwd<FwdCfg<3, 3, 3, 2, 2, 2, 2, 8, 8, 16, NB>>(a, stream);
  } else {
    if (wgs(2, 8, 8) >= 512) return launch_fwd<FwdCfg<3, 3, 3, 2, 2, 2, 2, 8, 8, 16, NB>>(a, stream);
    return launch_fwd<FwdCfg<3, 3, 3, 2, 2, 2, 2, 4, 8, 16, NB>>(a, stream);
  }
}

// Parity-decomposed data gradient of a stride-2, pad-1 3x3x3 convolution.  Along each axis dx[2j] only meets the
// centre tap (dy[j] * W[1]) and dx[2j+1] the two outer ones (dy[j] * W[2] + dy[j+1] * W[0]), so the 8 output parity
// classes are stride-1 correlations of the COARSE dy with 1..8 of the 27 taps (27 tap-visits for 8 outputs instead
// of the 216 of a zero-stuffed dense pass).  One launch: blockIdx.z = class, outputs scattered to 2*o + parity.
template <int NB>
int dispatch_par(const Conv3dFwdArgs &a, hipStream_t stream) {
  auto wgs = [&](int td, int th, int tw) {
    return 8LL * a.N * ((a.Do + td - 1) / td) * ((a.Ho + th - 1) / th) * ((a.Wo + tw - 1) / tw) * (a.Cout / NB);
  };
  if (wgs(4, 8, 8) >= 1024) return launch_fwd<FwdCfg<2, 2, 2, 1, 1, 1, 4, 8, 8, 16, NB, true>>(a, stream);
  if constexpr (NB == 32) {
    return launch_fwd<FwdCfg<2, 2, 2, 1, 1, 1, 2, 8, 8, 16, NB, true>>(a, stream);
  } else {
    if (wgs(2, 8, 8) >= 1024) return launch_fwd<FwdCfg<2, 2, 2, 1, 1, 1, 2, 8, 8, 16, NB, true>>(a, stream);
    return launch_fwd<FwdCfg<2, 2, 2, 1, 1, 1, 2, 4, 8, 16, NB, true>>(a, stream);
  }
}

}  // namespace

// Cout tile: 64 when Cout is a multiple of 64, else 32, else 16.
static int pick_nb(int cout) { return cout % 64 == 0 ? 64 : (cout % 32 == 0 ? 32 : (cout % 16 == 0 ? 16 : 0)); }

extern "C" int mr3d_conv3d_forward(const mr3d_conv3d_desc *d, mr3d_stream_t stream_) {
  if (!d || !d->x || !d->w || !d->y) return MR3D_ERR_INVALID_ARG;
  if (d->kd != d->kh || d->kh != d->kw) return MR3D_ERR_UNSUPPORTED;
  Conv3dFwdArgs a;
  a.x = d->x; a.w = d->w; a.y = d->y; a.shift = d->shift; a.add1 = d->add1; a.add2 = d->add2; a.up = d->up; a.mask = d->mask;
  a.N = d->n; a.D = d->d; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Cout = d->cout;
  a.pd = d->pd; a.ph = d->ph; a.pw = d->pw;
  a.Do = (d->d + 2 * d->pd - d->kd) / d->sd + 1;
  a.Ho = (d->h + 2 * d->ph - d->kh) / d->sh + 1;
  a.Wo = (d->w_in + 2 * d->pw - d->kw) / d->sw + 1;
  a.Du = d->up_d; a.Hu = d->up_h; a.Wu = d->up_w;
  a.relu = d->relu & MR3D_CONV_RELU; a.up_even = (d->relu & MR3D_CONV_UP_EVEN) ? 1 : 0;
  a.ysd = d->ys_d > 0 ? d->ys_d : 1; a.ysh = d->ys_h > 0 ? d->ys_h : 1; a.ysw = d->ys_w > 0 ? d->ys_w : 1;
  a.yD = d->y_d > 0 ? d->y_d : a.Do; a.yH = d->y_h > 0 ? d->y_h : a.Ho; a.yW = d->y_w > 0 ? d->y_w : a.Wo;
  a.ws = (float *)d->workspace; a.ws_bytes = d->workspace_bytes;
  a.x_live = d->x_live;
  if (a.ws && ((uintptr_t)a.ws & 255)) return MR3D_ERR_WORKSPACE;
  if ((a.Do - 1) * a.ysd >= a.yD || (a.Ho - 1) * a.ysh >= a.yH || (a.Wo - 1) * a.ysw >= a.yW) return MR3D_ERR_INVALID_ARG;
  if (a.N <= 0 || a.Do <= 0 || a.Ho <= 0 || a.Wo <= 0) return MR3D_ERR_INVALID_ARG;
  if (a.up && (a.Du <= 0 || a.Hu <= 0 || a.Wu <= 0)) return MR3D_ERR_INVALID_ARG;
  if (a.up && a.up_even && (a.Du != (a.Do + 1) / 2 || a.Hu != (a.Ho + 1) / 2 || a.Wu != (a.Wo + 1) / 2 || a.ysd != 1 || a.ysh != 1 || a.ysw != 1))
    return MR3D_ERR_INVALID_ARG;
  // Ranges of the 32-bit / 24-bit offset arithmetic inside the kernels (halo staging through buffer loads, tile-relative output
  // offsets): an input plane below 2^20 positions and 2^24 elements, an output plane (times its scatter stride) below 2^25
  // elements, weights below 2^29 elements.  Far above anything a 512 x 512 x 80 volume produces behind the stem.
  // (the streaming pointwise kernels index with 64 bits: they are tried first, in front of the guard)
  const bool offsets_fit = !((long long)a.H * a.W >= (1LL << 20) || (long long)a.H * a.W * a.Cin >= (1LL << 24) ||
                             (long long)a.yH * a.yW * a.Cout * a.ysd >= (1LL << 25) ||
                             (long long)d->kd * d->kh * d->kw * a.Cin * a.Cout >= (1LL << 29));
  const int nb = pick_nb(a.Cout);
  if (!nb) return MR3D_ERR_UNSUPPORTED;
  hipStream_t stream = mr3d_stream(stream_);
  const int k = d->kd, s = d->sd * 100 + d->sh * 10 + d->sw;
  if (k == 1 && s == 111) {
    if (a.pd == 0 && a.ph == 0 && a.pw == 0) {
      const int st = mr3d_pointwise_fwd_try(a, g_variant_probe ? nullptr : stream, g_variant_probe != nullptr);
      if (st != MR3D_ERR_UNSUPPORTED) {
        if (g_variant_probe && st == MR3D_OK) {
          *g_variant_probe = mr3d_conv3d_variant{};
          g_variant_probe->streaming_pointwise = 1; g_variant_probe->mfma_issue_factor = 1.f;
          g_variant_probe->cin_chunk = a.Cin; g_variant_probe->cout_tile = a.Cout;
        }
        return st;
      }
    }
    if (!offsets_fit) return MR3D_ERR_UNSUPPORTED;
    if (a.Cin % 64 == 0) return dispatch_nb<1, 1, 1, 1, 64>(a, nb, stream);
    if (a.Cin % 32 == 0) return dispatch_nb<1, 1, 1, 1, 32>(a, nb, stream);
    if (a.Cin % 16 == 0) return dispatch_nb<1, 1, 1, 1, 16>(a, nb, stream);
  } else if (!offsets_fit) {
    return MR3D_ERR_UNSUPPORTED;
  } else if (k == 1 && s == 222) {
    if (a.Cin % 64 == 0) return dispatch_nb<1, 2, 2, 2, 64>(a, nb, stream);
    if (a.Cin % 16 == 0) return dispatch_nb<1, 2, 2, 2, 16>(a, nb, stream);
  } else if (k == 3 && s == 111) {
    if (a.Cin % 16 == 0) return dispatch_nb<3, 1, 1, 1, 16>(a, nb, stream);
  } else if (k == 3 && s == 222) {
    if (a.Cin % 8 == 0) {
      switch (narrow_nb(a, nb)) {
        case 16: return dispatch_s2<16>(a, stream);
        case 32: return dispatch_s2<32>(a, stream);
        case 64: return dispatch_s2<64>(a, stream);
      }
    }
  } else if (k == 7 && s == 122) {
    if (a.Cin == 4 && nb == 16) return launch_fwd<FwdCfg<7, 7, 7, 1, 2, 2, 4, 8, 8, 4, 16>>(a, stream);
  } else if (k == 2 && s == 222 && a.pd == 0 && a.ph == 0 && a.pw == 0) {
    // kernel 2 / stride 2 (non-overlapping windows): the data gradient of ConvTranspose3d(k=2, s=2)
    // (fcn_mask_head_3d.py:60-66 `upsample`).  The halo is exactly 2T per axis: 8 channels per chunk keep it in LDS.
    if (a.Cin % 8 == 0 && nb == 64) {
      const long long t577 = (long long)((a.Do + 4) / 5) * ((a.Ho + 6) / 7) * ((a.Wo + 6) / 7);
      const long long t488 = (long long)((a.Do + 3) / 4) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8);
      if (t577 < t488) {
        // 2x7x7 outputs (98 of 128 GEMM rows) with 16 channels per chunk beat 5x7x7 with 8 (292 vs 473 us on the
        // 128-roi mask-head shape): half as many chunk passes over the 2x-sized halo, three workgroups per CU
        if (a.Cin % 16 == 0) return launch_fwd<FwdCfg<2, 2, 2, 2, 2, 2, 2, 7, 7, 16, 64>>(a, stream);
        return launch_fwd<FwdCfg<2, 2, 2, 2, 2, 2, 5, 7, 7, 8, 64>>(a, stream);
      }
      return launch_fwd<FwdCfg<2, 2, 2, 2, 2, 2, 4, 8, 8, 8, 64>>(a, stream);
    }
  }
  return MR3D_ERR_UNSUPPORTED;
}

extern "C" int mr3d_conv3d_dgrad_s2k3(const mr3d_conv3d_desc *d, mr3d_stream_t stream_) {
  if (!d || !d->x || !d->w || !d->y) return MR3D_ERR_INVALID_ARG;
  if (d->y_d <= 0 || d->y_h <= 0 || d->y_w <= 0 || d->shift || d->up || d->relu) return MR3D_ERR_INVALID_ARG;
  // dy must be the coarse grid of dx under kernel 3 / stride 2 / pad 1
  if (d->d != (d->y_d - 1) / 2 + 1 || d->h != (d->y_h - 1) / 2 + 1 || d->w_in != (d->y_w - 1) / 2 + 1) return MR3D_ERR_INVALID_ARG;
  if (d->n <= 0 || d->cin % 16 != 0) return MR3D_ERR_UNSUPPORTED;
  Conv3dFwdArgs a;
  a.x = d->x; a.w = d->w; a.y = d->y; a.shift = nullptr; a.add1 = d->add1; a.add2 = d->add2; a.up = nullptr; a.mask = d->mask;
  a.N = d->n; a.D = d->d; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Cout = d->cout;
  a.pd = a.ph = a.pw = 0;
  a.Do = d->d; a.Ho = d->h; a.Wo = d->w_in;          // class grid = dy's grid
  a.Du = a.Hu = a.Wu = 0; a.relu = 0;
  a.ysd = a.ysh = a.ysw = 2; a.yD = d->y_d; a.yH = d->y_h; a.yW = d->y_w;
  if ((long long)a.H * a.W >= (1LL << 20) || (long long)a.H * a.W * a.Cin >= (1LL << 24) ||
      (long long)a.yH * a.yW * a.Cout * 2 >= (1LL << 25) || 27LL * a.Cin * a.Cout >= (1LL << 29))
    return MR3D_ERR_UNSUPPORTED;                       // ranges of the kernels' 32-bit / 24-bit offset arithmetic (mr3d_conv3d_forward)
  hipStream_t stream = mr3d_stream(stream_);
  switch (narrow_nb(a, pick_nb(a.Cout))) {
    case 16: return dispatch_par<16>(a, stream);
    case 32: return dispatch_par<32>(a, stream);
    case 64: return dispatch_par<64>(a, stream);
  }
  return MR3D_ERR_UNSUPPORTED;
}

extern "C" int mr3d_conv3d_forward_variant(const mr3d_conv3d_desc *d, mr3d_conv3d_variant *out) {
  if (!d || !out) return MR3D_ERR_INVALID_ARG;
  *out = mr3d_conv3d_variant{};
  mr3d_conv3d_desc probe = *d;
  // the dispatch only looks at shapes; give it non-null operands so that the argument checks pass
  static float dummy;
  if (!probe.x) probe.x = &dummy;
  if (!probe.w) probe.w = &dummy;
  if (!probe.y) probe.y = &dummy;
  g_variant_probe = out;
  const int st = mr3d_conv3d_forward(&probe, nullptr);
  g_variant_probe = nullptr;
  return st;
}

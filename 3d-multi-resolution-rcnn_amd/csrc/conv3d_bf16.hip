// bf16 3D convolution forward / data gradient on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// -- BASELINE.json configs[4] ("bf16"): the same layers as conv3d_fwd.hip (mmdet/models/backbones/resnet3d.py:126,139,
// 173,251; utils/conv_module3d.py:34; necks/fpn3d.py:99-108 with the norm / ReLU / residual / upsample-add epilogue) with
// activations stored as bf16 NDHWC in HBM (half the bytes of every map) and 16x the fp32 matrix rate.
//
// GEMM view: M = output positions of a (TD x TH x TW) tile, N = Cout, K = taps x Cin.
//   * A: the tile's input halo for CC channels sits in LDS as [halo position][CC] bf16 (rows padded by 16 B); an
//     A fragment (8 consecutive channels of one position = 16 B) is ONE ds_read_b128, a tap is a constant LDS offset.
//   * B: weights pre-packed [tap][Cin/8][Cout32][8]: a lane's fragment (8 consecutive input channels of one output
//     channel) is one 16-B global load, 512 contiguous bytes per half-wave; fetched one tap ahead into registers.
//     Weights never touch LDS.
//   * a wave owns (M/128) x NT MFMA tiles of 32 x 32; 4 waves split the tile's positions.
//   * epilogue as in the fp32 kernel: + shift (folded BN / bias, fp32) + add1 + add2 + nearest-upsampled map, ReLU or
//     the ReLU mask of the layer a data gradient feeds; bf16 store (round to nearest even).
#include "conv3d.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
using bf16_t = uint16_t;

constexpr int kThreads = 256;

struct BfArgs {
  const bf16_t *x; const bf16_t *w; bf16_t *y;
  const float *shift; const bf16_t *add1; const bf16_t *add2; const bf16_t *up; const bf16_t *mask;
  int N, D, H, W, Cin;
  int Do, Ho, Wo, Cout, CoutPad;
  int pd, ph, pw;
  int Du, Hu, Wu;
  int relu;
  int ysd, ysh, ysw, yD, yH, yW;
  const unsigned char *x_live = nullptr;      // mr3d_voxel_liveness bitmap of x, or null
  int co_base = 0;                            // conv3d_bf16_c64_kernel: first output channel of this launch
};

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {          // round to nearest even, NaN stays NaN: v_cvt_pk_bf16_f32
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  int s = (int)floorf((float)dst * ((float)in / (float)out));      // F.interpolate(mode='nearest'), fpn3d.py:106-108
  return s < in - 1 ? s : in - 1;
}

// PAR_: the parity-decomposed data gradient of a kernel-3 / stride-2 / pad-1 convolution (mr3d_conv3d_dgrad_s2k3_bf16): K_ = 2 is
// then the per-axis MAXIMUM of taps an output class meets; blockIdx.z is the class, its 1..8 taps are walked, and the outputs go to
// positions 2 o + (class bit) of dx.
// PARF_: the FORWARD kernel-3 / stride-2 / pad-1 convolution from the 8 parity classes of its INPUT (x[2j + a] per axis): class a = 0
// meets tap 1 at j = o, class a = 1 taps 0 and 2 at j = o - 1, o -- every class is a dense 1..8-tap convolution on a halo of
// (T + 1)^3 positions read with step 2, walked one class after the other into the same accumulators.  (The direct form staged
// (2T + 1)^3 positions of 16 channels at a time: 2 x 8 x 8 outputs were all that fitted.)
template <int K_, int S_, int TD_, int TH_, int TW_, int CC_, int NT_, bool PAR_ = false, bool PARF_ = false>
struct BfCfg {
  static constexpr int K = K_, S = S_, TD = TD_, TH = TH_, TW = TW_, CC = CC_, NT = NT_;
  static constexpr bool PAR = PAR_, PARF = PARF_;
  static constexpr bool PW1 = (K_ == 1);
  static constexpr int LS = PW1 ? 1 : S_;          // LDS step per output position
  static constexpr int GS = PW1 ? S_ : 1;          // global step per halo position (a strided 1x1x1 reads every S-th voxel)
  static constexpr int HD = (TD - 1) * LS + K, HH = (TH - 1) * LS + K, HW = (TW - 1) * LS + K, HP = HD * HH * HW;
  // halo row pitch in LDS: the 16 lanes of a ds_read_b128 group hold positions of four tile rows, and with the plain pitch (10 for
  // the 3x3x3 / stride-1 tiles) three of them land on one 16-byte slot of the bank row (62 % of the LDS-active cycles were
  // conflicts); two spare positions per row leave 2-way (searched with the kernel's own address map)
  static constexpr int HWP = (K == 3 && S == 1 && TW == 8) ? HW + 2 : HW;
  static constexpr int HPP = HD * HH * HWP;
  static constexpr int ROWB = CC * 2 + 16;         // bytes per halo position (16 B of padding staggers the banks)
  static constexpr int M = TD * TH * TW;
  static constexpr int MTW = M / 128;              // 32-row MFMA tiles per wave
  static constexpr int KK = CC / 16;               // MFMA k-steps per tap and chunk
  static constexpr int TAPS = K * K * K;
  static constexpr int NB = 32 * NT;
  static constexpr int LDS_BYTES = (HPP * ROWB > M * 36 * 4) ? HPP * ROWB : M * 36 * 4;    // halo, later the epilogue's rows
  static constexpr int PIECES = HP * (CC / 8);     // 16-byte pieces of a halo chunk
  static constexpr int NIT = (PIECES + kThreads - 1) / kThreads;
  static_assert(M % 128 == 0 && CC % 16 == 0, "tile");
  static_assert(LDS_BYTES <= 160 * 1024, "halo exceeds LDS");
};

// LIVE (3x3x3 / stride 1): the caller lent a voxel liveness bitmap of x (mr3d_voxel_liveness) -- a tile whose halo holds no live voxel
// is written from its epilogue alone (data gradients of the sparse maps of detector training, cf. conv3d_wino2d_kernel<true>)
template <class C, bool LIVE = false>
__global__ __launch_bounds__(kThreads, 2) void conv3d_bf16_kernel(BfArgs p, int tiles_d, int tiles_h, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  int b = blockIdx.x;
  const int tw_i = b % tiles_w; b /= tiles_w;
  const int th_i = b % tiles_h; b /= tiles_h;
  const int td_i = b % tiles_d; b /= tiles_d;
  const int n = b;
  const int od0 = td_i * C::TD, oh0 = th_i * C::TH, ow0 = tw_i * C::TW;
  const int id0 = od0 * C::S - p.pd, ih0 = oh0 * C::S - p.ph, iw0 = ow0 * C::S - p.pw;
  const int n0 = blockIdx.y * C::NB;
  // PAR: parity class of the outputs this workgroup produces (the 8-tap class first); class-local tap t -> halo offset
  // (kd, kh, kw) in {0, 1}^3 and tap {1} / {0, 2} of the flipped 3x3x3 kernel per axis
  const int cls = C::PAR ? 7 - (int)blockIdx.z : 0;
  const int ad = (cls >> 2) & 1, ah = (cls >> 1) & 1, aw = cls & 1;
  const int tap_shift = ad + ah + aw;                       // PAR: the class walks 1 << tap_shift taps
  auto tap_of = [&](int c_, int t, int &kd, int &kh, int &kw) {     // (class, class-local tap) -> index into the weight array
    if constexpr (C::PAR || C::PARF) {
      const int ad = (c_ >> 2) & 1, ah = (c_ >> 1) & 1, aw = c_ & 1;
      kw = aw ? (t & 1) : 0;
      kh = ah ? ((t >> aw) & 1) : 0;
      kd = ad ? ((t >> (aw + ah)) & 1) : 0;
      return ((ad ? 2 * kd : 1) * 3 + (ah ? 2 * kh : 1)) * 3 + (aw ? 2 * kw : 1);
    } else {
      kw = t % C::K; kh = (t / C::K) % C::K; kd = t / (C::K * C::K);
      return t;
    }
  };
  bool dead = false;                                 // (workgroup-uniform: every wave tests the same halo rows)
  if constexpr (LIVE) {
    static_assert(C::HD * C::HH <= 64, "one lane per halo row");
    const int hd = lane / C::HH, hh = lane - hd * C::HH;
    const int id = id0 + hd, ih = ih0 + hh;
    unsigned bits = 0;
    if (lane < C::HD * C::HH && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H) {
      const int w_lo = max(iw0, 0), w_hi = min(iw0 + C::HW, p.W);
      const long long b0 = (((long long)n * p.D + id) * p.H + ih) * p.W + w_lo;
      const unsigned *wd = reinterpret_cast<const unsigned *>(p.x_live) + (b0 >> 5);
      const int sh = (int)(b0 & 31), cnt = w_hi - w_lo;                        // cnt <= 10: at most two words
      const unsigned long long two = (unsigned long long)wd[0] | ((unsigned long long)(sh + cnt > 32 ? wd[1] : 0u) << 32);
      bits = (unsigned)(two >> sh) & ((1u << cnt) - 1u);
    }
    dead = __builtin_amdgcn_ballot_w64(bits != 0) == 0;
  }

  int abase[C::MTW];
#pragma unroll
  for (int mt = 0; mt < C::MTW; ++mt) {
    const int m = (wave * C::MTW + mt) * 32 + r;
    const int tw = m % C::TW, th = (m / C::TW) % C::TH, td = m / (C::TW * C::TH);
    abase[mt] = (((td * C::LS) * C::HH + th * C::LS) * C::HWP + tw * C::LS) * C::ROWB + 16 * h;
  }
  // this lane's B fragment: rows of 8 input channels (k = 8h .. 8h+7 of a 16-wide k-step), column n0 + nt*32 + r
  const int cin8 = p.Cin >> 3;
  const bf16x8 *wlane = reinterpret_cast<const bf16x8 *>(p.w) + (size_t)h * p.CoutPad + n0 + r;

  f32x16_t acc[C::MTW][C::NT];
#pragma unroll
  for (int mt = 0; mt < C::MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

  // ---- halo staging through registers (the next chunk is fetched before the MFMA loop of the current one)
  uint4 stage[C::NIT];
  auto fetch = [&](int c_, int c0) {
    // PARF: the halo of input parity class c_ -- positions 2 (o0 + h) - a per axis
    const int gs = C::PARF ? 2 : C::GS;
    const int fd0 = C::PARF ? 2 * od0 - ((c_ >> 2) & 1) : id0, fh0 = C::PARF ? 2 * oh0 - ((c_ >> 1) & 1) : ih0, fw0 = C::PARF ? 2 * ow0 - (c_ & 1) : iw0;
#pragma unroll
    for (int j = 0; j < C::NIT; ++j) {
      const int it = tid + j * kThreads;
      constexpr int Q = C::CC / 8;
      const int pos = it / Q, q = it - pos * Q;
      const int hw = pos % C::HW, hh = (pos / C::HW) % C::HH, hd = pos / (C::HW * C::HH);
      const int id = fd0 + hd * gs, ih = fh0 + hh * gs, iw = fw0 + hw * gs;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (it < C::PIECES && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
        v = *reinterpret_cast<const uint4 *>(p.x + ((((size_t)n * p.D + id) * p.H + ih) * p.W + iw) * p.Cin + c0 + q * 8);
      stage[j] = v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < C::NIT; ++j) {
      const int it = tid + j * kThreads;
      constexpr int Q = C::CC / 8;
      if (it < C::PIECES) {
        const int pos = it / Q, q = it - pos * Q;
        const int row = pos / C::HW, hw = pos - row * C::HW;
        *reinterpret_cast<uint4 *>(lds + (row * C::HWP + hw) * C::ROWB + q * 16) = stage[j];
      }
    }
  };
  // weight fragments of one tap (all k-steps of the chunk, all N tiles)
  auto load_b = [&](bf16x8 (&dst)[C::KK][C::NT], int c_, int c0, int tap) {
    int kd_, kh_, kw_;
    const int wtap = tap_of(c_, tap, kd_, kh_, kw_);
    const bf16x8 *wt = wlane + ((size_t)wtap * cin8 + (c0 >> 3)) * p.CoutPad;
#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) dst[kk][nt] = wt[(size_t)kk * 2 * p.CoutPad + nt * 32];
  };
  auto mma_tap = [&](const bf16x8 (&bw)[C::KK][C::NT], int c_, int tap) {
    int kd, kh, kw;
    (void)tap_of(c_, tap, kd, kh, kw);
    const int tapoff = ((kd * C::HH + kh) * C::HWP + kw) * C::ROWB;
#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk) {
      bf16x8 a[C::MTW];
#pragma unroll
      for (int mt = 0; mt < C::MTW; ++mt) a[mt] = *reinterpret_cast<const bf16x8 *>(lds + abase[mt] + tapoff + kk * 32);
#pragma unroll
      for (int mt = 0; mt < C::MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], bw[kk][nt], acc[mt][nt], 0, 0, 0);
    }
  };

  // The weight fragments of the next RD - 1 (chunk, tap) steps are in flight while a step is multiplied.  One step ahead was enough
  // on the large maps (many MFMAs per tap) but on the small ones (one 32-row tile per wave: layer3 / layer4, P4 / P5) every tap
  // waited for an L2 round trip -- 27 taps x Cin / 32 chunks x ~0.7 us = the whole 40 us of a launch with 2-6 us of matrix work.
  constexpr int RD = C::MTW == 1 ? 4 : 2;
  bf16x8 bq[RD][C::KK][C::NT];
  const int nchunks = p.Cin / C::CC;
  const int ntaps = C::PAR ? (1 << tap_shift) : C::TAPS;
  const int total = dead ? 0 : (C::PARF ? nchunks * 27 : nchunks * ntaps);      // flattened (class,) (chunk, tap) walk; the B registers rotate across it
  auto split = [&](int cur, int &c_, int &chunk, int &tap) {          // (wave-uniform)
    if constexpr (C::PARF) {
      int base = 0;
      c_ = 0; chunk = 0; tap = 0;
      for (int k = 7; k >= 0; --k) {                                    // the 8-tap class first
        const int sh = __builtin_popcount((unsigned)k), span = nchunks << sh;
        if (cur < base + span) { c_ = k; chunk = (cur - base) >> sh; tap = (cur - base) & ((1 << sh) - 1); break; }
        base += span;
      }
    } else if constexpr (C::PAR) { c_ = cls; chunk = cur >> tap_shift; tap = cur & (ntaps - 1); }
    else { c_ = 0; chunk = cur / C::TAPS; tap = cur - chunk * C::TAPS; }
  };
  auto taps_of = [&](int c_) { return (C::PAR || C::PARF) ? (1 << __builtin_popcount((unsigned)c_)) : C::TAPS; };
  if (!dead) {
    int c_, ch_, t_;
    split(0, c_, ch_, t_);
    fetch(c_, 0);
  }
#pragma unroll
  for (int u = 0; u < RD - 1; ++u)
    if (u < total) {
      int k_, c_, t_;
      split(u, k_, c_, t_);
      load_b(bq[u], k_, c_ * C::CC, t_);
    }
  for (int it = 0; it < total; it += RD) {
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      const int cur = it + u;
      if (cur >= total) break;
      int kcls, chunk, tap;
      split(cur, kcls, chunk, tap);
      if (tap == 0) {
        __syncthreads();                        // every wave is done reading the previous chunk's halo
        commit();
        __syncthreads();
        const int nstage = cur + taps_of(kcls);  // first step of the next (class, chunk) halo
        if (nstage < total) {
          int k2, c2, t2;
          split(nstage, k2, c2, t2);
          fetch(k2, c2 * C::CC);
        }
      }
      const int nxt = cur + RD - 1;
      if (nxt < total) {
        int k3, nchunk, ntap;
        split(nxt, k3, nchunk, ntap);
        load_b(bq[(u + RD - 1) % RD], k3, nchunk * C::CC, ntap);
      }
      mma_tap(bq[u], kcls, tap);
    }
  }

  // ---- epilogue through LDS: the accumulators (column = Cout on the lane, rows = positions in the registers) are
  // transposed to [position][32 Cout] fp32 rows so that every thread finishes 8 consecutive channels of one position
  // with 16-byte bf16 loads / stores (4 adjacent lanes cover a position's 64 contiguous bytes)
  float *eps = reinterpret_cast<float *>(lds);
  constexpr int EROW = 36;                           // floats per position row (32 + 4: float4-aligned, bank-staggered)
  constexpr int ITEMS = C::M * 4;                    // (position, 8-channel group) pairs per N tile
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    __syncthreads();                                 // halo reads (first pass) / previous pass's row reads are done
#pragma unroll
    for (int mt = 0; mt < C::MTW; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        eps[((wave * C::MTW + mt) * 32 + row) * EROW + r] = acc[mt][nt][e];
      }
    __syncthreads();
    for (int item = tid; item < ITEMS; item += kThreads) {
      const int m = item >> 2, grp = item & 3;
      const int co = n0 + nt * 32 + grp * 8;
      const int tw = m % C::TW, th = (m / C::TW) % C::TH, td = m / (C::TW * C::TH);
      const int od = od0 + td, oh = oh0 + th, ow = ow0 + tw;
      if (co >= p.Cout || od >= p.Do || oh >= p.Ho || ow >= p.Wo) continue;
      if (C::PAR && (2 * od + ad >= p.yD || 2 * oh + ah >= p.yH || 2 * ow + aw >= p.yW)) continue;
      const float4 v0 = *reinterpret_cast<const float4 *>(eps + m * EROW + grp * 8);
      const float4 v1 = *reinterpret_cast<const float4 *>(eps + m * EROW + grp * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const size_t idx = C::PAR ? ((((size_t)n * p.yD + 2 * od + ad) * p.yH + 2 * oh + ah) * p.yW + 2 * ow + aw) * p.Cout + co
                                : ((((size_t)n * p.yD + od * p.ysd) * p.yH + oh * p.ysh) * p.yW + ow * p.ysw) * p.Cout + co;
      if (p.shift) {
        const float4 s0 = *reinterpret_cast<const float4 *>(p.shift + co), s1 = *reinterpret_cast<const float4 *>(p.shift + co + 4);
        v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
      }
      auto add8 = [&](const bf16_t *src, size_t at) {
        const uint4 q = *reinterpret_cast<const uint4 *>(src + at);
        const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[2 * j] += __uint_as_float(wds[j] << 16);
          v[2 * j + 1] += __uint_as_float(wds[j] & 0xffff0000u);
        }
      };
      if (p.add1) add8(p.add1, idx);
      if (p.add2) add8(p.add2, idx);
      if (p.up) {
        const int ud = nearest_src(od, p.Du, p.Do), uh = nearest_src(oh, p.Hu, p.Ho), uw = nearest_src(ow, p.Wu, p.Wo);
        add8(p.up, ((((size_t)n * p.Du + ud) * p.Hu + uh) * p.Wu + uw) * p.Cout + co);
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (p.mask) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p.mask + idx);
        const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!(__uint_as_float(wds[j] << 16) > 0.f)) v[2 * j] = 0.f;
          if (!(__uint_as_float(wds[j] & 0xffff0000u) > 0.f)) v[2 * j + 1] = 0.f;
        }
      }
      uint4 o;
      o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
      o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
      o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
      *reinterpret_cast<uint4 *>(p.y + idx) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ 3x3x3 / stride 1, 16 -> 16 channels
// conv2 of the layer1 bottlenecks and its data gradient (resnet3d.py:126-139 at 16 planes): six launches per step on the largest
// map.  In the generic kernel above they are its worst case -- K = 16 per tap is ONE MFMA per tap and 32-row tile, half of the
// 32-wide N tile is padding, and the epilogue transposes 32 columns through LDS for 16 -- 47-71 us per launch with the matrix pipe
// busy 13 % (profiles/r04/bench_bf16_sq_counters.md).  Here, on v_mfma_f32_16x16x32_bf16:
//   * A = weights [16 cout][32 k], k = 2 taps x 16 channels: the 14 A fragments of a lane are read once from the generic packed
//     layout ([tap][Cin/8][CoutPad][8]: a lane's 8 channels of one tap and cout are 16 contiguous bytes) and stay in registers;
//   * B = the halo [position][16 ch] in LDS (32 B per position): a lane's fragment -- 8 channels of one position under one tap --
//     is ONE ds_read_b128, conflict-free as it lies (16 consecutive positions x two 16-byte halves fill the 64 banks);
//   * the accumulator holds 4 consecutive output channels of one position per lane: shift / ReLU / ReLU-mask and an 8-byte store
//     straight from registers, 512 contiguous bytes per store instruction, no LDS pass.
// A wave owns one depth of the 4 x 8 x 16 tile: 8 rows of 16 positions x 14 k-steps = 112 MFMAs on 112 LDS reads.
struct C16 {
  static constexpr int TD = 4, TH = 8, TW = 16, HD = TD + 2, HH = TH + 2, HW = TW + 2, HP = HD * HH * HW;
  static constexpr int ROWB = 32, LDS_BYTES = HP * ROWB, NT = 256, STEPS = 14;
  static constexpr int PIECES = HP * 2, NIT = (PIECES + NT - 1) / NT;
};
typedef __attribute__((ext_vector_type(4))) float f32x4_c16;

__global__ __launch_bounds__(C16::NT, 3) void conv3d_bf16_c16_kernel(BfArgs p, int tiles_d, int tiles_h, int tiles_w) {
  using C = C16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  int b = blockIdx.x;
  const int tw_i = b % tiles_w; b /= tiles_w;
  const int th_i = b % tiles_h; b /= tiles_h;
  const int td_i = b % tiles_d; b /= tiles_d;
  const int n = b;
  const int od0 = td_i * C::TD, oh0 = th_i * C::TH, ow0 = tw_i * C::TW;

  // halo -> LDS through registers; a piece = 16 bytes (8 channels) of one position.  Buffer loads: a position outside the map
  // gets an offset past the end of the buffer and reads as zeros (the launcher bounds x below 2^31 bytes)
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, 0x7fffffff, 0x00020000);
  uint4 stage[C::NIT];
#pragma unroll
  for (int j = 0; j < C::NIT; ++j) {
    const int it = tid + j * C::NT;
    const int pos = it >> 1, half = it & 1;
    const int hw = pos % C::HW, hh = (pos / C::HW) % C::HH, hd = pos / (C::HW * C::HH);
    const int id = od0 - 1 + hd, ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
    const bool ok = it < C::PIECES && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    const unsigned off = ok ? (unsigned)((((n * p.D + id) * p.H + ih) * p.W + iw) * 32 + half * 16) : 0x80000000u;
    stage[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
  }
  // this lane's weight fragments: k-step s covers taps 2s and 2s + 1; lanes q = 0, 1 hold tap 2s (channels 0-7, 8-15), q = 2, 3
  // tap 2s + 1 (the 28th tap does not exist: zeros)
  bf16x8 wr[C::STEPS];
  int toff[C::STEPS];                                  // LDS byte offset of the lane's tap in a k-step
#pragma unroll
  for (int s = 0; s < C::STEPS; ++s) {
    const int tap = 2 * s + (q >> 1);
    const int tc = tap < 27 ? tap : 26;
    const bf16x8 w = reinterpret_cast<const bf16x8 *>(p.w)[(size_t)(tc * 2 + (q & 1)) * p.CoutPad + r];
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
    wr[s] = tap < 27 ? w : z;
    const int kw = tc % 3, kh = (tc / 3) % 3, kd = tc / 9;
    toff[s] = ((kd * C::HH + kh) * C::HW + kw) * C::ROWB;
  }
#pragma unroll
  for (int j = 0; j < C::NIT; ++j) {
    const int it = tid + j * C::NT;
    if (it < C::PIECES) *reinterpret_cast<uint4 *>(lds + it * 16) = stage[j];
  }
  __syncthreads();

  f32x4_c16 acc[C::TH];
#pragma unroll
  for (int i = 0; i < C::TH; ++i) acc[i] = f32x4_c16{0.f, 0.f, 0.f, 0.f};
  // lane (r, q): position column r of every row of depth `wave`, channel half q & 1
  const unsigned char *bbase = lds + ((wave * C::HH) * C::HW + r) * C::ROWB + (q & 1) * 16;
#pragma unroll
  for (int s = 0; s < C::STEPS; ++s) {
    const unsigned char *bs = bbase + toff[s];
#pragma unroll
    for (int i = 0; i < C::TH; ++i) {
      const bf16x8 f = *reinterpret_cast<const bf16x8 *>(bs + i * (C::HW * C::ROWB));
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s], f, acc[i], 0, 0, 0);
    }
  }
  // ---- epilogue: C/D layout of 16x16: column (position) = lane & 15, row (cout) = 4 * (lane >> 4) + reg
  const int od = od0 + wave, ow = ow0 + r;
  if (od >= p.Do || ow >= p.Wo) return;
  float sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.shift) {
    const float4 s4 = *reinterpret_cast<const float4 *>(p.shift + 4 * q);
    sh[0] = s4.x; sh[1] = s4.y; sh[2] = s4.z; sh[3] = s4.w;
  }
#pragma unroll
  for (int i = 0; i < C::TH; ++i) {
    const int oh = oh0 + i;
    if (oh >= p.Ho) continue;
    const size_t idx = ((((size_t)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * 16 + 4 * q;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = acc[i][e] + sh[e];
      if (p.relu) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.mask) {
      const uint2 m = *reinterpret_cast<const uint2 *>(p.mask + idx);
      if (!(__uint_as_float(m.x << 16) > 0.f)) v[0] = 0.f;
      if (!(__uint_as_float(m.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
      if (!(__uint_as_float(m.y << 16) > 0.f)) v[2] = 0.f;
      if (!(__uint_as_float(m.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
    }
    *reinterpret_cast<uint2 *>(p.y + idx) = make_uint2((uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16),
                                                      (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16));
  }
}

// ------------------------------------------------------------------------------------------------ 3x3x3 / stride 1, 32 -> 32 channels
// conv2 of the layer2 bottlenecks and its data gradient (eight launches per step): the same construction as the 16-channel kernel
// with one tap per k-step (k = 32 channels).  The two 16-wide output-channel tiles belong to different waves, so a wave's 27 weight
// fragments still fit its registers (108); a wave owns one depth of the 2 x 8 x 16 tile and one channel tile: 216 MFMAs on 216
// ds_read_b128.  Halo rows of 80 bytes (64 + 16: three of a read's sixteen 16-byte slots collide, against every fourth with 64).
struct C32 {
  static constexpr int TD = 2, TH = 8, TW = 16, HD = TD + 2, HH = TH + 2, HW = TW + 2, HP = HD * HH * HW;
  static constexpr int ROWB = 80, LDS_BYTES = HP * ROWB, NT = 256, STEPS = 27;
  static constexpr int PIECES = HP * 4, NIT = (PIECES + NT - 1) / NT;
};

__global__ __launch_bounds__(C32::NT, 2) void conv3d_bf16_c32_kernel(BfArgs p, int tiles_d, int tiles_h, int tiles_w) {
  using C = C32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int ct = wave & 1, td = wave >> 1;             // output-channel tile, depth inside the tile
  int b = blockIdx.x;
  const int tw_i = b % tiles_w; b /= tiles_w;
  const int th_i = b % tiles_h; b /= tiles_h;
  const int td_i = b % tiles_d; b /= tiles_d;
  const int n = b;
  const int od0 = td_i * C::TD, oh0 = th_i * C::TH, ow0 = tw_i * C::TW;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, 0x7fffffff, 0x00020000);
  uint4 stage[C::NIT];
#pragma unroll
  for (int j = 0; j < C::NIT; ++j) {
    const int it = tid + j * C::NT;
    const int pos = it >> 2, pc = it & 3;
    const int hw = pos % C::HW, hh = (pos / C::HW) % C::HH, hd = pos / (C::HW * C::HH);
    const int id = od0 - 1 + hd, ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
    const bool ok = it < C::PIECES && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    const unsigned off = ok ? (unsigned)((((n * p.D + id) * p.H + ih) * p.W + iw) * 64 + pc * 16) : 0x80000000u;
    stage[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
  }
  // this lane's weight fragments: tap s, channels 8q .. 8q+7, output channel 16 ct + r
  bf16x8 wr[C::STEPS];
#pragma unroll
  for (int s = 0; s < C::STEPS; ++s) wr[s] = reinterpret_cast<const bf16x8 *>(p.w)[(size_t)(s * 4 + q) * p.CoutPad + 16 * ct + r];
#pragma unroll
  for (int j = 0; j < C::NIT; ++j) {
    const int it = tid + j * C::NT;
    if (it < C::PIECES) *reinterpret_cast<uint4 *>(lds + (it >> 2) * C::ROWB + (it & 3) * 16) = stage[j];
  }
  __syncthreads();

  f32x4_c16 acc[C::TH];
#pragma unroll
  for (int i = 0; i < C::TH; ++i) acc[i] = f32x4_c16{0.f, 0.f, 0.f, 0.f};
  const unsigned char *bbase = lds + ((td * C::HH) * C::HW + r) * C::ROWB + q * 16;
#pragma unroll
  for (int s = 0; s < C::STEPS; ++s) {
    const int kw = s % 3, kh = (s / 3) % 3, kd = s / 9;
    const unsigned char *bs = bbase + ((kd * C::HH + kh) * C::HW + kw) * C::ROWB;
#pragma unroll
    for (int i = 0; i < C::TH; ++i) {
      const bf16x8 f = *reinterpret_cast<const bf16x8 *>(bs + i * (C::HW * C::ROWB));
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s], f, acc[i], 0, 0, 0);
    }
  }
  // ---- epilogue: column (position) = lane & 15, row (cout inside the wave's tile) = 4 * (lane >> 4) + reg
  const int od = od0 + td, ow = ow0 + r;
  if (od >= p.Do || ow >= p.Wo) return;
  const int co = 16 * ct + 4 * q;
  float sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.shift) {
    const float4 s4 = *reinterpret_cast<const float4 *>(p.shift + co);
    sh[0] = s4.x; sh[1] = s4.y; sh[2] = s4.z; sh[3] = s4.w;
  }
#pragma unroll
  for (int i = 0; i < C::TH; ++i) {
    const int oh = oh0 + i;
    if (oh >= p.Ho) continue;
    const size_t idx = ((((size_t)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * 32 + co;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = acc[i][e] + sh[e];
      if (p.relu) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.mask) {
      const uint2 m = *reinterpret_cast<const uint2 *>(p.mask + idx);
      if (!(__uint_as_float(m.x << 16) > 0.f)) v[0] = 0.f;
      if (!(__uint_as_float(m.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
      if (!(__uint_as_float(m.y << 16) > 0.f)) v[2] = 0.f;
      if (!(__uint_as_float(m.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
    }
    *reinterpret_cast<uint2 *>(p.y + idx) = make_uint2((uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16),
                                                      (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16));
  }
}

bool c32_ok(const BfArgs &a) {
  return a.Cin == 32 && a.Cout == 32 && a.pd == 1 && a.ph == 1 && a.pw == 1 && !a.add1 && !a.add2 && !a.up && !a.x_live &&
         a.ysd == 1 && a.ysh == 1 && a.ysw == 1 && a.yD == a.Do && a.yH == a.Ho && a.yW == a.Wo &&
         (long long)a.N * a.D * a.H * a.W * 64 < 0x7fffffffLL;
}

int launch_c32(const BfArgs &a, hipStream_t stream) {
  using C = C32;
  const int tiles_d = (a.Do + C::TD - 1) / C::TD, tiles_h = (a.Ho + C::TH - 1) / C::TH, tiles_w = (a.Wo + C::TW - 1) / C::TW;
  const long long tiles = (long long)a.N * tiles_d * tiles_h * tiles_w;
  if (tiles > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  conv3d_bf16_c32_kernel<<<(unsigned)tiles, C::NT, C::LDS_BYTES, stream>>>(a, tiles_d, tiles_h, tiles_w);
  return mr3d_launch_status();
}

// ------------------------------------------------------------------------------------------------ 3x3x3 / stride 1, 64 -> 64 channels
// conv2 of layer3's bottlenecks, the P2..P5 output convolutions of the FPN, and their data gradients.  The same construction once
// more with 8 waves: wave (ct, kh) owns the 16-wide output-channel tile ct and the input-channel half kh -- its 27 weight
// fragments (one per tap: 32 channels = one k-step) fit its registers, its 8 accumulators are the 2 x 4 rows of 16 positions of the
// 2 x 4 x 16 tile; the two halves of a channel tile meet in LDS before the epilogue.
// 512 volume, per launch: P2 map 181-202 -> 118-124 us, P3 41 -> 22, layer3 / P4 / P5 18 -> 12; bf16 backbone step 4.42 -> 4.26 ms.
// NCT = 16-wide output-channel tiles of a launch: 4 (Cout = 64), or 2 for the last 32 of a 96-wide layer (the DeformConv3d neck's
// offset convolutions, 81 channels padded to 96: one launch of each kind, 469 -> ~190 us on the P2 map)
template <int NCT_>
struct C64 {
  static constexpr int NCT = NCT_;
  static constexpr int TD = 2, TH = 4, TW = 16, HD = TD + 2, HH = TH + 2, HW = TW + 2, HP = HD * HH * HW;
  static constexpr int ROWB = 128, HALO_BYTES = HP * ROWB, NT = NCT_ * 2 * 64, STEPS = 27;
  static constexpr int RED_BYTES = NCT_ * 64 * 32 * 4;                       // NCT waves x 64 lanes x 8 accumulators x 4 floats
  static constexpr int LDS_BYTES = 2 * HALO_BYTES + RED_BYTES;
  static constexpr int NLOADS = HALO_BYTES / 1024;                           // 1 KB (8 positions) per wave-wide direct load
  static_assert(HALO_BYTES % 1024 == 0, "whole wave loads");
};

// Persistent workgroups (one per CU: the 27 weight fragments of a lane take 108 registers).  The halo of the NEXT tile travels
// global -> LDS directly (buffer_load_dwordx4 ... lds, no staging registers) into the second of two buffers while the current
// tile is multiplied.  Rows are 128 bytes, unpadded (a direct load writes 1 KB contiguously); the 16-byte piece c of halo position
// p sits in slot c ^ (p & 7) of its row instead -- the swizzle is applied on the GLOBAL side of the load (a lane fetches the piece
// that belongs in its slot), and it makes the sixteen slots of a ds_read_b128 lane group distinct.
template <int NCT>
__global__ __launch_bounds__(C64<NCT>::NT, 1) void conv3d_bf16_c64_kernel(BfArgs p, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  using C = C64<NCT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int ct = wave % NCT, kh = wave / NCT;           // output-channel tile, input-channel half

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, 0x7fffffff, 0x00020000);
  auto origin = [&](int tile, int &n, int &od0, int &oh0, int &ow0) {
    int b = tile;
    ow0 = (b % tiles_w) * C::TW; b /= tiles_w;
    oh0 = (b % tiles_h) * C::TH; b /= tiles_h;
    od0 = (b % tiles_d) * C::TD;
    n = b / tiles_d;
  };
  auto fetch = [&](int tile, int buf) {
    int n, od0, oh0, ow0;
    origin(tile, n, od0, oh0, ow0);
#pragma unroll
    for (int k = 0; k < (C::NLOADS + 2 * NCT - 1) / (2 * NCT); ++k) {
      const int ld = wave + 2 * NCT * k;               // wave-uniform
      if (ld < C::NLOADS) {
        const int pos = ld * 8 + (lane >> 3), slot = lane & 7;
        const int hw = pos % C::HW, hh = (pos / C::HW) % C::HH, hd = pos / (C::HW * C::HH);
        const int id = od0 - 1 + hd, ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
        const bool ok = (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const unsigned off = ok ? (unsigned)((((n * p.D + id) * p.H + ih) * p.W + iw) * 128 + ((slot ^ (pos & 7)) << 4)) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void *)(lds + buf * C::HALO_BYTES + ld * 1024), 16,
                                                 (int)off, 0, 0, 0);
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile, 0);
  // this lane's weight fragments: tap s, channels 32 kh + 8q .. + 7, output channel 16 ct + r
  bf16x8 wr[C::STEPS];
#pragma unroll
  for (int s = 0; s < C::STEPS; ++s) wr[s] = reinterpret_cast<const bf16x8 *>(p.w)[(size_t)(s * 8 + 4 * kh + q) * p.CoutPad + p.co_base + 16 * ct + r];
  const int co = p.co_base + 16 * ct + 4 * q;
  float sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.shift) {
    const float4 s4 = *reinterpret_cast<const float4 *>(p.shift + co);
    sh[0] = s4.x; sh[1] = s4.y; sh[2] = s4.z; sh[3] = s4.w;
  }
  // fragment address of halo position r + c (c: a constant of the tap and the accumulator): row (r + c) * 128, slot pc ^ ((r + c) & 7)
  // = base[c & 7] + c * 128 with eight per-lane bases
  const int pc = 4 * kh + q;
  unsigned base[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) base[m] = (unsigned)(r * C::ROWB + ((pc ^ ((r + m) & 7)) << 4));
  f32x4_c16 *red = reinterpret_cast<f32x4_c16 *>(lds + 2 * C::HALO_BYTES) + ct * 64 + lane;      // [accumulator][channel tile][lane]
  __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0): this wave's part of the first halo is in LDS
  __syncthreads();

  int buf = 0;
  for (; tile < ntiles; tile += (int)gridDim.x, buf ^= 1) {
    const int next = tile + (int)gridDim.x;
    if (next < ntiles) fetch(next, buf ^ 1);           // in flight during the MFMAs
    const unsigned char *hb = lds + buf * C::HALO_BYTES;
    f32x4_c16 acc[C::TD * C::TH];
#pragma unroll
    for (int i = 0; i < C::TD * C::TH; ++i) acc[i] = f32x4_c16{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s) {
      const int kw = s % 3, khh = (s / 3) % 3, kd = s / 9;
#pragma unroll
      for (int i = 0; i < C::TD * C::TH; ++i) {
        const int td = i / C::TH, th = i % C::TH;
        const int c = ((td + kd) * C::HH + th + khh) * C::HW + kw;
        const bf16x8 f = *reinterpret_cast<const bf16x8 *>(hb + base[c & 7] + c * C::ROWB);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s], f, acc[i], 0, 0, 0);
      }
    }
    __syncthreads();                                   // this halo has been read (and the previous tile's sums)
    // the input-channel halves meet: kh = 1 hands its sums over through LDS
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < C::TD * C::TH; ++i) red[i * (NCT * 64)] = acc[i];
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                // vmcnt(0): this wave's part of the next halo has landed
    __syncthreads();
    if (kh == 1) continue;
    // ---- epilogue: column (position) = lane & 15, row (cout inside the wave's tile) = 4 * (lane >> 4) + reg
    int n, od0, oh0, ow0;
    origin(tile, n, od0, oh0, ow0);
    const int ow = ow0 + r;
    if (ow >= p.Wo) continue;
#pragma unroll
    for (int i = 0; i < C::TD * C::TH; ++i) {
      const int od = od0 + i / C::TH, oh = oh0 + i % C::TH;
      if (od >= p.Do || oh >= p.Ho) continue;
      const size_t idx = ((((size_t)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * p.Cout + co;
      const f32x4_c16 o2 = red[i * (NCT * 64)];
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (acc[i][e] + o2[e]) + sh[e];
        if (p.relu) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.mask) {
        const uint2 m = *reinterpret_cast<const uint2 *>(p.mask + idx);
        if (!(__uint_as_float(m.x << 16) > 0.f)) v[0] = 0.f;
        if (!(__uint_as_float(m.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
        if (!(__uint_as_float(m.y << 16) > 0.f)) v[2] = 0.f;
        if (!(__uint_as_float(m.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
      }
      *reinterpret_cast<uint2 *>(p.y + idx) = make_uint2((uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16),
                                                        (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16));
    }
  }
}

bool c64_ok(const BfArgs &a) {
  return a.Cin == 64 && (a.Cout == 64 || a.Cout == 96) && a.pd == 1 && a.ph == 1 && a.pw == 1 && !a.add1 && !a.add2 && !a.up && !a.x_live &&
         a.ysd == 1 && a.ysh == 1 && a.ysw == 1 && a.yD == a.Do && a.yH == a.Ho && a.yW == a.Wo &&
         (long long)a.N * a.D * a.H * a.W * 128 < 0x7fffffffLL;
}

template <int NCT>
int launch_c64_part(BfArgs a, int co_base, hipStream_t stream) {
  using C = C64<NCT>;
  const int tiles_d = (a.Do + C::TD - 1) / C::TD, tiles_h = (a.Ho + C::TH - 1) / C::TH, tiles_w = (a.Wo + C::TW - 1) / C::TW;
  const long long tiles = (long long)a.N * tiles_d * tiles_h * tiles_w;
  if (tiles > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  static bool attr_done = false;
  static int cus = 0;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_bf16_c64_kernel<NCT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    attr_done = true;
  }
  a.co_base = co_base;
  const int grid = tiles < cus ? (int)tiles : cus;
  conv3d_bf16_c64_kernel<NCT><<<(unsigned)grid, C::NT, C::LDS_BYTES, stream>>>(a, tiles_d, tiles_h, tiles_w, (int)tiles);
  return mr3d_launch_status();
}

int launch_c64(const BfArgs &a, hipStream_t stream) {
  const int st = launch_c64_part<4>(a, 0, stream);
  if (st != MR3D_OK || a.Cout == 64) return st;
  return launch_c64_part<2>(a, 64, stream);            // channels 64 .. 95 of a 96-wide layer
}

// the shapes the 16-channel kernel takes: dense output, no residual / upsample inputs, the whole input below 2^31 bytes
bool c16_ok(const BfArgs &a) {
  return a.Cin == 16 && a.Cout == 16 && a.pd == 1 && a.ph == 1 && a.pw == 1 && !a.add1 && !a.add2 && !a.up && !a.x_live &&
         a.ysd == 1 && a.ysh == 1 && a.ysw == 1 && a.yD == a.Do && a.yH == a.Ho && a.yW == a.Wo &&
         (long long)a.N * a.D * a.H * a.W * 32 < 0x7fffffffLL;
}

int launch_c16(const BfArgs &a, hipStream_t stream) {
  using C = C16;
  const int tiles_d = (a.Do + C::TD - 1) / C::TD, tiles_h = (a.Ho + C::TH - 1) / C::TH, tiles_w = (a.Wo + C::TW - 1) / C::TW;
  const long long tiles = (long long)a.N * tiles_d * tiles_h * tiles_w;
  if (tiles > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  conv3d_bf16_c16_kernel<<<(unsigned)tiles, C::NT, C::LDS_BYTES, stream>>>(a, tiles_d, tiles_h, tiles_w);
  return mr3d_launch_status();
}

template <class C, bool LIVE = false>
int launch_bf(const BfArgs &a, hipStream_t stream) {
  const int tiles_d = (a.Do + C::TD - 1) / C::TD, tiles_h = (a.Ho + C::TH - 1) / C::TH, tiles_w = (a.Wo + C::TW - 1) / C::TW;
  static bool attr_done = false;
  if (!attr_done && C::LDS_BYTES > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_bf16_kernel<C, LIVE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  dim3 grid((unsigned)(a.N * tiles_d * tiles_h * tiles_w), (unsigned)(a.CoutPad / C::NB), C::PAR ? 8u : 1u);
  conv3d_bf16_kernel<C, LIVE><<<grid, kThreads, C::LDS_BYTES, stream>>>(a, tiles_d, tiles_h, tiles_w);
  return mr3d_launch_status();
}

template <int K, int S, int CC, int NT>
int dispatch_tile(const BfArgs &a, hipStream_t stream) {
  const long long wg488 = (long long)a.N * ((a.Do + 3) / 4) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8) * (a.CoutPad / (32 * NT));
  if constexpr (K == 3 && S == 2) {
    return launch_bf<BfCfg<K, S, 2, 8, 8, CC, NT>>(a, stream);       // the 4x8x8 halo of a stride-2 3x3x3 does not fit LDS
  } else {
    if constexpr (K == 3 && S == 1) {
      if (a.x_live && a.pd == 1 && a.ph == 1 && a.pw == 1) {
        if (wg488 >= 512) return launch_bf<BfCfg<K, S, 4, 8, 8, CC, NT>, true>(a, stream);
        return launch_bf<BfCfg<K, S, 2, 8, 8, CC, NT>, true>(a, stream);
      }
    }
    if (wg488 >= 512) return launch_bf<BfCfg<K, S, 4, 8, 8, CC, NT>>(a, stream);
    return launch_bf<BfCfg<K, S, 2, 8, 8, CC, NT>>(a, stream);
  }
}

template <int K, int S>
int dispatch_cc(const BfArgs &a, hipStream_t stream) {
  const bool wide = a.CoutPad % 64 == 0;
  if constexpr (K == 3 && S == 2) {
    return wide ? dispatch_tile<K, S, 16, 2>(a, stream) : dispatch_tile<K, S, 16, 1>(a, stream);
  } else {
    if constexpr (K == 1) {
      // 1x1x1 on the small maps: a launch is Cin / CC staging round trips per workgroup (one step ahead): 64-channel chunks halve them (bf16 backbone 175.2 -> 176.5 iters/s)
      if (a.Cin % 64 == 0) return wide ? dispatch_tile<K, S, 64, 2>(a, stream) : dispatch_tile<K, S, 64, 1>(a, stream);
    }
    if (a.Cin % 32 == 0) return wide ? dispatch_tile<K, S, 32, 2>(a, stream) : dispatch_tile<K, S, 32, 1>(a, stream);
    return wide ? dispatch_tile<K, S, 16, 2>(a, stream) : dispatch_tile<K, S, 16, 1>(a, stream);
  }
}

}  // namespace

// Data gradient of a kernel-3 / stride-2 / pad-1 convolution from the coarse dy, bf16 (the fp32 form: mr3d_conv3d_dgrad_s2k3 in
// csrc/conv3d_fwd.hip): the 8 output parity classes of dx are computed with the 1..8 taps each of them meets -- the zero-stuffed
// form multiplied 8x the positions, 7/8 of them zeros.
extern "C" int mr3d_conv3d_dgrad_s2k3_bf16(const mr3d_conv3d_desc *d, mr3d_stream_t stream_) {
  if (!d || !d->x || !d->w || !d->y) return MR3D_ERR_INVALID_ARG;
  if (d->y_d <= 0 || d->y_h <= 0 || d->y_w <= 0 || d->shift || d->up || d->relu) return MR3D_ERR_INVALID_ARG;
  if (d->d != (d->y_d - 1) / 2 + 1 || d->h != (d->y_h - 1) / 2 + 1 || d->w_in != (d->y_w - 1) / 2 + 1) return MR3D_ERR_INVALID_ARG;
  if (d->n <= 0 || d->cin % 16 != 0 || d->cout % 8 != 0) return MR3D_ERR_UNSUPPORTED;
  BfArgs a;
  a.x = (const bf16_t *)d->x; a.w = (const bf16_t *)d->w; a.y = (bf16_t *)d->y; a.shift = nullptr;
  a.add1 = (const bf16_t *)d->add1; a.add2 = (const bf16_t *)d->add2; a.up = nullptr; a.mask = (const bf16_t *)d->mask;
  a.N = d->n; a.D = d->d; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Cout = d->cout;
  a.CoutPad = (d->cout + 31) / 32 * 32;
  a.pd = a.ph = a.pw = 0;
  a.Do = d->d; a.Ho = d->h; a.Wo = d->w_in;          // class grid = dy's grid
  a.Du = a.Hu = a.Wu = 0; a.relu = 0;
  a.x_live = nullptr;
  a.ysd = a.ysh = a.ysw = 2; a.yD = d->y_d; a.yH = d->y_h; a.yW = d->y_w;
  hipStream_t stream = mr3d_stream(stream_);
  const bool wide = a.CoutPad % 64 == 0;
  const long long wg488 = (long long)a.N * ((a.Do + 3) / 4) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8) * (a.CoutPad / (wide ? 64 : 32)) * 8;
  const bool big = wg488 >= 512;
  if (a.Cin % 32 == 0) {
    if (wide) return big ? launch_bf<BfCfg<2, 1, 4, 8, 8, 32, 2, true>>(a, stream) : launch_bf<BfCfg<2, 1, 2, 8, 8, 32, 2, true>>(a, stream);
    return big ? launch_bf<BfCfg<2, 1, 4, 8, 8, 32, 1, true>>(a, stream) : launch_bf<BfCfg<2, 1, 2, 8, 8, 32, 1, true>>(a, stream);
  }
  if (wide) return big ? launch_bf<BfCfg<2, 1, 4, 8, 8, 16, 2, true>>(a, stream) : launch_bf<BfCfg<2, 1, 2, 8, 8, 16, 2, true>>(a, stream);
  return big ? launch_bf<BfCfg<2, 1, 4, 8, 8, 16, 1, true>>(a, stream) : launch_bf<BfCfg<2, 1, 2, 8, 8, 16, 1, true>>(a, stream);
}

extern "C" int mr3d_conv3d_forward_bf16(const mr3d_conv3d_desc *d, mr3d_stream_t stream_) {
  if (!d || !d->x || !d->w || !d->y) return MR3D_ERR_INVALID_ARG;
  if (d->kd != d->kh || d->kh != d->kw || d->sd != d->sh || d->sh != d->sw) return MR3D_ERR_UNSUPPORTED;
  BfArgs a;
  a.x = (const bf16_t *)d->x; a.w = (const bf16_t *)d->w; a.y = (bf16_t *)d->y; a.shift = d->shift;
  a.add1 = (const bf16_t *)d->add1; a.add2 = (const bf16_t *)d->add2; a.up = (const bf16_t *)d->up; a.mask = (const bf16_t *)d->mask;
  a.N = d->n; a.D = d->d; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Cout = d->cout;
  a.CoutPad = (d->cout + 31) / 32 * 32;
  a.pd = d->pd; a.ph = d->ph; a.pw = d->pw;
  a.Do = (d->d + 2 * d->pd - d->kd) / d->sd + 1;
  a.Ho = (d->h + 2 * d->ph - d->kh) / d->sh + 1;
  a.Wo = (d->w_in + 2 * d->pw - d->kw) / d->sw + 1;
  a.Du = d->up_d; a.Hu = d->up_h; a.Wu = d->up_w;
  a.relu = d->relu;
  a.x_live = d->x_live;
  a.ysd = d->ys_d > 0 ? d->ys_d : 1; a.ysh = d->ys_h > 0 ? d->ys_h : 1; a.ysw = d->ys_w > 0 ? d->ys_w : 1;
  a.yD = d->y_d > 0 ? d->y_d : a.Do; a.yH = d->y_h > 0 ? d->y_h : a.Ho; a.yW = d->y_w > 0 ? d->y_w : a.Wo;
  if ((a.Do - 1) * a.ysd >= a.yD || (a.Ho - 1) * a.ysh >= a.yH || (a.Wo - 1) * a.ysw >= a.yW) return MR3D_ERR_INVALID_ARG;
  if (a.N <= 0 || a.Do <= 0 || a.Ho <= 0 || a.Wo <= 0) return MR3D_ERR_INVALID_ARG;
  if (a.up && (a.Du <= 0 || a.Hu <= 0 || a.Wu <= 0)) return MR3D_ERR_INVALID_ARG;
  if (a.Cin % 16 != 0 || a.Cout % 8 != 0) return MR3D_ERR_UNSUPPORTED;
  if (d->relu & ~MR3D_CONV_RELU) return MR3D_ERR_UNSUPPORTED;          // MR3D_CONV_UP_EVEN: fp32 kernels only (mr3d.h)
  hipStream_t stream = mr3d_stream(stream_);
  const int k = d->kd, s = d->sd;
  if (k == 1 && s == 1) {
    const bool dense = a.pd == 0 && a.ph == 0 && a.pw == 0 && a.ysd == 1 && a.ysh == 1 && a.ysw == 1 && a.yD == a.Do && a.yH == a.Ho &&
                       a.yW == a.Wo;
    if (dense) {      // HBM-bound layers on the large maps: the barrier-free streaming kernel (csrc/pointwise_bf16.hip)
      const int st = mr3d_pointwise_bf16_try(a.x, a.w, a.y, a.shift, a.add1, a.add2, a.up, a.mask, (long long)a.N * a.Do * a.Ho * a.Wo,
                                             a.Cin, a.Cout, a.CoutPad, a.Do, a.Ho, a.Wo, a.Du, a.Hu, a.Wu, a.relu, stream);
      if (st != MR3D_ERR_UNSUPPORTED) return st;
    }
    return dispatch_cc<1, 1>(a, stream);
  }
  if (k == 1 && s == 2) return dispatch_cc<1, 2>(a, stream);
  if (k == 3 && s == 1 && c16_ok(a)) return launch_c16(a, stream);
  if (k == 3 && s == 1 && c32_ok(a)) return launch_c32(a, stream);
  if (k == 3 && s == 1 && c64_ok(a)) return launch_c64(a, stream);
  if (k == 3 && s == 1) return dispatch_cc<3, 1>(a, stream);
  if (k == 3 && s == 2 && a.pd == 1 && a.ph == 1 && a.pw == 1 && (long long)a.N * a.Do * a.Ho * a.Wo >= 8192) {
    // from the 8 parity classes of the input (BfCfg PARF): 32-channel chunks and 4 x 8 x 8 tiles fit now.  512 volume: layer2.0 conv2
    // 67 -> 31 us, layer3.0 40 -> 34; on layer4.0's 1280 outputs the 32 (class, chunk) stagings of a workgroup are a longer chain than
    // the direct form's 8 (62 against 37 us): small maps keep it
    const bool wide = a.CoutPad % 64 == 0;
    const long long wg488 = (long long)a.N * ((a.Do + 3) / 4) * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8) * (a.CoutPad / (wide ? 64 : 32));
    const bool big = wg488 >= 512;
    if (a.Cin % 32 == 0) {
      if (wide) return big ? launch_bf<BfCfg<2, 1, 4, 8, 8, 32, 2, false, true>>(a, stream) : launch_bf<BfCfg<2, 1, 2, 8, 8, 32, 2, false, true>>(a, stream);
      return big ? launch_bf<BfCfg<2, 1, 4, 8, 8, 32, 1, false, true>>(a, stream) : launch_bf<BfCfg<2, 1, 2, 8, 8, 32, 1, false, true>>(a, stream);
    }
    if (wide) return big ? launch_bf<BfCfg<2, 1, 4, 8, 8, 16, 2, false, true>>(a, stream) : launch_bf<BfCfg<2, 1, 2, 8, 8, 16, 2, false, true>>(a, stream);
    return big ? launch_bf<BfCfg<2, 1, 4, 8, 8, 16, 1, false, true>>(a, stream) : launch_bf<BfCfg<2, 1, 2, 8, 8, 16, 1, false, true>>(a, stream);
  }
  if (k == 3 && s == 2) return dispatch_cc<3, 2>(a, stream);
  return MR3D_ERR_UNSUPPORTED;
}

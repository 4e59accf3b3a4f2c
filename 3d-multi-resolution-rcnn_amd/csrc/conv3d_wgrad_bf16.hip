// bf16 weight gradient of the 3D convolutions (configs[4]; the wgrad half of autograd's backward of the Conv3d calls of
// mmdet/models/backbones/resnet3d.py:126,139,173,251 and necks/fpn3d.py:48-64):
//   dw[tap][ci][co] += sum_p x[p*stride + tap - pad][ci] * dy[p][co]         (fp32 accumulation, fp32 dw)
// GEMM per tap: M = input channels, N = output channels, K = positions.  x and dy are bf16 NDHWC, i.e. both operands have
// K as their SLOW axis in memory; the tile's x halo [position][32 ci] and dy tile [position][32 co] are staged in LDS as they
// come (coalesced 16-byte rows) and the MFMA fragments -- 8 consecutive positions of one channel -- are read with gfx950's
// transposing LDS read (ds_read_b64_tr_b16: a 4-position x 16-channel block per 16 lanes, delivered channel-major), two per
// fragment.  A k-step (16 positions) is one W-row of the tile (TW = 16).
//   * 3x3x3: the four waves split the 27 taps (7/7/7/6), one 32x32 fp32 accumulator tile per tap in registers; the dy
//     fragment of a k-step is shared by all taps; wave 3's spare slot multiplies a fragment of ones with dy: sum_p dy
//     (bias / BN-beta gradient) from the matrix pipe;
//   * 1x1x1: the waves split the k-steps and their tiles meet in LDS before ONE atomic per element and workgroup.
// Workgroups are persistent over position tiles and add their sums to dw with fp32 atomics at the end.
#include "conv3d.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
using bf16_t = uint16_t;
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

constexpr int kThreads = 256;

struct WgArgs {
  const bf16_t *x; const bf16_t *dy; float *dw; float *dbias;
  int N, D, H, W, Cin;
  int Do, Ho, Wo, Cout;
  int pd, ph, pw;
};

template <int K_, int S_, int TD_, int TH_>
struct WgCfg {
  static constexpr int K = K_, S = S_, TD = TD_, TH = TH_, TW = 16;
  static constexpr bool PW1 = (K_ == 1);
  static constexpr int LS = PW1 ? 1 : S_, GS = PW1 ? S_ : 1;
  static constexpr int HD = (TD - 1) * LS + K, HH = (TH - 1) * LS + K, HW = (TW - 1) * LS + K, HP = HD * HH * HW;
  static constexpr int M = TD * TH * TW, KS = TD * TH;          // positions, k-steps per tile
  static constexpr int TAPS = K * K * K;
  static constexpr int NTW = PW1 ? 1 : (TAPS + 3) / 4;          // accumulator slots per wave (taps, + the ones slot)
  // bytes per staged position.  A transposing read (ds_read_b64_tr_b16, 32 lanes per LDS pass) takes 4 positions x 64 B: unpadded
  // 64-byte rows put them on four disjoint 16-bank blocks of the 64-bank row (80-byte rows overlapped two of them: 2-way, 50 % of
  // the LDS-active cycles); a stride-2 halo (every second position) needs 96.
  static constexpr int XROWB = (LS == 2) ? 96 : 64, YROWB = 64;
  static constexpr int XBYTES = HP * XROWB, YBYTES = M * YROWB;
  static constexpr int RED_BYTES = PW1 ? 4 * 2 * 32 * 33 * 4 : 0;   // 1x1x1: the four waves' (dw, sum dy) tiles meet here
  static constexpr int LDS_BYTES = (XBYTES + YBYTES > RED_BYTES) ? XBYTES + YBYTES : RED_BYTES;
  // 3x3x3 stride 1: ONE workgroup per CU with the whole 512-register budget -- both operands of the next tile prefetched and the
  // MFMA fragments double-buffered across k-steps (nothing else on the CU would hide their latencies)
  static constexpr int MINWG = (!PW1 && S_ == 1) ? 1 : 2;
  static_assert(LDS_BYTES <= 160 * 1024, "tile exceeds LDS");
};

__device__ __forceinline__ bf16x8 tr_frag(unsigned char *lds, int off0, int off1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + off1));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// SPARSE (MR3D_WGRAD_SPARSE_DY; stride-1 instances): every dy tile is tested while it is staged and the MFMA loop of a zero tile is
// skipped -- in detector training the gradient at the RPN convs, the FPN output / offset convs is zero on most position tiles
template <class C, bool SPARSE>
__global__ __launch_bounds__(kThreads, C::MINWG) void conv3d_wgrad_bf16_kernel(WgArgs p, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *xs = lds, *ys = lds + C::XBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // transposing-read roles of this lane: 16-lane group g reads channel block cb = g&1 for k half hh = g>>1;
  // inside the group lane 4q+pp supplies the address of row q (a position), channels 4pp..4pp+3
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, cb = g & 1, hh = g >> 1;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
  const int a_lane = ((8 * hh + q) * C::LS) * C::XROWB + (16 * cb + 4 * pp) * 2;     // x halo: k-th position of a row is LS apart
  const int b_lane = (8 * hh + q) * C::YROWB + (16 * cb + 4 * pp) * 2;
  const bool want_bias = p.dbias != nullptr && blockIdx.y == 0;

  f32x16_t acc[C::NTW + (C::PW1 ? 1 : 0)];
#pragma unroll
  for (int t = 0; t < C::NTW + (C::PW1 ? 1 : 0); ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // ---- tile staging, software-pipelined: the NEXT tile's x halo block and dy tile are fetched into registers before the
  // MFMA loop of the current tile and written to LDS after it.  (Fetched and stored in one loop, the global round trip of
  // every tile sat between two barriers with the matrix pipe idle: 16 k-steps x 7 MFMAs are 1.5 us, the fetch 2-3 us.)
  // register budget: 1x1x1 (two workgroups per CU) and 3x3x3 stride 1 (one per CU) prefetch both operands; stride 2 neither.
  constexpr bool PFX = C::PW1 || C::S == 1, PFY = C::PW1 || C::MINWG == 1;
  constexpr int NXR = (C::HP * 4 + kThreads - 1) / kThreads, NYR = (C::M * 4 + kThreads - 1) / kThreads;
  uint4 xr[NXR], yr[NYR];
  // per-thread staging slots, tabulated once: element offset from the tile's corner and packed halo / tile coordinates
  int xoff[NXR], yoff[NYR];
  unsigned xcrd[NXR], ycrd[NYR];
  constexpr unsigned kNoSlot = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < NXR; ++j) {
    const int it = tid + j * kThreads;
    const int pos = it >> 2, qq = it & 3;
    const int hw = (pos % C::HW) * C::GS, hh_ = ((pos / C::HW) % C::HH) * C::GS, hd = (pos / (C::HW * C::HH)) * C::GS;
    xoff[j] = ((hd * p.H + hh_) * p.W + hw) * p.Cin + qq * 8;
    xcrd[j] = (it < C::HP * 4 && ci0 + qq * 8 < p.Cin) ? (unsigned)(hd | (hh_ << 8) | (hw << 16)) : kNoSlot;
  }
#pragma unroll
  for (int j = 0; j < NYR; ++j) {
    const int it = tid + j * kThreads;
    const int m = it >> 2, qq = it & 3;
    const int tw = m % C::TW, th = (m / C::TW) % C::TH, td = m / (C::TW * C::TH);
    yoff[j] = ((td * p.Ho + th) * p.Wo + tw) * p.Cout + qq * 8;
    ycrd[j] = (it < C::M * 4 && co0 + qq * 8 < p.Cout) ? (unsigned)(td | (th << 8) | (tw << 16)) : kNoSlot;
  }
  auto fetch_part = [&](int tile, bool do_x, bool do_y) {
    int b = tile;
    const int tw_i = b % tiles_w; b /= tiles_w;
    const int th_i = b % tiles_h; b /= tiles_h;
    const int td_i = b % tiles_d; b /= tiles_d;
    const int n = b;
    const int od0 = td_i * C::TD, oh0 = th_i * C::TH, ow0 = tw_i * C::TW;
    const int id0 = od0 * C::S - p.pd, ih0 = oh0 * C::S - p.ph, iw0 = ow0 * C::S - p.pw;
    const bool live = tile < ntiles;
    // corners of the halo block / dy tile (may lie outside the tensor: only in-range slots are dereferenced)
    const bf16_t *xb = p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * (long long)p.Cin + ci0;
    const bf16_t *yb = p.dy + ((((long long)n * p.Do + od0) * p.Ho + oh0) * p.Wo + ow0) * (long long)p.Cout + co0;
    if (do_x)
#pragma unroll
      for (int j = 0; j < NXR; ++j) {
        const unsigned c = xcrd[j];
        const bool ok = live && c != kNoSlot && (unsigned)(id0 + (int)(c & 255u)) < (unsigned)p.D &&
                        (unsigned)(ih0 + (int)((c >> 8) & 255u)) < (unsigned)p.H && (unsigned)(iw0 + (int)(c >> 16)) < (unsigned)p.W;
        xr[j] = ok ? *reinterpret_cast<const uint4 *>(xb + xoff[j]) : make_uint4(0u, 0u, 0u, 0u);
      }
    if (do_y)
#pragma unroll
      for (int j = 0; j < NYR; ++j) {
        const unsigned c = ycrd[j];
        const bool ok = live && c != kNoSlot && od0 + (int)(c & 255u) < p.Do && oh0 + (int)((c >> 8) & 255u) < p.Ho &&
                        ow0 + (int)(c >> 16) < p.Wo;
        yr[j] = ok ? *reinterpret_cast<const uint4 *>(yb + yoff[j]) : make_uint4(0u, 0u, 0u, 0u);
      }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int it = tid + j * kThreads;
      if (it < C::HP * 4) *reinterpret_cast<uint4 *>(xs + (it >> 2) * C::XROWB + (it & 3) * 16) = xr[j];
    }
#pragma unroll
    for (int j = 0; j < NYR; ++j) {
      const int it = tid + j * kThreads;
      if (it < C::M * 4) *reinterpret_cast<uint4 *>(ys + (it >> 2) * C::YROWB + (it & 3) * 16) = yr[j];
    }
  };

  // stride 2: the first round's plain staging (global -> LDS slot by slot, nothing held across the MFMA loop)
  auto stage_direct = [&](int tile) {
    int b = tile;
    const int tw_i = b % tiles_w; b /= tiles_w;
    const int th_i = b % tiles_h; b /= tiles_h;
    const int td_i = b % tiles_d; b /= tiles_d;
    const int n = b;
    const int od0 = td_i * C::TD, oh0 = th_i * C::TH, ow0 = tw_i * C::TW;
    const int id0 = od0 * C::S - p.pd, ih0 = oh0 * C::S - p.ph, iw0 = ow0 * C::S - p.pw;
    for (int it = tid; it < C::HP * 4; it += kThreads) {
      const int pos = it >> 2, qq = it & 3;
      const int hw = pos % C::HW, hh_ = (pos / C::HW) % C::HH, hd = pos / (C::HW * C::HH);
      const int id = id0 + hd * C::GS, ih = ih0 + hh_ * C::GS, iw = iw0 + hw * C::GS;
      const int ci = ci0 + qq * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ci < p.Cin && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
        v = *reinterpret_cast<const uint4 *>(p.x + ((((size_t)n * p.D + id) * p.H + ih) * p.W + iw) * p.Cin + ci);
      *reinterpret_cast<uint4 *>(xs + pos * C::XROWB + qq * 16) = v;
    }
    for (int it = tid; it < C::M * 4; it += kThreads) {
      const int m = it >> 2, qq = it & 3;
      const int tw = m % C::TW, th = (m / C::TW) % C::TH, td = m / (C::TW * C::TH);
      const int od = od0 + td, oh = oh0 + th, ow = ow0 + tw;
      const int co = co0 + qq * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (co < p.Cout && od < p.Do && oh < p.Ho && ow < p.Wo)
        v = *reinterpret_cast<const uint4 *>(p.dy + ((((size_t)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * p.Cout + co);
      *reinterpret_cast<uint4 *>(ys + m * C::YROWB + qq * 16) = v;
    }
  };

  static_assert(!SPARSE || PFX, "the tile test sits in the register-staged path");
  __shared__ int live_flag[2];                          // per tile parity: some thread staged a non-zero dy value
  if (SPARSE && tid == 0) live_flag[0] = 0;
  int par = 0;
  if constexpr (SPARSE) fetch_part(blockIdx.x, false, true);          // sparse: only dy travels ahead; x is fetched for live tiles
  else if constexpr (PFX) fetch_part(blockIdx.x, PFX, PFY);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    __syncthreads();                                    // the previous tile's fragments have been read
    if constexpr (SPARSE) {
      unsigned nz = 0;
#pragma unroll
      for (int j = 0; j < NYR; ++j) nz |= (yr[j].x | yr[j].y | yr[j].z | yr[j].w) & 0x7fff7fffu;
      if (nz) live_flag[par] = 1;
      if (tid == 0) live_flag[par ^ 1] = 0;
      __syncthreads();
      const bool lv = live_flag[par] != 0;              // (workgroup-uniform)
      if (lv) {
        fetch_part(tile, true, false);                  // a dead tile's x halo is never read: that was most of a skipped tile's time
        commit();
      }
      fetch_part(tile + gridDim.x, false, true);
      if (!lv) continue;
      __syncthreads();
    } else if constexpr (PFX) {
      fetch_part(tile, false, !PFY);                    // the dy tile where it is not prefetched
      commit();
      __syncthreads();
      fetch_part(tile + gridDim.x, PFX, PFY);           // in flight during the MFMA loop
    } else {
      stage_direct(tile);
      __syncthreads();
    }

    if constexpr (C::PW1) {
      for (int ks = wave; ks < C::KS; ks += 4) {        // waves split the k-steps
        const bf16x8 bf = tr_frag(ys, b_lane + ks * 16 * C::YROWB, b_lane + (ks * 16 + 4) * C::YROWB);
        const bf16x8 af = tr_frag(xs, a_lane + ks * 16 * C::XROWB, a_lane + (ks * 16 + 4) * C::XROWB);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[0], 0, 0, 0);
        if (want_bias) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bf, acc[1], 0, 0, 0);
      }
    } else {
      // fragments of k-step ks + 1 are requested before the MFMAs of k-step ks issue (read-then-multiply per tap left every
      // MFMA waiting for a full LDS round trip: 14 us per 256-position tile for 1.5 us of matrix work)
      // slot ti of wave w is tap w + 4 ti; only the LAST slot can fall off the tap list (27 taps: wave 3's seventh slot),
      // and it is handled without a branch: the fragment of tap 26 is read anyway and replaced by ones
      constexpr int LAST = C::NTW - 1;
      const bool spare = wave + 4 * LAST >= C::TAPS;
      int tapoff[C::NTW];
#pragma unroll
      for (int ti = 0; ti < C::NTW; ++ti) {
        int tap = wave + 4 * ti;
        if (tap >= C::TAPS) tap = C::TAPS - 1;
        const int kw = tap % C::K, kh = (tap / C::K) % C::K, kd = tap / (C::K * C::K);
        tapoff[ti] = ((kd * C::HH + kh) * C::HW + kw) * C::XROWB;
      }
      auto load_frags = [&](int ks, bf16x8 (&af)[C::NTW], bf16x8 &bf) {
        const int th = ks % C::TH, td = ks / C::TH;
        bf = tr_frag(ys, b_lane + ks * 16 * C::YROWB, b_lane + (ks * 16 + 4) * C::YROWB);
        const int rowbase = a_lane + (((td * C::LS) * C::HH + th * C::LS) * C::HW) * C::XROWB;
#pragma unroll
        for (int ti = 0; ti < C::NTW; ++ti) {
          const int off = rowbase + tapoff[ti];
          af[ti] = tr_frag(xs, off, off + 4 * C::LS * C::XROWB);
        }
        if (spare) af[LAST] = ones;                      // sum_p dy on the matrix pipe (wave-uniform select)
      };
      auto mma = [&](const bf16x8 (&af)[C::NTW], const bf16x8 &bf) {
#pragma unroll
        for (int ti = 0; ti < C::NTW; ++ti) acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ti], bf, acc[ti], 0, 0, 0);
      };
      static_assert(C::KS % 2 == 0, "k-steps are processed in pairs");
      bf16x8 afA[C::NTW], afB[C::NTW], bfA, bfB;
      load_frags(0, afA, bfA);
      for (int ks = 0; ks < C::KS; ks += 2) {
        load_frags(ks + 1, afB, bfB);
        __builtin_amdgcn_sched_barrier(0);
        mma(afA, bfA);
        if (ks + 2 < C::KS) load_frags(ks + 2, afA, bfA);
        __builtin_amdgcn_sched_barrier(0);
        mma(afB, bfB);
      }
    }
  }

  // ---- write-out: C/D layout of the 32x32 tile: column (co) = lane & 31, row (ci) = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int col = lane & 31, h2 = lane >> 5;
  if constexpr (C::PW1) {
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);        // [wave][2][32][33]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h2;
        red[((wave * 2 + t) * 32 + row) * 33 + col] = acc[t][e];
      }
    __syncthreads();
    for (int it = tid; it < 32 * 32; it += kThreads) {
      const int row = it >> 5, c = it & 31;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[((w * 2 + 0) * 32 + row) * 33 + c];
      if (ci0 + row < p.Cin && co0 + c < p.Cout) atomicAdd(p.dw + (size_t)(ci0 + row) * p.Cout + co0 + c, s);
    }
    if (want_bias && tid < 32 && co0 + tid < p.Cout) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[((w * 2 + 1) * 32 + 0) * 33 + tid];
      atomicAdd(p.dbias + co0 + tid, s);
    }
  } else {
#pragma unroll
    for (int ti = 0; ti < C::NTW; ++ti) {
      const int tap = wave + 4 * ti;
      if (tap < C::TAPS) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h2;
          if (ci0 + row < p.Cin && co0 + col < p.Cout)
            atomicAdd(p.dw + ((size_t)tap * p.Cin + ci0 + row) * p.Cout + co0 + col, acc[ti][e]);
        }
      } else if (tap == C::TAPS && want_bias) {
        if (h2 == 0 && co0 + col < p.Cout) atomicAdd(p.dbias + co0 + col, acc[ti][0]);      // every row of the ones tile is sum_p dy
      }
    }
  }
}

template <class C, bool SPARSE = false>
int launch_wg(const WgArgs &a, hipStream_t stream) {
  const int tiles_d = (a.Do + C::TD - 1) / C::TD, tiles_h = (a.Ho + C::TH - 1) / C::TH, tiles_w = (a.Wo + C::TW - 1) / C::TW;
  static bool attr_done = false;
  if (!attr_done && C::LDS_BYTES > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_wgrad_bf16_kernel<C, SPARSE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const long long ntiles = (long long)a.N * tiles_d * tiles_h * tiles_w;
  if (ntiles > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  const int by = (a.Cin + 31) / 32, bz = (a.Cout + 31) / 32;
  // persistent workgroups: about two per CU across the (ci, co) blocks, each walking many position tiles
  long long gx = (C::MINWG == 1 ? 256 : 512) / ((long long)by * bz);
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  dim3 grid((unsigned)gx, (unsigned)by, (unsigned)bz);
  conv3d_wgrad_bf16_kernel<C, SPARSE><<<grid, kThreads, C::LDS_BYTES, stream>>>(a, tiles_d, tiles_h, tiles_w, (int)ntiles);
  return mr3d_launch_status();
}

}  // namespace

extern "C" int mr3d_conv3d_wgrad_bf16(const mr3d_conv3d_wgrad_desc *d, mr3d_stream_t stream_) {
  if (!d || !d->x || !d->dy || !d->dw) return MR3D_ERR_INVALID_ARG;
  if (d->kd != d->kh || d->kh != d->kw || d->sd != d->sh || d->sh != d->sw) return MR3D_ERR_UNSUPPORTED;
  WgArgs a;
  a.x = (const bf16_t *)d->x; a.dy = (const bf16_t *)d->dy; a.dw = d->dw; a.dbias = d->dbias;
  a.N = d->n; a.D = d->d; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Cout = d->cout;
  a.pd = d->pd; a.ph = d->ph; a.pw = d->pw;
  a.Do = (d->d + 2 * d->pd - d->kd) / d->sd + 1;
  a.Ho = (d->h + 2 * d->ph - d->kh) / d->sh + 1;
  a.Wo = (d->w_in + 2 * d->pw - d->kw) / d->sw + 1;
  if (a.N <= 0 || a.Do <= 0 || a.Ho <= 0 || a.Wo <= 0) return MR3D_ERR_INVALID_ARG;
  if (a.Cin % 8 != 0 || a.Cout % 8 != 0) return MR3D_ERR_UNSUPPORTED;
  hipStream_t stream = mr3d_stream(stream_);
  const int k = d->kd, s = d->sd;
  const bool sparse = (d->flags & MR3D_WGRAD_SPARSE_DY) != 0;
  if (k == 3 && s == 1 && sparse) return launch_wg<WgCfg<3, 1, 2, 8>, true>(a, stream);
  if (k == 1 && s == 1 && sparse) return launch_wg<WgCfg<1, 1, 2, 8>, true>(a, stream);
  if (k == 3 && s == 1) return launch_wg<WgCfg<3, 1, 2, 8>>(a, stream);
  if (k == 3 && s == 2) return launch_wg<WgCfg<3, 2, 1, 4>>(a, stream);
  // (512-position tiles for the 1x1x1 records of the large maps: 41 -> 37 us for 64 -> 64 at 40x128x128 alone, but the bf16 backbone
  // step 4.61 -> 4.66 ms -- twice the LDS per workgroup leaves less room beside the main chain's kernels)
  if (k == 1 && s == 1) return launch_wg<WgCfg<1, 1, 2, 8>>(a, stream);
  if (k == 1 && s == 2) return launch_wg<WgCfg<1, 2, 2, 8>>(a, stream);
  return MR3D_ERR_UNSUPPORTED;
}

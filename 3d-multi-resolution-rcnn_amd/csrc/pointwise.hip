// Pointwise (1x1x1, stride 1) convolution forward / data gradient and weight gradient for the
// HBM-bound layers of the backbone (large position counts, Cin*Cout small): bottleneck conv1 / conv3 /
// downsample of resnet3d.py:126,173,251 and the FPN laterals fpn3d.py:48-55.
//
// These layers move 3-25 FLOP per byte: they live on the HBM roofline, not the MFMA one.  Design for
// bytes in flight, not for the matrix pipe:
//   * every WAVE is an independent streaming engine: it owns a private LDS slab, walks its own sequence of
//     row tiles (ROWS consecutive positions x all Cin channels -- contiguous in NDHWC memory, so the tile is
//     one fully coalesced run of 16-B-per-lane loads) and never meets a workgroup barrier;
//   * software pipeline per wave: the global loads of tile t+1 are issued into registers before the MFMAs
//     of tile t start, and land in LDS after them -> up to 16 KiB per wave / 128 KiB per CU in flight;
//   * the weight fragments of the whole layer stay in VGPRs for the kernel's lifetime when Cin*Cout <= 4096
//     (all 655k-position layers), otherwise in LDS, loaded once per (persistent) workgroup;
//   * fp32 MFMA (32x32x2, or 16x16x4 when Cout = 16) does the arithmetic -- far below its peak by design;
//   * epilogue fused exactly like conv3d_fwd.hip (shift, two residual adds, nearest-upsample add, ReLU, mask).
// wgrad: dW[ci][co] = sum_p x[p][ci]*dy[p][co]: both operands stream; per wave both tiles are staged in its
// LDS slab, the [Cin x Cout] accumulators stay in registers across all tiles, one atomic pass at the end;
// sum_p dy[p][co] (bias / BN-beta gradient) is reduced in the same pass.
#include "conv3d.h"
#include <type_traits>

namespace {

constexpr int kWavesPerWG = 4;

__device__ __forceinline__ int pw_nearest_src(int dst, int in, int out) {
  int s = (int)floorf((float)dst * ((float)in / (float)out));
  return s < in - 1 ? s : in - 1;
}

// EXTRA = the call carries residual / mask streams (add1, add2, mask): 32-row tiles, and the add1 / mask tiles of
// the CURRENT tile are requested at the top of the iteration so that their HBM latency hides behind the LDS
// round trip and the MFMAs instead of being exposed in the epilogue.
// C2_ > 0: the PAIR form -- a second pointwise product CO -> C2 on the activated output tile while it still sits in the wave's
// LDS slab (a bottleneck block's conv3 + identity + ReLU followed by the next block's conv1 + ReLU, and the mirror image in the
// backward pass): the 4x-wide block output is written once and not read back by another launch.
// BLK_: the launch covers a WIDER layer in column blocks of CO_ output channels (blockIdx.y = block): weights, shift, y and the
// residual / mask streams have a row pitch of p.Cout; whole tiles, no upsample add, no scatter (the 64 -> 256 convolutions of
// layer3 on the 768 x 768 x 120 pathway of the shipped model: 34,560 positions took the generic halo-tile kernel, 60-160 us each).
template <int CI_, int CO_, bool EXTRA_ = false, int C2_ = 0, bool BLK_ = false>
struct PwCfg {
  static constexpr int CI = CI_, CO = CO_, C2 = C2_;
  static constexpr bool EXTRA = EXTRA_, PAIR = C2_ > 0, BLK = BLK_;
  static_assert(!(BLK_ && C2_ > 0), "column blocks: no pair form");
  static constexpr int KK2 = CO_ / 4, MT2 = 2, OS2 = C2_ + 4;            // stage 2: 16x16x4, both 16-row tiles of the 32-row tile
  static constexpr int NOUT2 = C2_ > 0 ? 32 * C2_ / 4 / 64 : 1;
  static_assert(C2_ == 0 || (C2_ == 16 && EXTRA_ && CO_ % 4 == 0), "pair: 16-wide second product on 32-row tiles");
  static constexpr bool k16 = (CO == 16);
  static constexpr int TM = k16 ? 16 : 32;
  static constexpr int KSTEP = k16 ? 4 : 2;
  static constexpr int KK = CI / KSTEP;
  static constexpr int NT = CO / TM;
  static constexpr bool BREG = (KK * NT <= 64);            // weights resident in VGPRs
  static constexpr int ROWS = (!EXTRA && CI <= 64 && CO <= 64) ? 64 : 32;   // rows per wave tile (register budget)
  static constexpr int MT = ROWS / TM;
  static constexpr int RS = CI + 4;                        // LDS row stride (floats), keeps 16-B alignment
  static constexpr int OS = CO + 4;                        // LDS row stride of the output tile
  static constexpr int SLAB = ROWS * (RS > OS ? RS : OS);  // floats per wave: x tile, later reused for the y tile
  static constexpr int NOUT = ROWS * CO / 4 / 64;          // float4 per lane of the output tile
  static constexpr int WLDS = BREG ? 0 : CI * CO;          // floats of weights in LDS
  // Waves never talk to each other (weights in registers): ONE wave per workgroup, so that a workgroup needs 9-35 KB of
  // LDS and fits beside a 107 KB weight-gradient workgroup of the side stream -- with four waves (70-140 KB) the
  // data-gradient chain waited for whole CUs to drain (the 64->64 lateral data gradient took 460 us beside the P2
  // weight gradient, 80 us alone).  Weights in LDS (64x128, 128x64): four waves share one copy.
  static constexpr int WPW = BREG ? 1 : kWavesPerWG;
  static constexpr int LDS_BYTES = (WLDS + WPW * SLAB) * 4;
  static constexpr int NLOAD = ROWS * CI / 4 / 64;         // float4 loads per lane per tile
  static constexpr int ACC = k16 ? 4 : 16;
  static_assert(ROWS * CI % 256 == 0 && ROWS * CO % 256 == 0, "tiles must be whole numbers of wave-wide float4 accesses");
};

template <class C>
__global__ __launch_bounds__(64 * C::WPW) void pointwise_fwd_kernel(Conv3dFwdArgs p, long long rows_total) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float *slab = lds + C::WLDS + wave * C::SLAB;
  const int li = lane % C::TM, lk = lane / C::TM;
  using acc_t = typename std::conditional<C::k16, f32x4, f32x16>::type;

  // ---- weights: registers (whole layer) or LDS
  float breg[C::BREG ? C::KK : 1][C::BREG ? C::NT : 1];
  if constexpr (C::BREG) {
#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
        breg[kk][nt] = C::BLK ? p.w[(size_t)(kk * C::KSTEP + lk) * p.Cout + blockIdx.y * C::CO + nt * C::TM + li]
                              : p.w[(size_t)(kk * C::KSTEP + lk) * C::CO + nt * C::TM + li];
  } else {
    static_assert(!C::BLK, "column blocks: weights in registers only");
    for (int e = tid; e < C::CI * C::CO / 4; e += 64 * C::WPW)
      *reinterpret_cast<float4 *>(lds + e * 4) = *reinterpret_cast<const float4 *>(p.w + (size_t)e * 4);
    __syncthreads();
  }
  // float4 index of element f = (row, channel quad) of the tile that starts at row0, in y / add1 / add2 / mask
  auto oidx = [&](long long row0, int f) -> size_t {
    if constexpr (C::BLK) return (size_t)(row0 + f / (C::CO / 4)) * (size_t)(p.Cout / 4) + blockIdx.y * (C::CO / 4) + f % (C::CO / 4);
    else return (size_t)row0 * (C::CO / 4) + f;
  };

  float breg2[C::PAIR ? C::KK2 : 1];
  if constexpr (C::PAIR) {
#pragma unroll
    for (int kk = 0; kk < C::KK2; ++kk) breg2[kk] = p.w2[(size_t)(kk * 4 + (lane >> 4)) * C::C2 + (lane & 15)];
  }

  const long long ntiles = (rows_total + C::ROWS - 1) / C::ROWS;
  const long long wstride = (long long)gridDim.x * C::WPW;
  long long tile = (long long)blockIdx.x * C::WPW + wave;

  float4 pre[C::NLOAD];
  auto issue_loads = [&](long long t) {
    const long long f0 = t * (C::ROWS * C::CI / 4);               // float4 index of the tile start
    const long long fmax = rows_total * (C::CI / 4);
#pragma unroll
    for (int q = 0; q < C::NLOAD; ++q) {
      const long long f = f0 + q * 64 + lane;
      pre[q] = (f < fmax) ? reinterpret_cast<const float4 *>(p.x)[f] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (tile < ntiles) issue_loads(tile);

  for (; tile < ntiles; tile += wstride) {
    const bool full_tile = (tile + 1) * C::ROWS <= rows_total;
    float4 e1[C::EXTRA ? C::NOUT : 1], em[C::EXTRA ? C::NOUT : 1], eu[C::EXTRA ? C::NOUT : 1];
    if constexpr (C::EXTRA) {
      if (full_tile) {
        const size_t t4e = (size_t)tile * C::ROWS * (C::CO / 4);
        if (p.up) {        // nearest-upsampled coarser map (FPN top-down add): gathered rows, requested early too
#pragma unroll
          for (int q = 0; q < C::NOUT; ++q) {
            const int f = q * 64 + lane;
            unsigned pq = (unsigned)(tile * C::ROWS + f / (C::CO / 4));
            const unsigned ow = pq % (unsigned)p.Wo; pq /= (unsigned)p.Wo;
            const unsigned oh = pq % (unsigned)p.Ho; pq /= (unsigned)p.Ho;
            const unsigned od = pq % (unsigned)p.Do;
            const unsigned n = pq / (unsigned)p.Do;
            const bool ev = p.up_even != 0;
            const int ud = ev ? (int)(od >> 1) : pw_nearest_src((int)od, p.Du, p.Do), uh = ev ? (int)(oh >> 1) : pw_nearest_src((int)oh, p.Hu, p.Ho),
                      uw = ev ? (int)(ow >> 1) : pw_nearest_src((int)ow, p.Wu, p.Wo);
            eu[q] = (ev && ((od | oh | ow) & 1u)) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(
                p.up + ((((size_t)n * p.Du + ud) * p.Hu + uh) * p.Wu + uw) * C::CO + (f % (C::CO / 4)) * 4);
          }
        }
        if (p.add1) {
#pragma unroll
          for (int q = 0; q < C::NOUT; ++q)
            e1[q] = C::BLK ? reinterpret_cast<const float4 *>(p.add1)[oidx(tile * C::ROWS, q * 64 + lane)]
                           : (reinterpret_cast<const float4 *>(p.add1) + t4e)[q * 64 + lane];
        }
        if (p.mask) {
#pragma unroll
          for (int q = 0; q < C::NOUT; ++q)
            em[q] = C::BLK ? reinterpret_cast<const float4 *>(p.mask)[oidx(tile * C::ROWS, q * 64 + lane)]
                           : (reinterpret_cast<const float4 *>(p.mask) + t4e)[q * 64 + lane];
        }
      }
    }
    // ---- registers -> this wave's LDS slab (row-padded)
#pragma unroll
    for (int q = 0; q < C::NLOAD; ++q) {
      const int f = q * 64 + lane;
      const int row = f / (C::CI / 4), c4 = f % (C::CI / 4);
      *reinterpret_cast<float4 *>(slab + row * C::RS + c4 * 4) = pre[q];
    }
    if (tile + wstride < ntiles) issue_loads(tile + wstride);       // in flight during the MFMAs below

    acc_t acc[C::MT][C::NT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int r = 0; r < C::ACC; ++r) acc[mt][nt][r] = 0.f;

#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk) {
      float a[C::MT], b[C::NT];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) a[mt] = slab[(mt * C::TM + li) * C::RS + kk * C::KSTEP + lk];
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) {
        if constexpr (C::BREG) b[nt] = breg[kk][nt];
        else b[nt] = lds[(kk * C::KSTEP + lk) * C::CO + nt * C::TM + li];
      }
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          if constexpr (C::k16)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
          else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- epilogue.  Each optional term is one wave-uniform branch around a fully unrolled batch of
    // loads (so the loads of a term are all in flight together), never a per-element branch.
    const long long row0 = tile * C::ROWS;
    if (row0 + C::ROWS <= rows_total) {
      // (1) accumulators -> this wave's LDS slab as a [ROWS][CO] tile (the x tile is dead by now), then
      // (2) walk the tile in memory order, 16 B per lane: residual / mask loads and the store are all
      //     fully coalesced 1-KiB wave accesses instead of 4-B-per-lane column accesses.
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) {
        const float sh = p.shift ? p.shift[(C::BLK ? blockIdx.y * C::CO : 0) + nt * C::TM + li] : 0.f;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int r = 0; r < C::ACC; ++r) {
            const int row = mt * C::TM + (C::k16 ? (lk * 4 + r) : ((r & 3) + 8 * (r >> 2) + 4 * lk));
            slab[row * C::OS + nt * C::TM + li] = acc[mt][nt][r] + sh;
          }
      }
      const size_t t4 = (size_t)row0 * (C::CO / 4);           // float4 index of the tile in y / add / mask
      float4 v[C::NOUT];
#pragma unroll
      for (int q = 0; q < C::NOUT; ++q) {
        const int f = q * 64 + lane;
        v[q] = *reinterpret_cast<const float4 *>(slab + (f / (C::CO / 4)) * C::OS + (f % (C::CO / 4)) * 4);
      }
      if (p.add1) {
        const float4 *a4 = reinterpret_cast<const float4 *>(p.add1) + t4;
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) {
          float4 a;
          if constexpr (C::EXTRA) a = e1[q];
          else if constexpr (C::BLK) a = reinterpret_cast<const float4 *>(p.add1)[oidx(row0, q * 64 + lane)];
          else a = a4[q * 64 + lane];
          v[q].x += a.x; v[q].y += a.y; v[q].z += a.z; v[q].w += a.w;
        }
      }
      if (p.add2) {
        const float4 *a4 = reinterpret_cast<const float4 *>(p.add2) + t4;
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) {
          const float4 a = C::BLK ? reinterpret_cast<const float4 *>(p.add2)[oidx(row0, q * 64 + lane)] : a4[q * 64 + lane];
          v[q].x += a.x; v[q].y += a.y; v[q].z += a.z; v[q].w += a.w;
        }
      }
      if (p.up && C::EXTRA) {
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) { v[q].x += eu[q].x; v[q].y += eu[q].y; v[q].z += eu[q].z; v[q].w += eu[q].w; }
      } else if (p.up) {
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) {
          const int f = q * 64 + lane;
          unsigned pq = (unsigned)(row0 + f / (C::CO / 4));      // positions fit 32 bits (checked by the launcher)
          const unsigned ow = pq % (unsigned)p.Wo; pq /= (unsigned)p.Wo;
          const unsigned oh = pq % (unsigned)p.Ho; pq /= (unsigned)p.Ho;
          const unsigned od = pq % (unsigned)p.Do;
          const unsigned n = pq / (unsigned)p.Do;
          const bool ev = p.up_even != 0;
          const int ud = ev ? (int)(od >> 1) : pw_nearest_src((int)od, p.Du, p.Do), uh = ev ? (int)(oh >> 1) : pw_nearest_src((int)oh, p.Hu, p.Ho),
                    uw = ev ? (int)(ow >> 1) : pw_nearest_src((int)ow, p.Wu, p.Wo);
          const float4 a = (ev && ((od | oh | ow) & 1u)) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(
              p.up + ((((size_t)n * p.Du + ud) * p.Hu + uh) * p.Wu + uw) * C::CO + (f % (C::CO / 4)) * 4);
          v[q].x += a.x; v[q].y += a.y; v[q].z += a.z; v[q].w += a.w;
        }
      }
      if (p.relu) {
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) {
          v[q].x = fmaxf(v[q].x, 0.f); v[q].y = fmaxf(v[q].y, 0.f); v[q].z = fmaxf(v[q].z, 0.f); v[q].w = fmaxf(v[q].w, 0.f);
        }
      }
      if (p.mask) {
        const float4 *m4 = reinterpret_cast<const float4 *>(p.mask) + t4;
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) {
          float4 m;
          if constexpr (C::EXTRA) m = em[q];
          else if constexpr (C::BLK) m = reinterpret_cast<const float4 *>(p.mask)[oidx(row0, q * 64 + lane)];
          else m = m4[q * 64 + lane];
          v[q].x = m.x > 0.f ? v[q].x : 0.f; v[q].y = m.y > 0.f ? v[q].y : 0.f;
          v[q].z = m.z > 0.f ? v[q].z : 0.f; v[q].w = m.w > 0.f ? v[q].w : 0.f;
        }
      }
      if (p.ysd != 1 || p.ysh != 1 || p.ysw != 1 || p.yD != p.Do || p.yH != p.Ho || p.yW != p.Wo) {
        // strided scatter of whole rows (the taps of a stride-2 transposed conv, the stride-2 pointwise dgrad): every
        // row is still one contiguous CO*4-byte run, only its position changes
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) {
          const int f = q * 64 + lane;
          unsigned pq = (unsigned)(row0 + f / (C::CO / 4));
          const unsigned ow = pq % (unsigned)p.Wo; pq /= (unsigned)p.Wo;
          const unsigned oh = pq % (unsigned)p.Ho; pq /= (unsigned)p.Ho;
          const unsigned od = pq % (unsigned)p.Do;
          const unsigned n = pq / (unsigned)p.Do;
          const size_t orow = (((size_t)n * p.yD + od * p.ysd) * p.yH + oh * p.ysh) * p.yW + ow * p.ysw;
          *reinterpret_cast<float4 *>(p.y + orow * C::CO + (f % (C::CO / 4)) * 4) = v[q];
        }
      } else if constexpr (C::BLK) {
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) reinterpret_cast<float4 *>(p.y)[oidx(row0, q * 64 + lane)] = v[q];
      } else {
        float4 *y4 = reinterpret_cast<float4 *>(p.y) + t4;
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) y4[q * 64 + lane] = v[q];
      }
      if constexpr (C::PAIR) {
        // the activated tile back into the slab, second product from there
#pragma unroll
        for (int q = 0; q < C::NOUT; ++q) {
          const int f = q * 64 + lane;
          *reinterpret_cast<float4 *>(slab + (f / (C::CO / 4)) * C::OS + (f % (C::CO / 4)) * 4) = v[q];
        }
        f32x4 acc2[C::MT2];
#pragma unroll
        for (int m = 0; m < C::MT2; ++m) acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int li2 = lane & 15, lk2 = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < C::KK2; ++kk) {
          float a2[C::MT2];
#pragma unroll
          for (int m = 0; m < C::MT2; ++m) a2[m] = slab[(m * 16 + li2) * C::OS + kk * 4 + lk2];
#pragma unroll
          for (int m = 0; m < C::MT2; ++m) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[m], breg2[kk], acc2[m], 0, 0, 0);
        }
        const float sh2 = p.shift2 ? p.shift2[li2] : 0.f;
#pragma unroll
        for (int m = 0; m < C::MT2; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(m * 16 + lk2 * 4 + r) * C::OS2 + li2] = acc2[m][r] + sh2;
        const size_t t42 = (size_t)row0 * (C::C2 / 4);
#pragma unroll
        for (int q = 0; q < C::NOUT2; ++q) {
          const int f = q * 64 + lane;
          float4 o = *reinterpret_cast<const float4 *>(slab + (f / (C::C2 / 4)) * C::OS2 + (f % (C::C2 / 4)) * 4);
          if (p.relu2) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          if (p.mask2) {
            const float4 m = (reinterpret_cast<const float4 *>(p.mask2) + t42)[f];
            o.x = m.x > 0.f ? o.x : 0.f; o.y = m.y > 0.f ? o.y : 0.f; o.z = m.z > 0.f ? o.z : 0.f; o.w = m.w > 0.f ? o.w : 0.f;
          }
          (reinterpret_cast<float4 *>(p.y2) + t42)[f] = o;
        }
      }
    } else {
      // ragged last tile: per-element predicates (at most one tile per launch)
      for (int nt = 0; nt < C::NT; ++nt) {
        const int co = nt * C::TM + li;
        const float sh = p.shift ? p.shift[co] : 0.f;
        for (int mt = 0; mt < C::MT; ++mt)
          for (int r = 0; r < C::ACC; ++r) {
            const int row = C::k16 ? (lk * 4 + r) : ((r & 3) + 8 * (r >> 2) + 4 * lk);
            const long long pos = row0 + mt * C::TM + row;
            if (pos >= rows_total) continue;
            size_t idx = (size_t)pos * C::CO + co;
            if (p.ysd != 1 || p.ysh != 1 || p.ysw != 1 || p.yD != p.Do || p.yH != p.Ho || p.yW != p.Wo) {
              unsigned q = (unsigned)pos;
              const unsigned ow = q % (unsigned)p.Wo; q /= (unsigned)p.Wo;
              const unsigned oh = q % (unsigned)p.Ho; q /= (unsigned)p.Ho;
              const unsigned od = q % (unsigned)p.Do;
              const unsigned n = q / (unsigned)p.Do;
              idx = ((((size_t)n * p.yD + od * p.ysd) * p.yH + oh * p.ysh) * p.yW + ow * p.ysw) * C::CO + co;
            }
            float v = acc[mt][nt][r] + sh;
            if (p.add1) v += p.add1[idx];
            if (p.add2) v += p.add2[idx];
            if (p.up) {
              unsigned q = (unsigned)pos;
              const unsigned ow = q % (unsigned)p.Wo; q /= (unsigned)p.Wo;
              const unsigned oh = q % (unsigned)p.Ho; q /= (unsigned)p.Ho;
              const unsigned od = q % (unsigned)p.Do;
              const unsigned n = q / (unsigned)p.Do;
              if (p.up_even) {
                if (!((od | oh | ow) & 1u)) v += p.up[((((size_t)n * p.Du + (od >> 1)) * p.Hu + (oh >> 1)) * p.Wu + (ow >> 1)) * C::CO + co];
              } else {
                const int ud = pw_nearest_src((int)od, p.Du, p.Do), uh = pw_nearest_src((int)oh, p.Hu, p.Ho),
                          uw = pw_nearest_src((int)ow, p.Wu, p.Wo);
                v += p.up[((((size_t)n * p.Du + ud) * p.Hu + uh) * p.Wu + uw) * C::CO + co];
              }
            }
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.mask) v = p.mask[idx] > 0.f ? v : 0.f;
            p.y[idx] = v;
          }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ wgrad
template <int CI_, int CO_>
struct PwWgCfg {
  static constexpr int CI = CI_, CO = CO_;
  static constexpr int MT = CI / 16, NT = CO / 16;        // 16x16x4 tiles of the [CI x CO] result
  static constexpr int ROWS = 32;                         // positions per wave tile (8 MFMA k-steps)
  static constexpr int XS = CI + 4, YS = CO + 4;          // LDS row strides
  static constexpr int SLAB = ROWS * (XS + YS);
  static constexpr int LDS_BYTES = kWavesPerWG * SLAB * 4;
  static constexpr int NLX = ROWS * CI / 4 / 64, NLY = ROWS * CO / 4 / 64;
  static_assert(MT * NT <= 16, "accumulators: [CI x CO] <= 4096 per wave");
  static_assert((ROWS * CI) % 256 == 0 && (ROWS * CO) % 256 == 0, "whole wave-wide float4 loads");
};

template <class C>
__global__ __launch_bounds__(256) void pointwise_wgrad_kernel(Conv3dWgradArgs p, long long rows_total) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float *xs = lds + wave * C::SLAB;
  float *ys = xs + C::ROWS * C::XS;
  const int li = lane & 15, lk = lane >> 4;

  f32x4 acc[C::MT][C::NT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[C::NT];
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) bsum[nt] = 0.f;

  const long long ntiles = (rows_total + C::ROWS - 1) / C::ROWS;
  const long long wstride = (long long)gridDim.x * kWavesPerWG;
  long long tile = (long long)blockIdx.x * kWavesPerWG + wave;
  float4 px[C::NLX], py[C::NLY];
  auto issue_loads = [&](long long t) {
    const long long fx0 = t * (C::ROWS * C::CI / 4), fxmax = rows_total * (C::CI / 4);
    const long long fy0 = t * (C::ROWS * C::CO / 4), fymax = rows_total * (C::CO / 4);
#pragma unroll
    for (int q = 0; q < C::NLX; ++q) {
      const long long f = fx0 + q * 64 + lane;
      px[q] = (f < fxmax) ? reinterpret_cast<const float4 *>(p.x)[f] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < C::NLY; ++q) {
      const long long f = fy0 + q * 64 + lane;
      py[q] = (f < fymax) ? reinterpret_cast<const float4 *>(p.dy)[f] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (tile < ntiles) issue_loads(tile);

  for (; tile < ntiles; tile += wstride) {
#pragma unroll
    for (int q = 0; q < C::NLX; ++q) {
      const int f = q * 64 + lane;
      *reinterpret_cast<float4 *>(xs + (f / (C::CI / 4)) * C::XS + (f % (C::CI / 4)) * 4) = px[q];
    }
#pragma unroll
    for (int q = 0; q < C::NLY; ++q) {
      const int f = q * 64 + lane;
      *reinterpret_cast<float4 *>(ys + (f / (C::CO / 4)) * C::YS + (f % (C::CO / 4)) * 4) = py[q];
    }
    if (tile + wstride < ntiles) issue_loads(tile + wstride);
#pragma unroll
    for (int ks = 0; ks < C::ROWS / 4; ++ks) {
      const int row = ks * 4 + lk;                      // A[i = ci][k = position], B[k = position][j = co]
      float a[C::MT], b[C::NT];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) a[mt] = xs[row * C::XS + mt * 16 + li];
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) { b[nt] = ys[row * C::YS + nt * 16 + li]; bsum[nt] += b[nt]; }
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
  }

  // ---- the four waves' [CI x CO] partials are summed through LDS first, then ONE atomic per element per workgroup:
  // every workgroup of the launch adds into the same CI*CO addresses, and same-address device-scope atomics serialise
  // (per-wave atomics: 2048 waves x 1024 elements on the 16->64 layer)
  static_assert(kWavesPerWG * C::CI * C::CO * 4 <= C::LDS_BYTES, "partials fit the slab storage");
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        lds[wave * (C::CI * C::CO) + (mt * 16 + lk * 4 + r) * C::CO + nt * 16 + li] = acc[mt][nt][r];
  __syncthreads();
  float *part = p.part ? p.part + (size_t)blockIdx.x * (C::CI * C::CO) : nullptr;
  for (int e = tid; e < C::CI * C::CO; e += 256) {
    float s = lds[e];
#pragma unroll
    for (int w = 1; w < kWavesPerWG; ++w) s += lds[w * (C::CI * C::CO) + e];
    if (part) part[e] = s;               // two-stage reduction: a second kernel adds the workgroups' blocks
    else atomicAdd(p.dw + e, s);
  }
  if (p.dbias) {
    // wave-level shuffle reduction, then the 4 waves of the workgroup combine through LDS: one atomic per
    // channel per workgroup
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
      float s = bsum[nt];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lk == 0) lds[wave * C::CO + nt * 16 + li] = s;
    }
    __syncthreads();
    for (int c = tid; c < C::CO; c += 256)
      atomicAdd(p.dbias + c, lds[c] + lds[C::CO + c] + lds[2 * C::CO + c] + lds[3 * C::CO + c]);
  }
}

int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <class C, class K>
int set_lds_attr(K kernel) {
  static bool done = false;
  if (!done && C::LDS_BYTES > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    done = true;
  }
  return MR3D_OK;
}

template <int CI, int CO, bool EXTRA>
int launch_pw_fwd_impl(const Conv3dFwdArgs &a, hipStream_t stream) {
  using C = PwCfg<CI, CO, EXTRA>;
  if (set_lds_attr<C>(pointwise_fwd_kernel<C>) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const long long rows = (long long)a.N * a.Do * a.Ho * a.Wo;
  const long long ntiles = (rows + C::ROWS - 1) / C::ROWS;
  // about 8 waves per CU (as many as the four-wave form had), fewer when LDS does not allow it
  constexpr int lds_wgs = (160 * 1024) / C::LDS_BYTES < 1 ? 1 : (160 * 1024) / C::LDS_BYTES;
  constexpr int waves_per_cu = 8;   // tuning knob
  const int want_wgs = waves_per_cu / C::WPW < 1 ? 1 : waves_per_cu / C::WPW;
  const int wg_per_cu = lds_wgs < want_wgs ? lds_wgs : want_wgs;
  long long grid = (long long)num_cus() * wg_per_cu;
  const long long need = (ntiles + C::WPW - 1) / C::WPW;
  if (grid > need) grid = need;
  pointwise_fwd_kernel<C><<<(unsigned)grid, 64 * C::WPW, C::LDS_BYTES, stream>>>(a, rows);
  return mr3d_launch_status();
}

// column-blocked launch: a.Cout / CO blocks of the CI -> CO instance (whole tiles, no upsample add, no scatter: the caller checks)
template <int CI, int CO>
int launch_pw_fwd_blk(const Conv3dFwdArgs &a, hipStream_t stream) {
  const long long rows = (long long)a.N * a.Do * a.Ho * a.Wo;
  const int blocks = a.Cout / CO;
  auto go = [&](auto cfg) {
    using C = decltype(cfg);
    if (set_lds_attr<C>(pointwise_fwd_kernel<C>) != MR3D_OK) return (int)MR3D_ERR_LAUNCH;
    const long long ntiles = rows / C::ROWS;
    constexpr int lds_wgs = (160 * 1024) / C::LDS_BYTES < 1 ? 1 : (160 * 1024) / C::LDS_BYTES;
    const int wg_per_cu = lds_wgs < 8 ? lds_wgs : 8;
    long long gx = ((long long)num_cus() * wg_per_cu + blocks - 1) / blocks;
    if (gx > ntiles) gx = ntiles;
    pointwise_fwd_kernel<C><<<dim3((unsigned)gx, (unsigned)blocks), 64, C::LDS_BYTES, stream>>>(a, rows);
    return mr3d_launch_status();
  };
  if (a.add1 || a.add2 || a.mask) return go(PwCfg<CI, CO, true, 0, true>{});
  return go(PwCfg<CI, CO, false, 0, true>{});
}

template <int CI, int CO, int C2>
int launch_pw_pair(const Conv3dFwdArgs &a, hipStream_t stream) {
  using C = PwCfg<CI, CO, true, C2>;
  if (set_lds_attr<C>(pointwise_fwd_kernel<C>) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const long long rows = (long long)a.N * a.Do * a.Ho * a.Wo;
  const long long ntiles = rows / C::ROWS;
  constexpr int lds_wgs = (160 * 1024) / C::LDS_BYTES < 1 ? 1 : (160 * 1024) / C::LDS_BYTES;
  const int wg_per_cu = lds_wgs < 8 ? lds_wgs : 8;
  long long grid = (long long)num_cus() * wg_per_cu;
  if (grid > ntiles) grid = ntiles;
  pointwise_fwd_kernel<C><<<(unsigned)grid, 64 * C::WPW, C::LDS_BYTES, stream>>>(a, rows);
  return mr3d_launch_status();
}

template <int CI, int CO>
int launch_pw_fwd(const Conv3dFwdArgs &a, hipStream_t stream) {
  if (a.add1 || a.add2 || a.mask || a.up) return launch_pw_fwd_impl<CI, CO, true>(a, stream);
  return launch_pw_fwd_impl<CI, CO, false>(a, stream);
}

template <int CI, int CO>
int launch_pw_wgrad(const Conv3dWgradArgs &a, hipStream_t stream) {
  using C = PwWgCfg<CI, CO>;
  if (set_lds_attr<C>(pointwise_wgrad_kernel<C>) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const long long rows = (long long)a.N * a.Do * a.Ho * a.Wo;
  const long long ntiles = (rows + C::ROWS - 1) / C::ROWS;
  const int wg_per_cu = (160 * 1024) / C::LDS_BYTES > 2 ? 2 : ((160 * 1024) / C::LDS_BYTES < 1 ? 1 : (160 * 1024) / C::LDS_BYTES);
  long long grid = (long long)num_cus() * wg_per_cu;
  const long long need = (ntiles + kWavesPerWG * 4 - 1) / (kWavesPerWG * 4);   // >= 4 tiles per wave before the atomics
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  Conv3dWgradArgs k = a;
  if (!k.part || k.part_bytes < grid * (long long)(CI * CO * 4)) k.part = nullptr;
  pointwise_wgrad_kernel<C><<<(unsigned)grid, 256, C::LDS_BYTES, stream>>>(k, rows);
  const int st = mr3d_launch_status();
  if (st != MR3D_OK || !k.part) return st;
  return mr3d_reduce_partials(k.part, (int)grid, 1, CI, CO, 1, 1, 0, 0, CI, CO, a.dw, stream);
}

}  // namespace

// Returns MR3D_ERR_UNSUPPORTED when the shape is not one of the streaming instances: the caller then uses
// the generic halo-tile kernels of conv3d_fwd.hip / conv3d_wgrad.hip.
int mr3d_pointwise_fwd_try(const Conv3dFwdArgs &a, hipStream_t stream, bool probe_only) {
  const bool scatter = a.ysd != 1 || a.ysh != 1 || a.ysw != 1 || a.yD != a.Do || a.yH != a.Ho || a.yW != a.Wo;
  if (scatter && (a.add1 || a.add2 || a.mask || a.up)) return MR3D_ERR_UNSUPPORTED;   // residual streams index like y: generic path
  if (a.D != a.Do || a.H != a.Ho || a.W != a.Wo) return MR3D_ERR_UNSUPPORTED;      // stride 1, no padding
  const long long rows = (long long)a.N * a.Do * a.Ho * a.Wo;
  if (rows < 32768 || rows > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;             // small maps: not HBM-bound
  const int key = a.Cin * 1000 + a.Cout;
  // 64 -> 256 as four column blocks of the 64 -> 64 instance (whole 64-row tiles; the blocks re-read x from L2);
  // 128 -> 256 (layer4.0 conv1's data gradient on that pathway) as eight blocks of the 128 -> 32 instance, whose weights fit registers
  const bool blk_ok = (key == 64256 || key == 128256) && !a.up && !scatter && rows % 64 == 0;
  if (blk_ok) return probe_only ? MR3D_OK : (key == 64256 ? launch_pw_fwd_blk<64, 64>(a, stream) : launch_pw_fwd_blk<128, 32>(a, stream));
  if (probe_only) {          // would a streaming instance take this shape? (mr3d_conv3d_forward_variant)
    switch (key) {
      case 16016: case 16064: case 64016: case 64032: case 32064: case 64064: case 32128: case 128032: case 64128: case 128064:
        return MR3D_OK;
    }
    return MR3D_ERR_UNSUPPORTED;
  }
  switch (key) {
    case 16016: return launch_pw_fwd<16, 16>(a, stream);
    case 16064: return launch_pw_fwd<16, 64>(a, stream);
    case 64016: return launch_pw_fwd<64, 16>(a, stream);
    case 64032: return launch_pw_fwd<64, 32>(a, stream);
    case 32064: return launch_pw_fwd<32, 64>(a, stream);
    case 64064: return launch_pw_fwd<64, 64>(a, stream);
    case 32128: return launch_pw_fwd<32, 128>(a, stream);
    case 128032: return launch_pw_fwd<128, 32>(a, stream);
    case 64128: return launch_pw_fwd<64, 128>(a, stream);
    case 128064: return launch_pw_fwd<128, 64>(a, stream);
  }
  return MR3D_ERR_UNSUPPORTED;
}

int mr3d_pointwise_pair_stream_try(const Conv3dFwdArgs &a, int c2, hipStream_t stream, bool probe_only) {
  const long long rows = (long long)a.N * a.Do * a.Ho * a.Wo;
  // whole 32-row tiles only (the ragged-tile path of the kernel has no second product); the HBM-bound layer1 shape
  if (a.Cin != 16 || a.Cout != 64 || c2 != 16 || rows < 32768 || rows > 0x7fffffffLL || rows % 32 != 0) return MR3D_ERR_UNSUPPORTED;
  if (a.up || a.add2 || a.D != a.Do || a.H != a.Ho || a.W != a.Wo) return MR3D_ERR_UNSUPPORTED;
  if (probe_only) return MR3D_OK;
  return launch_pw_pair<16, 64, 16>(a, stream);
}

int mr3d_pointwise_wgrad_try(const Conv3dWgradArgs &a, hipStream_t stream) {
  if (a.D != a.Do || a.H != a.Ho || a.W != a.Wo) return MR3D_ERR_UNSUPPORTED;
  const long long rows = (long long)a.N * a.Do * a.Ho * a.Wo;
  if (rows < 32768) return MR3D_ERR_UNSUPPORTED;
  const int key = a.Cin * 1000 + a.Cout;
  switch (key) {
    case 16016: return launch_pw_wgrad<16, 16>(a, stream);
    case 16064: return launch_pw_wgrad<16, 64>(a, stream);
    case 64016: return launch_pw_wgrad<64, 16>(a, stream);
    case 64032: return launch_pw_wgrad<64, 32>(a, stream);
    case 64064: return launch_pw_wgrad<64, 64>(a, stream);
    case 32128: return launch_pw_wgrad<32, 128>(a, stream);
    case 128032: return launch_pw_wgrad<128, 32>(a, stream);
  }
  return MR3D_ERR_UNSUPPORTED;
}

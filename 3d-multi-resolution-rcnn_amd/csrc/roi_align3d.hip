// RoIAlign3D forward / backward for gfx950.
//
// Arithmetic follows the reference kernels (mmdet/ops/roi_align/src/roi_align_kernel.cu):
//   roi geometry :232-259, sample coordinates :271-281, trilinear taps :64-149 / :383-442,
//   mean over samples :288, backward scatter g_k = top*w_k/count :596-619,
//   backward's top_grad index bug :554-555 (kept behind MR3D_ROIALIGN_BUG_COMPAT).
//
// Two kernel families:
//   *_generic : one lane per output element, arbitrary feature strides -- the reference's
//               decomposition; used for NCDHW features and for sample_num == 0.
//   *_cl      : channels-last (NDHWC) features.  A wave owns one output bin and its 64 lanes
//               are 64 consecutive channels, so each trilinear tap is ONE coalesced 256-B read
//               (forward) or ONE 256-B contiguous fp32 atomic (backward: the full-rate atomic
//               shape on MI355X).  Per-axis tap tables (index/weight per sample, shared by all
//               channels and all bins of the RoI) are built once per workgroup in LDS; the
//               [bin][channel] tile is transposed through LDS so the [R,C,PD,PH,PW] output /
//               top_grad side is accessed in 256-B runs as well.
#include "mr3d_common.h"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include <type_traits>

namespace {

struct RoiGeom {
  int b;
  float sw, sh, sd;   // roi_start_{w,h,d}
  float bw, bh, bd;   // bin sizes
  int nw, nh, nd;     // samples per bin
};

// (roi_align_kernel.cu:232-259 op by op, WITHOUT fused multiply-adds like sample_coord below: `end * scale - start` contracted into
// one fma moves an extent by an ulp, i.e. a sample coordinate near x = 175 by 1.5e-5 and an output by up to 5e-5 of the map's scale)
__device__ __forceinline__ RoiGeom roi_geom(const float *__restrict__ roi, float scale, float scale_d, int sample_num,
                                            int PD, int PH, int PW) {
  RoiGeom g;
  g.b = (int)roi[0];
  g.sw = __fmul_rn(roi[1], scale);
  g.sh = __fmul_rn(roi[2], scale);
  const float ew = __fmul_rn(__fadd_rn(roi[3], 1.f), scale);
  const float eh = __fmul_rn(__fadd_rn(roi[4], 1.f), scale);
  g.sd = __fmul_rn(roi[5], scale_d);
  const float ed = __fmul_rn(__fadd_rn(roi[6], 1.f), scale_d);
  const float rw = fmaxf(__fsub_rn(ew, g.sw), 0.f);
  const float rh = fmaxf(__fsub_rn(eh, g.sh), 0.f);
  const float rd = fmaxf(__fsub_rn(ed, g.sd), 0.f);
  g.bd = rd / PD;
  g.bh = rh / PH;
  g.bw = rw / PW;
  g.nd = (sample_num > 0) ? sample_num : (int)ceilf(rd / PD);
  g.nh = (sample_num > 0) ? sample_num : (int)ceilf(rh / PH);
  g.nw = (sample_num > 0) ? sample_num : (int)ceilf(rw / PW);
  return g;
}

// one axis of bilinear_interpolate_3d (:69-103,:105-110): returns false when the coordinate is
// outside (-1, dim) (the whole sample then contributes 0 / scatters nothing).
struct AxisTap {
  int lo, hi;
  float l, h;   // weight of hi, weight of lo
};

__device__ __forceinline__ bool axis_tap(float v, int dim, AxisTap &t) {
  if (v < -1.0f || v > (float)dim) return false;
  if (v <= 0) v = 0;
  int lo = (int)v;
  if (lo >= dim - 1) {
    t.hi = t.lo = dim - 1;
    v = (float)t.lo;
  } else {
    t.lo = lo;
    t.hi = lo + 1;
  }
  t.l = v - (float)t.lo;
  t.h = 1.f - t.l;
  return true;
}

// roi_align_kernel.cu:271-281, op by op WITHOUT fused multiply-adds: a contracted `start + p * bin` moves a coordinate near 100
// by one ulp (8e-6), i.e. a tap weight by 1e-5 -- every kernel of this file then agrees with the unfused restatement to rounding
__device__ __forceinline__ float sample_coord(float start, int p, float bin, int i, int ns) {
  return __fadd_rn(__fadd_rn(start, __fmul_rn((float)p, bin)), __fdiv_rn(__fmul_rn((float)(i + .5f), bin), (float)ns));
}

// ------------------------------------------------------------------------------------------
// generic: lane per output element, feature strides (sN,sC,sD,sH,sW) in elements
// ------------------------------------------------------------------------------------------
struct Strides { long long n, c, d, h, w; };

__global__ __launch_bounds__(256) void roi_align3d_fwd_generic(long long total, const float *__restrict__ feat,
                                                               const float *__restrict__ rois, float scale, float scale_d,
                                                               int sample_num, int C, int D, int H, int W, int PD, int PH,
                                                               int PW, Strides st, float *__restrict__ out) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)blockDim.x * gridDim.x) {
    const int pw = (int)(index % PW);
    const int ph = (int)((index / PW) % PH);
    const int pd = (int)((index / PW / PH) % PD);
    const int c = (int)((index / PW / PH / PD) % C);
    const int n = (int)(index / PW / PH / PD / C);
    const RoiGeom g = roi_geom(rois + (size_t)n * 7, scale, scale_d, sample_num, PD, PH, PW);
    const float *src = feat + g.b * st.n + c * st.c;
    float acc = 0.f;
    for (int iz = 0; iz < g.nd; ++iz) {
      AxisTap tz;
      const bool vz = axis_tap(sample_coord(g.sd, pd, g.bd, iz, g.nd), D, tz);
      for (int iy = 0; iy < g.nh; ++iy) {
        AxisTap ty;
        const bool vy = axis_tap(sample_coord(g.sh, ph, g.bh, iy, g.nh), H, ty);
        for (int ix = 0; ix < g.nw; ++ix) {
          AxisTap tx;
          const bool vx = axis_tap(sample_coord(g.sw, pw, g.bw, ix, g.nw), W, tx);
          if (!(vz && vy && vx)) continue;
          const float *zl = src + tz.lo * st.d, *zh = src + tz.hi * st.d;
          const long long yl = ty.lo * st.h, yh = ty.hi * st.h, xl = tx.lo * st.w, xh = tx.hi * st.w;
          const float v1 = zl[yl + xl], v2 = zl[yl + xh], v3 = zl[yh + xl], v4 = zl[yh + xh];
          const float v5 = zh[yl + xl], v6 = zh[yl + xh], v7 = zh[yh + xl], v8 = zh[yh + xh];
          const float w1 = tx.h * ty.h * tz.h, w5 = tx.h * ty.h * tz.l;
          const float w2 = tx.l * ty.h * tz.h, w6 = tx.l * ty.h * tz.l;
          const float w3 = tx.h * ty.l * tz.h, w7 = tx.h * ty.l * tz.l;
          const float w4 = tx.l * ty.l * tz.h, w8 = tx.l * ty.l * tz.l;
          acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + w5 * v5 + w6 * v6 + w7 * v7 + w8 * v8);
        }
      }
    }
    acc /= (float)(g.nd * g.nh * g.nw);
    out[index] = acc;
  }
}

__global__ __launch_bounds__(256) void roi_align3d_bwd_generic(long long total, const float *__restrict__ top,
                                                               const float *__restrict__ rois, float scale, float scale_d,
                                                               int sample_num, int C, int D, int H, int W, int PD, int PH,
                                                               int PW, Strides st, int bug_compat,
                                                               float *__restrict__ bottom) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
       index += (long long)blockDim.x * gridDim.x) {
    const int pw = (int)(index % PW);
    const int ph = (int)((index / PW) % PH);
    const int pd = (int)((index / PW / PH) % PD);
    const int c = (int)((index / PW / PH / PD) % C);
    const int n = (int)(index / PW / PH / PD / C);
    const RoiGeom g = roi_geom(rois + (size_t)n * 7, scale, scale_d, sample_num, PD, PH, PW);
    float *dst = bottom + g.b * st.n + c * st.c;
    const long long off = (long long)(n * C + c) * PD * PH * PW + (bug_compat ? pd * PD * PW : pd * PH * PW) + ph * PW + pw;
    const float tg = top[off];
    const float count = (float)(g.nd * g.nh * g.nw);
    for (int iz = 0; iz < g.nd; ++iz) {
      AxisTap tz;
      const bool vz = axis_tap(sample_coord(g.sd, pd, g.bd, iz, g.nd), D, tz);
      for (int iy = 0; iy < g.nh; ++iy) {
        AxisTap ty;
        const bool vy = axis_tap(sample_coord(g.sh, ph, g.bh, iy, g.nh), H, ty);
        for (int ix = 0; ix < g.nw; ++ix) {
          AxisTap tx;
          const bool vx = axis_tap(sample_coord(g.sw, pw, g.bw, ix, g.nw), W, tx);
          if (!(vz && vy && vx)) continue;
          float *zl = dst + tz.lo * st.d, *zh = dst + tz.hi * st.d;
          const long long yl = ty.lo * st.h, yh = ty.hi * st.h, xl = tx.lo * st.w, xh = tx.hi * st.w;
          atomicAdd(zl + yl + xl, tg * (tx.h * ty.h * tz.h) / count);
          atomicAdd(zl + yl + xh, tg * (tx.l * ty.h * tz.h) / count);
          atomicAdd(zl + yh + xl, tg * (tx.h * ty.l * tz.h) / count);
          atomicAdd(zl + yh + xh, tg * (tx.l * ty.l * tz.h) / count);
          atomicAdd(zh + yl + xl, tg * (tx.h * ty.h * tz.l) / count);
          atomicAdd(zh + yl + xh, tg * (tx.l * ty.h * tz.l) / count);
          atomicAdd(zh + yh + xl, tg * (tx.h * ty.l * tz.l) / count);
          atomicAdd(zh + yh + xh, tg * (tx.l * ty.l * tz.l) / count);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// typed generic forms.  The reference dispatches float / double / half (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// roi_align_kernel.cu:323,674): features, rois and output share the element type T.  double computes in double; half
// computes in fp32 and rounds once per output element (forward) / per atomic add (backward; a 32-bit compare-and-swap on the
// word holding the half, as the reference's gpuAtomicAdd(Half*) does) -- the reference rounds after every half operation.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float rd_as(const __half *p, float) { return __half2float(*p); }
__device__ __forceinline__ double rd_as(const double *p, double) { return *p; }
__device__ __forceinline__ void wr_as(__half *p, float v) { *p = __float2half(v); }
__device__ __forceinline__ void wr_as(double *p, double v) { *p = v; }
__device__ __forceinline__ void add_as(double *p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void add_as(__half *p, float v) {
  unsigned int *word = reinterpret_cast<unsigned int *>(reinterpret_cast<size_t>(p) & ~(size_t)3);
  const bool hi = (reinterpret_cast<size_t>(p) & 2) != 0;
  unsigned int old = *word, assumed;
  do {
    assumed = old;
    const unsigned short cur = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
    const unsigned short nxt = __half_as_ushort(__float2half(__half2float(__ushort_as_half(cur)) + v));
    const unsigned int repl = hi ? ((assumed & 0xffffu) | ((unsigned int)nxt << 16)) : ((assumed & 0xffff0000u) | nxt);
    old = atomicCAS(word, assumed, repl);
  } while (old != assumed);
}

template <typename A>
struct RoiGeomT {
  int b;
  A sw, sh, sd, bw, bh, bd;
  int nw, nh, nd;
};

template <typename T, typename A>
__device__ __forceinline__ RoiGeomT<A> roi_geom_t(const T *roi, A scale, A scale_d, int sample_num, int PD, int PH, int PW) {
  RoiGeomT<A> g;
  const A zero = 0, one = 1;
  g.b = (int)rd_as(roi, zero);
  g.sw = rd_as(roi + 1, zero) * scale;
  g.sh = rd_as(roi + 2, zero) * scale;
  const A ew = (rd_as(roi + 3, zero) + one) * scale, eh = (rd_as(roi + 4, zero) + one) * scale;
  g.sd = rd_as(roi + 5, zero) * scale_d;
  const A ed = (rd_as(roi + 6, zero) + one) * scale_d;
  const A rw = ew - g.sw > zero ? ew - g.sw : zero, rh = eh - g.sh > zero ? eh - g.sh : zero, rd = ed - g.sd > zero ? ed - g.sd : zero;
  g.bd = rd / PD; g.bh = rh / PH; g.bw = rw / PW;
  g.nd = sample_num > 0 ? sample_num : (int)ceil(rd / PD);
  g.nh = sample_num > 0 ? sample_num : (int)ceil(rh / PH);
  g.nw = sample_num > 0 ? sample_num : (int)ceil(rw / PW);
  return g;
}

template <typename A>
struct AxisTapT { int lo, hi; A l, h; };

template <typename A>
__device__ __forceinline__ bool axis_tap_t(A v, int dim, AxisTapT<A> &t) {
  if (v < (A)-1 || v > (A)dim) return false;
  if (v <= 0) v = 0;
  const int lo = (int)v;
  if (lo >= dim - 1) { t.hi = t.lo = dim - 1; v = (A)t.lo; }
  else { t.lo = lo; t.hi = lo + 1; }
  t.l = v - (A)t.lo;
  t.h = (A)1 - t.l;
  return true;
}

template <typename T, typename A, bool kBackward>
__global__ __launch_bounds__(256) void roi_align3d_typed(long long total, const T *feat_or_top, const T *__restrict__ rois, A scale,
                                                        A scale_d, int sample_num, int C, int D, int H, int W, int PD, int PH, int PW,
                                                        Strides st, int bug_compat, T *out_or_bottom) {
  for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total; index += (long long)blockDim.x * gridDim.x) {
    const int pw = (int)(index % PW), ph = (int)((index / PW) % PH), pd = (int)((index / PW / PH) % PD);
    const int c = (int)((index / PW / PH / PD) % C), n = (int)(index / PW / PH / PD / C);
    const RoiGeomT<A> g = roi_geom_t<T, A>(rois + (size_t)n * 7, scale, scale_d, sample_num, PD, PH, PW);
    const long long base = g.b * st.n + c * st.c;
    const A count = (A)(g.nd * g.nh * g.nw);
    A acc = 0, tg = 0;
    if (kBackward) {
      const long long off = (long long)(n * C + c) * PD * PH * PW + (bug_compat ? pd * PD * PW : pd * PH * PW) + ph * PW + pw;
      tg = rd_as(feat_or_top + off, acc);
    }
    for (int iz = 0; iz < g.nd; ++iz) {
      AxisTapT<A> tz;
      const bool vz = axis_tap_t<A>(g.sd + pd * g.bd + (A)(iz + .5f) * g.bd / (A)g.nd, D, tz);
      for (int iy = 0; iy < g.nh; ++iy) {
        AxisTapT<A> ty;
        const bool vy = axis_tap_t<A>(g.sh + ph * g.bh + (A)(iy + .5f) * g.bh / (A)g.nh, H, ty);
        for (int ix = 0; ix < g.nw; ++ix) {
          AxisTapT<A> tx;
          const bool vx = axis_tap_t<A>(g.sw + pw * g.bw + (A)(ix + .5f) * g.bw / (A)g.nw, W, tx);
          if (!(vz && vy && vx)) continue;
          const long long zl = base + tz.lo * st.d, zh = base + tz.hi * st.d;
          const long long yl = ty.lo * st.h, yh = ty.hi * st.h, xl = tx.lo * st.w, xh = tx.hi * st.w;
          const A w1 = tx.h * ty.h * tz.h, w5 = tx.h * ty.h * tz.l, w2 = tx.l * ty.h * tz.h, w6 = tx.l * ty.h * tz.l;
          const A w3 = tx.h * ty.l * tz.h, w7 = tx.h * ty.l * tz.l, w4 = tx.l * ty.l * tz.h, w8 = tx.l * ty.l * tz.l;
          if (!kBackward) {
            const T *f = feat_or_top;
            acc += (w1 * rd_as(f + zl + yl + xl, acc) + w2 * rd_as(f + zl + yl + xh, acc) + w3 * rd_as(f + zl + yh + xl, acc) +
                    w4 * rd_as(f + zl + yh + xh, acc) + w5 * rd_as(f + zh + yl + xl, acc) + w6 * rd_as(f + zh + yl + xh, acc) +
                    w7 * rd_as(f + zh + yh + xl, acc) + w8 * rd_as(f + zh + yh + xh, acc));
          } else {
            T *bt = out_or_bottom;
            add_as(bt + zl + yl + xl, tg * w1 / count); add_as(bt + zl + yl + xh, tg * w2 / count);
            add_as(bt + zl + yh + xl, tg * w3 / count); add_as(bt + zl + yh + xh, tg * w4 / count);
            add_as(bt + zh + yl + xl, tg * w5 / count); add_as(bt + zh + yl + xh, tg * w6 / count);
            add_as(bt + zh + yh + xl, tg * w7 / count); add_as(bt + zh + yh + xh, tg * w8 / count);
          }
        }
      }
    }
    if (!kBackward) wr_as(out_or_bottom + index, acc / count);
  }
}

// ------------------------------------------------------------------------------------------
// multi-level form (SingleRoIExtractor, mmdet/models/roi_extractors/single_level.py:58-104): every RoI reads the
// pyramid level its size maps to.  The reference builds a boolean mask per level on the host (one sync per level)
// and launches RoIAlign once per level; here the level is computed in the kernel (same rule: scale = sqrt(w*h*d) --
// the square root of the VOLUME, :73-76 -- then floor(log2(scale / finest_scale + 1e-6)) clamped to the pyramid) and
// one launch serves all levels with no host round trip.
// ------------------------------------------------------------------------------------------
constexpr int kMaxLevels = 4;
struct LevelTable {
  int num;                 // 0: single-level call (the scalar arguments apply)
  float finest_scale;
  const float *feat[kMaxLevels];
  float *grad[kMaxLevels];
  int D[kMaxLevels], H[kMaxLevels], W[kMaxLevels];
  float scale[kMaxLevels], scale_d[kMaxLevels];
};

__device__ __forceinline__ int roi_level(const float *__restrict__ roi, float finest_scale, int num_levels) {
  const float sc = sqrtf((roi[3] - roi[1] + 1) * (roi[4] - roi[2] + 1) * (roi[6] - roi[5] + 1));
  const float l = floorf(log2f(sc / finest_scale + 1e-6f));
  int lvl = (int)fminf(fmaxf(l, 0.f), (float)(num_levels - 1));
  return lvl;
}

template <typename T>
__device__ __forceinline__ T pick4(int i, const T (&a)[kMaxLevels]) {
  return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

// ------------------------------------------------------------------------------------------
// channels-last fast path
// ------------------------------------------------------------------------------------------
constexpr int kClThreads = 256;          // 4 waves
constexpr int kClBins = 64;              // bins per workgroup (one LDS tile)
constexpr int kAxisCap = 256;            // table entries per axis (P * sample_num)

// table entry: lo/hi are ELEMENT offsets along the axis (index * stride), valid folded into weights:
// an out-of-range sample gets l = h = 0 and offsets 0, so it adds exactly 0 like the reference's `return 0`.
struct __attribute__((aligned(16))) TabEntry {
  int lo, hi;
  float l, h;
};

__device__ __forceinline__ void build_axis_table(TabEntry *tab, float start, float bin, int P, int ns, int dim,
                                                 int stride, int tid) {
  for (int e = tid; e < P * ns; e += kClThreads) {
    const int p = e / ns, i = e - p * ns;
    AxisTap t;
    TabEntry te;
    if (axis_tap(sample_coord(start, p, bin, i, ns), dim, t)) {
      te.lo = t.lo * stride; te.hi = t.hi * stride; te.l = t.l; te.h = t.h;
    } else {
      te.lo = 0; te.hi = 0; te.l = 0.f; te.h = 0.f;
    }
    tab[e] = te;
  }
}

template <bool kBackward>
__global__ __launch_bounds__(kClThreads) void roi_align3d_cl(const float *feat_or_top_src, float *out_or_bottom,
                                                            const float *__restrict__ rois, float scale, float scale_d,
                                                            int ns, int C, int D, int H, int W, int PD, int PH, int PW,
                                                            int bug_compat, LevelTable lt) {
  __shared__ TabEntry tabx[kAxisCap], taby[kAxisCap], tabz[kAxisCap];
  __shared__ float tile[kClBins][MR3D_WAVE + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = blockIdx.x;
  if (lt.num) {            // multi-level: this RoI's own feature map / gradient buffer and geometry
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    if (kBackward) out_or_bottom = pick4(lvl, lt.grad);
    else feat_or_top_src = pick4(lvl, lt.feat);
  }
  const int B = PD * PH * PW;
  const int bin0 = blockIdx.y * kClBins;
  const int nb = min(kClBins, B - bin0);
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, ns, PD, PH, PW);
  // element strides of the NDHWC tensor
  const int sW = C, sH = W * C;
  const long long sD = (long long)H * W * C;
  build_axis_table(tabx, g.sw, g.bw, PW, ns, W, sW, tid);
  build_axis_table(taby, g.sh, g.bh, PH, ns, H, sH, tid);
  build_axis_table(tabz, g.sd, g.bd, PD, ns, D, 1, tid);   // z offsets can exceed int: keep the index, scale below
  __syncthreads();
  if (kBackward) {
    // RoIs the separable sub-box kernel owns (decided once per call by roi_box_plan_kernel)
  }
  const float count = (float)(ns * ns * ns);
  const long long base_n = (long long)g.b * D * sD;

  for (int cg = 0; cg < C; cg += MR3D_WAVE) {
    const int c = cg + lane;
    const bool cok = c < C;
    if (kBackward) {
      // stage top_grad[r, cg..cg+63, bin0..bin0+nb) as tile[bin][channel]
      const float *top = feat_or_top_src;
      for (int cc = wave; cc < MR3D_WAVE; cc += kClThreads / MR3D_WAVE) {
        const int ch = cg + cc;
        if (ch < C && lane < nb) {
          const int bin = bin0 + lane;
          const int pw = bin % PW, ph = (bin / PW) % PH, pd = bin / (PW * PH);
          const int off = (bug_compat ? pd * PD * PW : pd * PH * PW) + ph * PW + pw;
          tile[lane][cc] = top[((size_t)r * C + ch) * B + off];
        }
      }
      __syncthreads();
    }
    for (int bi = wave; bi < nb; bi += kClThreads / MR3D_WAVE) {
      const int bin = bin0 + bi;
      const int pw = bin % PW, ph = (bin / PW) % PH, pd = bin / (PW * PH);
      float acc = 0.f;
      const float tg = kBackward ? tile[bi][lane] : 0.f;
      for (int iz = 0; iz < ns; ++iz) {
        const TabEntry tz = tabz[pd * ns + iz];
        const long long zl = base_n + tz.lo * sD, zh = base_n + tz.hi * sD;
        for (int iy = 0; iy < ns; ++iy) {
          const TabEntry ty = taby[ph * ns + iy];
          for (int ix = 0; ix < ns; ++ix) {
            const TabEntry tx = tabx[pw * ns + ix];
            // an out-of-range sample has l = h = 0 on that axis: it adds 0 / scatters nothing (:69-71, :600)
            if ((tx.l + tx.h) * (ty.l + ty.h) * (tz.l + tz.h) == 0.f) continue;
            const float w1 = tx.h * ty.h * tz.h, w5 = tx.h * ty.h * tz.l;
            const float w2 = tx.l * ty.h * tz.h, w6 = tx.l * ty.h * tz.l;
            const float w3 = tx.h * ty.l * tz.h, w7 = tx.h * ty.l * tz.l;
            const float w4 = tx.l * ty.l * tz.h, w8 = tx.l * ty.l * tz.l;
            const long long o1 = zl + ty.lo + tx.lo + c, o2 = zl + ty.lo + tx.hi + c;
            const long long o3 = zl + ty.hi + tx.lo + c, o4 = zl + ty.hi + tx.hi + c;
            const long long o5 = zh + ty.lo + tx.lo + c, o6 = zh + ty.lo + tx.hi + c;
            const long long o7 = zh + ty.hi + tx.lo + c, o8 = zh + ty.hi + tx.hi + c;
            if (!kBackward) {
              const float *f = feat_or_top_src;
              if (cok) {
                const float v1 = f[o1], v2 = f[o2], v3 = f[o3], v4 = f[o4];
                const float v5 = f[o5], v6 = f[o6], v7 = f[o7], v8 = f[o8];
                acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + w5 * v5 + w6 * v6 + w7 * v7 + w8 * v8);
              }
            } else {
              float *bt = out_or_bottom;
              if (cok) {
                atomicAdd(bt + o1, tg * w1 / count);
                atomicAdd(bt + o2, tg * w2 / count);
                atomicAdd(bt + o3, tg * w3 / count);
                atomicAdd(bt + o4, tg * w4 / count);
                atomicAdd(bt + o5, tg * w5 / count);
                atomicAdd(bt + o6, tg * w6 / count);
                atomicAdd(bt + o7, tg * w7 / count);
                atomicAdd(bt + o8, tg * w8 / count);
              }
            }
          }
        }
      }
      if (!kBackward) tile[bi][lane] = acc / count;
    }
    __syncthreads();
    if (!kBackward) {
      // tile[bin][channel] -> out[r, channel, bin]: 256-B runs of consecutive bins per channel
      for (int cc = wave; cc < MR3D_WAVE; cc += kClThreads / MR3D_WAVE) {
        const int ch = cg + cc;
        if (ch < C && lane < nb) out_or_bottom[((size_t)r * C + ch) * B + bin0 + lane] = tile[lane][cc];
      }
      __syncthreads();
    }
  }
}


struct IdxEntry { int lo, hi; float l, h; };     // plain voxel indices; l = h = 0 marks an out-of-range sample

__device__ __forceinline__ void build_idx_table(IdxEntry *tab, float start, float bin, int P, int ns, int dim, int tid,
                                                int nthreads) {
  for (int e = tid; e < P * ns; e += nthreads) {
    const int p = e / ns, i = e - p * ns;
    AxisTap t;
    IdxEntry te;
    if (axis_tap(sample_coord(start, p, bin, i, ns), dim, t)) { te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h; }
    else { te.lo = 0; te.hi = 0; te.l = 0.f; te.h = 0.f; }
    tab[e] = te;
  }
}

// ------------------------------------------------------------------------------------------
// channels-last FORWARD, footprint staged in LDS (the generic bf16-feature path; fp32 features never take it: measured slower than the wave-per-bin kernel, 141 vs 87 us on the bbox case).  The
// wave-per-bin kernel above re-reads every trilinear tap from L2 -- 64 x 256 B per output bin, ~22x the bytes the op has
// to move (the distinct voxels under a RoI's sample footprint once + the output once).  Here a workgroup owns
// (RoI, 16 channels): it finds the voxel box the RoI's
// samples touch (from the per-axis tap tables), stages that box ONCE as [z][y][x][16 ch] fp32 in LDS (64-byte rows
// from HBM; the other channel groups of the RoI read the rest of the same lines out of L2), evaluates every bin from LDS
// (a wave = 4 bins x 16 channels), and writes [R,C,PD,PH,PW] rows in 256-B runs through an LDS transpose.  Same
// arithmetic expression per sample as above.  Features may be fp32 or bf16 (configs[4]; the reference's kernel
// dispatches float / double / half, roi_align_kernel.cu:323); output fp32.  A RoI whose box exceeds the LDS budget
// (never with the shipped level rule) reads its taps from global memory instead.
// ------------------------------------------------------------------------------------------
constexpr int kStCh = 16;            // channels per workgroup
constexpr int kStCap = 1024;         // voxels of the staged box (64 KB of fp32 at 16 channels)
constexpr int kStAxis = 64;          // tap-table entries per axis (P * sample_num)
constexpr int kStLdsBytes = 3 * kStAxis * 16 + 64 + kStCap * kStCh * 4 + 64 * (kStCh + 1) * 4;

__device__ __forceinline__ float4 ld_feat4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld_feat4(const uint16_t *p) {
  const uint2 q = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                     __uint_as_float(q.y & 0xffff0000u));
}
__device__ __forceinline__ float ld_feat1(const float *p) { return *p; }
__device__ __forceinline__ float ld_feat1(const uint16_t *p) { return __uint_as_float((uint32_t)*p << 16); }

template <typename T>
__global__ __launch_bounds__(kClThreads) void roi_align3d_fwd_staged(const T *feat, float *__restrict__ out,
                                                                    const float *__restrict__ rois, float scale, float scale_d,
                                                                    int ns, int C, int D, int H, int W, int PD, int PH, int PW,
                                                                    LevelTable lt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char st_lds[];
  IdxEntry *tabx = reinterpret_cast<IdxEntry *>(st_lds), *taby = tabx + kStAxis, *tabz = taby + kStAxis;
  int *ext = reinterpret_cast<int *>(tabz + kStAxis);                 // x0,nx, y0,ny, z0,nz
  float *box = reinterpret_cast<float *>(st_lds + 3 * kStAxis * 16 + 64);
  float (*tile)[kStCh + 1] = reinterpret_cast<float (*)[kStCh + 1]>(box + kStCap * kStCh);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = blockIdx.x, c0 = blockIdx.y * kStCh;
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    feat = reinterpret_cast<const T *>(pick4(lvl, lt.feat));
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, ns, PD, PH, PW);
  build_idx_table(tabx, g.sw, g.bw, PW, ns, W, tid, kClThreads);
  build_idx_table(taby, g.sh, g.bh, PH, ns, H, tid, kClThreads);
  build_idx_table(tabz, g.sd, g.bd, PD, ns, D, tid, kClThreads);
  __syncthreads();
  if (tid < 3) {            // voxel range of the valid samples of one axis
    const IdxEntry *tab = tid == 0 ? tabx : (tid == 1 ? taby : tabz);
    const int n = (tid == 0 ? PW : (tid == 1 ? PH : PD)) * ns;
    int mn = 0x7fffffff, mx = -1;
    for (int e = 0; e < n; ++e)
      if (tab[e].l + tab[e].h != 0.f) { mn = min(mn, tab[e].lo); mx = max(mx, tab[e].hi); }
    ext[2 * tid] = mx < 0 ? 0 : mn;
    ext[2 * tid + 1] = mx < 0 ? 0 : mx - mn + 1;
  }
  __syncthreads();
  const int x0 = ext[0], nx = ext[1], y0 = ext[2], ny = ext[3], z0 = ext[4], nz = ext[5];
  const long long vox = (long long)nx * ny * nz;
  const bool staged = vox > 0 && vox <= kStCap;
  const size_t base_n = (size_t)g.b * D * H * W * C + c0;
  __syncthreads();                                                   // everyone has read ext / the absolute tables
  // tables -> element offsets: into the staged box, or into the feature map
  {
    const long long mx_ = staged ? kStCh : C, my_ = staged ? (long long)nx * kStCh : (long long)W * C;
    for (int e = tid; e < PW * ns; e += kClThreads) { tabx[e].lo = (int)((tabx[e].lo - (staged ? x0 : 0)) * mx_); tabx[e].hi = (int)((tabx[e].hi - (staged ? x0 : 0)) * mx_); }
    for (int e = tid; e < PH * ns; e += kClThreads) { taby[e].lo = (int)((taby[e].lo - (staged ? y0 : 0)) * my_); taby[e].hi = (int)((taby[e].hi - (staged ? y0 : 0)) * my_); }
    if (staged)
      for (int e = tid; e < PD * ns; e += kClThreads) { tabz[e].lo = (tabz[e].lo - z0) * ny * nx * kStCh; tabz[e].hi = (tabz[e].hi - z0) * ny * nx * kStCh; }
  }
  if (staged) {
    const int pieces = (int)vox * 4;
    for (int it = tid; it < pieces; it += kClThreads) {
      const int q = it & 3, v = it >> 2;
      const int x = v % nx, y = (v / nx) % ny, z = v / (nx * ny);
      const float4 f = ld_feat4(feat + base_n + (((size_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x)) * C + q * 4);
      *reinterpret_cast<float4 *>(box + v * kStCh + q * 4) = f;
    }
  }
  __syncthreads();

  const int B = PD * PH * PW;
  const float count = (float)(ns * ns * ns);
  const int bs = lane >> 4, ch = lane & 15;
  const long long sD = (long long)H * W * C;
  for (int pass0 = 0; pass0 < B; pass0 += 64) {
    const int nb = min(64, B - pass0);
    for (int bg = wave; bg * 4 < nb; bg += kClThreads / MR3D_WAVE) {
      const int bl = bg * 4 + bs;
      if (bl < nb) {
        const int bin = pass0 + bl;
        const int pw = bin % PW, ph = (bin / PW) % PH, pd = bin / (PW * PH);
        float acc = 0.f;
        for (int iz = 0; iz < ns; ++iz) {
          const IdxEntry tz = tabz[pd * ns + iz];
          for (int iy = 0; iy < ns; ++iy) {
            const IdxEntry ty = taby[ph * ns + iy];
            for (int ix = 0; ix < ns; ++ix) {
              const IdxEntry tx = tabx[pw * ns + ix];
              if ((tx.l + tx.h) * (ty.l + ty.h) * (tz.l + tz.h) == 0.f) continue;
              const float w1 = tx.h * ty.h * tz.h, w5 = tx.h * ty.h * tz.l;
              const float w2 = tx.l * ty.h * tz.h, w6 = tx.l * ty.h * tz.l;
              const float w3 = tx.h * ty.l * tz.h, w7 = tx.h * ty.l * tz.l;
              const float w4 = tx.l * ty.l * tz.h, w8 = tx.l * ty.l * tz.l;
              float v1, v2, v3, v4, v5, v6, v7, v8;
              if (staged) {
                const float *bl_ = box + tz.lo + ch, *bh_ = box + tz.hi + ch;
                v1 = bl_[ty.lo + tx.lo]; v2 = bl_[ty.lo + tx.hi]; v3 = bl_[ty.hi + tx.lo]; v4 = bl_[ty.hi + tx.hi];
                v5 = bh_[ty.lo + tx.lo]; v6 = bh_[ty.lo + tx.hi]; v7 = bh_[ty.hi + tx.lo]; v8 = bh_[ty.hi + tx.hi];
              } else {
                const T *fl = feat + base_n + tz.lo * sD + ch, *fh = feat + base_n + tz.hi * sD + ch;
                v1 = ld_feat1(fl + ty.lo + tx.lo); v2 = ld_feat1(fl + ty.lo + tx.hi); v3 = ld_feat1(fl + ty.hi + tx.lo); v4 = ld_feat1(fl + ty.hi + tx.hi);
                v5 = ld_feat1(fh + ty.lo + tx.lo); v6 = ld_feat1(fh + ty.lo + tx.hi); v7 = ld_feat1(fh + ty.hi + tx.lo); v8 = ld_feat1(fh + ty.hi + tx.hi);
              }
              acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + w5 * v5 + w6 * v6 + w7 * v7 + w8 * v8);
            }
          }
        }
        tile[bl][ch] = acc / count;
      }
    }
    __syncthreads();
    // tile[bin][channel] -> out[r, c0 + channel, pass0 + bin]: 256-B runs of consecutive bins per channel
    for (int cc = wave; cc < kStCh; cc += kClThreads / MR3D_WAVE)
      if (lane < nb) out[((size_t)r * C + c0 + cc) * B + pass0 + lane] = tile[lane][cc];
    __syncthreads();
  }
}

template <typename T>
int launch_fwd_staged(const T *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int ns, int C,
                      int D, int H, int W, int PD, int PH, int PW, const LevelTable &lt, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(roi_align3d_fwd_staged<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kStLdsBytes) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  roi_align3d_fwd_staged<T><<<dim3(num_rois, C / kStCh), kClThreads, kStLdsBytes, stream>>>(feat, out, rois, scale, scale_d, ns, C, D,
                                                                                         H, W, PD, PH, PW, lt);
  return mr3d_launch_status();
}

// ------------------------------------------------------------------------------------------
// channels-last FORWARD, separable.  The pooled value of bin (pd, ph, pw) is a sum over ns^3 samples of a trilinear tap sum:
// 64 loads per output element in the reference (and in the wave-per-bin kernel above).  But weights factor per axis
// (w = wz * wy * wx, a sample is dropped iff one of its axes is out of range), so per channel
//     out[pd][ph][pw] = 1/count * sum_z Wz[pd][z] * sum_y Wy[ph][y] * sum_x Wx[pw][x] * F[z][y][x]
// with Wx[pw][x] = the summed weights of bin pw's valid x-samples on voxel x (<= 2*ns non-zeros), Wy, Wz alike: three 1-D
// contractions.  A workgroup owns (RoI, 16 channels, a chunk of PDC pooled depths) and walks the z-planes that chunk touches:
//   stage 1  T1[y][pw] = sum_x Wx[pw][x] F[z][y][x]        one item per (row y, bin pw): <= 2*ns 64-byte global reads (16 ch)
//   stage 2  t2 = sum_y Wy[ph][y] T1[y][pw]                one item per (ph, pw): <= 2*ns LDS reads
//   stage 3  acc[pd][(ph,pw)] += Wz[pd][z] * t2            in REGISTERS (the thread that made t2 owns that bin column)
// so a (3,7,7) bin costs ~3+3+3 taps instead of 64, a (10,14,14) bin even fewer (its planes are shared by more bins).  The
// result goes through an LDS tile [bin][channel] so that [R,C,PD,PH,PW] rows are written in 256-byte runs.
// Same tap rule as bilinear_interpolate_3d (axis_tap); the fp32 sums are associated differently from the reference's
// per-sample order (~1e-6 relative, inside the 1e-4 parity bound).  RoIs whose footprint rows exceed the T1 tile fall back to
// the direct per-bin evaluation inside the same kernel.
// ------------------------------------------------------------------------------------------
constexpr int kSepCh = 16;                 // channels per workgroup
template <int PDC_, int J_, int T1CAP_>
struct SepCfg {
  static constexpr int PDC = PDC_, J = J_;                  // pooled depths per workgroup; (ph,pw) items per thread (16 workers)
  static constexpr int T1CAP = T1CAP_;                      // T1 items (rows y x bins pw) per z-plane
  static constexpr int T1_FLOATS = 2 * T1CAP * kSepCh;      // double-buffered
  static constexpr int TILE_FLOATS = PDC * J * 16 * (kSepCh + 1);
  static constexpr int BODY_FLOATS = T1_FLOATS > TILE_FLOATS ? T1_FLOATS : TILE_FLOATS;
  static constexpr int LDS_BYTES = 3 * kStAxis * 16 + 64 + BODY_FLOATS * 4;
};

template <typename T, class S>
__global__ __launch_bounds__(kClThreads, 2) void roi_align3d_fwd_sep(const T *feat, float *__restrict__ out, const float *__restrict__ rois,
                                                                 float scale, float scale_d, int ns, int C, int D, int H, int W,
                                                                 int PD, int PH, int PW, LevelTable lt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
  IdxEntry *tabx = reinterpret_cast<IdxEntry *>(sp_lds), *taby = tabx + kStAxis, *tabz = taby + kStAxis;
  int *ext = reinterpret_cast<int *>(tabz + kStAxis);                 // x0,nx, y0,ny
  float *body = reinterpret_cast<float *>(sp_lds + 3 * kStAxis * 16 + 64);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = tid & 15, worker = tid >> 4;                         // 16 channels x 16 workers
  const int r = blockIdx.x, c0 = blockIdx.y * kSepCh, pd0 = blockIdx.z * S::PDC;
  const int npd = min(S::PDC, PD - pd0);
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    feat = reinterpret_cast<const T *>(pick4(lvl, lt.feat));
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, ns, PD, PH, PW);
  build_idx_table(tabx, g.sw, g.bw, PW, ns, W, tid, kClThreads);
  build_idx_table(taby, g.sh, g.bh, PH, ns, H, tid, kClThreads);
  build_idx_table(tabz, g.sd, g.bd, PD, ns, D, tid, kClThreads);
  __syncthreads();
  if (tid < 2) {            // voxel range of the valid samples along x (tid 0) and y (tid 1)
    const IdxEntry *tab = tid == 0 ? tabx : taby;
    const int n = (tid == 0 ? PW : PH) * ns;
    int mn = 0x7fffffff, mx = -1;
    for (int e = 0; e < n; ++e)
      if (tab[e].l + tab[e].h != 0.f) { mn = min(mn, tab[e].lo); mx = max(mx, tab[e].hi); }
    ext[2 * tid] = mx < 0 ? 0 : mn;
    ext[2 * tid + 1] = mx < 0 ? 0 : mx - mn + 1;
  }
  __syncthreads();
  const int y0 = ext[2], ny = ext[3];
  // z range of this chunk's pooled depths (wave-uniform scalars)
  int zmin = 0x7fffffff, zmax = -1;
  for (int e = pd0 * ns; e < (pd0 + npd) * ns; ++e)
    if (tabz[e].l + tabz[e].h != 0.f) { zmin = min(zmin, tabz[e].lo); zmax = max(zmax, tabz[e].hi); }
  const int B = PD * PH * PW, PHW = PH * PW;
  const float inv_count = 1.f / (float)(ns * ns * ns);
  const size_t base_n = (size_t)g.b * D * H * W * C + c0 + ch;
  const long long sH = (long long)W * C, sD = (long long)H * W * C;

  float acc[S::PDC][S::J];
#pragma unroll
  for (int a = 0; a < S::PDC; ++a)
#pragma unroll
    for (int j = 0; j < S::J; ++j) acc[a][j] = 0.f;

  const bool separable = ns == 2 && ny * PW <= S::T1CAP && PHW <= 16 * S::J;
  if (separable) {
    // everything about a thread's items that does not depend on the plane is worked out once: the kernel is issue-bound, and an
    // integer division by the (runtime) PW per item and plane was a large share of its instructions
    const int q4 = tid & 3, w64 = tid >> 2;
    constexpr int U = sizeof(T) == 4 ? 2 : 4;                    // stage-1 items per thread and batch (fp32: 2, for two waves per SIMD)
    const int yy_w = w64 / PW, pw_w = w64 - yy_w * PW, q64 = 64 / PW, r64 = 64 - q64 * PW;
    // T1 element offsets of the two rows of item j's two y-samples, (hi << 16) | lo: every register counts here -- 281 of them
    // meant one wave per SIMD, and the kernel lives on hiding its plane-by-plane round trips behind other waves
    unsigned t1[S::J][2];
    static_assert(S::T1CAP * kSepCh <= 0xffff, "T1 offsets are packed in 16 bits");
    float wyh[S::J][2], wyl[S::J][2];
#pragma unroll
    for (int j = 0; j < S::J; ++j) {
      const int item = worker + 16 * j;
      const int ph = item < PHW ? item / PW : 0, pw = item < PHW ? item - ph * PW : 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const IdxEntry ty = taby[ph * 2 + i];
        const bool on = item < PHW && ty.l + ty.h != 0.f;
        t1[j][i] = on ? (unsigned)(((ty.lo - y0) * PW + pw) * kSepCh + ch) | ((unsigned)(((ty.hi - y0) * PW + pw) * kSepCh + ch) << 16) : 0u;
        wyh[j][i] = on ? ty.h : 0.f;
        wyl[j][i] = on ? ty.l : 0.f;
      }
    }
    int buf = 0;
    for (int z = zmin; z <= zmax; ++z, buf ^= 1) {
      float *T1 = body + buf * (S::T1CAP * kSepCh);
      // ---- stage 1: x-contraction of plane z, rows y0 .. y0+ny-1.  A thread = (channel quad, one of 64 workers): its four
      // taps are 16-byte loads (4 lanes cover the 64 bytes of a position's 16 channels; one channel per lane was 4x the
      // load instructions and 7 dependent L2 round trips per plane for a 28-row footprint, now 2).  U items per thread and
      // batch, all 16 tap loads issued before the first is used.
      {
        const T *fq = feat + (base_n - ch) + q4 * 4;
        int yy_u = yy_w, pw_u = pw_w;                            // (row, bin) of item w64; + 64 per step without a division
        for (int it0 = 0; it0 < ny * PW; it0 += 64 * U) {
          float4 v[U][4];
          float w[U][4];
          int yy4[U], pw4[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int item = it0 + w64 + 64 * u;
            const bool ok = item < ny * PW;
            yy4[u] = ok ? yy_u : 0; pw4[u] = ok ? pw_u : 0;
            pw_u += r64; yy_u += q64;
            if (pw_u >= PW) { pw_u -= PW; ++yy_u; }
            const T *row = fq + z * sD + (y0 + yy4[u]) * sH;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const IdxEntry tx = tabx[pw4[u] * 2 + i];               // an out-of-range sample has l = h = 0 and lo = hi = 0
              w[u][2 * i] = tx.h; w[u][2 * i + 1] = tx.l;
              v[u][2 * i] = ld_feat4(row + (long long)tx.lo * C);
              v[u][2 * i + 1] = ld_feat4(row + (long long)tx.hi * C);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int item = it0 + w64 + 64 * u;
            if (item < ny * PW) {
              float4 o;
              o.x = (w[u][0] * v[u][0].x + w[u][1] * v[u][1].x) + (w[u][2] * v[u][2].x + w[u][3] * v[u][3].x);
              o.y = (w[u][0] * v[u][0].y + w[u][1] * v[u][1].y) + (w[u][2] * v[u][2].y + w[u][3] * v[u][3].y);
              o.z = (w[u][0] * v[u][0].z + w[u][1] * v[u][1].z) + (w[u][2] * v[u][2].z + w[u][3] * v[u][3].z);
              o.w = (w[u][0] * v[u][0].w + w[u][1] * v[u][1].w) + (w[u][2] * v[u][2].w + w[u][3] * v[u][3].w);
              *reinterpret_cast<float4 *>(T1 + item * kSepCh + q4 * 4) = o;
            }
          }
        }
      }
      __syncthreads();
      // ---- stages 2 + 3: y-contraction, then the z-weights of this plane into the register accumulators
      float wz[S::PDC];
#pragma unroll
      for (int a = 0; a < S::PDC; ++a) {
        float w = 0.f;
        if (a < npd)
          for (int i = 0; i < ns; ++i) {
            const IdxEntry tz = tabz[(pd0 + a) * ns + i];
            w += (tz.lo == z ? tz.h : 0.f) + (tz.hi == z ? tz.l : 0.f);       // l = h = 0 for an out-of-range sample
          }
        wz[a] = w;
      }
#pragma unroll
      for (int j = 0; j < S::J; ++j) {
        // (a sample that is out of range, or an item beyond PHW, carries zero weights and reads T1 element 0)
        const float t2 = (wyh[j][0] * T1[t1[j][0] & 0xffffu] + wyl[j][0] * T1[t1[j][0] >> 16]) +
                         (wyh[j][1] * T1[t1[j][1] & 0xffffu] + wyl[j][1] * T1[t1[j][1] >> 16]);
#pragma unroll
        for (int a = 0; a < S::PDC; ++a) acc[a][j] += wz[a] * t2;
      }
      // the other T1 buffer is written next: one barrier per plane is enough (its readers passed the barrier above)
    }
  } else {
    // ---- fallback: direct per-bin evaluation (footprint rows do not fit the T1 tile)
#pragma unroll
    for (int a = 0; a < S::PDC; ++a)
#pragma unroll
      for (int j = 0; j < S::J; ++j) {
        const int item = worker + 16 * j;
        if (a < npd && item < PHW) {
          const int ph = item / PW, pw = item - ph * PW;
          float v = 0.f;
          for (int iz = 0; iz < ns; ++iz) {
            const IdxEntry tz = tabz[(pd0 + a) * ns + iz];
            for (int iy = 0; iy < ns; ++iy) {
              const IdxEntry ty = taby[ph * ns + iy];
              for (int ix = 0; ix < ns; ++ix) {
                const IdxEntry tx = tabx[pw * ns + ix];
                if ((tx.l + tx.h) * (ty.l + ty.h) * (tz.l + tz.h) == 0.f) continue;
                const T *f = feat + base_n;
                auto at = [&](int zz, int yy, int xx) { return ld_feat1(f + zz * sD + yy * sH + (long long)xx * C); };
                v += tz.h * (ty.h * (tx.h * at(tz.lo, ty.lo, tx.lo) + tx.l * at(tz.lo, ty.lo, tx.hi)) +
                             ty.l * (tx.h * at(tz.lo, ty.hi, tx.lo) + tx.l * at(tz.lo, ty.hi, tx.hi))) +
                     tz.l * (ty.h * (tx.h * at(tz.hi, ty.lo, tx.lo) + tx.l * at(tz.hi, ty.lo, tx.hi)) +
                             ty.l * (tx.h * at(tz.hi, ty.hi, tx.lo) + tx.l * at(tz.hi, ty.hi, tx.hi)));
              }
            }
          }
          acc[a][j] = v;
        }
      }
  }
  // ---- accumulators -> LDS tile [local bin][channel] -> out[r, c0 + channel, pd0 .. , :, :] in 256-byte runs
  __syncthreads();
  float (*tile)[kSepCh + 1] = reinterpret_cast<float (*)[kSepCh + 1]>(body);
#pragma unroll
  for (int a = 0; a < S::PDC; ++a)
#pragma unroll
    for (int j = 0; j < S::J; ++j) {
      const int item = worker + 16 * j;
      if (a < npd && item < PHW) tile[a * PHW + item][ch] = acc[a][j] * inv_count;
    }
  __syncthreads();
  const int nbins = npd * PHW;
  for (int cc = wave; cc < kSepCh; cc += kClThreads / MR3D_WAVE) {
    float *dst = out + ((size_t)r * C + c0 + cc) * B + (size_t)pd0 * PHW;
    for (int bI = lane; bI < nbins; bI += MR3D_WAVE) dst[bI] = tile[bI][cc];
  }
}

template <typename T, class S>
int launch_fwd_sep(const T *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int ns, int C, int D,
                   int H, int W, int PD, int PH, int PW, const LevelTable &lt, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done && S::LDS_BYTES > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(roi_align3d_fwd_sep<T, S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            S::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  dim3 grid(num_rois, C / kSepCh, (PD + S::PDC - 1) / S::PDC);
  roi_align3d_fwd_sep<T, S><<<grid, kClThreads, S::LDS_BYTES, stream>>>(feat, out, rois, scale, scale_d, ns, C, D, H, W, PD, PH, PW, lt);
  return mr3d_launch_status();
}

// picks the separable instance for a pooled shape: MR3D_ERR_UNSUPPORTED = none (the caller takes the wave-per-bin kernel)
template <typename T>
int try_fwd_sep(const T *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int ns, int C, int D, int H,
                int W, int PD, int PH, int PW, const LevelTable &lt, hipStream_t stream) {
  if (C % kSepCh != 0 || ns != 2 || PD * ns > kStAxis || PH * ns > kStAxis || PW * ns > kStAxis || PW > 14) return MR3D_ERR_UNSUPPORTED;
  const int phw = PH * PW;
  // Measured on the SURVEY 8(d) bench inputs: the separable form wins where many bins share the planes of a RoI -- the mask
  // extractor (10,14,14): 137 vs 232 us -- and only ties on the bbox extractor (3,7,7): 88 vs 88 us (3 pooled depths over up
  // to 20 planes: as many stage-1 taps as the direct form has in total; a variant that walks only the touched planes / rows in
  // groups was slower still, 110 us).  bf16 features take it for both shapes (there is no bf16 wave-per-bin kernel).
  if (phw <= 64 && PD <= 4) {     // bbox extractor (3,7,7): every pooled depth in one workgroup
    if (sizeof(T) == 4) return MR3D_ERR_UNSUPPORTED;          // fp32 features: the box-in-LDS kernel (try_fwd_box) is faster there
    return launch_fwd_sep<T, SepCfg<4, 4, 224>>(feat, out, rois, num_rois, scale, scale_d, ns, C, D, H, W, PD, PH, PW, lt, stream);
  }
  if (phw <= 208)                 // mask extractor (10,14,14): chunks of 5 pooled depths
    return launch_fwd_sep<T, SepCfg<5, 13, 448>>(feat, out, rois, num_rois, scale, scale_d, ns, C, D, H, W, PD, PH, PW, lt, stream);
  return MR3D_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// channels-last FORWARD, whole-voxel box in LDS.  A workgroup owns (RoI, CH = 64 or 32 channels, a chunk of pooled depths).  Its
// bins are walked in SEGMENTS of whole pooled depths whose sample footprint -- a dense voxel box [z][y][x] -- fits the LDS
// budget (as many consecutive depths as fit).  A segment's box is staged
// ONCE as fp32 -- whole x-rows of the NDHWC map, up to 16 row pieces per wave in flight -- and every bin of the segment is
// evaluated from LDS by CH/4 lanes x float4 channels (at CH = 64 a voxel is all 64 LDS banks, so the 16-byte reads never
// conflict; the per-sample weights are computed once per 4 channels; packed fp32 FMAs).  A pooled depth whose box alone does
// not fit reads its taps from global memory.  Same per-sample expression as the wave-per-bin kernel.  Results
// leave through an LDS tile [channel][tbins] so that [R,C,PD,PH,PW] rows are written in runs of tbins consecutive bins.
// Measurements: try_fwd_box below.
// ------------------------------------------------------------------------------------------
constexpr int kBxTabBytes = 5 * kStAxis * 16 + 64;      // tap tables (x, y in two forms; z) + extents
constexpr int kBxUnits = 8;                // row pieces (64 float4) a wave keeps in flight while staging

struct BoxRange { int lo, n; };

// voxels touched by the valid samples of pooled indices [a, b) of one axis (table with plain voxel indices)
__device__ __forceinline__ BoxRange tap_range(const IdxEntry *tab, int a, int b, int ns) {
  int mn = 0x7fffffff, mx = -1;
  for (int e = a * ns; e < b * ns; ++e)
    if (tab[e].l + tab[e].h != 0.f) { mn = min(mn, tab[e].lo); mx = max(mx, tab[e].hi); }
  BoxRange r;
  r.lo = mx < 0 ? 0 : mn;
  r.n = mx < 0 ? 0 : mx - mn + 1;
  return r;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// one bin (pd, ph, pw) of 4 channels.  STAGED: x / y tables hold float4 offsets into the box, `box` points at this lane's
// channel quad of voxel (z0, segment's first row, x0), plane4 = float4 per box plane; else they hold element offsets into the
// feature map and fbase points at this lane's channels of the RoI's image.  NS > 0 unrolls the sample loops; GUARD keeps the
// reference's "an out-of-range sample adds nothing" as a branch (without it every sample is known valid).
template <typename T, int NS, bool STAGED, bool GUARD>
__device__ __forceinline__ float4 box_bin(const IdxEntry *tabx, const IdxEntry *taby, const IdxEntry *tabz, int pd, int ph, int pw,
                                          int ns_rt, const float4 *box, int z0, int plane4, const T *fbase, long long sD) {
  const int ns = NS > 0 ? NS : ns_rt;
  v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll NS > 0 ? NS : 1
  for (int iz = 0; iz < ns; ++iz) {
    const IdxEntry tz = tabz[pd * ns + iz];
#pragma unroll NS > 0 ? NS : 1
    for (int iy = 0; iy < ns; ++iy) {
      const IdxEntry ty = taby[ph * ns + iy];
#pragma unroll NS > 0 ? NS : 1
      for (int ix = 0; ix < ns; ++ix) {
        const IdxEntry tx = tabx[pw * ns + ix];
        if (GUARD && (tx.l + tx.h) * (ty.l + ty.h) * (tz.l + tz.h) == 0.f) continue;
        const float w1 = tx.h * ty.h * tz.h, w5 = tx.h * ty.h * tz.l;
        const float w2 = tx.l * ty.h * tz.h, w6 = tx.l * ty.h * tz.l;
        const float w3 = tx.h * ty.l * tz.h, w7 = tx.h * ty.l * tz.l;
        const float w4 = tx.l * ty.l * tz.h, w8 = tx.l * ty.l * tz.l;
        float4 v1, v2, v3, v4, v5, v6, v7, v8;
        if (STAGED) {
          const float4 *bl_ = box + (tz.lo - z0) * plane4, *bh_ = box + (tz.hi - z0) * plane4;
          v1 = bl_[ty.lo + tx.lo]; v2 = bl_[ty.lo + tx.hi]; v3 = bl_[ty.hi + tx.lo]; v4 = bl_[ty.hi + tx.hi];
          v5 = bh_[ty.lo + tx.lo]; v6 = bh_[ty.lo + tx.hi]; v7 = bh_[ty.hi + tx.lo]; v8 = bh_[ty.hi + tx.hi];
        } else {
          const T *fl = fbase + tz.lo * sD, *fh = fbase + tz.hi * sD;
          v1 = ld_feat4(fl + ty.lo + tx.lo); v2 = ld_feat4(fl + ty.lo + tx.hi); v3 = ld_feat4(fl + ty.hi + tx.lo); v4 = ld_feat4(fl + ty.hi + tx.hi);
          v5 = ld_feat4(fh + ty.lo + tx.lo); v6 = ld_feat4(fh + ty.lo + tx.hi); v7 = ld_feat4(fh + ty.hi + tx.lo); v8 = ld_feat4(fh + ty.hi + tx.hi);
        }
#define MR3D_LO(v) ((v2f){(v).x, (v).y})
#define MR3D_HI(v) ((v2f){(v).z, (v).w})
        v2f s0 = w1 * MR3D_LO(v1), s1 = w1 * MR3D_HI(v1);
        s0 += w2 * MR3D_LO(v2); s1 += w2 * MR3D_HI(v2);
        s0 += w3 * MR3D_LO(v3); s1 += w3 * MR3D_HI(v3);
        s0 += w4 * MR3D_LO(v4); s1 += w4 * MR3D_HI(v4);
        s0 += w5 * MR3D_LO(v5); s1 += w5 * MR3D_HI(v5);
        s0 += w6 * MR3D_LO(v6); s1 += w6 * MR3D_HI(v6);
        s0 += w7 * MR3D_LO(v7); s1 += w7 * MR3D_HI(v7);
        s0 += w8 * MR3D_LO(v8); s1 += w8 * MR3D_HI(v8);
#undef MR3D_LO
#undef MR3D_HI
        a0 += s0; a1 += s1;
      }
    }
  }
  return make_float4(a0.x, a0.y, a1.x, a1.y);
}

template <typename T, int NS, int CH>
__global__ __launch_bounds__(kClThreads) void roi_align3d_fwd_box(const T *feat, float *__restrict__ out,
                                                                 const float *__restrict__ rois, float scale, float scale_d,
                                                                 int ns, int C, int D, int H, int W, int PD, int PH, int PW,
                                                                 int pdc, int cap, int tbins, LevelTable lt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bx_lds[];
  // tabx / taby: voxel indices at first, then float4 offsets into the box; tabxg / tabyg: element offsets into the map
  IdxEntry *tabx = reinterpret_cast<IdxEntry *>(bx_lds), *taby = tabx + kStAxis, *tabz = taby + kStAxis;
  IdxEntry *tabxg = tabz + kStAxis, *tabyg = tabxg + kStAxis;
  int *ext = reinterpret_cast<int *>(tabyg + kStAxis);                // x0,nx, y0,ny, -, -, invalid samples on x, y, z
  float4 *box = reinterpret_cast<float4 *>(bx_lds + kBxTabBytes);
  float *tile = reinterpret_cast<float *>(bx_lds + kBxTabBytes + (size_t)cap * CH * 4);   // [CH][tbins + 1] results
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int L = CH / 4, SLOTS = kClThreads / L;                   // lanes per voxel, bin slots
  const int TS = tbins + 1;
  const int q = tid % L, slot = tid / L;                              // float4 of channels, bin slot
  const int r = blockIdx.x, c0 = blockIdx.y * CH, pd0 = blockIdx.z * pdc;
  const int pd1 = min(pd0 + pdc, PD);
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    feat = reinterpret_cast<const T *>(pick4(lvl, lt.feat));
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, ns, PD, PH, PW);
  build_idx_table(tabx, g.sw, g.bw, PW, ns, W, tid, kClThreads);
  build_idx_table(taby, g.sh, g.bh, PH, ns, H, tid, kClThreads);
  build_idx_table(tabz, g.sd, g.bd, PD, ns, D, tid, kClThreads);
  __syncthreads();
  if (tid < 3) {            // voxel range of the valid samples along x (tid 0), y (tid 1); are all samples of the axis valid?
    const IdxEntry *tab = tid == 0 ? tabx : (tid == 1 ? taby : tabz);
    const int n = (tid == 0 ? PW : (tid == 1 ? PH : PD)) * ns;
    int mn = 0x7fffffff, mx = -1, bad = 0;
    for (int e = 0; e < n; ++e) {
      if (tab[e].l + tab[e].h != 0.f) { mn = min(mn, tab[e].lo); mx = max(mx, tab[e].hi); }
      else bad = 1;
    }
    ext[2 * tid] = mx < 0 ? 0 : mn;
    ext[2 * tid + 1] = mx < 0 ? 0 : mx - mn + 1;
    ext[6 + tid] = bad;
  }
  // element-offset copies of the x / y tables for rows that read from global memory
  for (int e = tid; e < PW * ns; e += kClThreads) { IdxEntry t = tabx[e]; t.lo *= C; t.hi *= C; tabxg[e] = t; }
  for (int e = tid; e < PH * ns; e += kClThreads) { IdxEntry t = taby[e]; t.lo *= W * C; t.hi *= W * C; tabyg[e] = t; }
  __syncthreads();
  const int x0 = ext[0], nx = ext[1], y0 = ext[2], ny = ext[3];
  const bool guard = (ext[6] | ext[7] | ext[8]) != 0;
  const int B = PD * PH * PW, PHW = PH * PW;
  const float inv_count = 1.f / (float)(ns * ns * ns);
  const size_t base_n = (size_t)g.b * D * H * W * C + c0;
  const long long sD = (long long)H * W * C;
  const int rowlen4 = nx * L, pieces = (rowlen4 + 63) >> 6;
  const T *fq = feat + base_n + q * 4;

  int flushed = pd0 * PHW;                                           // first bin not yet written to out
  bool rel = false;                                                  // tabx / taby already converted to box offsets?
  for (int row = pd0 * PH; row < pd1 * PH;) {                        // rows = (pd, ph) pairs in bin order
    const int pd = row / PH, ph = row - pd * PH;
    // ---- next segment [row, row_end): whole pooled depths from a depth boundary, else ph rows of this depth (wave-uniform)
    int row_end = row + 1;
    BoxRange zr = tap_range(tabz, pd, pd + 1, ns), yr;
    bool staged = false;
    if (ph == 0 && (long long)zr.n * ny * nx <= cap) {
      staged = true;
      int sb = pd + 1;
      while (sb < pd1) {
        const BoxRange nxt = tap_range(tabz, pd, sb + 1, ns);
        if ((long long)nxt.n * ny * nx > cap) break;
        zr = nxt; ++sb;
      }
      row_end = sb * PH;
    } else {                                                           // pooled depths that read their taps from global memory
      int sb = pd + 1;
      while (sb < pd1 && (long long)tap_range(tabz, sb, sb + 1, ns).n * ny * nx > cap) ++sb;
      row_end = sb * PH;
    }
    yr.lo = y0; yr.n = ny;
    __syncthreads();                                                 // the previous segment's bins are done with box and tables
    if (staged && !rel) {           // first staged segment: x / y tables -> float4 offsets relative to (y0, x0)
      for (int e = tid; e < PW * ns; e += kClThreads) { tabx[e].lo = (tabx[e].lo - x0) * L; tabx[e].hi = (tabx[e].hi - x0) * L; }
      for (int e = tid; e < PH * ns; e += kClThreads) { taby[e].lo = (taby[e].lo - y0) * rowlen4; taby[e].hi = (taby[e].hi - y0) * rowlen4; }
      rel = true;
    }
    if (staged) {
      const int units = zr.n * yr.n * pieces;
      for (int u0 = wave; u0 < units; u0 += (kClThreads / MR3D_WAVE) * kBxUnits) {
        float4 v[kBxUnits];
#pragma unroll
        for (int k = 0; k < kBxUnits; ++k) {
          const int u = u0 + k * (kClThreads / MR3D_WAVE);
          const int rr = u / pieces, idx = (u - rr * pieces) * 64 + lane;
          if (u < units && idx < rowlen4) {
            const int zz = rr / yr.n, yy = rr - zz * yr.n;
            v[k] = ld_feat4(feat + base_n + (((size_t)(zr.lo + zz) * H + (yr.lo + yy)) * W + x0 + idx / L) * C + (idx % L) * 4);
          }
        }
#pragma unroll
        for (int k = 0; k < kBxUnits; ++k) {
          const int u = u0 + k * (kClThreads / MR3D_WAVE);
          const int rr = u / pieces, idx = (u - rr * pieces) * 64 + lane;
          if (u < units && idx < rowlen4) box[rr * rowlen4 + idx] = v[k];
        }
      }
    }
    __syncthreads();
    // ---- bins of the segment: results go to tile[channel][bin - flushed], flushed to out[r, c0 + channel, ...] in runs
    const int bin_end = row_end * PW, plane4 = yr.n * rowlen4;
    const float4 *boxq = box + q - (yr.lo - y0) * rowlen4;            // taby offsets are relative to y0
    for (int b0 = row * PW; b0 < bin_end; b0 += SLOTS) {
      const int bin = b0 + slot;
      if (bin < bin_end) {
        const int bpd = bin / PHW, rem = bin - bpd * PHW, bph = rem / PW, bpw = rem - bph * PW;
        float4 acc;
        if (staged) {
          if (!guard) acc = box_bin<T, NS, true, false>(tabx, taby, tabz, bpd, bph, bpw, ns, boxq, zr.lo, plane4, fq, sD);
          else acc = box_bin<T, NS, true, true>(tabx, taby, tabz, bpd, bph, bpw, ns, boxq, zr.lo, plane4, fq, sD);
        } else {
          if (!guard) acc = box_bin<T, NS, false, false>(tabxg, tabyg, tabz, bpd, bph, bpw, ns, boxq, zr.lo, plane4, fq, sD);
          else acc = box_bin<T, NS, false, true>(tabxg, tabyg, tabz, bpd, bph, bpw, ns, boxq, zr.lo, plane4, fq, sD);
        }
        float *t = tile + (4 * q) * TS + (bin - flushed);
        t[0] = acc.x * inv_count; t[TS] = acc.y * inv_count; t[2 * TS] = acc.z * inv_count; t[3 * TS] = acc.w * inv_count;
      }
      const int done = min(b0 + SLOTS, bin_end);                     // bins [flushed, done) are in the tile
      if (done - flushed + SLOTS > tbins || done == pd1 * PHW) {
        __syncthreads();
        const int nb = done - flushed;
        if (tbins <= 32) {                                           // two channels x 32 bins per wave store
          for (int ch = wave * 2 + (lane >> 5); ch < CH; ch += 2 * (kClThreads / MR3D_WAVE))
            if ((lane & 31) < nb) out[((size_t)r * C + c0 + ch) * B + flushed + (lane & 31)] = tile[ch * TS + (lane & 31)];
        } else {
          for (int ch = wave; ch < CH; ch += kClThreads / MR3D_WAVE)
            for (int bI = lane; bI < nb; bI += MR3D_WAVE) out[((size_t)r * C + c0 + ch) * B + flushed + bI] = tile[ch * TS + bI];
        }
        flushed = done;
        __syncthreads();
      }
    }
    row = row_end;
  }
}

// launch of one (NS, CH) instance; the LDS attribute is raised once per instance
template <typename T, int NS, int CH>
int launch_fwd_box(const T *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int ns, int C, int D,
                   int H, int W, int PD, int PH, int PW, int pdc, int lds_budget, const LevelTable &lt, hipStream_t stream) {
  // results tile: 32 bins (two passes of the 16 bin slots at CH = 64); the box gets what is left of the LDS budget
  const int tbins = (kClThreads / (CH / 4)) > 32 ? (kClThreads / (CH / 4)) : 32;          // at least one pass of the bin slots
  const int tile_bytes = CH * (tbins + 1) * 4;
  const int cap = (lds_budget - kBxTabBytes - tile_bytes) / (CH * 4);
  if (cap < 16) return MR3D_ERR_UNSUPPORTED;
  const int lds = kBxTabBytes + cap * CH * 4 + tile_bytes;
  static std::mutex mu;
  static int attr_bytes = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (lds > attr_bytes) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(roi_align3d_fwd_box<T, NS, CH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              lds) != hipSuccess) {
        (void)hipGetLastError();
        return MR3D_ERR_LAUNCH;
      }
      attr_bytes = lds;
    }
  }
  dim3 grid(num_rois, C / CH, (PD + pdc - 1) / pdc);
  roi_align3d_fwd_box<T, NS, CH><<<grid, kClThreads, lds, stream>>>(feat, out, rois, scale, scale_d, ns, C, D, H, W, PD, PH, PW, pdc, cap, tbins, lt);
  return mr3d_launch_status();
}

// Measured on the SURVEY 8(d) inputs (MI355X, kernel time): bbox extractor (3,7,7) x 512 rois 57 us against 86 us wave-per-bin
// and 88 us separable (bf16 features 60 against 87); mask extractor (10,14,14) x 128 rois 157 us against 138 us separable, so
// only shapes with PD <= 4 come here by default (MR3D_ROIALIGN_BOX=2 sends every shape, =0 none).  Every workgroup of the bbox
// case is resident at once (512 workgroups, two per CU), so the launch lasts as long as its slowest RoI: median workgroup 22 us
// (geometry + tables 4.6, staging 5, bins 10, stores 2), RoIs whose pooled depths do not fit the box ~45 us.  Tried and not
// kept: 32 channels per workgroup with a 160-bin tile (two rounds of workgroups: 62 us), one pooled depth per workgroup
// (61 us), splitting a pooled depth that does not fit into ph rows (each segment pays a staging latency: 90 us), contracting
// the x-taps before the y/z weights (less VALU, 76 us: the bins are latency-, not issue-bound).
template <typename T>
int try_fwd_box(const T *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int ns, int C, int D,
                int H, int W, int PD, int PH, int PW, const LevelTable &lt, hipStream_t stream) {
  if (PD > 4) return MR3D_ERR_UNSUPPORTED;          // many pooled depths (mask extractor): the separable kernel wins (157 vs 138 us)
  if (C % 64 != 0 || ns <= 0 || PD * ns > kStAxis || PH * ns > kStAxis || PW * ns > kStAxis) return MR3D_ERR_UNSUPPORTED;
  const int pdc = PD;
  // (round 4: 16 channels per workgroup -- 4x the workgroups, six per CU, 26 KB boxes -- measured 84 us against 52 us on the bbox case:
  //  the per-workgroup geometry / table phases are paid four times and the 64-byte voxel rows conflict in LDS)
  const int budget = 80 * 1024;               // two workgroups per CU
  if (ns == 2) return launch_fwd_box<T, 2, 64>(feat, out, rois, num_rois, scale, scale_d, ns, C, D, H, W, PD, PH, PW, pdc, budget, lt, stream);
  return launch_fwd_box<T, 0, 64>(feat, out, rois, num_rois, scale, scale_d, ns, C, D, H, W, PD, PH, PW, pdc, budget, lt, stream);
}

// ------------------------------------------------------------------------------------------
// channels-last FORWARD, one WAVE per (RoI, 64 channels, chunk of bins) (round 4).
// The box kernel above is a chain of workgroup phases -- tables, extents, staging, bins, tile flush, each behind a barrier -- and a
// launch lasts as long as its slowest workgroup (rois whose box does not fit LDS read their taps from global memory: 45 of the
// 52 us).  RoIAlign3D's samples are separable: with the per-axis tap weights,
//     out[pd][ph][pw] = 1/count * sum_z Wz[pd][z] sum_y Wy[ph][y] sum_x Wx[pw][x] F[z][y][x].
// A lane owns ONE channel, so every weight, index and loop bound is wave-uniform (scalar registers, scalar branches).  The voxel
// rows of the footprint travel global -> LDS directly (`buffer_load_dword ... lds`: one 256-byte row of 64 channels per load, no
// vector registers, no vector address arithmetic), 64 voxels per batch, every DISTINCT voxel once; per voxel row (z, y) the wave
// forms T1[pw] = the row's x-contraction (2 taps per x-sample, read back from LDS) and adds wz * wy * T1 into the bins (pd, ph)
// whose samples touch plane z and row y.  No workgroup barrier inside the loop (the LDS rows are the wave's own).  The bin
// accumulators stay in registers and leave through an LDS tile [bin][channel] (it shares the row buffer's space), so that
// [R,C,PD,PH,PW] rows are written in runs of consecutive bins.  Same samples, taps and weights as the reference (axis_tap /
// sample_coord); the fp32 sums are associated per axis instead of per sample (<= 5e-7 against the box kernel on the bench rois).
// Measured on the way (bbox extractor, bench rois): one row of per-sample loads at a time 194 us; four rows of per-sample loads in
// flight 78 us (176 MB of load requests for 37 MB of distinct voxels: request-bound); this form: see try_fwd_wave.
// ------------------------------------------------------------------------------------------
constexpr int kWvSlots = 64;               // voxel rows (256 B each) of the wave's LDS row buffer: 16 KB

template <int PDC, int PHC, int PWT, int NS>
__global__ __launch_bounds__(MR3D_WAVE) void roi_align3d_fwd_wave(const float *feat, float *__restrict__ out, const float *__restrict__ rois,
                                                                  float scale, float scale_d, int C, int D, int H, int W, int PD, int PH,
                                                                  LevelTable lt) {
  constexpr int NXS = PWT * NS, NYS = PHC * NS, NZS = PDC * NS;
  constexpr int TILE = PHC * PWT * (MR3D_WAVE + 1), BUF = kWvSlots * MR3D_WAVE;
  __shared__ IdxEntry tabx[NXS], taby[NYS], tabz[NZS];
  __shared__ __attribute__((aligned(16))) float buf[TILE > BUF ? TILE : BUF];       // row buffer, then the output tile
  const int lane = threadIdx.x;
  const int r = blockIdx.x, c0 = blockIdx.y * MR3D_WAVE;
  const int phchunks = (PH + PHC - 1) / PHC;
  const int pd0 = ((int)blockIdx.z / phchunks) * PDC, ph0 = ((int)blockIdx.z % phchunks) * PHC;
  const int npd = min(PDC, PD - pd0), nph = min(PHC, PH - ph0);          // (scalar)
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    feat = pick4(lvl, lt.feat);
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, NS, PD, PH, PWT);
  // tap tables of the chunk's samples (lanes 0 .. NXS + NYS + NZS - 1, one sample each); an out-of-range sample has zero weights
  // and index -1 (it never matches a voxel)
  {
    const int e = lane;
    AxisTap t;
    IdxEntry te;
    te.lo = -1; te.hi = -1; te.l = 0.f; te.h = 0.f;
    if (e < NXS) {
      if (axis_tap(sample_coord(g.sw, e / NS, g.bw, e % NS, NS), W, t)) { te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h; }
      tabx[e] = te;
    } else if (e < NXS + NYS) {
      const int k = e - NXS;
      if (k / NS < nph && axis_tap(sample_coord(g.sh, ph0 + k / NS, g.bh, k % NS, NS), H, t)) { te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h; }
      taby[k] = te;
    } else if (e < NXS + NYS + NZS) {
      const int k = e - NXS - NYS;
      if (k / NS < npd && axis_tap(sample_coord(g.sd, pd0 + k / NS, g.bd, k % NS, NS), D, t)) { te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h; }
      tabz[k] = te;
    }
  }
  static_assert(NXS + NYS + NZS <= MR3D_WAVE, "one sample per lane");
  __syncthreads();
  auto sf = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  // footprint extents (scalar)
  int x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1, z0 = 0x7fffffff, z1 = -1;
#pragma unroll
  for (int e = 0; e < NXS; ++e) {
    const int lo = __builtin_amdgcn_readfirstlane(tabx[e].lo), hi = __builtin_amdgcn_readfirstlane(tabx[e].hi);
    if (lo >= 0) { x0 = min(x0, lo); x1 = max(x1, hi); }
  }
#pragma unroll
  for (int e = 0; e < NYS; ++e) {
    const int lo = __builtin_amdgcn_readfirstlane(taby[e].lo), hi = __builtin_amdgcn_readfirstlane(taby[e].hi);
    if (lo >= 0) { y0 = min(y0, lo); y1 = max(y1, hi); }
  }
#pragma unroll
  for (int e = 0; e < NZS; ++e) {
    const int lo = __builtin_amdgcn_readfirstlane(tabz[e].lo), hi = __builtin_amdgcn_readfirstlane(tabz[e].hi);
    if (lo >= 0) { z0 = min(z0, lo); z1 = max(z1, hi); }
  }
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
  // x taps as LDS byte offsets relative to the row's first voxel (lane's channel included) + the two weights
  int alo[NXS], ahi[NXS];
  float xwl[NXS], xwh[NXS];
#pragma unroll
  for (int e = 0; e < NXS; ++e) {
    const IdxEntry t = tabx[e];
    const bool ok = t.lo >= 0;
    alo[e] = ((ok ? t.lo - x0 : 0) * MR3D_WAVE + lane) * 4;
    ahi[e] = ((ok ? t.hi - x0 : 0) * MR3D_WAVE + lane) * 4;
    xwl[e] = sf(t.h);
    xwh[e] = sf(t.l);
  }
  // dense per-row y weights Wy[row][ph] in LDS (a clamped sample has lo == hi: both taps count)
  constexpr int YT = 64, PHP = 8;
  static_assert(PHC <= PHP, "row weights");
  __shared__ __attribute__((aligned(16))) float wyt[YT][PHP];
  auto row_weight = [&](int y, int a) {
    float w = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const IdxEntry t = taby[a * NS + i];
      if (t.lo == y) w += t.h;
      if (t.hi == y) w += t.l;
    }
    return w;
  };
  for (int idx = lane; idx < min(ny, YT) * PHP; idx += MR3D_WAVE) {
    const int row = idx / PHP, a = idx - row * PHP;
    wyt[row][a] = a < PHC ? row_weight(y0 + row, a) : 0.f;
  }
  __syncthreads();

  float acc[PDC][PHC * PWT];
#pragma unroll
  for (int a = 0; a < PDC; ++a)
#pragma unroll
    for (int b = 0; b < PHC * PWT; ++b) acc[a][b] = 0.f;

  const int row_bytes = W * C * 4;                       // (a feature plane stays below 2^31 bytes: checked by the launcher)
  const int xstride = C * 4;
  // rows per batch: as many whole voxel rows as the buffer holds (a row wider than the buffer is walked in 64-voxel pieces by the
  // per-sample path below)
  const bool wide = nx > kWvSlots;
  const int rpb = wide ? 1 : max(1, kWvSlots / max(nx, 1));
  for (int z = z0; z <= z1; ++z) {
    float wz[PDC];
    bool zany = false;
#pragma unroll
    for (int a = 0; a < PDC; ++a) {
      float w = 0.f;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const IdxEntry t = tabz[a * NS + i];
        if (t.lo == z) w += t.h;
        if (t.hi == z) w += t.l;
      }
      wz[a] = sf(w);
      zany = zany || wz[a] != 0.f;
    }
    if (!zany) continue;
    const float *plane = feat + (((size_t)g.b * D + z) * H) * (size_t)W * C + c0;
    const __amdgpu_buffer_rsrc_t pr = buffer_of(plane);
    for (int yb = y0; yb <= y1; yb += rpb) {
      const int rows = min(rpb, y1 - yb + 1);
      if (!wide) {
        // every distinct voxel of the batch's rows: global -> LDS, one 256-byte row of 64 channels per load
        for (int k = 0; k < rows; ++k) {
          const int gb = (yb + k) * row_bytes + x0 * xstride;
          for (int i = 0; i < nx; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (__attribute__((address_space(3))) void *)(buf + (k * nx + i) * MR3D_WAVE), 4,
                                                     lane * 4, gb + i * xstride, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0): the rows are in LDS (lgkmcnt / expcnt untouched)
      }
      for (int k = 0; k < rows; ++k) {
        const int y = yb + k;
        float wy[PHC];
        if (y - y0 < YT) {
          const float4 q0 = *reinterpret_cast<const float4 *>(&wyt[y - y0][0]), q1 = *reinterpret_cast<const float4 *>(&wyt[y - y0][4]);
          const float q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
          for (int a = 0; a < PHC; ++a) wy[a] = sf(q[a]);
        } else {
#pragma unroll
          for (int a = 0; a < PHC; ++a) wy[a] = sf(row_weight(y, a));
        }
        bool any = false;
#pragma unroll
        for (int a = 0; a < PHC; ++a) any = any || wy[a] != 0.f;
        if (!any) continue;
        // T1[pw] = sum over the bin's x-samples of (w_lo F[lo] + w_hi F[hi])
        float t1[PWT];
        if (!wide) {
          const char *rowp = reinterpret_cast<const char *>(buf) + k * nx * (MR3D_WAVE * 4);
#pragma unroll
          for (int pw = 0; pw < PWT; ++pw) {
            float s_ = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
              const int e = pw * NS + i;
              s_ += xwl[e] * *reinterpret_cast<const float *>(rowp + alo[e]) + xwh[e] * *reinterpret_cast<const float *>(rowp + ahi[e]);
            }
            t1[pw] = s_;
          }
        } else {
          const int rb = y * row_bytes + x0 * xstride;
#pragma unroll
          for (int pw = 0; pw < PWT; ++pw) {
            float s_ = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
              const int e = pw * NS + i;
              const int lo = __builtin_amdgcn_readfirstlane(alo[e] >> 8), hi = __builtin_amdgcn_readfirstlane(ahi[e] >> 8);     // voxel index relative to x0
              s_ += xwl[e] * buffer_load1(pr, (unsigned)lane * 4u, (unsigned)(rb + lo * xstride)) +
                    xwh[e] * buffer_load1(pr, (unsigned)lane * 4u, (unsigned)(rb + hi * xstride));
            }
            t1[pw] = s_;
          }
        }
#pragma unroll
        for (int a = 0; a < PDC; ++a) {
          if (wz[a] == 0.f) continue;
#pragma unroll
          for (int b = 0; b < PHC; ++b) {
            if (wy[b] == 0.f) continue;
            const float w = wz[a] * wy[b];
#pragma unroll
            for (int pw = 0; pw < PWT; ++pw) acc[a][b * PWT + pw] += w * t1[pw];
          }
        }
      }
    }
  }

  // ---- out[r][c0 + lane][pd][ph][pw]: per pooled depth through the LDS tile [bin][channel], then one run of the chunk's bins per
  // channel (the chunk's bins of one depth are contiguous in a channel's row)
  const float inv_count = 1.f / (float)(NS * NS * NS);
  const int B = PD * PH * PWT;
  float *orow = out + ((size_t)r * C + c0) * B;
  float (*tile)[MR3D_WAVE + 1] = reinterpret_cast<float (*)[MR3D_WAVE + 1]>(buf);
#pragma unroll
  for (int a = 0; a < PDC; ++a) {
    if (a >= npd) break;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < PHC * PWT; ++b) tile[b][lane] = acc[a][b] * inv_count;
    __syncthreads();
    const int nb = nph * PWT, first = ((pd0 + a) * PH + ph0) * PWT;
    if (lane < nb) {
      float *dst = orow + first + lane;
#pragma unroll 8
      for (int ch = 0; ch < MR3D_WAVE; ++ch) dst[(size_t)ch * B] = tile[lane][ch];
    }
  }
}

template <int PDC, int PHC, int PWT, int NS>
int launch_fwd_wave(const float *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int C, int D, int H, int W,
                    int PD, int PH, const LevelTable &lt, hipStream_t stream) {
  const dim3 grid(num_rois, C / MR3D_WAVE, ((PD + PDC - 1) / PDC) * ((PH + PHC - 1) / PHC));
  roi_align3d_fwd_wave<PDC, PHC, PWT, NS><<<grid, MR3D_WAVE, 0, stream>>>(feat, out, rois, scale, scale_d, C, D, H, W, PD, PH, lt);
  return mr3d_launch_status();
}

// fp32 channels-last features, C % 64 == 0, sample_num 2, the two extractor shapes of the shipped model; anything else: the kernels above
int try_fwd_wave(const float *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int ns, int C, int D, int H,
                 int W, int PD, int PH, int PW, const LevelTable &lt, hipStream_t stream) {
  if (C % MR3D_WAVE != 0 || ns != 2) return MR3D_ERR_UNSUPPORTED;
  // byte offsets inside one feature plane travel as 32-bit scalars
  long long plane = (long long)H * W * C * 4;
  for (int i = 0; i < lt.num; ++i) plane = std::max(plane, (long long)lt.H[i] * lt.W[i] * C * 4);
  if (plane >= (1LL << 31)) return MR3D_ERR_UNSUPPORTED;
  // Kernel times on the SURVEY 8(d) rois and on size classes of 512 / 128 rois (rocprofv3, us; box / sep = the kernels above):
  //   (3,7,7):    mixed 51 box, 74 wave | 8-16 px 36 / 28 | 8-32 px 47 / 39 | 48-128 px 39 / 32 | flat 64-128 x 2-4 px 54 / 68
  //   (10,14,14): mixed 72 sep, 63 wave (4 pooled rows per wave; 7 rows: 93) | 8-16 px 48 / 47 | 48-128 px 49 / 50 | flat 50 / 72
  // A wave is a serial chain (tables 5 us, one memory round trip per 64-voxel batch, tile flush 5-8 us), so rois with thousands of
  // footprint voxels (the log-uniform bench sizes produce 128 x 4 x 2 px slabs) take 37 batches: the box kernel keeps the (3,7,7)
  // extractor, this one takes the mask extractor's shape, where the footprint per (pooled depth, 4 pooled rows) chunk is small.
  if (PW == 14 && PH == 14) return launch_fwd_wave<1, 4, 14, 2>(feat, out, rois, num_rois, scale, scale_d, C, D, H, W, PD, PH, lt, stream);
  return MR3D_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// channels-last FORWARD, row streaming (round 6): one 4-wave workgroup per (RoI, 64 channels, PDC pooled depths) streams the voxel
// rows of the footprint through ONE shared LDS ring; wave w owns the pooled COLUMNS pw = w PWW .. w PWW + PWW - 1 of every bin.
// Why another kernel.  The box kernel's workgroups walk tables -> staging -> bins -> stores in lock-step, and its bins phase is
// bound by LDS BANDWIDTH, not latency: 147 bins x 64 taps x 256 B = 2.4 MB of ds_read_b128 per RoI, two RoIs per CU = 16 us at
// 128 B/clk.  The wave kernel above has the right arithmetic (separable: a few LDS reads per voxel row instead of 64 per bin) but a
// 10-13 us fixed cost per wave and one wave per (RoI, pooled depth): a RoI with a thousand footprint voxels is one serial chain.
// A first row-streaming form with one independent wave per (RoI, 2 pooled rows) (tools/diag/roi_rows_stamps.py, stamps per unit) showed
// what bounds this op on the chip: with every wave streaming, a CU's LDS-DMA path moves ~12 B/clk (an 8 KB batch took 3.8 us with
// eight waves per CU), so bytes loaded twice (rows shared by neighbouring units, 1.5x; whole 4-voxel pieces, 1.6x) are paid in
// full, and the launch ended with single waves walking 36-batch units at the latency-bound 8 KB per 1.2 us: 55 us.  Hence:
//   * every row of the footprint is loaded ONCE per workgroup, only rows and planes some sample touches, as 16-byte LDS-DMA
//     (`buffer_load_dwordx4 ... lds`: four voxels x 64 channels per instruction).  The ring holds four batches, three in flight
//     while one is contracted; batch i is requested by wave i % 4 alone, which therefore waits for it with a plain vmcnt(0), and
//     ONE barrier per batch publishes it (inline asm loads, so that the compiler's own LDS-alias wait does not drain the ring);
//   * the contraction is two small fp32 MFMA products instead of per-bin taps -- the bookkeeping of which voxel feeds which bin
//     becomes dense weight matrices: T1[pw][c] = sum_x Wx[pw][x] F[x][c] per voxel row (v_mfma_f32_16x16x4_f32, K = the row's
//     voxels, wave w owns channels 16 w .. 16 w + 15), then Out[(pd, ph)][pw, c] += sum_rows wz[pd][z] wy[ph][y] T1[row][pw, c]
//     for four rows at a time (K = the rows).  No run-time-indexed accumulator, no branch per row: what bounded the scalar
//     versions of this kernel was the instruction count of a wave (one instruction per ~4.5 cycles and ~25 cycles per taken
//     branch: with every load and every multiply switched off a 4-row batch still cost 0.8 us of cursor and weight code);
//   * tap tables never leave the registers: lane e holds sample e (x samples, then y, then z), extents come from one ballot per
//     axis (samples ascend, so the valid ones are a contiguous lane range), the touched rows / planes from two bit masks compacted
//     with one ds_permute each, the two A operands from ds_bpermute gathers (ALWAYS with all lanes active, see below);
//   * results leave straight from the accumulator registers: a lane holds whole pooled rows of its channel.
// Same samples, taps and weights as the reference (axis_tap / sample_coord); sums associated per axis as in the wave kernel.
// Rows wider than two ring buffers (> 8 LPB voxels) are evaluated bin by bin from global memory (correct, slow, never with the
// shipped level rule below 256-pixel RoIs).
// ------------------------------------------------------------------------------------------
typedef unsigned rows_desc_t __attribute__((ext_vector_type(4)));
#ifdef MR3D_ROWS_STAMPS                                // diagnostic builds: wall-clock stamps of a workgroup's phases (tools/diag/roi_rows_stamps.py)
__device__ unsigned long long g_rows_stamps[8192 * 16];
__device__ int g_rows_dbg;                              // timing-only ablations: 1 = no contraction, 2 = no row loads
#define ROWS_DBG(bit) (g_rows_dbg & (bit))
#define ROWS_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_rows_stamps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define ROWS_NOTE(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_rows_stamps[blockIdx.x * 16 + (k)] = (unsigned long long)(v); } while (0)
#define FOLD_MARK(k) do { const unsigned b_ = blockIdx.x + gridDim.x * blockIdx.z; if (threadIdx.x == 0 && b_ < 8192) g_rows_stamps[b_ * 16 + (k)] = wall_clock64(); } while (0)
#define FOLD_NOTE(k, v) do { const unsigned b_ = blockIdx.x + gridDim.x * blockIdx.z; if (threadIdx.x == 0 && b_ < 8192) g_rows_stamps[b_ * 16 + (k)] = (unsigned long long)(v); } while (0)
#else
#define FOLD_MARK(k) do { } while (0)
#define FOLD_NOTE(k, v) do { } while (0)
#define ROWS_MARK(k) do { } while (0)
#define ROWS_NOTE(k, v) do { } while (0)
#define ROWS_DBG(bit) 0
#endif

// one LDS-DMA piece: 64 lanes x 16 bytes from (descriptor base + per-lane byte offset) to LDS bytes [m0, m0 + 1024)
__device__ __forceinline__ void rows_dma16(unsigned m0, unsigned voff, rows_desc_t desc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(m0), "v"(voff), "s"(desc) : "memory");      // (m0: nothing else in these kernels uses it)
}

__device__ __forceinline__ float rows_rl(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ void rows_barrier() {                // raw barrier: the ring's LDS-DMA stays in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int kRowsWaves = 4, kRowsThreads = kRowsWaves * MR3D_WAVE;
typedef float rows_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int rows_bperm(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ float rows_bperm(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

template <int PDC, int PH_, int PW_, int NS, int LPB, int NB>
__global__ __launch_bounds__(kRowsThreads) void roi_align3d_fwd_rows(const float *feat, float *__restrict__ out, const float *__restrict__ rois,
                                                                     float scale, float scale_d, int C, int D, int H, int W, int PD,
                                                                     int num_rois, LevelTable lt) {
  constexpr int NXS = PW_ * NS, NYS = PH_ * NS, NZS = PDC * NS;
  static_assert(NS == 2, "two samples per bin and axis");
  static_assert(NXS + NYS + NZS <= MR3D_WAVE, "one sample per lane");
  static_assert(PW_ <= 16, "the pooled columns are the rows of one MFMA tile");
  static_assert((LPB == 4 || LPB == 8) && NB == 4, "at most three batches in flight: a wave waits for its batch with vmcnt(0)");
  constexpr int MP = PDC * PH_, MT = (MP + 15) / 16;              // (pooled depth, pooled row) pairs = rows of the second product, its tiles
  constexpr int PWP = PW_ <= 8 ? 8 : 16;                          // pooled columns of the T1 scratch
  constexpr int KXMAX = 2 * LPB;                                  // k-steps of the first product: rows of up to 8 LPB = 64 voxels
  constexpr int NBIN = PH_ * PW_;
  constexpr int RING = NB * LPB * 256;
  constexpr uint64_t YM = ((1ull << NYS) - 1) << NXS, ZM = ((1ull << NZS) - 1) << (NXS + NYS);
  __shared__ __attribute__((aligned(16))) float buf[RING];                           // the row ring
  __shared__ __attribute__((aligned(16))) float scr_all[kRowsWaves][4 * PWP * 16];   // per wave: T1 of a 4-row group, [row][pooled column][16 channels]
  __shared__ float wy_tab[2 * NYS * 16 * MT], wz_tab[2 * NZS * 16 * MT];             // the second product's A factors, [touched row / plane][pair]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;                          // MFMA 16x16x4: A row / B column, k; D row = 4 lk + r, column li
  // workgroup -> (RoI, channel group, pooled-depth chunk).  Workgroups are dealt round-robin over the 8 XCDs: the chunks of RoI r
  // sit on the workgroups b with b % 8 == r % 8, next to each other in dispatch order, so that the planes two chunks share come
  // out of one L2 (placement is a speed matter only)
  const int pdch = (PD + PDC - 1) / PDC, upr = (C / MR3D_WAVE) * pdch;
  const int bid = blockIdx.x, slot8 = bid >> 3;
  const int r = (slot8 / upr) * 8 + (bid & 7);
  ROWS_MARK(0);
  if (r >= num_rois) return;
  const int u = slot8 % upr;
  const int pd0 = (u % pdch) * PDC, c0 = (u / pdch) * MR3D_WAVE;
  const int npd = min(PDC, PD - pd0);                               // (scalar)
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    feat = pick4(lvl, lt.feat);
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, NS, PD, PH_, PW_);
  // ---- lane e = sample e: x samples of every bin, then the y samples, then the chunk's z samples (every wave holds the whole table).
  // An out-of-range sample (or one of a pooled depth beyond the map) has indices -1 (it never matches a voxel) and zero weights.
  int tlo = -1, thi = -1;
  float tl = 0.f, th = 0.f;
  {
    const bool isx = lane < NXS, isy = !isx && lane < NXS + NYS, isz = !isx && !isy && lane < NXS + NYS + NZS;
    const int k = isx ? lane : (isy ? lane - NXS : lane - NXS - NYS);
    const int p = (isz ? pd0 : 0) + k / NS;
    const bool in = isx || isy || (isz && k / NS < npd);
    AxisTap t;
    if (in && axis_tap(sample_coord(isx ? g.sw : (isy ? g.sh : g.sd), p, isx ? g.bw : (isy ? g.bh : g.bd), k % NS, NS),
                       isx ? W : (isy ? H : D), t)) {
      tlo = t.lo; thi = t.hi; tl = t.l; th = t.h;
    }
  }
  const uint64_t vm = __builtin_amdgcn_ballot_w64(tlo >= 0);
  ROWS_MARK(1);
  const uint64_t xm = vm & ((1ull << NXS) - 1), ym = vm & YM, zm = vm & ZM;
  const bool nonempty = xm != 0 && ym != 0 && zm != 0;            // else: every sample of the chunk is dropped, the bins are 0

  rows_f32x4 acc[MT][PW_];                                         // this wave's 16 channels of Out[(pd, ph)][pw]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int pw = 0; pw < PW_; ++pw) acc[mt][pw] = (rows_f32x4){0.f, 0.f, 0.f, 0.f};

  const int B = PD * NBIN;
  const float inv_count = 1.f / (float)(NS * NS * NS);
  float *oblock = out + ((size_t)r * C + c0) * B;                  // this RoI's [64 channels][B] block
  const float *fb = feat + (size_t)g.b * D * H * W * C + c0;       // channel group c0 of the RoI's image
  const unsigned nrec = (unsigned)(((size_t)D * H * W * C - c0) * 4);

  if (nonempty) {
    // samples ascend along an axis: the first / last valid lane of an axis holds its lowest / highest voxel
    const int x0 = __builtin_amdgcn_readlane(tlo, __builtin_ctzll(xm)), x1 = __builtin_amdgcn_readlane(thi, 63 - __builtin_clzll(xm));
    const int y0 = __builtin_amdgcn_readlane(tlo, __builtin_ctzll(ym)), y1 = __builtin_amdgcn_readlane(thi, 63 - __builtin_clzll(ym));
    const int z0 = __builtin_amdgcn_readlane(tlo, __builtin_ctzll(zm)), z1 = __builtin_amdgcn_readlane(thi, 63 - __builtin_clzll(zm));
    const int nx = x1 - x0 + 1, pieces = (nx + 3) >> 2;
    // touched planes / rows as bit masks relative to z0 / y0, then as lists (lane j: the j-th touched index): a row of the flattened
    // (plane, row) sequence is two readlanes -- the streaming loop below has no search loop and next to no scalar branch (a taken
    // branch costs ~25 cycles here: a first version that walked the rows with a cursor spent 0.8 us per 4-row batch on control flow
    // with every load and every multiply switched off, tools/diag/roi_rows_stamps.py ablations)
    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    if (pieces <= 2 * LPB && ny <= 64 && nz <= 64) {
      uint64_t ybits = 0, zbits = 0;
#pragma unroll
      for (int e = NXS; e < NXS + NYS; ++e) {
        const int lo = __builtin_amdgcn_readlane(tlo, e), hi = __builtin_amdgcn_readlane(thi, e);
        if (lo >= 0) ybits |= (1ull << (lo - y0)) | (1ull << (hi - y0));
      }
#pragma unroll
      for (int e = NXS + NYS; e < NXS + NYS + NZS; ++e) {
        const int lo = __builtin_amdgcn_readlane(tlo, e), hi = __builtin_amdgcn_readlane(thi, e);
        if (lo >= 0) zbits |= (1ull << (lo - z0)) | (1ull << (hi - z0));
      }
      const int nyt = __builtin_popcountll(ybits), nzt = __builtin_popcountll(zbits), nrows = nyt * nzt;
      auto compact = [&](uint64_t bits, int nset) {                 // lane j < nset: the index of the j-th set bit
        const bool set = (bits >> lane) & 1;
        const int below = __builtin_popcountll(bits & ((1ull << lane) - 1));        // set bits below this lane
        const int dst = set ? below : nset + (lane - below);       // a permutation of the lanes
        return __builtin_amdgcn_ds_permute(dst << 2, lane);
      };
      const int yrel = compact(ybits, nyt), zrel = compact(zbits, nzt);             // lane j: j-th touched row / plane, relative to y0 / z0
      const int yoffs = (y0 + yrel) * W * C * 4, zoffs = (z0 + zrel) * H * W * C * 4;   // ... as byte offsets inside the image

      // ---- A operand of the first product, T1[pw][c] = sum_x Wx[pw][x] F[x][c]: lane (li, lk) holds Wx[pw = li][x0 + 4 ks + lk], the
      // summed tap weights of bin column li's two samples on that voxel (a clamped sample has lo == hi: both taps count), zero
      // beyond the row's last voxel.  Rows of up to 16 voxels (the common case) run a loop unrolled for their k-step count; a longer
      // row owns a whole double-size buffer and runs all 2 LPB k-steps (the ring is zero-filled at the start: stale bytes are finite)
      const int kx = pieces;                                         // k-steps of a row
      float ax[KXMAX];
      {
        const int e0 = min(li * NS, NXS - 1), e1 = min(li * NS + 1, NXS - 1);
        const bool on = li < PW_;
        // (every ds_bpermute runs with all lanes active, the selects follow: a permute inside `on ? ... : ...` executes under a partial
        // mask and an inactive SOURCE lane reads as 0 -- the first build of this kernel lost every pooled column >= PW / 2 that way)
        const int lo0 = rows_bperm(tlo, e0), hi0 = rows_bperm(thi, e0), lo1 = rows_bperm(tlo, e1), hi1 = rows_bperm(thi, e1);
        const float l0p = rows_bperm(tl, e0), h0p = rows_bperm(th, e0), l1p = rows_bperm(tl, e1), h1p = rows_bperm(th, e1);
        const float l0 = on ? l0p : 0.f, h0 = on ? h0p : 0.f, l1 = on ? l1p : 0.f, h1 = on ? h1p : 0.f;
        auto wx = [&](int ks) {
          const int x = x0 + 4 * ks + lk;
          return ((lo0 == x ? h0 : 0.f) + (hi0 == x ? l0 : 0.f)) + ((lo1 == x ? h1 : 0.f) + (hi1 == x ? l1 : 0.f));
        };
#pragma unroll
        for (int ks = 0; ks < KXMAX; ++ks) ax[ks] = wx(ks);
      }
      // ---- A operand of the second product, Out[(pd, ph)][.] = sum_rows wz[pd][z_row] wy[ph][y_row] T1[row][.]: lane (li, lk) of tile mt
      // owns pair m = 16 mt + li and multiplies it with row lk of a 4-row group.  Its two factors come from tables in LDS,
      // WY[touched row][m] and WZ[touched plane][m] (the summed tap weights of the pair's samples on that row / plane; at most 2 NYS
      // rows and 2 NZS planes are touched), built here: lane (li, lk) of wave w fills rows lk + 4 w, + 16, ...
      constexpr int MTW = 16 * MT;
      {
        float *wyt = wy_tab, *wzt = wz_tab;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = 16 * mt + li, pa = m / PH_, ph = m - pa * PH_;
          const bool on = m < MP;
          int slo[2][NS], shi[2][NS];
          float sl[2][NS], sh[2][NS];
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            const int ey = min(NXS + ph * NS + i, 63), ez = min(NXS + NYS + pa * NS + i, 63);
            slo[0][i] = rows_bperm(tlo, ey); shi[0][i] = rows_bperm(thi, ey); sl[0][i] = rows_bperm(tl, ey); sh[0][i] = rows_bperm(th, ey);
            slo[1][i] = rows_bperm(tlo, ez); shi[1][i] = rows_bperm(thi, ez); sl[1][i] = rows_bperm(tl, ez); sh[1][i] = rows_bperm(th, ez);
          }
#pragma unroll
          for (int ax_ = 0; ax_ < 2; ++ax_) {
            const int nt = ax_ == 0 ? nyt : nzt, first = ax_ == 0 ? y0 : z0;
            float *tab = ax_ == 0 ? wyt : wzt;
            for (int j0 = 4 * wave; j0 < nt; j0 += 4 * kRowsWaves) {          // (uniform trip count: the permutes run with all lanes active)
              const int j = j0 + lk;
              const int v = first + rows_bperm(ax_ == 0 ? yrel : zrel, min(j, 63));
              const float w = ((slo[ax_][0] == v ? sh[ax_][0] : 0.f) + (shi[ax_][0] == v ? sl[ax_][0] : 0.f)) +
                              ((slo[ax_][1] == v ? sh[ax_][1] : 0.f) + (shi[ax_][1] == v ? sl[ax_][1] : 0.f));
              if (j < nt) tab[j * MTW + m] = on ? w : 0.f;
            }
          }
        }
      }
      const uint64_t fbu = mr3d_uniform64((uint64_t)(uintptr_t)fb);                 // (float geometry is uniform, but lives in vector registers)
      const rows_desc_t desc = {(unsigned)fbu, (unsigned)(fbu >> 32) & 0xffffu, (unsigned)__builtin_amdgcn_readfirstlane((int)nrec), 0x00020000u};
      const unsigned lds_base = (unsigned)(uintptr_t)buf;           // LDS byte address of the ring (the low half of its flat address)
      const unsigned vstride = (unsigned)C * 4u;                    // bytes between neighbouring voxels of a row
      const unsigned lanepart = (unsigned)(lane >> 4) * vstride + (unsigned)(lane & 15) * 16u + (unsigned)x0 * vstride;
      // ring geometry: NB buffers of LPB pieces; a row of more than 4 LPB voxels takes buffers of twice the size
      const int bufp = pieces <= 4 && pieces <= LPB ? LPB : 2 * LPB, nbuf = pieces <= 4 && pieces <= LPB ? NB : NB / 2;
      const int rpb = pieces <= 4 ? bufp / pieces : 1;              // whole rows per buffer (a long row is read KXMAX k-steps deep: alone in its buffer)
      int rl = 0, zil = 0, yil = 0;                                 // the loader's position in the row sequence: row, (plane, row) list indices
      // batch i is requested by wave i % 4 alone (the others only move the cursor): with at most three batches in flight a wave
      // has nothing younger than the batch it must wait for, so its wait is vmcnt(0) -- no per-wave piece counts, no counted waits
      auto issue = [&](int bsel, bool mine) {                       // the next batch of rows into buffer bsel
        const int n = min(rpb, nrows - rl);
        if (mine && !ROWS_DBG(2)) {
          int zi = zil, yi = yil;
          unsigned m0 = lds_base + (unsigned)((bsel * bufp) << 10);
          for (int k = 0; k < n; ++k) {
            unsigned voff = lanepart + (unsigned)(__builtin_amdgcn_readlane(zoffs, zi) + __builtin_amdgcn_readlane(yoffs, yi));
            for (int pc = 0; pc < pieces; ++pc, voff += 4 * vstride, m0 += 1024) rows_dma16(m0, voff, desc);
            const bool wrap = yi + 1 == nyt;
            yi = wrap ? 0 : yi + 1;
            zi += wrap ? 1 : 0;
          }
        }
        rl += n;
        yil += n;                                                   // (n <= 8 rows: a few wraps at most)
        while (yil >= nyt) { yil -= nyt; ++zil; }
      };
      // ---- contraction.  A voxel row of the ring becomes T1[pw][16 channels] (first product, K = the row's voxels) and parks in
      // the wave's scratch; every four rows the second product adds them into the accumulators (K = the four rows)
      float *scr = scr_all[wave];
      for (int i = lane; i < 4 * PWP * 16; i += MR3D_WAVE) scr[i] = 0.f;         // (a row slot never written multiplies a zero weight: keep it finite)
      if (kx > 4) {                                                  // long rows read their buffer to its end
        for (int i = tid; i < RING / 4; i += kRowsThreads) reinterpret_cast<float4 *>(buf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
      }
      const int fbase = lk * 256 + (16 * wave + li) * 4;             // this lane's B element of a k-step: voxel lk, channel 16 wave + li
      const int swr = (4 * lk * 16 + li) * 4, srd = (lk * PWP * 16 + li) * 4;       // scratch bytes: D rows 4 lk + r of a row slot / B row lk
      // this lane's row of the current 4-row group, as list indices (plane, row): row lk of group 0, then + 4 per group
      int gyi = lk, gzi = 0;
      while (gyi >= nyt) { gyi -= nyt; ++gzi; }
      const int gq4 = 4 / nyt, gr4 = 4 - gq4 * nyt;
      auto second = [&]() {
        const bool live = gzi < nzt;                                 // (beyond the last row: weight 0)
        const char *py = reinterpret_cast<const char *>(wy_tab) + ((live ? gyi : 0) * MTW + li) * 4;
        const char *pz = reinterpret_cast<const char *>(wz_tab) + ((live ? gzi : 0) * MTW + li) * 4;
        float aw[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float w = *reinterpret_cast<const float *>(py + mt * 64) * *reinterpret_cast<const float *>(pz + mt * 64);
          aw[mt] = live ? w : 0.f;
        }
#pragma unroll
        for (int pw = 0; pw < PW_; ++pw) {
          const float bv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(scr) + srd + pw * 64);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][pw] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], bv, acc[mt][pw], 0, 0, 0);
        }
        gyi += gr4; gzi += gq4;
        if (gyi >= nyt) { gyi -= nyt; ++gzi; }
      };
      int rp = 0;                                                    // rows contracted
      auto rows_of = [&](int bsel, auto KXC) {                       // the rows of buffer bsel; KXC: compile-time k-steps
        constexpr int KX = decltype(KXC)::value;
        const int n = ROWS_DBG(1) ? 0 : min(rpb, nrows - rp);
        const char *rowp = reinterpret_cast<const char *>(buf) + ((bsel * bufp) << 10) + fbase;
        const int rstride = pieces << 10;
        // R rows at a time while they stay inside one 4-row group: their B reads are issued together and their MFMA chains are
        // independent (row by row, a wave waited for an LDS round trip, two dependent MFMAs and the result's way back per row)
        auto rows_at = [&](int k, auto RC) {
          constexpr int R = decltype(RC)::value;
          float bq[R][KX];
#pragma unroll
          for (int q = 0; q < R; ++q)
#pragma unroll
            for (int ks = 0; ks < KX; ++ks) bq[q][ks] = *reinterpret_cast<const float *>(rowp + q * rstride + ks * 1024);
          rows_f32x4 t1[R];
#pragma unroll
          for (int q = 0; q < R; ++q) t1[q] = (rows_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KX; ++ks)
#pragma unroll
            for (int q = 0; q < R; ++q) t1[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[ks], bq[q][ks], t1[q], 0, 0, 0);
          rowp += R * rstride;
          const int gi = (rp + k) & 3;                                // the first row's slot in its 4-row group
          if (4 * lk < PWP) {                                         // D rows 4 lk .. 4 lk + 3 = pooled columns
            char *d = reinterpret_cast<char *>(scr) + swr + gi * (PWP * 64);
#pragma unroll
            for (int q = 0; q < R; ++q)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) *reinterpret_cast<float *>(d + q * (PWP * 64) + rr * 64) = t1[q][rr];
          }
          // (compiler barrier: hipcc 7.2 sank second()'s scratch READS into the lane-masked block above in one copy of this loop --
          //  the lanes of k-groups 2 and 3 then multiplied stale registers; tests: rows of <= 4 voxels with the 8-piece ring)
          asm volatile("" ::: "memory");
          if (gi + R == 4) second();
        };
        int k = 0;
        if constexpr (KX <= 4) {
#ifndef MR3D_ROWS_NO_R4
          for (; k + 4 <= n && ((rp + k) & 3) == 0; k += 4) rows_at(k, std::integral_constant<int, 4>{});
#endif
          for (; k + 2 <= n && ((rp + k) & 1) == 0; k += 2) rows_at(k, std::integral_constant<int, 2>{});
        }
        for (; k < n; ++k) rows_at(k, std::integral_constant<int, 1>{});
        rp += n;
      };
      ROWS_MARK(2);
      // the streaming loop, once per compile-time k-step count: ONE switch in front of it (a switch per batch made the compiler
      // shuffle all 56 accumulator registers between the cases' assignments -- 220 moves per batch)
      auto stream = [&](auto KXC) {
        int nissued = 0, nproc = 0, bi = 0, bp = 0;                 // batches requested / contracted; the buffers they go to
        for (; nissued < nbuf - 1 && rl < nrows; ++nissued) {
          issue(bi, (nissued & (kRowsWaves - 1)) == wave);
          bi = bi + 1 == nbuf ? 0 : bi + 1;
        }
        while (nproc < nissued) {
          if ((nproc & (kRowsWaves - 1)) == wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's batch has landed ...
          rows_barrier();                                           // ... everybody sees it, and everybody is done with batch nproc - 1
#ifdef MR3D_ROWS_STAMPS
          if (nproc == 0) ROWS_MARK(3);
#endif
          if (rl < nrows) {                                         // its buffer takes the next batch
            issue(bi, (nissued & (kRowsWaves - 1)) == wave);
            bi = bi + 1 == nbuf ? 0 : bi + 1;
            ++nissued;
          }
          rows_of(bp, KXC);
          bp = bp + 1 == nbuf ? 0 : bp + 1;
          ++nproc;
        }
        ROWS_NOTE(8, nproc);
      };
      switch (kx) {
        case 1: stream(std::integral_constant<int, 1>{}); break;
        case 2: stream(std::integral_constant<int, 2>{}); break;
        case 3: stream(std::integral_constant<int, 3>{}); break;
        case 4: stream(std::integral_constant<int, 4>{}); break;
        default: stream(std::integral_constant<int, KXMAX>{}); break;
      }
      if (nrows & 3) second();                                      // the last, partial group
      ROWS_NOTE(9, pieces); ROWS_NOTE(10, rpb);
    } else {
      // ---- rows wider than the ring: bin by bin from global memory, straight to the output (lane = channel: strided stores)
      const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(fb), 0, (int)nrec, 0x00020000);
      for (int f = wave; f < npd * NBIN; f += kRowsWaves) {
        const int a = f / NBIN, rem = f - a * NBIN, b = rem / PW_, pw = rem - b * PW_;
        float v = 0.f;
        for (int iz = 0; iz < NS; ++iz) {
          const int ez = NXS + NYS + a * NS + iz;
          const int zlo_ = __builtin_amdgcn_readlane(tlo, ez), zhi_ = __builtin_amdgcn_readlane(thi, ez);
          const float zl2 = rows_rl(tl, ez), zh2 = rows_rl(th, ez);
          for (int iy = 0; iy < NS; ++iy) {
            const int ey = NXS + b * NS + iy;
            const int ylo_ = __builtin_amdgcn_readlane(tlo, ey), yhi_ = __builtin_amdgcn_readlane(thi, ey);
            const float yl2 = rows_rl(tl, ey), yh2 = rows_rl(th, ey);
            for (int ix = 0; ix < NS; ++ix) {
              const int ex = pw * NS + ix;
              const int xlo_ = __builtin_amdgcn_readlane(tlo, ex), xhi_ = __builtin_amdgcn_readlane(thi, ex);
              const float xl2 = rows_rl(tl, ex), xh2 = rows_rl(th, ex);
              if (zlo_ < 0 || ylo_ < 0 || xlo_ < 0) continue;
              auto at = [&](int zz, int yy, int xx) { return buffer_load1(fr, (unsigned)(((zz * H + yy) * W + xx) * C + lane) * 4u, 0u); };
              v += zh2 * (yh2 * (xh2 * at(zlo_, ylo_, xlo_) + xl2 * at(zlo_, ylo_, xhi_)) + yl2 * (xh2 * at(zlo_, yhi_, xlo_) + xl2 * at(zlo_, yhi_, xhi_))) +
                   zl2 * (yh2 * (xh2 * at(zhi_, ylo_, xlo_) + xl2 * at(zhi_, ylo_, xhi_)) + yl2 * (xh2 * at(zhi_, yhi_, xlo_) + xl2 * at(zhi_, yhi_, xhi_)));
            }
          }
        }
        oblock[(size_t)lane * B + (pd0 + a) * NBIN + rem] = v * inv_count;
      }
      return;
    }
  }

  // ---- out[r][c0 + channel][pd0 + pa][ph][0 .. PW): accumulator row 4 lk + r of tile mt is the pair (pa, ph), the column is the
  // channel: a lane holds the PW pooled columns of its pairs -- one contiguous run of PW floats per pair in the channel's row of the
  // output block.  Straight from the registers, 8 bytes per store where the run is 8-byte aligned (PW even): no LDS tile, no barrier
  // (a tile [channel][bin] copied out 16 bytes per lane cost two barriers per pass and 1.5 us (3,7,7) / 3.5 us (10,14,14) per
  // workgroup; the lines are completed in L2 by the neighbouring lanes' runs)
  ROWS_MARK(4);
  {
    float *och = oblock + (size_t)(16 * wave + li) * B + (size_t)pd0 * NBIN;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int m = 16 * mt + 4 * lk + rr, pa = m / PH_, ph = m - pa * PH_;
        if (m < MP && pa < npd) {
          float *run = och + pa * NBIN + ph * PW_;
          if constexpr (PW_ % 2 == 0) {
#pragma unroll
            for (int pw = 0; pw < PW_; pw += 2)
              *reinterpret_cast<float2 *>(run + pw) = make_float2(acc[mt][pw][rr] * inv_count, acc[mt][pw + 1][rr] * inv_count);
          } else {
#pragma unroll
            for (int pw = 0; pw < PW_; ++pw) run[pw] = acc[mt][pw][rr] * inv_count;
          }
        }
      }
  }
  ROWS_MARK(5);
#ifdef MR3D_ROWS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ROWS_MARK(6);
#endif
}

#ifdef MR3D_ROWS_STAMPS
}  // namespace
extern "C" int mr3d_debug_rows_ablate(int bits) { return hipMemcpyToSymbol(HIP_SYMBOL(g_rows_dbg), &bits, sizeof(int)) == hipSuccess ? 0 : -1; }
extern "C" int mr3d_debug_rows_stamps(unsigned long long *dst) {     // 8192 x 16 stamps of the last row-streaming launch
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rows_stamps), sizeof(unsigned long long) * 8192 * 16) == hipSuccess ? 0 : -1;
}
namespace {
#endif

template <int PDC, int PH_, int PW_, int LPB, int NB>
int launch_fwd_rows(const float *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int C, int D, int H, int W,
                    int PD, const LevelTable &lt, hipStream_t stream) {
  const int upr = (C / MR3D_WAVE) * ((PD + PDC - 1) / PDC);
  const int grid = ((num_rois + 7) / 8) * 8 * upr;
  roi_align3d_fwd_rows<PDC, PH_, PW_, 2, LPB, NB><<<grid, kRowsThreads, 0, stream>>>(feat, out, rois, scale, scale_d, C, D, H, W, PD, num_rois, lt);
  return mr3d_launch_status();
}

// fp32 channels-last features, C % 64 == 0, sample_num 2, the two extractor shapes of the shipped model; anything else: the kernels above
struct LevelTable;
int try_fwd_fold(const float *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int C, int D, int H, int W, int PD,
                 int PH, int PW, const LevelTable &lt, hipStream_t stream);         // (defined next to the plane-owner backward)

int try_fwd_rows(const float *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int ns, int C, int D, int H,
                 int W, int PD, int PH, int PW, const LevelTable &lt, hipStream_t stream) {
  const int form = g_mr3d_tuning[MR3D_TUNE_ROIALIGN_FORM];
  if (form == 1 || C % MR3D_WAVE != 0 || ns != 2) return MR3D_ERR_UNSUPPORTED;
  // byte offsets inside one image travel as 32-bit values, and the buffer descriptor's range check covers one image
  long long image = (long long)D * H * W * C * 4;
  for (int i = 0; i < lt.num; ++i) image = std::max(image, (long long)lt.D[i] * lt.H[i] * lt.W[i] * C * 4);
  if (image >= (1LL << 31) || (long long)PD * PH * PW * MR3D_WAVE * 4 >= (1LL << 31)) return MR3D_ERR_UNSUPPORTED;
  // The plane-folded kernel (further down, next to the backward it mirrors): bbox extractor 30.5 us against 40.8 for the row-streaming
  // kernel (kernel times, 512 rois; 13.8 against 24.7 us at 64 rois); mask extractor 53.4 against 49.9 us (27.7 against 35.0 at 64 rois:
  // a (RoI, pooled depth) unit of the mask extractor is ~9 voxel rows x 2 planes of work under the same set-up, four barriers and a
  // 50 KB output tile that keeps it at two workgroups per CU).  Default: folded for (.,7,7), row streaming for (.,14,14);
  // MR3D_TUNE_ROIALIGN_FORM = 2: row streaming for both, 8: folded for both, 4 / 5 / 6: the row-streaming instances below.
  if ((form == 0 && PW == 7 && PH == 7) || form == 8) {
    const int sto = try_fwd_fold(feat, out, rois, num_rois, scale, scale_d, C, D, H, W, PD, PH, PW, lt, stream);
    if (sto != MR3D_ERR_UNSUPPORTED) return sto;
  }
#define MR3D_ROWS(PDC, PH_, PW_, LPB, NB) return launch_fwd_rows<PDC, PH_, PW_, LPB, NB>(feat, out, rois, num_rois, scale, scale_d, C, D, H, W, PD, lt, stream)
  // Kernel times on the SURVEY 8(d) rois (rocprofv3 --kernel-trace, tools/r6_roi_kt.sh; round-5 kernels: box 54.5, wave 64.4 us):
  //   (3,7,7)   x 512: one pooled depth per workgroup, 4-piece buffers (26 KB of LDS, six workgroups per CU: all 1536 resident) 40.7 us;
  //                    8-piece buffers (43 KB, three per CU) 41.8; the three depths in one workgroup (two M-tiles) 57.0
  //   (10,14,14) x 128: one pooled depth per workgroup 50.5 / 48.9 us (4- / 8-piece buffers); two depths 55.8
  // Summing the planes of a voxel row before the products (one pooled depth per workgroup: b = sum_z wz[z] F[z][y][x], 2-3x fewer
  // MFMAs) was measured too: 40.0 against 41.8 us with 8-piece buffers, 49.0 against 40.7 with 4-piece ones (a buffer then holds one
  // or two y-rows: more barriers, no 4-row pipelining) -- the products are not what bounds a batch; not kept.
  // A dynamic form was built and measured as well (round 6): a one-workgroup plan kernel wrote a unit list into a caller-lent workspace --
  // (RoI, pooled depth, range of pooled rows), largest first, large chunks split by an estimate of their rows x pieces -- and persistent
  // workgroups drew units from it (first unit = own index, then a ticket counter: 1536 workgroups drawing their first ticket from ONE
  // counter staggered the launch by 15 us).  bbox: 37.9 us unsplit / 38.0-43.7 split at 256 / 128 / 64 KB against 40.7 static, plus 7.4 us
  // for the plan kernel; mask: 54-98 us against 49.  Every split unit pays the whole skeleton again and the order buys 3 us: not kept.
  // A register-only form in the image of the plane-owner backward was built and measured too (round 6): one workgroup per (RoI, pooled
  // depth), lane = channel, wave w owning four (two) pooled rows whose 4 x 14 (2 x 7) accumulators never leave its registers; per voxel
  // row V[pw] = sum_x Wx[pw][x] F[x] with the voxels of 3 - 12 rows (48 registers) requested together through per-row buffer descriptors
  // and Wx broadcast from LDS, then T[ph] += wz wy V; output through LDS as 8-byte pieces.  Deterministic, no barrier between set-up
  // and output, parity green -- bbox 42.4 - 43.3 us, mask 64.3 - 71.3 us against 42.4 / 51.6 for the kernels below on the same box: ~3,000
  // instructions per wave at one per 4.5 cycles with two or three waves per SIMD (194 registers) is the same wall from the other side.
  // Not kept.
  // Sharing a large chunk between workgroups that leave early when the RoI is small (each slot estimates the chunk's rows x pieces
  // from the RoI's extents) was measured and dropped: once every slot of the chip is taken, a workgroup's turnover costs ~24 ns
  // whatever its lifetime (4077 early leavers of 0.9 us each stretched the launch from 48 to 95 us).
  if (PW == 7 && PH == 7 && PD <= 3) {                 // the bbox extractor
    switch (form) {
      case 4: MR3D_ROWS(1, 7, 7, 8, 4);
      case 5: MR3D_ROWS(3, 7, 7, 4, 4);
      case 6: MR3D_ROWS(3, 7, 7, 8, 4);
      default: MR3D_ROWS(1, 7, 7, 4, 4);
    }
  }
  if (PW == 14 && PH == 14) {                          // the mask extractor
    switch (form) {
      case 4: MR3D_ROWS(1, 14, 14, 4, 4);
      case 5: MR3D_ROWS(2, 14, 14, 4, 4);
      case 6: MR3D_ROWS(2, 14, 14, 8, 4);
      default: MR3D_ROWS(1, 14, 14, 8, 4);
    }
  }
#undef MR3D_ROWS
  return MR3D_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// channels-last backward, voxel-owner form (C % 64 == 0).  A workgroup owns (RoI, 64 channels, a chunk of pooled depths): the
// chunk's top-gradient planes are staged once as G[bin][64 ch] in LDS, the per-axis weight matrices Wx[x][pw] (Wy, Wz alike:
// the summed tap weights of bin pw on voxel x) are built once together with the range of bins that reach each voxel (the
// bins touching a voxel are consecutive), and then every voxel of the RoI's footprint is evaluated by 16 lanes x float4
// channels:   dX[z][y][x] = 1/count * sum_{pd in range(z)} Wz[z][pd] * sum_{ph in range(y)} Wy[y][ph] * sum_{pw in range(x)} Wx[x][pw] * G[pd][ph][pw]
// -- on the SURVEY 8(d) rois ~8 non-zero terms per voxel, where the scatter form issues 64 atomics per output element and
// the separable sub-box kernel above pays four barriers per pooled depth and (sub-box, 16 channels) item.  One global atomic per
// touched voxel-channel (RoIs overlap).  No plan kernel, no sub-boxes; footprints longer than 64 voxels on an axis are walked
// in 64-voxel segments.  Same weights as the scatter form (l / h of axis_tap, dropped samples contribute nothing); the fp32
// sums are associated differently (per voxel instead of per atomic arrival order).
// ------------------------------------------------------------------------------------------
constexpr int kVxSeg = 64;                 // voxels per axis covered by one set of weight matrices
constexpr int kVxRow = 68;                 // floats per G row: 64 channels + 4 (16-byte aligned rows, 4-way instead of 32-way store conflicts)

__host__ __device__ inline int vox_lds_bytes(int pdc, int PH, int PW) {
  return 3 * kStAxis * 16 + 64 + (kVxSeg * (PW + PH + pdc)) * 4 + 3 * kVxSeg * 8 + 16 * 64 * 4 + pdc * PH * PW * kVxRow * 4;
}

__global__ __launch_bounds__(kClThreads) void roi_align3d_bwd_vox(const float *__restrict__ top, float *bottom,
                                                                 const float *__restrict__ rois, float scale, float scale_d,
                                                                 int ns, int C, int D, int H, int W, int PD, int PH, int PW,
                                                                 int pdc, int parts, int bug_compat, LevelTable lt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vx_lds[];
  IdxEntry *tabx = reinterpret_cast<IdxEntry *>(vx_lds), *taby = tabx + kStAxis, *tabz = taby + kStAxis;
  int *ext = reinterpret_cast<int *>(tabz + kStAxis);                 // x0,nx, y0,ny, z0,nz
  float *Wx = reinterpret_cast<float *>(vx_lds + 3 * kStAxis * 16 + 64), *Wy = Wx + kVxSeg * PW, *Wz = Wy + kVxSeg * PH;
  int2 *rx = reinterpret_cast<int2 *>(Wz + kVxSeg * pdc), *ry = rx + kVxSeg, *rz = ry + kVxSeg;   // [first, last] bin per voxel
  float *F = reinterpret_cast<float *>(rz + kVxSeg);                  // [16 voxel slots][64 channels]: flush transpose, per wave
  float *G = F + 16 * 64;                                             // [local bin][kVxRow]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = tid & 15, slot = tid >> 4;                            // float4 of channels, voxel slot
  const int r = blockIdx.x, c0 = blockIdx.y * 64, pd0 = ((int)blockIdx.z / parts) * pdc;
  int part = (int)blockIdx.z % parts;
  const int pd1 = min(pd0 + pdc, PD), npd = pd1 - pd0;
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    bottom = pick4(lvl, lt.grad);
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, ns, PD, PH, PW);
  // `parts` workgroups share the voxel passes of a LARGE footprint (every workgroup of the launch is resident at once, so the
  // launch lasts as long as its largest RoI); a small one is done by part 0 alone.  The rule only has to be the same in all
  // parts: an upper bound of the footprint from the RoI's size.
  if (parts > 1) {
    const float bound = (fminf(ceilf(g.bw * PW), (float)W) + 2.f) * (fminf(ceilf(g.bh * PH), (float)H) + 2.f) *
                        (fminf(ceilf(g.bd * npd), (float)D) + 2.f);
    if (bound <= 384.f) {
      if (part > 0) return;
      parts = 1;
    }
  }
  const int B = PD * PH * PW, PHW = PH * PW;
  const float inv_count = 1.f / (float)(ns * ns * ns);
  // ---- this chunk's top-gradient planes -> G[(pd - pd0) * PHW + ph * PW + pw][channel], pre-divided by the sample count.
  // The RoI's [64][B] block is walked flat (coalesced dwords); the reference's index bug reads plane pd at pd*PD*PW.
  {
    const float *src = top + ((size_t)r * C + c0) * B;
    const int nb = npd * PHW;
    for (int lb0 = 0; lb0 < nb; lb0 += MR3D_WAVE) {                  // a wave = 64 consecutive bins of one channel: 256-byte runs
      const int lb = lb0 + lane;
      const int pd = pd0 + lb / PHW, rem = lb % PHW;
      const int off = (bug_compat ? pd * PD * PW : pd * PHW) + rem;
      if (lb < nb) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = src[(size_t)(wave + 4 * k) * B + off];
#pragma unroll
        for (int k = 0; k < 16; ++k) G[lb * kVxRow + wave + 4 * k] = v[k] * inv_count;
      }
    }
  }
  build_idx_table(tabx, g.sw, g.bw, PW, ns, W, tid, kClThreads);
  build_idx_table(taby, g.sh, g.bh, PH, ns, H, tid, kClThreads);
  build_idx_table(tabz, g.sd, g.bd, PD, ns, D, tid, kClThreads);
  __syncthreads();
  if (tid < 3) {            // voxel range of the valid samples along x, y (all bins) and z (this chunk's pooled depths)
    const IdxEntry *tab = tid == 0 ? tabx : (tid == 1 ? taby : tabz);
    const int a = tid == 2 ? pd0 * ns : 0, b = tid == 0 ? PW * ns : (tid == 1 ? PH * ns : pd1 * ns);
    int mn = 0x7fffffff, mx = -1;
    for (int e = a; e < b; ++e)
      if (tab[e].l + tab[e].h != 0.f) { mn = min(mn, tab[e].lo); mx = max(mx, tab[e].hi); }
    ext[2 * tid] = mx < 0 ? 0 : mn;
    ext[2 * tid + 1] = mx < 0 ? 0 : mx - mn + 1;
  }
  __syncthreads();
  const int x0 = ext[0], nx = ext[1], y0 = ext[2], ny = ext[3], z0 = ext[4], nz = ext[5];
  if (nx == 0 || ny == 0 || nz == 0) return;                         // nothing of this chunk lands in the map (uniform)
  const float4 *Gq = reinterpret_cast<const float4 *>(G) + q;

  // weight matrix + bin range of one axis segment: one thread per voxel index
  auto build_axis = [&](float *Wm, int2 *rng, const IdxEntry *tab, int p0, int np, int v0, int vn, int t) {
    if (t < vn) {
      int first = np, last = -1;
      for (int p = 0; p < np; ++p) {
        float a = 0.f;
        for (int i = 0; i < ns; ++i) {
          const IdxEntry te = tab[(p0 + p) * ns + i];
          if (te.l + te.h == 0.f) continue;
          if (te.lo == v0 + t) a += te.h;
          if (te.hi == v0 + t) a += te.l;
        }
        Wm[t * np + p] = a;
        if (a != 0.f) { first = min(first, p); last = p; }
      }
      rng[t] = make_int2(first, last);
    }
  };
  for (int zs = 0; zs < nz; zs += kVxSeg)
    for (int ys = 0; ys < ny; ys += kVxSeg)
      for (int xs = 0; xs < nx; xs += kVxSeg) {
        const int sz = min(kVxSeg, nz - zs), sy = min(kVxSeg, ny - ys), sx = min(kVxSeg, nx - xs);
        __syncthreads();                                             // the previous segment is done with the matrices
        if (tid < 64) build_axis(Wx, rx, tabx, 0, PW, x0 + xs, sx, tid);
        else if (tid < 128) build_axis(Wy, ry, taby, 0, PH, y0 + ys, sy, tid - 64);
        else if (tid < 192) build_axis(Wz, rz, tabz, pd0, npd, z0 + zs, sz, tid - 128);
        __syncthreads();
        const int nvox = sz * sy * sx;
        for (int vb = part * (kClThreads / 16); vb < nvox; vb += parts * (kClThreads / 16)) {      // 16 voxels per pass, 4 per wave
          const int v = vb + slot;
          const bool live = v < nvox;
          const int vv = live ? v : 0;
          const int x = vv % sx, y = (vv / sx) % sy, z = vv / (sx * sy);
          int2 bz = rz[z], by = ry[y], bx = rx[x];
          if (!live) bz.y = bz.x - 1;                                   // empty range
          v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
          for (int pd = bz.x; pd <= bz.y; ++pd) {
            const float wz = Wz[z * npd + pd];
            for (int ph = by.x; ph <= by.y; ++ph) {
              const float wzy = wz * Wy[y * PH + ph];
              const float4 *row = Gq + (size_t)((pd * PH + ph) * PW) * (kVxRow / 4);
              for (int pw = bx.x; pw <= bx.y; ++pw) {
                const float w = wzy * Wx[x * PW + pw];
                const float4 gv = row[pw * (kVxRow / 4)];
                a0 += w * (v2f){gv.x, gv.y};
                a1 += w * (v2f){gv.z, gv.w};
              }
            }
          }
          // flush: the wave's 4 voxels x 64 channels go through LDS so that one atomic instruction adds the 64 consecutive
          // channels of ONE voxel (the full-rate atomic shape; 16 lanes x 4 strided dwords ran at 0.4 of it)
          *reinterpret_cast<float4 *>(F + slot * 64 + q * 4) = make_float4(a0.x, a0.y, a1.x, a1.y);
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int vk = v - (slot & 3) + k;                       // voxel of the wave's k-th slot in this pass
            if (vk < nvox) {
              const int xk = vk % sx, yk = (vk / sx) % sy, zk = vk / (sx * sy);
              const float val = F[(wave * 4 + k) * 64 + lane];
              if (val != 0.f)
                atomicAdd(bottom + (size_t)g.b * D * H * W * C + ((((size_t)(z0 + zs + zk)) * H + (y0 + ys + yk)) * W + (x0 + xs + xk)) * C + c0 + lane, val);
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
}

// ------------------------------------------------------------------------------------------
// channels-last BACKWARD, one WAVE per (RoI, 64 channels, pooled depth, chunk of pooled rows) (round 4): the transposed form of
// roi_align3d_fwd_wave.  The chunk's top gradients G[ph][pw] sit in the lane's registers (lane = channel; read through an LDS tile
// so that [R,C,PD,PH,PW] rows are fetched in runs of consecutive bins); per voxel row (z, y) of the footprint
//     T1[pw] = wz * sum_ph Wy[ph][y] G[ph][pw],      dF[z][y][x] += sum over the x-samples' taps on x of their weight * T1[bin]
// is accumulated in a wave-private LDS row [x][channel] and flushed with ONE 256-byte-contiguous atomic per touched voxel (the
// full-rate atomic shape).  No workgroup barrier after the set-up, no memory round trip inside the loop; weights, indices and loop
// bounds are wave-uniform.  Same taps and weights as the scatter form (l / h of axis_tap, dropped samples contribute nothing);
// the reference's top-gradient index bug (:554-555) behind `bug_compat` as everywhere.
// ------------------------------------------------------------------------------------------
template <int PDC, int PHC, int PWT, int NS>
__global__ __launch_bounds__(MR3D_WAVE) void roi_align3d_bwd_wave(const float *__restrict__ top, float *bottom, const float *__restrict__ rois,
                                                                  float scale, float scale_d, int C, int D, int H, int W, int PD, int PH,
                                                                  int bug_compat, LevelTable lt) {
  constexpr int NXS = PWT * NS, NYS = PHC * NS, NZS = PDC * NS;
  constexpr int TILE = PHC * PWT * (MR3D_WAVE + 1), BUF = kWvSlots * MR3D_WAVE;
  __shared__ IdxEntry tabx[NXS], taby[NYS], tabz[NZS];
  __shared__ __attribute__((aligned(16))) float buf[TILE > BUF ? TILE : BUF];       // the G tile, then the voxel row
  const int lane = threadIdx.x;
  const int r = blockIdx.x, c0 = blockIdx.y * MR3D_WAVE;
  const int phchunks = (PH + PHC - 1) / PHC;
  const int pd0 = ((int)blockIdx.z / phchunks) * PDC, ph0 = ((int)blockIdx.z % phchunks) * PHC;
  const int npd = min(PDC, PD - pd0), nph = min(PHC, PH - ph0);        // (scalar)
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    bottom = pick4(lvl, lt.grad);
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, NS, PD, PH, PWT);
  {
    const int e = lane;
    AxisTap t;
    IdxEntry te;
    te.lo = -1; te.hi = -1; te.l = 0.f; te.h = 0.f;
    if (e < NXS) {
      if (axis_tap(sample_coord(g.sw, e / NS, g.bw, e % NS, NS), W, t)) { te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h; }
      tabx[e] = te;
    } else if (e < NXS + NYS) {
      const int k = e - NXS;
      if (k / NS < nph && axis_tap(sample_coord(g.sh, ph0 + k / NS, g.bh, k % NS, NS), H, t)) { te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h; }
      taby[k] = te;
    } else if (e < NXS + NYS + NZS) {
      const int k = e - NXS - NYS;
      if (k / NS < npd && axis_tap(sample_coord(g.sd, pd0 + k / NS, g.bd, k % NS, NS), D, t)) { te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h; }
      tabz[k] = te;
    }
  }
  static_assert(NXS + NYS + NZS <= MR3D_WAVE, "one sample per lane");
  // ---- the chunk's top gradients: tile[bin][channel] <- 64 channel rows of nph * PWT consecutive bins, then one register per bin
  const int B = PD * PH * PWT;
  const int nb = nph * PWT;
  float (*tile)[MR3D_WAVE + 1] = reinterpret_cast<float (*)[MR3D_WAVE + 1]>(buf);
  const float inv_count = 1.f / (float)(NS * NS * NS);
  float gq[PDC][PHC * PWT];
#pragma unroll
  for (int a = 0; a < PDC; ++a) {
    const int pd = pd0 + a;
    __syncthreads();                                                   // tables written / the previous depth's tile read out
    if (a < npd && lane < nb) {
      const float *src = top + ((size_t)r * C + c0) * B + (bug_compat ? pd * PD * PWT : pd * PH * PWT) + ph0 * PWT;
#pragma unroll 8
      for (int ch = 0; ch < MR3D_WAVE; ++ch) tile[lane][ch] = src[(size_t)ch * B + lane];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < PHC * PWT; ++b) gq[a][b] = (a < npd && b < nb) ? tile[b][lane] * inv_count : 0.f;
  }
  auto sf = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  int x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1, z0 = 0x7fffffff, z1 = -1;
#pragma unroll
  for (int e = 0; e < NXS; ++e) {
    const int lo = __builtin_amdgcn_readfirstlane(tabx[e].lo), hi = __builtin_amdgcn_readfirstlane(tabx[e].hi);
    if (lo >= 0) { x0 = min(x0, lo); x1 = max(x1, hi); }
  }
#pragma unroll
  for (int e = 0; e < NYS; ++e) {
    const int lo = __builtin_amdgcn_readfirstlane(taby[e].lo), hi = __builtin_amdgcn_readfirstlane(taby[e].hi);
    if (lo >= 0) { y0 = min(y0, lo); y1 = max(y1, hi); }
  }
#pragma unroll
  for (int e = 0; e < NZS; ++e) {
    const int lo = __builtin_amdgcn_readfirstlane(tabz[e].lo), hi = __builtin_amdgcn_readfirstlane(tabz[e].hi);
    if (lo >= 0) { z0 = min(z0, lo); z1 = max(z1, hi); }
  }
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
  if (nx <= 0 || ny <= 0 || z1 < z0) return;                           // nothing of this chunk lands in the map (uniform)
  constexpr int YT = 64, PHP = 8;
  static_assert(PHC <= PHP, "row weights");
  __shared__ __attribute__((aligned(16))) float wyt[YT][PHP];
  auto row_weight = [&](int y, int a) {
    float w = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const IdxEntry t = taby[a * NS + i];
      if (t.lo == y) w += t.h;
      if (t.hi == y) w += t.l;
    }
    return w;
  };
  for (int idx = lane; idx < min(ny, YT) * PHP; idx += MR3D_WAVE) {
    const int row = idx / PHP, a = idx - row * PHP;
    wyt[row][a] = a < PHC ? row_weight(y0 + row, a) : 0.f;
  }
  __syncthreads();                                                     // (also: every lane has read its G values out of the tile)

  // dense x weights Wx[x - x0][pw] of a 64-voxel piece of the footprint in LDS (lane = voxel; a clamped sample has lo == hi: both
  // taps count): a voxel's gradient is then PWT uniform-weight FMAs on T1 -- an LDS row accumulated tap by tap was a chain of 56
  // dependent read-modify-writes per voxel row (203 us on the mask extractor)
  constexpr int PWP = (PWT + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float wxt[kWvSlots][PWP];
  float *gbase = bottom + (size_t)g.b * D * H * W * C + c0 + lane;
  for (int xs = 0; xs < nx; xs += kWvSlots) {                          // (one piece unless the footprint is wider than 64 voxels)
    const int sx = min(kWvSlots, nx - xs);
    __syncthreads();
    if (lane < sx) {
      const int x = x0 + xs + lane;
#pragma unroll
      for (int pw = 0; pw < PWP; ++pw) {
        float w = 0.f;
        if (pw < PWT) {
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            const IdxEntry t = tabx[pw * NS + i];
            if (t.lo == x) w += t.h;
            if (t.hi == x) w += t.l;
          }
        }
        wxt[lane][pw] = w;
      }
    }
    __syncthreads();
    for (int z = z0; z <= z1; ++z) {
      float wz[PDC];
      bool zany = false;
#pragma unroll
      for (int a = 0; a < PDC; ++a) {
        float w = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const IdxEntry t = tabz[a * NS + i];
          if (t.lo == z) w += t.h;
          if (t.hi == z) w += t.l;
        }
        wz[a] = sf(w);
        zany = zany || wz[a] != 0.f;
      }
      if (!zany) continue;
      for (int y = y0; y <= y1; ++y) {
        float wy[PHC];
        if (y - y0 < YT) {
          const float4 q0 = *reinterpret_cast<const float4 *>(&wyt[y - y0][0]), q1 = *reinterpret_cast<const float4 *>(&wyt[y - y0][4]);
          const float q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
          for (int a = 0; a < PHC; ++a) wy[a] = sf(q[a]);
        } else {
#pragma unroll
          for (int a = 0; a < PHC; ++a) wy[a] = sf(row_weight(y, a));
        }
        bool any = false;
#pragma unroll
        for (int a = 0; a < PHC; ++a) any = any || wy[a] != 0.f;
        if (!any) continue;
        float t1[PWP];
#pragma unroll
        for (int pw = 0; pw < PWP; ++pw) t1[pw] = 0.f;
#pragma unroll
        for (int d = 0; d < PDC; ++d) {
          if (wz[d] == 0.f) continue;
#pragma unroll
          for (int a = 0; a < PHC; ++a) {
            if (wy[a] == 0.f) continue;
            const float w = wz[d] * wy[a];
#pragma unroll
            for (int pw = 0; pw < PWT; ++pw) t1[pw] += w * gq[d][a * PWT + pw];
          }
        }
        float *grow = gbase + (((size_t)z * H + y) * W + x0 + xs) * C;
        for (int i = 0; i < sx; ++i) {
          float val = 0.f;
#pragma unroll
          for (int q4 = 0; q4 < PWP / 4; ++q4) {
            const float4 w4 = *reinterpret_cast<const float4 *>(&wxt[i][q4 * 4]);          // (uniform address: broadcast)
            val += w4.x * t1[q4 * 4] + w4.y * t1[q4 * 4 + 1] + w4.z * t1[q4 * 4 + 2] + w4.w * t1[q4 * 4 + 3];
          }
          if (val != 0.f) atomicAdd(grow + (size_t)i * C, val);
        }
      }
    }
  }
}

template <int PDC, int PHC, int PWT, int NS>
int launch_bwd_wave(const float *top, float *bottom, const float *rois, int num_rois, float scale, float scale_d, int C, int D, int H, int W,
                    int PD, int PH, int bug, const LevelTable &lt, hipStream_t stream) {
  const dim3 grid(num_rois, C / MR3D_WAVE, ((PD + PDC - 1) / PDC) * ((PH + PHC - 1) / PHC));
  roi_align3d_bwd_wave<PDC, PHC, PWT, NS><<<grid, MR3D_WAVE, 0, stream>>>(top, bottom, rois, scale, scale_d, C, D, H, W, PD, PH, bug, lt);
  return mr3d_launch_status();
}

// ------------------------------------------------------------------------------------------
// channels-last BACKWARD, one workgroup per (RoI, 64 channels, every K-th voxel PLANE of the footprint) (round 6).
// roi_align3d_bwd_wave adds a voxel once per (pooled-depth pair, pooled-row quad) chunk that reaches it: 2.4 atomic passes over
// the footprint on the mask extractor (90 MB of 256-byte atomics at the chip's ~1 TB/s atomic rate = the launch).  Here a voxel is
// added ONCE per RoI: per voxel plane z
//     U[ph][pw]  = sum_pd Wz[pd][z] G[pd][ph][pw]          (the pooled depths that reach z: a ballot over the 2 taps x NS samples)
//     V[pw]      = sum_ph Wy[ph][y] U[ph][pw]               per voxel row y of the plane
//     dF[z][y][x] += sum_pw Wx[pw][x] V[pw]                 one 256-byte atomic per voxel (lane = channel)
// The top gradient is [R, C, PD, PH, PW]: the 256 threads read the (64 channels x PH x PW) block of a pooled depth as 8-byte
// pieces of consecutive bins, accumulate U in registers over the depths of the plane and transpose it through LDS ONCE per plane
// (U[channel][ph][PWP], channel rows of RS floats with RS / 4 odd: the 16 lanes of a ds_read_b128 pass land on 64 distinct banks);
// the four waves then split the plane's rows, V in registers, Wy / Wx from dense LDS tables (uniform addresses: broadcast reads).
// Same taps and weights as every other form (axis_tap's l / h, dropped samples contribute nothing, a clamped sample counts both
// taps); the reference's top-gradient index bug (:554-555) behind `bug_compat`.
// ------------------------------------------------------------------------------------------
constexpr int kPlThreads = 256, kPlPdMax = 16, kPlYCap = 128, kPlXCap = 128;

template <int PH_, int PW_, int NS>
struct PlCfg {
  static constexpr int PWP = PW_ <= 8 ? 8 : 16, PHP = PH_ <= 8 ? 8 : 16;
  static constexpr int RS0 = PH_ * PWP;
  static constexpr int RS = ((RS0 / 4) % 2 == 0) ? RS0 + 4 : RS0;
  static constexpr int VEC = PW_ % 2 == 0 ? 2 : 1;
  static constexpr int ROWI = PW_ / VEC;                       // pieces per pooled row
  static constexpr int CHI = PH_ * ROWI;                       // pieces per channel
  static constexpr int ITEMS = 64 * CHI;
  static constexpr int PER = (ITEMS + kPlThreads - 1) / kPlThreads;
  static constexpr int NXS = PW_ * NS, NYS = PH_ * NS, NZS = kPlPdMax * NS;
  static constexpr int FLOATS = 64 * RS + kPlYCap * PHP + kPlXCap * PWP;
  static constexpr int LDS_BYTES = FLOATS * 4 + (NXS + NYS + NZS) * 16 + 32;
  static_assert(PW_ <= 16 && PH_ <= 16, "padded table rows");
};

__device__ __forceinline__ float pl_axis_weight(const IdxEntry *tab, int bin, int v, int ns) {
  float w = 0.f;
  for (int i = 0; i < ns; ++i) {
    const IdxEntry t = tab[bin * ns + i];
    if (t.lo == v) w += t.h;
    if (t.hi == v) w += t.l;
  }
  return w;
}

template <int PH_, int PW_, int NS>
__global__ __launch_bounds__(kPlThreads) void roi_align3d_bwd_planes(const float *__restrict__ top, float *bottom, const float *__restrict__ rois,
                                                                    float scale, float scale_d, int C, int D, int H, int W, int PD, int K,
                                                                    int bug_compat, LevelTable lt) {
  using S = PlCfg<PH_, PW_, NS>;
  constexpr int PWP = S::PWP, PHP = S::PHP, RS = S::RS, VEC = S::VEC, NXS = S::NXS, NYS = S::NYS, PER = S::PER;
  extern __shared__ __attribute__((aligned(16))) float pl_lds[];
  float *U = pl_lds;                                           // [64 channels][RS]
  float *wyt = U + 64 * RS;                                    // [kPlYCap][PHP]
  float *wxt = wyt + kPlYCap * PHP;                            // [kPlXCap][PWP]
  IdxEntry *tabx = reinterpret_cast<IdxEntry *>(wxt + kPlXCap * PWP), *taby = tabx + NXS, *tabz = taby + NYS;
  int *ext = reinterpret_cast<int *>(tabz + S::NZS);           // x0, x1, y0, y1, z0, z1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = blockIdx.x, c0 = blockIdx.y * 64, k = blockIdx.z;
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    bottom = pick4(lvl, lt.grad);
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, NS, PD, PH_, PW_);
  if (tid < 6) ext[tid] = (tid & 1) ? -1 : 0x7fffffff;
  __syncthreads();
  if (tid < NXS + NYS + PD * NS) {
    const int axis = tid < NXS ? 0 : (tid < NXS + NYS ? 1 : 2);
    const int e = tid - (axis == 0 ? 0 : (axis == 1 ? NXS : NXS + NYS));
    const float v = axis == 0 ? sample_coord(g.sw, e / NS, g.bw, e % NS, NS)
                              : (axis == 1 ? sample_coord(g.sh, e / NS, g.bh, e % NS, NS) : sample_coord(g.sd, e / NS, g.bd, e % NS, NS));
    AxisTap t;
    IdxEntry te;
    te.lo = -1; te.hi = -1; te.l = 0.f; te.h = 0.f;
    if (axis_tap(v, axis == 0 ? W : (axis == 1 ? H : D), t)) {
      te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h;
      atomicMin(&ext[2 * axis], t.lo);
      atomicMax(&ext[2 * axis + 1], t.hi);
    }
    (axis == 0 ? tabx : (axis == 1 ? taby : tabz))[e] = te;
  }
  __syncthreads();
  const int x0 = ext[0], x1 = ext[1], y0 = ext[2], y1 = ext[3], z0 = ext[4], z1 = ext[5];
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
  if (nx <= 0 || ny <= 0 || z0 + k > z1) return;              // nothing lands in the map / no plane for this workgroup (uniform)
  for (int idx = tid; idx < min(ny, kPlYCap) * PHP; idx += kPlThreads) {
    const int row = idx / PHP, a = idx - row * PHP;
    wyt[idx] = a < PH_ ? pl_axis_weight(taby, a, y0 + row, NS) : 0.f;
  }
  for (int idx = tid; idx < min(nx, kPlXCap) * PWP; idx += kPlThreads) {
    const int col = idx / PWP, a = idx - col * PWP;
    wxt[idx] = a < PW_ ? pl_axis_weight(tabx, a, x0 + col, NS) : 0.f;
  }
  // (the first plane's barrier below also covers the tables)
  const int B = PD * PH_ * PW_;
  const float inv_count = 1.f / (float)(NS * NS * NS);
  const float *tbase = top + ((size_t)r * C + c0) * B;
  const int pd_stride = bug_compat ? PD * PW_ : PH_ * PW_;
  float *gbase = bottom + (size_t)g.b * D * H * W * C + c0 + lane;
  auto sf = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };

  for (int z = z0 + k; z <= z1; z += K) {
    const float wzl = lane < PD ? pl_axis_weight(tabz, lane, z, NS) * inv_count : 0.f;
    unsigned long long mz = __ballot(wzl != 0.f);
    float acc[PER][VEC];
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[i][v] = 0.f;
    while (mz) {
      const int pd = __builtin_ctzll(mz);
      mz &= mz - 1;
      const float w = sf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wzl), pd)));
      const float *src = tbase + pd * pd_stride;
      float val[PER][VEC];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int item = tid + i * kPlThreads;
        const int c = item / S::CHI, rem = item - c * S::CHI;
        const bool on = (PER * kPlThreads == S::ITEMS) || item < S::ITEMS;
        const float *p = src + (size_t)c * B + rem * VEC;
        if (VEC == 2) {
          const float2 q = on ? *reinterpret_cast<const float2 *>(p) : make_float2(0.f, 0.f);
          val[i][0] = q.x; val[i][VEC - 1] = q.y;
        } else {
          val[i][0] = on ? *p : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < PER; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[i][v] += w * val[i][v];
    }
    __syncthreads();                                           // the previous plane's rows are done with U (first plane: the tables)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int item = tid + i * kPlThreads;
      const int c = item / S::CHI, rem = item - c * S::CHI;
      const int ph = rem / S::ROWI, j = rem - ph * S::ROWI;
      if ((PER * kPlThreads == S::ITEMS) || item < S::ITEMS) {
        float *d = U + c * RS + ph * PWP + j * VEC;
        if (VEC == 2) *reinterpret_cast<float2 *>(d) = make_float2(acc[i][0], acc[i][VEC - 1]);
        else d[0] = acc[i][0];
      }
    }
    __syncthreads();
    for (int y = y0 + wave; y <= y1; y += kPlThreads / 64) {
      const int yi = y - y0;
      float wyl = 0.f;
      if (yi < kPlYCap) { if (lane < PHP) wyl = wyt[yi * PHP + lane]; }
      else if (lane < PH_) wyl = pl_axis_weight(taby, lane, y, NS);
      unsigned long long my = __ballot(wyl != 0.f);
      if (!my) continue;
      float V[PWP];
#pragma unroll
      for (int pw = 0; pw < PWP; ++pw) V[pw] = 0.f;
      while (my) {
        const int a = __builtin_ctzll(my);
        my &= my - 1;
        const float w = sf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyl), a)));
        const float4 *u = reinterpret_cast<const float4 *>(U + lane * RS + a * PWP);
#pragma unroll
        for (int q = 0; q < PWP / 4; ++q) {
          const float4 v4 = u[q];
          V[q * 4] += w * v4.x; V[q * 4 + 1] += w * v4.y; V[q * 4 + 2] += w * v4.z; V[q * 4 + 3] += w * v4.w;
        }
      }
      float *grow = gbase + (((size_t)z * H + y) * W + x0) * C;
#pragma unroll 2
      for (int i = 0; i < nx; ++i) {
        float val = 0.f;
        if (i < kPlXCap) {
#pragma unroll
          for (int q = 0; q < PWP / 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wxt + i * PWP + q * 4);          // (uniform address: broadcast)
            if (q * 4 < PW_) val += w4.x * V[q * 4];
            if (q * 4 + 1 < PW_) val += w4.y * V[q * 4 + 1];
            if (q * 4 + 2 < PW_) val += w4.z * V[q * 4 + 2];
            if (q * 4 + 3 < PW_) val += w4.w * V[q * 4 + 3];
          }
        } else {
#pragma unroll
          for (int pw = 0; pw < PW_; ++pw) val += pl_axis_weight(tabx, pw, x0 + i, NS) * V[pw];
        }
        if (val != 0.f) atomicAdd(grow + (size_t)i * C, val);
      }
    }
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: one attempt per (kernel, device), up to 16
// devices in one process (a 17th takes the attempt on every call)
template <class K>
bool raise_lds_limit(K kernel, int bytes) {
  static std::mutex mu;
  static signed char state[16] = {};                           // 0 not tried, 1 set, -1 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  std::lock_guard<std::mutex> lock(mu);
  if (dev >= 0 && dev < 16 && state[dev]) return state[dev] > 0;
  const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  if (dev >= 0 && dev < 16) state[dev] = ok ? 1 : -1;
  return ok;
}

template <int PH_, int PW_, int NS>
int launch_bwd_planes(const float *top, float *bottom, const float *rois, int num_rois, float scale, float scale_d, int C, int D, int H, int W,
                      int PD, int K, int bug, const LevelTable &lt, hipStream_t stream) {
  using S = PlCfg<PH_, PW_, NS>;
  if (!raise_lds_limit(roi_align3d_bwd_planes<PH_, PW_, NS>, S::LDS_BYTES)) return MR3D_ERR_LAUNCH;
  const dim3 grid(num_rois, C / 64, K);
  roi_align3d_bwd_planes<PH_, PW_, NS><<<grid, kPlThreads, S::LDS_BYTES, stream>>>(top, bottom, rois, scale, scale_d, C, D, H, W, PD, K, bug, lt);
  return mr3d_launch_status();
}

// ------------------------------------------------------------------------------------------
// channels-last FORWARD, planes folded first (round 6): one workgroup per (RoI, 64 channels, pooled depth), lane = channel.
// The transposed form of roi_align3d_bwd_planes: every voxel row of the footprint is multiplied ONCE.
//   phase A (the waves split the voxel rows y):   F'[x]  = sum_z Wz[z] F[z][y][x]        z over the <= 2 NS planes of the pooled depth;
//                                                 V[pw]  = sum_x Wx[pw][x] F'[x]         (Wx: dense LDS table, uniform address = broadcast)
//                                                 -> LDS  Vt[y][channel][pw]
//   phase B (wave w owns pooled rows 4w .. 4w+3): T[ph][pw] = sum over the 2 NS taps of ph  weight * Vt[tap row][pw]     (registers)
//   output: registers -> LDS [channel][ph][PWP] -> 8-byte pieces of consecutive bins of the [R, C, PD, PH, PW] output.
// The voxels of a row piece (<= 16 voxels x <= 4 planes) are requested together through one buffer descriptor per (plane, piece):
// a voxel behind the row's end is out of the descriptor's range and reads 0 -- no branch, no select.  Footprints of more than
// kFoldRows voxel rows take several passes (the accumulators stay in registers); the sum order is fixed by the RoI alone.
// The row-streaming kernel above multiplies every voxel row of every plane (two MFMA products with padded K); the owner-wave
// form read a row once per wave whose pooled rows reach it.
// Phase stamps (tools/diag/roi_fold_stamps.py, SURVEY 8(d) rois; median / max per workgroup): roi + planes 1.7, samples 0.3, x table 0.6,
// phase A 2.9 / 16.4, phase B 1.2 (mask 2.4), tile + stores 1.3 (mask 2.1) -- lifetime 8.5 / 21.6 us (mask 10.2 / 24.6): a wave issues one
// instruction per ~4.5 cycles and there are two or three waves per SIMD.  Measured and not kept: two row pieces per step when the
// pooled depth has two planes (half the memory round trips of the mask extractor: phase A 2.68 -> 2.58 us -- the step is issue-bound,
// 64 load slots + fold + 11 instructions per voxel -- and 164 -> 239 registers); the four taps of a pooled row read together in
// phase B (2.36 -> 2.48 us); the voxels of the NEXT row piece requested before the current one is multiplied (two register sets of
// 4 planes x 8 voxels, ping-pong: phase A 2.9 -> 2.96 us median, 16.4 -> 15.4 max; launch 33.9 -> 34.5 / 51 -> 56 us); for the
// (.,14,14) shape three workgroups per CU instead of two (46 KB of LDS: Vt rows of 16 floats with the quads of a channel rotated
// by channel / 4, the output tile in two halves of eight pooled rows, a 64-column weight table): 53.8 -> 51.2 us, roi_align3d_fwd_rows 49.9.
// ------------------------------------------------------------------------------------------

template <int PH_, int PW_, int NS, int ROWS>
struct FoldCfg {
  static constexpr int kFoldRows = ROWS;
  using P = PlCfg<PH_, PW_, NS>;
  static constexpr int PWP = P::PWP;
  static constexpr int VP = ((PWP / 4) % 2 == 0) ? PWP + 4 : PWP;              // floats per (row, channel) of Vt: VP / 4 odd
  static constexpr int UNION = (kFoldRows * 64 * VP > 64 * P::RS) ? kFoldRows * 64 * VP : 64 * P::RS;
  static constexpr int FLOATS = UNION + kPlXCap * PWP;
  static constexpr int LDS_BYTES = FLOATS * 4 + (P::NXS + P::NYS) * 16 + 64;
};

template <int PH_, int PW_, int NS, int ROWS>
__global__ __launch_bounds__(kPlThreads, (PH_ > 8 ? 2 : 4)) void roi_align3d_fwd_fold(const float *feat, float *__restrict__ out,
                                                                                     const float *__restrict__ rois, float scale, float scale_d,
                                                                                     int C, int D, int H, int W, int PD, LevelTable lt) {
  using S = PlCfg<PH_, PW_, NS>;
  using Fc = FoldCfg<PH_, PW_, NS, ROWS>;
  constexpr int kFoldRows = ROWS;
  constexpr int PWP = S::PWP, RS = S::RS, VEC = S::VEC, NXS = S::NXS, NYS = S::NYS, PER = S::PER, VP = Fc::VP;
  constexpr int PHW = (PH_ + 3) / 4, NZ = 2 * NS;
  extern __shared__ __attribute__((aligned(16))) float pl_lds[];
  float *U = pl_lds;                                           // [64 channels][RS]: the output tile ...
  float *Vt = pl_lds;                                          // ... and before it [kFoldRows][64 channels][VP]
  float *wxt = pl_lds + Fc::UNION;                             // [kPlXCap][PWP]
  IdxEntry *tabx = reinterpret_cast<IdxEntry *>(wxt + kPlXCap * PWP), *taby = tabx + NXS;
  int *ext = reinterpret_cast<int *>(taby + NYS);              // x0, x1, y0, y1, -, -, planes | plane index x4 | plane weight x4
  int *zl = ext + 8;
  float *wzl = reinterpret_cast<float *>(ext + 12);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = blockIdx.x, c0 = blockIdx.y * 64, pd = blockIdx.z;
  FOLD_MARK(0);
  if (lt.num) {
    const int lvl = roi_level(rois + (size_t)r * 7, lt.finest_scale, lt.num);
    D = pick4(lvl, lt.D); H = pick4(lvl, lt.H); W = pick4(lvl, lt.W);
    scale = pick4(lvl, lt.scale); scale_d = pick4(lvl, lt.scale_d);
    feat = pick4(lvl, lt.feat);
  }
  const RoiGeom g = roi_geom(rois + (size_t)r * 7, scale, scale_d, NS, PD, PH_, PW_);
  if (tid < 6) ext[tid] = (tid & 1) ? -1 : 0x7fffffff;
  if (tid == 64) {                                             // the planes of this pooled depth: <= 2 per sample, merged
    AxisTap t[NS];
    bool ok[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) ok[s] = axis_tap(sample_coord(g.sd, pd, g.bd, s, NS), D, t[s]);
    const float inv_count = 1.f / (float)(NS * NS * NS);
    int n = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!ok[s]) continue;
        const int z = e ? t[s].hi : t[s].lo;
        bool dup = false;
        for (int q = 0; q < n; ++q) dup = dup || zl[q] == z;
        if (dup) continue;
        float w = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
          if (!ok[s2]) continue;
          if (t[s2].lo == z) w += t[s2].h;
          if (t[s2].hi == z) w += t[s2].l;
        }
        zl[n] = z; wzl[n] = w * inv_count;
        ++n;
      }
    for (int q = n; q < NZ; ++q) { zl[q] = 0; wzl[q] = 0.f; }
    ext[6] = n;
  }
  __syncthreads();
  FOLD_MARK(1);
  if (tid < NXS + NYS) {
    const int axis = tid < NXS ? 0 : 1;
    const int e = tid - (axis == 0 ? 0 : NXS);
    const float v = axis == 0 ? sample_coord(g.sw, e / NS, g.bw, e % NS, NS) : sample_coord(g.sh, e / NS, g.bh, e % NS, NS);
    AxisTap t;
    IdxEntry te;
    te.lo = -1; te.hi = -1; te.l = 0.f; te.h = 0.f;
    if (axis_tap(v, axis == 0 ? W : H, t)) {
      te.lo = t.lo; te.hi = t.hi; te.l = t.l; te.h = t.h;
      atomicMin(&ext[2 * axis], t.lo);
      atomicMax(&ext[2 * axis + 1], t.hi);
    }
    (axis == 0 ? tabx : taby)[e] = te;
  }
  __syncthreads();
  FOLD_MARK(2);
  const int x0 = __builtin_amdgcn_readfirstlane(ext[0]), y0 = __builtin_amdgcn_readfirstlane(ext[2]);
  const int nzp = __builtin_amdgcn_readfirstlane(ext[6]);
  const int nx = __builtin_amdgcn_readfirstlane(ext[1]) - x0 + 1, ny = __builtin_amdgcn_readfirstlane(ext[3]) - y0 + 1;
  const float *fbase = feat + (size_t)g.b * D * H * W * C + c0;
  const int ph0 = wave * PHW;
  const int nphw = min(PHW, PH_ - ph0);
  auto sf = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };

  if (nx > kPlXCap) {
    // a footprint wider than the weight table (a RoI of > 128 voxels along x of its level): bin by bin, tap by tap
    for (int bin = wave; bin < PH_ * PW_; bin += kPlThreads / 64) {
      const int ph = bin / PW_, pw = bin - ph * PW_;
      float acc = 0.f;
      for (int k = 0; k < nzp; ++k) {
        const int z = zl[k];
        for (int sy = 0; sy < 2 * NS; ++sy) {
          const IdxEntry ty = taby[ph * NS + (sy >> 1)];
          if (ty.lo < 0) continue;
          const float wzy = wzl[k] * ((sy & 1) ? ty.l : ty.h);
          const float *prow = fbase + ((size_t)z * H + ((sy & 1) ? ty.hi : ty.lo)) * W * C + lane;
          for (int sx = 0; sx < 2 * NS; ++sx) {
            const IdxEntry tx = tabx[pw * NS + (sx >> 1)];
            if (tx.lo < 0) continue;
            acc += wzy * ((sx & 1) ? tx.l : tx.h) * prow[(size_t)((sx & 1) ? tx.hi : tx.lo) * C];
          }
        }
      }
      U[lane * RS + ph * PWP + pw] = acc;
    }
  } else {
    const int nxt = min((nx + 15) & ~15, kPlXCap);             // (the columns behind the row's end carry weight 0; kPlXCap % 16 == 0)
    for (int idx = tid; idx < nxt * PWP; idx += kPlThreads) {
      const int col = idx / PWP, a = idx - col * PWP;
      wxt[idx] = a < PW_ ? pl_axis_weight(tabx, a, x0 + col, NS) : 0.f;
    }
    float T[PHW][PW_];
#pragma unroll
    for (int j = 0; j < PHW; ++j)
#pragma unroll
      for (int pw = 0; pw < PW_; ++pw) T[j][pw] = 0.f;
    const unsigned cstep = (unsigned)C * 4u, lane4 = (unsigned)lane * 4u;
    const int zq[NZ] = {__builtin_amdgcn_readfirstlane(zl[0]), __builtin_amdgcn_readfirstlane(zl[1]), __builtin_amdgcn_readfirstlane(zl[NZ > 2 ? 2 : 0]),
                        __builtin_amdgcn_readfirstlane(zl[NZ > 3 ? 3 : 0])};
    const float wq[NZ] = {sf(wzl[0]), sf(wzl[1]), sf(wzl[NZ > 2 ? 2 : 0]), sf(wzl[NZ > 3 ? 3 : 0])};
    static_assert(NZ == 4, "two samples per bin along z");

    // phase A for one voxel row: V[pw] of the folded row, pieces of XB voxels
    auto row = [&](auto xb_tag, int y, float (&V)[PW_]) {
      constexpr int XB = decltype(xb_tag)::value;
      const int nch = (nx + XB - 1) / XB;
      for (int pc = 0; pc < nch; ++pc) {
        float f[NZ][XB];
        const int nvox = min(XB, nx - pc * XB);
#pragma unroll
        for (int k = 0; k < NZ; ++k) {
          const bool on = k < nzp;
          const float *pbase = fbase + (size_t)(((zq[k] * H + y) * W + x0 + pc * XB)) * C;
          const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(on ? pbase : fbase), 0, on ? nvox * (int)cstep : 0, 0x00020000);
#pragma unroll
          for (int i = 0; i < XB; ++i) f[k][i] = buffer_load1(pr, lane4, (unsigned)i * cstep);
        }
        __builtin_amdgcn_sched_barrier(0);                     // every request of the piece before the first multiply-add
        float ff[XB];
#pragma unroll
        for (int i = 0; i < XB; ++i) {
          ff[i] = wq[0] * f[0][i];
#pragma unroll
          for (int k = 1; k < NZ; ++k) ff[i] += wq[k] * f[k][i];
        }
        const float *wrow = wxt + pc * XB * PWP;
        // the weights of voxel i + 1 are requested before voxel i's multiply-adds and no further ahead (left alone the scheduler
        // hoists every read of the piece)
        float4 wc[PWP / 4], wn[PWP / 4];
#pragma unroll
        for (int q = 0; q < PWP / 4; ++q) wc[q] = *reinterpret_cast<const float4 *>(wrow + q * 4);      // (uniform address: broadcast)
#pragma unroll
        for (int i = 0; i < XB; ++i) {
          if (i + 1 < XB) {
#pragma unroll
            for (int q = 0; q < PWP / 4; ++q) wn[q] = *reinterpret_cast<const float4 *>(wrow + (i + 1) * PWP + q * 4);
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < PWP / 4; ++q) {
            if (q * 4 < PW_) V[q * 4] += wc[q].x * ff[i];
            if (q * 4 + 1 < PW_) V[q * 4 + 1] += wc[q].y * ff[i];
            if (q * 4 + 2 < PW_) V[q * 4 + 2] += wc[q].z * ff[i];
            if (q * 4 + 3 < PW_) V[q * 4 + 3] += wc[q].w * ff[i];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < PWP / 4; ++q) wc[q] = wn[q];
        }
      }
    };

    __syncthreads();                                           // the weight table
    FOLD_MARK(3);
    FOLD_NOTE(10, nx); FOLD_NOTE(11, ny); FOLD_NOTE(12, nzp);
    for (int yb = 0; yb < ny && nzp > 0; yb += kFoldRows) {
      const int nrows = min(kFoldRows, ny - yb);
      if (yb) __syncthreads();                                 // the previous pass's phase B is done with Vt
      for (int yr = wave; yr < nrows; yr += kPlThreads / 64) {
        float V[PW_];
#pragma unroll
        for (int pw = 0; pw < PW_; ++pw) V[pw] = 0.f;
        if (nx <= 4) row(std::integral_constant<int, 4>{}, y0 + yb + yr, V);
        else if (nx <= 8) row(std::integral_constant<int, 8>{}, y0 + yb + yr, V);
        else if (nx <= 12) row(std::integral_constant<int, 12>{}, y0 + yb + yr, V);
        else row(std::integral_constant<int, 16>{}, y0 + yb + yr, V);
        float *d = Vt + (yr * 64 + lane) * VP;
#pragma unroll
        for (int q = 0; q < PWP / 4; ++q) {
          float4 v;
          v.x = q * 4 < PW_ ? V[q * 4 < PW_ ? q * 4 : 0] : 0.f;
          v.y = q * 4 + 1 < PW_ ? V[q * 4 + 1 < PW_ ? q * 4 + 1 : 0] : 0.f;
          v.z = q * 4 + 2 < PW_ ? V[q * 4 + 2 < PW_ ? q * 4 + 2 : 0] : 0.f;
          v.w = q * 4 + 3 < PW_ ? V[q * 4 + 3 < PW_ ? q * 4 + 3 : 0] : 0.f;
          *reinterpret_cast<float4 *>(d + q * 4) = v;
        }
      }
      FOLD_MARK(4);
      __syncthreads();
      FOLD_MARK(5);
      // phase B: the 2 NS taps of each of the wave's pooled rows that fall into this pass
#pragma unroll
      for (int j = 0; j < PHW; ++j) {
        if (j < nphw) {
#pragma unroll
          for (int e = 0; e < 2 * NS; ++e) {
            const IdxEntry ty = taby[(ph0 + j) * NS + (e >> 1)];
            const int yy = __builtin_amdgcn_readfirstlane((e & 1) ? ty.hi : ty.lo) - y0 - yb;
            const float w = sf((e & 1) ? ty.l : ty.h);
            if (yy >= 0 && yy < nrows && w != 0.f) {           // (an out-of-range sample has lo = hi = -1)
              const float4 *s = reinterpret_cast<const float4 *>(Vt + (yy * 64 + lane) * VP);
#pragma unroll
              for (int q = 0; q < PWP / 4; ++q) {
                const float4 v4 = s[q];
                if (q * 4 < PW_) T[j][q * 4 < PW_ ? q * 4 : 0] += w * v4.x;
                if (q * 4 + 1 < PW_) T[j][q * 4 + 1 < PW_ ? q * 4 + 1 : 0] += w * v4.y;
                if (q * 4 + 2 < PW_) T[j][q * 4 + 2 < PW_ ? q * 4 + 2 : 0] += w * v4.z;
                if (q * 4 + 3 < PW_) T[j][q * 4 + 3 < PW_ ? q * 4 + 3 : 0] += w * v4.w;
              }
            }
          }
        }
      }
    }
    FOLD_MARK(6);
    __syncthreads();                                           // every wave is done with Vt: the output tile takes its place
#pragma unroll
    for (int j = 0; j < PHW; ++j) {
      if (j < nphw) {
        float *d = U + lane * RS + (ph0 + j) * PWP;
#pragma unroll
        for (int q = 0; q < PWP / 4; ++q) {
          float4 v;
          v.x = q * 4 < PW_ ? T[j][q * 4 < PW_ ? q * 4 : 0] : 0.f;
          v.y = q * 4 + 1 < PW_ ? T[j][q * 4 + 1 < PW_ ? q * 4 + 1 : 0] : 0.f;
          v.z = q * 4 + 2 < PW_ ? T[j][q * 4 + 2 < PW_ ? q * 4 + 2 : 0] : 0.f;
          v.w = q * 4 + 3 < PW_ ? T[j][q * 4 + 3 < PW_ ? q * 4 + 3 : 0] : 0.f;
          *reinterpret_cast<float4 *>(d + q * 4) = v;
        }
      }
    }
  }
  __syncthreads();
  FOLD_MARK(7);
  // ---- the output tile: LDS [channel][ph][PWP] -> 8-byte pieces of consecutive bins
  const int B = PD * PH_ * PW_;
  float *obase = out + ((size_t)r * C + c0) * B + pd * PH_ * PW_;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int item = tid + i * kPlThreads;
    if ((PER * kPlThreads == S::ITEMS) || item < S::ITEMS) {
      const int c = item / S::CHI, rem = item - c * S::CHI;
      const int ph = rem / S::ROWI, j = rem - ph * S::ROWI;
      const float *s = U + c * RS + ph * PWP + j * VEC;
      float *d = obase + (size_t)c * B + rem * VEC;
      if (VEC == 2) *reinterpret_cast<float2 *>(d) = *reinterpret_cast<const float2 *>(s);
      else d[0] = s[0];
    }
  }
  FOLD_MARK(8);
#ifdef MR3D_ROWS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FOLD_MARK(9);
#endif
}

template <int PH_, int PW_, int NS, int ROWS>
int launch_fwd_fold(const float *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int C, int D, int H, int W, int PD,
                    const LevelTable &lt, hipStream_t stream) {
  using Fc = FoldCfg<PH_, PW_, NS, ROWS>;
  if (!raise_lds_limit(roi_align3d_fwd_fold<PH_, PW_, NS, ROWS>, Fc::LDS_BYTES)) return MR3D_ERR_LAUNCH;
  const dim3 grid(num_rois, C / 64, PD);
  roi_align3d_fwd_fold<PH_, PW_, NS, ROWS><<<grid, kPlThreads, Fc::LDS_BYTES, stream>>>(feat, out, rois, scale, scale_d, C, D, H, W, PD, lt);
  return mr3d_launch_status();
}

int try_fwd_fold(const float *feat, float *out, const float *rois, int num_rois, float scale, float scale_d, int C, int D, int H, int W, int PD,
                 int PH, int PW, const LevelTable &lt, hipStream_t stream) {
  // voxel rows per pass (SURVEY 8(d) rois, (3,7,7) x 512, kernel time): 8 / 10 / 12 / 16 rows = 34.7 / 33.3 / 32.8 / 34.4 us with two
  // piece widths; 30.5 us with the four (12 rows)
  if (PW == 14 && PH == 14) return launch_fwd_fold<14, 14, 2, 12>(feat, out, rois, num_rois, scale, scale_d, C, D, H, W, PD, lt, stream);
  if (PW == 7 && PH == 7) return launch_fwd_fold<7, 7, 2, 12>(feat, out, rois, num_rois, scale, scale_d, C, D, H, W, PD, lt, stream);
  return MR3D_ERR_UNSUPPORTED;
}

int try_bwd_wave(const float *top, float *bottom, const float *rois, int num_rois, float scale, float scale_d, int ns, int C, int D, int H,
                 int W, int PD, int PH, int PW, int bug, const LevelTable &lt, hipStream_t stream) {
  if (C % MR3D_WAVE != 0 || ns != 2) return MR3D_ERR_UNSUPPORTED;
  if (bug && (PD - 1) * PD * PW + PH * PW > PD * PH * PW) return MR3D_ERR_UNSUPPORTED;      // the bug's offsets stay inside the roi's block
  // Mask extractor (10,14,14) x 128 rois on the SURVEY 8(d) inputs, launch alone: voxel-owner kernel 158 us; this kernel with
  // (pooled depths, pooled rows) per wave = (1,4) 123, (2,4) 89, (3,4) 98 (269 registers), (5,2) 113 -- fewer chunks mean fewer
  // atomic passes over a voxel plane that several pooled depths reach, until the registers halve the waves per SIMD.
  // The (3,7,7) extractor keeps the voxel-owner kernel (73 us against 107 us for (3,7) per wave: its rois include the footprints of
  // thousands of voxels that a single wave walks alone).
  // Plane-owner kernel, K = planes in flight per RoI (SURVEY 8(d) inputs, launch alone): mask extractor (10,14,14) x 128 rois
  // K = 1 / 2 / 4 / 8 / 12 / 16: 96.7 / 52.9 / 33.7 / 31.9 / 32.3 / 33.9 us (roi_align3d_bwd_wave<2,4,14,2>: 87.6); bbox extractor
  // (3,7,7) x 512 rois K = 2 / 8 / 12 / 16: 50.0 / 43.0 / 45.1 / 47.7 us (voxel-owner kernel: 74.0).  MR3D_TUNE_ROIALIGN_FORM = 1
  // keeps the round-5 kernels.  Splitting a RoI's pooled rows over 2 / 4 workgroups as well (each holds a slice of U and adds the voxel
  // rows its pooled rows reach; 30 KB of LDS, three per CU) was measured on the mask extractor: (slices, K) = (1,8) 36.4, (2,8) 34.8,
  // (2,4) 34.1, (4,4) 42.0, (4,2) 43.9 us on one box -- not kept.
  if (PD <= kPlPdMax && g_mr3d_tuning[MR3D_TUNE_ROIALIGN_FORM] != 1) {
    if (PW == 14 && PH == 14) return launch_bwd_planes<14, 14, 2>(top, bottom, rois, num_rois, scale, scale_d, C, D, H, W, PD, 8, bug, lt, stream);
    if (PW == 7 && PH == 7) return launch_bwd_planes<7, 7, 2>(top, bottom, rois, num_rois, scale, scale_d, C, D, H, W, PD, 8, bug, lt, stream);
  }
  if (PW == 14 && PH == 14) return launch_bwd_wave<2, 4, 14, 2>(top, bottom, rois, num_rois, scale, scale_d, C, D, H, W, PD, PH, bug, lt, stream);
  return MR3D_ERR_UNSUPPORTED;
}

// the voxel-owner backward for a pooled shape: MR3D_ERR_UNSUPPORTED = not applicable (the caller takes the plan + sub-box path)
int try_bwd_vox(const float *top, float *bottom, const float *rois, int num_rois, float scale, float scale_d, int ns, int C, int D,
                int H, int W, int PD, int PH, int PW, int bug, const LevelTable &lt, hipStream_t stream) {
  if (C % 64 != 0 || ns <= 0 || PD * ns > kStAxis || PH * ns > kStAxis || PW * ns > kStAxis) return MR3D_ERR_UNSUPPORTED;
  const int budget = 80 * 1024;               // two workgroups per CU
  int pdc = PD;
  while (pdc > 1 && vox_lds_bytes(pdc, PH, PW) > budget) --pdc;
  const int lds = vox_lds_bytes(pdc, PH, PW);
  if (lds > 160 * 1024) return MR3D_ERR_UNSUPPORTED;
  static std::mutex mu;
  static int attr_bytes = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (lds > attr_bytes) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(roi_align3d_bwd_vox), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        (void)hipGetLastError();
        return MR3D_ERR_LAUNCH;
      }
      attr_bytes = lds;
    }
  }
  const int chunks = (PD + pdc - 1) / pdc;
  const int parts = chunks == 1 ? 3 : 1;   // bbox extractor: 105 (1 part) -> 84 -> 76 us; mask extractor (10 chunks): 1
  dim3 grid(num_rois, C / 64, chunks * parts);
  roi_align3d_bwd_vox<<<grid, kClThreads, lds, stream>>>(top, bottom, rois, scale, scale_d, ns, C, D, H, W, PD, PH, PW, pdc, parts, bug, lt);
  return mr3d_launch_status();
}

int check_common(const void *a, const void *b, const void *c, int num_rois, int batch, int C, int D, int H, int W, int PD,
                 int PH, int PW, int sample_num, int layout) {
  if (num_rois < 0 || batch <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || PD <= 0 || PH <= 0 || PW <= 0 ||
      sample_num < 0)
    return MR3D_ERR_INVALID_ARG;
  if (layout != MR3D_LAYOUT_NCDHW && layout != MR3D_LAYOUT_NDHWC) return MR3D_ERR_INVALID_ARG;
  if (num_rois > 0 && (!a || !b || !c)) return MR3D_ERR_INVALID_ARG;
  return MR3D_OK;
}

Strides strides_for(int layout, int C, int D, int H, int W) {
  Strides s;
  if (layout == MR3D_LAYOUT_NCDHW) {
    s.w = 1; s.h = W; s.d = (long long)H * W; s.c = (long long)D * H * W; s.n = (long long)C * D * H * W;
  } else {
    s.c = 1; s.w = C; s.h = (long long)W * C; s.d = (long long)H * W * C; s.n = (long long)D * H * W * C;
  }
  return s;
}

bool use_cl(int layout, int sample_num, int PD, int PH, int PW) {
  return layout == MR3D_LAYOUT_NDHWC && sample_num > 0 && PD * sample_num <= kAxisCap && PH * sample_num <= kAxisCap &&
         PW * sample_num <= kAxisCap;
}

}  // namespace

extern "C" int mr3d_roi_align3d_forward(const float *features, const float *rois, int num_rois, int batch, int channels,
                                        int depth, int height, int width, int pooled_depth, int pooled_height,
                                        int pooled_width, float spatial_scale, float spatial_scale_depth, int sample_num,
                                        int layout, float *output, mr3d_stream_t stream_) {
  int st = check_common(features, rois, output, num_rois, batch, channels, depth, height, width, pooled_depth,
                        pooled_height, pooled_width, sample_num, layout);
  if (st != MR3D_OK) return st;
  if (num_rois == 0) return MR3D_OK;
  hipStream_t stream = mr3d_stream(stream_);
  const int B = pooled_depth * pooled_height * pooled_width;
  if (use_cl(layout, sample_num, pooled_depth, pooled_height, pooled_width)) {
    const int str = try_fwd_rows(features, output, rois, num_rois, spatial_scale, spatial_scale_depth, sample_num, channels, depth, height,
                                 width, pooled_depth, pooled_height, pooled_width, LevelTable{}, stream);
    if (str != MR3D_ERR_UNSUPPORTED) return str;
    const int st0 = try_fwd_wave(features, output, rois, num_rois, spatial_scale, spatial_scale_depth, sample_num, channels, depth, height,
                                 width, pooled_depth, pooled_height, pooled_width, LevelTable{}, stream);
    if (st0 != MR3D_ERR_UNSUPPORTED) return st0;
    const int st1 = try_fwd_box<float>(features, output, rois, num_rois, spatial_scale, spatial_scale_depth, sample_num, channels, depth,
                                         height, width, pooled_depth, pooled_height, pooled_width, LevelTable{}, stream);
    if (st1 != MR3D_ERR_UNSUPPORTED) return st1;
    const int st2 = try_fwd_sep<float>(features, output, rois, num_rois, spatial_scale, spatial_scale_depth, sample_num, channels, depth,
                                       height, width, pooled_depth, pooled_height, pooled_width, LevelTable{}, stream);
    if (st2 != MR3D_ERR_UNSUPPORTED) return st2;
  }
  if (use_cl(layout, sample_num, pooled_depth, pooled_height, pooled_width)) {
    dim3 grid(num_rois, (B + kClBins - 1) / kClBins);
    roi_align3d_cl<false><<<grid, kClThreads, 0, stream>>>(features, output, rois, spatial_scale, spatial_scale_depth,
                                                          sample_num, channels, depth, height, width, pooled_depth,
                                                          pooled_height, pooled_width, 0, LevelTable{});
  } else {
    const long long total = (long long)num_rois * channels * B;
    const long long blocks = (total + 255) / 256;
    const int grid = (int)(blocks < 65536 ? blocks : 65536);
    roi_align3d_fwd_generic<<<grid, 256, 0, stream>>>(total, features, rois, spatial_scale, spatial_scale_depth,
                                                      sample_num, channels, depth, height, width, pooled_depth,
                                                      pooled_height, pooled_width,
                                                      strides_for(layout, channels, depth, height, width), output);
  }
  return mr3d_launch_status();
}

extern "C" int mr3d_roi_align3d_backward(const float *top_grad, const float *rois, int num_rois, int batch, int channels,
                                         int depth, int height, int width, int pooled_depth, int pooled_height,
                                         int pooled_width, float spatial_scale, float spatial_scale_depth, int sample_num,
                                         int layout, int flags, float *bottom_grad, mr3d_stream_t stream_) {
  int st = check_common(top_grad, rois, bottom_grad, num_rois, batch, channels, depth, height, width, pooled_depth,
                        pooled_height, pooled_width, sample_num, layout);
  if (st != MR3D_OK) return st;
  if (num_rois == 0) return MR3D_OK;
  hipStream_t stream = mr3d_stream(stream_);
  const int bug = (flags & MR3D_ROIALIGN_BUG_COMPAT) ? 1 : 0;
  const int B = pooled_depth * pooled_height * pooled_width;
  if (use_cl(layout, sample_num, pooled_depth, pooled_height, pooled_width)) {
    const int stw = try_bwd_wave(top_grad, bottom_grad, rois, num_rois, spatial_scale, spatial_scale_depth, sample_num, channels, depth,
                                 height, width, pooled_depth, pooled_height, pooled_width, bug, LevelTable{}, stream);
    if (stw != MR3D_ERR_UNSUPPORTED) return stw;
    const int stv = try_bwd_vox(top_grad, bottom_grad, rois, num_rois, spatial_scale, spatial_scale_depth, sample_num, channels, depth,
                                height, width, pooled_depth, pooled_height, pooled_width, bug, LevelTable{}, stream);
    if (stv != MR3D_ERR_UNSUPPORTED) return stv;
    dim3 grid(num_rois, (B + kClBins - 1) / kClBins);
    // other channel counts: the wave-per-bin kernel scatters with 256-byte-contiguous fp32 atomics
    roi_align3d_cl<true><<<grid, kClThreads, 0, stream>>>(top_grad, bottom_grad, rois, spatial_scale,
                                                         spatial_scale_depth, sample_num, channels, depth, height, width,
                                                         pooled_depth, pooled_height, pooled_width, bug, LevelTable{});
  } else {
    const long long total = (long long)num_rois * channels * B;
    const long long blocks = (total + 255) / 256;
    const int grid = (int)(blocks < 65536 ? blocks : 65536);
    roi_align3d_bwd_generic<<<grid, 256, 0, stream>>>(total, top_grad, rois, spatial_scale, spatial_scale_depth,
                                                      sample_num, channels, depth, height, width, pooled_depth,
                                                      pooled_height, pooled_width,
                                                      strides_for(layout, channels, depth, height, width), bug,
                                                      bottom_grad);
  }
  return mr3d_launch_status();
}

extern "C" int mr3d_roi_align3d_forward_dtype(const void *features, const void *rois, int num_rois, int batch, int channels, int depth,
                                              int height, int width, int pooled_depth, int pooled_height, int pooled_width,
                                              float spatial_scale, float spatial_scale_depth, int sample_num, int layout, int dtype,
                                              void *output, mr3d_stream_t stream_) {
  if (dtype == MR3D_DTYPE_F32)
    return mr3d_roi_align3d_forward(static_cast<const float *>(features), static_cast<const float *>(rois), num_rois, batch, channels,
                                    depth, height, width, pooled_depth, pooled_height, pooled_width, spatial_scale,
                                    spatial_scale_depth, sample_num, layout, static_cast<float *>(output), stream_);
  if (dtype != MR3D_DTYPE_F16 && dtype != MR3D_DTYPE_F64) return MR3D_ERR_INVALID_ARG;
  const int st = check_common(features, rois, output, num_rois, batch, channels, depth, height, width, pooled_depth, pooled_height,
                              pooled_width, sample_num, layout);
  if (st != MR3D_OK) return st;
  if (num_rois == 0) return MR3D_OK;
  hipStream_t stream = mr3d_stream(stream_);
  const long long total = (long long)num_rois * channels * pooled_depth * pooled_height * pooled_width;
  const long long blocks = (total + 255) / 256;
  const int grid = (int)(blocks < 65536 ? blocks : 65536);
  const Strides sd = strides_for(layout, channels, depth, height, width);
  if (dtype == MR3D_DTYPE_F64)
    roi_align3d_typed<double, double, false><<<grid, 256, 0, stream>>>(total, static_cast<const double *>(features),
        static_cast<const double *>(rois), (double)spatial_scale, (double)spatial_scale_depth, sample_num, channels, depth, height, width,
        pooled_depth, pooled_height, pooled_width, sd, 0, static_cast<double *>(output));
  else
    roi_align3d_typed<__half, float, false><<<grid, 256, 0, stream>>>(total, static_cast<const __half *>(features),
        static_cast<const __half *>(rois), spatial_scale, spatial_scale_depth, sample_num, channels, depth, height, width,
        pooled_depth, pooled_height, pooled_width, sd, 0, static_cast<__half *>(output));
  return mr3d_launch_status();
}

extern "C" int mr3d_roi_align3d_backward_dtype(const void *top_grad, const void *rois, int num_rois, int batch, int channels, int depth,
                                               int height, int width, int pooled_depth, int pooled_height, int pooled_width,
                                               float spatial_scale, float spatial_scale_depth, int sample_num, int layout, int flags,
                                               int dtype, void *bottom_grad, mr3d_stream_t stream_) {
  if (dtype == MR3D_DTYPE_F32)
    return mr3d_roi_align3d_backward(static_cast<const float *>(top_grad), static_cast<const float *>(rois), num_rois, batch, channels,
                                     depth, height, width, pooled_depth, pooled_height, pooled_width, spatial_scale,
                                     spatial_scale_depth, sample_num, layout, flags, static_cast<float *>(bottom_grad), stream_);
  if (dtype != MR3D_DTYPE_F16 && dtype != MR3D_DTYPE_F64) return MR3D_ERR_INVALID_ARG;
  const int st = check_common(top_grad, rois, bottom_grad, num_rois, batch, channels, depth, height, width, pooled_depth,
                              pooled_height, pooled_width, sample_num, layout);
  if (st != MR3D_OK) return st;
  if (num_rois == 0) return MR3D_OK;
  hipStream_t stream = mr3d_stream(stream_);
  const int bug = (flags & MR3D_ROIALIGN_BUG_COMPAT) ? 1 : 0;
  const long long total = (long long)num_rois * channels * pooled_depth * pooled_height * pooled_width;
  const long long blocks = (total + 255) / 256;
  const int grid = (int)(blocks < 65536 ? blocks : 65536);
  const Strides sd = strides_for(layout, channels, depth, height, width);
  if (dtype == MR3D_DTYPE_F64)
    roi_align3d_typed<double, double, true><<<grid, 256, 0, stream>>>(total, static_cast<const double *>(top_grad),
        static_cast<const double *>(rois), (double)spatial_scale, (double)spatial_scale_depth, sample_num, channels, depth, height, width,
        pooled_depth, pooled_height, pooled_width, sd, bug, static_cast<double *>(bottom_grad));
  else
    roi_align3d_typed<__half, float, true><<<grid, 256, 0, stream>>>(total, static_cast<const __half *>(top_grad),
        static_cast<const __half *>(rois), spatial_scale, spatial_scale_depth, sample_num, channels, depth, height, width,
        pooled_depth, pooled_height, pooled_width, sd, bug, static_cast<__half *>(bottom_grad));
  return mr3d_launch_status();
}

namespace {
int fill_levels(const mr3d_roi_level *levels, int num_levels, float finest_scale, bool backward, LevelTable &lt) {
  if (!levels || num_levels <= 0 || num_levels > kMaxLevels || !(finest_scale > 0.f)) return MR3D_ERR_INVALID_ARG;
  lt = LevelTable{};
  lt.num = num_levels;
  lt.finest_scale = finest_scale;
  for (int i = 0; i < kMaxLevels; ++i) {
    const mr3d_roi_level &L = levels[i < num_levels ? i : num_levels - 1];
    if (L.depth <= 0 || L.height <= 0 || L.width <= 0) return MR3D_ERR_INVALID_ARG;
    if (backward ? !L.grad : !L.features) return MR3D_ERR_INVALID_ARG;
    lt.feat[i] = L.features; lt.grad[i] = L.grad;
    lt.D[i] = L.depth; lt.H[i] = L.height; lt.W[i] = L.width;
    lt.scale[i] = L.spatial_scale; lt.scale_d[i] = L.spatial_scale_depth;
  }
  return MR3D_OK;
}
}  // namespace

extern "C" int mr3d_roi_align3d_forward_ml(const mr3d_roi_level *levels_host, int num_levels, float finest_scale,
                                           const float *rois, int num_rois, int channels, int pooled_depth, int pooled_height,
                                           int pooled_width, int sample_num, float *output, mr3d_stream_t stream_) {
  LevelTable lt;
  const int st = fill_levels(levels_host, num_levels, finest_scale, false, lt);
  if (st != MR3D_OK) return st;
  if (num_rois < 0 || channels <= 0 || pooled_depth <= 0 || pooled_height <= 0 || pooled_width <= 0) return MR3D_ERR_INVALID_ARG;
  if (num_rois == 0) return MR3D_OK;
  if (!rois || !output) return MR3D_ERR_INVALID_ARG;
  if (!use_cl(MR3D_LAYOUT_NDHWC, sample_num, pooled_depth, pooled_height, pooled_width)) return MR3D_ERR_UNSUPPORTED;
  {
    const int str = try_fwd_rows(nullptr, output, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth, pooled_height,
                                 pooled_width, lt, mr3d_stream(stream_));
    if (str != MR3D_ERR_UNSUPPORTED) return str;
    const int st0 = try_fwd_wave(nullptr, output, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth, pooled_height,
                                 pooled_width, lt, mr3d_stream(stream_));
    if (st0 != MR3D_ERR_UNSUPPORTED) return st0;
    const int st1 = try_fwd_box<float>(nullptr, output, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth,
                                         pooled_height, pooled_width, lt, mr3d_stream(stream_));
    if (st1 != MR3D_ERR_UNSUPPORTED) return st1;
    const int st2 = try_fwd_sep<float>(nullptr, output, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth,
                                       pooled_height, pooled_width, lt, mr3d_stream(stream_));
    if (st2 != MR3D_ERR_UNSUPPORTED) return st2;
  }
  const int B = pooled_depth * pooled_height * pooled_width;
  dim3 grid(num_rois, (B + kClBins - 1) / kClBins);
  roi_align3d_cl<false><<<grid, kClThreads, 0, mr3d_stream(stream_)>>>(nullptr, output, rois, 0.f, 0.f, sample_num, channels, 1, 1, 1,
                                                                      pooled_depth, pooled_height, pooled_width, 0, lt);
  return mr3d_launch_status();
}

extern "C" int mr3d_roi_align3d_backward_ml(const mr3d_roi_level *levels_host, int num_levels, float finest_scale,
                                            const float *top_grad, const float *rois, int num_rois, int channels,
                                            int pooled_depth, int pooled_height, int pooled_width, int sample_num, int flags,
                                            mr3d_stream_t stream_) {
  LevelTable lt;
  const int st = fill_levels(levels_host, num_levels, finest_scale, true, lt);
  if (st != MR3D_OK) return st;
  if (num_rois < 0 || channels <= 0 || pooled_depth <= 0 || pooled_height <= 0 || pooled_width <= 0) return MR3D_ERR_INVALID_ARG;
  if (num_rois == 0) return MR3D_OK;
  if (!rois || !top_grad) return MR3D_ERR_INVALID_ARG;
  if (!use_cl(MR3D_LAYOUT_NDHWC, sample_num, pooled_depth, pooled_height, pooled_width)) return MR3D_ERR_UNSUPPORTED;
  hipStream_t stream = mr3d_stream(stream_);
  const int bug = (flags & MR3D_ROIALIGN_BUG_COMPAT) ? 1 : 0;
  const int B = pooled_depth * pooled_height * pooled_width;
  {
    const int stw = try_bwd_wave(top_grad, nullptr, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth, pooled_height,
                                 pooled_width, bug, lt, stream);
    if (stw != MR3D_ERR_UNSUPPORTED) return stw;
    const int stv = try_bwd_vox(top_grad, nullptr, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth, pooled_height,
                                pooled_width, bug, lt, stream);
    if (stv != MR3D_ERR_UNSUPPORTED) return stv;
  }
  dim3 grid(num_rois, (B + kClBins - 1) / kClBins);
  roi_align3d_cl<true><<<grid, kClThreads, 0, stream>>>(top_grad, nullptr, rois, 0.f, 0.f, sample_num, channels, 1, 1, 1,
                                                       pooled_depth, pooled_height, pooled_width, bug, lt);
  return mr3d_launch_status();
}

// bf16 feature maps (configs[4]; roi_align_kernel.cu:323 dispatches half as well): same call, `features` of every level
// record points to bf16 NDHWC data; output fp32.  channels % 16 == 0, pooled * sample_num <= 64 per axis.
extern "C" int mr3d_roi_align3d_forward_ml_bf16(const mr3d_roi_level *levels_host, int num_levels, float finest_scale,
                                                const float *rois, int num_rois, int channels, int pooled_depth,
                                                int pooled_height, int pooled_width, int sample_num, float *output,
                                                mr3d_stream_t stream_) {
  LevelTable lt;
  const int st = fill_levels(levels_host, num_levels, finest_scale, false, lt);
  if (st != MR3D_OK) return st;
  if (num_rois < 0 || channels <= 0 || pooled_depth <= 0 || pooled_height <= 0 || pooled_width <= 0) return MR3D_ERR_INVALID_ARG;
  if (num_rois == 0) return MR3D_OK;
  if (!rois || !output) return MR3D_ERR_INVALID_ARG;
  if (channels % kStCh != 0 || sample_num <= 0 || pooled_depth * sample_num > kStAxis || pooled_height * sample_num > kStAxis ||
      pooled_width * sample_num > kStAxis)
    return MR3D_ERR_UNSUPPORTED;
  {
    const int st1 = try_fwd_box<uint16_t>(nullptr, output, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth,
                                            pooled_height, pooled_width, lt, mr3d_stream(stream_));
    if (st1 != MR3D_ERR_UNSUPPORTED) return st1;
    const int st2 = try_fwd_sep<uint16_t>(nullptr, output, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth,
                                          pooled_height, pooled_width, lt, mr3d_stream(stream_));
    if (st2 != MR3D_ERR_UNSUPPORTED) return st2;
  }
  return launch_fwd_staged<uint16_t>(nullptr, output, rois, num_rois, 0.f, 0.f, sample_num, channels, 1, 1, 1, pooled_depth,
                                     pooled_height, pooled_width, lt, mr3d_stream(stream_));
}

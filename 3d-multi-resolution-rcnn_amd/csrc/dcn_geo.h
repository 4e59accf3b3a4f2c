// Geometry shared by the column-free DeformConv3d kernels (deform_conv3d_fused.hip: fp32 sampling from global memory;
// deform_conv3d_bf16.hip: bf16 sampling from a voxel box in LDS): descriptor -> FGeo, the 4 x 4 x 8 position tiles and their
// XCD-contiguous order, the sampling rule of one (position, tap) (mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:83-108,228
// extended by a depth axis).  Internal header: everything lives in the including file's anonymous namespace.
#pragma once
#include "mr3d_common.h"
#include <mutex>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kC = 64;             // Cin = Cout
constexpr int kM = 128;            // output positions per tile
constexpr int kRow = kC + 4;       // LDS row stride (floats) of the column / dcol tile: 16-lane b128 groups hit 64 distinct banks
constexpr int kThreads = 256;
constexpr int kTapsPerWg = 3;      // weight gradient: taps whose 64x64 accumulators one workgroup holds

struct FGeo {
  int N, D, H, W, Do, Ho, Wo;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int K;
  long long total;                 // N * Do * Ho * Wo
  int tiles_d, tiles_h, tiles_w, ntiles;       // 4 x 4 x 8 position blocks of the gather kernels
};

struct Sample {
  bool any;
  int z[2], y[2], x[2];            // clamped corner coordinates
  bool vz[2], vy[2], vx[2];
  float lz, ly, lx;
  int n;
};

// an output position, decoded once per tile (32-bit arithmetic: total < 2^31 is checked on the host)
struct Coord {
  int n, od, oh, ow;
  unsigned pos;
  bool in;
};

// The gather kernels tile the output positions in 4 x 4 x 8 blocks (128 positions): the 27 x 8 corner reads of a block
// stay inside a 6 x 6 x 10 voxel neighbourhood (+ offsets), and blocks are handed to the 8 XCDs in contiguous ranges
// (hardware deals workgroups round-robin: blockIdx & 7 = XCD) so that the blocks in flight on one XCD are neighbours
// and share their halos in that XCD's L2.
constexpr int kGd = 4, kGh = 4, kGw = 8;
static_assert(kGd * kGh * kGw == kM, "tile");

struct TileOrg { int n, od0, oh0, ow0; };

__device__ __forceinline__ int xcd_contiguous(int bid, int ntiles) {     // -> tile id, or -1
  const int per = (ntiles + 7) >> 3;
  const int tile = (bid & 7) * per + (bid >> 3);
  return ((bid >> 3) < per && tile < ntiles) ? tile : -1;
}

__device__ __forceinline__ TileOrg tile_origin(const FGeo &g, int tile) {
  TileOrg o;
  o.ow0 = (tile % g.tiles_w) * kGw; tile /= g.tiles_w;
  o.oh0 = (tile % g.tiles_h) * kGh; tile /= g.tiles_h;
  o.od0 = (tile % g.tiles_d) * kGd;
  o.n = tile / g.tiles_d;
  return o;
}

__device__ __forceinline__ Coord tile_pos(const FGeo &g, const TileOrg &o, int m) {
  Coord c;
  c.n = o.n; c.ow = o.ow0 + (m & (kGw - 1)); c.oh = o.oh0 + ((m >> 3) & (kGh - 1)); c.od = o.od0 + (m >> 5);
  c.in = c.od < g.Do && c.oh < g.Ho && c.ow < g.Wo;
  c.pos = c.in ? (unsigned)(((o.n * g.Do + c.od) * g.Ho + c.oh) * g.Wo + c.ow) : 0u;
  return c;
}

// the three offsets of (position, tap), requested one tap ahead of their use
struct OffsetStream {
  const float *p;
  float nx[3];
  bool on;
  __device__ __forceinline__ void start(const float *__restrict__ offset, const Coord &c, int K, bool active, int tap) {
    on = active && c.in;
    p = offset + (size_t)c.pos * (K * 3);
    nx[0] = nx[1] = nx[2] = 0.f;
    if (on) { nx[0] = p[3 * tap]; nx[1] = p[3 * tap + 1]; nx[2] = p[3 * tap + 2]; }
  }
  __device__ __forceinline__ void take(float &oz, float &oy, float &ox, int next_tap, int K) {
    oz = nx[0]; oy = nx[1]; ox = nx[2];
    if (on && next_tap < K) { nx[0] = p[3 * next_tap]; nx[1] = p[3 * next_tap + 1]; nx[2] = p[3 * next_tap + 2]; }
  }
};

using bf16_t = uint16_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }

// bf16 offsets [position][stride] (the offset conv's padded output), requested one tap ahead of their use
struct OffStream {
  const bf16_t *p;
  float nx[3];
  bool on;
  __device__ __forceinline__ void start(const bf16_t *__restrict__ offset, int stride, const Coord &c, bool active, int tap) {
    on = active && c.in;
    p = offset + (size_t)c.pos * stride;
    nx[0] = nx[1] = nx[2] = 0.f;
    if (on) { nx[0] = bf2f(p[3 * tap]); nx[1] = bf2f(p[3 * tap + 1]); nx[2] = bf2f(p[3 * tap + 2]); }
  }
  __device__ __forceinline__ void take(float &oz, float &oy, float &ox, int next_tap, int K) {
    oz = nx[0]; oy = nx[1]; ox = nx[2];
    if (on && next_tap < K) { nx[0] = bf2f(p[3 * next_tap]); nx[1] = bf2f(p[3 * next_tap + 1]); nx[2] = bf2f(p[3 * next_tap + 2]); }
  }
};

__device__ __forceinline__ Sample sample_at(const FGeo &g, const Coord &c, int kd_, int kh_, int kw_, float oz, float oy, float ox) {
  Sample s;
  s.any = false;
  s.lz = s.ly = s.lx = 0.f;
  s.n = c.n;
#pragma unroll
  for (int a = 0; a < 2; ++a) { s.z[a] = s.y[a] = s.x[a] = 0; s.vz[a] = s.vy[a] = s.vx[a] = false; }
  if (!c.in) return s;
  const float pz = (float)(c.od * g.sd - g.pd + kd_ * g.dd) + oz;
  const float py = (float)(c.oh * g.sh - g.ph + kh_ * g.dh) + oy;
  const float px = (float)(c.ow * g.sw - g.pw + kw_ * g.dw) + ox;
  s.any = pz > -1.f && py > -1.f && px > -1.f && pz < (float)g.D && py < (float)g.H && px < (float)g.W;
  if (!s.any) return s;
  const int z0 = (int)floorf(pz), y0 = (int)floorf(py), x0 = (int)floorf(px);
  s.lz = pz - (float)z0; s.ly = py - (float)y0; s.lx = px - (float)x0;
  s.vz[0] = z0 >= 0; s.vz[1] = z0 + 1 <= g.D - 1;
  s.vy[0] = y0 >= 0; s.vy[1] = y0 + 1 <= g.H - 1;
  s.vx[0] = x0 >= 0; s.vx[1] = x0 + 1 <= g.W - 1;
  s.z[0] = max(z0, 0); s.z[1] = min(z0 + 1, g.D - 1);
  s.y[0] = max(y0, 0); s.y[1] = min(y0 + 1, g.H - 1);
  s.x[0] = max(x0, 0); s.x[1] = min(x0 + 1, g.W - 1);
  return s;
}

// tap -> (kd, kh, kw), advanced incrementally by the tap loops (no division per tap)
struct TapIdx {
  int kd, kh, kw;
  __device__ __forceinline__ void set(const FGeo &g, int tap) { kw = tap % g.kw; kh = (tap / g.kw) % g.kh; kd = tap / (g.kw * g.kh); }
  __device__ __forceinline__ void next(const FGeo &g) {
    if (++kw == g.kw) { kw = 0; if (++kh == g.kh) { kh = 0; ++kd; } }
  }
};

inline int fused_geo(const mr3d_deform_conv3d_desc *d, int cout, FGeo &g) {
  if (!d || d->n <= 0 || d->c != kC || cout != kC || d->dg != 1) return MR3D_ERR_UNSUPPORTED;
  g.N = d->n; g.D = d->d; g.H = d->h; g.W = d->w;
  g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
  g.pd = d->pd; g.ph = d->ph; g.pw = d->pw; g.dd = d->dil_d; g.dh = d->dil_h; g.dw = d->dil_w;
  g.K = d->kd * d->kh * d->kw;
  if (g.K <= 0 || g.D <= 0 || g.H <= 0 || g.W <= 0 || g.sd <= 0 || g.sh <= 0 || g.sw <= 0) return MR3D_ERR_INVALID_ARG;
  g.Do = (d->d + 2 * d->pd - (d->dil_d * (d->kd - 1) + 1)) / d->sd + 1;
  g.Ho = (d->h + 2 * d->ph - (d->dil_h * (d->kh - 1) + 1)) / d->sh + 1;
  g.Wo = (d->w + 2 * d->pw - (d->dil_w * (d->kw - 1) + 1)) / d->sw + 1;
  if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MR3D_ERR_INVALID_ARG;
  if ((long long)g.N * g.D * g.H * g.W * kC >= (1ll << 31)) return MR3D_ERR_UNSUPPORTED;      // 32-bit element offsets in the tables
  g.total = (long long)g.N * g.Do * g.Ho * g.Wo;
  if (g.total >= (1ll << 31) / (g.K * 3)) return MR3D_ERR_UNSUPPORTED;                        // 32-bit position / offset indices
  g.tiles_d = (g.Do + kGd - 1) / kGd; g.tiles_h = (g.Ho + kGh - 1) / kGh; g.tiles_w = (g.Wo + kGw - 1) / kGw;
  g.ntiles = g.N * g.tiles_d * g.tiles_h * g.tiles_w;
  return MR3D_OK;
}

// dynamic LDS above 64 KB needs the opt-in attribute; set once per kernel and size (the largest request so far)
template <class Kern>
int allow_lds(Kern k, size_t bytes) {
  static const void *fns[8];
  static size_t granted[8];
  static int n = 0;
  static std::mutex mu;
  const void *f = reinterpret_cast<const void *>(k);
  std::lock_guard<std::mutex> lock(mu);
  int at = -1;
  for (int i = 0; i < n; ++i)
    if (fns[i] == f) at = i;
  if (at >= 0 && bytes <= granted[at]) return MR3D_OK;
  if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return MR3D_ERR_LAUNCH;
  }
  if (at < 0 && n < 8) { at = n++; fns[at] = f; }
  if (at >= 0) granted[at] = bytes;
  return MR3D_OK;
}

}  // namespace

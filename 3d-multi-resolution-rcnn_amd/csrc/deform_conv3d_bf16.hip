// DeformConv3d with bf16 activation storage, sampled from a voxel box in LDS -- BASELINE.json configs[4] (bf16 + DeformConv3d
// neck).  Same operator and sampling rule as csrc/deform_conv3d_fused.hip (the reference's 2D rule extended by a depth axis,
// mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:83-108,228: 0 unless -1 < p < dim on every axis, floor(), corners outside the
// volume contribute 0), Cin = Cout = 64, one deformable group, no modulation.
//
// The fp32 kernels gather every corner from global memory: 27 taps x 8 corners x 256 B per output position, ~14 TB/s of L1 / L2
// reads on P2 -- they are bound by the vector memory path, not by HBM or the matrix cores.  Here a workgroup owns a 4 x 4 x 8 block
// of output positions and first copies the voxels its samples can reach with |offset| < 1 -- the block's tap footprint plus one
// voxel on every side, 8 x 8 x 12 voxels x 64 channels bf16 = 96 KB -- into LDS, ONCE for all 27 taps (36x fewer bytes through
// L1 / L2); corner reads are ds_read_b128 (8 lanes x 16 B = the 64 channels of a voxel).  A sample that leaves the box (|offset|
// >= 1) takes the corner values from global memory instead: rare, and exact.
//
// Bank conflicts.  ds_read_b128 is served in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 --
// and a group is conflict-free when its lanes cover the 64 banks of a 256-B row.  A voxel is 128 B, half a row, the half given by
// the parity of its x coordinate in the box (even box width).  Eight lanes share a sample (slot s = lane >> 3); a group holds
// halves of slots {0,1,2,3} (or {4..7}), and it is conflict-free iff slots 0 / 3 and slots 1 / 2 read voxels of opposite
// parity.  The two x-corners of a sample always have opposite parity, so every slot reads its x-corners in the order that
// puts an EVEN voxel first for slots 0, 1, 4, 5 and an ODD voxel first for slots 2, 3, 6, 7: no conflicts whatever the offsets.
//
//   forward   columns[128 pos][64 ci] (bf16, LDS) = weighted corner sums;  y += columns W[tap]   v_mfma_f32_32x32x16_bf16
//   wgrad     the columns are written TRANSPOSED ([ci][pos], rows permuted so that the 2-byte stores of a wave spread over the
//             banks) next to a transposed dy tile;  dW[tap] += columns^T dy;  a workgroup keeps 9 taps x (64 x 64) fp32 over its
//             share of the tiles (5 or 4 taps per wave) and adds them to dW once
//   doffset   dcol = dy W[tap]^T (MFMA, bf16 tile in LDS), then per (position, tap): sum_c dcol[c] d(sample[c]) / d(coordinate)
//             from the box's corner values, reduced over the sample's 8 lanes with DPP
// The input gradient (a scatter) stays with the fp32 LDS-box accumulator of deform_conv3d_fused.hip (dcn_dx_tiled_kernel).
// Arithmetic: corner values and weights are bf16, columns / dcol are rounded to bf16 (they are MFMA operands), every sum is fp32.
#include "dcn_geo.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kBT = 512;                       // 8 waves: 16 positions of the tile each
constexpr int kBW = kBT / 64;
constexpr int kBoxMax = 768;                   // voxels of the LDS box (x 128 B)
constexpr int kBoxBytes = kBoxMax * 128;
constexpr int kColB = 144;                     // bytes per row of a [128 pos][64 ch] bf16 tile: +16 B so that the A-fragment reads
constexpr int kColBytes = kM * kColB;          //   of 16 rows hit 16 different 16-byte slots of the 256-B bank row
constexpr int kTrB = 272;                      // bytes per row of a transposed [64 ch][128 pos] bf16 tile (+16 B, same reason)
constexpr int kTrBytes = kC * kTrB;
constexpr int kTabBytes = kBW * 2 * 64 * 16;   // per wave, double-buffered: 4 taps x 16 positions, 16 B each
constexpr int kDead = -1, kFar = -2;           // entry codes: sample outside the validity box / outside the LDS box

struct BoxGeo { int rz, ry, rx, nbox; };       // rx even

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ v2f unpack2(uint32_t d) {
  v2f r;
  r.x = __uint_as_float(d << 16);
  r.y = __uint_as_float(d & 0xffff0000u);
  return r;
}
__device__ __forceinline__ void fma8(v2f (&acc)[4], float w, const uint4 &v) {
  const v2f ww = {w, w};
  acc[0] += ww * unpack2(v.x); acc[1] += ww * unpack2(v.y); acc[2] += ww * unpack2(v.z); acc[3] += ww * unpack2(v.w);
}
__device__ __forceinline__ float dot8(const v2f (&a)[4], const uint4 &v) {
  v2f s = a[0] * unpack2(v.x);
  s += a[1] * unpack2(v.y); s += a[2] * unpack2(v.z); s += a[3] * unpack2(v.w);
  return s.x + s.y;
}
__device__ __forceinline__ uint4 pack8(const v2f (&a)[4]) {
  return make_uint4(pack2(a[0].x, a[0].y), pack2(a[1].x, a[1].y), pack2(a[2].x, a[2].y), pack2(a[3].x, a[3].y));
}

struct BoxOrg { int n, z0, y0, x0; };          // image and the volume coordinates of box voxel (0, 0, 0)

__device__ __forceinline__ BoxOrg box_origin(const FGeo &g, const TileOrg &o) {
  BoxOrg b;
  b.n = o.n; b.z0 = o.od0 * g.sd - g.pd - 1; b.y0 = o.oh0 * g.sh - g.ph - 1; b.x0 = o.ow0 * g.sw - g.pw - 1;
  return b;
}

// the tile's voxel box -> LDS (zeros outside the volume: a corner there then contributes 0 without a validity test)
__device__ __forceinline__ void load_box(const bf16_t *__restrict__ x, const FGeo &g, const BoxGeo &b, const BoxOrg &o, uint4 *box,
                                         int tid) {
  const int per_row = b.rx * 8, total = b.nbox * 8;
  constexpr int NIT = kBoxMax * 8 / kBT;
  uint4 v[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int e = tid + j * kBT;
    const int row = e / per_row, i = e - row * per_row;
    const int bz = row / b.ry, by = row - bz * b.ry;
    const int z = o.z0 + bz, y = o.y0 + by, xx = o.x0 + (i >> 3);
    v[j] = make_uint4(0u, 0u, 0u, 0u);
    if (e < total && (unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H && (unsigned)xx < (unsigned)g.W)
      v[j] = *reinterpret_cast<const uint4 *>(x + ((((size_t)o.n * g.D + z) * g.H + y) * g.W + xx) * kC + (i & 7) * 8);
  }
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int e = tid + j * kBT;
    if (e < total) box[e] = v[j];
  }
}

// table entry of one (position, tap): code = box voxel index of corner (0,0,0) and the fractions, or kFar and the sample's volume
// coordinates, or kDead
__device__ __forceinline__ void make_entry(const FGeo &g, const BoxGeo &b, const BoxOrg &o, const Coord &c, const TapIdx &ti,
                                           float oz, float oy, float ox, int4 *dst) {
  int code = kDead;
  float fz = 0.f, fy = 0.f, fx = 0.f;
  if (c.in) {
    const float pz = (float)(c.od * g.sd - g.pd + ti.kd * g.dd) + oz;
    const float py = (float)(c.oh * g.sh - g.ph + ti.kh * g.dh) + oy;
    const float px = (float)(c.ow * g.sw - g.pw + ti.kw * g.dw) + ox;
    if (pz > -1.f && py > -1.f && px > -1.f && pz < (float)g.D && py < (float)g.H && px < (float)g.W) {
      const int z0 = (int)floorf(pz), y0 = (int)floorf(py), x0 = (int)floorf(px);
      const int bz = z0 - o.z0, by = y0 - o.y0, bx = x0 - o.x0;
      if ((unsigned)bz < (unsigned)(b.rz - 1) && (unsigned)by < (unsigned)(b.ry - 1) && (unsigned)bx < (unsigned)(b.rx - 1)) {
        code = (bz * b.ry + by) * b.rx + bx;
        fz = pz - (float)z0; fy = py - (float)y0; fx = px - (float)x0;
      } else {
        code = kFar;
        fz = pz; fy = py; fx = px;
      }
    }
  }
  *dst = make_int4(code, __float_as_int(fz), __float_as_int(fy), __float_as_int(fx));
}

// Table entries are made four taps at a time by all 64 lanes of a wave (lane = 16 * (tap & 3) + position: one lane per entry;
// 16 active lanes per tap cost the same instructions), into the wave's own table (batch & 1): no barrier involved.  The three
// offsets of the NEXT batch are requested right after a batch is made, four taps before they are needed.
struct EntryBatches {
  const bf16_t *p;                // this lane's position row of the offsets
  Coord c;
  float nx[3];
  int tap0, ntap, sub;            // taps [tap0, tap0 + ntap) are walked; sub = lane >> 4
  __device__ __forceinline__ void fetch(int t) {            // t = tap - tap0 of this lane's next entry
    nx[0] = nx[1] = nx[2] = 0.f;
    if (c.in && t < ntap) { const int k = 3 * (tap0 + t); nx[0] = bf2f(p[k]); nx[1] = bf2f(p[k + 1]); nx[2] = bf2f(p[k + 2]); }
  }
  __device__ __forceinline__ void start(const bf16_t *__restrict__ offset, int stride, const Coord &me, int lane, int tap0_, int ntap_) {
    c = me; tap0 = tap0_; ntap = ntap_; sub = lane >> 4;
    p = offset + (size_t)me.pos * stride;
    fetch(sub);
  }
  // entries of taps tap0 + 4 * batch .. + 3 -> tab[(batch & 1) * 64 + lane]
  __device__ __forceinline__ void make(const FGeo &g, const BoxGeo &b, const BoxOrg &o, int batch, int4 *tab, int lane) {
    const int t = 4 * batch + sub;
    TapIdx ti;
    ti.set(g, min(tap0 + t, g.K - 1));
    Coord cc = c;
    cc.in = c.in && t < ntap;
    make_entry(g, b, o, cc, ti, nx[0], nx[1], nx[2], tab + (batch & 1) * 64 + lane);
    fetch(t + 4);
  }
};

// the 8 corner values (this lane's 8 channels) of an in-box sample, x-corners in the lane's conflict-free order:
// v[(a * 2 + b) * 2 + i] = corner (z0 + a, y0 + b, x0 + (i ? 1 - cf : cf))
__device__ __forceinline__ void box_corners(const uint4 *box, const BoxGeo &b, int code, int cf, int piece, uint4 (&v)[8]) {
  const uint4 *p0 = box + (code + cf) * 8 + piece, *p1 = box + (code + (cf ^ 1)) * 8 + piece;
  const int sy = b.rx * 8, sz = b.ry * b.rx * 8;
  v[0] = p0[0]; v[1] = p1[0]; v[2] = p0[sy]; v[3] = p1[sy];
  v[4] = p0[sz]; v[5] = p1[sz]; v[6] = p0[sz + sy]; v[7] = p1[sz + sy];
}

// a sample outside the LDS box: the same sum from global memory, corner by corner
struct Acc8 { v2f a[4]; };
__device__ __noinline__ Acc8 far_columns_nv(const bf16_t *__restrict__ x, int D, int H, int W, int n, float pz, float py, float px, int piece) {
  v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  const int z0 = (int)floorf(pz), y0 = (int)floorf(py), x0 = (int)floorf(px);
  const float lz = pz - (float)z0, ly = py - (float)y0, lx = px - (float)x0;
  const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
  for (int a = 0; a < 2; ++a)
    for (int bb = 0; bb < 2; ++bb)
      for (int c = 0; c < 2; ++c) {
        const int z = z0 + a, y = y0 + bb, xx = x0 + c;
        if ((unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
          fma8(acc, wz[a] * wy[bb] * wx[c], *reinterpret_cast<const uint4 *>(x + ((((size_t)n * D + z) * H + y) * W + xx) * kC + piece * 8));
      }
  Acc8 r;
  r.a[0] = acc[0]; r.a[1] = acc[1]; r.a[2] = acc[2]; r.a[3] = acc[3];
  return r;
}
__device__ __forceinline__ void far_columns(const bf16_t *__restrict__ x, const FGeo &g, int n, float pz, float py, float px, int piece,
                                            v2f (&acc)[4]) {
  const Acc8 r = far_columns_nv(x, g.D, g.H, g.W, n, pz, py, px, piece);
  acc[0] = r.a[0]; acc[1] = r.a[1]; acc[2] = r.a[2]; acc[3] = r.a[3];
}

// The weighted corner sums of this wave's 16 positions for one tap: two groups of 8 samples (slot = lane >> 3), 8 channels per
// lane (piece = lane & 7); the 16 corner reads of both groups are issued before the first is used.
// sink(local row 0..15, piece, sums of the 8 channels).
template <bool BOTH, class Sink>
__device__ __forceinline__ void gather_columns(const uint4 *box, const BoxGeo &b, const int4 *tab, int lane, const bf16_t *__restrict__ x,
                                               const FGeo &g, int n, Sink &&sink) {
  const int s = lane >> 3, piece = lane & 7, pol = (s >> 1) & 1;
  constexpr int NG = BOTH ? 2 : 1;                  // groups whose reads are in flight together (registers: 32 per group)
#pragma unroll
  for (int g0 = 0; g0 < 2; g0 += NG) {
    int4 e[NG];
    uint4 v[NG][8];
    int cf[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) e[u] = tab[(g0 + u) * 8 + s];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      cf[u] = (e[u].x & 1) ^ pol;
      box_corners(box, b, max(e[u].x, 0), cf[u], piece, v[u]);        // a dead / far sample reads voxel 0 and drops it
    }
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      const float lz = __int_as_float(e[u].y), ly = __int_as_float(e[u].z), lx = __int_as_float(e[u].w);
      if (e[u].x >= 0) {
        const float wx0 = cf[u] ? lx : 1.f - lx, wx1 = cf[u] ? 1.f - lx : lx;
        const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const float wab = wz[a] * wy[bb];
            fma8(acc, wab * wx0, v[u][(a * 2 + bb) * 2]);
            fma8(acc, wab * wx1, v[u][(a * 2 + bb) * 2 + 1]);
          }
      } else if (e[u].x == kFar) {
        far_columns(x, g, n, lz, ly, lx, piece, acc);
      }
      sink((g0 + u) * 8 + s, piece, acc);
    }
  }
}

// physical row of channel c in a transposed tile: the 8 channels of a lane (piece * 8 + j) land on rows j * 8 + piece, so the
// 2-byte stores of a wave's 8 pieces go to 8 consecutive rows (272 B apart: 8 different bank quads)
__device__ __forceinline__ int tr_row(int c) { return (c & 7) * 8 + (c >> 3); }

// ---------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(kBT) void dcn_bf16_fwd_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ offset, int off_stride,
                                                          const bf16_t *__restrict__ wf, bf16_t *__restrict__ y, FGeo g, BoxGeo b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint4 *box = reinterpret_cast<uint4 *>(lds);
  unsigned char *col = lds + kBoxBytes;                                  // 2 x [128][kColB]
  int4 *tab = reinterpret_cast<int4 *>(col + 2 * kColBytes);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int mt = wave >> 1, nt = wave & 1;                               // this wave's 32 x 32 tile of the [128 pos][64 co] product
  const int tile_id = xcd_contiguous(blockIdx.x, g.ntiles);
  if (tile_id < 0) return;
  const TileOrg org = tile_origin(g, tile_id);
  const BoxOrg bo = box_origin(g, org);
  load_box(x, g, b, bo, box, tid);

  EntryBatches eb;
  eb.start(offset, off_stride, tile_pos(g, org, wave * 16 + (lane & 15)), lane, 0, g.K);
  int4 *mytab = tab + wave * 128;
  auto gather = [&](int tap, unsigned char *dst) {
    gather_columns<true>(box, b, mytab + ((tap >> 2) & 1) * 64 + (tap & 3) * 16, lane, x, g, org.n, [&](int rl, int piece, const v2f (&acc)[4]) {
      *reinterpret_cast<uint4 *>(dst + (wave * 16 + rl) * kColB + piece * 16) = pack8(acc);
    });
  };
  // B fragments of a tap: W[tap][ci = 16 kk + 8 h .. + 7][co = 32 nt + r], packed [tap][ci / 8][co][8]
  const bf16x8 *wlane = reinterpret_cast<const bf16x8 *>(wf) + (size_t)h * kC + nt * 32 + r;
  auto load_b = [&](bf16x8 (&dst)[4], int tap) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dst[kk] = wlane[((size_t)tap * 8 + kk * 2) * kC];
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  bf16x8 bcur[4], bnxt[4];
  load_b(bcur, 0);
  eb.make(g, b, bo, 0, mytab, lane);
  __syncthreads();                                  // the box is complete
  gather(0, col);
  __syncthreads();
  for (int tap = 0; tap < g.K; ++tap) {
    if (tap + 1 < g.K) load_b(bnxt, tap + 1);
    if ((tap & 3) == 1 && 4 * ((tap >> 2) + 1) < g.K) eb.make(g, b, bo, (tap >> 2) + 1, mytab, lane);
    const unsigned char *arow = col + (tap & 1) * kColBytes + (mt * 32 + r) * kColB + 16 * h;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(arow + kk * 32), bcur[kk], acc, 0, 0, 0);
    if (tap + 1 < g.K) gather(tap + 1, col + ((tap + 1) & 1) * kColBytes);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bcur[kk] = bnxt[kk];
  }
  // result -> bf16 rows in LDS (the column buffers are free) -> 16-byte stores
  unsigned char *out = col;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    *reinterpret_cast<bf16_t *>(out + row * kColB + (nt * 32 + r) * 2) = __builtin_bit_cast(bf16_t, (__bf16)acc[e]);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int it = tid + j * kBT, row = it >> 3, piece = it & 7;
    const Coord c = tile_pos(g, org, row);
    if (c.in) *reinterpret_cast<uint4 *>(y + (size_t)c.pos * kC + piece * 8) = *reinterpret_cast<const uint4 *>(out + row * kColB + piece * 16);
  }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
constexpr int kWgTaps = 9;                          // taps per workgroup (their 64 x 64 accumulators: 5 or 4 per wave)

__global__ __launch_bounds__(kBT) void dcn_bf16_wgrad_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ offset, int off_stride,
                                                            const bf16_t *__restrict__ dy, float *__restrict__ dw, FGeo g, BoxGeo b,
                                                            int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint4 *box = reinterpret_cast<uint4 *>(lds);
  unsigned char *colT = lds + kBoxBytes;                                 // 2 x [64 ci rows][kTrB]
  unsigned char *dyT = colT + 2 * kTrBytes;                              // [64 co rows][kTrB]; once its fragments sit in registers
  int4 *tab = reinterpret_cast<int4 *>(dyT);                             //   the region holds the waves' sample tables
  static_assert(kTabBytes <= kTrBytes, "the tables live in the dy tile's LDS");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int mb = (wave >> 1) & 1, nb = wave & 1, par = wave >> 2;        // (ci block, co block) and the tap parity this wave sums
  const int groups = (g.K + kWgTaps - 1) / kWgTaps;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tg = slot % groups, chunk = xcd * nsplit + slot / groups;    // the tap groups of a chunk share an XCD (same voxels)
  const int nchunks = 8 * nsplit;
  const int per = (g.ntiles + nchunks - 1) / nchunks;
  const int t_begin = chunk * per, t_end = min(g.ntiles, t_begin + per);
  const int tap0 = tg * kWgTaps, ntap = min(kWgTaps, g.K - tap0);

  constexpr int NA = (kWgTaps + 1) / 2;
  f32x16 acc[NA];
#pragma unroll
  for (int t = 0; t < NA; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  int4 *mytab = tab + wave * 128;

  for (int tile_id = t_begin; tile_id < t_end; ++tile_id) {
    const TileOrg org = tile_origin(g, tile_id);
    const BoxOrg bo = box_origin(g, org);
    uint4 dyv[2];
    uint32_t live = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int it = tid + j * kBT, row = it >> 3, piece = it & 7;
      const Coord c = tile_pos(g, org, row);
      dyv[j] = make_uint4(0u, 0u, 0u, 0u);
      if (c.in) dyv[j] = *reinterpret_cast<const uint4 *>(dy + (size_t)c.pos * kC + piece * 8);
      live |= (dyv[j].x | dyv[j].y | dyv[j].z | dyv[j].w) & 0x7fff7fffu;
    }
    // (a barrier: the previous tile's box, dy tile and columns are done with.)  dy == 0 on the whole tile -- the usual case in
    // detector training, where the gradient of a pyramid level comes from 256 sampled anchors and the sampled rois' footprints --
    // adds nothing to dW: skip the box and the gather passes
    if (!__syncthreads_or((int)live)) continue;
    load_box(x, g, b, bo, box, tid);
#pragma unroll
    for (int j = 0; j < 2; ++j) {                   // dy tile [pos][co] -> dyT[co row][pos]
      const int it = tid + j * kBT, row = it >> 3, piece = it & 7;
      const uint4 v = dyv[j];
      const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 8; ++q)
        *reinterpret_cast<bf16_t *>(dyT + (q * 8 + piece) * kTrB + row * 2) = (bf16_t)(d[q >> 1] >> (16 * (q & 1)));
    }
    EntryBatches eb;
    eb.start(offset, off_stride, tile_pos(g, org, wave * 16 + (lane & 15)), lane, tap0, ntap);
    auto gather = [&](int t, unsigned char *dst) {        // t = tap - tap0
      gather_columns<false>(box, b, mytab + ((t >> 2) & 1) * 64 + (t & 3) * 16, lane, x, g, org.n, [&](int rl, int piece, const v2f (&a)[4]) {
        const int pos = wave * 16 + rl;
        const float f[8] = {a[0].x, a[0].y, a[1].x, a[1].y, a[2].x, a[2].y, a[3].x, a[3].y};
#pragma unroll
        for (int q = 0; q < 8; ++q)
          *reinterpret_cast<bf16_t *>(dst + (q * 8 + piece) * kTrB + pos * 2) = __builtin_bit_cast(bf16_t, (__bf16)f[q]);
      });
    };
    __syncthreads();                                // box and dy tile complete
    bf16x8 bfr[8];                                  // B[k = position 16 kk + 8 h ..][n = co row 32 nb + r]
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) bfr[kk] = *reinterpret_cast<const bf16x8 *>(dyT + (nb * 32 + r) * kTrB + kk * 32 + 16 * h);
    __syncthreads();                                // every wave has its dy fragments: the region becomes the tables
    eb.make(g, b, bo, 0, mytab, lane);
    gather(0, colT);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kWgTaps; ++t) {
      if (t < ntap) {
        if ((t & 3) == 1 && 4 * ((t >> 2) + 1) < ntap) eb.make(g, b, bo, (t >> 2) + 1, mytab, lane);
        if ((t & 1) == par) {
          const unsigned char *arow = colT + (t & 1) * kTrBytes + (mb * 32 + r) * kTrB + 16 * h;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk)
            acc[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(arow + kk * 32), bfr[kk], acc[t >> 1], 0, 0, 0);
        }
        if (t + 1 < ntap) gather(t + 1, colT + ((t + 1) & 1) * kTrBytes);
        __syncthreads();
      }
    }
  }
#pragma unroll
  for (int t = 0; t < kWgTaps; ++t) {
    if (t < ntap && (t & 1) == par) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, nn = nb * 32 + r;        // physical rows of colT / dyT
        const int ci = (m & 7) * 8 + (m >> 3), co = (nn & 7) * 8 + (nn >> 3);            // tr_row is its own inverse
        atomicAdd(dw + ((size_t)(tap0 + t) * kC + ci) * kC + co, acc[t >> 1][e]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- offset gradient
__device__ __forceinline__ float sum8(float t) {       // over the 8 lanes of a sample: every lane gets the total
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xf, 0xf, true));   // row_half_mirror
  return t;
}

struct Grad3 { float z, y, x; };
__device__ __noinline__ Grad3 far_doffset_nv(const bf16_t *__restrict__ x, int D, int H, int W, int n, float pz, float py, float px, int piece,
                                            v2f dc0, v2f dc1, v2f dc2, v2f dc3) {
  const v2f dc[4] = {dc0, dc1, dc2, dc3};
  float gz = 0.f, gy = 0.f, gx = 0.f;
  struct { int D, H, W; } g = {D, H, W};
  const int z0 = (int)floorf(pz), y0 = (int)floorf(py), x0 = (int)floorf(px);
  const float lz = pz - (float)z0, ly = py - (float)y0, lx = px - (float)x0;
  const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx}, sg[2] = {-1.f, 1.f};
  for (int a = 0; a < 2; ++a)
    for (int bb = 0; bb < 2; ++bb)
      for (int c = 0; c < 2; ++c) {
        const int z = z0 + a, y = y0 + bb, xx = x0 + c;
        if ((unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) {
          const float dot = dot8(dc, *reinterpret_cast<const uint4 *>(x + ((((size_t)n * g.D + z) * g.H + y) * g.W + xx) * kC + piece * 8));
          gz = fmaf(sg[a] * wy[bb] * wx[c], dot, gz);
          gy = fmaf(wz[a] * sg[bb] * wx[c], dot, gy);
          gx = fmaf(wz[a] * wy[bb] * sg[c], dot, gx);
        }
      }
  Grad3 r = {gz, gy, gx};
  return r;
}
__device__ __forceinline__ void far_doffset(const bf16_t *__restrict__ x, const FGeo &g, int n, float pz, float py, float px, int piece,
                                            const v2f (&dc)[4], float &gz, float &gy, float &gx) {
  const Grad3 r = far_doffset_nv(x, g.D, g.H, g.W, n, pz, py, px, piece, dc[0], dc[1], dc[2], dc[3]);
  gz = r.z; gy = r.y; gx = r.x;
}

__global__ __launch_bounds__(kBT) void dcn_bf16_doffset_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ offset, int off_stride,
                                                              const bf16_t *__restrict__ dy, const bf16_t *__restrict__ wb,
                                                              bf16_t *__restrict__ doffset, int doff_stride, FGeo g, BoxGeo b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint4 *box = reinterpret_cast<uint4 *>(lds);
  unsigned char *dcol = lds + kBoxBytes;                                 // 2 x [128 pos][kColB]: dy W[tap]^T, bf16
  int4 *tab = reinterpret_cast<int4 *>(dcol + 2 * kColBytes);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int mt = wave >> 1, nt = wave & 1;
  const int tile_id = xcd_contiguous(blockIdx.x, g.ntiles);
  if (tile_id < 0) return;
  const TileOrg org = tile_origin(g, tile_id);
  const BoxOrg bo = box_origin(g, org);

  // A operand: dy rows 32 mt + r, k = co: registers for the whole tap loop
  bf16x8 afr[4];
  {
    const Coord c = tile_pos(g, org, mt * 32 + r);
    uint32_t live = 0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (c.in) v = *reinterpret_cast<const uint4 *>(dy + (size_t)c.pos * kC + kk * 16 + 8 * h);
      live |= (v.x | v.y | v.z | v.w) & 0x7fff7fffu;
      afr[kk] = __builtin_bit_cast(bf16x8, v);
    }
    // dy == 0 on the whole tile (the usual case in detector training, see the weight gradient): the offset gradient keeps the
    // zeros the caller put there
    if (!__syncthreads_or((int)live)) return;
  }
  load_box(x, g, b, bo, box, tid);
  // B fragments of a tap: W[tap]^T, B[k = co][n = ci = 32 nt + r], packed [tap][co / 8][ci][8]
  const bf16x8 *wlane = reinterpret_cast<const bf16x8 *>(wb) + (size_t)h * kC + nt * 32 + r;
  auto load_b = [&](bf16x8 (&dst)[4], int tap) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dst[kk] = wlane[((size_t)tap * 8 + kk * 2) * kC];
  };
  auto mma = [&](const bf16x8 (&bw)[4], unsigned char *dst) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kk], bw[kk], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      *reinterpret_cast<bf16_t *>(dst + row * kColB + (nt * 32 + r) * 2) = __builtin_bit_cast(bf16_t, (__bf16)acc[e]);
    }
  };
  EntryBatches eb;
  eb.start(offset, off_stride, tile_pos(g, org, wave * 16 + (lane & 15)), lane, 0, g.K);
  int4 *mytab = tab + wave * 128;
  const int s = lane >> 3, piece = lane & 7, pol = (s >> 1) & 1;
  bf16x8 bcur[4], bnxt[4];
  load_b(bcur, 0);
  eb.make(g, b, bo, 0, mytab, lane);
  mma(bcur, dcol);
  __syncthreads();                                  // box, dcol of tap 0
  for (int tap = 0; tap < g.K; ++tap) {
    if (tap + 1 < g.K) load_b(bnxt, tap + 1);
    if ((tap & 3) == 1 && 4 * ((tap >> 2) + 1) < g.K) eb.make(g, b, bo, (tap >> 2) + 1, mytab, lane);
    const unsigned char *dc_tile = dcol + (tap & 1) * kColBytes;
    const int4 *tb = mytab + ((tap >> 2) & 1) * 64 + (tap & 3) * 16;
    // both groups' table entries, dcol rows and 16 corner reads are requested before the first is used
    int4 e2[2];
    uint4 dcv2[2], v2[2][8];
    int cf2[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      e2[gi] = tb[gi * 8 + s];
      dcv2[gi] = *reinterpret_cast<const uint4 *>(dc_tile + (wave * 16 + gi * 8 + s) * kColB + piece * 16);
    }
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      cf2[gi] = (e2[gi].x & 1) ^ pol;
      box_corners(box, b, max(e2[gi].x, 0), cf2[gi], piece, v2[gi]);
    }
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int rl = gi * 8 + s, row = wave * 16 + rl;
      const int4 e = e2[gi];
      const uint4 dcv = dcv2[gi];
      const v2f dc[4] = {unpack2(dcv.x), unpack2(dcv.y), unpack2(dcv.z), unpack2(dcv.w)};
      float gz = 0.f, gy = 0.f, gx = 0.f;
      if (e.x >= 0) {
        const float lz = __int_as_float(e.y), ly = __int_as_float(e.z), lx = __int_as_float(e.w);
        const int cf = cf2[gi];
        const uint4 (&v)[8] = v2[gi];
        const float wx0 = cf ? lx : 1.f - lx, wx1 = cf ? 1.f - lx : lx;       // weight of the first / second x-corner read
        const float sx0 = cf ? 1.f : -1.f;                                    // d(weight of the first) / d(lx); the second: -sx0
        const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly}, sg[2] = {-1.f, 1.f};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const float d0 = dot8(dc, v[(a * 2 + bb) * 2]), d1 = dot8(dc, v[(a * 2 + bb) * 2 + 1]);
            const float dx_ = wx0 * d0 + wx1 * d1;                            // sum over the x-corners
            gz = fmaf(sg[a] * wy[bb], dx_, gz);
            gy = fmaf(wz[a] * sg[bb], dx_, gy);
            gx = fmaf(wz[a] * wy[bb] * sx0, d0 - d1, gx);
          }
      } else if (e.x == kFar) {
        far_doffset(x, g, org.n, __int_as_float(e.y), __int_as_float(e.z), __int_as_float(e.w), piece, dc, gz, gy, gx);
      }
      gz = sum8(gz); gy = sum8(gy); gx = sum8(gx);
      const Coord rc = tile_pos(g, org, row);
      if (piece == 0 && rc.in) {
        bf16_t *d = doffset + (size_t)rc.pos * doff_stride + tap * 3;
        d[0] = __builtin_bit_cast(bf16_t, (__bf16)gz);
        d[1] = __builtin_bit_cast(bf16_t, (__bf16)gy);
        d[2] = __builtin_bit_cast(bf16_t, (__bf16)gx);
      }
    }
    if (tap + 1 < g.K) mma(bnxt, dcol + ((tap + 1) & 1) * kColBytes);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bcur[kk] = bnxt[kk];
  }
}

int box_geo(const FGeo &g, BoxGeo &b) {
  b.rz = (kGd - 1) * g.sd + (g.kd - 1) * g.dd + 3;
  b.ry = (kGh - 1) * g.sh + (g.kh - 1) * g.dh + 3;
  b.rx = (kGw - 1) * g.sw + (g.kw - 1) * g.dw + 3;
  b.rx += b.rx & 1;                                  // even: the parity of a voxel's index is the parity of its x coordinate
  b.nbox = b.rz * b.ry * b.rx;
  return b.nbox <= kBoxMax ? MR3D_OK : MR3D_ERR_UNSUPPORTED;
}

constexpr size_t kFwdLdsB = (size_t)kBoxBytes + 2 * kColBytes + kTabBytes;
constexpr size_t kWgLdsB = (size_t)kBoxBytes + 3 * kTrBytes;
static_assert(kFwdLdsB <= 160 * 1024 && kWgLdsB <= 160 * 1024, "LDS");

}  // namespace

extern "C" int mr3d_dcn_bf16_supported(const mr3d_deform_conv3d_desc *d, int cout) {
  FGeo g;
  BoxGeo b;
  return fused_geo(d, cout, g) == MR3D_OK && box_geo(g, b) == MR3D_OK ? 1 : 0;
}

extern "C" int mr3d_dcn_bf16_forward(const mr3d_deform_conv3d_desc *d, int cout, const uint16_t *x, const uint16_t *offset,
                                     int off_stride, const uint16_t *w_packed, uint16_t *y, mr3d_stream_t s) {
  FGeo g;
  BoxGeo b;
  int st = fused_geo(d, cout, g);
  if (st == MR3D_OK) st = box_geo(g, b);
  if (st != MR3D_OK) return st;
  if (!x || !offset || !w_packed || !y || off_stride < g.K * 3) return MR3D_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_packed)) & 15) return MR3D_ERR_INVALID_ARG;
  if (allow_lds(dcn_bf16_fwd_kernel, kFwdLdsB) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const unsigned tiles = 8u * (unsigned)((g.ntiles + 7) / 8);
  dcn_bf16_fwd_kernel<<<tiles, kBT, kFwdLdsB, mr3d_stream(s)>>>(x, offset, off_stride, w_packed, y, g, b);
  return mr3d_launch_status();
}

extern "C" int mr3d_dcn_bf16_wgrad(const mr3d_deform_conv3d_desc *d, int cout, const uint16_t *x, const uint16_t *offset, int off_stride,
                                   const uint16_t *dy, float *dw, mr3d_stream_t s) {
  FGeo g;
  BoxGeo b;
  int st = fused_geo(d, cout, g);
  if (st == MR3D_OK) st = box_geo(g, b);
  if (st != MR3D_OK) return st;
  if (!x || !offset || !dy || !dw || off_stride < g.K * 3) return MR3D_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) return MR3D_ERR_INVALID_ARG;
  if (allow_lds(dcn_bf16_wgrad_kernel, kWgLdsB) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const int groups = (g.K + kWgTaps - 1) / kWgTaps;
  int per_xcd = 32 / groups;                          // one workgroup per CU (LDS): 32 CUs per XCD
  if (per_xcd < 1) per_xcd = 1;
  while (per_xcd > 1 && 8 * (per_xcd - 1) >= g.ntiles) --per_xcd;
  dcn_bf16_wgrad_kernel<<<(unsigned)(8 * per_xcd * groups), kBT, kWgLdsB, mr3d_stream(s)>>>(x, offset, off_stride, dy, dw, g, b, per_xcd);
  return mr3d_launch_status();
}

extern "C" int mr3d_dcn_bf16_doffset(const mr3d_deform_conv3d_desc *d, int cout, const uint16_t *x, const uint16_t *offset,
                                     int off_stride, const uint16_t *dy, const uint16_t *wt_packed, uint16_t *doffset, int doff_stride,
                                     mr3d_stream_t s) {
  FGeo g;
  BoxGeo b;
  int st = fused_geo(d, cout, g);
  if (st == MR3D_OK) st = box_geo(g, b);
  if (st != MR3D_OK) return st;
  if (!x || !offset || !dy || !wt_packed || !doffset || off_stride < g.K * 3 || doff_stride < g.K * 3) return MR3D_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(wt_packed)) & 15) return MR3D_ERR_INVALID_ARG;
  if (allow_lds(dcn_bf16_doffset_kernel, kFwdLdsB) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const unsigned tiles = 8u * (unsigned)((g.ntiles + 7) / 8);
  dcn_bf16_doffset_kernel<<<tiles, kBT, kFwdLdsB, mr3d_stream(s)>>>(x, offset, off_stride, dy, wt_packed, doffset, doff_stride, g, b);
  return mr3d_launch_status();
}

// DeformConv3d, fused form for the FPN neck shape (Cin = Cout = 64, one deformable group, no modulation):
// forward, weight gradient and input/offset gradient WITHOUT the column matrix of csrc/deform_conv3d.hip.
//
// The column matrix of one 3x3x3 DeformConv3d over P2 of a 512x512x80 volume is 327,680 x 1,728 fp32 = 2.3 GB (7.6 GB on
// the 1.5x pathway); the im2col form writes it, reads it twice and writes its gradient.  Here a workgroup owns a tile of
// 128 consecutive output positions and walks the taps:
//     sample table  one thread per position: the 8 corner offsets + trilinear weights of (position, tap) -> LDS
//     gather        16 lanes x float4 = the 64 channels of one corner (256-B coalesced read); 4 positions per wave
//                   instruction; the weighted sum of the 8 corners = one row of the tap's column tile  -> LDS [128][64]
//     MFMA          v_mfma_f32_32x32x2_f32:  out[128 x 64] += columns[128 x 64] * W[tap][64 x 64]
// so the columns never leave the CU.  The sampling rule is the one of csrc/deform_conv3d.hip (the reference's 2D rule,
// mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:83-108,228: 0 unless -1 < p < dim on every axis, floor(), corners masked
// one by one); like the reference every IN-RANGE corner is read even when its weight is 0.
//
//   forward   y[pos][co]   = sum_tap col_tap[pos][:] . W[tap][:][co]
//   wgrad     dW[tap]      = col_tap^T dy       a workgroup keeps 3 taps x (64 x 64) in registers over its share of the
//                                               position tiles and adds them to dW once (fp32 atomics)
//   dgrad     dcol_tap     = dy W[tap]^T  (MFMA, tile in LDS), then per (position, tap):
//               dx[corner] += w_corner * dcol            (fp32 atomics, 256-B contiguous per corner)
//               doffset     = sum_c dcol[c] * d(sample[c]) / d(coordinate)     (16-lane shuffle reduction)
//
// K permutation: the MFMA A operand of k-step j holds A[row][2j + (lane >> 5)]; reading the column tile with one
// ds_read_b128 per four k-steps needs a lane's four values adjacent, so GEMM-k index r = 2j + h is mapped to channel
//     sigma(r) = 8 * (j >> 2) + 4 * h + (j & 3)
// and the packed weights (mr3d_deform_conv3d_pack_weights) are stored in that order.
#include "dcn_geo.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {


// table entry of the forward / weight-gradient kernels: off[8] (element offsets of channel 0; off[0] = -1 marks a sample
// outside the validity box), w[8]
__device__ __forceinline__ void fill_entry(const FGeo &g, const Coord &c, const TapIdx &ti, float oz, float oy, float ox, int *eo,
                                           float *ew) {
  const Sample s = sample_at(g, c, ti.kd, ti.kh, ti.kw, oz, oy, ox);
  int o[8];
  float w[8];
  const float wz[2] = {1.f - s.lz, s.lz}, wy[2] = {1.f - s.ly, s.ly}, wx[2] = {1.f - s.lx, s.lx};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = a * 4 + b * 2 + e;
        const bool v = s.any && s.vz[a] && s.vy[b] && s.vx[e];
        w[i] = v ? wz[a] * wy[b] * wx[e] : 0.f;
        o[i] = s.any ? (((s.n * g.D + s.z[a]) * g.H + s.y[b]) * g.W + s.x[e]) * kC : 0;
      }
  if (!s.any) o[0] = -1;
  *reinterpret_cast<int4 *>(eo) = make_int4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<int4 *>(eo + 4) = make_int4(o[4], o[5], o[6], o[7]);
  *reinterpret_cast<float4 *>(ew) = make_float4(w[0], w[1], w[2], w[3]);
  *reinterpret_cast<float4 *>(ew + 4) = make_float4(w[4], w[5], w[6], w[7]);
}

// the weighted corner sum of G consecutive 4-position groups of this wave: rows row0 + 4 * gi + (lane >> 4)
template <int G>
__device__ __forceinline__ void gather_rows(const float *__restrict__ x, const int *__restrict__ toff, const float *__restrict__ tw,
                                            float *__restrict__ tile, int row0, int lane) {
  const int p = lane >> 4, q = lane & 15;
  int o[G][8];
  float w[G][8];
  float4 v[G][8];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const int row = row0 + 4 * gi + p;
    const int4 oa = *reinterpret_cast<const int4 *>(toff + row * 8), ob = *reinterpret_cast<const int4 *>(toff + row * 8 + 4);
    const float4 wa = *reinterpret_cast<const float4 *>(tw + row * 8), wb = *reinterpret_cast<const float4 *>(tw + row * 8 + 4);
    o[gi][0] = oa.x; o[gi][1] = oa.y; o[gi][2] = oa.z; o[gi][3] = oa.w; o[gi][4] = ob.x; o[gi][5] = ob.y; o[gi][6] = ob.z; o[gi][7] = ob.w;
    w[gi][0] = wa.x; w[gi][1] = wa.y; w[gi][2] = wa.z; w[gi][3] = wa.w; w[gi][4] = wb.x; w[gi][5] = wb.y; w[gi][6] = wb.z; w[gi][7] = wb.w;
  }
#pragma unroll
  for (int gi = 0; gi < G; ++gi)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[gi][i] = *reinterpret_cast<const float4 *>(x + (i == 0 ? max(o[gi][0], 0) : o[gi][i]) + 4 * q);
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s.x = fmaf(w[gi][i], v[gi][i].x, s.x); s.y = fmaf(w[gi][i], v[gi][i].y, s.y);
      s.z = fmaf(w[gi][i], v[gi][i].z, s.z); s.w = fmaf(w[gi][i], v[gi][i].w, s.w);
    }
    if (o[gi][0] < 0) s = make_float4(0.f, 0.f, 0.f, 0.f);      // outside the validity box: 0 whatever voxel 0 holds
    *reinterpret_cast<float4 *>(tile + (row0 + 4 * gi + p) * kRow + 4 * q) = s;
  }
}

// ---------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void dcn_fwd_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                              const float *__restrict__ wf, float *__restrict__ y, FGeo g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tile = lds;                                   // [128][kRow]
  float *bs = tile + kM * kRow;                        // [64 r][64 co]
  int *toff = reinterpret_cast<int *>(bs + kC * kC);   // [2][128][8]
  float *tw = reinterpret_cast<float *>(toff + 2 * kM * 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int tile_id = xcd_contiguous(blockIdx.x, g.ntiles);
  if (tile_id < 0) return;
  const TileOrg org = tile_origin(g, tile_id);

  f32x16 acc[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;

  const Coord me = tile_pos(g, org, tid & (kM - 1));
  OffsetStream os;
  os.start(offset, me, g.K, tid < kM, 0);
  TapIdx ti;
  ti.set(g, 0);
  for (int tap = 0; tap < g.K; ++tap, ti.next(g)) {
    const int buf = tap & 1;
    if (tid < kM) {
      float oz, oy, ox;
      os.take(oz, oy, ox, tap + 1, g.K);
      fill_entry(g, me, ti, oz, oy, ox, toff + (buf * kM + tid) * 8, tw + (buf * kM + tid) * 8);
    }
    __syncthreads();
    // the tap's weights ride along with the gather (nobody reads bs between the two barriers)
    {
      const float4 *src = reinterpret_cast<const float4 *>(wf + (size_t)tap * kC * kC);
      float4 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = src[tid + i * kThreads];
      gather_rows<4>(x, toff + buf * kM * 8, tw + buf * kM * 8, tile, wave * 32, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) reinterpret_cast<float4 *>(bs)[tid + i * kThreads] = t[i];
      gather_rows<4>(x, toff + buf * kM * 8, tw + buf * kM * 8, tile, wave * 32 + 16, lane);
    }
    __syncthreads();
    const float *arow = tile + (wave * 32 + l31) * kRow + 4 * h;
    const float *bcol = bs + h * kC + l31;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 a = *reinterpret_cast<const float4 *>(arow + 8 * i);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * i + jj;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bcol[(2 * j) * kC + nb * 32], acc[nb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const Coord c = tile_pos(g, org, wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h);
      if (c.in) y[(size_t)c.pos * kC + nb * 32 + l31] = acc[nb][e];
    }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void dcn_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                                const float *__restrict__ dy, float *__restrict__ dw, FGeo g,
                                                                int nsplit, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tile = lds;                                   // [128 pos][kRow]
  int *toff = reinterpret_cast<int *>(tile + kM * kRow);
  float *tw = reinterpret_cast<float *>(toff + 2 * kM * 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int mb = wave >> 1, nb = wave & 1;             // this wave's (ci block, co block)
  // workgroup -> (chunk of consecutive tiles, tap group): the tap groups of one chunk gather the same voxels for
  // different taps, so they sit on the same XCD (blockIdx & 7) and walk the chunk together
  const int groups = (g.K + kTapsPerWg - 1) / kTapsPerWg;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tg = slot % groups, chunk = xcd * nsplit + slot / groups;          // nsplit = chunks per XCD
  const int nchunks = 8 * nsplit;
  const int per = (ntiles + nchunks - 1) / nchunks;
  const int t_begin = chunk * per, t_end = min(ntiles, t_begin + per);
  const int tap0 = tg * kTapsPerWg;

  f32x16 acc[kTapsPerWg];
#pragma unroll
  for (int t = 0; t < kTapsPerWg; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  int buf = 0;
  for (int tile_id = t_begin; tile_id < t_end; ++tile_id) {
    const TileOrg org = tile_origin(g, tile_id);
    const Coord me = tile_pos(g, org, tid & (kM - 1));
    OffsetStream os;
    os.start(offset, me, g.K, tid < kM, tap0);
    TapIdx tx;
    tx.set(g, tap0);
    float bfr[kM / 2];                                 // B[k = position 2j + h][n = co]
    int live = 0;
#pragma unroll
    for (int j = 0; j < kM / 2; ++j) {
      const Coord c = tile_pos(g, org, 2 * j + h);
      bfr[j] = c.in ? dy[(size_t)c.pos * kC + nb * 32 + l31] : 0.f;
      live |= bfr[j] != 0.f;
    }
    // The gradient that reaches a pyramid level in detector training is sparse -- it comes from 256 sampled anchors and from the
    // RoIAlign footprints of the sampled rois -- so most tiles carry dy == 0 and add nothing to dW: skip their 27 gather passes
    if (!__syncthreads_or(live)) continue;
#pragma unroll
    for (int t = 0; t < kTapsPerWg; ++t) {
      const int tap = tap0 + t;
      if (tap >= g.K) break;
      if (tid < kM) {
        float oz, oy, ox;
        os.take(oz, oy, ox, t + 1 < kTapsPerWg ? tap + 1 : g.K, g.K);
        fill_entry(g, me, tx, oz, oy, ox, toff + (buf * kM + tid) * 8, tw + (buf * kM + tid) * 8);
      }
      tx.next(g);
      __syncthreads();
#pragma unroll
      for (int part = 0; part < 4; ++part)
        gather_rows<2>(x, toff + buf * kM * 8, tw + buf * kM * 8, tile, wave * 32 + part * 8, lane);
      __syncthreads();
      const float *acol = tile + h * kRow + mb * 32 + l31;       // A[m = ci][k = position]
#pragma unroll
      for (int j = 0; j < kM / 2; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(acol[(2 * j) * kRow], bfr[j], acc[t], 0, 0, 0);
      buf ^= 1;
    }
  }
#pragma unroll
  for (int t = 0; t < kTapsPerWg; ++t) {
    const int tap = tap0 + t;
    if (tap >= g.K) break;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ci = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      atomicAdd(dw + ((size_t)tap * kC + ci) * kC + nb * 32 + l31, acc[t][e]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------- input / offset gradient
// table entry: off[8], (lz, ly, lx), flags (bit 0..5: vz0 vz1 vy0 vy1 vx0 vx1, bit 6: inside the validity box)
constexpr int kEntD = 12;

__device__ __forceinline__ void fill_entry_d(const FGeo &g, const Coord &cd, const TapIdx &ti, float oz, float oy, float ox, int *e) {
  const Sample s = sample_at(g, cd, ti.kd, ti.kh, ti.kw, oz, oy, ox);
  int o[8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        o[a * 4 + b * 2 + c] = s.any ? (((s.n * g.D + s.z[a]) * g.H + s.y[b]) * g.W + s.x[c]) * kC : 0;
  const int flags = (int)s.vz[0] | ((int)s.vz[1] << 1) | ((int)s.vy[0] << 2) | ((int)s.vy[1] << 3) | ((int)s.vx[0] << 4) |
                    ((int)s.vx[1] << 5) | ((int)s.any << 6);
  *reinterpret_cast<int4 *>(e) = make_int4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<int4 *>(e + 4) = make_int4(o[4], o[5], o[6], o[7]);
  *reinterpret_cast<int4 *>(e + 8) = make_int4(__float_as_int(s.lz), __float_as_int(s.ly), __float_as_int(s.lx), flags);
}

// WITH_DX = false: the offset gradient alone (the dx half then runs in the tiled kernel below): four position groups of
// corner reads in flight per step, reductions on DPP row shifts
template <bool WITH_DX>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void dcn_dgrad_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                                const float *__restrict__ dy, const float *__restrict__ wb,
                                                                float *__restrict__ dx, float *__restrict__ doffset, FGeo g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tile = lds;                                   // dcol [128][kRow]
  float *bs = tile + kM * kRow;                        // W[tap]^T: [64 r (co, permuted)][64 ci]
  int *tab = reinterpret_cast<int *>(bs + kC * kC);    // [2][128][kEntD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int p = lane >> 4, q = lane & 15;
  const int tile_id = xcd_contiguous(blockIdx.x, g.ntiles);
  if (tile_id < 0) return;
  const TileOrg org = tile_origin(g, tile_id);

  // A operand: this wave's 32 rows of dy, all 64 co, in the permuted k order -- registers for the whole tap loop
  float4 afr[8];
  {
    const Coord c = tile_pos(g, org, wave * 32 + l31);
    int live = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      afr[i] = c.in ? *reinterpret_cast<const float4 *>(dy + (size_t)c.pos * kC + 8 * i + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      live |= (afr[i].x != 0.f) | (afr[i].y != 0.f) | (afr[i].z != 0.f) | (afr[i].w != 0.f);
    }
    // dy == 0 on the whole tile (the usual case in detector training, see the weight gradient): dx gets nothing and doffset keeps
    // the zeros the caller put there
    if (!__syncthreads_or(live)) return;
  }

  const Coord me = tile_pos(g, org, tid & (kM - 1));
  OffsetStream os;
  os.start(offset, me, g.K, tid < kM, 0);
  TapIdx ti;
  ti.set(g, 0);
  for (int tap = 0; tap < g.K; ++tap, ti.next(g)) {
    const int buf = tap & 1;
    if (tid < kM) {
      float oz, oy, ox;
      os.take(oz, oy, ox, tap + 1, g.K);
      fill_entry_d(g, me, ti, oz, oy, ox, tab + (buf * kM + tid) * kEntD);
    }
    {
      const float4 *src = reinterpret_cast<const float4 *>(wb + (size_t)tap * kC * kC);
#pragma unroll
      for (int i = 0; i < 4; ++i) reinterpret_cast<float4 *>(bs)[tid + i * kThreads] = src[tid + i * kThreads];
    }
    __syncthreads();
    {
      f32x16 acc[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
      const float *bcol = bs + h * kC + l31;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float av[4] = {afr[i].x, afr[i].y, afr[i].z, afr[i].w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = 4 * i + jj;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bcol[(2 * j) * kC + nb * 32], acc[nb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          tile[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * kRow + nb * 32 + l31] = acc[nb][e];
    }
    __syncthreads();
    const int *tb = tab + buf * kM * kEntD;
    constexpr int GB = WITH_DX ? 2 : 4;              // position groups whose corner reads are in flight together
#pragma unroll 1
    for (int g0 = 0; g0 < 8; g0 += GB) {
      int o[GB][8];
      float lzyx[GB][3];
      int flg[GB];
      float4 v[GB][8];
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int row = wave * 32 + 4 * (g0 + u) + p;
        const int4 oa = *reinterpret_cast<const int4 *>(tb + row * kEntD), ob = *reinterpret_cast<const int4 *>(tb + row * kEntD + 4);
        const int4 fr = *reinterpret_cast<const int4 *>(tb + row * kEntD + 8);
        o[u][0] = oa.x; o[u][1] = oa.y; o[u][2] = oa.z; o[u][3] = oa.w; o[u][4] = ob.x; o[u][5] = ob.y; o[u][6] = ob.z; o[u][7] = ob.w;
        lzyx[u][0] = __int_as_float(fr.x); lzyx[u][1] = __int_as_float(fr.y); lzyx[u][2] = __int_as_float(fr.z);
        flg[u] = fr.w;
      }
#pragma unroll
      for (int u = 0; u < GB; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[u][i] = *reinterpret_cast<const float4 *>(x + o[u][i] + 4 * q);
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int row = wave * 32 + 4 * (g0 + u) + p;
        const float lz = lzyx[u][0], ly = lzyx[u][1], lx = lzyx[u][2];
        const int flags = flg[u];
        const bool any = (flags >> 6) & 1;
        const float4 gc = *reinterpret_cast<const float4 *>(tile + row * kRow + 4 * q);
        const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
        const float sg[2] = {-1.f, 1.f};
        float gz = 0.f, gy = 0.f, gx = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int i = a * 4 + b * 2 + c;
              const bool valid = any && ((flags >> a) & 1) && ((flags >> (2 + b)) & 1) && ((flags >> (4 + c)) & 1);
              const float dot = valid ? gc.x * v[u][i].x + gc.y * v[u][i].y + gc.z * v[u][i].z + gc.w * v[u][i].w : 0.f;
              gz = fmaf(sg[a] * wy[b] * wx[c], dot, gz);
              gy = fmaf(wz[a] * sg[b] * wx[c], dot, gy);
              gx = fmaf(wz[a] * wy[b] * sg[c], dot, gx);
              if constexpr (WITH_DX) {
                const float w = wz[a] * wy[b] * wx[c];
                if (valid && w != 0.f) {
                  float *d = dx + o[u][i] + 4 * q;
                  atomicAdd(d + 0, w * gc.x); atomicAdd(d + 1, w * gc.y); atomicAdd(d + 2, w * gc.z); atomicAdd(d + 3, w * gc.w);
                }
              }
            }
        // sum over the 16 lanes of the position (one DPP row): the total lands in lane 15 of the row
        auto row_sum = [](float t) {
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x111, 0xf, 0xf, true));
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x112, 0xf, 0xf, true));
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x114, 0xf, 0xf, true));
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x118, 0xf, 0xf, true));
          return t;
        };
        gz = row_sum(gz); gy = row_sum(gy); gx = row_sum(gx);
        const Coord rc = tile_pos(g, org, row);
        if (doffset && q == 15 && rc.in) {
          float *d = doffset + (size_t)rc.pos * (g.K * 3) + tap * 3;
          d[0] = gz; d[1] = gy; d[2] = gx;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- input gradient, tiled
// The direct kernel above runs at the chip's fp32 atomic rate (~1 TB/s of added bytes: 216 corner adds per position and
// channel = 18 GB for P2).  Tiled form of the dx half: a workgroup owns a 2 x 4 x 8 block of output positions and one
// 8-channel slice of dx; the voxels its samples can reach with |offset| < 1 -- the block's tap footprint plus one voxel
// on every side -- are an LDS accumulator; corners outside the box fall back to global atomics, and the box is added to dx
// once at the end (non-zero entries only): ~30x fewer global atomics.  (The offset gradient needs the corner VALUES of
// all 64 channels and no scatter: it comes from the direct kernel run without dx.)
//
// LDS accumulation is integer: ds_add_f32 serialises its 64 lanes on gfx950 (~190 cycles per instruction, against ~6 for
// ds_add_u32 / ds_max_f32 and ~14 for a read + compare-and-swap pair that then has to retry whenever two positions meet
// in a voxel, which neighbouring positions do all the time -- tools/lds_atomic_probe.hip).  A contribution v = w * dcol is
// scaled by 2^S, S = 40 - E with 2^E >= |dcol| for every value this workgroup can produce (Cauchy-Schwarz:
// |dcol[pos][tap][ci]| <= ||dy[pos]||_2 * max_(tap,ci) ||W[tap][:,ci]||_2), and split into two words
//     hi = floor(v 2^S / 2^20)     (|sum| < 2^31: at most 64 x 27 = 1728 < 2^11 contributions reach one voxel, each < 2^20)
//     lo = v 2^S - hi 2^20  in [0, 2^20]          (sum < 1728 * 2^20 < 2^31; carries are resolved at the end)
// each added with ds_add_u32.  The split is exact in fp32, so the sum is the exact sum of the contributions rounded to
// 2^(E-40) -- 16 bits below an fp32 ulp of the largest value -- independent of the order: the LDS part is deterministic.
//   dcol = dy[64 pos x 64 co] * [W[tap]^T | W[tap+1]^T][64 co x (8 + 8) ci]     v_mfma_f32_16x16x4_f32, two taps per pass
//   GEMM-k index kk = 4 * step + (lane >> 4)  <->  co = sigma16(kk) = 16 * (step >> 2) + 4 * (lane >> 4) + (step & 3)
//   (a lane's dy values of four steps are one float4); the third block of w_packed is stored in that order and a lane's
//   16 B values of a tap pair are loaded straight into registers one pair ahead.
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kTd = 2, kTh = 4, kTw = 8, kTm = kTd * kTh * kTw;     // 64 positions
constexpr int kCs = 8;                                              // channels per slice
constexpr int kEntT = 16;                                           // loc[8], w[8]
constexpr size_t kTiledLdsMax = 53 * 1024;                          // three workgroups per CU

struct TGeo {
  FGeo g;
  int tiles_d, tiles_h, tiles_w;
  int rz, ry, rx;                 // box extent
  const float *wnorm;             // max over (tap, ci) of ||W[tap][:, ci]||_2
};

// BF: offsets [position][off_stride] and dy are bf16 (the bf16 neck, csrc/deform_conv3d_bf16.hip); the arithmetic is the same fp32
template <bool BF>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
void dcn_dx_tiled_kernel(const void *__restrict__ offset_v, int off_stride, const void *__restrict__ dy_v, const float *__restrict__ wb16,
                         float *__restrict__ dx, TGeo t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[4];
  const FGeo &g = t.g;
  const int nbox = t.rz * t.ry * t.rx;
  int *bhi = reinterpret_cast<int *>(lds);                  // [nbox][8]
  int *blo = bhi + nbox * kCs;                              // [nbox][8]
  float *dt = reinterpret_cast<float *>(blo + nbox * kCs);  // [64][16]: columns 0-7 tap t, 8-15 tap t + 1
  int *tab = reinterpret_cast<int *>(dt + kTm * 16);        // [2 taps][64][kEntT]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int p8 = lane >> 3, c8 = lane & 7;
  int b = blockIdx.x;
  const int slice = b & 7; b >>= 3;
  const int tw_i = b % t.tiles_w; b /= t.tiles_w;
  const int th_i = b % t.tiles_h; b /= t.tiles_h;
  const int td_i = b % t.tiles_d; b /= t.tiles_d;
  const int n = b;
  const int od0 = td_i * kTd, oh0 = th_i * kTh, ow0 = tw_i * kTw;
  // box origin: first tap of the first position, one voxel lower
  const int bz0 = od0 * g.sd - g.pd - 1, by0 = oh0 * g.sh - g.ph - 1, bx0 = ow0 * g.sw - g.pw - 1;

  auto coord_of = [&](int m) {
    Coord c;
    c.n = n; c.ow = ow0 + (m & (kTw - 1)); c.oh = oh0 + ((m >> 3) & (kTh - 1)); c.od = od0 + (m >> 5);
    c.in = c.od < g.Do && c.oh < g.Ho && c.ow < g.Wo;
    c.pos = c.in ? (unsigned)(((n * g.Do + c.od) * g.Ho + c.oh) * g.Wo + c.ow) : 0u;
    return c;
  };
  // A operand: this wave's 16 positions x 64 co (permuted), registers for the whole kernel
  float4 afr[4];
  float nrm = 0.f;
  {
    const Coord c = coord_of(wave * 16 + l15);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      afr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c.in) {
        if constexpr (BF) {
          const uint2 v = *reinterpret_cast<const uint2 *>(static_cast<const bf16_t *>(dy_v) + (size_t)c.pos * kC + 16 * i + 4 * kq);
          afr[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                               __uint_as_float(v.y & 0xffff0000u));
        } else {
          afr[i] = *reinterpret_cast<const float4 *>(static_cast<const float *>(dy_v) + (size_t)c.pos * kC + 16 * i + 4 * kq);
        }
      }
      nrm += afr[i].x * afr[i].x + afr[i].y * afr[i].y + afr[i].z * afr[i].z + afr[i].w * afr[i].w;
    }
  }
  // ||dy[pos]||_2^2: the four kq lanes of a row, then the largest row of the workgroup
  nrm += __shfl_xor(nrm, 16);
  nrm += __shfl_xor(nrm, 32);
#pragma unroll
  for (int m = 8; m > 0; m >>= 1) nrm = fmaxf(nrm, __shfl_xor(nrm, m));
  if (lane == 0) red[wave] = nrm;
  for (int e = tid; e < 2 * nbox * kCs / 4; e += kThreads) reinterpret_cast<int4 *>(bhi)[e] = make_int4(0, 0, 0, 0);
  __syncthreads();
  const float bound = sqrtf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))) * t.wnorm[0] * 1.0001f;
  if (bound == 0.f) return;                                 // dy == 0 on the whole block
  const bool finite = bound < 3.0e38f;                      // non-finite dy / weights: every add goes straight to dx (fp32 atomics)
  int E = 0;
  if (finite) frexpf(bound, &E);                            // bound < 2^E
  const float up_hi = ldexpf(1.f, 20 - E);                  // v -> v 2^S / 2^20
  const float kLo = 1048576.f;

  const Coord me = coord_of(tid & (kTm - 1));               // the position whose table entries this thread fills
  const int my_tap = tid >> 6;                              // threads 0-63: first tap of the pair, 64-127: second
  typename std::conditional<BF, OffStream, OffsetStream>::type os;
  if constexpr (BF) os.start(static_cast<const bf16_t *>(offset_v), off_stride, me, tid < 2 * kTm && my_tap < g.K, my_tap);
  else os.start(static_cast<const float *>(offset_v), me, g.K, tid < 2 * kTm && my_tap < g.K, my_tap);
  TapIdx ti;
  ti.set(g, tid < 2 * kTm ? min(my_tap, g.K - 1) : 0);

  // B operand of a tap pair: B[kk = 4 * step + kq][n = l15]: n < 8 -> tap t, channel slice * 8 + n; else tap t + 1
  const int bt = l15 >> 3;
  const float *bsrc = wb16 + (size_t)kq * kC + slice * kCs + (l15 & 7);
  float bcur[16];
#pragma unroll
  for (int st = 0; st < 16; ++st) bcur[st] = bt < g.K ? bsrc[((size_t)bt * kC + 4 * st) * kC] : 0.f;

  for (int tap = 0; tap < g.K; tap += 2) {
    if (tid < 2 * kTm) {
      const int mt = tap + my_tap;
      float oz, oy, ox;
      os.take(oz, oy, ox, mt + 2, g.K);
      int lc[8];
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { lc[i] = -1; w[i] = 0.f; }
      const float pz = (float)(me.od * g.sd - g.pd + ti.kd * g.dd) + oz;
      const float py = (float)(me.oh * g.sh - g.ph + ti.kh * g.dh) + oy;
      const float px = (float)(me.ow * g.sw - g.pw + ti.kw * g.dw) + ox;
      if (mt < g.K && me.in && pz > -1.f && py > -1.f && px > -1.f && pz < (float)g.D && py < (float)g.H && px < (float)g.W) {
        const int z0 = (int)floorf(pz), y0 = (int)floorf(py), x0 = (int)floorf(px);
        const float lz = pz - (float)z0, ly = py - (float)y0, lx = px - (float)x0;
        const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int i = a * 4 + bb * 2 + c;
              const int z = z0 + a, y = y0 + bb, xx = x0 + c;
              const bool valid = (unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
              const float wi = wz[a] * wy[bb] * wx[c];
              if (valid && wi != 0.f) {
                const int bz = z - bz0, by = y - by0, bx = xx - bx0;
                const bool inb = finite && (unsigned)bz < (unsigned)t.rz && (unsigned)by < (unsigned)t.ry && (unsigned)bx < (unsigned)t.rx;
                w[i] = wi;
                lc[i] = inb ? ((bz * t.ry + by) * t.rx + bx) * kCs : -2 - (((n * g.D + z) * g.H + y) * g.W + xx);
              }
            }
      }
      ti.next(g); ti.next(g);
      int *e = tab + tid * kEntT;
      *reinterpret_cast<int4 *>(e) = make_int4(lc[0], lc[1], lc[2], lc[3]);
      *reinterpret_cast<int4 *>(e + 4) = make_int4(lc[4], lc[5], lc[6], lc[7]);
      *reinterpret_cast<float4 *>(e + 8) = make_float4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<float4 *>(e + 12) = make_float4(w[4], w[5], w[6], w[7]);
    }
    {
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av[4] = {afr[i].x, afr[i].y, afr[i].z, afr[i].w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], bcur[4 * i + jj], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) dt[(wave * 16 + 4 * kq + e) * 16 + l15] = acc[e];
    }
    {                                     // next pair's B operand: in flight during the adds below
      const int nt = tap + 2 + bt;
#pragma unroll
      for (int st = 0; st < 16; ++st) bcur[st] = nt < g.K ? bsrc[((size_t)nt * kC + 4 * st) * kC] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {         // (tap of the pair, half of the wave's 16 positions)
      const int tp = u >> 1;
      const int row = wave * 16 + 8 * (u & 1) + p8;
      const int *ent = tab + (tp * kTm + row) * kEntT;
      const int4 la = *reinterpret_cast<const int4 *>(ent), lb = *reinterpret_cast<const int4 *>(ent + 4);
      const float4 wa = *reinterpret_cast<const float4 *>(ent + 8), wb = *reinterpret_cast<const float4 *>(ent + 12);
      const int lc[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
      const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
      const float gc = dt[row * 16 + tp * 8 + c8];
      const float gs = gc * up_hi;
      bool far = false;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (lc[i] >= 0) {
          const float sv = w[i] * gs;                       // v 2^S / 2^20
          const float fh = floorf(sv);
          const int lo = (int)rintf((sv - fh) * kLo);       // exact difference, [0, 2^20]
          atomicAdd(bhi + lc[i] + c8, (int)fh);
          atomicAdd(blo + lc[i] + c8, lo);
        }
        far |= lc[i] < -1;
      }
      if (far) {                          // some corner beyond the box (|offset| >= 1): straight to dx
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (lc[i] < -1) atomicAdd(dx + (size_t)(-2 - lc[i]) * kC + slice * kCs + c8, w[i] * gc);
      }
    }
    __syncthreads();
  }
  const double down = (double)ldexpf(1.f, E - 40);
  for (int e = tid; e < nbox * kCs; e += kThreads) {
    const int hi = bhi[e], lo = blo[e];
    if ((hi | lo) != 0) {
      const float val = (float)(((double)hi * 1048576.0 + (double)lo) * down);
      const int c = e & (kCs - 1);
      int vox = e >> 3;
      const int lx = vox % t.rx; vox /= t.rx;
      const int ly = vox % t.ry;
      const int lz = vox / t.ry;
      atomicAdd(dx + ((((size_t)n * g.D + (bz0 + lz)) * g.H + (by0 + ly)) * g.W + (bx0 + lx)) * kC + slice * kCs + c, val);
    }
  }
}

__global__ void dcn_pack_weights_kernel(const float *__restrict__ w, int K, float *__restrict__ wf, float *__restrict__ wb,
                                        float *__restrict__ wb16) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;            // (tap, r, n)
  if (idx >= K * kC * kC) return;
  const int n = idx % kC, r = (idx / kC) % kC, tap = idx / (kC * kC);
  const int j = r >> 1, hh = r & 1;
  const int c = 8 * (j >> 2) + 4 * hh + (j & 3);                    // sigma(r)
  wf[idx] = w[((size_t)n * kC + c) * K + tap];                      // W[co = n][ci = sigma(r)][tap]
  wb[idx] = w[((size_t)c * kC + n) * K + tap];                      // W[co = sigma(r)][ci = n][tap]
  const int step = r >> 2, q = r & 3;
  const int c16 = 16 * (step >> 2) + 4 * q + (step & 3);            // sigma16(r)
  wb16[idx] = w[((size_t)c16 * kC + n) * K + tap];
}

// max over (tap, ci) of ||W[:, ci, tap]||_2 (the bound of the tiled input gradient's fixed-point scale)
__global__ void dcn_weight_norm_kernel(const float *__restrict__ w, int K, float *__restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;            // (tap, ci)
  if (idx >= K * kC) return;
  const int ci = idx % kC, tap = idx / kC;
  float s = 0.f;
  for (int co = 0; co < kC; ++co) {
    const float v = w[((size_t)co * kC + ci) * K + tap];
    s = fmaf(v, v, s);
  }
  atomicMax(reinterpret_cast<unsigned *>(out), __float_as_uint(sqrtf(s)));       // non-negative floats order like their bits
}


constexpr size_t kFwdLds = (size_t)(kM * kRow + kC * kC) * 4 + 2 * kM * 8 * 4 * 2;
constexpr size_t kWgLds = (size_t)(kM * kRow) * 4 + 2 * kM * 8 * 4 * 2;
constexpr size_t kDgLds = (size_t)(kM * kRow + kC * kC) * 4 + 2 * kM * kEntD * 4;


}  // namespace

extern "C" int mr3d_deform_conv3d_fused_supported(const mr3d_deform_conv3d_desc *d, int cout) {
  FGeo g;
  return fused_geo(d, cout, g) == MR3D_OK ? 1 : 0;
}

extern "C" int mr3d_deform_conv3d_pack_weights(const float *w, int cout, int cin, int taps, float *w_packed, mr3d_stream_t s) {
  if (!w || !w_packed || taps <= 0) return MR3D_ERR_INVALID_ARG;
  if (cout != kC || cin != kC) return MR3D_ERR_UNSUPPORTED;
  const int n = taps * kC * kC;
  dcn_pack_weights_kernel<<<(n + 255) / 256, 256, 0, mr3d_stream(s)>>>(w, taps, w_packed, w_packed + n, w_packed + 2 * (size_t)n);
  if (hipMemsetAsync(w_packed + 3 * (size_t)n, 0, 4, mr3d_stream(s)) != hipSuccess) return MR3D_ERR_LAUNCH;
  dcn_weight_norm_kernel<<<(taps * kC + 255) / 256, 256, 0, mr3d_stream(s)>>>(w, taps, w_packed + 3 * (size_t)n);
  return mr3d_launch_status();
}

extern "C" int mr3d_deform_conv3d_forward_fused(const mr3d_deform_conv3d_desc *d, int cout, const float *x, const float *offset,
                                                const float *w_packed, float *y, mr3d_stream_t s) {
  FGeo g;
  const int st = fused_geo(d, cout, g);
  if (st != MR3D_OK) return st;
  if (!x || !offset || !w_packed || !y) return MR3D_ERR_INVALID_ARG;
  if (allow_lds(dcn_fwd_kernel, kFwdLds) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const unsigned tiles = 8u * (unsigned)((g.ntiles + 7) / 8);                  // XCD-contiguous tile ranges (xcd_contiguous)
  dcn_fwd_kernel<<<tiles, kThreads, kFwdLds, mr3d_stream(s)>>>(x, offset, w_packed, y, g);
  return mr3d_launch_status();
}

extern "C" int mr3d_deform_conv3d_wgrad_fused(const mr3d_deform_conv3d_desc *d, int cout, const float *x, const float *offset,
                                              const float *dy, float *dw_packed, mr3d_stream_t s) {
  FGeo g;
  const int st = fused_geo(d, cout, g);
  if (st != MR3D_OK) return st;
  if (!x || !offset || !dy || !dw_packed) return MR3D_ERR_INVALID_ARG;
  if (allow_lds(dcn_wgrad_kernel, kWgLds) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const int groups = (g.K + kTapsPerWg - 1) / kTapsPerWg;
  int per_xcd = 64 / groups;                            // chunks per XCD: two workgroups on each of its 32 CUs
  if (per_xcd < 1) per_xcd = 1;
  while (per_xcd > 1 && 8 * (per_xcd - 1) >= g.ntiles) --per_xcd;
  dcn_wgrad_kernel<<<(unsigned)(8 * per_xcd * groups), kThreads, kWgLds, mr3d_stream(s)>>>(x, offset, dy, dw_packed, g, per_xcd,
                                                                                           g.ntiles);
  return mr3d_launch_status();
}

extern "C" int mr3d_deform_conv3d_dgrad_fused(const mr3d_deform_conv3d_desc *d, int cout, const float *x, const float *offset,
                                              const float *dy, const float *w_packed, float *dx, float *doffset, mr3d_stream_t s) {
  TGeo t;
  FGeo &g = t.g;
  const int st = fused_geo(d, cout, g);
  if (st != MR3D_OK) return st;
  if (!x || !offset || !dy || !w_packed || (!dx && !doffset)) return MR3D_ERR_INVALID_ARG;
  const size_t wn = (size_t)g.K * kC * kC;
  // tiled form when the reach of a 2x4x8 block of positions (+1 voxel each side) fits the LDS accumulator
  t.rz = (kTd - 1) * g.sd + (g.kd - 1) * g.dd + 3;
  t.ry = (kTh - 1) * g.sh + (g.kh - 1) * g.dh + 3;
  t.rx = (kTw - 1) * g.sw + (g.kw - 1) * g.dw + 3;
  const size_t lds_tiled = ((size_t)2 * t.rz * t.ry * t.rx * kCs + kTm * 16 + 2 * kTm * kEntT) * 4;
  t.wnorm = w_packed + 3 * wn;
  bool tiled = dx && lds_tiled <= kTiledLdsMax;
  if (allow_lds(dcn_dgrad_kernel<true>, kDgLds) != MR3D_OK || allow_lds(dcn_dgrad_kernel<false>, kDgLds) != MR3D_OK) return MR3D_ERR_LAUNCH;
  const unsigned tiles = 8u * (unsigned)((g.ntiles + 7) / 8);
  if (tiled) {
    t.tiles_d = (g.Do + kTd - 1) / kTd; t.tiles_h = (g.Ho + kTh - 1) / kTh; t.tiles_w = (g.Wo + kTw - 1) / kTw;
    const long long wgs = 8ll * g.N * t.tiles_d * t.tiles_h * t.tiles_w;
    if (wgs >= (1ll << 31)) return MR3D_ERR_UNSUPPORTED;
    if (allow_lds(dcn_dx_tiled_kernel<false>, lds_tiled) != MR3D_OK) return MR3D_ERR_LAUNCH;
    if (doffset) dcn_dgrad_kernel<false><<<tiles, kThreads, kDgLds, mr3d_stream(s)>>>(x, offset, dy, w_packed + wn, nullptr, doffset, g);
    dcn_dx_tiled_kernel<false><<<(unsigned)wgs, kThreads, lds_tiled, mr3d_stream(s)>>>(offset, 0, dy, w_packed + 2 * wn, dx, t);
    return mr3d_launch_status();
  }
  if (dx) dcn_dgrad_kernel<true><<<tiles, kThreads, kDgLds, mr3d_stream(s)>>>(x, offset, dy, w_packed + wn, dx, doffset, g);
  else dcn_dgrad_kernel<false><<<tiles, kThreads, kDgLds, mr3d_stream(s)>>>(x, offset, dy, w_packed + wn, nullptr, doffset, g);
  return mr3d_launch_status();
}

// the input gradient of the bf16 neck (csrc/deform_conv3d_bf16.hip): bf16 offsets [position][off_stride] and dy, fp32 weights
// (mr3d_deform_conv3d_pack_weights) and an fp32 dx the caller has zeroed -- the tiled LDS accumulator above
extern "C" int mr3d_dcn_bf16_dx(const mr3d_deform_conv3d_desc *d, int cout, const uint16_t *offset, int off_stride, const uint16_t *dy,
                                const float *w_packed, float *dx, mr3d_stream_t s) {
  TGeo t;
  FGeo &g = t.g;
  const int st = fused_geo(d, cout, g);
  if (st != MR3D_OK) return st;
  if (!offset || !dy || !w_packed || !dx || off_stride < g.K * 3) return MR3D_ERR_INVALID_ARG;
  const size_t wn = (size_t)g.K * kC * kC;
  t.rz = (kTd - 1) * g.sd + (g.kd - 1) * g.dd + 3;
  t.ry = (kTh - 1) * g.sh + (g.kh - 1) * g.dh + 3;
  t.rx = (kTw - 1) * g.sw + (g.kw - 1) * g.dw + 3;
  const size_t lds_tiled = ((size_t)2 * t.rz * t.ry * t.rx * kCs + kTm * 16 + 2 * kTm * kEntT) * 4;
  if (lds_tiled > kTiledLdsMax) return MR3D_ERR_UNSUPPORTED;
  t.wnorm = w_packed + 3 * wn;
  t.tiles_d = (g.Do + kTd - 1) / kTd; t.tiles_h = (g.Ho + kTh - 1) / kTh; t.tiles_w = (g.Wo + kTw - 1) / kTw;
  const long long wgs = 8ll * g.N * t.tiles_d * t.tiles_h * t.tiles_w;
  if (wgs >= (1ll << 31)) return MR3D_ERR_UNSUPPORTED;
  if (allow_lds(dcn_dx_tiled_kernel<true>, lds_tiled) != MR3D_OK) return MR3D_ERR_LAUNCH;
  dcn_dx_tiled_kernel<true><<<(unsigned)wgs, kThreads, lds_tiled, mr3d_stream(s)>>>(offset, off_stride, dy, w_packed + 2 * wn, dx, t);
  return mr3d_launch_status();
}

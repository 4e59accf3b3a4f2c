// Weight gradient of the 3x3x3 / stride-1 convolution as a TWO-dimensional Winograd form on the fp32 matrix cores:
// F(3,2) along H x F(3,4) along W -- the 3 x 3 (kh, kw) taps of one kd are the "outputs", a 2 x 4 patch of the output gradient
// is the "filter", the 4 x 6 input window is the same as in the forward kernel (conv3d_wino2d.hip).  Same contract as
// conv3d_wgrad.hip (Conv3dWgradArgs: NDHWC fp32 x and dy, dw [27][Cin][Cout] PRE-ZEROED and added to atomically, dbias);
// replaces the autograd weight gradients of the FPN output convs and the RPN conv on the large maps
// (mmdet/models/necks/fpn3d.py:56-64, anchor_heads/rpn_head_3d.py:19-24).
//
//   V = Bh^T d Bw          24 values per (patch, ci): the forward kernel's input transform (last row of Bh^T negated: d3 - d1)
//   E = Gh g Gw^T          24 values per (patch, co) from the 2 x 4 dy patch g: Gh = F(3,2)'s (g0, (g0+g1)/2, (g0-g1)/2, g1),
//                          Gw = F(3,4)'s c_f * (1, p_f, p_f^2, p_f^3) over the points 0, +-1, +-2, inf (c = 1/4, -1/6, -1/6, 1/24, 1/24, 1)
//   M[fh][fw][ci][co] += sum over patches V[fh][fw][patch][ci] * E[fh][fw][patch][co]      on the matrix cores: K = patches
//   dw[kd][kh][kw] = Ah^T M Aw   (Ah^T = F(3,2)'s (m0+m1+m2, m1-m2, m1+m2+m3), Aw^T rows (1,1,1,1,1,0), (0,1,-1,2,-2,0), (0,1,1,4,4,1))
// -- 24 MFMA products per patch and kd where the F(3,2)-along-W kernel of conv3d_wgrad.hip issues 48 and the direct form 72.
//
// v_mfma_f32_16x16x4_f32 with A[i = ci][k = patch], B[k = patch][j = co]: a wave accumulates the 24 product matrices of a 16 ci x
// 16 co block (96 registers).  A workgroup = (kd, 16-channel Cin tile, 64-wide Cout tile: one 16-column block per wave); it walks
// tiles of 2 x 8 x 8 outputs (16 patches = 4 k-steps): the x halo of the two input planes its kd needs and the dy tile are staged
// in LDS, V is formed once per tile (one item per thread) into LDS, E is formed in registers by the lane that owns (co, patch).
// The inverse transform (24 -> 9 taps) is applied once per workgroup before its atomics.
#include "conv3d.h"
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
constexpr int TD = 2, TH = 8, TW = 8;                  // output tile: 16 patches of 2 x 4
constexpr int HH = TH + 2, HW = TW + 2;                // halo rows / columns of one input plane
constexpr int CC = 16, ROW = CC + 1;                   // Cin tile, halo row pitch (floats)
constexpr int NF = 24;
constexpr int HALO = TD * HH * HW * ROW;               // floats
constexpr int DYP = 64 + 4;                            // dy row pitch (floats): 16-byte aligned rows
constexpr int DYL = TD * TH * TW * DYP;
constexpr int VP = NF * CC + 16;                       // V pitch per patch: lanes (ci, patch) of a k-step cover all 32 banks
constexpr int VL = 16 * VP;
constexpr int LDS_BYTES = (HALO + DYL + VL) * 4;

// SPARSE (MR3D_WGRAD_SPARSE_DY): dy tiles are tested and zero tiles skipped; the dense instance carries no test
template <bool SPARSE>
__global__ __launch_bounds__(kThreads, 2) void conv3d_wgrad_wino2d_kernel(Conv3dWgradArgs p, int tiles_d, int tiles_h, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *halo = lds, *dyl = lds + HALO, *vl = dyl + DYL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  // Dispatch order -> (tile sequence bx, Cin tile by, (kd, Cout tile) bz).  The 12 workgroup TYPES (by, bz) of one tile sequence read
  // the same dy tiles (and overlapping x planes); in the grid's natural order they are 128 ids apart -- three dispatch rounds --
  // and every round re-read its tiles from HBM (855 MiB per launch on the P2 map against 190 MB of operands).  Here consecutive
  // ids of one XCD (id % 8) are the types of ONE sequence: they are resident together, walk their tiles at the same pace, and the
  // first to touch a tile brings it into the XCD's L2 for the other eleven.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if ((gridDim.x & 7) == 0) {
    const int id = bx + (int)gridDim.x * (by + (int)gridDim.y * bz), types = (int)(gridDim.y * gridDim.z);
    const int xcd = id & 7, sq = id >> 3, type = sq % types;
    bx = (sq / types) * 8 + xcd; by = type % (int)gridDim.y; bz = type / (int)gridDim.y;
  }
  const int c0 = by * CC;
  const int cotiles = p.Cout / 64;
  const int kd = bz / cotiles, n0 = (bz % cotiles) * 64;
  const int co = n0 + wave * 16 + li;                    // B operand column of this lane
  const bool want_bias = p.dbias && kd == 0 && by == 0;

  f32x4 acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bias_sum = 0.f;

  const int ntiles = p.N * tiles_d * tiles_h * tiles_w;
  // staging pieces of a tile, per thread: x halo 2 x 10 x 10 positions x 4 float4 = 800 (4 slots, the last one partial), dy tile
  // 2 x 8 x 8 positions x 16 float4 = 2048 (8 slots).  The NEXT tile's pieces are fetched into registers in front of the
  // current tile's k-steps (their latency rides under 96 MFMAs per wave) and committed to LDS after them.
  constexpr int NX = (TD * HH * HW * (CC / 4) + kThreads - 1) / kThreads, NY = TD * TH * TW * 16 / kThreads;
  float4 xs[NX], ys[NY];
  // Addresses: this kernel is VALU-issue bound (two waves per SIMD, ~1200 vector instructions per tile and wave against 96
  // MFMAs), and a third of those were the 64-bit multiply chains of these 12 loads.  A load is now a per-tile SCALAR 64-bit
  // base plus a 32-bit lane offset that does not depend on the tile (x: one register per piece, hoisted; dy: one register, the
  // pieces differ by scalars), and tiles whose halo lies inside the volume (most) carry no per-lane bounds tests.
  int xrel[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int it = tid + j * kThreads, pos = it >> 2, q = it & 3;
    const int hw = pos % HW, hh = (pos / HW) % HH, hd = pos / (HW * HH);
    xrel[j] = ((hd * p.H + hh) * p.W + hw) * p.Cin + q * 4;
  }
  const int yrel0 = ((tid >> 7) * p.Wo + ((tid >> 4) & 7)) * p.Cout + (tid & 15) * 4;      // piece 0: od 0, oh tid >> 7, ow
  auto fetch = [&](int t, bool do_x, bool do_y) {
    int b = t;
    const int tw_i = b % tiles_w; b /= tiles_w;
    const int th_i = b % tiles_h; b /= tiles_h;
    const int td_i = b % tiles_d; b /= tiles_d;
    const int n = b;
    const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
    const int id0 = od0 + kd - p.pd, ih0 = oh0 - p.ph, iw0 = ow0 - p.pw;
    if (do_x) {
      const float *xb = p.x + ((((long long)n * p.D + id0) * p.H + ih0) * p.W + iw0) * p.Cin + c0;      // may point in front of x: masked
      const __amdgpu_buffer_rsrc_t xr = buffer_of(xb);   // (scalar descriptor: the loads below carry no vector address arithmetic)
      const bool inside = id0 >= 0 && id0 + TD <= p.D && ih0 >= 0 && ih0 + HH <= p.H && iw0 >= 0 && iw0 + HW <= p.W;   // (scalar)
      if (inside) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          xs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((j + 1) * kThreads <= TD * HH * HW * (CC / 4) || tid + j * kThreads < TD * HH * HW * (CC / 4))
            xs[j] = buffer_load4(xr, (unsigned)xrel[j] * 4u, 0u);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          const int it = tid + j * kThreads, pos = it >> 2;
          const int hw = pos % HW, hh = (pos / HW) % HH, hd = pos / (HW * HH);
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (it < TD * HH * HW * (CC / 4) && (unsigned)(id0 + hd) < (unsigned)p.D && (unsigned)(ih0 + hh) < (unsigned)p.H &&
              (unsigned)(iw0 + hw) < (unsigned)p.W)
            v = buffer_load4(xr, (unsigned)xrel[j] * 4u, 0u);
          xs[j] = v;
        }
      }
    }
    if (do_y) {
      const float *yb = p.dy + ((((long long)n * p.Do + od0) * p.Ho + oh0) * p.Wo + ow0) * p.Cout + n0;
      const __amdgpu_buffer_rsrc_t yr = buffer_of(yb);
      const bool inside = od0 + TD <= p.Do && oh0 + TH <= p.Ho && ow0 + TW <= p.Wo;
      const int row2 = 2 * p.Wo * p.Cout, plane = p.Ho * p.Wo * p.Cout;                 // pieces: oh += 2 per j, od = j >> 2
#pragma unroll
      for (int j = 0; j < NY; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inside || (od0 + (j >> 2) < p.Do && oh0 + (tid >> 7) + 2 * (j & 3) < p.Ho && ow0 + ((tid >> 4) & 7) < p.Wo))
          v = buffer_load4(yr, (unsigned)yrel0 * 4u, (unsigned)(((j & 3) * row2 + (j >> 2) * plane) * 4));      // the piece is a scalar offset
        ys[j] = v;
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int it = tid + j * kThreads;
      if (it < TD * HH * HW * (CC / 4)) {
        float *dd = halo + (it >> 2) * ROW + (it & 3) * 4;
        dd[0] = xs[j].x; dd[1] = xs[j].y; dd[2] = xs[j].z; dd[3] = xs[j].w;
      }
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int it = tid + j * kThreads;
      *reinterpret_cast<float4 *>(dyl + (it >> 4) * DYP + (it & 15) * 4) = ys[j];
    }
  };
  // A tile whose dy is zero adds nothing: in detector training the gradient that reaches the RPN conv and the FPN output convs is
  // sparse (256 sampled anchors, the RoIAlign footprints of the sampled rois), so most tiles are.  live[] is a workgroup flag per
  // tile parity: raised by any thread that fetched a non-zero dy value, read after the barrier every tile passes anyway.
  __shared__ int live[2];
  if (tid == 0) live[0] = 0;
  int tile_par = 0;
  if (bx < ntiles) fetch(bx, !SPARSE, true);      // sparse: only dy travels ahead; x is fetched for live tiles
  for (int t = bx; t < ntiles; t += gridDim.x, tile_par ^= 1) {
    __syncthreads();                                     // the previous tile's k-steps are done with dyl and V
    bool tile_live = true;                               // (workgroup-uniform)
    if constexpr (SPARSE) {
      bool nz = false;
#pragma unroll
      for (int j = 0; j < NY; ++j) nz |= (ys[j].x != 0.f) | (ys[j].y != 0.f) | (ys[j].z != 0.f) | (ys[j].w != 0.f);
      if (nz) live[tile_par] = 1;
      if (tid == 0) live[tile_par ^ 1] = 0;
      __syncthreads();
      tile_live = live[tile_par] != 0;
      if (tile_live) fetch(t, true, false);              // a dead tile's x halo is never read
    }
    if (tile_live) commit();
    __syncthreads();
    if (tile_live)
    // ---- V = Bh^T d Bw of (patch, ci): 16 x 16 items, one per thread -> vl[patch][f][ci]
    {
      const int ci = tid & 15, patch = tid >> 4;          // patch = plane * 8 + th2 * 2 + tw4
      const float *src = halo + ((((patch >> 3) * HH) + ((patch >> 1) & 3) * 2) * HW + (patch & 1) * 4) * ROW + ci;
      float d[4][6];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[i][j] = src[(i * HW + j) * ROW];
      float *dst = vl + patch * VP + ci;
#pragma unroll
      for (int fh = 0; fh < 4; ++fh) {
        float tt[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
          tt[j] = fh == 0 ? d[0][j] - d[2][j] : (fh == 1 ? d[1][j] + d[2][j] : (fh == 2 ? d[2][j] - d[1][j] : d[3][j] - d[1][j]));
        const float e = tt[4] - tt[2], o = tt[3] - tt[1];
        dst[(fh * 6 + 0) * CC] = 4.f * tt[0] - 5.f * tt[2] + tt[4];
        dst[(fh * 6 + 1) * CC] = (tt[3] + tt[4]) - 4.f * (tt[1] + tt[2]);
        dst[(fh * 6 + 2) * CC] = (tt[4] - tt[3]) + 4.f * (tt[1] - tt[2]);
        dst[(fh * 6 + 3) * CC] = e + 2.f * o;
        dst[(fh * 6 + 4) * CC] = e - 2.f * o;
        dst[(fh * 6 + 5) * CC] = 4.f * tt[1] - 5.f * tt[3] + tt[5];
      }
    }
    if (tile_live) __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x, !SPARSE, true);
    // ---- 4 k-steps of 4 patches: A = V[patch][f][ci = li] from LDS, B = E(co = li, patch) formed in registers
#pragma unroll 1
    for (int ks = 0; ks < (tile_live ? 4 : 0); ++ks) {    // not unrolled: four k-steps in flight at once spill
      const int patch = ks * 4 + lk;
      const float *va = vl + patch * VP + li;          // A fragments: one column of products (4 values) at a time, a column ahead
      // dy patch g[2][4] of (co, patch)
      const float *gsrc = dyl + ((((patch >> 3) * TH) + ((patch >> 1) & 3) * 2) * TW + (patch & 1) * 4) * DYP + wave * 16 + li;
      float g[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = gsrc[(i * TW + j) * DYP];
      if (want_bias) bias_sum += ((g[0][0] + g[0][1]) + (g[0][2] + g[0][3])) + ((g[1][0] + g[1][1]) + (g[1][2] + g[1][3]));
      // along W: Gw g WITHOUT its constants -- the column factors c_f = (1/4, -1/6, -1/6, 1/24, 1/24, 1) and the 1/2 of the two middle
      // rows of Gh are applied once, to the 24 accumulators, in front of the inverse transform (22 of the 52 vector instructions of
      // this transform per k-step were those multiplies, and they do not hide under the MFMAs)
      float r[2][6];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float g0 = g[i][0], g1 = g[i][1], g2 = g[i][2], g3 = g[i][3];
        const float s02 = g0 + g2, s13 = g1 + g3, a = g0 + 4.f * g2, bq = 2.f * g1 + 8.f * g3;
        r[i][0] = g0;
        r[i][1] = s02 + s13;
        r[i][2] = s02 - s13;
        r[i][3] = a + bq;
        r[i][4] = a - bq;
        r[i][5] = g3;
      }
      float ac[2][4];
#pragma unroll
      for (int fh = 0; fh < 4; ++fh) ac[0][fh] = va[(fh * 6) * CC];
#pragma unroll
      for (int fw = 0; fw < 6; ++fw) {                   // along H: Gh (2 -> 4), then the 4 MFMAs of this column of products
        if (fw + 1 < 6) {
#pragma unroll
          for (int fh = 0; fh < 4; ++fh) ac[(fw + 1) & 1][fh] = va[(fh * 6 + fw + 1) * CC];
        }
        const float e0 = r[0][fw], e3 = r[1][fw], e1 = e0 + e3, e2 = e0 - e3;
        acc[0 * 6 + fw] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[fw & 1][0], e0, acc[0 * 6 + fw], 0, 0, 0);
        acc[1 * 6 + fw] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[fw & 1][1], e1, acc[1 * 6 + fw], 0, 0, 0);
        acc[2 * 6 + fw] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[fw & 1][2], e2, acc[2 * 6 + fw], 0, 0, 0);
        acc[3 * 6 + fw] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[fw & 1][3], e3, acc[3 * 6 + fw], 0, 0, 0);
      }
    }
  }

  // ---- the constants of G that the k-steps left out
  {
    const float cw[6] = {0.25f, -1.f / 6.f, -1.f / 6.f, 1.f / 24.f, 1.f / 24.f, 1.f};
#pragma unroll
    for (int fh = 0; fh < 4; ++fh)
#pragma unroll
      for (int fw = 0; fw < 6; ++fw) {
        const float c = cw[fw] * ((fh == 1 || fh == 2) ? 0.5f : 1.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[fh * 6 + fw][r] *= c;
      }
  }
  // ---- dw[kd][kh][kw][ci][co] += Ah^T M Aw: accumulator element r of lane (li, lk) is (ci = 4 lk + r, co = li)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ci = c0 + lk * 4 + r;
    float z[4][3];                                       // along W: z[fh][kw]
#pragma unroll
    for (int fh = 0; fh < 4; ++fh) {
      const float m0 = acc[fh * 6][r], m1 = acc[fh * 6 + 1][r], m2 = acc[fh * 6 + 2][r], m3 = acc[fh * 6 + 3][r],
                  m4 = acc[fh * 6 + 4][r], m5 = acc[fh * 6 + 5][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      z[fh][0] = (m0 + s12) + s34; z[fh][1] = d12 + 2.f * d34; z[fh][2] = (s12 + 4.f * s34) + m5;
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float w0 = (z[0][kw] + z[1][kw]) + z[2][kw], w1 = z[1][kw] - z[2][kw], w2 = (z[1][kw] + z[2][kw]) + z[3][kw];
      float *dst = p.dw + ((size_t)(kd * 9 + kw) * p.Cin + ci) * p.Cout + co;
      atomicAdd(dst, w0);
      atomicAdd(dst + (size_t)3 * p.Cin * p.Cout, w1);
      atomicAdd(dst + (size_t)6 * p.Cin * p.Cout, w2);
    }
  }
  if (want_bias) {                                       // lanes lk = 0..3 of a column hold partial sums over their patches
    bias_sum += __shfl_xor(bias_sum, 16);
    bias_sum += __shfl_xor(bias_sum, 32);
    if (lk == 0) atomicAdd(p.dbias + co, bias_sum);
  }
}


// Round 4, measured and rejected: ALL THREE kd per workgroup (one staging of the dy tile and one E transform for the 72 products of
// the three kd, the four input planes of a tile transformed once): 2.0 instead of 4.8 vector instructions per MFMA, but the 3 x 24
// product matrices of a 16 x 64 block are 288 accumulator registers per lane with four waves (the compiler swapped them between
// the two register files around every MFMA) or 144 with eight waves and 113 KB of LDS -- either way ONE workgroup per CU, whose
// staging / transform / MFMA phases then have nothing to overlap with: P2 map 1170 us against 729 us for this kernel (two
// workgroups per CU), P3 191 against 127.  The CU's register file cannot hold two such blocks, so the form above stays.
}  // namespace

bool mr3d_conv3d_wgrad_wino2d_ok(const Conv3dWgradArgs &a) {
  return a.Cin % CC == 0 && a.Cout % 64 == 0 && !a.part && a.Do == a.D + 2 * a.pd - 2 && a.Ho == a.H + 2 * a.ph - 2 &&
         a.Wo == a.W + 2 * a.pw - 2 &&
         (long long)a.H * a.W * a.Cin < (1LL << 27) && (long long)a.Ho * a.Wo * a.Cout < (1LL << 27);      // byte offsets of a tile (two planes) below 2^31
}

int mr3d_conv3d_wgrad_wino2d(const Conv3dWgradArgs &a, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_wgrad_wino2d_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_wgrad_wino2d_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int tiles_d = (a.Do + TD - 1) / TD, tiles_h = (a.Ho + TH - 1) / TH, tiles_w = (a.Wo + TW - 1) / TW;
  const int ntiles = a.N * tiles_d * tiles_h * tiles_w;
  const int types = (a.Cin / CC) * 3 * (a.Cout / 64);
  int chunks = (512 * 3) / types;   // ~3 rounds of 512 resident workgroups (128 at 64 -> 64: 888 us; 170: 953)
  if (chunks > ntiles) chunks = ntiles;
  if (chunks < 1) chunks = 1;
  dim3 grid((unsigned)chunks, (unsigned)(a.Cin / CC), (unsigned)(3 * (a.Cout / 64)));
  if (a.sparse_dy) conv3d_wgrad_wino2d_kernel<true><<<grid, kThreads, LDS_BYTES, stream>>>(a, tiles_d, tiles_h, tiles_w);
  else conv3d_wgrad_wino2d_kernel<false><<<grid, kThreads, LDS_BYTES, stream>>>(a, tiles_d, tiles_h, tiles_w);
  return mr3d_launch_status();
}

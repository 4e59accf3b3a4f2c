// The stem's weight gradient taken straight from the gradient of the MAX-POOLED map, as a gather on the vector ALUs.
// (mmdet/models/backbones/resnet3d.py:397-409 conv1 / norm1 / relu, :471 nn.MaxPool3d(3, 2, 1).)
//
// Behind the stem sits a 3x3x3 / stride-2 max-pool, so the gradient of the stem's output has exactly ONE non-zero per
// pooled element: g[p][c] lands on window p's arg-max and nowhere else -- at most 1/8 of the elements of dy are non-zero,
// by construction, in every training step.  A dense GEMM over output positions (stem_wgrad_kernel: 172.6 GFLOP on the
// 1x3x80x512x512 volume, 1.5 ms of fp32 MFMAs) multiplies those zeros too; the non-zeros of a position sit in different
// channels with different arg-max positions, so no row of the GEMM can be dropped.  Written per pooled element the sum is
//
//   dW[c][kd][kh][kw][ci] = sum_p  relu'(pooled[p][c]) g[p][c] * x[ci][ z(p,c) - 3 + kd ][ 2 y(p,c) - 3 + kh ][ 2 x(p,c) - 3 + kw ]
//
// with (z, y, x)(p, c) = 2p - 1 + tap(p, c) the arg-max: 1029 multiply-adds per (p, c), 21.6 GFLOP in all (1/8), and the
// operand of all of them is the same scalar.  That is vector-ALU work with a wave-uniform multiplier:
//   * a wave owns an output channel c; its 343 x 3 accumulators are spread over the lanes (lane = (kd parity, kh, kw pair),
//     four depth pairs x three input channels = 12 float2 registers);
//   * the input halo of a tile of 4 x 4 x 8 pooled positions (15 x 23 x 40 floats of ONE input channel) sits in LDS; per
//     (p, c) a lane reads its four float2 at  item offset + lane offset + immediate  and issues four v_pk_fma_f32 with the
//     broadcast g.  Row stride 40 and plane stride 920 (= 24 mod 64) put the 56 active lanes of a ds_read_b64 on 64
//     distinct banks for every item offset (all even): no conflicts;
//   * (g, item offset) pairs are prepared once per tile ([channel][position] in LDS, read back as one broadcast
//     ds_read_b64 per item) -- a pooled element belongs to exactly one tile, so there is no halo of windows to visit.
// Bound: LDS bandwidth, 10 LDS cycles per (item, input channel) and CU.  The matrix pipe is idle, which is what the last
// weight gradients of the side queues need while this kernel closes the backward pass.
#include <atomic>
#include <mutex>
#include "conv3d.h"

namespace {

struct GatherGeo {
  static constexpr int TPD = 4, TPH = 4, TPW = 8;                  // pooled positions per tile
  static constexpr int TP = TPD * TPH * TPW;                       // 128
  static constexpr int PD = 2 * TPD + 7, PH = 4 * TPH + 7;         // 15 planes x 23 rows of one input channel
  static constexpr int ROW = 40, PLANE = PH * ROW, SLAB = PD * PLANE;
  static constexpr int PAIRS = ROW / 2;                            // float2 per row
  // column c' of a row holds input column 4*pw0 - 6 + c' (pw0 = the tile's first pooled column): the halo origin
  // 4*pw0 - 5 sits at c' = 1, so that tap pairs (kw = 2j - 1, 2j) start on even columns (8-byte aligned ds_read_b64)
  static_assert(4 * (TPW - 1) + 2 * 2 + 6 + 1 < ROW, "row holds the halo");
  static_assert(ROW % 16 == 8 && PLANE % 64 == (7 * ROW) % 64, "bank layout: (7 kd + kh) * ROW mod 64 enumerates the multiples of 8");
};
constexpr int kGatherThreads = 1024;                               // 16 waves = 16 output channels
constexpr int kGatherLds = GatherGeo::SLAB * 4 + 16 * GatherGeo::TP * 8;       // 71,584 bytes: two workgroups per CU

struct GatherArgs {
  const float *x;             // [N,3,D,H,W] planar
  const void *g;              // pooled gradient [N,Dp,Hp,Wp,16], fp32 or bf16
  const uint32_t *idx;        // [N,Dp,Hp,Wp,4]: four arg-max tap ids (kd*9 + kh*3 + kw) per word
  const void *pooled;         // pooled activation (ReLU mask) or null
  float *dw;                  // [343][4][16]
  float *dbias;               // [16] or null
  float *part;                // [grid][343][3][16] or null (atomics)
  int N, D, H, W, Dp, Hp, Wp;
  unsigned *queue;            // 8 tile-queue heads + 1 finished-workgroup count, 0 at launch: the tail of the caller's workspace (zeroed by a
                              // memset in the launch's own stream) or, without a workspace, one of the library's 64 rotating slots
};

__device__ __forceinline__ f32x2 buffer_load2(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff_bytes, (int)soff_bytes, 0));
}

// per slot: 8 tile-queue heads + 1 finished-workgroup count, all 0 between launches
__device__ unsigned g_stem_gather_queues[64 * 16];
constexpr int kGatherMaxDevices = 16;

template <bool BF16>
__global__ __launch_bounds__(kGatherThreads, 8) void stem_wgrad_gather_kernel(GatherArgs p, int tiles_d, int tiles_h, int tiles_w) {
  using G = GatherGeo;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *slab = lds;
  uint2 *tab = reinterpret_cast<uint2 *>(lds + G::SLAB);           // [16 channels][TP] (g bits, item byte offset)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = output channel
  const size_t vox = (size_t)p.D * p.H * p.W;
  const int num_tiles = p.N * tiles_d * tiles_h * tiles_w;

  // lane -> (kd parity, kh, kw pair); lanes 56..63 repeat lane 32's address (a broadcast, no bank of their own)
  const int l = lane < 56 ? lane : 32;
  const int kdl = l / 28, kh = (l % 28) >> 2, j = l & 3;
  const unsigned lane_byte = (unsigned)(kdl * G::PLANE + kh * G::ROW + 2 * j) * 4u;

  f32x2 acc[3][4];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[ci][i] = f32x2{0.f, 0.f};
  float bsum = 0.f;

  // staging ownership: thread (half, r, k) loads the float2 k of row r of every second plane
  constexpr int PER = G::PH * G::PAIRS;                            // 460 float2 per plane
  static_assert(2 * PER <= kGatherThreads, "two planes per pass");
  const int s_half = tid >= PER ? 1 : 0, s_t = tid - s_half * PER;
  const int s_r = s_t / G::PAIRS, s_k = s_t - s_r * G::PAIRS;
  const bool s_lane = tid < 2 * PER;

  // tiles are pulled from eight queues: the depth-fastest tile order cut into one contiguous range per XCD (stem.hip)
  __shared__ int s_tile;
  unsigned *queue = p.queue;
  const int per_xcd = (num_tiles + 7) >> 3;
  const int home = blockIdx.x & 7;
  auto range = [&](int x, int &lo) { lo = x * per_xcd; return max(min(lo + per_xcd, num_tiles) - lo, 0); };
  auto take = [&](unsigned ticket) {
    int lo, cnt = range(home, lo);
    if (ticket < (unsigned)cnt) return lo + (int)ticket;
    for (int a = 1; a < 8; ++a) {
      cnt = range((home + a) & 7, lo);
      if (cnt > 0) {
        const unsigned v = atomicAdd(queue + ((home + a) & 7), 1u);
        if (v < (unsigned)cnt) return lo + (int)v;
      }
    }
    return -1;
  };
  unsigned ticket = 0;
  if (tid == 0) s_tile = take(atomicAdd(queue + home, 1u));
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane(s_tile);
  while (tile >= 0) {
    int seq = tile;
    const int td = seq % tiles_d; seq /= tiles_d;
    const int tw = seq % tiles_w; seq /= tiles_w;
    const int th = seq % tiles_h;
    const int n = seq / tiles_h;
    const int pd0 = td * G::TPD, ph0 = th * G::TPH, pw0 = tw * G::TPW;
    if (tid == 0) ticket = atomicAdd(queue + home, 1u);

    // (g, offset) of the tile's 128 x 16 items
    if (tid < G::TP * 4) {
      const int pp = tid >> 2, q4 = tid & 3;
      const int pwl = pp & 7, phl = (pp >> 3) & 3, pdl = pp >> 5;
      const int od = pd0 + pdl, oh = ph0 + phl, ow = pw0 + pwl;
      float gv[4] = {0.f, 0.f, 0.f, 0.f};
      uint32_t ix = 0;
      if (od < p.Dp && oh < p.Hp && ow < p.Wp) {
        const size_t o = ((((size_t)n * p.Dp + od) * p.Hp + oh) * p.Wp + ow) * 4 + q4;
        ix = p.idx[o];
        float pv[4] = {1.f, 1.f, 1.f, 1.f};
        if (BF16) {
          const uint2 q = reinterpret_cast<const uint2 *>(p.g)[o];
          gv[0] = __uint_as_float(q.x << 16); gv[1] = __uint_as_float(q.x & 0xffff0000u);
          gv[2] = __uint_as_float(q.y << 16); gv[3] = __uint_as_float(q.y & 0xffff0000u);
          if (p.pooled) {
            const uint2 a = reinterpret_cast<const uint2 *>(p.pooled)[o];
            pv[0] = __uint_as_float(a.x << 16); pv[1] = __uint_as_float(a.x & 0xffff0000u);
            pv[2] = __uint_as_float(a.y << 16); pv[3] = __uint_as_float(a.y & 0xffff0000u);
          }
        } else {
          const float4 q = reinterpret_cast<const float4 *>(p.g)[o];
          gv[0] = q.x; gv[1] = q.y; gv[2] = q.z; gv[3] = q.w;
          if (p.pooled) {
            const float4 a = reinterpret_cast<const float4 *>(p.pooled)[o];
            pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(pv[r] > 0.f)) gv[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tap = (ix >> (8 * r)) & 255;
        const int a = tap / 9, b = (tap / 3) % 3, c = tap % 3;
        const unsigned off = (unsigned)((2 * pdl + a) * G::PLANE + (4 * phl + 2 * b) * G::ROW + 4 * pwl + 2 * c) * 4u;
        tab[(q4 * 4 + r) * G::TP + pp] = make_uint2(__float_as_uint(gv[r]), off);
      }
    }

#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci) {
      if (ci) __syncthreads();                         // every wave is done with the previous channel's slab
      {
        const float *xc = p.x + ((size_t)n * 3 + ci) * vox;
        const int ih = 4 * ph0 - 5 + s_r, iw = 4 * pw0 - 6 + 2 * s_k;
        const bool mine = s_lane && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;      // W even: a pair is in or out
        const unsigned voff = mine ? (unsigned)(ih * p.W + iw) : 0u;
        const int plane = p.H * p.W;
        const __amdgpu_buffer_rsrc_t xr = buffer_of(xc);
        f32x2 v[(G::PD + 1) / 2];
#pragma unroll
        for (int k = 0; k < (G::PD + 1) / 2; ++k) {
          const int pl = 2 * k + s_half;
          const int id = 2 * pd0 - 4 + pl;
          v[k] = f32x2{0.f, 0.f};
          if (pl < G::PD && (unsigned)id < (unsigned)p.D && mine) v[k] = buffer_load2(xr, voff * 4u, (unsigned)(id * plane) * 4u);
        }
        if (s_lane) {
          float *d = slab + s_half * G::PLANE + s_r * G::ROW + 2 * s_k;
#pragma unroll
          for (int k = 0; k < (G::PD + 1) / 2; ++k)
            if (2 * k + s_half < G::PD) *reinterpret_cast<f32x2 *>(d + 2 * k * G::PLANE) = v[k];
        }
      }
      __syncthreads();                                 // slab (and, for ci == 0, the item table) complete
      const uint2 *items = tab + wave * G::TP;
      const char *base = reinterpret_cast<const char *>(slab) + lane_byte;
      f32x2 a0 = acc[0][0], a1 = acc[0][1], a2 = acc[0][2], a3 = acc[0][3];
      // Software pipeline, three items deep: item it's multiply-adds run while the four reads of item it + 1 and the table
      // entry of item it + 2 are in flight (the compiler, left alone, waits for every read right behind its issue; the
      // sched_barriers pin the order, its own counted s_waitcnt lgkmcnt does the rest).
      auto rd = [&](const char *q, int i) { return *reinterpret_cast<const f32x2 *>(q + 2 * i * G::PLANE * 4); };
      // the wave's 128 (g, offset) entries are read ONCE into two register pairs (entry lane and lane + 64 per lane) and come back
      // per item through v_readlane: no LDS read per item for the table (2 of the 10 LDS cycles per item and input channel)
      const uint2 ea = items[lane], eb = items[lane + 64];
      if (ci == 0) bsum += __uint_as_float(ea.x) + __uint_as_float(eb.x);      // bias gradient: per-lane partial sums
      auto g_of = [&](int it) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)(it < 64 ? ea.x : eb.x), it & 63)); };
      auto o_of = [&](int it) { return (unsigned)__builtin_amdgcn_readlane((int)(it < 64 ? ea.y : eb.y), it & 63); };
      f32x2 v0, v1, v2, v3;
      {
        const char *q = base + o_of(0);
        v0 = rd(q, 0); v1 = rd(q, 1); v2 = rd(q, 2); v3 = rd(q, 3);
      }
#pragma unroll
      for (int it = 0; it < G::TP; ++it) {
        const char *qn = base + o_of(it + 1 < G::TP ? it + 1 : G::TP - 1);
        const f32x2 n0 = rd(qn, 0), n1 = rd(qn, 1), n2 = rd(qn, 2), n3 = rd(qn, 3);
        __builtin_amdgcn_sched_barrier(0);
        const float g = g_of(it);
        const f32x2 gg = f32x2{g, g};
        a0 = __builtin_elementwise_fma(v0, gg, a0);
        a1 = __builtin_elementwise_fma(v1, gg, a1);
        a2 = __builtin_elementwise_fma(v2, gg, a2);
        a3 = __builtin_elementwise_fma(v3, gg, a3);
        __builtin_amdgcn_sched_barrier(0);
        v0 = n0; v1 = n1; v2 = n2; v3 = n3;
      }
      // rotate the accumulator groups: the body always works on group 0 (no dynamic register indexing)
      acc[0][0] = acc[1][0]; acc[0][1] = acc[1][1]; acc[0][2] = acc[1][2]; acc[0][3] = acc[1][3];
      acc[1][0] = acc[2][0]; acc[1][1] = acc[2][1]; acc[1][2] = acc[2][2]; acc[1][3] = acc[2][3];
      acc[2][0] = a0; acc[2][1] = a1; acc[2][2] = a2; acc[2][3] = a3;
    }
    if (tid == 0) s_tile = take(ticket);
    __syncthreads();                                   // also: every wave is done with this tile's slab and table
    tile = __builtin_amdgcn_readfirstlane(s_tile);
  }
  if (tid == 0) {                                      // the last workgroup to run dry re-arms the queues
    const unsigned done = atomicAdd(queue + 8, 1u);
    if (done == gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 9; ++x) atomicExch(queue + x, 0u);
    }
  }

  // lane (kdl, kh, j), register i: kd = 2i + kdl, taps kw = 2j - 1 (.x, none for j = 0) and 2j (.y).  The block leaves through LDS as
  // [tap][ci][16 channels] (the slab is free now): a wave's own values are 64 bytes apart in that layout, and stored straight from
  // the registers the 16 waves' 4-byte pieces of a line reached HBM as separate partial writes (217 MiB written per launch for
  // 34 MB of partial sums).
  float *stage = lds;                                  // 343 * 3 * 16 floats = 65,856 bytes <= slab + table
  static_assert(343 * 3 * 16 * 4 <= kGatherLds, "the partial block fits the workgroup's LDS");
  if (lane < 56) {
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kd = 2 * i + kdl;
        if (kd > 6) continue;
        const int t = (kd * 7 + kh) * 7 + 2 * j;
        if (j > 0) stage[((t - 1) * 3 + ci) * 16 + wave] = acc[ci][i].x;
        stage[(t * 3 + ci) * 16 + wave] = acc[ci][i].y;
      }
  }
  __syncthreads();
  {
    constexpr int PER4 = 343 * 3 * 4;                  // float4 of the block: element e = (tap * 3 + ci) * 4 + channel quad
    float4 *part4 = p.part ? reinterpret_cast<float4 *>(p.part + (size_t)blockIdx.x * (343 * 3 * 16)) : nullptr;
    for (int e = tid; e < PER4; e += kGatherThreads) {
      const float4 v = reinterpret_cast<const float4 *>(stage)[e];
      if (part4) {
        part4[e] = v;
      } else {
        const int tc = e >> 2;
        float *d = p.dw + ((size_t)(tc / 3) * 4 + tc % 3) * 16 + (e & 3) * 4;
        atomicAdd(d, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
      }
    }
  }
  if (p.dbias) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bsum += __shfl_xor(bsum, o);
    if (lane == 0) atomicAdd(p.dbias + wave, bsum);
  }
}

// adds the workgroups' partial blocks [nparts][343][3][16] into dw [343][4][16]: 32 slices of the partials x 4116 float4
// columns, every thread sums its slice with all loads in flight and adds the result atomically (dw is pre-zeroed by the caller).
// (mr3d_reduce_partials walks all 512 blocks serially in 65 workgroups: 38 us.)
__global__ __launch_bounds__(256) void stem_gather_reduce_kernel(const float *__restrict__ part, int nparts, float *__restrict__ dw) {
  constexpr int PER4 = 343 * 3 * 4;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= PER4) return;
  const int per_slice = (nparts + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per_slice, b1 = min(b0 + per_slice, nparts);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int b = b0; b < b1; ++b) {
    const float4 v = reinterpret_cast<const float4 *>(part)[(size_t)b * PER4 + e];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int q = e & 3, tc = e >> 2;                    // tc = tap * 3 + ci
  float *d = dw + ((size_t)(tc / 3) * 4 + tc % 3) * 16 + q * 4;
  atomicAdd(d, s.x); atomicAdd(d + 1, s.y); atomicAdd(d + 2, s.z); atomicAdd(d + 3, s.w);
}

}  // namespace

bool mr3d_stem_wgrad_gather_ok(int d, int h, int w) {
  return (w & 1) == 0 && (long long)d * h * w < (1LL << 28) && g_mr3d_tuning[MR3D_TUNE_STEM_WGRAD_DENSE] == 0;
}

// x [N,3,D,H,W]; g / pooled [N,Dp,Hp,Wp,16] fp32 (bf16 = 0) or bf16; idx [N,Dp,Hp,Wp,4]; dw [343][4][16] and dbias PRE-ZEROED
int mr3d_stem_wgrad_gather(const float *x, const void *g, const uint32_t *idx, const void *pooled, int bf16, float *dw, float *dbias,
                           int n, int d, int h, int w, float *workspace, long long workspace_bytes, hipStream_t stream) {
  using G = GatherGeo;
  GatherArgs a;
  a.x = x; a.g = g; a.idx = idx; a.pooled = pooled; a.dw = dw; a.dbias = dbias;
  a.N = n; a.D = d; a.H = h; a.W = w;
  const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
  a.Dp = (d - 1) / 2 + 1; a.Hp = (ho - 1) / 2 + 1; a.Wp = (wo - 1) / 2 + 1;
  // per device, once: the > 64 KiB LDS attribute of both instances and the CU count (hipGetDeviceProperties must not run inside a
  // stream capture: a plan's first, eager run comes before any capture)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kGatherMaxDevices) dev = 0;
  static std::once_flag once[kGatherMaxDevices];
  static int cus_of[kGatherMaxDevices];
  static bool attr_ok[kGatherMaxDevices];
  static unsigned *slots_of[kGatherMaxDevices];
  std::call_once(once[dev], [dev]() {
    attr_ok[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(stem_wgrad_gather_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kGatherLds) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void *>(stem_wgrad_gather_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kGatherLds) == hipSuccess;
    hipDeviceProp_t prop;
    cus_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    if (hipGetSymbolAddress(reinterpret_cast<void **>(&slots_of[dev]), HIP_SYMBOL(g_stem_gather_queues)) != hipSuccess) slots_of[dev] = nullptr;
    (void)hipGetLastError();
  });
  if (!attr_ok[dev]) return MR3D_ERR_LAUNCH;
  const int cus = cus_of[dev];
  const int td = (a.Dp + G::TPD - 1) / G::TPD, th = (a.Hp + G::TPH - 1) / G::TPH, tw = (a.Wp + G::TPW - 1) / G::TPW;
  const int tiles = n * td * th * tw;
  // Two 1024-thread workgroups per CU take every wave slot of the chip, and the workgroups are persistent: whatever is queued beside
  // this kernel -- the gradient finishing of layer1 / layer2 (side queue), a collective -- starts with its last tile.  That is the
  // cheaper order at N = 1 (48 free slots: the kernel 6 % longer, the 75 us of finishing behind it 50 us shorter, step 9.09 ->
  // 9.04 ms; with the finishing of layer3.. moved to the middle of the backward pass the rest is too short to matter), and the
  // gradient exchange does not depend on it: its large bucket is final in mid-backward (extract_feat.MARK_GRADS_EARLY).
  // MR3D_TUNE_STEM_GATHER_SPARE leaves n slots free.
  const int spare = g_mr3d_tuning[MR3D_TUNE_STEM_GATHER_SPARE] > 0 ? g_mr3d_tuning[MR3D_TUNE_STEM_GATHER_SPARE] : 0;
  int slots = 2 * cus - spare;
  if (slots < cus) slots = cus;
  int grid = tiles < slots ? tiles : slots;
  if (grid >= 8) grid &= ~7;
  const long long part_bytes = (long long)grid * 343 * 3 * 16 * 4;
  a.part = (workspace && workspace_bytes >= part_bytes) ? workspace : nullptr;
  if (workspace && workspace_bytes >= part_bytes + 64) {
    // the tile queues live behind the partial sums in the caller's workspace and are zeroed in this launch's own stream: no state
    // shared with any other launch (a replayed hipGraph and an eager launch used to be able to draw the same rotating slot)
    a.queue = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(workspace) + part_bytes);
    if (hipMemsetAsync(a.queue, 0, 64, stream) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
  } else {
    static std::atomic<unsigned> next_slot{0};           // no workspace: one of 64 self-re-arming slots (launches on ONE stream never overlap)
    if (!slots_of[dev]) return MR3D_ERR_LAUNCH;
    a.queue = slots_of[dev] + 16 * (next_slot.fetch_add(1) & 63u);
  }
  if (bf16) stem_wgrad_gather_kernel<true><<<(unsigned)grid, kGatherThreads, kGatherLds, stream>>>(a, td, th, tw);
  else stem_wgrad_gather_kernel<false><<<(unsigned)grid, kGatherThreads, kGatherLds, stream>>>(a, td, th, tw);
  const int st = mr3d_launch_status();
  if (st != MR3D_OK || !a.part) return st;
  stem_gather_reduce_kernel<<<dim3((343 * 3 * 4 + 255) / 256, 32), 256, 0, stream>>>(a.part, grid, dw);
  return mr3d_launch_status();
}

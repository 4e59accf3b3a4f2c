// ResNet3D stem on the bf16 matrix cores (BASELINE configs[4]): Conv3d(3, 16, kernel 7, stride (1,2,2), padding 3) +
// folded BN + ReLU forward, and its weight gradient (mmdet/models/backbones/resnet3d.py:397-409,467-469).
// Input: the reference's planar fp32 volume [N,3,D,H,W] (converted to bf16 while it is staged); output / dy: bf16
// NDHWC [N,D,H/2,W/2,16]; fp32 accumulation (v_mfma_f32_16x16x32_bf16: N = 16 output channels is exactly one tile).
//
// Forward.  GEMM K is walked as 49 (kd,kh) steps of 32 = 4 channel slots x 8 kw slots (3 x 7 real: 66 % of the issued
// MFMA work is useful, still ~4x the fp32 kernel's rate).  The weights are the A operand (pre-packed [kd*7+kh][c4][cout][kw8] bf16,
// 50 KB, staged in LDS once per workgroup), the input the B operand: a lane's fragment -- 8 consecutive kw taps of one channel
// for one output position -- is 8 CONSECUTIVE input columns starting at 2*ow - 3, 16 bytes of the planar bf16 slab
// [c][depth slot][h][w] at a 4-byte-aligned address.  A wave owns two output depths x four consecutive output rows of the
// 8 x 8 x 16 tile: the fragment of input plane pl and row j feeds depth t with tap kd = pl - t and row i with tap kh = j - 2i, so
// ONE fragment read serves up to eight MFMAs (104 reads for 392 MFMAs) against the seven kh weight fragments of two kd taps,
// which wait in registers.  The accumulator holds 4 consecutive output channels of one position
// per lane: 8-byte stores, 512 contiguous bytes per store instruction.
//
// Weight gradient.  GEMM M = (channel, kd, kh, kw) rows, N = 16 output channels, K = positions, 32 consecutive ow per step.
// The slab keeps the even and the odd input columns as separate images (iw = 2*ow + kw - 3 has the parity of kw + 1): the
// fragment of tap kw -- 8 consecutive positions -- is 8 consecutive bf16 of an image, starting kw >> 1 elements into the
// ALIGNED 24-byte window of its (channel, kd, kh) group.  A lane serves one group per row tile: one ds_read_b128 + one
// ds_read_b64 deliver the fragments of all 4 (even kw) or 3 (odd kw) taps, the odd starts are shifted out with v_perm --
// 24 bytes of LDS per 3-4 MFMAs instead of 16 per MFMA (the first form, a 16-byte read per (tap, lane) from four shifted
// images, was LDS-bound: 59 % LDS-active, 41 % of it bank conflicts, 0.81 ms).  B = dy [position][16] read with the
// transposing LDS read (ds_read_b64_tr_b16).  8 waves split the 70 (row tile, kw) accumulators (+ one tile of ones: sum_p dy);
// persistent workgroups, fp32 atomics at the end into the generic packed layout [343][4][16] that mr3d_finish_grads consumes.
//
// Both kernels walk depth-adjacent tiles and keep the slab's depth planes in a ring: a tile stages only its 4 new planes (12 of
// the 30 (channel, depth) planes), fetched with buffer loads (zeros past the borders from the hardware: no branch per load)
// that stay in flight during the MFMA phase of the tile before.  512x512x80: forward 0.59 -> 0.34 ms, weight gradient (with the
// max-pool backward folded in) 0.81 -> 0.41 ms.
#include <atomic>
#include "mr3d_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
using bf16_t = uint16_t;
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

__device__ __forceinline__ bf16_t f2bf(float f) {          // round to nearest even, NaN stays NaN: v_cvt_pk_bf16_f32
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}

// 16 bytes at 4-byte alignment: the compiler emits dword-granular LDS reads (ds_read2_b32 pairs).  A 16-byte read whose address
// is only 2-byte aligned does execute on gfx950, but the DS unit then stays busy ~8x longer (SQ_LDS_IDX_ACTIVE: 79 % of the
// weight-gradient kernel's run time) -- every fragment address in this file is an even bf16 element.
struct __attribute__((packed, aligned(4))) U16x8 { uint16_t v[8]; };

__device__ __forceinline__ bf16x8 lds_frag_unaligned(const unsigned char *p) {
  const U16x8 raw = *reinterpret_cast<const U16x8 *>(p);
  return __builtin_bit_cast(bf16x8, raw);
}

// ------------------------------------------------------------------------------------------------ weight packing
// wf: generic packing [343][4][16] fp32 (BN scale folded, channel 3 = 0) -> [49][4][16][8] bf16 (kw 7 = 0)
__global__ __launch_bounds__(256) void stem_pack_bf16_kernel(const float *__restrict__ wf, bf16_t *__restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 49 * 4 * 16 * 8) return;
  const int kw = e & 7, co = (e >> 3) & 15, c = (e >> 7) & 3, kdkh = e >> 9;
  float v = 0.f;
  if (kw < 7 && c < 3) v = wf[((kdkh * 7 + kw) * 4 + c) * 16 + co];
  out[e] = f2bf(v);
}

// ------------------------------------------------------------------------------------------------ forward
// Persistent workgroups (one per CU, 8 waves) walk the 4x8x16 output tiles.  The packed weights (50 KB) are staged in LDS
// once; the input slab of the NEXT tile is fetched into registers (all loads in flight at once) while the current tile is
// multiplied, then converted to bf16 and written to LDS -- the global latency never sits between two MFMA phases.
struct FwdGeo {
  static constexpr int TD = 8, TH = 8, TW = 16, NT = 512;
  static constexpr int HD = TD + 6, HH = 2 * (TH - 1) + 7, HW = 40;          // 14 ring slots x 21 rows x 38 columns used (+1 read by the zero kw slot)
  static constexpr int PLANE = HD * HH * HW;                                  // bf16 elements per channel
  static constexpr int W_BYTES = 49 * 4 * 16 * 8 * 2;
  static constexpr int LDS_BYTES = W_BYTES + 3 * PLANE * 2 + 64;
  static constexpr int PAIRS = 20;                                           // 21 rows x 20 pairs = 420 of the 512 threads stage a plane
  static_assert(HH * PAIRS <= NT, "one (channel, depth) plane per staging step");
};

struct StemBfArgs {
  const float *x; const bf16_t *w16; const float *shift; bf16_t *y;
  int N, D, H, W, Ho, Wo, relu;
};


// Tile sequence of the persistent stem kernels: depth-fastest inside a (n, h, w) column; a workgroup walks a contiguous
// run, and the runs of one XCD's workgroups (blockIdx & 7 = XCD) are adjacent -- the input planes shared by depth
// neighbours (10 of 10 + 4 ...) then come from that XCD's L2 (see csrc/stem.hip).
struct BfTile { int n, td, th, tw; };
__device__ __forceinline__ BfTile bf_tile_of(int seq, int tiles_d, int tiles_h, int tiles_w) {
  BfTile t;
  t.td = seq % tiles_d; seq /= tiles_d;
  t.tw = seq % tiles_w; seq /= tiles_w;
  t.th = seq % tiles_h;
  t.n = seq / tiles_h;
  return t;
}
__device__ __forceinline__ void bf_tile_run(int bid, int nwg, int ntiles, int &begin, int &end) {
  if ((nwg & 7) == 0) {
    const int per_xcd = (ntiles + 7) >> 3, wgs = nwg >> 3;
    const int per_wg = (per_xcd + wgs - 1) / wgs;
    const int x0 = (bid & 7) * per_xcd;
    begin = x0 + (bid >> 3) * per_wg;
    end = min(min(begin + per_wg, x0 + per_xcd), ntiles);
  } else {
    const int per_wg = (ntiles + nwg - 1) / nwg;
    begin = bid * per_wg;
    end = min(begin + per_wg, ntiles);
  }
  if (begin > end) begin = end;
}

__global__ __launch_bounds__(512, 1) void stem_fwd_bf16_kernel(StemBfArgs p, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  using G = FwdGeo;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *wl = lds;
  unsigned char *slab_b = lds + G::W_BYTES;
  bf16_t *slab = reinterpret_cast<bf16_t *>(slab_b);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const size_t vox = (size_t)p.D * p.H * p.W;
  const bool even_w = (p.W & 1) == 0;

  for (int i = tid; i < G::W_BYTES / 16; i += G::NT)
    reinterpret_cast<uint4 *>(wl)[i] = reinterpret_cast<const uint4 *>(p.w16)[i];

  // Staging map: iteration u = one (channel, depth) plane of the slab (compile-time), thread = (row hh, column pair pr) of it:
  // the per-thread part of every address is ONE register, the plane part is wave-uniform.
  // A workgroup's tiles are depth-adjacent inside a (n, h, w) column and the slab's depth planes form a ring (input depth id lives
  // in slot (id + 14) % 14): a tile (8 output depths, 14 input planes) stages only its 8 new planes; the first tile of a column (or
  // of the workgroup's run) takes one more staging round in front for the other 6.
  constexpr int RCNT = G::TD, SCNT = G::HD - G::TD;                   // planes per channel of a regular / a start round: 8 / 6
  constexpr int NPL = 3 * RCNT;
  const int s_hh = tid / G::PAIRS, s_pr = tid - s_hh * G::PAIRS;
  const bool s_act = s_hh < G::HH;
  float2 st[NPL];
  // plane u of 24 in a round: ONE code path for both round kinds (see stem_wgrad_bf16_kernel)
  auto plane_of = [&](int u, bool start, int &c, int &j) {
    c = start ? u / SCNT : u / RCNT;
    j = start ? u % SCNT : u % RCNT;
    return start ? u < 3 * SCNT : true;
  };
  auto first_id = [&](int td_i, bool start) { return td_i * G::TD + (start ? -3 : 3); };
  auto fetch = [&](int tile, bool start) {
    const BfTile bt = bf_tile_of(tile, tiles_d, tiles_h, tiles_w);
    const int tw_i = bt.tw, th_i = bt.th, n = bt.n;
    const int id_first = first_id(bt.td, start);
    const int ih = 2 * th_i * G::TH - 3 + s_hh, iw = 2 * tw_i * G::TW - 3 + 2 * s_pr - 1;   // iw is even
    const bool row_ok = s_act && (unsigned)ih < (unsigned)p.H;
    const long long off = (long long)ih * p.W + iw;
    if (even_w) {
      // even W: iw is even, a pair is inside the row or outside it as a whole.  Buffer loads: a lane outside its row gets an offset
      // past the end of the image and the hardware returns zeros -- no bounds test around a load, all of them are in flight together
      // (a load inside a per-lane bounds test is a branch per load and the wave walks them one after the other)
      const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x + (size_t)n * 3 * vox), 0,
                                                                          (unsigned)(3 * vox * 4), 0x00020000);
      const unsigned voff = (row_ok && iw >= 0 && iw + 1 < p.W) ? (unsigned)off * 4u : 0x80000000u;
#pragma unroll
      for (int u = 0; u < NPL; ++u) {
        int c, j;                                                        // wave-uniform
        const bool used = plane_of(u, start, c, j);
        const int id = id_first + j;
        st[u] = make_float2(0.f, 0.f);
        if (used && (unsigned)id < (unsigned)p.D)
          st[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(
                                                 xr, (int)voff, (int)(((unsigned)c * (unsigned)vox + (unsigned)id * (unsigned)(p.H * p.W)) * 4u), 0));
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPL; ++u) {
        int c, j;
        const bool used = plane_of(u, start, c, j);
        const int id = id_first + j;
        st[u] = make_float2(0.f, 0.f);
        if (used && row_ok && (unsigned)id < (unsigned)p.D) {
          const float *src = p.x + ((size_t)n * 3 + c) * vox + (size_t)id * p.H * p.W + off;
          if ((unsigned)iw < (unsigned)p.W) st[u].x = src[0];
          if ((unsigned)(iw + 1) < (unsigned)p.W) st[u].y = src[1];
        }
      }
    }
  };
  auto commit = [&](int td_i, bool start) {
    if (!s_act) return;
    bf16_t *dst = slab + s_hh * G::HW;                                // slab columns 2*pr - 1, 2*pr
    const int s0 = (first_id(td_i, start) + G::HD) % G::HD;           // wave-uniform ring slot of the round's first plane
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      int c, j;
      if (!plane_of(u, start, c, j)) continue;
      int slot = s0 + j;
      slot -= slot >= G::HD ? G::HD : 0;
      bf16_t *d = dst + (c * G::HD + slot) * (G::HH * G::HW);
      if (s_pr >= 1) d[2 * s_pr - 1] = f2bf(st[u].x);
      d[2 * s_pr] = f2bf(st[u].y);
    }
  };

  // this wave's outputs: two depths (2 * (wave >> 1), + 1) x four consecutive rows ((wave & 1) * 4 ..) of 16 columns; lane (r, q):
  // output column r, channel slot q.  Slot 3 reads channel 0's columns: its weights are zero (stem_pack_bf16_kernel)
  const int w_td0 = (wave >> 1) * 2, w_th0 = (wave & 1) * 4;
  const int abase = (((q < 3 ? q : 0) * G::HD * G::HH + 2 * w_th0) * G::HW + 2 * r) * 2;
  const unsigned char *wlane = wl + (q * 16 + r) * 16;
  float sh4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) sh4[e] = p.shift ? p.shift[4 * q + e] : 0.f;

  // Multiply one tile.  The weights are the A operand: the accumulator holds 4 consecutive output channels of one position per
  // lane (8-byte stores, 512 contiguous bytes per store instruction).  A wave's eight accumulators are 2 output depths x 4 output
  // rows: the fragment of input plane pl and input row j feeds depth t with tap kd = pl - t and row i with tap kh = j - 2i -- ONE
  // fragment read per (plane, row) serves up to 8 MFMAs (104 reads for 392 MFMAs; a 4-deep tile with one depth per wave read 91
  // for 196 and its MFMA phase ran at the LDS array's pace), against the seven kh weight fragments of two kd taps in registers.
  f32x4_t acc[2][4];
  auto mma = [&](int tile) {
    const BfTile bt = bf_tile_of(tile, tiles_d, tiles_h, tiles_w);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int ring0 = (bt.td * G::TD - 3 + G::HD) % G::HD;            // slot of input depth 8 td - 3
    bf16x8 bw[2][7];                                                  // kh fragments of kd = pl (bw[pl & 1]) and kd = pl - 1
#pragma unroll
    for (int pl = 0; pl < 8; ++pl) {
      if (pl < 7) {
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) bw[pl & 1][kh] = *reinterpret_cast<const bf16x8 *>(wlane + (pl * 7 + kh) * 1024);
      }
      int slot = ring0 + w_td0 + pl;                                  // wave-uniform
      slot -= slot >= G::HD ? G::HD : 0;
      const unsigned char *arow = slab_b + abase + slot * (G::HH * G::HW * 2);
#pragma unroll
      for (int j = 0; j < 13; ++j) {
        const bf16x8 a = lds_frag_unaligned(arow + j * (G::HW * 2));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kh = j - 2 * i;
          if (kh >= 0 && kh < 7) {
            if (pl < 7) acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[pl & 1][kh], a, acc[0][i], 0, 0, 0);
            if (pl >= 1) acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[(pl - 1) & 1][kh], a, acc[1][i], 0, 0, 0);
          }
        }
      }
    }
  };
  // ---- epilogue: C/D layout of 16x16: column (output column) = lane & 15, row (cout) = 4 * (lane >> 4) + reg.  Called BEHIND the
  // commit of the next round, so that the stores drain under the next MFMA phase and not in front of a barrier.  (Streaming
  // stores -- cache policy slc -- make this launch 40 us shorter alone and the max-pool behind it as much longer: 4.471 / 4.476 ms.)
  auto epilogue = [&](int tile) {
    const BfTile bt = bf_tile_of(tile, tiles_d, tiles_h, tiles_w);
    const int d0 = bt.td * G::TD, h0 = bt.th * G::TH, w0 = bt.tw * G::TW;
    // stores through a buffer descriptor: the output row (n, od, oh) is a SCALAR byte offset, the lane's (column, channel quad) a
    // constant one (a column past the row's end gets an offset past the end of the buffer: the store is dropped) -- no 64-bit
    // index arithmetic per store
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0x7fffffff, 0x00020000);
    const int ow = w0 + r;
    const unsigned voff = ow < p.Wo ? (unsigned)(ow * 16 + 4 * q) * 2u : 0x80000000u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int od = d0 + w_td0 + t;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oh = h0 + w_th0 + i;
        if (od >= p.D || oh >= p.Ho) continue;                 // (wave-uniform)
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[t][i][e] + sh4[e];
          if (p.relu) v[e] = fmaxf(v[e], 0.f);
        }
        typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_st;
        const u32x2_st o = {(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
        __builtin_amdgcn_raw_buffer_store_b64(o, yr, (int)voff, (int)((((bt.n * p.D + od) * p.Ho + oh) * p.Wo) * 32), 0);
      }
    }
  };

  int t_begin, t_end;
  bf_tile_run((int)blockIdx.x, (int)gridDim.x, ntiles, t_begin, t_end);
  auto starts = [&](int tile) { return tile == t_begin || tile % tiles_d == 0; };      // depth is the fastest tile index
  // pipeline: [fetch round k + 1 into registers | multiply the tile round k completed] -> barrier -> [commit round k + 1] -> barrier
  bool have_cur = false, have_nxt = t_begin < t_end;
  int cur = t_begin, nxt = t_begin;
  bool nxt_start = true;                                              // the next round is the 6-plane round in front of a column's first tile
  while (true) {
    if (have_nxt) fetch(nxt, nxt_start);                              // in flight during the MFMA phase
    if (have_cur) mma(cur);
    if (!have_nxt) {
      if (have_cur) epilogue(cur);
      break;
    }
    __syncthreads();                                                  // the previous tile's fragments have been read
    commit(bf_tile_of(nxt, tiles_d, tiles_h, tiles_w).td, nxt_start);
    __syncthreads();
    if (have_cur) epilogue(cur);
    if (nxt_start) {
      have_cur = false;
      nxt_start = false;                                              // same tile: its 8 new planes
    } else {
      have_cur = true;
      cur = nxt;
      nxt = nxt + 1;
      have_nxt = nxt < t_end;
      nxt_start = have_nxt && starts(nxt);
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgGeo {
  static constexpr int TD = 4, TH = 4, TW = 32, NT = 512;
  static constexpr int HD = TD + 6, HH = 2 * (TH - 1) + 7;                    // 10 x 13 rows
  // The slab holds the even and the odd input columns as two separate images of [channel][depth slot][row][48] bf16 (36 used:
  // column 2*ow + kw of the slab = image kw & 1, index ow + (kw >> 1)).  A lane serves ONE (channel, kd, kh) group per row tile: it
  // reads the aligned window of 12 elements that holds the fragments of ALL the group's kw taps of one column parity -- 24 bytes
  // for 4 (even kw) or 3 (odd kw) fragments instead of 16 bytes each -- and shifts the odd starts out with v_perm / v_alignbit.
  // Strides in dwords, searched with the LDS's own lane groups (ds_read_b128: 4 groups of 16 lanes over 64 banks, MI355X guide):
  // rows of 24 put the 16-byte reads of a group's lanes r and r' on different banks (2 r + q (mod 16) chunks), the pads keep most
  // of that across the kh / kd / channel wraps and the depth ring: 5.0 LDS cycles per ds_read_b128 (4 is the floor; rows of 20
  // dwords: 8.0), 4.3 per tail ds_read_b64 (2-way by construction: 8 elements between k-groups).
  static constexpr int ROWD = 24, DSTR = HH * ROWD + 48, PLN = HD * DSTR + 8;
  static constexpr int IMG_BYTES = 3 * PLN * 4;
  static_assert(IMG_BYTES % 16 == 0 && (DSTR % 4) == 0 && (PLN % 4) == 0, "16-byte aligned windows");
  static constexpr int DYROW = 24;                                            // bf16 per dy row: 16 + 8 of padding (48 B)
  static constexpr int M = TD * TH * TW;
  static constexpr int SLAB_BYTES = 2 * IMG_BYTES, DY_BYTES = M * DYROW * 2;
  static constexpr int LDS_BYTES = SLAB_BYTES + DY_BYTES + 64;
  static constexpr int SCATTER_BYTES = M * 16 * 4;                           // POOLED: fp32 [position][16] the windows add into
  static constexpr int WINDOWS = 3 * 3 * 17;                                  // pooled windows that reach a 4 x 4 x 32 tile
  static constexpr int GROUPS = 3 * 49;                                       // 147 (channel, kd, kh) groups -> 10 row tiles x 7 kw
  static constexpr int PAIRS = 36;                                            // columns -1 .. 70; 13 rows x 36 pairs = 468 threads per plane
  static_assert(HH * PAIRS <= NT, "one (channel, depth) plane per staging step");
};

struct StemBfWgArgs {
  const float *x; const bf16_t *dy; float *dw; float *dbias;
  int N, D, H, W, Ho, Wo;
  // POOLED (mr3d_stem_conv_wgrad_bf16_pooled): dy is the gradient of the max-pooled map [N, Dp, Hp, Wp, 16]
  const uint32_t *pool_idx; const bf16_t *pooled;
  int Dp, Hp, Wp;
  int queue_slot;                      // g_stem_bf16_queues slot of this launch
};

// tile queues of the weight-gradient kernel, as in csrc/stem.hip: per slot 8 queue heads + 1 finished-workgroup count,
// all 0 between launches
__device__ unsigned g_stem_bf16_queues[64 * 16];

template <int V> struct IntC { static constexpr int value = V; };

// Work split of a k-step over the 8 waves: a unit = (row tile of 16 groups, column parity) = 4 (even kw) or 3 (odd kw) MFMAs on one
// window.  Waves 0-3: two even units (wave 0 also the tile of ones: sum_p dy); waves 4-5: three odd units; waves 6-7: one even +
// two odd -- 8..10 MFMAs per wave, 17-18 per SIMD (waves w and w + 4).
//
// POOLED: the max-pool backward (3x3x3 / stride 2 / pad 1) and the stem's ReLU mask are folded into the dy staging: the
// windows that reach the tile add relu'(pooled) * g at their arg-max into an fp32 tile in LDS, which is rounded to the bf16 dy
// rows once (the un-pooled bf16 gradient -- 168 MB on the 512 volume -- is neither written nor read).
template <bool POOLED>
__global__ __launch_bounds__(512, 1) void stem_wgrad_bf16_kernel(StemBfWgArgs p, int tiles_d, int tiles_h, int tiles_w, int run_len,
                                                                   int runs_per_col, int nunits) {
  using G = WgGeo;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *dyt = lds + G::SLAB_BYTES;
  float *scat = reinterpret_cast<float *>(lds + G::LDS_BYTES);       // POOLED only
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const size_t vox = (size_t)p.D * p.H * p.W;
  const bool even_w = (p.W & 1) == 0;

  // units of this wave (wave-uniform row tiles) and the lane's window offsets (bytes; k = 8q folded in, the depth slot is added
  // per k-step: the slab is a ring of 10 depth planes, see below)
  int egt[2] = {0, 0}, ogt[3] = {0, 0, 0};
  if (wave < 4) { egt[0] = 2 * wave; egt[1] = 2 * wave + 1; }
  else if (wave < 6) { ogt[0] = 3 * (wave - 4); ogt[1] = ogt[0] + 1; ogt[2] = ogt[0] + 2; }
  else { egt[0] = 8 + (wave - 6); ogt[0] = 6 + 2 * (wave - 6); ogt[1] = ogt[0] + 1; }
  auto window_off = [&](int gt, int &kd) {               // a lane past the last group reads the last group's window and drops the result
    const int g = min(16 * gt + r, G::GROUPS - 1);
    const int c = g / 49, kdkh = g - c * 49;
    kd = kdkh / 7;
    return (c * G::PLN + (kdkh - kd * 7) * G::ROWD + 4 * q) * 4;
  };
  int ekd[2], okd[3];
  const int eoff[2] = {window_off(egt[0], ekd[0]), window_off(egt[1], ekd[1])};
  const int ooff[3] = {window_off(ogt[0], okd[0]) + G::IMG_BYTES, window_off(ogt[1], okd[1]) + G::IMG_BYTES,
                       window_off(ogt[2], okd[2]) + G::IMG_BYTES};
  // transposing read of dy [position][16]: group q reads positions 8q..8q+7; inside the group lane 4a+b gives row a, columns 4b..4b+3
  const int i16 = lane & 15, ra = i16 >> 2, cb = i16 & 3;
  const int b_lane = (8 * q + ra) * (G::DYROW * 2) + cb * 8;

  f32x4_t acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // A workgroup walks RUNS of depth-adjacent tiles of one (n, h, w) column.  The slab's depth planes form a ring (input depth id
  // lives in slot (id + 10) % 10): the first tile of a run stages its 10 planes, every later tile only the 4 new ones -- 12 instead
  // of 30 (channel, depth) planes fetched, converted and written per tile.
  // The next tile's operands wait in registers: x as aligned float2 pairs, dy as two 16-byte halves of one position row.  Staging
  // map: iteration u = one (channel, depth) plane, thread = (row hh, column pair pr).
  struct Pos { int n, th, tw, t, t_end; };               // wave-uniform
  auto unit_pos = [&](int unit) {
    Pos o;
    int col = unit / runs_per_col;
    const int ri = unit - col * runs_per_col;
    o.tw = col % tiles_w; col /= tiles_w;
    o.th = col % tiles_h;
    o.n = col / tiles_h;
    o.t = ri * run_len;
    o.t_end = min(o.t + run_len, tiles_d);
    return o;
  };
  const int s_hh = tid / G::PAIRS, s_pr = tid - s_hh * G::PAIRS;
  const bool s_act = s_hh < G::HH;
  float2 st[3 * G::HD];
  uint4 sdy[2];
  // planes per channel: 10 at the start of a run (start), 4 inside it.  ONE code path for both (plane u of 30 is skipped by a
  // wave-uniform test inside a run): two instantiations met in register copies of the loaded values behind the fetch, and a copy
  // waits for the load -- the fetch then no longer ran beside the MFMA phase
  auto plane_of = [&](int u, bool start, int id_first, int &c, int &id) {
    c = start ? u / G::HD : u / G::TD;
    id = id_first + (start ? u % G::HD : u % G::TD);
    return start || u < 3 * G::TD;
  };
  auto fetch_x = [&](bool start, const Pos &o, int id_first) {
    const int h0 = o.th * G::TH, w0 = o.tw * G::TW;
    const int ih = 2 * h0 - 3 + s_hh, iw = 2 * w0 - 3 + 2 * s_pr - 1;
    const bool row_ok = s_act && (unsigned)ih < (unsigned)p.H;
    const long long off = (long long)ih * p.W + iw;
    if (even_w) {
      // even W: iw is even, a pair is inside the row or outside it as a whole.  Buffer loads: a lane outside its row gets an offset
      // past the end of the image and the hardware returns zeros -- no bounds test around a load, all of them are in flight together
      // (a load inside a per-lane bounds test is a branch per load and the wave walks them one after the other)
      const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x + (size_t)o.n * 3 * vox), 0,
                                                                          (unsigned)(3 * vox * 4), 0x00020000);
      const unsigned voff = (row_ok && iw >= 0 && iw + 1 < p.W) ? (unsigned)off * 4u : 0x80000000u;
#pragma unroll
      for (int u = 0; u < 3 * G::HD; ++u) {
        int c, id;                                       // wave-uniform
        const bool used = plane_of(u, start, id_first, c, id);
        st[u] = make_float2(0.f, 0.f);
        if (used && (unsigned)id < (unsigned)p.D)
          st[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(
                                                 xr, (int)voff, (int)(((unsigned)c * (unsigned)vox + (unsigned)id * (unsigned)(p.H * p.W)) * 4u), 0));
      }
    } else {
#pragma unroll
      for (int u = 0; u < 3 * G::HD; ++u) {
        int c, id;
        const bool used = plane_of(u, start, id_first, c, id);
        st[u] = make_float2(0.f, 0.f);
        if (used && row_ok && (unsigned)id < (unsigned)p.D) {
          const float *src = p.x + ((size_t)o.n * 3 + c) * vox + (size_t)id * p.H * p.W + off;
          if ((unsigned)iw < (unsigned)p.W) st[u].x = src[0];
          if ((unsigned)(iw + 1) < (unsigned)p.W) st[u].y = src[1];
        }
      }
    }
  };
  auto fetch = [&](const Pos &o, bool start) {
    const int d0 = o.t * G::TD, h0 = o.th * G::TH, w0 = o.tw * G::TW, n = o.n;
    if constexpr (!POOLED) {
      const int m = tid;                                 // 512 positions, one per thread
      const int tw = m & 31, th = (m >> 5) % G::TH, td = m / (32 * G::TH);
      const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
      sdy[0] = sdy[1] = make_uint4(0u, 0u, 0u, 0u);
      if (od < p.D && oh < p.Ho && ow < p.Wo) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.dy + ((((size_t)n * p.D + od) * p.Ho + oh) * p.Wo + ow) * 16);
        sdy[0] = src[0]; sdy[1] = src[1];
      }
    } else {
      // (window, channel quad) items: 3 x 3 x 17 windows x 4 = 612 for 512 threads; sdy[j] = {tap ids, g (2 dwords), ReLU bits}
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int it = tid + j * G::NT, q4 = it & 3, pp = it >> 2;
        const int pw = pp % 17, ph = (pp / 17) % 3, pd = pp / 51;
        const int od = (d0 >> 1) + pd, oh = (h0 >> 1) + ph, ow = (w0 >> 1) + pw;
        sdy[j] = make_uint4(0u, 0u, 0u, 0u);
        if (it < G::WINDOWS * 4 && od < p.Dp && oh < p.Hp && ow < p.Wp) {
          const size_t o4 = ((((size_t)n * p.Dp + od) * p.Hp + oh) * p.Wp + ow) * 4 + q4;
          const uint2 g = *reinterpret_cast<const uint2 *>(p.dy + o4 * 4);
          uint32_t pos = 0xfu;                           // bit r: pooled value of channel r is > 0
          if (p.pooled) {
            const uint2 a = *reinterpret_cast<const uint2 *>(p.pooled + o4 * 4);
            auto gt0 = [](uint32_t h) { return (h & 0x7fffu) != 0u && (h & 0x8000u) == 0u && (h & 0x7fffu) <= 0x7f80u; };
            pos = (gt0(a.x & 0xffffu) ? 1u : 0u) | (gt0(a.x >> 16) ? 2u : 0u) | (gt0(a.y & 0xffffu) ? 4u : 0u) | (gt0(a.y >> 16) ? 8u : 0u);
          }
          sdy[j] = make_uint4(p.pool_idx[o4], g.x, g.y, pos);
        }
      }
    }
    // the x loads come last: the registers of the dy operands are cleared while nothing is in flight (clearing a register that a
    // load of the previous round MAY still own costs an s_waitcnt vmcnt(0) -- behind the x loads it waited for all of them)
    fetch_x(start, o, start ? d0 - 3 : d0 + 3);
  };
  auto commit_x = [&](bool start, int id_first) {
    // slab column s = iw - iw0: the pair (2*pr - 1, 2*pr) lands at odd-image index pr - 1 and even-image index pr
    if (s_act) {
      bf16_t *dst = reinterpret_cast<bf16_t *>(lds) + s_hh * (2 * G::ROWD);
      const int s0 = (id_first + 10) % 10;               // wave-uniform ring slot of the first plane
#pragma unroll
      for (int u = 0; u < 3 * G::HD; ++u) {
        int c, id;
        if (!plane_of(u, start, 0, c, id)) continue;     // wave-uniform
        int slot = s0 + id;
        slot -= slot >= 10 ? 10 : 0;
        bf16_t *d = dst + (c * G::PLN + slot * G::DSTR) * 2;
        d[s_pr] = f2bf(st[u].y);
        if (s_pr >= 1) d[G::IMG_BYTES / 2 + s_pr - 1] = f2bf(st[u].x);
      }
    }
  };
  auto commit = [&](const Pos &o, bool start) {
    commit_x(start, start ? o.t * G::TD - 3 : o.t * G::TD + 3);
    if constexpr (!POOLED) {
      *reinterpret_cast<uint4 *>(dyt + tid * (G::DYROW * 2)) = sdy[0];
      *reinterpret_cast<uint4 *>(dyt + tid * (G::DYROW * 2) + 16) = sdy[1];
    }
  };
  // POOLED: every window adds its gradient at its arg-max position of the tile
  auto scatter = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int it = tid + j * G::NT, q4 = it & 3, pp = it >> 2;
      const int pw = pp % 17, ph = (pp / 17) % 3, pd = pp / 51;
      const uint32_t gq[2] = {sdy[j].y, sdy[j].z};
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const uint32_t gb = (rr & 1) ? (gq[rr >> 1] & 0xffff0000u) : (gq[rr >> 1] << 16);
        const int tap = (sdy[j].x >> (8 * rr)) & 255;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        // window (pd, ph, pw) of the tile's window block covers tile positions 2 p - 1 + k
        const int td = 2 * pd - 1 + kd, th = 2 * ph - 1 + kh, tw = 2 * pw - 1 + kw;
        if (((sdy[j].w >> rr) & 1u) && (gb & 0x7fffffffu) != 0u && (unsigned)td < (unsigned)G::TD && (unsigned)th < (unsigned)G::TH &&
            (unsigned)tw < (unsigned)G::TW)
          atomicAdd(scat + ((td * G::TH + th) * G::TW + tw) * 16 + q4 * 4 + rr, __uint_as_float(gb));
      }
    }
  };
  // after the barrier behind scatter(): a thread rounds its position's 16 sums into its bf16 dy row and clears them
  auto round_rows = [&]() {
    float4 *src = reinterpret_cast<float4 *>(scat + tid * 16);
    auto pk = [](const float4 &a, const float4 &b) {
      return make_uint4((uint32_t)f2bf(a.x) | ((uint32_t)f2bf(a.y) << 16), (uint32_t)f2bf(a.z) | ((uint32_t)f2bf(a.w) << 16),
                        (uint32_t)f2bf(b.x) | ((uint32_t)f2bf(b.y) << 16), (uint32_t)f2bf(b.z) | ((uint32_t)f2bf(b.w) << 16));
    };
    const float4 a = src[0], b = src[1], c = src[2], d = src[3];
    src[0] = src[1] = src[2] = src[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<uint4 *>(dyt + tid * (G::DYROW * 2)) = pk(a, b);
    *reinterpret_cast<uint4 *>(dyt + tid * (G::DYROW * 2) + 16) = pk(c, d);
  };

  // ---- the 16 k-steps of a tile for a wave with NE even and NO odd units (+ the tile of ones); ring0 = slot of the tile's first
  // input depth plane
  auto mma_phase = [&](auto ne_, auto no_, auto ones_, int ring0) {
    constexpr int NE = decltype(ne_)::value, NO = decltype(no_)::value;
    constexpr bool ONES = decltype(ones_)::value != 0;
    struct Ops { u32x4 ea[NE ? NE : 1]; u32x2 eb[NE ? NE : 1]; u32x4 oa[NO ? NO : 1]; u32x2 ob[NO ? NO : 1]; bf16x8 b; };
    // window addresses of the units for one td (4 k-steps: th = 0..3 is a constant offset of the LDS read)
    struct Base { uint32_t e[NE ? NE : 1], o[NO ? NO : 1], y; };
    auto bases = [&](int td, Base &b) {
      auto slot_off = [&](int kd) {                      // bytes of ring slot (ring0 + td + kd) % 10
        int sl = ring0 + td + kd;
        sl -= sl >= 10 ? 10 : 0;
        return sl * (G::DSTR * 4);
      };
#pragma unroll
      for (int u = 0; u < NE; ++u) b.e[u] = (uint32_t)(eoff[u] + slot_off(ekd[u]));
#pragma unroll
      for (int u = 0; u < NO; ++u) b.o[u] = (uint32_t)(ooff[u] + slot_off(okd[u]));
      b.y = (uint32_t)(G::SLAB_BYTES + td * G::TH * 32 * (G::DYROW * 2) + b_lane);
    };
    auto load_step = [&](auto th_, const Base &b, Ops &o) {
      constexpr int TH = decltype(th_)::value;
      const unsigned char *yb = lds + b.y + TH * 32 * (G::DYROW * 2);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(yb));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(yb + 4 * (G::DYROW * 2)));
      o.b = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      constexpr int rowoff = 2 * TH * G::ROWD * 4;
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const unsigned char *w = lds + b.e[u] + rowoff;
        o.ea[u] = *reinterpret_cast<const u32x4 *>(w);
        o.eb[u] = *reinterpret_cast<const u32x2 *>(w + 16);
      }
#pragma unroll
      for (int u = 0; u < NO; ++u) {
        const unsigned char *w = lds + b.o[u] + rowoff;
        o.oa[u] = *reinterpret_cast<const u32x4 *>(w);
        o.ob[u] = *reinterpret_cast<const u32x2 *>(w + 16);          // (the second dword is not used: an aligned register pair for v_pk_mov)
      }
    };
    // Operand assembly.  An MFMA holds the SIMD's vector issue for 8 of its 16 cycles: two plain vector instructions per MFMA are
    // free, every further one costs its 4 cycles.  The tuple of a fragment that starts one dword into the window is two v_pk_mov_b32
    // of aligned register pairs ((a.hi, b.lo)), not four v_mov; the element-shifted tuples are v_perm (alignbit) of neighbouring dwords.
    auto mfma = [&](f32x4_t &c, const u32x4 &a, const bf16x8 &b) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), b, c, 0, 0, 0);
    };
    auto hi_lo = [](const u32x2 &a, const u32x2 &b) {      // (a[1], b[0])
      return (u32x2)__builtin_shufflevector(a, b, 1, 2);   // v_pk_mov_b32 op_sel:[1,0]
    };
    auto lo2 = [](const u32x4 &v) { return (u32x2)__builtin_shufflevector(v, v, 0, 1); };
    auto hi2 = [](const u32x4 &v) { return (u32x2)__builtin_shufflevector(v, v, 2, 3); };
    auto cat = [](const u32x2 &a, const u32x2 &b) { return (u32x4)__builtin_shufflevector(a, b, 0, 1, 2, 3); };
    auto shifted = [&](const u32x4 &w, uint32_t d4) {      // elements 1..8 of the window whose dwords are w[0..3], d4
      return u32x4{__builtin_amdgcn_alignbit(w[1], w[0], 16), __builtin_amdgcn_alignbit(w[2], w[1], 16),
                   __builtin_amdgcn_alignbit(w[3], w[2], 16), __builtin_amdgcn_alignbit(d4, w[3], 16)};
    };
    auto mma_step = [&](const Ops &o) {
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const u32x4 &w = o.ea[u];                           // dwords 0..3; o.eb[u] = dwords 4, 5
        const u32x4 t1 = shifted(w, o.eb[u][0]);            // kw 2: elements 1..8
        const u32x4 w1 = cat(hi_lo(lo2(w), hi2(w)), hi_lo(hi2(w), o.eb[u]));      // kw 4: elements 2..9 = dwords 1..4
        u32x4 t2 = cat(hi_lo(lo2(t1), hi2(t1)), hi2(t1));   // kw 6: elements 3..10
        t2[2] = t1[3];
        t2[3] = __builtin_amdgcn_alignbit(o.eb[u][1], o.eb[u][0], 16);
        mfma(acc[4 * u + 0], w, o.b);
        mfma(acc[4 * u + 1], t1, o.b);
        mfma(acc[4 * u + 2], w1, o.b);
        mfma(acc[4 * u + 3], t2, o.b);
      }
#pragma unroll
      for (int u = 0; u < NO; ++u) {
        const u32x4 &w = o.oa[u];
        mfma(acc[4 * NE + 3 * u + 0], w, o.b);                                     // kw 1
        mfma(acc[4 * NE + 3 * u + 1], shifted(w, o.ob[u][0]), o.b);               // kw 3
        mfma(acc[4 * NE + 3 * u + 2], cat(hi_lo(lo2(w), hi2(w)), hi_lo(hi2(w), o.ob[u])), o.b);   // kw 5
      }
      if constexpr (ONES) mfma(acc[4 * NE + 3 * NO], u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, o.b);
    };
    // the windows of k-step ks + 1 are requested before the MFMAs of k-step ks
    Ops o0, o1;
    Base cur, nxt;
    bases(0, cur);
    load_step(IntC<0>{}, cur, o0);
#pragma unroll 1
    for (int td = 0; td < G::TD; ++td) {
      bases(td + 1, nxt);
      load_step(IntC<1>{}, cur, o1);
      mma_step(o0);
      load_step(IntC<2>{}, cur, o0);
      mma_step(o1);
      load_step(IntC<3>{}, cur, o1);
      mma_step(o0);
      if (td + 1 < G::TD) load_step(IntC<0>{}, nxt, o0);
      mma_step(o1);
      cur = nxt;
    }
  };
  // ---- write-out of one accumulator: column (cout) = lane & 15, group = 16*gt + 4*(lane>>4) + reg
  auto write_acc = [&](const f32x4_t &a, int gt, int kw) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = 16 * gt + 4 * q + e;
      const int c = g / 49, kdkh = g - c * 49;
      if (g < G::GROUPS) atomicAdd(p.dw + ((size_t)(kdkh * 7 + kw) * 4 + c) * 16 + r, a[e]);
    }
  };

  // Runs are pulled from eight queues (one contiguous range of the run order per XCD, the workgroup's own XCD first): with a fixed
  // share per workgroup the launch ended with its latest workgroup, and in the training step the other layers' last weight
  // gradients and their finishing run beside this kernel (see stem_wgrad_kernel in csrc/stem.hip).
  __shared__ int s_unit;
  unsigned *queue = g_stem_bf16_queues + 16 * p.queue_slot;
  const int per_xcd = (nunits + 7) >> 3, home = blockIdx.x & 7;
  auto range = [&](int x, int &lo) { lo = x * per_xcd; return max(min(lo + per_xcd, nunits) - lo, 0); };
  auto take = [&](unsigned ticket) {                     // thread 0
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
  if constexpr (POOLED) {
    for (int e = tid; e < G::M * 4; e += G::NT) reinterpret_cast<float4 *>(scat)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  unsigned ticket = 0;
  if (tid == 0) s_unit = take(atomicAdd(queue + home, 1u));
  __syncthreads();
  const int first = __builtin_amdgcn_readfirstlane(s_unit);   // wave-uniform: the tile arithmetic below stays on the scalar unit
  if (tid == 0 && first >= 0) ticket = atomicAdd(queue + home, 1u);
  // Pipeline: [fetch tile k + 1 into registers | multiply tile k] -> barrier -> [commit tile k + 1 to LDS] -> barrier
  bool have_cur = false, have_nxt = first >= 0, nxt_start = true;
  Pos cur = unit_pos(0), nxt = unit_pos(first >= 0 ? first : 0);
  while (true) {
    if (have_nxt) fetch(nxt, nxt_start);                 // in flight during the MFMA phase
    if (have_cur) {
      const int ring0 = (cur.t * G::TD + 7) % 10;        // slot of input depth 4t - 3
      if (wave == 0) mma_phase(IntC<2>{}, IntC<0>{}, IntC<1>{}, ring0);
      else if (wave < 4) mma_phase(IntC<2>{}, IntC<0>{}, IntC<0>{}, ring0);
      else if (wave < 6) mma_phase(IntC<0>{}, IntC<3>{}, IntC<0>{}, ring0);
      else mma_phase(IntC<1>{}, IntC<2>{}, IntC<0>{}, ring0);
    }
    if (!have_nxt) break;
    __syncthreads();
    commit(nxt, nxt_start);
    if constexpr (POOLED) scatter();
    const bool last = nxt.t + 1 >= nxt.t_end;            // the tile after `nxt` starts a new run
    if (tid == 0 && last) s_unit = take(ticket);         // issued a run ago
    __syncthreads();
    if constexpr (POOLED) {
      round_rows();
      __syncthreads();
    }
    cur = nxt; have_cur = true;
    if (last) {
      const int nu = __builtin_amdgcn_readfirstlane(s_unit);
      have_nxt = nu >= 0;
      nxt_start = true;
      if (have_nxt) {
        nxt = unit_pos(nu);
        if (tid == 0) ticket = atomicAdd(queue + home, 1u);
      }
    } else {
      nxt.t += 1;
      nxt_start = false;
    }
  }
  if (tid == 0) {                                        // the last workgroup to run dry re-arms the queues
    const unsigned done = atomicAdd(queue + 8, 1u);
    if (done == gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 9; ++x) atomicExch(queue + x, 0u);
    }
  }
  if (wave < 4) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) write_acc(acc[4 * u + k], egt[u], 2 * k);
    if (wave == 0 && p.dbias && q == 0) atomicAdd(p.dbias + r, acc[8][0]);
  } else if (wave < 6) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) write_acc(acc[3 * u + k], ogt[u], 2 * k + 1);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) write_acc(acc[k], egt[0], 2 * k);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) write_acc(acc[4 + 3 * u + k], ogt[u], 2 * k + 1);
  }
}

}  // namespace

extern "C" int mr3d_stem_pack_weights_bf16(const float *w_packed, void *w16, mr3d_stream_t stream_) {
  if (!w_packed || !w16) return MR3D_ERR_INVALID_ARG;
  stem_pack_bf16_kernel<<<(49 * 4 * 16 * 8 + 255) / 256, 256, 0, mr3d_stream(stream_)>>>(w_packed, (bf16_t *)w16);
  return mr3d_launch_status();
}

extern "C" int mr3d_stem_conv_forward_bf16(const float *x_ncdhw, const void *w16, const float *shift, void *y_bf16, int n, int d, int h,
                                           int w, int relu, mr3d_stream_t stream_) {
  if (!x_ncdhw || !w16 || !y_bf16 || n <= 0 || d <= 0 || h <= 0 || w <= 0) return MR3D_ERR_INVALID_ARG;
  StemBfArgs a;
  a.x = x_ncdhw; a.w16 = (const bf16_t *)w16; a.shift = shift; a.y = (bf16_t *)y_bf16;
  a.N = n; a.D = d; a.H = h; a.W = w; a.Ho = (h - 1) / 2 + 1; a.Wo = (w - 1) / 2 + 1; a.relu = relu;
  using G = FwdGeo;
  // buffer-load offsets of the slab fetch: one image (3 channels, fp32) below 2^31 bytes; buffer-store offsets: the output too
  if ((unsigned long long)d * h * w * 12ull > 0x7fffffffull) return MR3D_ERR_UNSUPPORTED;
  if ((unsigned long long)n * d * a.Ho * a.Wo * 32ull > 0x7fffffffull) return MR3D_ERR_UNSUPPORTED;
  static bool attr = false;
  static int cus = 0;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(stem_fwd_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    attr = true;
  }
  const int td = (d + G::TD - 1) / G::TD, th = (a.Ho + G::TH - 1) / G::TH, tw = (a.Wo + G::TW - 1) / G::TW;
  const long long ntiles = (long long)n * td * th * tw;
  if (ntiles > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  int grid = ntiles < cus ? (int)ntiles : cus;               // one persistent 8-wave workgroup per CU (100 KB of LDS)
  if (grid >= 8) grid &= ~7;                                 // XCD-contiguous tile runs (bf_tile_run)
  stem_fwd_bf16_kernel<<<(unsigned)grid, G::NT, G::LDS_BYTES, mr3d_stream(stream_)>>>(a, td, th, tw, (int)ntiles);
  return mr3d_launch_status();
}

static int stem_wgrad_bf16_impl(const float *x_ncdhw, const void *dy_bf16, const uint32_t *pool_idx, const void *pooled, float *dw_packed,
                                float *dbias, int n, int d, int h, int w, mr3d_stream_t stream_) {
  if (!x_ncdhw || !dy_bf16 || !dw_packed || n <= 0 || d <= 0 || h <= 0 || w <= 0) return MR3D_ERR_INVALID_ARG;
  StemBfWgArgs a;
  a.x = x_ncdhw; a.dy = (const bf16_t *)dy_bf16; a.dw = dw_packed; a.dbias = dbias;
  a.N = n; a.D = d; a.H = h; a.W = w; a.Ho = (h - 1) / 2 + 1; a.Wo = (w - 1) / 2 + 1;
  a.pool_idx = pool_idx; a.pooled = (const bf16_t *)pooled;
  a.Dp = (d - 1) / 2 + 1; a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;
  using G = WgGeo;
  // buffer-load offsets of the slab fetch: one image (3 channels, fp32) below 2^31 bytes
  if ((unsigned long long)d * h * w * 12ull > 0x7fffffffull) return MR3D_ERR_UNSUPPORTED;
  static bool attr = false;
  static int cus = 0;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(stem_wgrad_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(stem_wgrad_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS_BYTES + G::SCATTER_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    attr = true;
  }
  const int td = (d + G::TD - 1) / G::TD, th = (a.Ho + G::TH - 1) / G::TH, tw = (a.Wo + G::TW - 1) / G::TW;
  const long long ncols = (long long)n * th * tw;
  if (ncols * td > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
  // runs of depth-adjacent tiles: whole columns when there are at least 4 per workgroup, else the columns are cut so that there
  // are (the first tile of a run stages 10 depth planes, the others 4)
  int rpc = (int)((4ll * cus + ncols - 1) / ncols);
  if (rpc < 1) rpc = 1;
  if (rpc > (td + 1) / 2) rpc = (td + 1) / 2;
  const int run_len = (td + rpc - 1) / rpc;
  rpc = (td + run_len - 1) / run_len;
  const long long nunits = ncols * rpc;
  int grid = nunits < cus ? (int)nunits : cus;               // one persistent workgroup per CU (88 / 120 KB of LDS)
  if (grid >= 8) grid &= ~7;
  static std::atomic<unsigned> next_slot{0};                 // ring of queue slots: see stem_wgrad_impl in csrc/stem.hip
  a.queue_slot = (int)(next_slot.fetch_add(1) & 63u);
  if (pool_idx)
    stem_wgrad_bf16_kernel<true><<<(unsigned)grid, 512, G::LDS_BYTES + G::SCATTER_BYTES, mr3d_stream(stream_)>>>(a, td, th, tw, run_len, rpc, (int)nunits);
  else
    stem_wgrad_bf16_kernel<false><<<(unsigned)grid, 512, G::LDS_BYTES, mr3d_stream(stream_)>>>(a, td, th, tw, run_len, rpc, (int)nunits);
  return mr3d_launch_status();
}

extern "C" int mr3d_stem_conv_wgrad_bf16(const float *x_ncdhw, const void *dy_bf16, float *dw_packed, float *dbias, int n, int d, int h,
                                         int w, mr3d_stream_t stream_) {
  return stem_wgrad_bf16_impl(x_ncdhw, dy_bf16, nullptr, nullptr, dw_packed, dbias, n, d, h, w, stream_);
}

extern "C" int mr3d_stem_conv_wgrad_bf16_pooled(const float *x_ncdhw, const void *g_pooled_bf16, const uint32_t *argmax,
                                                const void *pooled_bf16, float *dw_packed, float *dbias, int n, int d, int h, int w,
                                                mr3d_stream_t stream_) {
  if (!argmax) return MR3D_ERR_INVALID_ARG;
  return stem_wgrad_bf16_impl(x_ncdhw, g_pooled_bf16, argmax, pooled_bf16, dw_packed, dbias, n, d, h, w, stream_);
}

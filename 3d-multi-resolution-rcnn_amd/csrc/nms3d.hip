// NMS3D for gfx950: score sort, 64x64 IoU bit-matrix, greedy sweep and index
// compaction, all on the device (no D2H mask copy, no host loop).
//
// Semantics follow the reference's nms_cuda_3d (mmdet/ops/nms/src/nms_kernel.cu:196-257):
//   sort by score descending (:199-201) -> bit (i,j) = IoU3D(i,j) > thr, strict (:122)
//   -> greedy sweep in sorted order (:237-249) -> original indices, ascending (:253-256).
// The IoU arithmetic is devIoU3d's fp32 op order (:23-33) and must stay bit-exact:
// this file is compiled with -ffp-contract=off and IEEE fp32 division.
//
// A CDNA wavefront is 64 lanes, so one wave owns one 64-bit row word of the bit-matrix
// and lane == bit position; the sweep resolves one 64-box block at a time with scalar
// (SGPR) bit tricks and v_readlane instead of a 64k-iteration host loop.
#include "mr3d_common.h"
#include "sort_regs.h"

namespace {

constexpr int kSortLdsMax = 4096;   // boxes sorted inside one workgroup's LDS (32 KiB of keys)
constexpr int kSortThreads = 1024;
constexpr int kMaxSeg = 16;         // independent box sets per batched call (pyramid levels x pathways)

// One launch serves several independent NMS problems ("segments": the 5 pyramid levels of rpn_head_3d.py:81-135 are
// 5 separate nms() calls in the reference).  Segment s owns rows [row0[s], row0[s] + n[s]) of dets / order / keep and
// its own slices of the sorted-box and bit-matrix workspaces.
struct SegTable {
  int num;
  int row0[kMaxSeg], n[kMaxSeg];
  long long box_off[kMaxSeg];       // in float4 units
  long long mask_off[kMaxSeg];      // in u64 units
  const int32_t *n_dev;             // optional device array: only the first n_dev[s] (<= n[s]) rows of set s are real
};

__device__ __forceinline__ int seg_rows(const SegTable &seg, int sg) {
  int n = seg.n[sg];
  if (seg.n_dev) {
    const int m = seg.n_dev[sg];
    n = m < 0 ? 0 : (m < n ? m : n);
  }
  return n;
}

// (score desc, index asc) as one ascending u64 key
__device__ __forceinline__ uint64_t make_key(float s, int idx) {
  uint32_t u = __float_as_uint(s);
  if (u == 0x80000000u) u = 0u;                       // -0.0 compares equal to +0.0
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // order-preserving float -> uint
  u = ~u;                                             // descending score
  return ((uint64_t)u << 32) | (uint32_t)idx;
}

// sorted box record: {x1,y1,x2,y2} {z1,z2,volume,0}.  volume = (x2-x1+1)*(y2-y1+1)*(z2-z1+1)
// evaluated exactly like Sa/Sb in devIoU3d, so hoisting it out of the pair loop is bit-neutral.
__device__ __forceinline__ void store_box(const float *dets, int stride, int src, float4 *boxes, int dst) {
  const float *p = dets + (size_t)src * stride;
  float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3], z1 = p[4], z2 = p[5];
  float vol = (x2 - x1 + 1) * (y2 - y1 + 1) * (z2 - z1 + 1);
  boxes[2 * dst] = make_float4(x1, y1, x2, y2);
  boxes[2 * dst + 1] = make_float4(z1, z2, vol, 0.f);
}

// ---- sort, n <= kSortLdsMax: one workgroup, bitonic network in registers (sort_regs.h) ---------------------
__global__ __launch_bounds__(kSortThreads) void nms3d_sort_lds_kernel(const float *__restrict__ dets_all, SegTable seg,
                                                                     int stride, int *__restrict__ order_all,
                                                                     float4 *__restrict__ boxes_all) {
  __shared__ uint64_t keys[kSortLdsMax];
  const int tid = threadIdx.x;
  const int sg = blockIdx.x;
  const int n = seg_rows(seg, sg);
  if (n == 0) return;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  const float *dets = dets_all + (size_t)seg.row0[sg] * stride;
  int *order = order_all + seg.row0[sg];
  float4 *boxes = boxes_all + seg.box_off[sg];
  auto key_of = [&](int i) { return (i < n) ? make_key(dets[(size_t)i * stride + 6], i) : ~0ull; };
  if (np2 <= 2 * kSortThreads) {
    uint64_t v[2] = {key_of(tid * 2), key_of(tid * 2 + 1)};
    bitonic_sort_regs<2>(v, tid, np2 < 2 ? 2 : np2, keys);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = tid * 2 + e;
      if (i < n) {
        const int src = (int)(uint32_t)v[e];
        order[i] = src;
        store_box(dets, stride, src, boxes, i);
      }
    }
  } else {
    uint64_t v[4] = {key_of(tid * 4), key_of(tid * 4 + 1), key_of(tid * 4 + 2), key_of(tid * 4 + 3)};
    bitonic_sort_regs<4>(v, tid, np2, keys);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = tid * 4 + e;
      if (i < n) {
        const int src = (int)(uint32_t)v[e];
        order[i] = src;
        store_box(dets, stride, src, boxes, i);
      }
    }
  }
}

// ---- sort, n > kSortLdsMax: bitonic network over global memory (rare path) -----------
__global__ void nms3d_keys_init_kernel(const float *__restrict__ dets, int n, int stride, int np2,
                                       uint64_t *__restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np2) keys[i] = (i < n) ? make_key(dets[(size_t)i * stride + 6], i) : ~0ull;
}

__global__ void nms3d_bitonic_step_kernel(uint64_t *__restrict__ keys, int np2, int j, int k) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np2) return;
  int p = i ^ j;
  if (p > i) {
    uint64_t a = keys[i], b = keys[p];
    bool up = (i & k) == 0;
    if ((a > b) == up) { keys[i] = b; keys[p] = a; }
  }
}

__global__ void nms3d_gather_kernel(const float *__restrict__ dets, int n, int stride,
                                    const uint64_t *__restrict__ keys, int *__restrict__ order,
                                    float4 *__restrict__ boxes) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int src = (int)(uint32_t)keys[i];
  order[i] = src;
  store_box(dets, stride, src, boxes, i);
}

// ---- 64x64 suppression words; one wave per (row block, col block), upper triangle only ---
// (the reference also fills the blocks below the diagonal, nms_kernel.cu:86, but its sweep
//  never reads them: `for (j = nblock; ...)`, :244)
__global__ __launch_bounds__(MR3D_WAVE) void nms3d_mask_kernel(const float4 *__restrict__ boxes_all, SegTable seg,
                                                              float thr, uint64_t *__restrict__ mask_all) {
  const int sg = blockIdx.z;
  const int n = seg_rows(seg, sg), cb = (n + 63) / 64;
  const int c = blockIdx.x, r = blockIdx.y;
  if (c < r || c >= cb) return;
  const float4 *boxes = boxes_all + seg.box_off[sg];
  uint64_t *mask = mask_all + seg.mask_off[sg];
  __shared__ float4 cbx[2 * MR3D_WAVE];
  const int lane = threadIdx.x;
  const int cidx = c * 64 + lane;
  if (cidx < n) {
    cbx[2 * lane] = boxes[2 * cidx];
    cbx[2 * lane + 1] = boxes[2 * cidx + 1];
  }
  __syncthreads();
  const int ridx = r * 64 + lane;
  if (ridx >= n) return;
  const float4 a0 = boxes[2 * ridx], a1 = boxes[2 * ridx + 1];
  const int csz = min(n - c * 64, 64);
  const int start = (r == c) ? lane + 1 : 0;
  uint64_t t = 0;
  for (int i = 0; i < csz; ++i) {
    const float4 b0 = cbx[2 * i], b1 = cbx[2 * i + 1];
    float left = fmaxf(a0.x, b0.x), right = fminf(a0.z, b0.z);
    float top = fmaxf(a0.y, b0.y), bottom = fminf(a0.w, b0.w);
    float front = fmaxf(a1.x, b1.x), back = fminf(a1.y, b1.y);
    float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
    float depth = fmaxf(back - front + 1, 0.f);
    float interS = width * height * depth;
    float iou = interS / (a1.z + b1.z - interS);
    if (i >= start && iou > thr) t |= 1ull << i;
  }
  mask[(size_t)ridx * cb + c] = t;
}

// ---- greedy sweep + compaction: one workgroup ---------------------------------------------
// LDS: remv[cb] (suppressed bits per sorted block), flags[cb] (kept bits in ORIGINAL index space).
constexpr int kSweepThreads = 256;
constexpr int kRowsPerItem = 8;     // mask rows OR-ed by one thread per work item

__global__ __launch_bounds__(kSweepThreads) void nms3d_sweep_kernel(const uint64_t *__restrict__ mask_all,
                                                                   const int *__restrict__ order_all, SegTable seg,
                                                                   int64_t *__restrict__ keep_all,
                                                                   int32_t *__restrict__ num_keep_all) {
  extern __shared__ __attribute__((aligned(16))) uint64_t sm[];
  const int sg = blockIdx.x;
  const int n = seg_rows(seg, sg), cb = (n + 63) / 64;
  int32_t *num_keep = num_keep_all + sg;
  if (n == 0) {
    if (threadIdx.x == 0) *num_keep = 0;
    return;
  }
  const uint64_t *mask = mask_all + seg.mask_off[sg];
  const int *order = order_all + seg.row0[sg];
  int64_t *keep_out = keep_all + seg.row0[sg];
  uint64_t *remv = sm;
  uint64_t *flags = sm + cb;
  uint64_t *bcast = sm + 2 * cb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * cb + 1; i += kSweepThreads) sm[i] = 0;
  __syncthreads();

  for (int b = 0; b < cb; ++b) {
    if (wave == 0) {
      const int row = b * 64 + lane;
      const uint64_t diag = (row < n) ? mask[(size_t)row * cb + b] : 0ull;
      const int rows = min(64, n - b * 64);
      const uint64_t valid = (rows == 64) ? ~0ull : ((1ull << rows) - 1ull);
      // greedy resolution inside the block.  Only rows whose diagonal word is non-zero can suppress anything, so the
      // serial chain walks those rows only (a row with an empty word is kept iff nobody before it suppressed it):
      //   cur = bits suppressed so far;  for t in (rows that may suppress, ascending, still alive): cur |= diag[t]
      uint64_t cur = mr3d_uniform64(remv[b]);              // wave-uniform: lives in SGPRs
      uint64_t cand = (uint64_t)__ballot(diag != 0ull) & ~cur & valid;
      while (cand) {
        const int t = __builtin_ctzll(cand);
        const uint64_t d = mr3d_readlane64(diag, t);
        cur |= d;
        cand &= ~d;
        cand &= ~(1ull << t);
      }
      const uint64_t kept = ~cur & valid;
      if (lane == 0) *bcast = kept;
      if ((kept >> lane) & 1ull) {
        const int oi = order[row];
        atomicOr((unsigned int *)flags + (oi >> 5), 1u << (oi & 31));
      }
    }
    __syncthreads();
    const uint64_t kept = *bcast;
    // remv[w] |= OR_{t in kept} mask[b*64+t][w] for every later block w.
    // work item = (word w, group of kRowsPerItem rows); loads are unconditional (a dropped row
    // re-reads row 0 of the block, an L1 hit) so the compiler can keep all of them in flight.
    const int later = cb - (b + 1);
    const int items = later * (64 / kRowsPerItem);
    for (int it = tid; it < items; it += kSweepThreads) {
      const int w = b + 1 + it % later;
      const int g = it / later;
      uint64_t acc = 0;
#pragma unroll
      for (int q = 0; q < kRowsPerItem; ++q) {
        const int t = g * kRowsPerItem + q;
        const bool on = (kept >> t) & 1ull;           // kept bits only exist for rows < n
        const uint64_t v = mask[(size_t)(b * 64 + (on ? t : 0)) * cb + w];
        acc |= on ? v : 0ull;
      }
      if (acc) atomicOr((unsigned long long *)&remv[w], (unsigned long long)acc);
    }
    __syncthreads();
  }

  if (wave == 0) {
    int base = 0;
    for (int w = 0; w < cb; ++w) {
      const uint64_t f = flags[w];
      if ((f >> lane) & 1ull) keep_out[base + __popcll(f & ((1ull << lane) - 1ull))] = (int64_t)w * 64 + lane;
      base += __popcll(f);
    }
    if (lane == 0) *num_keep = base;
  }
}

// ---- greedy sweep, cb <= 64 blocks (n <= 4096: every batched call): the serial chain without global latency ---------
// The generic sweep above pays two dependent global round trips per 64-box block (the diagonal words, then the mask rows
// of the kept boxes): 32 blocks x ~1.4 us.  Here (a) all diagonal words sit in LDS before the chain starts, (b) the mask
// rows a block MAY need are fetched one block ahead, unconditionally (which rows are kept is only known later; the words
// of dropped rows are masked out after the fact), so the chain itself is LDS + scalar work between two barriers per block,
// (c) the order[] gather for the output flags happens once, after the chain.
constexpr int kFastMaxCb = 64;
__global__ __launch_bounds__(kSweepThreads) void nms3d_sweep_fast_kernel(const uint64_t *__restrict__ mask_all,
                                                                        const int *__restrict__ order_all, SegTable seg,
                                                                        int64_t *__restrict__ keep_all,
                                                                        int32_t *__restrict__ num_keep_all) {
  extern __shared__ __attribute__((aligned(16))) uint64_t sm[];
  const int sg = blockIdx.x;
  const int n = seg_rows(seg, sg), cb = (n + 63) / 64;
  int32_t *num_keep = num_keep_all + sg;
  if (n == 0) {
    if (threadIdx.x == 0) *num_keep = 0;
    return;
  }
  const uint64_t *mask = mask_all + seg.mask_off[sg];
  const int *order = order_all + seg.row0[sg];
  int64_t *keep_out = keep_all + seg.row0[sg];
  uint64_t *remv = sm;                     // [cb] suppressed bits per sorted block
  uint64_t *flags = sm + cb;               // [cb] kept bits in ORIGINAL index space
  uint64_t *keptb = sm + 2 * cb;           // [cb] kept bits per sorted block
  uint64_t *diag = sm + 3 * cb;            // [cb * 64] diagonal word of every row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * cb; i += kSweepThreads) sm[i] = 0;
  for (int r = tid; r < cb * 64; r += kSweepThreads) diag[r] = r < n ? mask[(size_t)r * cb + (r >> 6)] : 0ull;

  // work items of block b: (word w in (b, cb), group g of kRowsPerItem rows); this thread's items: tid, tid + 256
  constexpr int G = 64 / kRowsPerItem;
  uint64_t pre[2][kRowsPerItem];
  auto fetch = [&](int b) {
    const int later = cb - (b + 1);
    const int items = later * G;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int it = tid + s_ * kSweepThreads;
      const bool on = it < items;
      const int w = on ? b + 1 + it % later : 0, g = on ? it / later : 0;
#pragma unroll
      for (int q = 0; q < kRowsPerItem; ++q) {
        const int row = b * 64 + g * kRowsPerItem + q;
        pre[s_][q] = (on && row < n) ? mask[(size_t)row * cb + w] : 0ull;
      }
    }
  };
  fetch(0);
  __syncthreads();
  for (int b = 0; b < cb; ++b) {
    uint64_t cur_rows[2][kRowsPerItem];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int q = 0; q < kRowsPerItem; ++q) cur_rows[s_][q] = pre[s_][q];
    if (b + 1 < cb) fetch(b + 1);          // in flight while this block is resolved
    if (wave == 0) {
      const int row = b * 64 + lane;
      const uint64_t d = diag[row];
      const int rows = min(64, n - b * 64);
      const uint64_t valid = (rows == 64) ? ~0ull : ((1ull << rows) - 1ull);
      uint64_t cur = mr3d_uniform64(remv[b]);
      uint64_t cand = (uint64_t)__ballot(d != 0ull) & ~cur & valid;
      while (cand) {
        const int t = __builtin_ctzll(cand);
        const uint64_t dd = mr3d_readlane64(d, t);
        cur |= dd;
        cand &= ~dd;
        cand &= ~(1ull << t);
      }
      if (lane == 0) keptb[b] = ~cur & valid;
    }
    __syncthreads();
    const uint64_t kept = keptb[b];
    const int later = cb - (b + 1);
    const int items = later * G;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int it = tid + s_ * kSweepThreads;
      if (it < items) {
        const int w = b + 1 + it % later, g = it / later;
        uint64_t acc = 0;
#pragma unroll
        for (int q = 0; q < kRowsPerItem; ++q) acc |= ((kept >> (g * kRowsPerItem + q)) & 1ull) ? cur_rows[s_][q] : 0ull;
        if (acc) atomicOr((unsigned long long *)&remv[w], (unsigned long long)acc);
      }
    }
    __syncthreads();
  }
  // kept bits -> ORIGINAL index space (one gather of order[] for the whole set), then ascending compaction
  for (int r = tid; r < n; r += kSweepThreads)
    if ((keptb[r >> 6] >> (r & 63)) & 1ull) {
      const int oi = order[r];
      atomicOr((unsigned int *)flags + (oi >> 5), 1u << (oi & 31));
    }
  __syncthreads();
  if (wave == 0) {
    int base = 0;
    for (int w = 0; w < cb; ++w) {
      const uint64_t f = flags[w];
      if ((f >> lane) & 1ull) keep_out[base + __popcll(f & ((1ull << lane) - 1ull))] = (int64_t)w * 64 + lane;
      base += __popcll(f);
    }
    if (lane == 0) *num_keep = base;
  }
}

// ---- greedy sweep, cb <= 32 blocks (n <= 2048: the RPN's nms_pre = 2000 sets): the whole upper triangle on chip -----------
// Prefetching one block ahead still waits ~1 us per block for the mask rows (44 us for 32 blocks).  Here the chain never
// touches global memory: 1024 threads fetch ALL words above the diagonal once -- an item = 8 mask rows x 1 word, <= 4 items
// per thread, two of them kept in registers and two in LDS (128 KB; all four in registers spill) -- with every load in
// flight at the same time; then per block: wave 0 resolves the diagonal from LDS, one barrier, the threads that own items
// of this block OR their kept rows into the later blocks' suppression words in LDS, one barrier.
constexpr int kRegThreads = 1024;
constexpr int kRegMaxCb = 32;
constexpr int kRegSlots = 4;        // ceil(31 * 32 / 2 * 8 items / 1024 threads)
constexpr int kRegSlotsR = 2;       // slots held in registers; the rest in LDS as [slot][row][thread]
constexpr size_t kRegLdsBytes = sizeof(uint64_t) * (3 * kRegMaxCb + kRegMaxCb * 64 + (kRegSlots - kRegSlotsR) * kRowsPerItem * kRegThreads);
__global__ __launch_bounds__(kRegThreads) void nms3d_sweep_reg_kernel(const uint64_t *__restrict__ mask_all,
                                                                     const int *__restrict__ order_all, SegTable seg,
                                                                     int64_t *__restrict__ keep_all,
                                                                     int32_t *__restrict__ num_keep_all) {
  extern __shared__ __attribute__((aligned(16))) uint64_t sm[];
  uint64_t *remv = sm, *flags = sm + kRegMaxCb, *keptb = sm + 2 * kRegMaxCb, *diag = sm + 3 * kRegMaxCb;
  uint64_t *lrows = diag + kRegMaxCb * 64;
  const int sg = blockIdx.x;
  const int n = seg_rows(seg, sg), cb = (n + 63) / 64;
  int32_t *num_keep = num_keep_all + sg;
  if (n == 0) {
    if (threadIdx.x == 0) *num_keep = 0;
    return;
  }
  const uint64_t *mask = mask_all + seg.mask_off[sg];
  const int *order = order_all + seg.row0[sg];
  int64_t *keep_out = keep_all + seg.row0[sg];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 3 * kRegMaxCb) sm[tid] = 0;
  for (int r = tid; r < cb * 64; r += kRegThreads) diag[r] = r < n ? mask[(size_t)r * cb + (r >> 6)] : 0ull;
  // item id -> (block b, word w > b, row group g): ids run block by block; block b has (cb - 1 - b) * 8 items
  constexpr int G = 64 / kRowsPerItem;
  uint64_t rows[kRegSlotsR][kRowsPerItem];
  int meta[kRegSlots];                 // b << 16 | w << 8 | g, or -1
#pragma unroll
  for (int s_ = kRegSlots - 1; s_ >= 0; --s_) {      // the LDS-resident slots first: their data passes through registers
    int id = s_ * kRegThreads + tid, b = 0;
    while (b < cb && id >= (cb - 1 - b) * G) { id -= (cb - 1 - b) * G; ++b; }
    const bool on = b < cb - 1;
    const int later = cb - 1 - b;
    const int w = on ? b + 1 + id % later : 0, g = on ? id / later : 0;
    meta[s_] = on ? (b << 16 | w << 8 | g) : -1;
    const unsigned off = (unsigned)((b * 64 + g * kRowsPerItem) * cb + w);
#pragma unroll
    for (int q = 0; q < kRowsPerItem; ++q) {
      const int row = b * 64 + g * kRowsPerItem + q;
      const uint64_t v = (on && row < n) ? mask[off + (unsigned)(q * cb)] : 0ull;
      if (s_ >= kRegSlotsR) lrows[((s_ - kRegSlotsR) * kRowsPerItem + q) * kRegThreads + tid] = v;
      else rows[s_][q] = v;
    }
  }
  int blk_lo[kRegSlots], blk_hi[kRegSlots];            // wave-uniform block range of each slot (an idle lane has meta = -1)
#pragma unroll
  for (int s_ = 0; s_ < kRegSlots; ++s_) {
    const int first = __builtin_amdgcn_readfirstlane(meta[s_]), last = __builtin_amdgcn_readlane(meta[s_], 63);
    blk_lo[s_] = first < 0 ? 0x7fffffff : (first >> 16);
    blk_hi[s_] = first < 0 ? -1 : (last < 0 ? cb : (last >> 16));
  }
  __syncthreads();
  for (int b = 0; b < cb; ++b) {
    if (wave == 0) {
      const uint64_t d = diag[b * 64 + lane];
      const int nr = min(64, n - b * 64);
      const uint64_t valid = (nr == 64) ? ~0ull : ((1ull << nr) - 1ull);
      uint64_t cur = mr3d_uniform64(remv[b]);
      uint64_t cand = (uint64_t)__ballot(d != 0ull) & ~cur & valid;
      while (cand) {
        const int t = __builtin_ctzll(cand);
        const uint64_t dd = mr3d_readlane64(d, t);
        cur |= dd;
        cand &= ~dd;
        cand &= ~(1ull << t);
      }
      if (lane == 0) keptb[b] = ~cur & valid;
    }
    __syncthreads();
    const uint64_t kept = keptb[b];
    // a wave's 64 items of one slot are consecutive ids, i.e. they belong to one block or two neighbouring ones: the test
    // whether ANY of them is due is two scalar compares (16 waves evaluating four per-lane tests per block was the largest
    // share of the loop)
#pragma unroll
    for (int s_ = 0; s_ < kRegSlots; ++s_)
      if (b >= blk_lo[s_] && b <= blk_hi[s_] && (meta[s_] >> 16) == b) {
        const int w = (meta[s_] >> 8) & 255, g = meta[s_] & 255;
        const unsigned kb = (unsigned)(kept >> (g * kRowsPerItem)) & 255u;
        uint64_t acc = 0;
#pragma unroll
        for (int q = 0; q < kRowsPerItem; ++q) {
          const uint64_t v = s_ >= kRegSlotsR ? lrows[((s_ - kRegSlotsR) * kRowsPerItem + q) * kRegThreads + tid] : rows[s_ < kRegSlotsR ? s_ : 0][q];
          acc |= ((kb >> q) & 1u) ? v : 0ull;
        }
        if (acc) atomicOr((unsigned long long *)&remv[w], (unsigned long long)acc);
      }
    __syncthreads();
  }
  for (int r = tid; r < n; r += kRegThreads)
    if ((keptb[r >> 6] >> (r & 63)) & 1ull) {
      const int oi = order[r];
      atomicOr((unsigned int *)flags + (oi >> 5), 1u << (oi & 31));
    }
  __syncthreads();
  if (wave == 0) {
    int base = 0;
    for (int w = 0; w < cb; ++w) {
      const uint64_t f = flags[w];
      if ((f >> lane) & 1ull) keep_out[base + __popcll(f & ((1ull << lane) - 1ull))] = (int64_t)w * 64 + lane;
      base += __popcll(f);
    }
    if (lane == 0) *num_keep = base;
  }
}


// =====================================================================================================================
// Round 4: the n <= 2048 pipeline (every RPN set: nms_pre = 2000) in suppressor form.
//   rank sort   : a box's sorted position is the number of keys below its own -- no sorting network; the whole chip counts
//   T2 words    : T2[w][j] = bits of the boxes i of sorted block w (i < j) that suppress box j (IoU3D(i, j) > thr): the
//                 transpose of the reference's mask words (nms_kernel.cu:86-126), same bits -- devIoU3d is symmetric in its
//                 two boxes, op by op (max / min / the commutative Sa + Sb)
//   chain sweep : a wave owns a 64-box block and keeps its boxes' words in registers; the kept word of block w is consumed
//                 as soon as it is published, so the serial chain per block is ONE and + ballot + the in-block resolution --
//                 no barrier, no atomics, no global memory (the barrier form spent 1.15 us per block, this one ~0.15)
// =====================================================================================================================
constexpr int kRankThreads = 256;
constexpr int kChainMaxCb = 32;
constexpr int kChainThreads = 1024;

// grid (64-box group, segment).  The workgroup stages every key of its segment in LDS; wave q counts the keys of its quarter
// that sort before each of the group's 64 boxes (uniform LDS reads: broadcast), the four partial counts add up to the rank.
__global__ __launch_bounds__(kRankThreads) void nms3d_rank_sort_kernel(const float *__restrict__ dets_all, SegTable seg, int stride,
                                                                      int *__restrict__ order_all, float4 *__restrict__ boxes_all) {
  __shared__ __attribute__((aligned(16))) uint64_t keys[kSortLdsMax + 8];
  __shared__ int part[4][64];
  const int sg = blockIdx.y;
  const int n = seg_rows(seg, sg);
  const int i0 = blockIdx.x * 64;
  if (i0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *dets = dets_all + (size_t)seg.row0[sg] * stride;
  const int npad = (n + 7) & ~7;
  for (int j = tid; j < npad; j += kRankThreads) keys[j] = j < n ? make_key(dets[(size_t)j * stride + 6], j) : ~0ull;
  __syncthreads();
  const int i = i0 + lane;
  const uint64_t mine = i < n ? keys[i] : 0ull;
  // quarter of the key range, in pairs (ds_read_b128); padding keys are ~0: never below a real key
  const int pairs = npad >> 1, per = (pairs + 3) >> 2;
  const int p0 = wave * per, p1 = min(pairs, p0 + per);
  int cnt = 0;
  const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(keys);
#pragma unroll 4
  for (int p = p0; p < p1; ++p) {
    const ulonglong2 kk = k2[p];
    cnt += (kk.x < mine) ? 1 : 0;
    cnt += (kk.y < mine) ? 1 : 0;
  }
  part[wave][lane] = cnt;
  __syncthreads();
  if (wave == 0 && i < n) {
    const int rank = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    order_all[seg.row0[sg] + rank] = i;
    store_box(dets, stride, i, boxes_all + seg.box_off[sg], rank);
  }
}

// one wave per (suppressor block r, column block c >= r): lane = column box j, loop over the 64 boxes of block r (LDS).
// SKIP: threshold >= 0 -- an iteration in which no lane's pair intersects leaves every bit clear (0 / x > thr is false), so
// the division is skipped wave-uniformly; sets of small boxes spend most iterations there.
// v_max_f32 / v_min_f32 as the hardware executes them (IEEE maxNum / minNum: a quiet NaN operand yields the other one, like CUDA's
// fmaxf / fminf in devIoU3d).  Through fmaxf() the compiler first canonicalises every operand it cannot prove quiet -- six extra
// `v_max x, x` per pair for the box read from LDS, 15 % of the loop.
__device__ __forceinline__ float hw_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hw_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// grid (ceil(pairs / 4), 1, segment), 256 threads: wave w of workgroup b takes block pair p = 4 b + w of the UPPER TRIANGLE only,
// p = c (c + 1) / 2 + r with r <= c (round 6: the square grid of one-wave workgroups launched 1024 workgroups per set of which 496
// left at once -- a workgroup's turnover costs ~24 ns of dispatch whatever it does, tools/diag/roi_rows_stamps.py -- and 4x the
// workgroups for the pairs that stay).  The waves of a workgroup are independent: no barrier after the wave's own LDS staging.
constexpr int kT2Waves = 4;
template <bool SKIP>
__global__ __launch_bounds__(kT2Waves * MR3D_WAVE) void nms3d_mask_t2_kernel(const float4 *__restrict__ boxes_all, SegTable seg, float thr,
                                                                            uint64_t *__restrict__ mask_all) {
  const int sg = blockIdx.z;
  const int n = seg_rows(seg, sg), cb = (n + 63) / 64;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.x * kT2Waves + wave;
  int c = (int)((sqrtf(8.f * (float)pair + 1.f) - 1.f) * 0.5f);      // p = c (c + 1) / 2 + r: the float estimate is off by at most one
  while (c * (c + 1) / 2 > pair) --c;
  while ((c + 1) * (c + 2) / 2 <= pair) ++c;
  const int r = pair - c * (c + 1) / 2;
  if (c >= cb) return;                                               // (wave-uniform)
  const float4 *boxes = boxes_all + seg.box_off[sg];
  uint64_t *t2 = mask_all + seg.mask_off[sg];
  __shared__ float4 rbx_all[kT2Waves][2 * MR3D_WAVE];
  float4 *rbx = rbx_all[wave];
  const int lane = threadIdx.x & 63;
  const int ridx = r * 64 + lane;
  if (ridx < n) {
    rbx[2 * lane] = boxes[2 * ridx];
    rbx[2 * lane + 1] = boxes[2 * ridx + 1];
  }
  __builtin_amdgcn_wave_barrier();                                    // (the wave's own LDS rows: its DS operations complete in order)
  const int j = c * 64 + lane;
  uint64_t t = 0;
  if (j < n) {
    const float4 b0 = boxes[2 * j], b1 = boxes[2 * j + 1];          // b = the later box (column), a = the earlier (row): devIoU3d(a, b)
    const int rsz = min(n - r * 64, 64);
    const int stop = (r == c) ? lane : rsz;                           // diagonal block: earlier boxes only
    float4 a0 = rbx[0], a1 = rbx[1];
    for (int k = 0; k < rsz; ++k) {
      const float4 n0 = rbx[2 * min(k + 1, rsz - 1)], n1 = rbx[2 * min(k + 1, rsz - 1) + 1];      // the next box travels under this pair's arithmetic
      float left = hw_max(a0.x, b0.x), right = hw_min(a0.z, b0.z);
      float top = hw_max(a0.y, b0.y), bottom = hw_min(a0.w, b0.w);
      float front = hw_max(a1.x, b1.x), back = hw_min(a1.y, b1.y);
      float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
      float depth = fmaxf(back - front + 1, 0.f);
      float interS = width * height * depth;
      const float sa = a1.z;
      a0 = n0; a1 = n1;
      if (SKIP && __ballot(interS > 0.f) == 0ull) continue;
      float iou = interS / (sa + b1.z - interS);
      if (k < stop && iou > thr) t |= 1ull << k;
    }
  }
  t2[(size_t)r * (cb * 64) + j] = t;          // word-major: a wave of the sweep reads 64 consecutive boxes' words in one 512-byte run
}

// kept word of a sorted block, handed from the wave that resolved it to the waves of the later blocks through LDS: the word, then
// (release) its ready flag; a consumer spins on the flag (acquire) and reads the word.  Workgroup-scope atomics on LDS.
__device__ __forceinline__ void chain_publish(uint64_t *kept_w, unsigned *ready_w, uint64_t kept) {
  __hip_atomic_store(kept_w, kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_store(ready_w, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint64_t chain_wait(const uint64_t *kept_w, const unsigned *ready_w) {
  while (__hip_atomic_load(ready_w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(1);
  return __hip_atomic_load(kept_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(kChainThreads) void nms3d_sweep_chain_kernel(const uint64_t *__restrict__ mask_all,
                                                                         const int *__restrict__ order_all, SegTable seg,
                                                                         int64_t *__restrict__ keep_all, int32_t *__restrict__ num_keep_all) {
  __shared__ uint64_t pub[kChainMaxCb];               // kept bits of sorted block w ...
  __shared__ unsigned ready[kChainMaxCb];             // ... valid once ready[w] != 0
  __shared__ uint64_t flags[kChainMaxCb];             // kept bits in ORIGINAL index space
  __shared__ int wbase[kChainMaxCb + 1];
  const int sg = blockIdx.x;
  const int n = seg_rows(seg, sg), cb = (n + 63) / 64;
  int32_t *num_keep = num_keep_all + sg;
  if (n == 0) {
    if (threadIdx.x == 0) *num_keep = 0;
    return;
  }
  const uint64_t *t2 = mask_all + seg.mask_off[sg];
  const int *order = order_all + seg.row0[sg];
  int64_t *keep_out = keep_all + seg.row0[sg];
  const int tid = threadIdx.x, lane = tid & 63;
  const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < kChainMaxCb) { pub[tid] = 0ull; ready[tid] = 0u; flags[tid] = 0ull; }
  // wave v owns sorted blocks v and 31 - v: 33 words between them whatever v is.  X[x], x <= v: word x of block v's boxes (x == v:
  // the diagonal word); x > v: word x - v - 1 of block 31 - v's boxes (x == 32: its diagonal word).  Every index is wave-uniform.
  const int bA = v, bB = kChainMaxCb - 1 - v;
  const int ldt = cb * 64;
  uint64_t X[kChainMaxCb + 1];
#pragma unroll
  for (int x = 0; x <= kChainMaxCb; ++x) {
    const bool isA = x <= v;
    const int b = isA ? bA : bB, w = isA ? x : x - v - 1;
    X[x] = (b < cb) ? t2[(size_t)w * ldt + b * 64 + lane] : 0ull;
  }
  __syncthreads();
  uint64_t acc = 0;
  bool keptA = false, keptB = false;
#pragma unroll
  for (int x = 0; x <= kChainMaxCb; ++x) {
    const bool isA = x <= v;
    const int b = isA ? bA : bB, w = isA ? x : x - v - 1;
    if (b >= cb) continue;                              // wave-uniform
    if (w < b) {
      const uint64_t kept = chain_wait(&pub[w], &ready[w]);
      acc |= X[x] & mr3d_uniform64(kept);
    } else {
      // the block's own resolution: a box is kept iff no kept box of an earlier block (acc) and no kept EARLIER box of this
      // block suppresses it.  Boxes without an in-block suppressor are decided at once; the others in ascending order.
      const int i = b * 64 + lane;
      const uint64_t D = X[x];
      const bool free_ = i < n && acc == 0ull;
      uint64_t kept = (uint64_t)__ballot(free_ && D == 0ull);
      uint64_t cand = (uint64_t)__ballot(free_ && D != 0ull);
      while (cand) {
        const int t = __builtin_ctzll(cand);
        const uint64_t d = mr3d_readlane64(D, t);
        if ((d & kept) == 0ull) kept |= 1ull << t;
        cand &= cand - 1ull;
      }
      if (lane == 0) chain_publish(&pub[b], &ready[b], kept);
      const bool mine = (kept >> lane) & 1ull;
      if (isA) keptA = mine; else keptB = mine;
      acc = 0ull;
    }
  }
  // kept bits -> ORIGINAL index space, then ascending compaction (nms_kernel.cu:253-256 returns order[keep] sorted)
  if (keptA) { const int oi = order[bA * 64 + lane]; atomicOr((unsigned int *)flags + (oi >> 5), 1u << (oi & 31)); }
  if (keptB) { const int oi = order[bB * 64 + lane]; atomicOr((unsigned int *)flags + (oi >> 5), 1u << (oi & 31)); }
  __syncthreads();
  if (v == 0) {
    const int pc = lane < cb ? __popcll(flags[lane]) : 0;
    int incl = pc;
#pragma unroll
    for (int dlt = 1; dlt < 32; dlt <<= 1) {
      const int o = __shfl_up(incl, dlt);
      if (lane >= dlt) incl += o;
    }
    if (lane < kChainMaxCb) wbase[lane + 1] = incl;
    if (lane == 0) wbase[0] = 0;
    if (lane == kChainMaxCb - 1) *num_keep = incl;
  }
  __syncthreads();
  for (int w = v; w < cb; w += kChainThreads / 64) {
    const uint64_t f = flags[w];
    if ((f >> lane) & 1ull) keep_out[wbase[w] + __popcll(f & ((1ull << lane) - 1ull))] = (int64_t)w * 64 + lane;
  }
}

void launch_sweep(const uint64_t *mask, const int *order, const SegTable &seg, int num_segments, int max_cb, int64_t *keep,
                  int32_t *num_keep, hipStream_t stream) {
  static bool reg_ok = false, reg_tried = false;      // 144 KB of LDS needs the opt-in attribute (one CU has 160 KB)
  if (!reg_tried) {
    reg_tried = true;
    reg_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(nms3d_sweep_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)kRegLdsBytes) == hipSuccess;
    if (!reg_ok) (void)hipGetLastError();
  }
  if (max_cb <= kRegMaxCb && reg_ok) {
    nms3d_sweep_reg_kernel<<<num_segments, kRegThreads, kRegLdsBytes, stream>>>(mask, order, seg, keep, num_keep);
  } else if (max_cb <= kFastMaxCb) {
    const size_t lds = sizeof(uint64_t) * (3 * (size_t)max_cb + 64 * (size_t)max_cb);
    nms3d_sweep_fast_kernel<<<num_segments, kSweepThreads, lds, stream>>>(mask, order, seg, keep, num_keep);
  } else {
    const size_t lds = sizeof(uint64_t) * (2 * (size_t)max_cb + 1);
    nms3d_sweep_kernel<<<num_segments, kSweepThreads, lds, stream>>>(mask, order, seg, keep, num_keep);
  }
}

// (Round 4, measured and rejected: the three phases as ONE persistent launch -- one 16-wave workgroup per CU walking the phases' work
// items, two grid barriers on a device counter where the kernel boundaries are.  Correct, but 82 us against 62 us for the ten
// sets and 45 us against 22 us for a single 192-box set: a grid barrier costs more than a kernel boundary here -- every
// workgroup's release / acquire fence writes back and invalidates its XCD's L2, and the barrier waits for the launch ramp of all
// 256 workgroups twice.)
// n <= 2048 per set: T2 words + chain sweep
void launch_chain(const float4 *boxes, const int *order, const SegTable &seg, int num_segments, int max_cb, float thr, uint64_t *mask,
                  int64_t *keep, int32_t *num_keep, hipStream_t stream) {
  const int pairs = max_cb * (max_cb + 1) / 2;
  const dim3 grid((pairs + kT2Waves - 1) / kT2Waves, 1, num_segments);
  if (thr >= 0.f) nms3d_mask_t2_kernel<true><<<grid, kT2Waves * MR3D_WAVE, 0, stream>>>(boxes, seg, thr, mask);
  else nms3d_mask_t2_kernel<false><<<grid, kT2Waves * MR3D_WAVE, 0, stream>>>(boxes, seg, thr, mask);
  nms3d_sweep_chain_kernel<<<num_segments, kChainThreads, 0, stream>>>(mask, order, seg, keep, num_keep);
}

struct NmsWorkspace {
  uint64_t *keys;
  int *order;
  float4 *boxes;
  uint64_t *mask;
  size_t total;
};

NmsWorkspace carve(void *ws, int n) {
  const int np2 = mr3d_next_pow2(n);
  const size_t cb = (size_t)(n + 63) / 64;
  size_t off = 0;
  NmsWorkspace w;
  char *base = (char *)ws;
  w.keys = (uint64_t *)(base + off);  off += mr3d_align_up(n > kSortLdsMax ? sizeof(uint64_t) * np2 : 0, 256);
  w.order = (int *)(base + off);      off += mr3d_align_up(sizeof(int) * (size_t)n, 256);
  w.boxes = (float4 *)(base + off);   off += mr3d_align_up(sizeof(float) * 8 * (size_t)n, 256);
  w.mask = (uint64_t *)(base + off);  off += mr3d_align_up(sizeof(uint64_t) * 64 * cb * cb, 256);          // T2 form: cb words x 64 cb boxes (>= n * cb)
  w.total = off;
  return w;
}

}  // namespace

extern "C" size_t mr3d_nms3d_workspace_bytes(int n) {
  if (n <= 0) return 256;
  return carve(nullptr, n).total;
}

extern "C" int mr3d_nms3d(const float *dets, int n, int row_stride, float iou_threshold, int64_t *keep_out,
                          int32_t *num_keep_out, void *workspace, size_t workspace_bytes, mr3d_stream_t stream_) {
  hipStream_t stream = mr3d_stream(stream_);
  if (n < 0 || !num_keep_out) return MR3D_ERR_INVALID_ARG;
  if (n == 0) {
    return hipMemsetAsync(num_keep_out, 0, sizeof(int32_t), stream) == hipSuccess ? MR3D_OK : MR3D_ERR_LAUNCH;
  }
  if (!dets || !keep_out || row_stride < 7) return MR3D_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < mr3d_nms3d_workspace_bytes(n) || ((uintptr_t)workspace & 255))
    return MR3D_ERR_WORKSPACE;
  const int cb = (n + 63) / 64;
  const size_t sweep_lds = sizeof(uint64_t) * (2 * (size_t)cb + 1);
  if (sweep_lds > 64 * 1024) return MR3D_ERR_UNSUPPORTED;   // n > ~262k boxes
  NmsWorkspace w = carve(workspace, n);
  const int np2 = mr3d_next_pow2(n);
  SegTable seg;
  seg.num = 1; seg.row0[0] = 0; seg.n[0] = n; seg.box_off[0] = 0; seg.mask_off[0] = 0; seg.n_dev = nullptr;

  if (n <= kSortLdsMax) {
    nms3d_rank_sort_kernel<<<dim3(cb, 1), kRankThreads, 0, stream>>>(dets, seg, row_stride, w.order, w.boxes);
  } else {
    const int th = 256, bl = (np2 + th - 1) / th;
    nms3d_keys_init_kernel<<<bl, th, 0, stream>>>(dets, n, row_stride, np2, w.keys);
    for (int k = 2; k <= np2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) nms3d_bitonic_step_kernel<<<bl, th, 0, stream>>>(w.keys, np2, j, k);
    nms3d_gather_kernel<<<(n + th - 1) / th, th, 0, stream>>>(dets, n, row_stride, w.keys, w.order, w.boxes);
  }
  if (cb <= kChainMaxCb) {          // (the sort above also serves: the chain only needs sorted boxes + order)
    launch_chain(w.boxes, w.order, seg, 1, cb, iou_threshold, w.mask, keep_out, num_keep_out, stream);
    return mr3d_launch_status();
  }
  nms3d_mask_kernel<<<dim3(cb, cb, 1), MR3D_WAVE, 0, stream>>>(w.boxes, seg, iou_threshold, w.mask);
  launch_sweep(w.mask, w.order, seg, 1, cb, keep_out, num_keep_out, stream);
  return mr3d_launch_status();
}

// ---- batched: several independent box sets in one launch sequence ----------------------------------------------
namespace {
struct BatchLayout {
  SegTable seg;
  size_t order_off, boxes_off, mask_off, total;
  int max_cb;
};
// 0 on success
int batch_layout(const int *seg_rows_host, int num_segments, BatchLayout &L) {
  if (!seg_rows_host || num_segments <= 0 || num_segments > kMaxSeg) return MR3D_ERR_INVALID_ARG;
  long long rows = 0, boxes = 0, mask = 0;
  L.max_cb = 0;
  L.seg.num = num_segments;
  for (int s = 0; s < num_segments; ++s) {
    const int n = seg_rows_host[s];
    if (n < 0) return MR3D_ERR_INVALID_ARG;
    if (n > kSortLdsMax) return MR3D_ERR_UNSUPPORTED;          // larger sets: one mr3d_nms3d call each
    const int cb = (n + 63) / 64;
    L.seg.row0[s] = (int)rows; L.seg.n[s] = n; L.seg.box_off[s] = boxes; L.seg.mask_off[s] = mask;
    rows += n; boxes += 2LL * n; mask += 64LL * cb * cb;          // T2 form (>= n * cb words)
    L.max_cb = cb > L.max_cb ? cb : L.max_cb;
  }
  for (int s = num_segments; s < kMaxSeg; ++s) { L.seg.row0[s] = 0; L.seg.n[s] = 0; L.seg.box_off[s] = 0; L.seg.mask_off[s] = 0; }
  size_t off = 0;
  L.order_off = off; off += mr3d_align_up(sizeof(int) * (size_t)rows, 256);
  L.boxes_off = off; off += mr3d_align_up(sizeof(float4) * (size_t)boxes, 256);
  L.mask_off = off;  off += mr3d_align_up(sizeof(uint64_t) * (size_t)mask, 256);
  L.total = off + 256;
  return MR3D_OK;
}
}  // namespace

extern "C" size_t mr3d_nms3d_batched_workspace_bytes(const int *seg_rows_host, int num_segments) {
  BatchLayout L;
  return batch_layout(seg_rows_host, num_segments, L) == MR3D_OK ? L.total : 0;
}

extern "C" int mr3d_nms3d_batched(const float *dets, const int *seg_rows_host, int num_segments, int row_stride,
                                  float iou_threshold, int64_t *keep_out, int32_t *num_keep_out, void *workspace,
                                  size_t workspace_bytes, mr3d_stream_t stream_) {
  return mr3d_nms3d_batched_dev(dets, seg_rows_host, nullptr, num_segments, row_stride, iou_threshold, keep_out, num_keep_out,
                                workspace, workspace_bytes, stream_);
}

extern "C" int mr3d_nms3d_batched_dev(const float *dets, const int *seg_rows_host, const int32_t *seg_rows_dev, int num_segments,
                                      int row_stride, float iou_threshold, int64_t *keep_out, int32_t *num_keep_out,
                                      void *workspace, size_t workspace_bytes, mr3d_stream_t stream_) {
  hipStream_t stream = mr3d_stream(stream_);
  BatchLayout L;
  const int st = batch_layout(seg_rows_host, num_segments, L);
  if (st != MR3D_OK) return st;
  L.seg.n_dev = seg_rows_dev;
  if (!num_keep_out || row_stride < 7) return MR3D_ERR_INVALID_ARG;
  if (L.max_cb == 0)          // every segment empty
    return hipMemsetAsync(num_keep_out, 0, sizeof(int32_t) * num_segments, stream) == hipSuccess ? MR3D_OK : MR3D_ERR_LAUNCH;
  if (!dets || !keep_out) return MR3D_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return MR3D_ERR_WORKSPACE;
  char *base = (char *)workspace;
  int *order = (int *)(base + L.order_off);
  float4 *boxes = (float4 *)(base + L.boxes_off);
  uint64_t *mask = (uint64_t *)(base + L.mask_off);
  nms3d_rank_sort_kernel<<<dim3(L.max_cb, num_segments), kRankThreads, 0, stream>>>(dets, L.seg, row_stride, order, boxes);
  if (L.max_cb <= kChainMaxCb) {
    launch_chain(boxes, order, L.seg, num_segments, L.max_cb, iou_threshold, mask, keep_out, num_keep_out, stream);
    return mr3d_launch_status();
  }
  nms3d_mask_kernel<<<dim3(L.max_cb, L.max_cb, num_segments), MR3D_WAVE, 0, stream>>>(boxes, L.seg, iou_threshold, mask);
  launch_sweep(mask, order, L.seg, num_segments, L.max_cb, keep_out, num_keep_out, stream);
  return mr3d_launch_status();
}

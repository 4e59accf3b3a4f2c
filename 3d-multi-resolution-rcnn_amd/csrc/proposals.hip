// RPN proposal selection on the device, without a host round trip.
//
// RPNHead3D.get_bboxes_single (mmdet/models/anchor_heads/rpn_head_3d.py:72-149) per pyramid level: sigmoid -> drop the anchors
// flagged out by the last loss() call (:95-106) -> top-`nms_pre` by score (:108-112) -> delta2bbox3D -> nms -> first
// `nms_post` (:133-135); across levels: top-`max_num` by score (:141-148).  In the reference every step synchronises (boolean
// gathers, torch.topk on 655,360 P2 scores, nms' D2H mask copy).  Here:
//   * mr3d_rpn_select_topk: an exact radix select over the order-preserving bits of the LOGITS of all levels at once (the
//     sigmoid is monotone, so the k largest logits are k largest scores; which of several EQUAL scores are taken is as
//     unspecified as in torch.topk).  Three histogram passes (12 + 12 + 8 bits), each started by every workgroup re-deriving
//     the previous pass's pivot bin from the histogram (no separate scan kernel), then a counting pass and an ordered
//     compaction: the selected anchors come out in ascending anchor order with their count in device memory.
//   * the decode (bbox_ops.hip) and NMS3D (nms3d.hip, mr3d_nms3d_batched_dev) kernels take that count from device memory;
//   * mr3d_rpn_merge_levels: first nms_post kept rows of every level, concatenated, top max_num by (score desc, position asc)
//     with one register-resident bitonic sort, written as a fixed [max_num, 7] block (zero rows behind the count).
#include "mr3d_common.h"
#include "sort_regs.h"

namespace {

constexpr int kT = 256;
constexpr int kPerThread = 16;                       // elements of one thread: a contiguous run (output order = anchor order)
constexpr int kPerWg = kT * kPerThread;
constexpr int kMaxLevels = 8;
constexpr int kBins0 = 4096, kBins1 = 4096, kBins2 = 256;      // 12 + 12 + 8 bits of the 32-bit key
constexpr int kHistWords = kBins0 + kBins1 + kBins2;

struct LevelTab {
  int num;
  const float *logits[kMaxLevels];
  const uint8_t *valid[kMaxLevels];
  long long sa[kMaxLevels], sd[kMaxLevels], sh[kMaxLevels], sw[kMaxLevels];
  int A[kMaxLevels], D[kMaxLevels], H[kMaxLevels], W[kMaxLevels];
  int n[kMaxLevels];             // anchors of the level
  int k[kMaxLevels];             // min(n, nms_pre): capacity of its output block
  int select[kMaxLevels];        // 1: n > nms_pre (flags + top-k); 0: every anchor, in order
  int out0[kMaxLevels];          // first output row
  int wg0[kMaxLevels];           // first workgroup (of the select levels)
  long long key0[kMaxLevels];    // first element of its key block
};

// per-level state in the workspace: [hist 12 | hist 12 | hist 8] uint32, then per-workgroup (gt, eq) counts
struct Pivot { uint32_t prefix; uint32_t k_rem; };

__device__ __forceinline__ uint32_t key_of(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // ascending, order-preserving; > 0 for every non-NaN
}
__device__ __forceinline__ float float_of(uint32_t key) {
  return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

// flat (h, w, d, a) index (rpn_head_3d.py:87: permute(2, 3, 1, 0)) -> element offset in the [A, D, H, W] logits
__device__ __forceinline__ long long logit_offset(const LevelTab &t, int l, int f) {
  const int a = f % t.A[l];
  int q = f / t.A[l];
  const int d = q % t.D[l]; q /= t.D[l];
  const int w = q % t.W[l];
  const int h = q / t.W[l];
  return a * t.sa[l] + d * t.sd[l] + h * t.sh[l] + w * t.sw[l];
}

// workgroups are dealt to the select levels in order; the first select level starts at workgroup 0
__device__ __forceinline__ int level_of_wg(const LevelTab &t, int wg, int &local) {
  int l = 0;
  for (int j = 0; j < t.num; ++j)
    if (t.select[j] && wg >= t.wg0[j]) l = j;
  local = wg - t.wg0[l];
  return l;
}

// Pivot of a finished histogram pass: scanning the bins from the top, the bin in which the running count reaches k_rem.
// Every workgroup computes it for itself (4096 counters from L2, one block reduction).  Returns (bin, count above the bin).
__device__ void find_pivot(const uint32_t *__restrict__ hist, int bins, uint32_t k_rem, uint32_t *sh, int &bin, uint32_t &above) {
  const int tid = threadIdx.x;
  const int per = bins / kT > 0 ? bins / kT : 1;              // bins per thread; thread 0 owns the TOP bins
  uint32_t mine = 0;
  if (tid * per < bins)
    for (int j = 0; j < per; ++j) mine += hist[bins - 1 - (tid * per + j)];
  sh[tid] = mine;
  __syncthreads();
  // inclusive scan over the 256 partial sums (Hillis-Steele; 8 rounds)
  for (int o = 1; o < kT; o <<= 1) {
    const uint32_t v = tid >= o ? sh[tid - o] : 0u;
    __syncthreads();
    sh[tid] += v;
    __syncthreads();
  }
  const uint32_t incl = sh[tid], excl = incl - mine;
  __shared__ int s_bin;
  __shared__ uint32_t s_above;
  if (tid == 0) { s_bin = 0; s_above = sh[kT - 1]; }          // fewer than k_rem elements in total: bin 0, all above it
  __syncthreads();
  if (tid * per < bins && excl < k_rem && incl >= k_rem) {
    uint32_t run = excl;
    for (int j = 0; j < per; ++j) {
      const int b = bins - 1 - (tid * per + j);
      const uint32_t c = hist[b];
      if (run + c >= k_rem) { s_bin = b; s_above = run; break; }
      run += c;
    }
  }
  __syncthreads();
  bin = s_bin; above = s_above;
  __syncthreads();
}

// PASS 0: keys of a workgroup's 4096 anchors (flags applied: a flagged-out anchor gets key 0, below every real key) are
// written to the compact key block and counted into the 12-bit histogram.  PASS 1 / 2: re-derive the pivot of the pass
// before, count the next digit of the keys that match the prefix.
template <int PASS>
__global__ __launch_bounds__(kT) void rpn_select_hist_kernel(LevelTab t, int nms_pre, uint32_t *__restrict__ keys, uint32_t *__restrict__ ws) {
  __shared__ uint32_t lh[kBins0];
  __shared__ uint32_t sh[kT];
  int local;
  const int l = level_of_wg(t, blockIdx.x, local);
  uint32_t *hist = ws + (size_t)l * kHistWords;
  const int tid = threadIdx.x;
  constexpr int BINS = PASS == 2 ? kBins2 : kBins0;
  for (int i = tid; i < BINS; i += kT) lh[i] = 0;
  uint32_t prefix = 0;
  if constexpr (PASS >= 1) {
    int b0; uint32_t above0;
    find_pivot(hist, kBins0, (uint32_t)nms_pre, sh, b0, above0);
    prefix = (uint32_t)b0;
    if constexpr (PASS == 2) {
      int b1; uint32_t above1;
      find_pivot(hist + kBins0, kBins1, (uint32_t)nms_pre - above0, sh, b1, above1);
      prefix = (prefix << 12) | (uint32_t)b1;
    }
  }
  __syncthreads();
  uint32_t *kb = keys + t.key0[l];
  const int base = local * kPerWg + tid * kPerThread;
#pragma unroll 4
  for (int j = 0; j < kPerThread; ++j) {
    const int f = base + j;
    if (f >= t.n[l]) break;
    uint32_t key;
    if constexpr (PASS == 0) {
      const float x = t.logits[l][logit_offset(t, l, f)];
      key = (t.valid[l] && !t.valid[l][f]) ? 0u : key_of(x);
      kb[f] = key;
      atomicAdd(&lh[key >> 20], 1u);
    } else if constexpr (PASS == 1) {
      key = kb[f];
      if ((key >> 20) == prefix) atomicAdd(&lh[(key >> 8) & 4095u], 1u);
    } else {
      key = kb[f];
      if ((key >> 8) == prefix) atomicAdd(&lh[key & 255u], 1u);
    }
  }
  __syncthreads();
  uint32_t *gh = hist + (PASS == 0 ? 0 : (PASS == 1 ? kBins0 : kBins0 + kBins1));
  for (int i = tid; i < BINS; i += kT)
    if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// the full pivot key T and how many anchors equal to it are still wanted
__device__ void final_pivot(const uint32_t *hist, int nms_pre, uint32_t *sh, uint32_t &T, uint32_t &eq_take) {
  int b0, b1, b2; uint32_t a0, a1, a2;
  find_pivot(hist, kBins0, (uint32_t)nms_pre, sh, b0, a0);
  find_pivot(hist + kBins0, kBins1, (uint32_t)nms_pre - a0, sh, b1, a1);
  find_pivot(hist + kBins0 + kBins1, kBins2, (uint32_t)nms_pre - a0 - a1, sh, b2, a2);
  T = ((uint32_t)b0 << 20) | ((uint32_t)b1 << 8) | (uint32_t)b2;
  eq_take = (uint32_t)nms_pre - a0 - a1 - a2;
}

// per workgroup: how many of its anchors lie above the pivot / on it
__global__ __launch_bounds__(kT) void rpn_select_count_kernel(LevelTab t, int nms_pre, const uint32_t *__restrict__ keys, uint32_t *__restrict__ ws,
                                                             uint32_t *__restrict__ wgc) {
  __shared__ uint32_t sh[kT];
  __shared__ uint32_t s_gt, s_eq;
  int local;
  const int l = level_of_wg(t, blockIdx.x, local);
  const int tid = threadIdx.x;
  uint32_t T, eq_take;
  final_pivot(ws + (size_t)l * kHistWords, nms_pre, sh, T, eq_take);
  if (tid == 0) { s_gt = 0; s_eq = 0; }
  __syncthreads();
  const uint32_t *kb = keys + t.key0[l];
  const int base = local * kPerWg + tid * kPerThread;
  uint32_t gt = 0, eq = 0;
  for (int j = 0; j < kPerThread; ++j) {
    const int f = base + j;
    if (f >= t.n[l]) break;
    const uint32_t key = kb[f];
    gt += key > T;
    eq += key == T && T != 0u;
  }
  if (gt) atomicAdd(&s_gt, gt);
  if (eq) atomicAdd(&s_eq, eq);
  __syncthreads();
  if (tid == 0) { wgc[2 * blockIdx.x] = s_gt; wgc[2 * blockIdx.x + 1] = s_eq; }
}

// ordered compaction: an anchor above the pivot lands at (#above before it) + min(#equal before it, eq_take); an anchor ON
// the pivot is taken while fewer than eq_take equal ones precede it.  "Before" = lower anchor index: workgroups in order,
// threads in order, a thread's run in order.
__global__ __launch_bounds__(kT) void rpn_select_write_kernel(LevelTab t, int nms_pre, const uint32_t *__restrict__ keys,
                                                             const uint32_t *__restrict__ ws, const uint32_t *__restrict__ wgc,
                                                             long long *__restrict__ flat_out, float *__restrict__ score_out,
                                                             int32_t *__restrict__ counts) {
  __shared__ uint32_t sh[kT];
  __shared__ uint32_t sg[kT], se[kT];
  int local;
  const int l = level_of_wg(t, blockIdx.x, local);
  const int tid = threadIdx.x;
  uint32_t T, eq_take;
  final_pivot(ws + (size_t)l * kHistWords, nms_pre, sh, T, eq_take);
  // counts of the workgroups of this level in front of this one (<= 160 of them) and of the whole level
  const int nwg = (t.n[l] + kPerWg - 1) / kPerWg;
  uint32_t bg = 0, be = 0, tg = 0, te = 0;
  for (int j = tid; j < nwg; j += kT) {
    const uint32_t g = wgc[2 * (t.wg0[l] + j)], e = wgc[2 * (t.wg0[l] + j) + 1];
    tg += g; te += e;
    if (j < local) { bg += g; be += e; }
  }
  sg[tid] = bg; se[tid] = be;
  __syncthreads();
  for (int o = kT / 2; o > 0; o >>= 1) {
    if (tid < o) { sg[tid] += sg[tid + o]; se[tid] += se[tid + o]; }
    __syncthreads();
  }
  const uint32_t wg_gt = sg[0], wg_eq = se[0];
  __syncthreads();
  sg[tid] = tg; se[tid] = te;
  __syncthreads();
  for (int o = kT / 2; o > 0; o >>= 1) {
    if (tid < o) { sg[tid] += sg[tid + o]; se[tid] += se[tid + o]; }
    __syncthreads();
  }
  const uint32_t all_gt = sg[0], all_eq = se[0];
  __syncthreads();
  const uint32_t total = all_gt + (all_eq < eq_take ? all_eq : eq_take);
  if (local == 0 && tid == 0) counts[l] = (int32_t)total;
  // this thread's run
  const uint32_t *kb = keys + t.key0[l];
  const int base = local * kPerWg + tid * kPerThread;
  uint32_t key[kPerThread];
  uint32_t gt = 0, eq = 0;
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const int f = base + j;
    key[j] = f < t.n[l] ? kb[f] : 0u;
    gt += key[j] > T;
    eq += key[j] == T && T != 0u;
  }
  // exclusive scan of (gt, eq) over the threads
  sg[tid] = gt; se[tid] = eq;
  __syncthreads();
  for (int o = 1; o < kT; o <<= 1) {
    const uint32_t g = tid >= o ? sg[tid - o] : 0u, e = tid >= o ? se[tid - o] : 0u;
    __syncthreads();
    sg[tid] += g; se[tid] += e;
    __syncthreads();
  }
  uint32_t pg = wg_gt + sg[tid] - gt, pe = wg_eq + se[tid] - eq;
  long long *fo = flat_out + t.out0[l];
  float *so = score_out + t.out0[l];
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const uint32_t kk = key[j];
    const bool is_gt = kk > T, is_eq = kk == T && T != 0u;
    if (is_gt || (is_eq && pe < eq_take)) {
      const uint32_t pos = pg + (pe < eq_take ? pe : eq_take);
      const float x = float_of(kk);
      fo[pos] = base + j;
      so[pos] = 1.f / (1.f + expf(-x));                    // sigmoid (rpn_head_3d.py:90), torch's expression
    }
    pg += is_gt; pe += is_eq;
  }
  // rows behind the count: a valid anchor index and a score no real row has (the NMS never looks at them)
  for (int r = (int)total + local * kT + tid; r < t.k[l]; r += nwg * kT) { fo[r] = 0; so[r] = -1.f; }
}

// levels that keep every anchor (n <= nms_pre; the reference applies neither flags nor top-k there, :94)
__global__ __launch_bounds__(kT) void rpn_select_all_kernel(LevelTab t, long long *__restrict__ flat_out, float *__restrict__ score_out,
                                                           int32_t *__restrict__ counts) {
  const int l = blockIdx.y;
  if (t.select[l]) return;
  const int f = blockIdx.x * kT + threadIdx.x;
  if (f == 0) counts[l] = t.n[l];
  if (f >= t.n[l]) return;
  const float x = t.logits[l][logit_offset(t, l, f)];
  flat_out[t.out0[l] + f] = f;
  score_out[t.out0[l] + f] = 1.f / (1.f + expf(-x));
}

int fill_table(const mr3d_rpn_level *lv, int num_levels, int nms_pre, LevelTab &t, int &total_wgs, long long &total_keys, int &total_out) {
  if (!lv || num_levels <= 0 || num_levels > kMaxLevels || nms_pre <= 0) return MR3D_ERR_INVALID_ARG;
  t.num = num_levels;
  total_wgs = 0; total_keys = 0; total_out = 0;
  for (int l = 0; l < kMaxLevels; ++l) {
    if (l >= num_levels) {
      t.logits[l] = nullptr; t.valid[l] = nullptr; t.sa[l] = t.sd[l] = t.sh[l] = t.sw[l] = 0;
      t.A[l] = t.D[l] = t.H[l] = t.W[l] = 1; t.n[l] = 0; t.k[l] = 0; t.select[l] = 0; t.out0[l] = 0; t.wg0[l] = 0; t.key0[l] = 0;
      continue;
    }
    const mr3d_rpn_level &L = lv[l];
    if (!L.logits || L.anchors <= 0 || L.depth <= 0 || L.height <= 0 || L.width <= 0) return MR3D_ERR_INVALID_ARG;
    const long long n = (long long)L.anchors * L.depth * L.height * L.width;
    if (n > 0x7fffffffLL) return MR3D_ERR_UNSUPPORTED;
    t.logits[l] = L.logits; t.valid[l] = (const uint8_t *)L.valid;
    t.sa[l] = L.stride_a; t.sd[l] = L.stride_d; t.sh[l] = L.stride_h; t.sw[l] = L.stride_w;
    t.A[l] = L.anchors; t.D[l] = L.depth; t.H[l] = L.height; t.W[l] = L.width;
    t.n[l] = (int)n;
    t.select[l] = n > nms_pre ? 1 : 0;
    t.k[l] = t.select[l] ? nms_pre : (int)n;
    t.out0[l] = total_out; total_out += t.k[l];
    t.wg0[l] = total_wgs; t.key0[l] = total_keys;
    if (t.select[l]) { total_wgs += (int)((n + kPerWg - 1) / kPerWg); total_keys += (n + 63) / 64 * 64; }
  }
  return MR3D_OK;
}

size_t select_ws_bytes(int num_levels, int total_wgs, long long total_keys) {
  return mr3d_align_up((size_t)num_levels * kHistWords * 4, 256) + mr3d_align_up((size_t)total_wgs * 8 + 8, 256) +
         mr3d_align_up((size_t)total_keys * 4 + 4, 256);
}

}  // namespace

extern "C" size_t mr3d_rpn_select_topk_workspace_bytes(const mr3d_rpn_level *levels_host, int num_levels, int nms_pre) {
  LevelTab t; int wgs, out; long long keys;
  if (fill_table(levels_host, num_levels, nms_pre, t, wgs, keys, out) != MR3D_OK) return 0;
  return select_ws_bytes(num_levels, wgs, keys);
}

extern "C" int mr3d_rpn_select_topk(const mr3d_rpn_level *levels_host, int num_levels, int nms_pre, long long *flat_out,
                                    float *scores_out, int32_t *counts_out, void *workspace, size_t workspace_bytes,
                                    mr3d_stream_t stream_) {
  LevelTab t; int wgs, out; long long nkeys;
  const int st = fill_table(levels_host, num_levels, nms_pre, t, wgs, nkeys, out);
  if (st != MR3D_OK) return st;
  if (!flat_out || !scores_out || !counts_out) return MR3D_ERR_INVALID_ARG;
  if (!workspace || ((uintptr_t)workspace & 255) || workspace_bytes < select_ws_bytes(num_levels, wgs, nkeys)) return MR3D_ERR_WORKSPACE;
  hipStream_t stream = mr3d_stream(stream_);
  char *base = (char *)workspace;
  const size_t hist_bytes = mr3d_align_up((size_t)num_levels * kHistWords * 4, 256);
  const size_t wgc_bytes = mr3d_align_up((size_t)wgs * 8 + 8, 256);
  uint32_t *hist = (uint32_t *)base;
  uint32_t *wgc = (uint32_t *)(base + hist_bytes);
  uint32_t *keys = (uint32_t *)(base + hist_bytes + wgc_bytes);
  int max_small = 0;
  for (int l = 0; l < num_levels; ++l)
    if (!t.select[l] && t.n[l] > max_small) max_small = t.n[l];
  if (max_small > 0)
    rpn_select_all_kernel<<<dim3((max_small + kT - 1) / kT, num_levels), kT, 0, stream>>>(t, flat_out, scores_out, counts_out);
  if (wgs > 0) {
    if (hipMemsetAsync(hist, 0, hist_bytes, stream) != hipSuccess) return MR3D_ERR_LAUNCH;
    rpn_select_hist_kernel<0><<<wgs, kT, 0, stream>>>(t, nms_pre, keys, hist);
    rpn_select_hist_kernel<1><<<wgs, kT, 0, stream>>>(t, nms_pre, keys, hist);
    rpn_select_hist_kernel<2><<<wgs, kT, 0, stream>>>(t, nms_pre, keys, hist);
    rpn_select_count_kernel<<<wgs, kT, 0, stream>>>(t, nms_pre, keys, hist, wgc);
    rpn_select_write_kernel<<<wgs, kT, 0, stream>>>(t, nms_pre, keys, hist, wgc, flat_out, scores_out, counts_out);
  }
  return mr3d_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------- merge
namespace {

constexpr int kMergeThreads = 1024;
constexpr int kMergeE = 16;                      // 16,384 candidates at most
constexpr int kMergeMax = kMergeThreads * kMergeE;
constexpr int kMergeSel = 2 * kMergeThreads;     // the sorted part: max_num <= 2048

struct MergeTab { int num; int row0[kMaxLevels * 2]; };

// Candidates = the first min(num_keep, nms_post) kept rows of every level, position = running index in level order.  Their
// keys (~score bits, position) are held 16 per thread.  Sorting all of them in one workgroup cost 220 us (91 compare-exchange
// rounds over 8,192 keys); only the max_num best are wanted, so: an 8-bit-per-pass radix select of the max_num-th key over
// LDS histograms (4 passes), a compaction of the keys below it (+ the first few ON it, in position order) into 2,048 slots,
// and the register bitonic sort of those.
__global__ __launch_bounds__(kMergeThreads) void rpn_merge_kernel(const float *__restrict__ dets, MergeTab mt,
                                                                 const long long *__restrict__ keep, const int32_t *__restrict__ num_keep,
                                                                 int nms_post, int max_num, float *__restrict__ out,
                                                                 int32_t *__restrict__ count_out) {
  __shared__ uint64_t sel[kMergeSel];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t scan[kMergeThreads];
  __shared__ int s_start[kMaxLevels * 2 + 1];
  __shared__ uint32_t s_bin, s_below, s_slot;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int run = 0;
    for (int l = 0; l < mt.num; ++l) {
      s_start[l] = run;
      int k = num_keep[l];
      k = k < 0 ? 0 : (k > nms_post ? nms_post : k);             // proposals[:nms_post] (:135)
      run += k;
    }
    s_start[mt.num] = run > kMergeMax ? kMergeMax : run;
    s_slot = 0;
  }
  for (int i = tid; i < kMergeSel; i += kMergeThreads) sel[i] = ~0ull;
  __syncthreads();
  const int total = s_start[mt.num];
  auto row_of = [&](int c) {
    int l = 0;
    for (int j = 1; j < mt.num; ++j)
      if (c >= s_start[j]) l = j;
    return (long long)mt.row0[l] + keep[mt.row0[l] + (c - s_start[l])];
  };
  uint32_t k32[kMergeE];                       // ~(order-preserving score bits): smaller = better; 0xffffffff = no candidate
#pragma unroll
  for (int e = 0; e < kMergeE; ++e) {
    const int c = tid * kMergeE + e;
    uint32_t key = 0xffffffffu;
    if (c < total) {
      uint32_t u = __float_as_uint(dets[row_of(c) * 7 + 6]);
      if (u == 0x80000000u) u = 0u;
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      key = ~u;
      if (key == 0xffffffffu) key = 0xfffffffeu;
    }
    k32[e] = key;
  }
  const int take = total < max_num ? total : max_num;
  uint32_t T = 0xffffffffu, eq_take = 0;       // selected: key < T, or key == T and fewer than eq_take equal keys in front
  if (total > max_num) {
    uint32_t prefix = 0, krem = (uint32_t)max_num;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < kMergeE; ++e) {
        const uint32_t key = k32[e];
        if (key != 0xffffffffu && (pass == 0 || (key >> (shift + 8)) == prefix)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid < 64) {                          // one wave: 4 bins per lane, ascending
        uint32_t h[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = hist[tid * 4 + j]; sum += h[j]; }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
          if (tid >= o) incl += v;
        }
        uint32_t run = incl - sum;
        if (run < krem && incl >= krem) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (run + h[j] >= krem) { s_bin = (uint32_t)(tid * 4 + j); s_below = run; break; }
            run += h[j];
          }
        }
      }
      __syncthreads();
      prefix = (prefix << 8) | s_bin;
      krem -= s_below;
      __syncthreads();
    }
    T = prefix; eq_take = krem;
  }
  // rank of this thread's keys ON the pivot among all such keys, in position order
  uint32_t neq = 0;
#pragma unroll
  for (int e = 0; e < kMergeE; ++e) neq += k32[e] == T && T != 0xffffffffu;
  scan[tid] = neq;
  __syncthreads();
  for (int o = 1; o < kMergeThreads; o <<= 1) {
    const uint32_t v = tid >= o ? scan[tid - o] : 0u;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  uint32_t eq_before = scan[tid] - neq;
#pragma unroll
  for (int e = 0; e < kMergeE; ++e) {
    const uint32_t key = k32[e];
    const bool is_eq = key == T && T != 0xffffffffu;
    if (key < T || (is_eq && eq_before < eq_take)) {
      const uint32_t slot = atomicAdd(&s_slot, 1u);
      if (slot < (uint32_t)kMergeSel) sel[slot] = ((uint64_t)key << 32) | (uint32_t)(tid * kMergeE + e);
    }
    eq_before += is_eq;
  }
  __syncthreads();
  // the selected keys (<= max_num of them, unordered) -> (score desc, position asc)
  extern __shared__ __attribute__((aligned(16))) uint64_t sm[];
  uint64_t v[2] = {sel[tid * 2], sel[tid * 2 + 1]};
  __syncthreads();
  int np2 = 2;
  while (np2 < take) np2 <<= 1;
  bitonic_sort_regs<2>(v, tid, np2 > kMergeSel ? kMergeSel : np2, sm);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = tid * 2 + e;
    if (i < take) {
      const long long row = row_of((int)(uint32_t)v[e]);
#pragma unroll
      for (int q = 0; q < 7; ++q) out[(size_t)i * 7 + q] = dets[row * 7 + q];
    }
  }
  for (int i = take + tid; i < max_num; i += kMergeThreads)
#pragma unroll
    for (int q = 0; q < 7; ++q) out[(size_t)i * 7 + q] = 0.f;
  if (tid == 0) *count_out = take;
}

}  // namespace

extern "C" int mr3d_rpn_merge_levels(const float *dets, const int *seg_rows_host, int num_levels, const long long *keep,
                                     const int32_t *num_keep, int nms_post, int max_num, float *proposals_out,
                                     int32_t *count_out, mr3d_stream_t stream_) {
  if (!dets || !seg_rows_host || num_levels <= 0 || num_levels > kMaxLevels * 2 || !keep || !num_keep || nms_post <= 0 ||
      max_num <= 0 || !proposals_out || !count_out)
    return MR3D_ERR_INVALID_ARG;
  MergeTab mt;
  mt.num = num_levels;
  int run = 0;
  long long cand = 0;
  for (int l = 0; l < kMaxLevels * 2; ++l) {
    mt.row0[l] = run;
    if (l < num_levels) {
      if (seg_rows_host[l] < 0) return MR3D_ERR_INVALID_ARG;
      run += seg_rows_host[l];
      cand += seg_rows_host[l] < nms_post ? seg_rows_host[l] : nms_post;
    }
  }
  if (cand > kMergeMax || max_num > kMergeSel) return MR3D_ERR_UNSUPPORTED;
  constexpr int lds = kMergeSel * 8;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(rpn_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  rpn_merge_kernel<<<1, kMergeThreads, lds, mr3d_stream(stream_)>>>(dets, mt, keep, num_keep, nms_post, max_num, proposals_out, count_out);
  return mr3d_launch_status();
}

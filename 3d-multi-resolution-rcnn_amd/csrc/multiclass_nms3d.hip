// multiclass_nms_3d on the device -- replaces the per-class Python loop of
// mmdet/core/post_processing/bbox_nms.py:57-106 (score filter -> nms() -> labels -> cat -> top-`max_num` by score),
// which launches ~10 ATen kernels and synchronises with the host (`cls_inds.any()`, boolean-mask gathers) once per
// foreground class.
//
// Own design: nothing is compacted before the NMS.  Every foreground class becomes one fixed-size segment of ALL n
// rows; rows that fail the score filter keep their slot with score -FLT_MAX and a zero box.  NMS3D processes boxes in
// (score desc, index asc) order and suppression only flows from earlier to later boxes, so the filtered rows -- all
// sorted behind every real one -- can never suppress a real box: the kept REAL rows are exactly those of an NMS over
// the filtered subset, in the same ascending-index order.  One mr3d_nms3d_batched call serves all classes; a final
// single-workgroup kernel drops the filler rows, concatenates the classes, and, when more than max_num survive, ranks
// them by (score desc, position asc) and keeps the first max_num.  The only host read-back is the final count.
#include "mr3d_common.h"
#include <float.h>

namespace {

constexpr int kT = 1024;
constexpr int kMaxSegments = 16;        // mr3d_nms3d_batched's limit
constexpr int kMaxRows = 4096;

__global__ __launch_bounds__(256) void mc_build_kernel(const float *__restrict__ boxes, int box_cols, const float *__restrict__ scores,
                                                       int n, int C, float thr, float *__restrict__ dets) {
  const long long total = (long long)(C - 1) * n;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int s = (int)(t / n), r = (int)(t - (long long)s * n), cls = s + 1;
    const float sc = scores[(size_t)r * C + cls];
    float *d = dets + (size_t)t * 7;
    if (sc > thr) {
      const float *b = boxes + (size_t)r * box_cols + (box_cols == 6 ? 0 : cls * 6);
#pragma unroll
      for (int k = 0; k < 6; ++k) d[k] = b[k];
      d[6] = sc;
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) d[k] = 0.f;
      d[6] = -FLT_MAX;
    }
  }
}

// one workgroup: filter the kept lists, concatenate the classes, optional top-max_num
__global__ __launch_bounds__(kT) void mc_finalize_kernel(const float *__restrict__ dets, const long long *__restrict__ keep,
                                                        const int *__restrict__ num_keep, int n, int C, float thr, int max_num,
                                                        int *__restrict__ cand, float *__restrict__ dets_out,
                                                        long long *__restrict__ labels_out, int *__restrict__ count_out) {
  __shared__ int wsum[kT / 64];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  // ---- pass 1: cand[pos] = flat index (segment * n + row) of every kept real row, classes in order
  for (int s = 0; s < C - 1; ++s) {
    const int k = num_keep[s];
    for (int j0 = 0; j0 < k; j0 += kT) {
      const int j = j0 + tid;
      int flat = -1;
      if (j < k) {
        const int r = (int)keep[(size_t)s * n + j];
        if (dets[((size_t)s * n + r) * 7 + 6] > thr) flat = s * n + r;
      }
      const unsigned long long m = __ballot(flat >= 0);
      const int before = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[wave] = __popcll(m);
      __syncthreads();
      int off = base_s;
      for (int w = 0; w < wave; ++w) off += wsum[w];
      if (flat >= 0) cand[off + before] = flat;
      __syncthreads();
      if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < kT / 64; ++w) tot += wsum[w];
        base_s += tot;
      }
      __syncthreads();
    }
  }
  const int T = base_s;
  // bbox_nms.py:96-101: `if bboxes.shape[0] > max_num: sort by score, keep inds[:max_num]`.  With the signature's
  // default max_num = -1 the test is always true and `inds[:-1]` drops the lowest-scoring row: reproduced.
  const bool ranked = T > max_num;
  int keepN = T;
  if (ranked) keepN = max_num >= 0 ? max_num : (T + max_num > 0 ? T + max_num : 0);
  for (int i = tid; i < T; i += kT) {
    const int flat = cand[i];
    const float sc = dets[(size_t)flat * 7 + 6];
    int dst = i;
    if (ranked) {
      int rank = 0;
      for (int j = 0; j < T; ++j) {
        const float sj = dets[(size_t)cand[j] * 7 + 6];
        rank += (sj > sc) || (sj == sc && j < i);
      }
      dst = rank;
    }
    if (dst < keepN) {
#pragma unroll
      for (int k = 0; k < 7; ++k) dets_out[(size_t)dst * 7 + k] = dets[(size_t)flat * 7 + k];
      labels_out[dst] = flat / n;            // segment index == class - 1 (bbox_nms.py:92-94)
    }
  }
  if (tid == 0) *count_out = keepN;
}

struct McLayout { size_t dets, keep, num, cand, nms, total; size_t nms_bytes; };

bool mc_layout(int n, int C, McLayout &L) {
  const int S = C - 1;
  if (n <= 0 || S <= 0 || S > kMaxSegments || n > kMaxRows) return false;
  int seg[kMaxSegments];
  for (int s = 0; s < S; ++s) seg[s] = n;
  L.nms_bytes = mr3d_nms3d_batched_workspace_bytes(seg, S);
  size_t off = 0;
  L.dets = off; off += mr3d_align_up(sizeof(float) * 7 * (size_t)S * n, 256);
  L.keep = off; off += mr3d_align_up(sizeof(long long) * (size_t)S * n, 256);
  L.num = off;  off += mr3d_align_up(sizeof(int) * S, 256);
  L.cand = off; off += mr3d_align_up(sizeof(int) * (size_t)S * n, 256);
  L.nms = off;  off += mr3d_align_up(L.nms_bytes, 256);
  L.total = off;
  return true;
}

}  // namespace

extern "C" size_t mr3d_multiclass_nms3d_workspace_bytes(int n, int num_classes) {
  McLayout L;
  return mc_layout(n, num_classes, L) ? L.total : 0;
}

extern "C" int mr3d_multiclass_nms3d(const float *multi_bboxes, int box_cols, const float *multi_scores, int n, int num_classes,
                                     float score_thr, float iou_thr, int max_num, float *dets_out, long long *labels_out,
                                     int *count_out, void *workspace, size_t workspace_bytes, mr3d_stream_t stream_) {
  hipStream_t stream = mr3d_stream(stream_);
  if (!count_out || n < 0 || num_classes < 1) return MR3D_ERR_INVALID_ARG;
  if (n == 0 || num_classes == 1)
    return hipMemsetAsync(count_out, 0, sizeof(int), stream) == hipSuccess ? MR3D_OK : MR3D_ERR_LAUNCH;
  if (!multi_bboxes || !multi_scores || !dets_out || !labels_out) return MR3D_ERR_INVALID_ARG;
  if (box_cols != 6 && box_cols != 6 * num_classes) return MR3D_ERR_INVALID_ARG;
  McLayout L;
  if (!mc_layout(n, num_classes, L)) return MR3D_ERR_UNSUPPORTED;      // > 4096 rows or > 16 foreground classes
  if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return MR3D_ERR_WORKSPACE;
  char *base = (char *)workspace;
  float *dets = (float *)(base + L.dets);
  long long *keep = (long long *)(base + L.keep);
  int *num = (int *)(base + L.num);
  int *cand = (int *)(base + L.cand);
  const int S = num_classes - 1;
  const long long total = (long long)S * n;
  mc_build_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(multi_bboxes, box_cols, multi_scores, n, num_classes,
                                                                      score_thr, dets);
  int seg[kMaxSegments];
  for (int s = 0; s < S; ++s) seg[s] = n;
  const int st = mr3d_nms3d_batched(dets, seg, S, 7, iou_thr, (int64_t *)keep, num, base + L.nms, L.nms_bytes, stream_);
  if (st != MR3D_OK) return st;
  mc_finalize_kernel<<<1, kT, 0, stream>>>(dets, keep, num, n, num_classes, score_thr, max_num, cand, dets_out, labels_out,
                                          count_out);
  return mr3d_launch_status();
}

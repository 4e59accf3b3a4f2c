// Detection losses, value and gradient in one pass (mmdet/core/loss/losses.py:17-36,69-74,77-118; callers
// anchor_head_3d.py:127-177, bbox_head_3d.py:86-135, fcn_mask_head_3d.py:108-124).  The reference evaluates each as
// 5-10 elementwise ATen kernels plus a reduction and lets autograd replay them; here one kernel per loss streams its
// inputs once, writes d(loss)/d(prediction) and leaves per-workgroup partial sums that a second, single-workgroup
// kernel adds in a FIXED order in double precision (run-to-run deterministic, no float atomics).
// The normaliser (`avg_factor`) may live on the device: the sample counts it derives from never visit the host.
#include "mr3d_common.h"

namespace {

constexpr int kT = 256;
constexpr int kMaxBlocks = 1024;

struct Norm {
  const float *dev;   // device scalar, or null
  float host;
  __device__ __forceinline__ float get() const { return dev ? *dev : host; }
};

__device__ __forceinline__ void block_partial(double v, double *partials) {
  __shared__ double s[kT / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < kT / 64; ++w) t += s[w];
    partials[blockIdx.x] = t;
  }
}

// loss_out[0] = sum(partials) * scale ; optional second sum (e.g. a count) to aux_out
__global__ __launch_bounds__(kT) void finish_sum(const double *__restrict__ partials, int nblocks, Norm norm, int divide,
                                                 float *__restrict__ loss_out) {
  __shared__ double s[kT];
  double v = 0;
  for (int i = threadIdx.x; i < nblocks; i += kT) v += partials[i];
  s[threadIdx.x] = v;
  __syncthreads();
  for (int o = kT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = (float)(divide ? s[0] / (double)norm.get() : s[0]);
}

// F.binary_cross_entropy_with_logits(pred, onehot(label), weight, 'sum') / avg  (losses.py:28-36, _expand_binary_labels
// :140-147: target[i][c] = (label[i] == c + 1), weight broadcast over the channels)
__global__ __launch_bounds__(kT) void bce_kernel(const float *__restrict__ score, const long long *__restrict__ labels,
                                                 const float *__restrict__ weights, long long total, int C, Norm norm,
                                                 float *__restrict__ grad, double *__restrict__ partials) {
  const float inv = 1.f / norm.get();
  double acc = 0;
  for (long long e = (long long)blockIdx.x * kT + threadIdx.x; e < total; e += (long long)gridDim.x * kT) {
    const long long i = e / C;
    const int c = (int)(e - i * C);
    const float w = weights[i];
    float g = 0.f;
    if (w != 0.f) {
      const float x = score[e];
      const float t = labels[i] == c + 1 ? 1.f : 0.f;
      const float l = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      acc += (double)(l * w);
      g = (1.f / (1.f + expf(-x)) - t) * w * inv;
    }
    if (grad) grad[e] = g;
  }
  block_partial(acc, partials);
}

// smooth-L1 (losses.py:77-99) * weight, summed / avg (:102-118)
__global__ __launch_bounds__(kT) void smoothl1_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                      const float *__restrict__ weights, long long total, float beta, Norm norm,
                                                      float *__restrict__ grad, double *__restrict__ partials) {
  const float inv = 1.f / norm.get();
  double acc = 0;
  for (long long e = (long long)blockIdx.x * kT + threadIdx.x; e < total; e += (long long)gridDim.x * kT) {
    const float w = weights[e];
    float g = 0.f;
    if (w != 0.f) {
      const float d = pred[e] - target[e];
      const float ad = fabsf(d);
      const float l = ad < beta ? 0.5f * ad * ad / beta : ad - 0.5f * beta;
      acc += (double)(l * w);
      g = (ad < beta ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f))) * w * inv;
    }
    if (grad) grad[e] = g;
  }
  block_partial(acc, partials);
}

// number of rows with weight > 0 -> *out = max(count, 1)   (losses.py:18-19)
__global__ __launch_bounds__(kT) void count_positive_weights(const float *__restrict__ w, int n, float *__restrict__ out) {
  __shared__ int s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  int c = 0;
  for (int i = threadIdx.x; i < n; i += kT) c += w[i] > 0.f;
  atomicAdd(&s, c);
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)(s > 1 ? s : 1);
}

// F.cross_entropy(score, label, 'none') * weight, summed / avg (losses.py:17-25); row-wise log-softmax, C small.
// Also counts top-1 hits over the rows with weight > 0 ... no: accuracy() (losses.py:121-137) counts ALL rows it is
// given; rows >= *valid_rows (padding) are excluded from both.
__global__ __launch_bounds__(kT) void softmax_ce_kernel(const float *__restrict__ score, const long long *__restrict__ labels,
                                                        const float *__restrict__ weights, int n, int C,
                                                        const int *__restrict__ valid_rows, Norm norm, float *__restrict__ grad,
                                                        double *__restrict__ partials, double *__restrict__ hit_partials) {
  const float inv = 1.f / norm.get();
  const int nv = valid_rows ? min(*valid_rows, n) : n;
  double acc = 0, hits = 0;
  for (int i = blockIdx.x * kT + threadIdx.x; i < n; i += gridDim.x * kT) {
    const float *x = score + (size_t)i * C;
    const int lab = (int)labels[i];
    float m = x[0];
    int arg = 0;
    for (int c = 1; c < C; ++c)
      if (x[c] > m) { m = x[c]; arg = c; }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - m);
    const float lse = m + logf(se);
    const float w = i < nv ? weights[i] : 0.f;
    acc += (double)((lse - x[lab]) * w);
    if (i < nv && arg == lab) hits += 1.0;
    if (grad)
      for (int c = 0; c < C; ++c) grad[(size_t)i * C + c] = (expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) * w * inv;
  }
  block_partial(acc, partials);
  __syncthreads();
  block_partial(hits, hit_partials);
}

__global__ __launch_bounds__(kT) void finish_accuracy(const double *__restrict__ hit_partials, int nblocks, int n,
                                                      const int *__restrict__ valid_rows, float *__restrict__ acc_out) {
  __shared__ double s[kT];
  double v = 0;
  for (int i = threadIdx.x; i < nblocks; i += kT) v += hit_partials[i];
  s[threadIdx.x] = v;
  __syncthreads();
  for (int o = kT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int nv = valid_rows ? min(*valid_rows, n) : n;
    acc_out[0] = nv > 0 ? (float)(s[0] * 100.0 / nv) : 0.f;
  }
}

// mask_cross_entropy (losses.py:69-74): mean BCE-with-logits over pred[i, label[i], :] for the first *valid_rows rois
__global__ __launch_bounds__(kT) void mask_bce_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                      const long long *__restrict__ labels, int P, int C, long long S,
                                                      const int *__restrict__ valid_rows, float *__restrict__ grad,
                                                      double *__restrict__ partials) {
  const int nv = valid_rows ? min(*valid_rows, P) : P;
  const float inv = nv > 0 ? 1.f / ((float)nv * (float)S) : 0.f;
  const long long total = (long long)P * C * S;
  double acc = 0;
  for (long long e = (long long)blockIdx.x * kT + threadIdx.x; e < total; e += (long long)gridDim.x * kT) {
    const long long sidx = e % S;
    const long long pc = e / S;
    const int c = (int)(pc % C);
    const int i = (int)(pc / C);
    float g = 0.f;
    if (i < nv && c == (int)labels[i]) {
      const float x = pred[e], t = target[(long long)i * S + sidx];
      acc += (double)(fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
      g = (1.f / (1.f + expf(-x)) - t) * inv;
    }
    if (grad) grad[e] = g;
  }
  block_partial(acc, partials);
}

__global__ void mask_norm_kernel(const int *valid_rows, int P, long long S, float *out) {
  const int nv = valid_rows ? min(*valid_rows, P) : P;
  *out = nv > 0 ? (float)nv * (float)S : 1.f;
}

// sum_i max(pos_i, 1) + max(neg_i, 1) over images (anchor_target.py:69-70) or sum_i (pos_i + neg_i) (#sampled rois)
__global__ void avg_factor_kernel(const int *counts, int count_stride, int num_imgs, int mode, float *out) {
  int t = 0;
  for (int i = 0; i < num_imgs; ++i) {
    const int p = counts[i * count_stride], n = counts[i * count_stride + 1];
    t += mode == 0 ? (p > 1 ? p : 1) + (n > 1 ? n : 1) : p + n;
  }
  *out = (float)(t > 1 ? t : 1);
}

inline int grid_for(long long total) {
  long long b = (total + kT - 1) / kT;
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

}  // namespace

extern "C" size_t mr3d_loss_workspace_bytes(void) { return (size_t)(2 * kMaxBlocks + 8) * sizeof(double); }

extern "C" int mr3d_avg_factor_from_counts(const int *counts, int count_stride, int num_imgs, int mode, float *out,
                                           mr3d_stream_t s) {
  if (!counts || !out || num_imgs <= 0 || count_stride < 2 || (mode != 0 && mode != 1)) return MR3D_ERR_INVALID_ARG;
  avg_factor_kernel<<<1, 1, 0, mr3d_stream(s)>>>(counts, count_stride, num_imgs, mode, out);
  return mr3d_launch_status();
}

extern "C" int mr3d_weighted_bce_logits(const float *score, const long long *labels, const float *weights, long long n, int c,
                                        const float *avg_factor_dev, float avg_factor_host, float *loss_out, float *grad,
                                        void *workspace, mr3d_stream_t s) {
  if (!score || !labels || !weights || n <= 0 || c <= 0 || !loss_out || !workspace) return MR3D_ERR_INVALID_ARG;
  if (!avg_factor_dev && !(avg_factor_host > 0.f)) return MR3D_ERR_INVALID_ARG;
  hipStream_t st = mr3d_stream(s);
  double *partials = (double *)workspace;
  const Norm norm{avg_factor_dev, avg_factor_host};
  const int g = grid_for(n * c);
  bce_kernel<<<g, kT, 0, st>>>(score, labels, weights, n * c, c, norm, grad, partials);
  finish_sum<<<1, kT, 0, st>>>(partials, g, norm, 1, loss_out);
  return mr3d_launch_status();
}

extern "C" int mr3d_weighted_smoothl1(const float *pred, const float *target, const float *weights, long long count, float beta,
                                      const float *avg_factor_dev, float avg_factor_host, float *loss_out, float *grad,
                                      void *workspace, mr3d_stream_t s) {
  if (!pred || !target || !weights || count <= 0 || !(beta > 0.f) || !loss_out || !workspace) return MR3D_ERR_INVALID_ARG;
  if (!avg_factor_dev && !(avg_factor_host > 0.f)) return MR3D_ERR_INVALID_ARG;
  hipStream_t st = mr3d_stream(s);
  double *partials = (double *)workspace;
  const Norm norm{avg_factor_dev, avg_factor_host};
  const int g = grid_for(count);
  smoothl1_kernel<<<g, kT, 0, st>>>(pred, target, weights, count, beta, norm, grad, partials);
  finish_sum<<<1, kT, 0, st>>>(partials, g, norm, 1, loss_out);
  return mr3d_launch_status();
}

extern "C" int mr3d_weighted_softmax_ce(const float *score, const long long *labels, const float *weights, int n, int c,
                                        const int *valid_rows_dev, float *loss_out, float *accuracy_out, float *grad,
                                        void *workspace, mr3d_stream_t s) {
  if (!score || !labels || !weights || n <= 0 || c <= 1 || !loss_out || !workspace) return MR3D_ERR_INVALID_ARG;
  hipStream_t st = mr3d_stream(s);
  double *partials = (double *)workspace, *hit_partials = partials + kMaxBlocks;
  float *avg = (float *)(hit_partials + kMaxBlocks);
  count_positive_weights<<<1, kT, 0, st>>>(weights, n, avg);          // padding rows carry weight 0
  const Norm norm{avg, 0.f};
  const int g = grid_for(n);
  softmax_ce_kernel<<<g, kT, 0, st>>>(score, labels, weights, n, c, valid_rows_dev, norm, grad, partials, hit_partials);
  finish_sum<<<1, kT, 0, st>>>(partials, g, norm, 1, loss_out);
  if (accuracy_out) finish_accuracy<<<1, kT, 0, st>>>(hit_partials, g, n, valid_rows_dev, accuracy_out);
  return mr3d_launch_status();
}

extern "C" int mr3d_mask_bce(const float *pred, const float *target, const long long *labels, int num_rois, int c,
                             long long spatial, const int *valid_rows_dev, float *loss_out, float *grad, void *workspace,
                             mr3d_stream_t s) {
  if (!pred || !target || !labels || num_rois <= 0 || c <= 0 || spatial <= 0 || !loss_out || !workspace)
    return MR3D_ERR_INVALID_ARG;
  hipStream_t st = mr3d_stream(s);
  double *partials = (double *)workspace;
  float *normv = (float *)(partials + 2 * kMaxBlocks);
  mask_norm_kernel<<<1, 1, 0, st>>>(valid_rows_dev, num_rois, spatial, normv);
  const int g = grid_for((long long)num_rois * c * spatial);
  mask_bce_kernel<<<g, kT, 0, st>>>(pred, target, labels, num_rois, c, spatial, valid_rows_dev, grad, partials);
  finish_sum<<<1, kT, 0, st>>>(partials, g, Norm{normv, 0.f}, 1, loss_out);
  return mr3d_launch_status();
}

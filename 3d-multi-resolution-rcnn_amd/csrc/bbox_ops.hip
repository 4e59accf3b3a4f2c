// Box arithmetic of the proposal path (HBM-trivial, launch-bound in the reference: ~30 ATen launches per decode):
//   delta2bbox3D / bbox2delta3d   mmdet/core/bbox/transforms.py:105-160, :34-65
//   bbox_overlaps (6-column)      mmdet/core/bbox/geometry.py:49-60
//   grid anchors                  mmdet/core/anchor/anchor_generator_3d.py:55-71
//   fused RPN level decode        mmdet/models/anchor_heads/rpn_head_3d.py:87-132 (gather top-k -> anchors ->
//                                 delta2bbox3D -> [box, score] rows ready for NMS3D), one launch per level.
// fp32 op order follows the reference expressions; compiled with -ffp-contract=off so that x*y+z is not fused
// (torch evaluates these as separate elementwise ops).
#include "mr3d_common.h"

namespace {

constexpr int kT = 256;

struct DecodeCfg {
  float mean[6], stdv[6];
  float max_ratio;
  int has_max, max_h, max_w, max_d;
};

// rois r[6] = x1,y1,x2,y2,z1,z2 ; d[6] = dx,dy,dw,dh,dz,dd (already denormalised)
__device__ __forceinline__ void decode_one(const float *r, const float *dl, const DecodeCfg &c, float *o) {
  float dx = dl[0] * c.stdv[0] + c.mean[0];
  float dy = dl[1] * c.stdv[1] + c.mean[1];
  float dw = dl[2] * c.stdv[2] + c.mean[2];
  float dh = dl[3] * c.stdv[3] + c.mean[3];
  float dz = dl[4] * c.stdv[4] + c.mean[4];
  float dd = dl[5] * c.stdv[5] + c.mean[5];
  // transforms.py:122-128: dw, dh AND both depth deltas are clamped to +-|log(wh_ratio_clip)|
  dw = fminf(fmaxf(dw, -c.max_ratio), c.max_ratio);
  dh = fminf(fmaxf(dh, -c.max_ratio), c.max_ratio);
  dz = fminf(fmaxf(dz, -c.max_ratio), c.max_ratio);
  dd = fminf(fmaxf(dd, -c.max_ratio), c.max_ratio);
  const float px = (r[0] + r[2]) * 0.5f, py = (r[1] + r[3]) * 0.5f, pz = (r[4] + r[5]) * 0.5f;
  const float pw = r[2] - r[0] + 1.0f, ph = r[3] - r[1] + 1.0f, pd = r[5] - r[4] + 1.0f;
  const float gw = pw * expf(dw), gh = ph * expf(dh), gd = pd * expf(dd);
  const float gx = px + pw * dx, gy = py + ph * dy, gz = pz + pd * dz;
  float x1 = gx - gw * 0.5f + 0.5f, y1 = gy - gh * 0.5f + 0.5f, z1 = gz - gd * 0.5f + 0.5f;
  float x2 = gx + gw * 0.5f - 0.5f, y2 = gy + gh * 0.5f - 0.5f, z2 = gz + gd * 0.5f - 0.5f;
  if (c.has_max) {
    x1 = fminf(fmaxf(x1, 0.f), (float)(c.max_w - 1)); x2 = fminf(fmaxf(x2, 0.f), (float)(c.max_w - 1));
    y1 = fminf(fmaxf(y1, 0.f), (float)(c.max_h - 1)); y2 = fminf(fmaxf(y2, 0.f), (float)(c.max_h - 1));
    z1 = fminf(fmaxf(z1, 0.f), (float)(c.max_d - 1)); z2 = fminf(fmaxf(z2, 0.f), (float)(c.max_d - 1));
  }
  o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = z1; o[5] = z2;
}

__global__ __launch_bounds__(kT) void delta2bbox3d_kernel(const float *__restrict__ rois, const float *__restrict__ deltas,
                                                         int n, int k, DecodeCfg cfg, float *__restrict__ out) {
  const int t = blockIdx.x * kT + threadIdx.x;
  if (t >= n * k) return;
  const int i = t / k, c = t % k;
  float r[6], dl[6], o[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) { r[q] = rois[(size_t)i * 6 + q]; dl[q] = deltas[((size_t)i * k + c) * 6 + q]; }
  decode_one(r, dl, cfg, o);
#pragma unroll
  for (int q = 0; q < 6; ++q) out[((size_t)i * k + c) * 6 + q] = o[q];
}

__global__ __launch_bounds__(kT) void bbox2delta3d_kernel(const float *__restrict__ prop, const float *__restrict__ gt, int n,
                                                         DecodeCfg cfg, float *__restrict__ out) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const float *p = prop + (size_t)i * 6, *g = gt + (size_t)i * 6;
  const float px = (p[0] + p[2]) * 0.5f, py = (p[1] + p[3]) * 0.5f, pw = p[2] - p[0] + 1.0f, ph = p[3] - p[1] + 1.0f;
  const float pz = (p[4] + p[5]) * 0.5f, pd = p[5] - p[4] + 1.0f;
  const float gx = (g[0] + g[2]) * 0.5f, gy = (g[1] + g[3]) * 0.5f, gw = g[2] - g[0] + 1.0f, gh = g[3] - g[1] + 1.0f;
  const float gz = (g[4] + g[5]) * 0.5f, gd = g[5] - g[4] + 1.0f;
  float d[6];
  d[0] = (gx - px) / pw; d[1] = (gy - py) / ph;
  d[2] = logf(gw / pw); d[3] = logf(gh / ph);
  d[4] = (gz - pz) / pd; d[5] = logf(gd / pd);
#pragma unroll
  for (int q = 0; q < 6; ++q) out[(size_t)i * 6 + q] = (d[q] - cfg.mean[q]) / cfg.stdv[q];
}

// IoU matrix [m, n]; one thread per pair; b2 rows are read coalesced along n
__global__ __launch_bounds__(kT) void bbox_overlaps3d_kernel(const float *__restrict__ b1, const float *__restrict__ b2, int m,
                                                            int n, float *__restrict__ out) {
  const long long t = (long long)blockIdx.x * kT + threadIdx.x;
  if (t >= (long long)m * n) return;
  const int i = (int)(t / n), j = (int)(t % n);
  const float *a = b1 + (size_t)i * 6, *b = b2 + (size_t)j * 6;
  const float xA = fmaxf(a[0], b[0]), yA = fmaxf(a[1], b[1]), xB = fminf(a[2], b[2]), yB = fminf(a[3], b[3]);
  const float zA = fmaxf(a[4], b[4]), zB = fminf(a[5], b[5]);
  const float inter = fmaxf(xB - xA + 1, 0.f) * fmaxf(yB - yA + 1, 0.f) * fmaxf(zB - zA + 1, 0.f);
  const float areaA = (a[2] - a[0] + 1) * (a[3] - a[1] + 1) * (a[5] - a[4] + 1);
  const float areaB = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) * (b[5] - b[4] + 1);
  out[t] = inter / (areaA + areaB - inter);
}

// anchors in the reference's flat order ((h*W + w)*D + d)*A + a   (np.meshgrid(x, y, z) 'xy' indexing)
__device__ __forceinline__ void anchor_at(const float *__restrict__ base, int A, int fd, int fh, int fw, int stride,
                                          int dstride, long long flat, float *r) {
  const int a = (int)(flat % A);
  long long q = flat / A;
  const int d = (int)(q % fd); q /= fd;
  const int w = (int)(q % fw);
  const int h = (int)(q / fw);
  const float sx = (float)(w * stride), sy = (float)(h * stride), sz = (float)(d * dstride);
  const float *b = base + a * 6;
  r[0] = b[0] + sx; r[1] = b[1] + sy; r[2] = b[2] + sx; r[3] = b[3] + sy; r[4] = b[4] + sz; r[5] = b[5] + sz;
}

__global__ __launch_bounds__(kT) void grid_anchors3d_kernel(const float *__restrict__ base, int A, int fd, int fh, int fw,
                                                           int stride, int dstride, float *__restrict__ out) {
  const long long total = (long long)fd * fh * fw * A;
  const long long t = (long long)blockIdx.x * kT + threadIdx.x;
  if (t >= total) return;
  float r[6];
  anchor_at(base, A, fd, fh, fw, stride, dstride, t, r);
#pragma unroll
  for (int q = 0; q < 6; ++q) out[t * 6 + q] = r[q];
}

// fused RPN level decode: for the k selected flat anchor indices (reference order) build the anchor, gather its
// 6 deltas from the regression map, decode, append the score -> dets[k, 7]
//   reg map element (flat index f -> h,w,d,a ; component q): NCDHW [A*6, D, H, W] -> ((a*6+q)*D + d)*H*W + h*W + w
//                                                            NDHWC [D, H, W, A*6] -> ((d*H + h)*W + w)*A*6 + a*6 + q
__global__ __launch_bounds__(kT) void rpn_decode_kernel(const long long *__restrict__ idx, const float *__restrict__ scores, int k,
                                                       const float *__restrict__ reg, int reg_layout,
                                                       const float *__restrict__ base, int A, int fd, int fh, int fw,
                                                       int stride, int dstride, DecodeCfg cfg, float *__restrict__ dets) {
  const int t = blockIdx.x * kT + threadIdx.x;
  if (t >= k) return;
  const long long f = idx[t];
  float r[6], dl[6], o[6];
  anchor_at(base, A, fd, fh, fw, stride, dstride, f, r);
  const int a = (int)(f % A);
  long long q = f / A;
  const int d = (int)(q % fd); q /= fd;
  const int w = (int)(q % fw);
  const int h = (int)(q / fw);
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const size_t off = reg_layout == MR3D_LAYOUT_NCDHW
                           ? ((size_t)(a * 6 + c) * fd + d) * fh * fw + (size_t)h * fw + w
                           : (((size_t)d * fh + h) * fw + w) * (A * 6) + a * 6 + c;
    dl[c] = reg[off];
  }
  decode_one(r, dl, cfg, o);
#pragma unroll
  for (int c = 0; c < 6; ++c) dets[(size_t)t * 7 + c] = o[c];
  dets[(size_t)t * 7 + 6] = scores[t];
}

DecodeCfg make_cfg(const float *means, const float *stds, const int *max_shape_hwd, double wh_ratio_clip) {
  DecodeCfg c;
  for (int q = 0; q < 6; ++q) { c.mean[q] = means ? means[q] : 0.f; c.stdv[q] = stds ? stds[q] : 1.f; }
  // np.abs(np.log(wh_ratio_clip)) is float64; torch casts the clamp bound to fp32
  const double lr = wh_ratio_clip > 0 ? __builtin_log(wh_ratio_clip) : 0.0;
  c.max_ratio = (float)(lr < 0 ? -lr : lr);
  c.has_max = max_shape_hwd != nullptr;
  c.max_h = max_shape_hwd ? max_shape_hwd[0] : 0;
  c.max_w = max_shape_hwd ? max_shape_hwd[1] : 0;
  c.max_d = max_shape_hwd ? max_shape_hwd[2] : 0;
  return c;
}

}  // namespace

extern "C" int mr3d_delta2bbox3d(const float *rois, const float *deltas, int n, int num_classes, const float *means_host,
                                 const float *stds_host, const int *max_shape_hwd_host, double wh_ratio_clip, float *out,
                                 mr3d_stream_t s) {
  if (n < 0 || num_classes <= 0) return MR3D_ERR_INVALID_ARG;
  if (n == 0) return MR3D_OK;
  if (!rois || !deltas || !out) return MR3D_ERR_INVALID_ARG;
  const DecodeCfg c = make_cfg(means_host, stds_host, max_shape_hwd_host, wh_ratio_clip);
  delta2bbox3d_kernel<<<(n * num_classes + kT - 1) / kT, kT, 0, mr3d_stream(s)>>>(rois, deltas, n, num_classes, c, out);
  return mr3d_launch_status();
}

extern "C" int mr3d_bbox2delta3d(const float *proposals, const float *gt, int n, const float *means_host,
                                 const float *stds_host, float *out, mr3d_stream_t s) {
  if (n < 0) return MR3D_ERR_INVALID_ARG;
  if (n == 0) return MR3D_OK;
  if (!proposals || !gt || !out) return MR3D_ERR_INVALID_ARG;
  const DecodeCfg c = make_cfg(means_host, stds_host, nullptr, 1.0);
  bbox2delta3d_kernel<<<(n + kT - 1) / kT, kT, 0, mr3d_stream(s)>>>(proposals, gt, n, c, out);
  return mr3d_launch_status();
}

extern "C" int mr3d_bbox_overlaps3d(const float *b1, const float *b2, int m, int n, float *out, mr3d_stream_t s) {
  if (m < 0 || n < 0) return MR3D_ERR_INVALID_ARG;
  if (m == 0 || n == 0) return MR3D_OK;
  if (!b1 || !b2 || !out) return MR3D_ERR_INVALID_ARG;
  const long long total = (long long)m * n;
  bbox_overlaps3d_kernel<<<(unsigned)((total + kT - 1) / kT), kT, 0, mr3d_stream(s)>>>(b1, b2, m, n, out);
  return mr3d_launch_status();
}

extern "C" int mr3d_grid_anchors3d(const float *base_anchors, int num_base, int feat_d, int feat_h, int feat_w, int stride,
                                   int depth_stride, float *out, mr3d_stream_t s) {
  if (!base_anchors || !out || num_base <= 0 || feat_d <= 0 || feat_h <= 0 || feat_w <= 0) return MR3D_ERR_INVALID_ARG;
  const long long total = (long long)feat_d * feat_h * feat_w * num_base;
  grid_anchors3d_kernel<<<(unsigned)((total + kT - 1) / kT), kT, 0, mr3d_stream(s)>>>(base_anchors, num_base, feat_d, feat_h,
                                                                                  feat_w, stride, depth_stride, out);
  return mr3d_launch_status();
}

extern "C" int mr3d_rpn_decode_level(const long long *flat_idx, const float *scores, int k, const float *reg_map, int reg_layout,
                                     const float *base_anchors, int num_base, int feat_d, int feat_h, int feat_w, int stride,
                                     int depth_stride, const float *means_host, const float *stds_host,
                                     const int *max_shape_hwd_host, double wh_ratio_clip, float *dets_out, mr3d_stream_t s) {
  if (k < 0) return MR3D_ERR_INVALID_ARG;
  if (k == 0) return MR3D_OK;
  if (!flat_idx || !scores || !reg_map || !base_anchors || !dets_out || num_base <= 0) return MR3D_ERR_INVALID_ARG;
  if (reg_layout != MR3D_LAYOUT_NCDHW && reg_layout != MR3D_LAYOUT_NDHWC) return MR3D_ERR_INVALID_ARG;
  const DecodeCfg c = make_cfg(means_host, stds_host, max_shape_hwd_host, wh_ratio_clip);
  rpn_decode_kernel<<<(k + kT - 1) / kT, kT, 0, mr3d_stream(s)>>>(flat_idx, scores, k, reg_map, reg_layout, base_anchors, num_base,
                                                               feat_d, feat_h, feat_w, stride, depth_stride, c, dets_out);
  return mr3d_launch_status();
}

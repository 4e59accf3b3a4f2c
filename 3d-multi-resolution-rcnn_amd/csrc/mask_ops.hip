// Mask-crop on the GPU:
//   mask_target3d  training targets of the mask head: per positive RoI crop the assigned GT instance volume and
//                  resample it to (md, mh, mw)    (mmdet/core/mask/mask_target.py:17-49, called from
//                  FCNMaskHead3D.get_target, mmdet/models/mask_heads/fcn_mask_head_3d.py:99-106)
//   mask_paste3d   test-time: resample each predicted (md, mh, mw) probability map to its box and paste the
//                  thresholded result into a full-volume uint8 mask  (fcn_mask_head_3d.py:126-191)
// The reference does both on the CPU with skimage.transform.resize in a Python loop (D2H of boxes, per-RoI
// call, H2D of the result).  skimage is neither vendored nor installable here, so the exact resampling is
// "parity unpinned" in the strict sense; both ops follow scikit-image 0.18's published resize() algorithm (restated at the
// kernels below) and are checked against scipy.ndimage, the routines it calls.  Crop / box arithmetic: bbox.astype(int32),
// size max(x2 - x1 + 1, 1); the target crop is clipped to the volume like a numpy slice.
#include "mr3d_common.h"

namespace {

constexpr int kT = 256;

struct Crop { int x1, y1, z1, w, h, d; };

__device__ __forceinline__ Crop crop_of(const float *__restrict__ box, int W, int H, int D) {
  Crop c;
  const int x1 = (int)box[0], y1 = (int)box[1], x2 = (int)box[2], y2 = (int)box[3], z1 = (int)box[4], z2 = (int)box[5];
  int w = max(x2 - x1 + 1, 1), h = max(y2 - y1 + 1, 1), d = max(z2 - z1 + 1, 1);
  // numpy slice semantics: [lo, lo+size) intersected with [0, dim)
  const int xa = max(x1, 0), ya = max(y1, 0), za = max(z1, 0);
  const int xb = min(x1 + w, W), yb = min(y1 + h, H), zb = min(z1 + d, D);
  c.x1 = xa; c.y1 = ya; c.z1 = za; c.w = xb - xa; c.h = yb - ya; c.d = zb - za;
  return c;
}

// Target rule, restated from the published source of what the reference calls (scikit-image 0.18 transform/_warps.py
// resize(), n-dimensional branch, defaults): the uint8 0/1 crop becomes x/255 (img_as_float), is Gaussian-filtered along
// every axis that is being shrunk (anti_aliasing: sigma = (in/out - 1) / 2, scipy radius int(4 sigma + 0.5), mode
// 'mirror'), sampled at c = (o + 0.5) in/out - 0.5 with order-1 interpolation in mode 'mirror', clipped to the input
// range, multiplied by 255, truncated to uint8 and compared with 0 (mask_target.py:41-44).  All weights are positive and
// sum to 1, so the truncated value is 1 only where EVERY crop voxel the two stencils touch is set: an eroded mask.
//   target(o) = AND over axes-product of crop[mirror(i)], i in [floor(c) - r, (frac(c) > 0 ? floor(c) + 1 : floor(c)) + r]
// (In the reference the float64 sums of an all-ones stencil come out as 1/255 or one ulp below it depending on the
// interpolation weights, and one ulp below truncates to 0: its targets are a rounding-dependent SUBSET of this rule --
// tests/test_mask_crop.py reproduces that with scipy.ndimage, the routines scikit-image calls.)
struct AxisStencil { int lo, hi; };

__device__ __forceinline__ AxisStencil stencil_of(int o, int out, int in) {
  const double f = (double)in / (double)out;
  const double c = f * ((double)o + 0.5) - 0.5;      // numpy's two roundings: this file is built with -ffp-contract=off
  const double fl = floor(c);
  const int r = in > out ? (int)(4.0 * ((f - 1.0) / 2.0) + 0.5) : 0;
  AxisStencil a;
  a.lo = (int)fl - r;
  a.hi = (int)fl + (c > fl ? 1 : 0) + r;
  return a;
}

__device__ __forceinline__ int mirror_idx(int i, int n) {          // scipy.ndimage 'mirror': about the centre of the edge voxel
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i = (i < 0 ? -i : i) % p;
  return i >= n ? p - i : i;
}

__global__ __launch_bounds__(kT) void mask_target3d_kernel(const uint8_t *__restrict__ gt, int D, int H, int W,
                                                          const float *__restrict__ boxes, int box_stride,
                                                          const long long *__restrict__ gt_inds, int P, int md, int mh, int mw,
                                                          float *__restrict__ out) {
  const long long total = (long long)P * md * mh * mw;
  for (long long t = (long long)blockIdx.x * kT + threadIdx.x; t < total; t += (long long)gridDim.x * kT) {
    const int ow = (int)(t % mw), oh = (int)((t / mw) % mh), od = (int)((t / ((long long)mw * mh)) % md);
    const int p = (int)(t / ((long long)mw * mh * md));
    const Crop c = crop_of(boxes + (size_t)p * box_stride, W, H, D);
    bool all = false;
    if (c.w > 0 && c.h > 0 && c.d > 0) {
      const uint8_t *g = gt + (size_t)gt_inds[p] * D * H * W;
      const AxisStencil sx = stencil_of(ow, mw, c.w), sy = stencil_of(oh, mh, c.h), sz = stencil_of(od, md, c.d);
      all = true;
      for (int z = sz.lo; z <= sz.hi && all; ++z) {
        const int zz = c.z1 + mirror_idx(z, c.d);
        for (int y = sy.lo; y <= sy.hi && all; ++y) {
          const uint8_t *row = g + ((size_t)zz * H + (c.y1 + mirror_idx(y, c.h))) * W + c.x1;
          for (int x = sx.lo; x <= sx.hi; ++x)
            if (!row[mirror_idx(x, c.w)]) { all = false; break; }
        }
      }
    }
    out[t] = all ? 1.f : 0.f;
  }
}

// Test-time paste (fcn_mask_head_3d.py:163-189): bbox_mask = resize(prob, (d, h, w)) > thr with (d, h, w) the UNCLIPPED box
// size, assigned to im_mask[z1:z1+d, y1:y1+h, x1:x1+w].  resize() = scikit-image 0.18's n-dimensional branch (see above):
// on every axis where the box is SMALLER than the mask map a Gaussian pre-filter (sigma = (in/out - 1) / 2, radius
// int(4 sigma + 0.5), mode 'mirror', axes in order z, y, x, every pass rounded to the map's float32), then order-1
// sampling at c = (o + 0.5) in/out - 0.5 in mode 'mirror', double arithmetic, result rounded to float32, compared with thr.
// One workgroup per detection: the (md, mh, mw) map and its filtered copy live in LDS.  Voxels of a box that leave the
// volume are skipped (the reference's slice assignment raises there).
constexpr int kPasteT = 512;

__device__ __forceinline__ void gauss_axis(const float *__restrict__ src, float *__restrict__ dst, int nd, int nh, int nw, int axis,
                                           double sigma, int tid) {
  const int n = axis == 0 ? nd : (axis == 1 ? nh : nw);
  const int stride = axis == 0 ? nh * nw : (axis == 1 ? nw : 1);
  const int r = (int)(4.0 * sigma + 0.5);
  // scipy.ndimage._gaussian_kernel1d: exp(-0.5 / sigma^2 * x^2) / sum
  double wsum = 0.0;
  for (int k = -r; k <= r; ++k) wsum += exp(-0.5 / (sigma * sigma) * (double)(k * k));
  const int total = nd * nh * nw;
  for (int e = tid; e < total; e += kPasteT) {
    const int x = e % nw, y = (e / nw) % nh, z = e / (nw * nh);
    const int l = axis == 0 ? z : (axis == 1 ? y : x);
    const float *line = src + (e - l * stride);
    // correlate1d's symmetric form: centre tap, then pairs from the farthest to the nearest
    double acc = (double)line[l * stride] * (1.0 / wsum);
    for (int k = r; k >= 1; --k) {
      const double w = exp(-0.5 / (sigma * sigma) * (double)(k * k)) / wsum;
      acc += ((double)line[mirror_idx(l - k, n) * stride] + (double)line[mirror_idx(l + k, n) * stride]) * w;
    }
    dst[e] = (float)acc;
  }
}

__global__ __launch_bounds__(kPasteT) void mask_paste3d_kernel(const float *__restrict__ prob, int md, int mh, int mw,
                                                              const float *__restrict__ boxes, int box_stride, int P, float thr,
                                                              int D, int H, int W, uint8_t *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float pl[];
  const int p = blockIdx.x, tid = threadIdx.x;
  const float *box = boxes + (size_t)p * box_stride;
  const int x1 = (int)box[0], y1 = (int)box[1], x2 = (int)box[2], y2 = (int)box[3], z1 = (int)box[4], z2 = (int)box[5];
  const int bw = max(x2 - x1 + 1, 1), bh = max(y2 - y1 + 1, 1), bd = max(z2 - z1 + 1, 1);
  const int M = md * mh * mw;
  float *cur = pl, *nxt = pl + M;
  for (int e = tid; e < M; e += kPasteT) cur[e] = prob[(size_t)p * M + e];
  __syncthreads();
  const int sizes[3] = {bd, bh, bw}, maps[3] = {md, mh, mw};
  for (int axis = 0; axis < 3; ++axis) {
    const double f = (double)maps[axis] / (double)sizes[axis];
    const double sigma = fmax(0.0, (f - 1.0) / 2.0);
    if (sigma > 1e-15) {
      gauss_axis(cur, nxt, md, mh, mw, axis, sigma, tid);
      __syncthreads();
      float *t = cur; cur = nxt; nxt = t;
    }
  }
  uint8_t *o = out + (size_t)p * D * H * W;
  const double fz = (double)md / (double)bd, fy = (double)mh / (double)bh, fx = (double)mw / (double)bw;
  const long long n = (long long)bw * bh * bd;
  for (long long t = tid; t < n; t += kPasteT) {
    const int x = (int)(t % bw), y = (int)((t / bw) % bh), z = (int)(t / ((long long)bw * bh));
    const int vx = x1 + x, vy = y1 + y, vz = z1 + z;
    if ((unsigned)vx >= (unsigned)W || (unsigned)vy >= (unsigned)H || (unsigned)vz >= (unsigned)D) continue;
    const double cz = fz * ((double)z + 0.5) - 0.5, cy = fy * ((double)y + 0.5) - 0.5, cx = fx * ((double)x + 0.5) - 0.5;
    const double flz = floor(cz), fly = floor(cy), flx = floor(cx);
    const double tz = cz - flz, ty = cy - fly, tx = cx - flx;
    const int z0 = mirror_idx((int)flz, md), z1i = mirror_idx((int)flz + 1, md);
    const int y0 = mirror_idx((int)fly, mh), y1i = mirror_idx((int)fly + 1, mh);
    const int x0 = mirror_idx((int)flx, mw), x1i = mirror_idx((int)flx + 1, mw);
    auto at = [&](int zz, int yy, int xx) { return (double)cur[(zz * mh + yy) * mw + xx]; };
    const double c00 = at(z0, y0, x0) * (1.0 - tx) + at(z0, y0, x1i) * tx, c01 = at(z0, y1i, x0) * (1.0 - tx) + at(z0, y1i, x1i) * tx;
    const double c10 = at(z1i, y0, x0) * (1.0 - tx) + at(z1i, y0, x1i) * tx, c11 = at(z1i, y1i, x0) * (1.0 - tx) + at(z1i, y1i, x1i) * tx;
    const float v = (float)((c00 * (1.0 - ty) + c01 * ty) * (1.0 - tz) + (c10 * (1.0 - ty) + c11 * ty) * tz);
    o[((size_t)vz * H + vy) * W + vx] = v > thr ? 1 : 0;
  }
}

}  // namespace

extern "C" int mr3d_mask_target3d(const uint8_t *gt_masks, int depth, int height, int width, const float *pos_boxes,
                                  int box_stride, const long long *pos_assigned_gt_inds, int num_pos, int mask_d, int mask_h,
                                  int mask_w, float *targets, mr3d_stream_t s) {
  if (num_pos < 0 || mask_d <= 0 || mask_h <= 0 || mask_w <= 0 || box_stride < 6) return MR3D_ERR_INVALID_ARG;
  if (num_pos == 0) return MR3D_OK;
  if (!gt_masks || !pos_boxes || !pos_assigned_gt_inds || !targets) return MR3D_ERR_INVALID_ARG;
  const long long total = (long long)num_pos * mask_d * mask_h * mask_w;
  long long blocks = (total + kT - 1) / kT;
  if (blocks > 8192) blocks = 8192;
  mask_target3d_kernel<<<(unsigned)blocks, kT, 0, mr3d_stream(s)>>>(gt_masks, depth, height, width, pos_boxes, box_stride,
                                                                pos_assigned_gt_inds, num_pos, mask_d, mask_h, mask_w, targets);
  return mr3d_launch_status();
}

extern "C" int mr3d_mask_paste3d(const float *mask_prob, int mask_d, int mask_h, int mask_w, const float *det_boxes,
                                 int box_stride, int num_dets, float threshold, int depth, int height, int width,
                                 uint8_t *out_masks, mr3d_stream_t s) {
  if (num_dets < 0 || box_stride < 6) return MR3D_ERR_INVALID_ARG;
  if (num_dets == 0) return MR3D_OK;
  if (!mask_prob || !det_boxes || !out_masks) return MR3D_ERR_INVALID_ARG;
  const size_t lds = (size_t)2 * mask_d * mask_h * mask_w * 4;
  if (lds > 160 * 1024) return MR3D_ERR_UNSUPPORTED;                 // the shipped 20 x 28 x 28 map: 125 KB
  static size_t granted = 0;
  if (lds > granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(mask_paste3d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    granted = lds;
  }
  mask_paste3d_kernel<<<(unsigned)num_dets, kPasteT, lds, mr3d_stream(s)>>>(mask_prob, mask_d, mask_h, mask_w, det_boxes,
                                                                             box_stride, num_dets, threshold, depth, height, width,
                                                                             out_masks);
  return mr3d_launch_status();
}

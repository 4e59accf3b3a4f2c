/*
 * mr3d_oracle.c -- CPU restatement of the reference's hot-path operators.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or call it, and only
 * as the checker.  The shipped path is the HIP library (include/mr3d.h).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  The reference's CUDA extensions cannot be built
 * in this image (no nvcc, torch-1.0 THC API), so this restatement is pinned by
 *   - tests/golden/ .npz fixtures produced by tests/golden/make_golden.py, which imports
 *     the reference's own Python (coco_utils.nms_3d_python, geometry.bbox_overlaps
 *     incl. its known-answer IoUs 1.0/0.798/0.7899/0.6125, transforms.delta2bbox3D,
 *     AnchorGenerator3D, SingleRoIExtractor.map_roi_levels, ResNet3D, FPN3D), and
 *   - an independent float64 numpy trilinear RoIAlign in tests/ (the reference
 *     ships no RoIAlign3D test or vector: that op is "parity unpinned" by the
 *     reference's own tests).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction so the
 * fp32 op order is the one written here).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* 3D IoU, inclusive "+1" convention, all fp32.                                */
/* mmdet/ops/nms/src/nms_kernel.cu:23-33 (devIoU3d)                            */
/* box layout: [x1, y1, x2, y2, z1, z2, (score)]                               */
/* ------------------------------------------------------------------------- */
float oracle_iou3d(const float *a, const float *b) {
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float front = fmaxf(a[4], b[4]), back = fminf(a[5], b[5]);
  float width = fmaxf(right - left + 1, 0.f);
  float height = fmaxf(bottom - top + 1, 0.f);
  float depth = fmaxf(back - front + 1, 0.f);
  float interS = width * height * depth;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1) * (a[5] - a[4] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) * (b[5] - b[4] + 1);
  return interS / (Sa + Sb - interS);
}

/* ------------------------------------------------------------------------- */
/* NMS3D.  mmdet/ops/nms/src/nms_kernel.cu:196-257 (nms_cuda_3d) +             */
/* :81-129 (nms_kernel_3d).                                                    */
/*  (1) order = argsort(score, descending) (:199-200).  torch.sort leaves the  */
/*      order of equal scores unspecified; this restatement (and the HIP path) */
/*      fix it as "lower original index first" (a stable descending sort).     */
/*  (2) bit (i,j), j>i in sorted order, is set iff IoU > thr (strict, :122).   */
/*  (3) greedy sweep in sorted order: box i is kept iff no kept box before it  */
/*      has its bit set (:237-249).                                            */
/*  (4) result = original indices of kept boxes sorted ascending (:253-256).   */
/* dets: [n,7] row-major fp32 (row stride `stride` floats).  keep: int64[n].   */
/* returns number kept.                                                        */
/* ------------------------------------------------------------------------- */
typedef struct { float s; int32_t i; } oracle_si;

static int oracle_cmp_desc(const void *pa, const void *pb) {
  const oracle_si *a = (const oracle_si *)pa, *b = (const oracle_si *)pb;
  if (a->s > b->s) return -1;
  if (a->s < b->s) return 1;
  return (a->i > b->i) - (a->i < b->i);
}

static int oracle_cmp_i64(const void *pa, const void *pb) {
  int64_t a = *(const int64_t *)pa, b = *(const int64_t *)pb;
  return (a > b) - (a < b);
}

int oracle_nms3d(const float *dets, int n, int stride, float thr, int64_t *keep) {
  if (n <= 0) return 0;
  oracle_si *ord = (oracle_si *)malloc(sizeof(oracle_si) * (size_t)n);
  unsigned char *removed = (unsigned char *)calloc((size_t)n, 1);
  for (int i = 0; i < n; ++i) { ord[i].s = dets[(size_t)i * stride + 6]; ord[i].i = i; }
  qsort(ord, (size_t)n, sizeof(oracle_si), oracle_cmp_desc);
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    const float *bi = dets + (size_t)ord[i].i * stride;
    keep[nk++] = ord[i].i;
    for (int j = i + 1; j < n; ++j) {
      if (removed[j]) continue; /* same result as OR-ing the full row: removed is monotone */
      if (oracle_iou3d(bi, dets + (size_t)ord[j].i * stride) > thr) removed[j] = 1;
    }
  }
  qsort(keep, (size_t)nk, sizeof(int64_t), oracle_cmp_i64);
  free(ord); free(removed);
  return nk;
}

/* The 64x64-block suppression bit-matrix itself (nms_kernel.cu:81-129), for    */
/* tests that want to check the intermediate: boxes must already be sorted.    */
/* mask: uint64[n * col_blocks]; full matrix (incl. below-diagonal blocks, as  */
/* the reference computes them, :86 commented out).                            */
void oracle_nms3d_mask(const float *sorted_boxes, int n, int stride, float thr, uint64_t *mask) {
  const int cb = (n + 63) / 64;
  for (int r = 0; r < n; ++r) {
    for (int c = 0; c < cb; ++c) {
      uint64_t t = 0;
      int cs = (n - c * 64 < 64) ? n - c * 64 : 64;
      int start = (r / 64 == c) ? (r % 64) + 1 : 0;
      for (int i = start; i < cs; ++i)
        if (oracle_iou3d(sorted_boxes + (size_t)r * stride, sorted_boxes + (size_t)(c * 64 + i) * stride) > thr)
          t |= 1ULL << i;
      mask[(size_t)r * cb + c] = t;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Trilinear tap set for one sample point.                                     */
/* mmdet/ops/roi_align/src/roi_align_kernel.cu:64-149 (forward value) and      */
/* :383-442 (gradient weights) share this logic.                               */
/* returns 0 when the point is outside (-1, dim) on any axis (:69-71).         */
/* ------------------------------------------------------------------------- */
typedef struct {
  int xl, xh, yl, yh, zl, zh;
  float w[8]; /* w1..w8 in the reference's order (:134-137) */
} oracle_taps;

static int oracle_trilinear_taps(int depth, int height, int width, float z, float y, float x, oracle_taps *t) {
  if (z < -1.0 || z > depth || y < -1.0 || y > height || x < -1.0 || x > width) return 0;
  if (z <= 0) z = 0;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int z_low = (int)z, y_low = (int)y, x_low = (int)x, z_high, y_high, x_high;
  if (z_low >= depth - 1) { z_high = z_low = depth - 1; z = (float)z_low; } else z_high = z_low + 1;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else x_high = x_low + 1;
  float lz = z - z_low, ly = y - y_low, lx = x - x_low;
  float hz = 1.f - lz, hy = 1.f - ly, hx = 1.f - lx;
  t->xl = x_low; t->xh = x_high; t->yl = y_low; t->yh = y_high; t->zl = z_low; t->zh = z_high;
  t->w[0] = hx * hy * hz; t->w[4] = hx * hy * lz;
  t->w[1] = lx * hy * hz; t->w[5] = lx * hy * lz;
  t->w[2] = hx * ly * hz; t->w[6] = hx * ly * lz;
  t->w[3] = lx * ly * hz; t->w[7] = lx * ly * lz;
  return 1;
}

/* roi geometry shared by forward and backward (:232-259 / :539-575) */
typedef struct {
  int b;
  float sw, sh, sd;          /* roi_start_{w,h,d} */
  float bw, bh, bd;          /* bin sizes */
  int nw, nh, nd;            /* samples per bin per axis */
} oracle_roi_geom;

static oracle_roi_geom oracle_geom(const float *roi, float scale, float scale_d, int sample_num,
                                   int pd, int ph, int pw) {
  oracle_roi_geom g;
  g.b = (int)roi[0];
  g.sw = roi[1] * scale;
  g.sh = roi[2] * scale;
  float ew = (roi[3] + 1) * scale;
  float eh = (roi[4] + 1) * scale;
  g.sd = roi[5] * scale_d;
  float ed = (roi[6] + 1) * scale_d;
  float rw = fmaxf(ew - g.sw, 0.f);
  float rh = fmaxf(eh - g.sh, 0.f);
  float rd = fmaxf(ed - g.sd, 0.f);
  g.bd = rd / pd; g.bh = rh / ph; g.bw = rw / pw;
  g.nd = (sample_num > 0) ? sample_num : (int)ceilf(rd / pd);
  g.nh = (sample_num > 0) ? sample_num : (int)ceilf(rh / ph);
  g.nw = (sample_num > 0) ? sample_num : (int)ceilf(rw / pw);
  return g;
}

/* ------------------------------------------------------------------------- */
/* RoIAlign3D forward.  roi_align_kernel.cu:214-291 (ROIAlignForward3D).       */
/* feat [N,C,D,H,W] contiguous, rois [R,7] = [b,x1,y1,x2,y2,z1,z2],            */
/* out [R,C,PD,PH,PW] (fully overwritten).                                     */
/* ------------------------------------------------------------------------- */
void oracle_roi_align3d_forward(const float *feat, const float *rois, int R, int C, int D, int H, int W,
                                int PD, int PH, int PW, float scale, float scale_d, int sample_num,
                                float *out) {
  const long long total = (long long)R * C * PD * PH * PW;
#pragma omp parallel for schedule(static)
  for (long long index = 0; index < total; ++index) {
    int pw = (int)(index % PW);
    int ph = (int)((index / PW) % PH);
    int pd = (int)((index / PW / PH) % PD);
    int c = (int)((index / PW / PH / PD) % C);
    int n = (int)(index / PW / PH / PD / C);
    oracle_roi_geom g = oracle_geom(rois + (size_t)n * 7, scale, scale_d, sample_num, PD, PH, PW);
    const float *src = feat + ((size_t)g.b * C + c) * D * H * W;
    float acc = 0;
    for (int iz = 0; iz < g.nd; ++iz) {
      const float z = g.sd + pd * g.bd + (float)(iz + .5f) * g.bd / (float)g.nd;
      for (int iy = 0; iy < g.nh; ++iy) {
        const float y = g.sh + ph * g.bh + (float)(iy + .5f) * g.bh / (float)g.nh;
        for (int ix = 0; ix < g.nw; ++ix) {
          const float x = g.sw + pw * g.bw + (float)(ix + .5f) * g.bw / (float)g.nw;
          oracle_taps t;
          float val = 0;
          if (oracle_trilinear_taps(D, H, W, z, y, x, &t)) {
            const float v1 = src[t.xl + W * (t.yl + H * t.zl)];
            const float v2 = src[t.xh + W * (t.yl + H * t.zl)];
            const float v3 = src[t.xl + W * (t.yh + H * t.zl)];
            const float v4 = src[t.xh + W * (t.yh + H * t.zl)];
            const float v5 = src[t.xl + W * (t.yl + H * t.zh)];
            const float v6 = src[t.xh + W * (t.yl + H * t.zh)];
            const float v7 = src[t.xl + W * (t.yh + H * t.zh)];
            const float v8 = src[t.xh + W * (t.yh + H * t.zh)];
            val = (t.w[0] * v1 + t.w[1] * v2 + t.w[2] * v3 + t.w[3] * v4 +
                   t.w[4] * v5 + t.w[5] * v6 + t.w[6] * v7 + t.w[7] * v8);
          }
          acc += val;
        }
      }
    }
    acc /= (g.nd * g.nh * g.nw);
    out[index] = acc;
  }
}

/* ------------------------------------------------------------------------- */
/* RoIAlign3D backward.  roi_align_kernel.cu:519-636 (ROIAlignBackward3D).     */
/* bug_compat != 0 reproduces the reference's top_diff index (:554-555):       */
/*   offset_top = (n*C + c)*PD*PH*PW + pd*PD*PW + ph*PW + pw                   */
/* (depth stride PD*PW instead of PH*PW).  bug_compat == 0 uses the correct    */
/* index.  bottom_grad [N,C,D,H,W] must be pre-zeroed by the caller            */
/* (functions/roi_align_3d.py:84-85); serial accumulation order = index order. */
/* ------------------------------------------------------------------------- */
void oracle_roi_align3d_backward(const float *top, const float *rois, int R, int C, int D, int H, int W,
                                 int PD, int PH, int PW, float scale, float scale_d, int sample_num,
                                 int bug_compat, float *bottom) {
  const long long total = (long long)R * C * PD * PH * PW;
  for (long long index = 0; index < total; ++index) {
    int pw = (int)(index % PW);
    int ph = (int)((index / PW) % PH);
    int pd = (int)((index / PW / PH) % PD);
    int c = (int)((index / PW / PH / PD) % C);
    int n = (int)(index / PW / PH / PD / C);
    oracle_roi_geom g = oracle_geom(rois + (size_t)n * 7, scale, scale_d, sample_num, PD, PH, PW);
    float *dst = bottom + ((size_t)g.b * C + c) * D * H * W;
    long long off = (long long)(n * C + c) * PD * PH * PW +
                    (bug_compat ? pd * PD * PW : pd * PH * PW) + ph * PW + pw;
    const float tg = top[off];
    const float count = (float)(g.nd * g.nh * g.nw);
    for (int iz = 0; iz < g.nd; ++iz) {
      const float z = g.sd + pd * g.bd + (float)(iz + .5f) * g.bd / (float)g.nd;
      for (int iy = 0; iy < g.nh; ++iy) {
        const float y = g.sh + ph * g.bh + (float)(iy + .5f) * g.bh / (float)g.nh;
        for (int ix = 0; ix < g.nw; ++ix) {
          const float x = g.sw + pw * g.bw + (float)(ix + .5f) * g.bw / (float)g.nw;
          oracle_taps t;
          if (!oracle_trilinear_taps(D, H, W, z, y, x, &t)) continue; /* all lows = -1 => no scatter (:600) */
          dst[(t.zl * H + t.yl) * W + t.xl] += tg * t.w[0] / count;
          dst[(t.zl * H + t.yl) * W + t.xh] += tg * t.w[1] / count;
          dst[(t.zl * H + t.yh) * W + t.xl] += tg * t.w[2] / count;
          dst[(t.zl * H + t.yh) * W + t.xh] += tg * t.w[3] / count;
          dst[(t.zh * H + t.yl) * W + t.xl] += tg * t.w[4] / count;
          dst[(t.zh * H + t.yl) * W + t.xh] += tg * t.w[5] / count;
          dst[(t.zh * H + t.yh) * W + t.xl] += tg * t.w[6] / count;
          dst[(t.zh * H + t.yh) * W + t.xh] += tg * t.w[7] / count;
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* delta2bbox3D.  mmdet/core/bbox/transforms.py:105-160.                       */
/* rois [n,6], deltas [n,6] in the order [dx,dy,dw,dh,dz,dd]; means/stds [6];  */
/* max_shape = (H, W, 3, D) or NULL-equivalent (has_max_shape = 0).            */
/* Clamp: dw, dh, dz AND dd to +-|log(wh_ratio_clip)| (:122-128: the z-centre  */
/* delta is clamped like a log-size, reproduced on purpose).                   */
/* ------------------------------------------------------------------------- */
void oracle_delta2bbox3d(const float *rois, const float *deltas, int n, const float *means, const float *stds,
                         int has_max_shape, int max_h, int max_w, int max_d, double wh_ratio_clip, float *out) {
  /* np.abs(np.log(wh_ratio_clip)) is a float64 scalar that torch casts to fp32 for clamp */
  const float max_ratio = (float)fabs(log(wh_ratio_clip));
  for (int i = 0; i < n; ++i) {
    const float *r = rois + (size_t)i * 6, *dl = deltas + (size_t)i * 6;
    float dx = dl[0] * stds[0] + means[0];
    float dy = dl[1] * stds[1] + means[1];
    float dw = dl[2] * stds[2] + means[2];
    float dh = dl[3] * stds[3] + means[3];
    float dz = dl[4] * stds[4] + means[4];
    float dd = dl[5] * stds[5] + means[5];
    dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
    dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
    dz = fminf(fmaxf(dz, -max_ratio), max_ratio);
    dd = fminf(fmaxf(dd, -max_ratio), max_ratio);
    float px = (r[0] + r[2]) * 0.5f, py = (r[1] + r[3]) * 0.5f, pz = (r[4] + r[5]) * 0.5f;
    float pw = r[2] - r[0] + 1.0f, ph = r[3] - r[1] + 1.0f, pdp = r[5] - r[4] + 1.0f;
    float gw = pw * expf(dw), gh = ph * expf(dh), gd = pdp * expf(dd);
    float gx = px + pw * dx, gy = py + ph * dy, gz = pz + pdp * dz;
    float x1 = gx - gw * 0.5f + 0.5f, y1 = gy - gh * 0.5f + 0.5f, z1 = gz - gd * 0.5f + 0.5f;
    float x2 = gx + gw * 0.5f - 0.5f, y2 = gy + gh * 0.5f - 0.5f, z2 = gz + gd * 0.5f - 0.5f;
    if (has_max_shape) {
      x1 = fminf(fmaxf(x1, 0.f), (float)(max_w - 1)); x2 = fminf(fmaxf(x2, 0.f), (float)(max_w - 1));
      y1 = fminf(fmaxf(y1, 0.f), (float)(max_h - 1)); y2 = fminf(fmaxf(y2, 0.f), (float)(max_h - 1));
      z1 = fminf(fmaxf(z1, 0.f), (float)(max_d - 1)); z2 = fminf(fmaxf(z2, 0.f), (float)(max_d - 1));
    }
    float *o = out + (size_t)i * 6;
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = z1; o[5] = z2;
  }
}

/* ------------------------------------------------------------------------- */
/* Direct 3D convolution (cross-correlation), fp32, NCDHW, no groups --        */
/* the arithmetic of torch.nn.Conv3d at the reference's call sites             */
/* (resnet3d.py:126,139,173,251,398; conv_module3d.py:34).  Used on tiny       */
/* shapes only (tests); accumulation in double to serve as a tight reference.  */
/* ------------------------------------------------------------------------- */
void oracle_conv3d(const float *x, const float *w, const float *bias, int N, int Ci, int D, int H, int W,
                   int Co, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                   float *y) {
  const int Do = (D + 2 * pd - kd) / sd + 1, Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (int od = 0; od < Do; ++od)
        for (int oh = 0; oh < Ho; ++oh)
          for (int ow = 0; ow < Wo; ++ow) {
            double acc = bias ? bias[co] : 0.0;
            for (int ci = 0; ci < Ci; ++ci)
              for (int a = 0; a < kd; ++a) {
                int id = od * sd - pd + a; if (id < 0 || id >= D) continue;
                for (int b = 0; b < kh; ++b) {
                  int ih = oh * sh - ph + b; if (ih < 0 || ih >= H) continue;
                  for (int c = 0; c < kw; ++c) {
                    int iw = ow * sw - pw + c; if (iw < 0 || iw >= W) continue;
                    acc += (double)x[(((size_t)n * Ci + ci) * D + id) * H * W + (size_t)ih * W + iw] *
                           (double)w[((((size_t)co * Ci + ci) * kd + a) * kh + b) * kw + c];
                  }
                }
              }
            y[(((size_t)n * Co + co) * Do + od) * Ho * Wo + (size_t)oh * Wo + ow] = (float)acc;
          }
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

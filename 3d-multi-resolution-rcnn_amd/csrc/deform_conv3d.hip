// DeformConv3d: deformable convolution (v1) and its modulated form (v2) in three dimensions.
//
// The reference ships only the 2D operator (mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:83-435,569-766;
// functions/deform_conv.py:21-24 rejects non-4D input) -- the 3D operator is this build's extension of it
// (SURVEY.md section 8 row A11).  The arithmetic template is kept: a tap samples the input at
//     p = out_pos * stride - pad + tap * dilation + offset[tap]        (per deformable group)
// the sample is 0 unless -1 < p < dim on every axis (:228), interpolation uses floor() and masks each
// corner individually (low >= 0, high <= dim-1, :97-108); offsets are stored per tap as consecutive
// (d, h, w) channels (the 2D layout's (h, w) pairs, :221-223, plus a leading depth component).  With
// kd = 1 and zero depth offsets the operator reduces to the reference's 2D one (tests check that slice).
//
// MI355X design: sampled im2col in channels-last form feeds the existing fp32-MFMA GEMM kernels.
//   * columns col[pos][tap][ci] (ci innermost): one (pos, tap) sample is a 256-B coalesced read per corner for
//     64 channels, and the row layout equals the packed weight layout [tap*Cin + ci][Cout], so
//         out  = col  x Wpacked            (mr3d_conv3d_forward, 1x1x1 over Cin' = taps*Cin)
//         dcol = dy   x Wpacked^T          (same kernel)
//         dW   = col^T x dy                (mr3d_conv3d_wgrad)
//   * col2im scatters dcol back with 256-B contiguous fp32 atomics and reduces the offset / mask gradients
//     across the channels of a deformable group with wave shuffles.
#include "mr3d_common.h"

namespace {

constexpr int kT = 256;

struct DcnGeo {
  int N, D, H, W, C;        // input NDHWC
  int Do, Ho, Wo;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int dg;                   // deformable groups
  long long pos0, npos;     // chunk of output positions [pos0, pos0 + npos) (flattened n, od, oh, ow)
};

struct Corners {
  long long off[8];   // element offsets of the 8 corners (channel 0), valid only where w != 0
  float w[8];
  bool any;
  // for gradients w.r.t. the sample coordinate
  float ld, lh, lw;
  bool vd0, vd1, vh0, vh1, vw0, vw1;
};

__device__ __forceinline__ Corners corners_at(const DcnGeo &g, int n, float pz, float py, float px) {
  Corners c;
  c.any = pz > -1.f && py > -1.f && px > -1.f && pz < (float)g.D && py < (float)g.H && px < (float)g.W;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c.off[i] = 0; c.w[i] = 0.f; }
  c.ld = c.lh = c.lw = 0.f;
  c.vd0 = c.vd1 = c.vh0 = c.vh1 = c.vw0 = c.vw1 = false;
  if (!c.any) return c;
  const int z0 = (int)floorf(pz), y0 = (int)floorf(py), x0 = (int)floorf(px);
  const int z1 = z0 + 1, y1 = y0 + 1, x1 = x0 + 1;
  c.ld = pz - (float)z0; c.lh = py - (float)y0; c.lw = px - (float)x0;
  c.vd0 = z0 >= 0; c.vd1 = z1 <= g.D - 1; c.vh0 = y0 >= 0; c.vh1 = y1 <= g.H - 1; c.vw0 = x0 >= 0; c.vw1 = x1 <= g.W - 1;
  const float wd[2] = {1.f - c.ld, c.ld}, wh[2] = {1.f - c.lh, c.lh}, ww[2] = {1.f - c.lw, c.lw};
  const bool vd[2] = {c.vd0, c.vd1}, vh[2] = {c.vh0, c.vh1}, vw[2] = {c.vw0, c.vw1};
  const int zz[2] = {z0, z1}, yy[2] = {y0, y1}, xx[2] = {x0, x1};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = a * 4 + b * 2 + e;
        if (vd[a] && vh[b] && vw[e]) {
          c.w[i] = wd[a] * wh[b] * ww[e];
          c.off[i] = ((((long long)n * g.D + zz[a]) * g.H + yy[b]) * g.W + xx[e]) * g.C;
        }
      }
  return c;
}

__device__ __forceinline__ void decode_pos(const DcnGeo &g, long long pos, int &n, int &od, int &oh, int &ow) {
  ow = (int)(pos % g.Wo); pos /= g.Wo;
  oh = (int)(pos % g.Ho); pos /= g.Ho;
  od = (int)(pos % g.Do);
  n = (int)(pos / g.Do);
}

// one wave per (position, tap, deformable group); lanes stride over the group's channels
__global__ __launch_bounds__(kT) void deform_im2col3d_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                            const float *__restrict__ mask, DcnGeo g, float *__restrict__ col) {
  const int K = g.kd * g.kh * g.kw, cpg = g.C / g.dg;
  const long long items = g.npos * K * g.dg;
  const int lane = threadIdx.x & 63;
  for (long long it = (long long)blockIdx.x * (kT / 64) + (threadIdx.x >> 6); it < items; it += (long long)gridDim.x * (kT / 64)) {
    const int grp = (int)(it % g.dg);
    const int tap = (int)((it / g.dg) % K);
    const long long lp = it / ((long long)g.dg * K);
    const long long pos = g.pos0 + lp;
    int n, od, oh, ow;
    decode_pos(g, pos, n, od, oh, ow);
    const int kw_ = tap % g.kw, kh_ = (tap / g.kw) % g.kh, kd_ = tap / (g.kw * g.kh);
    const float *op = offset + (size_t)pos * (g.dg * K * 3) + ((size_t)grp * K + tap) * 3;
    const float pz = (float)(od * g.sd - g.pd + kd_ * g.dd) + op[0];
    const float py = (float)(oh * g.sh - g.ph + kh_ * g.dh) + op[1];
    const float px = (float)(ow * g.sw - g.pw + kw_ * g.dw) + op[2];
    const Corners c = corners_at(g, n, pz, py, px);
    const float m = mask ? mask[(size_t)pos * (g.dg * K) + (size_t)grp * K + tap] : 1.f;
    float *dst = col + ((size_t)lp * K + tap) * g.C + grp * cpg;
    for (int ch = lane; ch < cpg; ch += 64) {
      float v = 0.f;
      if (c.any) {
        const float *xc = x + grp * cpg + ch;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (c.w[i] != 0.f) v += c.w[i] * xc[c.off[i]];
      }
      dst[ch] = v * m;
    }
  }
}

// backward of the sampling: dx (atomics), doffset, dmask.  dcol is the gradient of the MODULATED columns.
__global__ __launch_bounds__(kT) void deform_col2im3d_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                            const float *__restrict__ mask, const float *__restrict__ dcol,
                                                            DcnGeo g, float *__restrict__ dx, float *__restrict__ doffset,
                                                            float *__restrict__ dmask) {
  const int K = g.kd * g.kh * g.kw, cpg = g.C / g.dg;
  const long long items = g.npos * K * g.dg;
  const int lane = threadIdx.x & 63;
  for (long long it = (long long)blockIdx.x * (kT / 64) + (threadIdx.x >> 6); it < items; it += (long long)gridDim.x * (kT / 64)) {
    const int grp = (int)(it % g.dg);
    const int tap = (int)((it / g.dg) % K);
    const long long lp = it / ((long long)g.dg * K);
    const long long pos = g.pos0 + lp;
    int n, od, oh, ow;
    decode_pos(g, pos, n, od, oh, ow);
    const int kw_ = tap % g.kw, kh_ = (tap / g.kw) % g.kh, kd_ = tap / (g.kw * g.kh);
    const size_t oidx = (size_t)pos * (g.dg * K * 3) + ((size_t)grp * K + tap) * 3;
    const float pz = (float)(od * g.sd - g.pd + kd_ * g.dd) + offset[oidx + 0];
    const float py = (float)(oh * g.sh - g.ph + kh_ * g.dh) + offset[oidx + 1];
    const float px = (float)(ow * g.sw - g.pw + kw_ * g.dw) + offset[oidx + 2];
    const Corners c = corners_at(g, n, pz, py, px);
    const size_t midx = (size_t)pos * (g.dg * K) + (size_t)grp * K + tap;
    const float m = mask ? mask[midx] : 1.f;
    const float *src = dcol + ((size_t)lp * K + tap) * g.C + grp * cpg;
    float gz = 0.f, gy = 0.f, gx = 0.f, gm = 0.f;
    if (c.any) {
      // d(sample)/d(coord): derivative of the (individually masked) trilinear weights
      const float wd[2] = {1.f - c.ld, c.ld}, wh[2] = {1.f - c.lh, c.lh}, ww[2] = {1.f - c.lw, c.lw};
      const float sd_[2] = {-1.f, 1.f};
      for (int ch = lane; ch < cpg; ch += 64) {
        const float gcol = src[ch];
        const float *xc = x + grp * cpg + ch;
        float *dxc = dx ? dx + grp * cpg + ch : nullptr;
        float val = 0.f, dz = 0.f, dy_ = 0.f, dxx = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int i = a * 4 + b * 2 + e;
              if (c.w[i] != 0.f || ((a ? c.vd1 : c.vd0) && (b ? c.vh1 : c.vh0) && (e ? c.vw1 : c.vw0))) {
                const float xv = xc[c.off[i]];
                val += c.w[i] * xv;
                dz += sd_[a] * wh[b] * ww[e] * xv;
                dy_ += wd[a] * sd_[b] * ww[e] * xv;
                dxx += wd[a] * wh[b] * sd_[e] * xv;
                if (dxc && c.w[i] != 0.f) atomicAdd(dxc + c.off[i], c.w[i] * gcol * m);
              }
            }
        gz += gcol * m * dz; gy += gcol * m * dy_; gx += gcol * m * dxx; gm += gcol * val;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      gz += __shfl_xor(gz, o); gy += __shfl_xor(gy, o); gx += __shfl_xor(gx, o); gm += __shfl_xor(gm, o);
    }
    if (lane == 0) {
      if (doffset) { doffset[oidx + 0] = gz; doffset[oidx + 1] = gy; doffset[oidx + 2] = gx; }
      if (dmask) dmask[midx] = gm;
    }
  }
}

int fill_geo(const mr3d_deform_conv3d_desc *d, DcnGeo &g) {
  if (!d || d->n <= 0 || d->c <= 0 || d->dg <= 0 || d->c % d->dg) return MR3D_ERR_INVALID_ARG;
  g.N = d->n; g.D = d->d; g.H = d->h; g.W = d->w; g.C = d->c;
  g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
  g.pd = d->pd; g.ph = d->ph; g.pw = d->pw; g.dd = d->dil_d; g.dh = d->dil_h; g.dw = d->dil_w; g.dg = d->dg;
  g.Do = (d->d + 2 * d->pd - (d->dil_d * (d->kd - 1) + 1)) / d->sd + 1;
  g.Ho = (d->h + 2 * d->ph - (d->dil_h * (d->kh - 1) + 1)) / d->sh + 1;
  g.Wo = (d->w + 2 * d->pw - (d->dil_w * (d->kw - 1) + 1)) / d->sw + 1;
  if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MR3D_ERR_INVALID_ARG;
  const long long total = (long long)g.N * g.Do * g.Ho * g.Wo;
  if (d->pos0 < 0 || d->npos <= 0 || d->pos0 + d->npos > total) return MR3D_ERR_INVALID_ARG;
  g.pos0 = d->pos0; g.npos = d->npos;
  return MR3D_OK;
}

}  // namespace

extern "C" int mr3d_deform_im2col3d(const mr3d_deform_conv3d_desc *d, const float *x, const float *offset, const float *mask,
                                    float *columns, mr3d_stream_t s) {
  DcnGeo g;
  const int st = fill_geo(d, g);
  if (st != MR3D_OK) return st;
  if (!x || !offset || !columns) return MR3D_ERR_INVALID_ARG;
  const long long items = g.npos * g.kd * g.kh * g.kw * g.dg;
  long long blocks = (items + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  deform_im2col3d_kernel<<<(unsigned)blocks, kT, 0, mr3d_stream(s)>>>(x, offset, mask, g, columns);
  return mr3d_launch_status();
}

extern "C" int mr3d_deform_col2im3d(const mr3d_deform_conv3d_desc *d, const float *x, const float *offset, const float *mask,
                                    const float *grad_columns, float *grad_x, float *grad_offset, float *grad_mask,
                                    mr3d_stream_t s) {
  DcnGeo g;
  const int st = fill_geo(d, g);
  if (st != MR3D_OK) return st;
  if (!x || !offset || !grad_columns) return MR3D_ERR_INVALID_ARG;
  const long long items = g.npos * g.kd * g.kh * g.kw * g.dg;
  long long blocks = (items + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  deform_col2im3d_kernel<<<(unsigned)blocks, kT, 0, mr3d_stream(s)>>>(x, offset, mask, grad_columns, g, grad_x, grad_offset, grad_mask);
  return mr3d_launch_status();
}

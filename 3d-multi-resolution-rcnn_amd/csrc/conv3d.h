// Argument blocks shared by the conv3d kernels (conv3d_fwd.hip, conv3d_wgrad.hip) and the
// plan executor.  All activations are NDHWC fp32 (channels-last); weights are packed
// [tap][Cin][Cout] (tap = (kd*KH + kh)*KW + kw, Cout innermost).
#pragma once
#include "mr3d_common.h"

struct Conv3dFwdArgs {
  const float *x;       // [N, D, H, W, Cin]
  const float *w;       // [KD*KH*KW][Cin][Cout]
  float *y;             // [N, Do, Ho, Wo, Cout]
  const float *shift;   // [Cout] added to every output position (bias / folded BN shift), may be null
  const float *add1;    // same shape as y, added before the activation, may be null
  const float *add2;    // idem
  const float *up;      // [N, Du, Hu, Wu, Cout]: nearest-upsampled to y's grid and added, may be null
  const float *mask;    // same shape as y: y = mask > 0 ? y : 0 (ReLU backward), may be null
  int N, D, H, W, Cin;
  int Do, Ho, Wo, Cout;
  int pd, ph, pw;
  int Du, Hu, Wu;
  int relu;
  // up_even: `up` is the COARSE result of a stride-2 adjoint (a zero-stuffed map that is never materialised): it is added at
  // the even positions (2u) only, nothing elsewhere.  0: nearest-neighbour replication (the FPN top-down add).
  int up_even = 0;
  // y is a [N, yD, yH, yW, Cout] buffer; output (od,oh,ow) lands at (od*ysd, oh*ysh, ow*ysw): lets a
  // coarse-grid result be scattered onto the even positions of a pre-zeroed fine grid (stride-2 dgrad)
  int ysd, ysh, ysw, yD, yH, yW;
  // caller-lent scratch for instances that transform the weights (two-dimensional Winograd); ws_bytes = -1: variant query
  // that assumes it
  float *ws = nullptr;
  long long ws_bytes = 0;
  // streaming pointwise PAIR (pointwise.hip): a second 1x1x1 product on the activated tile while it still sits in LDS,
  // y2 = act2(y * w2 + shift2); w2 packed [Cout][C2]
  const float *w2 = nullptr, *shift2 = nullptr, *mask2 = nullptr;
  float *y2 = nullptr;
  int relu2 = 0;
  // optional liveness bitmap of x (mr3d_voxel_liveness: one bit per position): a 3x3x3 / stride-1 tile whose halo holds no live
  // voxel is its epilogue only (conv3d_wino2d_kernel<true>)
  const unsigned char *x_live = nullptr;
};

struct Conv3dWgradArgs {
  const float *x;       // [N, D, H, W, Cin]   forward input
  const float *dy;      // [N, Do, Ho, Wo, Cout] upstream gradient (already ReLU-masked)
  float *dw;            // [KD*KH*KW][Cin][Cout], PRE-ZEROED; partial sums are added atomically
  float *dbias;         // [Cout] += sum over positions of dy (bias / BN-beta gradient), PRE-ZEROED, may be null
  int N, D, H, W, Cin;
  int Do, Ho, Wo, Cout;
  int pd, ph, pw;
  // optional two-stage reduction: every workgroup stores its partial sums to its own block of `part` with plain
  // stores and a second kernel adds the blocks into dw (null or too small: fp32 atomics straight into dw)
  float *part;
  long long part_bytes;
  int sparse_dy;        // MR3D_WGRAD_SPARSE_DY: test the dy tiles and skip the zero ones
};

// adds `nparts` partial blocks per (chunk y, Cout tile z) into dw: block element e = (g*R + row)*NBc + col goes to
// dw[(g*Cin + y*cstep + row)*Cout + z*nstep + col]
int mr3d_reduce_partials(const float *part, int nparts, int G, int R, int NBc, int chunks, int ntiles, int cstep, int nstep,
                         int Cin, int Cout, float *dw, hipStream_t stream);

using f32x2 = __attribute__((ext_vector_type(2))) float;      // a register pair: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 operands
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// two-dimensional Winograd form of the 3x3x3 / stride-1 convolution (conv3d_wino2d.hip): F(4,3) along W x F(2,3) along H
bool mr3d_conv3d_wino2d_ok(const Conv3dFwdArgs &a);
int mr3d_conv3d_wino2d(const Conv3dFwdArgs &a, hipStream_t stream);
bool mr3d_conv3d_wino2d_wide(const Conv3dFwdArgs &a);            // the launch takes the wide form (4 x 8 x 8 tiles, persistent workgroups)
bool mr3d_conv3d_wino2d_rows_ok(const Conv3dFwdArgs &a);          // Cout 16 / 32
int mr3d_conv3d_wino2d_rows(const Conv3dFwdArgs &a, hipStream_t stream);
// plan pre-pass: the weight transforms of n convolutions in one launch into the caller's workspace (see conv3d_wino2d.hip)
int mr3d_conv3d_wino2d_prepare(const float *const *w, const int *cin, const int *cout, int n, float *ws, long long ws_bytes,
                               hipStream_t stream);
void mr3d_conv3d_wino2d_prepare_end();

// two-dimensional Winograd form of the 3x3x3 / stride-1 weight gradient (conv3d_wgrad_wino2d.hip): F(3,2) along H x F(3,4) along W
bool mr3d_conv3d_wgrad_wino2d_ok(const Conv3dWgradArgs &a);
int mr3d_conv3d_wgrad_wino2d(const Conv3dWgradArgs &a, hipStream_t stream);

// streaming pointwise instances (pointwise.hip); MR3D_ERR_UNSUPPORTED = use the generic kernels
int mr3d_pointwise_fwd_try(const Conv3dFwdArgs &a, hipStream_t stream, bool probe_only = false);
// the PAIR form (a.w2 / a.y2 set): 16 -> 64 -> 16 on whole 32-row tiles; MR3D_ERR_UNSUPPORTED otherwise
int mr3d_pointwise_pair_stream_try(const Conv3dFwdArgs &a, int c2, hipStream_t stream, bool probe_only = false);
int mr3d_pointwise_wgrad_try(const Conv3dWgradArgs &a, hipStream_t stream);

// streaming bf16 pointwise instances (pointwise_bf16.hip); MR3D_ERR_UNSUPPORTED = use the generic bf16 kernel
int mr3d_pointwise_bf16_try(const void *x, const void *w, void *y, const float *shift, const void *add1, const void *add2,
                            const void *up, const void *mask, long long rows, int cin, int cout, int cout_pad, int D, int H, int W,
                            int Du, int Hu, int Wu, int relu, hipStream_t stream);

// the stem's weight gradient from the pooled gradient as a gather on the vector ALUs (stem_wgrad_gather.hip)
bool mr3d_stem_wgrad_gather_ok(int d, int h, int w);
int mr3d_stem_wgrad_gather(const float *x, const void *g, const uint32_t *idx, const void *pooled, int bf16, float *dw, float *dbias,
                           int n, int d, int h, int w, float *workspace, long long workspace_bytes, hipStream_t stream);

/*
 * mr3d.h -- C ABI of libmr3d_hip.so: the MI355X (gfx950) implementation of the
 * reference's detection hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only
 *     enqueues work on that stream: no allocation, no host synchronisation, graph-capturable;
 *   - the return value is an mr3d_status; no C++ exception crosses this boundary;
 *   - outputs that the reference's Python layer pre-zeroes (RoIAlign3D backward's bottom_grad,
 *     mmdet/ops/roi_align/functions/roi_align_3d.py:84-85) must be pre-zeroed by the caller here too.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference
 * checkout).  INTEGRATION.md shows the reference-side binding for each.
 */
#ifndef MR3D_H_
#define MR3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mr3d_stream_t;

typedef enum {
  MR3D_OK = 0,
  MR3D_ERR_INVALID_ARG = 1,   /* bad shape / null pointer / unsupported size        */
  MR3D_ERR_WORKSPACE = 2,     /* workspace missing or too small                     */
  MR3D_ERR_LAUNCH = 3,        /* hipGetLastError() != hipSuccess after the launch   */
  MR3D_ERR_UNSUPPORTED = 4    /* valid request that this build does not implement   */
} mr3d_status;

/* memory layout of a 5-D feature tensor */
typedef enum {
  MR3D_LAYOUT_NCDHW = 0,      /* the reference's contiguous layout                                  */
  MR3D_LAYOUT_NDHWC = 1       /* channels-last (torch.channels_last_3d strides): C innermost        */
} mr3d_layout;

const char *mr3d_version(void);
const char *mr3d_status_string(int status);
/* 1 when the library was built for gfx950 and a gfx950 device is current, else 0 (host-only query) */
int mr3d_device_ok(void);

/* ------------------------------------------------------------------------------------------------
 * NMS3D -- replaces nms_cuda.nms_3d(dets, threshold) -> LongTensor
 *   mmdet/ops/nms/src/nms_cuda.cpp:16-21, mmdet/ops/nms/src/nms_kernel.cu:23-33,81-129,196-257
 * dets: [n, 7] fp32 rows [x1,y1,x2,y2,z1,z2,score], row stride `row_stride` floats (>= 7; the
 *       reference re-gathers a possibly non-contiguous input, nms_kernel.cu:201).
 * keep_out: int64[n]; on return the first *num_keep_out entries hold the ORIGINAL indices of the
 *       kept boxes in ascending order (nms_kernel.cu:253-256).  num_keep_out: device int32[1].
 * Everything (score sort, 64x64 IoU bit-matrix, greedy sweep, index compaction) runs on the
 * device: the reference's 512 KB D2H mask copy + host loop (nms_kernel.cu:225-249) is gone.
 * Equal scores are ordered by ascending original index (torch.sort leaves it unspecified).
 * workspace: mr3d_nms3d_workspace_bytes(n) bytes, 256-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
size_t mr3d_nms3d_workspace_bytes(int n);
int mr3d_nms3d(const float *dets, int n, int row_stride, float iou_threshold,
               int64_t *keep_out, int32_t *num_keep_out,
               void *workspace, size_t workspace_bytes, mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RoIAlign3D forward -- replaces roi_align_cuda.forward3d(features, rois, pooled_depth,
 *   pooled_height, pooled_width, spatial_scale, spatial_scale_depth, sample_num, output)
 *   mmdet/ops/roi_align/src/roi_align_cuda.cpp:68-94, roi_align_kernel.cu:64-149,214-291,316-337
 * features: [N,C,D,H,W] fp32 in `layout`; rois: [num_rois,7] fp32 [batch,x1,y1,x2,y2,z1,z2];
 * output: [num_rois,C,PD,PH,PW] fp32 contiguous (fully overwritten).
 * ---------------------------------------------------------------------------------------------- */
int mr3d_roi_align3d_forward(const float *features, const float *rois, int num_rois,
                             int batch, int channels, int depth, int height, int width,
                             int pooled_depth, int pooled_height, int pooled_width,
                             float spatial_scale, float spatial_scale_depth, int sample_num,
                             int layout, float *output, mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RoIAlign3D backward -- replaces roi_align_cuda.backward3d(top_grad, rois, pd, ph, pw, scale,
 *   scale_depth, sample_num, bottom_grad)
 *   mmdet/ops/roi_align/src/roi_align_cuda.cpp:123-150, roi_align_kernel.cu:383-442,519-636,666-692
 * top_grad: [num_rois,C,PD,PH,PW] fp32 contiguous; bottom_grad: [N,C,D,H,W] fp32 in `layout`,
 * PRE-ZEROED by the caller; gradients are accumulated with fp32 atomics (order-nondeterministic
 * in the last bits, like the reference's atomicAdd, roi_align_kernel.cu:611-619).
 * flags: MR3D_ROIALIGN_BUG_COMPAT reproduces the reference's top_grad index
 *   (pd*PD*PW instead of pd*PH*PW, roi_align_kernel.cu:554-555); 0 = mathematically correct.
 * ---------------------------------------------------------------------------------------------- */
#define MR3D_ROIALIGN_BUG_COMPAT 1
int mr3d_roi_align3d_backward(const float *top_grad, const float *rois, int num_rois,
                              int batch, int channels, int depth, int height, int width,
                              int pooled_depth, int pooled_height, int pooled_width,
                              float spatial_scale, float spatial_scale_depth, int sample_num,
                              int layout, int flags, float *bottom_grad, mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * conv3d forward (+ fused epilogue) -- replaces the torch.nn.Conv3d / cuDNN calls on the reference's
 * backbone-neck path and the elementwise ops that follow them:
 *   mmdet/models/backbones/resnet3d.py:126,139,173,251,398 (Conv3d), :198-226 (norm, ReLU, residual add);
 *   mmdet/models/utils/conv_module3d.py:34; mmdet/models/necks/fpn3d.py:99-108 (lateral + nearest-upsample add).
 * Layout: activations NDHWC fp32 (channels_last_3d), weights packed [kd*kh*kw][Cin][Cout].
 * y = act( conv(x, w) + shift[co] + add1 + add2 + nearest_up(up) ), act = ReLU if relu; then, if mask,
 * y = mask > 0 ? y : 0  (the ReLU backward of a data-gradient call: dgrad = conv with flipped,
 * transposed weights).  Output size: floor((in + 2*pad - k)/stride) + 1 per axis.
 * Supported here: cubic kernels 1 / 3 (stride 1 or 2, Cin % 16 == 0) and 7 (stride (1,2,2), Cin == 4,
 * i.e. the 3-channel stem zero-padded to 4); Cout % 16 == 0.  Anything else: MR3D_ERR_UNSUPPORTED.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float *x; const float *w; float *y;
  const float *shift; const float *add1; const float *add2; const float *up; const float *mask;
  int n, d, h, w_in, cin, cout;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int up_d, up_h, up_w;
  int relu;
} mr3d_conv3d_desc;
int mr3d_conv3d_forward(const mr3d_conv3d_desc *desc, mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * conv3d weight gradient -- the wgrad half of autograd's backward of the torch.nn.Conv3d calls above
 * (reference: loss.backward() in mmdet/core/utils/dist_utils.py:52-53 reaching the Conv3d modules).
 *   dw[tap][ci][co] += sum_{n,p} x[n, p*stride + tap - pad][ci] * dy[n, p][co]
 * x, dy NDHWC fp32; dw packed [kd*kh*kw][Cin][Cout], PRE-ZEROED by the caller (partial sums of the
 * workgroups are added with fp32 atomics: last-bit nondeterministic like any atomic reduction).
 * Same kernel-size / stride / channel support as mr3d_conv3d_forward.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float *x; const float *dy; float *dw;
  int n, d, h, w_in, cin, cout;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
} mr3d_conv3d_wgrad_desc;
int mr3d_conv3d_wgrad(const mr3d_conv3d_wgrad_desc *desc, mr3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MR3D_H_ */

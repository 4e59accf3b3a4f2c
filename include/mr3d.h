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
/* Measurement knobs (A/B runs of kernel variants from the benchmark tools; 0 = the shipped default everywhere).  Not part of the
 * reference's interface: the drop-in entry points never need them.  Keys: */
enum {
  MR3D_TUNE_RESERVED_0 = 0,
  MR3D_TUNE_PLAN_SKIP = 1,        /* sensitivity runs ONLY (results are wrong): bit mask of plan records mr3d_plan_run leaves out --
                                     1: CONV_WGRAD 3x3x3 16->16, 2: CONV_WGRAD 3x3x3 64->64, 4: every CONV_WGRAD, 8: MAXPOOL_FWD,
                                     16: FINISH_GRADS, 32: STEM_WGRAD, 64: CONV_WGRAD 1x1x1, 128: CONV_FWD 3x3x3 with cin >= 64 on
                                     maps below 32768 positions (layer3 / layer4 convs and data gradients), 256: STEM_FWD,
                                     512: ZERO_STUFF2 / UPSAMPLE_ADJ / MAXPOOL_BWD */
  MR3D_TUNE_FORK_SYSTEM_FENCE = 2, /* 1 (set BEFORE the first mr3d_plan_run of the process): the plan executor's fork / join events keep
                                     their system-scope fence (the round-3 behaviour; default 0 = hipEventDisableSystemFence) */
  MR3D_TUNE_STEM_WGRAD_DENSE = 3, /* 1: mr3d_stem_conv_wgrad_pooled runs the dense MFMA kernel (the pooled gradient scattered into its dy
                                     tiles: the round-3/4 form) instead of the gather over the pooled elements (stem_wgrad_gather.hip) */
  MR3D_TUNE_STEM_GATHER_SPARE = 4, /* workgroup slots the gather form of the stem weight gradient leaves free (two 1024-thread workgroups per CU
                                     would occupy every wave slot of the chip: whatever is queued beside it -- gradient finishing, the
                                     collective of the overlapped exchange -- waits for its end).  0 = the library's default (none: see
                                     stem_wgrad_gather.hip), n > 0 = n slots */
  MR3D_TUNE_WINO2D_FORM = 5,      /* 64-wide F(4,3) x F(2,3) convolution (conv3d_wino2d.hip): 0 = the library's choice per map, 1 = always the
                                     2 x 8 x 8-tile kernel of round 2, 2 = always the wide kernel (4 x 8 x 8 tiles, 512 threads) */
  MR3D_TUNE_SIDE_PRIORITY = 6,    /* priority class of the plan executor's side queues, read at the first plan run of the process: 0 = high
                                     (default: a hardware queue of their own whatever other libraries hold), 1 = middle, 2 = low */
  MR3D_TUNE_ROIALIGN_FORM = 7,    /* channels-last fp32 RoIAlign3D (roi_align3d.hip): 0 = the library's choice per pooled shape (round 6: forward =
                                     the plane-folded kernel for (.,7,7), the row-streaming MFMA kernel for (.,14,14); backward = the plane-owner
                                     kernel), 1 = the round-5 kernels (forward box / wave, backward voxel-owner / wave), 2 = row streaming for both
                                     forward shapes, 8 = plane-folded for both, 4..6 = other (pooled-depth chunk, ring) instances of the
                                     row-streaming kernel, see try_fwd_rows */
  MR3D_TUNE_LATE_GRAD_FINISH = 8, /* read when an ExtractFeat3D engine is built: 1 = ONE gradient finishing at the end of the backward plan (rounds
                                     1-4) instead of the early bucket in the middle of it (the exchange then overlaps only the stem weight gradient) */
  MR3D_TUNE_FPN_AFTER_LATERALS = 9, /* read when a plan is built: 1 = the P3..P5 output convolutions are queued behind ALL lateral convolutions
                                     (rounds 3-4) instead of right behind their own lateral */
  MR3D_TUNE_BF16_ONE_SIDE_QUEUE = 10, /* plan executor: 1 = the bf16 weight-gradient records stay on ONE side queue (rounds 4-5: the 5.5 ms bf16 step
                                     had slack there); 0 = they alternate between the two side queues like the fp32 ones */
  MR3D_TUNE_NUM_KEYS = 16
};
void mr3d_tuning_set(int key, int value);
int mr3d_tuning_get(int key);

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

/* Batched form: `num_segments` (<= 16) independent box sets -- the 5 pyramid levels of RPNHead3D.get_bboxes_single
 * (rpn_head_3d.py:81-135) are 5 separate nms() calls in the reference -- stored back to back in `dets`
 * ([sum n_s, row_stride]); seg_rows_host[s] = n_s (HOST array, each <= 4096).  keep_out receives, at the same row
 * offset as each segment's boxes, that segment's kept indices RELATIVE to the segment (ascending), num_keep_out[s] the
 * counts.  Same semantics per segment as mr3d_nms3d; one sort / bit-matrix / sweep launch for all of them. */
size_t mr3d_nms3d_batched_workspace_bytes(const int *seg_rows_host, int num_segments);
int mr3d_nms3d_batched(const float *dets, const int *seg_rows_host, int num_segments, int row_stride, float iou_threshold,
                       int64_t *keep_out, int32_t *num_keep_out, void *workspace, size_t workspace_bytes,
                       mr3d_stream_t stream);
/* As mr3d_nms3d_batched, with the number of REAL rows of every set read from device memory: set s owns seg_rows_host[s] rows of
 * `dets` (its capacity; workspace and keep_out are sized by it) of which only the first seg_rows_dev[s] take part (NULL: all).
 * Lets a producer kernel decide how many boxes a set has (a top-k over fewer valid anchors than k) without a host round
 * trip; num_keep_out stays on the device for the consumer kernel. */
int mr3d_nms3d_batched_dev(const float *dets, const int *seg_rows_host, const int32_t *seg_rows_dev, int num_segments, int row_stride,
                           float iou_threshold, int64_t *keep_out, int32_t *num_keep_out, void *workspace, size_t workspace_bytes,
                           mr3d_stream_t stream);

/* multiclass_nms_3d (mmdet/core/post_processing/bbox_nms.py:57-106) as ONE device op: per foreground class c = 1..C-1
 * keep rows with multi_scores[r][c] > score_thr, NMS3D them (boxes multi_bboxes[r][6c..6c+5], or [0..5] when
 * box_cols == 6), label c-1, concatenate the classes (each in ascending row order, as nms() returns them) and, if more
 * than max_num survive, keep the max_num best by score (ties: earlier position first; torch.sort leaves them
 * unspecified).  max_num < 0 reproduces the reference's `inds[:max_num]` slice on an always-true test (:96-98).
 * multi_bboxes [n, box_cols], multi_scores [n, num_classes] fp32 contiguous; dets_out [(C-1)*n, 7] (or max_num rows if
 * smaller and >= 0), labels_out int64 same rows; count_out device int32[1].  n <= 4096, C-1 <= 16.
 * workspace: mr3d_multiclass_nms3d_workspace_bytes(n, num_classes) bytes, 256-byte aligned. */
size_t mr3d_multiclass_nms3d_workspace_bytes(int n, int num_classes);
int mr3d_multiclass_nms3d(const float *multi_bboxes, int box_cols, const float *multi_scores, int n, int num_classes,
                          float score_thr, float iou_thr, int max_num, float *dets_out, long long *labels_out,
                          int *count_out, void *workspace, size_t workspace_bytes, mr3d_stream_t stream);

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

/* Typed forms: the reference dispatches float, double and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * roi_align_kernel.cu:323,674); features / top_grad, rois and output / bottom_grad share the element type `dtype`.
 * MR3D_DTYPE_F32 forwards to the calls above; MR3D_DTYPE_F64 computes in double (double atomics); MR3D_DTYPE_F16 computes in
 * fp32 and rounds once per output element / per accumulated contribution (a compare-and-swap on the 32-bit word holding the
 * half, like the reference's gpuAtomicAdd(Half*); the reference rounds after every half operation).  Both take the
 * one-lane-per-output-element kernels (any layout). */
#define MR3D_DTYPE_F32 0
#define MR3D_DTYPE_F16 2
#define MR3D_DTYPE_F64 3
int mr3d_roi_align3d_forward_dtype(const void *features, const void *rois, int num_rois, int batch, int channels, int depth,
                                   int height, int width, int pooled_depth, int pooled_height, int pooled_width,
                                   float spatial_scale, float spatial_scale_depth, int sample_num, int layout, int dtype,
                                   void *output, mr3d_stream_t stream);
int mr3d_roi_align3d_backward_dtype(const void *top_grad, const void *rois, int num_rois, int batch, int channels, int depth,
                                    int height, int width, int pooled_depth, int pooled_height, int pooled_width,
                                    float spatial_scale, float spatial_scale_depth, int sample_num, int layout, int flags,
                                    int dtype, void *bottom_grad, mr3d_stream_t stream);

/* Multi-level form = the whole SingleRoIExtractor.forward / backward (mmdet/models/roi_extractors/single_level.py:58-104)
 * in ONE launch without a host round trip: every RoI reads (scatters into) the pyramid level its size maps to --
 * scale = sqrt(w*h*d) (the square root of the VOLUME, :73-76), level = clamp(floor(log2(scale / finest_scale + 1e-6)))
 * -- with that level's spatial scales.  levels_host: HOST array of num_levels (<= 4) records; features / grad are
 * NDHWC [batch, d, h, w, channels] device buffers (grad PRE-ZEROED by the caller, accumulated atomically);
 * sample_num > 0.  Other arguments as in the single-level calls. */
typedef struct {
  const float *features;     /* forward */
  float *grad;               /* backward */
  int depth, height, width;
  float spatial_scale, spatial_scale_depth;
} mr3d_roi_level;
int mr3d_roi_align3d_forward_ml(const mr3d_roi_level *levels_host, int num_levels, float finest_scale, const float *rois,
                                int num_rois, int channels, int pooled_depth, int pooled_height, int pooled_width,
                                int sample_num, float *output, mr3d_stream_t stream);
int mr3d_roi_align3d_backward_ml(const mr3d_roi_level *levels_host, int num_levels, float finest_scale,
                                 const float *top_grad, const float *rois, int num_rois, int channels, int pooled_depth,
                                 int pooled_height, int pooled_width, int sample_num, int flags, mr3d_stream_t stream);

/* bf16 feature maps (BASELINE configs[4]; the reference's kernel dispatches half too, roi_align_kernel.cu:323): as
 * mr3d_roi_align3d_forward_ml with every level's `features` pointing to bf16 NDHWC data; output fp32.
 * channels % 16 == 0, pooled * sample_num <= 64 per axis. */
int mr3d_roi_align3d_forward_ml_bf16(const mr3d_roi_level *levels_host, int num_levels, float finest_scale, const float *rois,
                                     int num_rois, int channels, int pooled_depth, int pooled_height, int pooled_width,
                                     int sample_num, float *output, mr3d_stream_t stream);

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
#define MR3D_CONV_RELU 1      /* y = max(y, 0) */
/* `up` is the COARSE result of a stride-2 adjoint (e.g. the data gradient of a strided 1x1x1 downsample conv,
 * resnet3d.py:251): added at the even positions (2u) of y only -- the zero-stuffed fine map is never materialised.
 * up_d/h/w = ceil(out / 2).  fp32 kernels only. */
#define MR3D_CONV_UP_EVEN 2
typedef struct {
  const float *x; const float *w; float *y;
  const float *shift; const float *add1; const float *add2; const float *up; const float *mask;
  int n, d, h, w_in, cin, cout;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int up_d, up_h, up_w;
  int relu;            /* flags: MR3D_CONV_RELU | MR3D_CONV_UP_EVEN */
  /* optional output scatter: y is a [n, y_d, y_h, y_w, cout] buffer and output (od,oh,ow) is written at
   * (od*ys_d, oh*ys_h, ow*ys_w) (0 = dense: ys = 1, y dims = output dims); add1/add2/mask index like y */
  int ys_d, ys_h, ys_w, y_d, y_h, y_w;
  /* Caller-lent device scratch (256-byte aligned) for kernel instances that transform the weights first: the two-dimensional
   * Winograd forms of the 3x3x3 / stride-1 convolution need mr3d_conv3d_wino2d_workspace_bytes(cin, cout).  NULL or too
   * small: the dispatch takes an instance that needs none (nothing is ever allocated inside a call).  Inside a plan run
   * the WINO2D_PREP record serves every convolution of the plan from ONE plan-owned buffer instead.
   * workspace_bytes = -1 in a mr3d_conv3d_forward_variant query means "assume the scratch is there". */
  void *workspace;
  long long workspace_bytes;
  /* Optional liveness bitmap of x (mr3d_voxel_liveness: one bit per position): the large-map 3x3x3 / stride-1 fp32 kernel writes a
   * tile whose input halo holds no live voxel from its epilogue alone.  For data-gradient calls on the sparse gradients of detector
   * training (RPN conv, FPN output convs).  NULL: no test.  Other kernel instances ignore it. */
  const unsigned char *x_live;
} mr3d_conv3d_desc;
/* Size limits (MR3D_ERR_UNSUPPORTED beyond them; the kernels address a tile with 32-bit byte offsets and 24-bit products):
 * an input plane h * w below 2^20 positions and h * w * cin below 2^24 elements, an output plane y_h * y_w * cout * ys_d below
 * 2^25 elements, the weights below 2^29 elements. */
int mr3d_conv3d_forward(const mr3d_conv3d_desc *desc, mr3d_stream_t stream);
/* bits: one bit per position p = ((n*D + d)*H + h)*W + w of the NDHWC map x (fp32, or bf16 if `bf16`; a position's channels must be
 * 16 * 2^k bytes, k <= 6) -- bit (p & 31) of word p >> 5 is set iff a channel of p is non-zero.  The buffer holds
 * mr3d_voxel_liveness_words(n*d*h*w) 32-bit words (two of them padding, zeroed). */
long long mr3d_voxel_liveness_words(long long positions);
int mr3d_voxel_liveness(const void *x, int bf16, int n, int d, int h, int w, int c, unsigned *bits, mr3d_stream_t s);
/* bytes of transformed weights U = Gh g Gw^T of one cin -> cout 3x3x3 convolution (72 floats per (ci, co), rounded up to 256 B) */
size_t mr3d_conv3d_wino2d_workspace_bytes(int cin, int cout);

/* Which kernel instance mr3d_conv3d_forward would launch for `desc` (nothing is launched; pointers may be NULL):
 * the measurement harness prices a launch with it -- a Winograd instance issues only `mfma_issue_factor` of the
 * convolution's algorithmic FLOPs (2*MAC) as MFMA work. */
typedef struct {
  int streaming_pointwise;      /* 1: the HBM-bound 1x1x1 streaming kernel (csrc/pointwise.hip) */
  int winograd;                 /* 0 direct, 2 = F(2,3) along W, 4 = F(4,3) along W, 8 = F(4,3) along W x F(2,3) along H */
  int tile_d, tile_h, tile_w;   /* output positions per workgroup tile (winograd 8, Cout 64: 4 x 8 x 8 = the wide kernel with persistent
                                   workgroups, 2 x 8 x 8 = one workgroup per tile; csrc/conv3d_wino2d.hip) */
  int cin_chunk, cout_tile;
  int parity_decomposed;
  float mfma_issue_factor;      /* MFMA FLOPs issued / algorithmic FLOPs: 1, 2/3, 1/2 or 1/3 */
} mr3d_conv3d_variant;
/* The probe sees the descriptor only: a launch whose descriptor arms a liveness bitmap (`x_live`, the sparse data gradients of the
 * pyramid) runs the 2 x 8 x 8 kernel where the probe reports the wide 4 x 8 x 8 one -- same mfma_issue_factor, other tile. */
int mr3d_conv3d_forward_variant(const mr3d_conv3d_desc *desc, mr3d_conv3d_variant *out);

/* Data gradient of a kernel-3 / stride-2 / pad-1 convolution (resnet3d.py:139, the strided conv2 of the first
 * bottleneck of layer2-4) WITHOUT zero-stuffing: the 8 output parity classes are computed from the coarse dy with the
 * 1..8 taps each of them meets.  Same descriptor: x = dy [n,d,h,w_in,cin] (cin = the forward conv's Cout), w = the
 * flipped/transposed 27-tap weights of mr3d_prepare_weights (wb), y = dx [n,y_d,y_h,y_w,cout] with y_d/y_h/y_w
 * REQUIRED and d == (y_d-1)/2+1 etc.; add1/add2/mask as in mr3d_conv3d_forward; shift/up/relu must be unset. */
int mr3d_conv3d_dgrad_s2k3(const mr3d_conv3d_desc *desc, mr3d_stream_t stream);
/* the same on bf16 maps (x = dy, y = dx, add1 / add2 / mask bf16; w = the bf16 packing of mr3d_prepare_weights_bf16's `wb`);
 * plan record CONV_DGRAD_S2 with i[25] = MR3D_DTYPE_BF16 */
int mr3d_conv3d_dgrad_s2k3_bf16(const mr3d_conv3d_desc *desc, mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * conv3d weight gradient -- the wgrad half of autograd's backward of the torch.nn.Conv3d calls above
 * (reference: loss.backward() in mmdet/core/utils/dist_utils.py:52-53 reaching the Conv3d modules).
 *   dw[tap][ci][co] += sum_{n,p} x[n, p*stride + tap - pad][ci] * dy[n, p][co]
 * x, dy NDHWC fp32; dw packed [kd*kh*kw][Cin][Cout], PRE-ZEROED by the caller (partial sums of the
 * workgroups are added with fp32 atomics: last-bit nondeterministic like any atomic reduction).
 * workspace (optional, MR3D_WGRAD_WORKSPACE_BYTES is always enough): the workgroups store their partial sums there
 * with plain stores and a second kernel adds them into dw -- deterministic, and faster than ~10^7 same-address
 * atomics; it must not be shared by calls that can run concurrently (different streams).
 * Same kernel-size / stride / channel support as mr3d_conv3d_forward.
 * ---------------------------------------------------------------------------------------------- */
#define MR3D_WGRAD_WORKSPACE_BYTES (64ll << 20)
/* flags: the caller expects dy to be zero on most position tiles -- the gradient that reaches the RPN conv and the FPN output convs
 * in detector training is non-zero around the 256 sampled anchors and under the sampled rois only -- and the large-map kernels
 * then test every dy tile and skip the zero ones (same sums: a zero tile adds zero).  Without it no test is made. */
#define MR3D_WGRAD_SPARSE_DY 1
typedef struct {
  const float *x; const float *dy; float *dw;
  float *dbias;   /* [cout] += sum_p dy[p][co] (bias / BN-beta gradient); pre-zeroed; may be NULL */
  int n, d, h, w_in, cin, cout;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int flags;                    /* MR3D_WGRAD_SPARSE_DY or 0 */
  float *workspace;             /* NULL: atomics */
  long long workspace_bytes;
} mr3d_conv3d_wgrad_desc;
int mr3d_conv3d_wgrad(const mr3d_conv3d_wgrad_desc *desc, mr3d_stream_t stream);
/* MFMA FLOPs a launch of `desc` issues / its algorithmic FLOPs (2 * MAC): 1 (direct), 2/3 (F(3,2) along W) or 1/3 (the two-dimensional
 * Winograd form); a host-only query for the measurement harness (bench.py's whole_step_issued_frac), nothing is launched */
float mr3d_conv3d_wgrad_issue_factor(const mr3d_conv3d_wgrad_desc *desc);

/* ------------------------------------------------------------------------------------------------
 * HBM-bound helpers of the backbone / neck path (activations NDHWC fp32, channels % 4 == 0).
 *   maxpool3d: nn.MaxPool3d(kernel 3, stride 2, padding 1) (mmdet/models/backbones/resnet3d.py:409,471);
 *     forward stores the argmax tap (4 x u8 per 4 channels; first maximum in (kd,kh,kw) scan order, as
 *     torch); backward gathers dy through it and, if pooled != NULL (the forward's output y), drops the windows whose
 *     pooled value is <= 0: the ReLU in front of the pool (the stem's) seen through the argmax -- y[o] IS the input at the
 *     argmax, so no un-pooled activation has to be read.  Output size floor((in-1)/2)+1 per axis.
 *   zero_stuff2: dst[2o] = src[o], 0 elsewhere (stride-2 data gradient as a stride-1 conv).
 *   upsample_nearest_adjoint_add: grad_coarse[u] += sum of grad_fine[f] over f with nearest_src(f) == u
 *     (backward of F.interpolate(mode='nearest') in mmdet/models/necks/fpn3d.py:106-108).
 *   subsample2: y[o] = x[2o] (F.max_pool3d(x, 1, stride=2), fpn3d.py:121-123) and its adjoint (+=).
 *   volume_to_ndhwc4: [N,3,D,H,W] -> [N,D,H,W,4] (4th channel 0) for the stem kernel.
 *   axpy: dst += alpha * src (count % 4 == 0).
 * ---------------------------------------------------------------------------------------------- */
int mr3d_maxpool3d_forward(const float *x, float *y, uint32_t *argmax, int n, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_maxpool3d_backward(const float *dy, const uint32_t *argmax, const float *pooled, float *dx,
                            int n, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_zero_stuff2(const float *src, float *dst, int n, int do_, int ho, int wo, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_upsample_nearest_adjoint_add(const float *grad_fine, float *grad_coarse, int n, int df, int hf, int wf,
                                      int dc, int hc, int wc, int c, mr3d_stream_t s);
int mr3d_subsample2_forward(const float *x, float *y, int n, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_subsample2_backward_add(const float *dy, float *dx, int n, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_volume_to_ndhwc4(const float *ncdhw3, float *ndhwc4, int n, int d, int h, int w, mr3d_stream_t s);
int mr3d_axpy(float *dst, const float *src, float alpha, long long count, mr3d_stream_t s);
/* ReLU backward of a layer whose forward output `act` [rows, c] was kept: out = act > 0 ? g : 0 (the `grad * (y > 0)`
 * of torch's ReLU, one pass instead of a compare and a multiply); dbias (may be NULL, PRE-ZEROED, c floats) += the
 * per-channel column sums of out (bias gradient of the layer below the ReLU).  c % 4 == 0; with dbias 256 % (c/4) == 0. */
int mr3d_relu_mask(const float *g, const float *act, float *out, long long rows, int c, float *dbias, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Table-driven parameter handling for conv (+ eval-mode BatchNorm3d or bias) layers.  All parameters and
 * BN buffers live in ONE flat fp32 arena (`params`), gradients in a mirror arena with identical offsets
 * (`grads`: the buffer the data-parallel all-reduce runs on, mmdet/core/utils/dist_utils.py:9-41).
 *   prepare_weights: folds BN-eval (norm_eval=True, resnet3d.py:330,480-486) into the conv:
 *       s = gamma / sqrt(running_var + eps);  packed wf[tap][ci_pad][co] = s[co] * W[co][ci][tap];
 *       dgrad weights wb[taps-1-tap][co][ci_pad] = same value; shift = beta - running_mean*s | bias | 0.
 *   finish_grads: from dwraw (= mr3d_conv3d_wgrad output w.r.t. the FOLDED weights) and sumdy (dbias):
 *       grads[W] += s[co]*dwraw;  dgamma += invstd*(sum_k W[c,k]*dwraw[k,c] - mean*sumdy);  dbeta += sumdy;
 *       conv bias: dbias += sumdy.   (BN gamma/beta keep receiving gradients: normalize frozen=False, :329.)
 * Offsets are in floats; -1 = absent.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  long long w_off, gamma_off, beta_off, mean_off, var_off, bias_off;   /* into params / grads          */
  long long wf_off, wb_off, shift_off, scale_off;                      /* into the packed-weight arena */
  long long dwraw_off, dbias_off;                                      /* into the scratch arena       */
  int co, ci, ci_pad, taps;
  float eps;
  int reserved;
} mr3d_layer_desc;
int mr3d_prepare_weights(const mr3d_layer_desc *table, int num_layers, const float *params, float *packed, mr3d_stream_t s);
int mr3d_finish_grads(const mr3d_layer_desc *table, int num_layers, int max_cout, const float *params,
                      const float *scratch, float *grads, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Box arithmetic of the proposal path (each replaces a chain of ~10-30 elementwise ATen launches).
 * Boxes are [x1,y1,x2,y2,z1,z2] fp32 with the reference's inclusive "+1" size convention.
 *   delta2bbox3d  mmdet/core/bbox/transforms.py:105-160: deltas [n, 6*num_classes] in the order
 *                 [dx,dy,dw,dh,dz,dd]; dw, dh AND dz, dd are clamped to +-|log(wh_ratio_clip)| (:122-128, the
 *                 centre delta dz included, reproduced on purpose); max_shape_hwd = {H, W, D} or NULL.
 *   bbox2delta3d  transforms.py:34-65 (encode).
 *   bbox_overlaps3d  mmdet/core/bbox/geometry.py:49-60: IoU matrix [m, n].
 *   grid_anchors3d   mmdet/core/anchor/anchor_generator_3d.py:55-71: flat order ((h*W + w)*D + d)*A + a.
 *   rpn_decode_level mmdet/models/anchor_heads/rpn_head_3d.py:108-132: for the k selected flat anchor indices
 *                 (int64, reference order) build anchor, gather the 6 deltas from the regression map
 *                 ([A*6, D, H, W] NCDHW or [D, H, W, A*6] NDHWC of ONE image), decode, emit dets [k,7] = box + score.
 * means/stds/max_shape are HOST arrays (6 / 6 / 3 values), copied by value into the launch.
 * ---------------------------------------------------------------------------------------------- */
int mr3d_delta2bbox3d(const float *rois, const float *deltas, int n, int num_classes, const float *means_host,
                      const float *stds_host, const int *max_shape_hwd_host, double wh_ratio_clip, float *out,
                      mr3d_stream_t s);
int mr3d_bbox2delta3d(const float *proposals, const float *gt, int n, const float *means_host, const float *stds_host,
                      float *out, mr3d_stream_t s);
int mr3d_bbox_overlaps3d(const float *b1, const float *b2, int m, int n, float *out, mr3d_stream_t s);
int mr3d_grid_anchors3d(const float *base_anchors, int num_base, int feat_d, int feat_h, int feat_w, int stride,
                        int depth_stride, float *out, mr3d_stream_t s);
int mr3d_rpn_decode_level(const long long *flat_idx, const float *scores, int k, const float *reg_map, int reg_layout,
                          const float *base_anchors, int num_base, int feat_d, int feat_h, int feat_w, int stride,
                          int depth_stride, const float *means_host, const float *stds_host,
                          const int *max_shape_hwd_host, double wh_ratio_clip, float *dets_out, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Input preparation on the device (csrc/input_ops.hip) -- SURVEY.md section 8(f) rank 3.  Replaces the per-slice host
 * loop of mmdet/datasets/coco_3d_2scales.py:245-289 (grey -> 3 channels -> mmcv.imnormalize -> impad_to_multiple ->
 * [3,D,H,W]; transforms.py:33-51) and, with out_* != d/h/w, the skimage.transform.resize of the second pathway
 * (coco_3d_2scales.py:219; scikit-image 0.18's algorithm for an up-scaling: order 1, half-pixel centres, 'mirror' border --
 * checked against scipy.ndimage.map_coordinates, the routine it calls; scikit-image itself is absent).
 *   grey: ONE-channel volume on the device, u8 or f32, axis order [H,W,D] (the .npy files, src_is_hwd = 1) or [D,H,W];
 *   out:  [3, out_d, pad_h, pad_w] fp32 planar; channel c = (g - mean[c]) / std[c]; rows/columns beyond out_h/out_w
 *         are 0 (impad pads AFTER normalisation).  mean/std: HOST arrays of 3.
 * ---------------------------------------------------------------------------------------------- */
int mr3d_volume_prepare(const void *grey, int src_is_u8, int src_is_hwd, int d, int h, int w, int out_d, int out_h, int out_w,
                        int pad_h, int pad_w, const float *mean_host, const float *std_host, float *out, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Training targets on the device (csrc/assign_ops.hip) -- SURVEY.md section 8(f) rank 1.  Nothing returns to the
 * host: counts stay in device memory and list outputs have a fixed capacity (padded with -1).
 *   max_iou_assign   mmdet/core/bbox/assigners/max_iou_assigner.py:47-196.  boxes [n, box_stride>=6], optional
 *                    `inside` u8 [n] (anchor_target.py:138-144: boxes with 0 do not take part and get -1 / 0.0),
 *                    gt [g,6], optional gt_labels i64 [g].  neg range = [neg_iou_lo, neg_iou_hi) (lo = 0 for the
 *                    reference's float form).  Outputs assigned_gt_inds i64 [n] (-1 ignore, 0 negative, k>0 = gt
 *                    k-1; the IoU matrix is never materialised), max_overlaps f32 [n], assigned_labels i64 [n] or
 *                    NULL.  On max-IoU ties between gts the lowest gt index wins (torch CPU max).  gt_max_assign_all
 *                    must be 1 (the only mode the shipped configs use) else MR3D_ERR_UNSUPPORTED; n == 0 or g == 0 is
 *                    MR3D_ERR_INVALID_ARG (the reference raises ValueError).
 *   random_sample    samplers/base_sampler.py:31-101 + random_sampler.py:18-59.  The reference draws from numpy's
 *                    global generator on the host; here the draws are an INPUT: draws_pos u32 [int(num*pos_fraction)],
 *                    draws_neg u32 [num]; draw j picks candidate draws[j] % #candidates, duplicates collapse.
 *                    Outputs: pos_inds i64 [int(num*pos_fraction)] and neg_inds i64 [num], ascending, padded with -1;
 *                    counts i32 [4] = {#pos sampled, #neg sampled, #pos candidates, #neg candidates};
 *                    sel_out i32 [n] or NULL: 1 = sampled positive, 2 = sampled negative, 0 = neither.
 *   anchor_target    anchor/anchor_target.py:146-189 on the FULL anchor set (the reference's unmap step): labels
 *                    i64 [n] (1 or gt_labels[..] for sampled positives), label_weights [n] (pos_weight<=0 -> 1),
 *                    bbox_targets / bbox_weights [n,6].  means/stds are HOST arrays of 6.
 *   concat_gt_assign assign_result.py:13-19 add_gt_: assigned_all[0:g] = 1..g, [g:] = assigned_proposals.
 *   bbox_target3d    two_stage_3d.py:179-190 + bbox/bbox_target.py:139-165 for one image, candidates = [gts; proposals]
 *                    (add_gt_as_proposals): rois [cap,7] = (batch_index, box), rows = sampled positives, then
 *                    negatives, then padding (weights 0); labels / label_weights [cap], bbox_targets / bbox_weights
 *                    [cap,6], pos_assigned_gt_inds i64 [cap] (-1 beyond the positives) or NULL.
 * ---------------------------------------------------------------------------------------------- */
size_t mr3d_max_iou_assign_workspace_bytes(int num_boxes, int num_gt);
int mr3d_max_iou_assign(const float *boxes, int box_stride, int num_boxes, const uint8_t *inside, const float *gt, int num_gt,
                        const long long *gt_labels, float pos_iou_thr, float neg_iou_lo, float neg_iou_hi, float min_pos_iou,
                        int gt_max_assign_all, long long *assigned_gt_inds, float *max_overlaps, long long *assigned_labels,
                        void *workspace, size_t workspace_bytes, mr3d_stream_t s);
size_t mr3d_random_sample_workspace_bytes(int n);
int mr3d_random_sample(const long long *assigned_gt_inds, int n, int num, float pos_fraction, float neg_pos_ub,
                       const uint32_t *draws_pos, const uint32_t *draws_neg, long long *pos_inds, long long *neg_inds,
                       int *counts, int *sel_out, void *workspace, size_t workspace_bytes, mr3d_stream_t s);
int mr3d_anchor_target(const float *anchors, int anchor_stride, int n, const long long *assigned_gt_inds, const int *sel,
                       const float *gt, const long long *gt_labels, const float *means_host, const float *stds_host,
                       float pos_weight, long long *labels, float *label_weights, float *bbox_targets, float *bbox_weights,
                       mr3d_stream_t s);
int mr3d_concat_gt_assign(const long long *assigned_proposals, int num_proposals, int num_gt, long long *assigned_all,
                          mr3d_stream_t s);
int mr3d_bbox_target3d(const float *proposals, int proposal_stride, const float *gt, int num_gt, const long long *gt_labels,
                       const long long *assigned_all, const long long *pos_inds, const long long *neg_inds, const int *counts,
                       int cap, int batch_index, const float *means_host, const float *stds_host, float pos_weight,
                       float *rois, long long *labels, float *label_weights, float *bbox_targets, float *bbox_weights,
                       long long *pos_assigned_gt_inds, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Losses, value + gradient w.r.t. the prediction in one pass (csrc/loss_ops.hip; mmdet/core/loss/losses.py).
 * `workspace` = mr3d_loss_workspace_bytes() bytes of device memory; loss_out is ONE device float (overwritten);
 * grad may be NULL.  Sums are reduced in a fixed order (deterministic).  avg_factor: device scalar if
 * avg_factor_dev != NULL else the host value.
 *   weighted_bce_logits  losses.py:28-36 (+ _expand_binary_labels :140-147): score [n,c], labels i64 [n], weights [n].
 *   weighted_smoothl1    losses.py:77-118: pred/target/weights [count] elementwise.
 *   weighted_softmax_ce  losses.py:17-25 (reduce=True, avg_factor = max(#(weight>0), 1)) + accuracy :121-137 (top-1,
 *                        percent) over the first *valid_rows_dev rows (NULL: all n).
 *   mask_bce             losses.py:69-74: pred [p,c,spatial], target [p,spatial], labels i64 [p]; mean over the first
 *                        *valid_rows_dev rois (NULL: all p).
 *   avg_factor_from_counts  mode 0: sum_i max(pos_i,1) + max(neg_i,1) (anchor_target.py:69-70, anchor_head_3d.py:213);
 *                        mode 1: sum_i pos_i + neg_i (number of sampled rois = bbox_targets.size(0), bbox_head_3d.py:122);
 *                        counts = num_imgs records of count_stride ints, {pos, neg, ...} as written by random_sample.
 * ---------------------------------------------------------------------------------------------- */
size_t mr3d_loss_workspace_bytes(void);
int mr3d_avg_factor_from_counts(const int *counts, int count_stride, int num_imgs, int mode, float *out, mr3d_stream_t s);
int mr3d_weighted_bce_logits(const float *score, const long long *labels, const float *weights, long long n, int c,
                             const float *avg_factor_dev, float avg_factor_host, float *loss_out, float *grad,
                             void *workspace, mr3d_stream_t s);
int mr3d_weighted_smoothl1(const float *pred, const float *target, const float *weights, long long count, float beta,
                           const float *avg_factor_dev, float avg_factor_host, float *loss_out, float *grad, void *workspace,
                           mr3d_stream_t s);
int mr3d_weighted_softmax_ce(const float *score, const long long *labels, const float *weights, int n, int c,
                             const int *valid_rows_dev, float *loss_out, float *accuracy_out, float *grad, void *workspace,
                             mr3d_stream_t s);
int mr3d_mask_bce(const float *pred, const float *target, const long long *labels, int num_rois, int c, long long spatial,
                  const int *valid_rows_dev, float *loss_out, float *grad, void *workspace, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Mask-crop (replaces the CPU skimage loops of mmdet/core/mask/mask_target.py:17-49 and
 * mmdet/models/mask_heads/fcn_mask_head_3d.py:163-189).  skimage.transform.resize is a third-party dependency
 * (scikit-image==0.18.0) absent from the reference tree: both ops restate its published algorithm (Gaussian pre-filter on
 * shrinking axes, order-1 'mirror' sampling; csrc/mask_ops.hip) and are checked against scipy.ndimage, which it calls.
 *   mask_target3d: gt_masks u8 [G,D,H,W] (device), pos_boxes [P, box_stride>=6] fp32, pos_assigned_gt_inds i64 [P]
 *                  -> targets fp32 [P, md, mh, mw] in {0,1}.
 *   mask_paste3d:  mask_prob fp32 [P, md, mh, mw] (already sigmoid-ed, the det's class channel), det_boxes
 *                  [P, box_stride>=6] -> out_masks u8 [P, D, H, W], PRE-ZEROED by the caller.
 * ---------------------------------------------------------------------------------------------- */
int mr3d_mask_target3d(const uint8_t *gt_masks, int depth, int height, int width, const float *pos_boxes, int box_stride,
                       const long long *pos_assigned_gt_inds, int num_pos, int mask_d, int mask_h, int mask_w,
                       float *targets, mr3d_stream_t s);
int mr3d_mask_paste3d(const float *mask_prob, int mask_d, int mask_h, int mask_w, const float *det_boxes, int box_stride,
                      int num_dets, float threshold, int depth, int height, int width, uint8_t *out_masks, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Fully connected layers of the bbox heads (mmdet/models/bbox_heads/convfc_bbox_head_3d.py:104-140: shared_fcs, fc_cls,
 * fc_reg -- torch.nn.Linear in the reference): fp32 MFMA GEMMs, row-major operands exactly as torch.nn.Linear holds them,
 * no transposed copies (csrc/linear.hip).
 *   forward: y [rows][out] = x [rows][in] * weight[out][in]^T + bias (bias may be NULL), optional ReLU
 *   dgrad:   dx [rows][in] = dy [rows][out] * weight[out][in]
 *   wgrad:   dw [out][in]  = dy^T x          (the bias gradient is a column sum of dy: mr3d callers use torch.sum)
 * Outputs are WRITTEN.  Launches whose output has few tiles split the K loop: with `workspace` (>= mr3d_linear_workspace_bytes
 * of the layer; device memory, contents irrelevant) the partial products are summed in a fixed order -- bit-reproducible;
 * with workspace == NULL they are added with fp32 atomics (last-bit run-to-run differences).
 * ---------------------------------------------------------------------------------------------- */
size_t mr3d_linear_workspace_bytes(int rows, int in_features, int out_features);
int mr3d_linear_forward(const float *x, const float *weight, const float *bias, float *y, int rows, int in_features,
                        int out_features, int relu, float *workspace, long long workspace_bytes, mr3d_stream_t s);
int mr3d_linear_dgrad(const float *dy, const float *weight, float *dx, int rows, int in_features, int out_features,
                      float *workspace, long long workspace_bytes, mr3d_stream_t s);
int mr3d_linear_wgrad(const float *dy, const float *x, float *dw, int rows, int in_features, int out_features,
                      float *workspace, long long workspace_bytes, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * DeformConv3d (v1) / modulated (v2): the sampling halves of the operator; the GEMM halves are
 * mr3d_conv3d_forward / mr3d_conv3d_wgrad over the column matrix (see csrc/deform_conv3d.hip).
 * Extends the reference's 2D operator (mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:83-435,569-766 and
 * deform_conv_cuda.cpp:151-479): same sampling rule (0 outside (-1, dim), floor(), corners masked one by one),
 * offsets per tap as consecutive (d, h, w) values.  Build-defined in 3D: the reference never runs DCN on 5-D input.
 *   x [N,D,H,W,C] NDHWC; offset [N,Do,Ho,Wo, dg*K*3]; mask [N,Do,Ho,Wo, dg*K] or NULL (K = kd*kh*kw);
 *   columns / grad_columns [npos][K][C] for the chunk of output positions [pos0, pos0+npos) (flattened n,od,oh,ow).
 *   col2im: grad_x [N,D,H,W,C] += (fp32 atomics; PRE-ZEROED by the caller; may be NULL), grad_offset / grad_mask
 *   are WRITTEN for the chunk's positions (may be NULL).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int n, d, h, w, c;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dil_d, dil_h, dil_w;
  int dg;
  long long pos0, npos;
} mr3d_deform_conv3d_desc;
int mr3d_deform_im2col3d(const mr3d_deform_conv3d_desc *d, const float *x, const float *offset, const float *mask,
                         float *columns, mr3d_stream_t s);
int mr3d_deform_col2im3d(const mr3d_deform_conv3d_desc *d, const float *x, const float *offset, const float *mask,
                         const float *grad_columns, float *grad_x, float *grad_offset, float *grad_mask, mr3d_stream_t s);

/* Fused form for the FPN-neck shape (c == cout == 64, dg == 1, no modulation; any kernel / stride / padding / dilation):
 * no column matrix -- a workgroup samples a 128-position tile tap by tap into LDS and multiplies it on the MFMA units
 * (csrc/deform_conv3d_fused.hip).  pos0 / npos of the descriptor are ignored (whole output).  Same sampling rule as above.
 *   mr3d_deform_conv3d_fused_supported: 1 if the three entry points accept the descriptor, else 0 (use the column form).
 *   pack_weights: w [cout][cin][taps] (the reference's OIDHW weight flattened) -> w_packed, 3 * taps * 64 * 64 + 1 floats: the
 *                 forward operand and the two backward operands (direct / tiled kernel), each [taps][64][64] in its kernel's
 *                 GEMM-k order, then max_(tap,ci) ||W[:,ci,tap]||_2 (the tiled kernel's accumulation scale).
 *   forward:  y [N*Do*Ho*Wo][64] written.
 *   wgrad:    dw_packed [taps][cin][cout] += (fp32 atomics, PRE-ZEROED by the caller).
 *   dgrad:    dx [N,D,H,W,64] += (fp32 atomics, PRE-ZEROED; may be NULL), doffset [N,Do,Ho,Wo,taps*3] += (PRE-ZEROED; may
 *             be NULL).  A 2x4x8 block of positions whose reach fits the LDS budget (the neck: 6x8x12 voxels) accumulates
 *             dx in LDS (exact two-word integer sums, see the kernel) and touches HBM once per voxel; wider kernels /
 *             strides use direct atomics (MR3D_DCN_DGRAD=direct forces it).
 * Other shapes: MR3D_ERR_UNSUPPORTED. */
int mr3d_deform_conv3d_fused_supported(const mr3d_deform_conv3d_desc *d, int cout);
int mr3d_deform_conv3d_pack_weights(const float *w, int cout, int cin, int taps, float *w_packed, mr3d_stream_t s);
int mr3d_deform_conv3d_forward_fused(const mr3d_deform_conv3d_desc *d, int cout, const float *x, const float *offset,
                                     const float *w_packed, float *y, mr3d_stream_t s);
int mr3d_deform_conv3d_wgrad_fused(const mr3d_deform_conv3d_desc *d, int cout, const float *x, const float *offset,
                                   const float *dy, float *dw_packed, mr3d_stream_t s);
int mr3d_deform_conv3d_dgrad_fused(const mr3d_deform_conv3d_desc *d, int cout, const float *x, const float *offset,
                                   const float *dy, const float *w_packed, float *dx, float *doffset, mr3d_stream_t s);

/* bf16 form of the same operator (BASELINE configs[4]: bf16 storage + DeformConv3d neck; csrc/deform_conv3d_bf16.hip): x, the
 * offsets, y / dy and the offset gradient are bf16 (NDHWC; the offsets [position][off_stride] with the taps' (dd, dh, dw) in the
 * first taps*3 entries of a row -- the offset conv's output padded to its N tile), sums are fp32.  A workgroup copies the voxel
 * box a 4x4x8 block of positions can reach with |offset| < 1 into LDS once for all taps and samples from there; samples that leave
 * the box read global memory.  c == cout == 64, dg == 1, the box must fit 768 voxels (the neck's 3x3x3 / stride 1: 8x8x12).
 *   w_packed / wt_packed: bf16 [taps][k / 8][n = 64][8], the B-operand layout of mr3d_conv3d_forward_bf16, of W (k = ci, n = co)
 *            resp. of its transpose (k = co, n = ci).
 *   wgrad:   dw [taps][cin][cout] fp32 += (atomics, PRE-ZEROED).
 *   doffset: [position][doff_stride] bf16, PRE-ZEROED: tiles of 4x4x8 positions whose dy is zero are skipped (so are they in wgrad and
 *            in the fp32 form: the gradient of a pyramid level in detector training is sparse).
 *   dx:      fp32 [N,D,H,W,64] += (PRE-ZEROED), the tiled LDS accumulator of the fp32 form; w_packed_f32 =
 *            mr3d_deform_conv3d_pack_weights. */
int mr3d_dcn_bf16_supported(const mr3d_deform_conv3d_desc *d, int cout);
int mr3d_dcn_bf16_forward(const mr3d_deform_conv3d_desc *d, int cout, const uint16_t *x, const uint16_t *offset, int off_stride,
                          const uint16_t *w_packed, uint16_t *y, mr3d_stream_t s);
int mr3d_dcn_bf16_wgrad(const mr3d_deform_conv3d_desc *d, int cout, const uint16_t *x, const uint16_t *offset, int off_stride,
                        const uint16_t *dy, float *dw, mr3d_stream_t s);
int mr3d_dcn_bf16_doffset(const mr3d_deform_conv3d_desc *d, int cout, const uint16_t *x, const uint16_t *offset, int off_stride,
                          const uint16_t *dy, const uint16_t *wt_packed, uint16_t *doffset, int doff_stride, mr3d_stream_t s);
int mr3d_dcn_bf16_dx(const mr3d_deform_conv3d_desc *d, int cout, const uint16_t *offset, int off_stride, const uint16_t *dy,
                     const float *w_packed_f32, float *dx, mr3d_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * ResNet3D stem -- Conv3d(3, 16, kernel 7, stride (1,2,2), padding 3) + folded BN + ReLU, forward and weight
 * gradient (mmdet/models/backbones/resnet3d.py:397-409,467-469), specialised: reads the volume in the
 * reference's own planar [N,3,D,H,W] layout, writes / reads NDHWC [N,D,H/2,W/2,16].
 * w_packed / dw_packed: [343][4][16] (the generic packing of mr3d_prepare_weights with ci_pad = 4).
 * dw_packed and dbias must be pre-zeroed (atomic accumulation).
 * ---------------------------------------------------------------------------------------------- */
int mr3d_stem_conv_forward(const float *x_ncdhw, const float *w_packed, const float *shift, float *y,
                           int n, int d, int h, int w, int relu, mr3d_stream_t stream);
int mr3d_stem_conv_wgrad(const float *x_ncdhw, const float *dy, float *dw_packed, float *dbias,
                         int n, int d, int h, int w, mr3d_stream_t stream);
/* same, with the optional partial-sum workspace of mr3d_conv3d_wgrad_desc (NULL / too small: atomics) */
int mr3d_stem_conv_wgrad_ws(const float *x_ncdhw, const float *dy, float *dw_packed, float *dbias,
                            int n, int d, int h, int w, float *workspace, long long workspace_bytes, mr3d_stream_t stream);
/* the stem's weight gradient with the max-pool backward (resnet3d.py:471, nn.MaxPool3d(3, 2, 1)) and the stem's ReLU mask
 * folded into its dy staging: g_pooled [N, Dp, Hp, Wp, 16] is the gradient of the POOLED map, argmax / pooled are
 * mr3d_maxpool3d_forward's outputs (pooled = NULL: no ReLU mask).  Equals mr3d_maxpool3d_backward followed by
 * mr3d_stem_conv_wgrad_ws up to the summation order of windows that share an arg-max; the un-pooled gradient
 * (335 MB on a 512x512x80 volume) is never materialised. */
int mr3d_stem_conv_wgrad_pooled(const float *x_ncdhw, const float *g_pooled, const uint32_t *argmax, const float *pooled,
                                float *dw_packed, float *dbias, int n, int d, int h, int w, float *workspace,
                                long long workspace_bytes, mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 storage path (BASELINE.json configs[4]): the same layers with activations and activation gradients stored as bf16
 * NDHWC in HBM, multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; parameters, parameter gradients, folded
 * shifts and the stem's input volume stay fp32.  Descriptors are the fp32 ones with every activation / weight pointer
 * reinterpreted as bf16 (shift, dw, dbias: fp32).  Reference: the autocast-style run of the Conv3d / BatchNorm3d / ReLU
 * graph of resnet3d.py:198-233,467-478 and fpn3d.py:95-134 (the reference itself ships fp32 only; its RoIAlign kernels
 * dispatch half, roi_align_kernel.cu:323).
 *   conv3d_forward_bf16   cubic kernels 1 / 3, stride 1 / 2, Cin % 16 == 0, Cout % 8 == 0; weights packed by
 *                         mr3d_prepare_weights_bf16: wf16[tap][Cin16/8][Cout32][8] (dgrad: wb16[taps-1-tap][Cout16/8][Cin32][8]),
 *                         channel counts rounded up to 16 (K) / 32 (N), zero filled.
 *   conv3d_wgrad_bf16     x, dy bf16; dw fp32 [taps][Cin][Cout] PRE-ZEROED (atomics), dbias fp32 or NULL.
 *   prepare_weights_bf16  table as mr3d_prepare_weights with wf_off / wb_off = offsets in bf16 ELEMENTS into packed16
 *                         (shift / scale are produced by the fp32 call).
 *   stem: fp32 volume and fp32 MFMA arithmetic, bf16 output (forward) / bf16 dy (weight gradient).
 * ---------------------------------------------------------------------------------------------- */
int mr3d_conv3d_forward_bf16(const mr3d_conv3d_desc *desc, mr3d_stream_t stream);
int mr3d_conv3d_wgrad_bf16(const mr3d_conv3d_wgrad_desc *desc, mr3d_stream_t stream);
int mr3d_prepare_weights_bf16(const mr3d_layer_desc *table, int num_layers, const float *params, void *packed16, mr3d_stream_t s);
int mr3d_maxpool3d_forward_bf16(const void *x, void *y, uint32_t *argmax, int n, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_maxpool3d_backward_bf16(const void *dy, const uint32_t *argmax, const void *pooled, void *dx, int n, int d, int h, int w,
                                 int c, mr3d_stream_t s);
int mr3d_zero_stuff2_bf16(const void *src, void *dst, int n, int do_, int ho, int wo, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_upsample_nearest_adjoint_add_bf16(const void *grad_fine, void *grad_coarse, int n, int df, int hf, int wf, int dc, int hc,
                                           int wc, int c, mr3d_stream_t s);
int mr3d_subsample2_forward_bf16(const void *x, void *y, int n, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_subsample2_backward_add_bf16(const void *dy, void *dx, int n, int d, int h, int w, int c, mr3d_stream_t s);
int mr3d_stem_conv_forward_bf16out(const float *x_ncdhw, const float *w_packed, const float *shift, void *y_bf16, int n, int d,
                                   int h, int w, int relu, mr3d_stream_t stream);
int mr3d_stem_conv_wgrad_bf16dy(const float *x_ncdhw, const void *dy_bf16, float *dw_packed, float *dbias, int n, int d, int h,
                                int w, mr3d_stream_t stream);
/* The stem on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16; csrc/stem_bf16.hip): the fp32 volume is converted to bf16 while
 * it is staged, accumulation fp32.  w16 = mr3d_stem_pack_weights_bf16(w_packed): [49 (kd,kh)][4 c][16 cout][8 kw] bf16 (50 KB)
 * from the generic fp32 packing [343][4][16].  Plan records: STEM_FWD with i[25] = bf16 and i[5] = 1 (p[1] = w16), STEM_WGRAD with
 * i[25] = bf16 and i[6] = 1; dw_packed / dbias fp32, PRE-ZEROED, same layout as the fp32 stem's.
 * Size limit: one image of the volume (3 x d x h x w fp32) below 2^31 bytes -- the slab fetch addresses it with 32-bit buffer
 * offsets; MR3D_ERR_UNSUPPORTED beyond (178 M voxels; the second pathway's 120 x 768 x 768 volume has 71 M). */
int mr3d_stem_pack_weights_bf16(const float *w_packed, void *w16, mr3d_stream_t stream);
int mr3d_stem_conv_forward_bf16(const float *x_ncdhw, const void *w16, const float *shift, void *y_bf16, int n, int d, int h, int w,
                                int relu, mr3d_stream_t stream);
int mr3d_stem_conv_wgrad_bf16(const float *x_ncdhw, const void *dy_bf16, float *dw_packed, float *dbias, int n, int d, int h, int w,
                              mr3d_stream_t stream);
/* bf16 form of mr3d_stem_conv_wgrad_pooled: g_pooled / pooled are bf16 [N, Dp, Hp, Wp, 16] (mr3d_maxpool3d_forward_bf16's
 * outputs); window sums are formed in fp32 and rounded to bf16 once, like mr3d_maxpool3d_backward_bf16 does.
 * Plan record: STEM_WGRAD with i[25] = bf16, i[6] = 1, i[7] = 1, p[1] = g_pooled, p[5] = argmax, p[6] = pooled. */
int mr3d_stem_conv_wgrad_bf16_pooled(const float *x_ncdhw, const void *g_pooled_bf16, const uint32_t *argmax, const void *pooled_bf16,
                                     float *dw_packed, float *dbias, int n, int d, int h, int w, mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Two chained pointwise (1x1x1, stride 1) convolutions in one launch (csrc/pw_pair.hip), for the small maps of layer3
 * where every separate launch is latency-bound: between two bottleneck blocks (resnet3d.py:198-233)
 *     y1 = act1(x  * w1 + shift1 + add1)      [P, c0] -> [P, c1]     e.g. conv3 + bn3 + identity + ReLU
 *     y2 = act2(y1 * w2 + shift2)             [P, c1] -> [P, c2]     e.g. the next block's conv1 + bn1 + ReLU
 * act = ReLU if relu, then, if mask != NULL, y = mask > 0 ? y : 0 (the backward pass: conv1'^T + residual gradient with the
 * mask of the block input, then conv3^T with the mask of a2).  y1 is written to global memory and kept in LDS for the second
 * product.  Weights packed [cin][cout] as for mr3d_conv3d_forward.  Shape: (64, 256, 64) with fewer than
 * 32,768 positions (layer3); anything else: MR3D_ERR_UNSUPPORTED (mr3d_pointwise_pair_supported answers without a launch).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float *x; const float *w1; const float *shift1; const float *add1; const float *mask1; float *y1;
  const float *w2; const float *shift2; const float *mask2; float *y2;
  long long positions;
  int c0, c1, c2, relu1, relu2;
} mr3d_pw_pair_desc;
int mr3d_pointwise_pair_supported(int c0, int c1, int c2, long long positions);
int mr3d_pointwise_pair(const mr3d_pw_pair_desc *desc, mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RPN proposal selection without a host round trip -- RPNHead3D.get_bboxes_single
 * (mmdet/models/anchor_heads/rpn_head_3d.py:72-149): per level sigmoid -> drop flagged-out anchors (:95-106) -> top nms_pre
 * (:108-112) -> decode -> nms -> first nms_post (:133-135); then top max_num over all levels (:141-148).
 *   mr3d_rpn_select_topk: for every level with more than nms_pre anchors, the nms_pre highest-scoring anchors among those
 *     whose `valid` byte is non-zero (NULL: all) -- an exact radix select on the logits; levels with at most nms_pre anchors
 *     contribute all of them (no flags, as the reference).  Level l owns rows [sum_{j<l} k_j, +k_l) of flat_out / scores_out,
 *     k_l = min(anchors_l, nms_pre); its first counts_out[l] rows are real, in ascending anchor order: flat index
 *     ((h*W + w)*D + d)*A + a (the reference's permute(2,3,1,0) order) and sigmoid(logit); the rest is filler.
 *   The rows then go through mr3d_rpn_decode_level and mr3d_nms3d_batched_dev (seg_rows_dev = counts_out).
 *   mr3d_rpn_merge_levels: the first min(num_keep[l], nms_post) kept rows of every level, concatenated in level order, then
 *     the max_num best by (score descending, position ascending): proposals_out [max_num, 7], zero rows behind *count_out.
 *     dets / keep are laid out as for mr3d_nms3d_batched (seg_rows_host = the sets' capacities).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float *logits;                 /* [A, D, H, W] objectness logits, element strides below (any memory format) */
  const void *valid;                   /* uint8 per anchor in flat (h, w, d, a) order, or NULL */
  long long stride_a, stride_d, stride_h, stride_w;
  int anchors, depth, height, width;
} mr3d_rpn_level;
size_t mr3d_rpn_select_topk_workspace_bytes(const mr3d_rpn_level *levels_host, int num_levels, int nms_pre);
int mr3d_rpn_select_topk(const mr3d_rpn_level *levels_host, int num_levels, int nms_pre, long long *flat_out, float *scores_out,
                         int32_t *counts_out, void *workspace, size_t workspace_bytes, mr3d_stream_t stream);
int mr3d_rpn_merge_levels(const float *dets, const int *seg_rows_host, int num_levels, const long long *keep,
                          const int32_t *num_keep, int nms_post, int max_num, float *proposals_out, int32_t *count_out,
                          mr3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Plan executor: runs a pre-built list of the operations above with ONE call (the host-side loop of
 * ~500 launches per training step stays in native code; the call only enqueues, so it can be captured
 * in a hipGraph).  Replaces the Python module-by-module dispatch of ResNet3D.forward / FPN3D.forward
 * (resnet3d.py:467-478, fpn3d.py:95-134) and of autograd's backward over them.
 * Field use per kind (unused = 0):
 *   CONV_FWD      p = {x,w,y,shift,add1,add2,up,mask}   i = the 25 ints of mr3d_conv3d_desc in order
 *   CONV_WGRAD    p = {x,dy,dw,dbias,workspace}         i = the 15 ints of mr3d_conv3d_wgrad_desc, i[15..16] = workspace bytes (lo, hi), i[17] = flags
 *   MAXPOOL_FWD   p = {x,y,argmax}                      i = {n,d,h,w,c}
 *   MAXPOOL_BWD   p = {dy,argmax,pooled,dx}             i = {n,d,h,w,c}
 *   ZERO_STUFF2   p = {src,dst}                         i = {n,do,ho,wo,d,h,w,c}
 *   UPSAMPLE_ADJ  p = {grad_fine,grad_coarse}           i = {n,df,hf,wf,dc,hc,wc,c}
 *   SUBSAMPLE_FWD p = {x,y}   SUBSAMPLE_BWD p = {dy,dx} i = {n,d,h,w,c}
 *   VOLUME_PACK   p = {ncdhw3,ndhwc4}                   i = {n,d,h,w}
 *   MEMSET0       p = {dst}                             i = {bytes_lo, bytes_hi}
 *   AXPY          p = {dst,src}  f[0] = alpha           i = {count_lo, count_hi}
 *   PREP_WEIGHTS  p = {table,params,packed}             i = {num_layers}
 *   FINISH_GRADS  p = {table,params,scratch,grads}      i = {num_layers,max_cout}
 *   STEM_FWD      p = {x_ncdhw,w_packed,shift,y}        i = {n,d,h,w,relu}
 *   STEM_WGRAD    p = {x_ncdhw,dy,dw_packed,dbias,workspace}  i = {n,d,h,w, workspace bytes lo, hi}
 *                 i[7] = 1: mr3d_stem_conv_wgrad_pooled with p[1] = g_pooled, p[5] = argmax, p[6] = pooled (fp32 plans)
 *   CONV_DGRAD_S2 p / i as CONV_FWD (mr3d_conv3d_dgrad_s2k3: x = dy, w = wb, y = dx; y_d/y_h/y_w = dx's grid)
 *   PW_PAIR       p = {x,w1,add1,mask1,y1,w2,mask2,y2}  i = {c0,c1,c2,relu1,relu2, positions lo,hi, shift1 lo,hi, shift2 lo,hi}
 *                 (mr3d_pointwise_pair; the two shift pointers travel as 64-bit integers)
 *   VOXEL_LIVENESS p = {x, bits}  i = {n,d,h,w,c, arm}: mr3d_voxel_liveness of x (storage type i[25]); arm = 1: the NEXT CONV_FWD
 *                 record of the run takes the bitmap as its mr3d_conv3d_desc.x_live (a record has no pointer slot left for it)
 *   SYNC          i = {mode, slot}: ordering between the caller's stream and the internal side stream (see below)
 *   WINO2D_PREP   p = {workspace}  i = {bytes_lo, bytes_hi}: transforms, in ONE launch, the weights of every fp32 CONV_FWD
 *                 record between here and the next PREP_WEIGHTS record that qualifies for the two-dimensional Winograd
 *                 kernels, into the caller's workspace (sum of mr3d_conv3d_wino2d_workspace_bytes over the distinct weight
 *                 pointers; MR3D_ERR_WORKSPACE if too small).  CONV_FWD records whose weights were not prepared take the
 *                 one-dimensional Winograd / direct instances.
 * PREP_WEIGHTS / FINISH_GRADS take any sub-range of the layer table (p[0] = first descriptor of the range).
 * Returns the first non-OK status; *failed_index (host int, may be NULL) receives the op index.
 * ---------------------------------------------------------------------------------------------- */
typedef enum {
  MR3D_OP_CONV_FWD = 1, MR3D_OP_CONV_WGRAD = 2, MR3D_OP_MAXPOOL_FWD = 3, MR3D_OP_MAXPOOL_BWD = 4,
  MR3D_OP_ZERO_STUFF2 = 5, MR3D_OP_UPSAMPLE_ADJ = 6, MR3D_OP_SUBSAMPLE_FWD = 7, MR3D_OP_SUBSAMPLE_BWD = 8,
  MR3D_OP_VOLUME_PACK = 9, MR3D_OP_MEMSET0 = 10, MR3D_OP_AXPY = 11, MR3D_OP_PREP_WEIGHTS = 12,
  MR3D_OP_FINISH_GRADS = 13, MR3D_OP_STEM_FWD = 14, MR3D_OP_STEM_WGRAD = 15, MR3D_OP_CONV_DGRAD_S2 = 16,
  MR3D_OP_PREP_WEIGHTS_BF16 = 17,     /* p = {table16, params, packed16}   i = {num_layers} */
  MR3D_OP_STEM_PACK_BF16 = 18,        /* p = {w_packed (fp32 [343][4][16]), w16} */
  MR3D_OP_WINO2D_PREP = 20,
  MR3D_OP_PW_PAIR = 21,
  MR3D_OP_VOXEL_LIVENESS = 22,
  MR3D_OP_SYNC = 19,                  /* i[0] = 0: the caller's stream waits for everything queued on the side stream;
                                         1: remember this point of the side queue in slot i[1] (0..7);
                                         2: the caller's stream waits for the point remembered in slot i[1] */
  MR3D_OP_KIND_END
} mr3d_op_kind;
/* i[25] of a record = storage type of the activations it touches: 0 = fp32, MR3D_DTYPE_BF16 = bf16 (the *_bf16 entry
 * points below; CONV_FWD / CONV_WGRAD / MAXPOOL_* / ZERO_STUFF2 / UPSAMPLE_ADJ / SUBSAMPLE_* / STEM_FWD (bf16 output) /
 * STEM_WGRAD (bf16 dy)).  CONV_DGRAD_S2 has no bf16 form: strided 3x3x3 data gradients take ZERO_STUFF2 + CONV_FWD. */
#define MR3D_DTYPE_BF16 1
typedef struct {
  int32_t kind;
  int32_t i[27];
  float f[4];
  void *p[8];
} mr3d_op;
int mr3d_plan_run(const mr3d_op *ops_host, int num_ops, int *failed_index_host, mr3d_stream_t stream);
/* CONV_WGRAD / STEM_WGRAD / PREP_WEIGHTS(_BF16) / MEMSET0 / FINISH_GRADS / CONV_FWD / SUBSAMPLE_FWD / VOXEL_LIVENESS records
 * with i[26] == 1 run on an internal side stream beside the caller's chain (fork by event before the op; the side stream is
 * one in-order queue; a record on the caller's stream that reads a side record's output must sit behind a SYNC).  The caller's
 * stream re-joins it at SYNC records, before a FINISH_GRADS record that runs on the caller's stream, and at the end of
 * the call.  0 disables the side stream (everything on `stream`, in list order); default 1. */
void mr3d_plan_set_side_stream(int enable);
/* mr3d_plan_run with HIP timing events around the records `marks[0..num_marks)` (ascending op indices, no SYNC records),
 * recorded on the stream each record really runs on -- the plan is NOT split, so the timed step is the production step.
 * pairs[j] receives two events the library creates; read them with mr3d_event_pair_elapsed_ms after synchronising the
 * device and release them with mr3d_event_pair_destroy. */
/* `stream` waits for the point of the side queue that the last SYNC(mark, slot) record of a plan run on the current device
 * remembered (a never-recorded slot is no wait).  The gradient exchange uses it: the all-reduce of everything but the
 * stem's parameters starts on a communication stream as soon as their gradient finishing is done, underneath the stem's
 * weight-gradient kernel. */
int mr3d_plan_stream_wait_mark(int slot, mr3d_stream_t stream);
/* The same wait for a consumer that must KNOW whether it is ordered: *waited = 1 when `stream` now waits for the point a
 * SYNC(mark, slot) record really recorded since the last checked wait on that slot; *waited = 0 (and no wait) when the slot
 * was not recorded by a live run -- side stream disabled (mr3d_plan_set_side_stream(0): SYNC records are skipped), the plan
 * was captured into a hipGraph (a replay never re-records the event), or no plan has run.  The caller then orders itself
 * behind the stream the plan ran on.  A recorded mark is consumed by one checked wait. */
int mr3d_plan_stream_wait_mark_checked(int slot, mr3d_stream_t stream, int *waited);
typedef struct { void *start, *stop; } mr3d_event_pair;
int mr3d_plan_run_marked(const mr3d_op *ops_host, int num_ops, int *failed_index_host, mr3d_stream_t stream, const int *marks,
                         int num_marks, mr3d_event_pair *pairs);
int mr3d_event_pair_elapsed_ms(const mr3d_event_pair *pair, float *ms);
void mr3d_event_pair_destroy(mr3d_event_pair *pair);
/* hipGraph replay of a plan: capture the launch sequence once (every buffer it names must stay allocated; run the
 * plan normally at least once BEFORE capturing so that one-time kernel attributes are set), then replay it with
 * one call per pass.  The graph is launched into `stream`. */
typedef struct mr3d_graph mr3d_graph;
int mr3d_plan_capture(const mr3d_op *ops_host, int num_ops, mr3d_graph **out);
int mr3d_graph_launch(mr3d_graph *g, mr3d_stream_t stream);
void mr3d_graph_destroy(mr3d_graph *g);

#ifdef __cplusplus
}
#endif
#endif /* MR3D_H_ */

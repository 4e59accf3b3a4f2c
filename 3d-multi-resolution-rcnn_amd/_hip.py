"""ctypes binding of libmr3d_hip.so (C ABI: include/mr3d.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is
raised.  PyTorch is used for device memory and streams only.
"""
import ctypes
import os
import subprocess

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# MR3D_LIB: A/B measurements of two builds of the library on one GPU box (tools/ab.sh); the default is the in-tree build
_SO = os.environ.get('MR3D_LIB') or os.path.join(_PKG, 'libmr3d_hip.so')
_CSRC = os.path.join(_PKG, 'csrc')
_lib = None

OK = 0
LAYOUT_NCDHW = 0
LAYOUT_NDHWC = 1
ROIALIGN_BUG_COMPAT = 1


class Mr3dError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into libmr3d_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ['make', '-C', _CSRC, '-j', str(min(8, os.cpu_count() or 1))]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise Mr3dError(
                f'{_SO} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C {_CSRC}`). There is no CPU fallback for the HIP hot path.')
        L = ctypes.CDLL(_SO)
        L.mr3d_version.restype = ctypes.c_char_p
        L.mr3d_status_string.restype = ctypes.c_char_p
        L.mr3d_status_string.argtypes = [ctypes.c_int]
        L.mr3d_nms3d_workspace_bytes.restype = ctypes.c_size_t
        L.mr3d_nms3d_workspace_bytes.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def check(status, what):
    if status != OK:
        raise Mr3dError(f'{what}: {lib().mr3d_status_string(int(status)).decode()} (status {status})')


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def feature_layout(t):
    """NCDHW-contiguous or channels_last_3d; anything else is an error (the reference demands
    contiguous inputs: mmdet/ops/roi_align/src/roi_align_cuda.cpp:34-39)."""
    if t.is_contiguous():
        return LAYOUT_NCDHW
    if t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
        return LAYOUT_NDHWC
    raise RuntimeError('features must be contiguous (NCDHW) or channels_last_3d')


# --------------------------------------------------------------------------------------------
def nms3d(dets, iou_threshold):
    """dets: [n,7] fp32 device tensor (any row stride) -> int64 tensor of kept ORIGINAL indices, ascending.
    (nms_cuda.nms_3d, mmdet/ops/nms/src/nms_cuda.cpp:16-21)"""
    if not dets.is_cuda:
        raise RuntimeError('dets must be a CUDAtensor ')          # CHECK_CUDA, nms_cuda.cpp:4
    if dets.dim() != 2 or dets.size(1) != 7:
        raise RuntimeError('nms_3d expects dets of shape [n, 7]')
    if dets.dtype != torch.float32:
        raise RuntimeError('nms_3d expects float32 dets')         # nms_kernel.cu:197,207
    n = dets.size(0)
    if n == 0:
        return torch.empty((0,), dtype=torch.long, device=dets.device)
    if dets.stride(1) != 1:
        dets = dets.contiguous()
    L = lib()
    with torch.cuda.device(dets.device):
        ws_bytes = int(L.mr3d_nms3d_workspace_bytes(n))
        ws = torch.empty((ws_bytes + 256,), dtype=torch.uint8, device=dets.device)
        off = (-ws.data_ptr()) % 256
        keep = torch.empty((n,), dtype=torch.long, device=dets.device)
        num = torch.empty((1,), dtype=torch.int32, device=dets.device)
        st = L.mr3d_nms3d(ptr(dets), ctypes.c_int(n), ctypes.c_int(dets.stride(0)), ctypes.c_float(iou_threshold),
                          ptr(keep), ptr(num), ctypes.c_void_p(ws.data_ptr() + off), ctypes.c_size_t(ws_bytes),
                          stream_ptr(dets.device))
        check(st, 'mr3d_nms3d')
        k = int(num.item())      # the only host sync: 4 bytes (the reference copies the whole bit-matrix)
    return keep[:k]


class RoiLevel(ctypes.Structure):
    """mr3d_roi_level (include/mr3d.h)"""
    _fields_ = [('features', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('depth', ctypes.c_int), ('height', ctypes.c_int),
                ('width', ctypes.c_int), ('spatial_scale', ctypes.c_float), ('spatial_scale_depth', ctypes.c_float)]


def _roi_levels(tensors, scales, scales_depth, backward, dtype=torch.float32):
    arr = (RoiLevel * len(tensors))()
    for i, (t, s, sd) in enumerate(zip(tensors, scales, scales_depth)):
        if not (t.is_cuda and t.dtype == dtype and t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d)):
            raise RuntimeError(f'multi-level RoIAlign3D needs {dtype} channels_last_3d feature maps')
        if backward:
            arr[i].grad = t.data_ptr()
        else:
            arr[i].features = t.data_ptr()
        arr[i].depth, arr[i].height, arr[i].width = t.size(2), t.size(3), t.size(4)
        arr[i].spatial_scale, arr[i].spatial_scale_depth = float(s), float(sd)
    return arr


def roi_align3d_forward_ml(feats, rois, out_d, out_h, out_w, scales, scales_depth, sample_num, finest_scale):
    """SingleRoIExtractor.forward in one launch (mr3d_roi_align3d_forward_ml); feats: logical [N,C,d,h,w] channels_last_3d"""
    rois = rois.contiguous()
    C = feats[0].size(1)
    out = torch.empty((rois.size(0), C, out_d, out_h, out_w), dtype=torch.float32, device=rois.device)
    bf16 = feats[0].dtype == torch.bfloat16            # configs[4]: bf16 pyramid, fp32 roi features
    lv = _roi_levels(feats, scales, scales_depth, False, feats[0].dtype)
    fn = lib().mr3d_roi_align3d_forward_ml_bf16 if bf16 else lib().mr3d_roi_align3d_forward_ml
    with torch.cuda.device(rois.device):
        st = fn(lv, len(feats), ctypes.c_float(finest_scale), ptr(rois), rois.size(0), C, out_d, out_h,
                                               out_w, int(sample_num), ptr(out), stream_ptr(rois.device))
    check(st, 'mr3d_roi_align3d_forward_ml')
    return out


def roi_align3d_backward_ml(top_grad, rois, grads, out_d, out_h, out_w, scales, scales_depth, sample_num, finest_scale,
                            bug_compat=True):
    """grads: PRE-ZEROED channels_last_3d buffers, one per level; accumulated in place"""
    top_grad, rois = top_grad.contiguous(), rois.contiguous()
    lv = _roi_levels(grads, scales, scales_depth, True)
    with torch.cuda.device(rois.device):
        st = lib().mr3d_roi_align3d_backward_ml(lv, len(grads), ctypes.c_float(finest_scale), ptr(top_grad), ptr(rois), rois.size(0),
                                                top_grad.size(1), out_d, out_h, out_w, int(sample_num),
                                                ROIALIGN_BUG_COMPAT if bug_compat else 0, stream_ptr(rois.device))
    check(st, 'mr3d_roi_align3d_backward_ml')


def nms3d_batched(dets_list, iou_threshold, real_rows=None):
    """independent NMS3D problems (e.g. the pyramid levels of one image) in ONE launch sequence and ONE 4*S-byte read-back.
    dets_list: [n_s, 7] fp32 device tensors -> list of int64 kept-index tensors (ascending, relative to their own set).
    Sets larger than 4096 boxes fall back to one mr3d_nms3d call each.
    real_rows: optional list (None or a 0-d device integer tensor per set): only the first real_rows[s] rows of set s are
    real, the rows behind them are fillers that score below every real row.  Fillers cannot suppress a real row (suppression
    flows from higher to lower scores), so the kept real rows are exactly those of an NMS over the real rows alone; kept
    fillers are cut from the returned list.  The counts travel in the same read-back."""
    S = len(dets_list)
    if S == 0:
        return []
    ns = [int(d.size(0)) for d in dets_list]
    if max(ns) > 4096 or S > 16:
        outs = [nms3d(d, iou_threshold) for d in dets_list]
        if real_rows is not None:
            outs = [o if r is None else o[o < r] for o, r in zip(outs, real_rows)]
        return outs
    dev = dets_list[0].device
    for d in dets_list:
        if not (d.is_cuda and d.dtype == torch.float32 and d.dim() == 2 and d.size(1) == 7):
            raise RuntimeError('nms_3d expects float32 CUDA dets of shape [n, 7]')
    total = sum(ns)
    if total == 0:
        return [torch.empty((0,), dtype=torch.long, device=dev) for _ in ns]
    dets = torch.cat([d for d in dets_list], 0).contiguous()
    seg = (ctypes.c_int * S)(*ns)
    L = lib()
    L.mr3d_nms3d_batched_workspace_bytes.restype = ctypes.c_size_t
    with torch.cuda.device(dev):
        ws_bytes = int(L.mr3d_nms3d_batched_workspace_bytes(seg, S))
        ws = torch.empty((ws_bytes + 256,), dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        keep = torch.empty((total,), dtype=torch.long, device=dev)
        num = torch.empty((S,), dtype=torch.int32, device=dev)
        st = L.mr3d_nms3d_batched(ptr(dets), seg, S, ctypes.c_int(dets.stride(0)), ctypes.c_float(iou_threshold), ptr(keep),
                                  ptr(num), ctypes.c_void_p(ws.data_ptr() + off), ctypes.c_size_t(ws_bytes), stream_ptr(dev))
        check(st, 'mr3d_nms3d_batched')
        if real_rows is not None and any(r is not None for r in real_rows):
            # kept indices are ascending, so the kept fillers are the tail of each list: count the kept REAL rows on the device
            row, cut = 0, []
            for s_, (n, r) in enumerate(zip(ns, real_rows)):
                if r is None or n == 0:
                    cut.append(num[s_])
                else:
                    live = torch.arange(n, device=dev) < num[s_]
                    cut.append(((keep[row:row + n] < r) & live).sum().to(torch.int32))
                row += n
            num = torch.stack(cut)
        counts = num.tolist()                    # one host sync for all segments
    out, row = [], 0
    for n, k in zip(ns, counts):
        out.append(keep[row:row + k])
        row += n
    return out


class RpnLevel(ctypes.Structure):
    """mr3d_rpn_level (include/mr3d.h)"""
    _fields_ = [('logits', ctypes.c_void_p), ('valid', ctypes.c_void_p)] + \
               [(n, ctypes.c_longlong) for n in ('stride_a', 'stride_d', 'stride_h', 'stride_w')] + \
               [(n, ctypes.c_int) for n in ('anchors', 'depth', 'height', 'width')]


def rpn_select_topk(cls_maps, valid_flags, nms_pre, counts_out=None):
    """mr3d_rpn_select_topk over all pyramid levels at once.  cls_maps[l]: [A, D, H, W] fp32 device logits (any strides);
    valid_flags[l]: uint8 / bool [A*D*H*W] in flat (h, w, d, a) order or None.  Returns (caps, flat int64 [sum caps], scores
    fp32 [sum caps], counts int32 [L] ON THE DEVICE): level l owns rows [sum caps[:l], +caps[l]), the first counts[l] real."""
    L = len(cls_maps)
    dev = cls_maps[0].device
    lv = (RpnLevel * L)()
    keep = []
    for l, c in enumerate(cls_maps):
        if not (c.is_cuda and c.dtype == torch.float32 and c.dim() == 4):
            raise RuntimeError('rpn_select_topk: cls maps must be float32 device tensors [A, D, H, W]')
        lv[l].logits = c.data_ptr()
        lv[l].stride_a, lv[l].stride_d, lv[l].stride_h, lv[l].stride_w = c.stride()
        lv[l].anchors, lv[l].depth, lv[l].height, lv[l].width = c.shape
        vf = None if valid_flags is None else valid_flags[l]
        if vf is not None and vf.numel() == c.numel():
            vf = vf.reshape(-1)
            vf = (vf if vf.dtype in (torch.uint8, torch.bool) else (vf != 0)).contiguous()
            keep.append(vf)
            lv[l].valid = vf.data_ptr()
    caps = [min(int(c.numel()), int(nms_pre)) for c in cls_maps]
    K = sum(caps)
    flat = torch.empty(K, dtype=torch.long, device=dev)
    scores = torch.empty(K, dtype=torch.float32, device=dev)
    # counts_out: a caller-owned int32 [L] slice (the dual-pathway detector keeps both pathways' counts in one tensor)
    counts = torch.empty(L, dtype=torch.int32, device=dev) if counts_out is None else counts_out
    assert counts.dtype == torch.int32 and counts.numel() == L and counts.is_contiguous()
    lib_ = lib()
    wsb = _sz(lib_.mr3d_rpn_select_topk_workspace_bytes, lv, L, int(nms_pre))
    if wsb == 0:
        raise Mr3dError('mr3d_rpn_select_topk: unsupported level table')
    ws_keep, ws = _aligned_bytes(wsb, dev)
    with torch.cuda.device(dev):
        st = lib_.mr3d_rpn_select_topk(lv, L, int(nms_pre), ptr(flat), ptr(scores), ptr(counts), ctypes.c_void_p(ws),
                                       ctypes.c_size_t(wsb), stream_ptr(dev))
    check(st, 'mr3d_rpn_select_topk')
    del ws_keep, keep
    return caps, flat, scores, counts


def nms3d_batched_dev(dets, caps, counts, iou_threshold):
    """mr3d_nms3d_batched_dev: set s = rows [sum caps[:s], +caps[s]) of dets [K, 7], of which the first counts[s] (device
    int32) are real.  Returns (keep int64 [K] -- per set, ascending, relative to the set -- and num_keep int32 [S]), both
    on the device: nothing is read back."""
    S = len(caps)
    dev = dets.device
    assert dets.is_contiguous() and dets.dtype == torch.float32 and dets.size(1) == 7 and dets.size(0) == sum(caps)
    seg = (ctypes.c_int * S)(*caps)
    L = lib()
    L.mr3d_nms3d_batched_workspace_bytes.restype = ctypes.c_size_t
    wsb = int(L.mr3d_nms3d_batched_workspace_bytes(seg, S))
    if wsb == 0:
        raise Mr3dError('mr3d_nms3d_batched_dev: sets of more than 4096 boxes / more than 16 sets')
    ws_keep, ws = _aligned_bytes(wsb, dev)
    keep = torch.empty(max(sum(caps), 1), dtype=torch.long, device=dev)
    num = torch.empty(S, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = L.mr3d_nms3d_batched_dev(ptr(dets), seg, ctypes.c_void_p(_opt(counts)), S, 7, ctypes.c_float(iou_threshold), ptr(keep),
                                      ptr(num), ctypes.c_void_p(ws), ctypes.c_size_t(wsb), stream_ptr(dev))
    check(st, 'mr3d_nms3d_batched_dev')
    del ws_keep
    return keep, num


def rpn_merge_levels(dets, caps, keep, num_keep, nms_post, max_num):
    """mr3d_rpn_merge_levels -> (proposals [max_num, 7], zero rows behind the count; count int32 [1] on the device)"""
    S = len(caps)
    seg = (ctypes.c_int * S)(*caps)
    out = torch.empty((int(max_num), 7), dtype=torch.float32, device=dets.device)
    count = torch.empty(1, dtype=torch.int32, device=dets.device)
    with torch.cuda.device(dets.device):
        st = lib().mr3d_rpn_merge_levels(ptr(dets), seg, S, ptr(keep), ptr(num_keep), int(nms_post), int(max_num), ptr(out),
                                         ptr(count), stream_ptr(dets.device))
    check(st, 'mr3d_rpn_merge_levels')
    return out, count


def multiclass_nms3d_fits(n, num_classes):
    """whether mr3d_multiclass_nms3d takes n rows x num_classes columns (its LDS-resident ranking: <= 4096 rows, <= 16 classes)"""
    return num_classes < 2 or _sz(lib().mr3d_multiclass_nms3d_workspace_bytes, int(n), int(num_classes)) != 0


def multiclass_nms3d(multi_bboxes, multi_scores, score_thr, iou_thr, max_num):
    """mr3d_multiclass_nms3d -> (dets [k,7], labels int64 [k]); one 4-byte read-back (the output's row count)"""
    for t, what in ((multi_bboxes, 'multi_bboxes'), (multi_scores, 'multi_scores')):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
            raise RuntimeError(f'{what} must be a 2-D float32 CUDAtensor ')
    b, sc = multi_bboxes.contiguous(), multi_scores.contiguous()
    n, C = sc.shape
    dev = sc.device
    if b.size(0) != n or b.size(1) not in (6, 6 * C):
        raise RuntimeError('multi_bboxes must be [n, 6] or [n, 6 * num_classes]')
    if n == 0 or C < 2:
        return b.new_zeros((0, 7)), torch.zeros((0,), dtype=torch.long, device=dev)
    L = lib()
    wsb = _sz(L.mr3d_multiclass_nms3d_workspace_bytes, n, C)
    if wsb == 0:
        raise Mr3dError(f'mr3d_multiclass_nms3d: {n} rows x {C - 1} classes exceeds the 4096-row / 16-class device limits')
    cap = (C - 1) * n if max_num < 0 else min((C - 1) * n, int(max_num))
    dets = torch.empty((max(cap, 1), 7), dtype=torch.float32, device=dev)
    labels = torch.empty((max(cap, 1),), dtype=torch.long, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((wsb + 256,), dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(dev):
        st = L.mr3d_multiclass_nms3d(ptr(b), int(b.size(1)), ptr(sc), n, C, ctypes.c_float(score_thr), ctypes.c_float(iou_thr),
                                     int(max_num), ptr(dets), ptr(labels), ptr(count), ctypes.c_void_p(ws.data_ptr() + off),
                                     ctypes.c_size_t(wsb), stream_ptr(dev))
        check(st, 'mr3d_multiclass_nms3d')
        k = int(count.item())
    return dets[:k], labels[:k]


_ROI_DTYPES = {torch.float32: 0, torch.float16: 2, torch.float64: 3}      # MR3D_DTYPE_F32 / _F16 / _F64


def roi_align3d_forward(features, rois, out_d, out_h, out_w, spatial_scale, spatial_scale_depth, sample_num, output):
    layout = feature_layout(features)
    N, C, D, H, W = features.shape
    if features.dtype != torch.float32:          # double / half: features, rois and output share the type (roi_align_kernel.cu:323)
        with torch.cuda.device(features.device):
            st = lib().mr3d_roi_align3d_forward_dtype(
                ptr(features), ptr(rois), ctypes.c_int(rois.size(0)), ctypes.c_int(N), ctypes.c_int(C), ctypes.c_int(D),
                ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(out_d), ctypes.c_int(out_h), ctypes.c_int(out_w),
                ctypes.c_float(spatial_scale), ctypes.c_float(spatial_scale_depth), ctypes.c_int(sample_num),
                ctypes.c_int(layout), ctypes.c_int(_ROI_DTYPES[features.dtype]), ptr(output), stream_ptr(features.device))
        check(st, 'mr3d_roi_align3d_forward_dtype')
        return
    with torch.cuda.device(features.device):
        st = lib().mr3d_roi_align3d_forward(
            ptr(features), ptr(rois), ctypes.c_int(rois.size(0)), ctypes.c_int(N), ctypes.c_int(C), ctypes.c_int(D),
            ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(out_d), ctypes.c_int(out_h), ctypes.c_int(out_w),
            ctypes.c_float(spatial_scale), ctypes.c_float(spatial_scale_depth), ctypes.c_int(sample_num),
            ctypes.c_int(layout), ptr(output), stream_ptr(features.device))
    check(st, 'mr3d_roi_align3d_forward')


def roi_align3d_backward(top_grad, rois, out_d, out_h, out_w, spatial_scale, spatial_scale_depth, sample_num,
                         bottom_grad, bug_compat=True):
    layout = feature_layout(bottom_grad)
    N, C, D, H, W = bottom_grad.shape
    if bottom_grad.dtype != torch.float32:
        with torch.cuda.device(bottom_grad.device):
            st = lib().mr3d_roi_align3d_backward_dtype(
                ptr(top_grad), ptr(rois), ctypes.c_int(rois.size(0)), ctypes.c_int(N), ctypes.c_int(C), ctypes.c_int(D),
                ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(out_d), ctypes.c_int(out_h), ctypes.c_int(out_w),
                ctypes.c_float(spatial_scale), ctypes.c_float(spatial_scale_depth), ctypes.c_int(sample_num),
                ctypes.c_int(layout), ctypes.c_int(ROIALIGN_BUG_COMPAT if bug_compat else 0),
                ctypes.c_int(_ROI_DTYPES[bottom_grad.dtype]), ptr(bottom_grad), stream_ptr(bottom_grad.device))
        check(st, 'mr3d_roi_align3d_backward_dtype')
        return
    with torch.cuda.device(bottom_grad.device):
        st = lib().mr3d_roi_align3d_backward(
            ptr(top_grad), ptr(rois), ctypes.c_int(rois.size(0)), ctypes.c_int(N), ctypes.c_int(C), ctypes.c_int(D),
            ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(out_d), ctypes.c_int(out_h), ctypes.c_int(out_w),
            ctypes.c_float(spatial_scale), ctypes.c_float(spatial_scale_depth), ctypes.c_int(sample_num),
            ctypes.c_int(layout), ctypes.c_int(ROIALIGN_BUG_COMPAT if bug_compat else 0), ptr(bottom_grad),
            stream_ptr(bottom_grad.device))
    check(st, 'mr3d_roi_align3d_backward')


# --------------------------------------------------------------------------------------------
class Conv3dDesc(ctypes.Structure):
    """mr3d_conv3d_desc (include/mr3d.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ('x', 'w', 'y', 'shift', 'add1', 'add2', 'up', 'mask')] + \
               [(n, ctypes.c_int) for n in ('n', 'd', 'h', 'w_in', 'cin', 'cout', 'kd', 'kh', 'kw', 'sd', 'sh', 'sw',
                                            'pd', 'ph', 'pw', 'up_d', 'up_h', 'up_w', 'relu',
                                            'ys_d', 'ys_h', 'ys_w', 'y_d', 'y_h', 'y_w')] + \
               [('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_longlong), ('x_live', ctypes.c_void_p)]


def _aligned_bytes(nbytes, device):
    """torch-owned device scratch of `nbytes`, 256-byte aligned: (tensor to keep alive, address)"""
    t = torch.empty((int(nbytes) + 256,), dtype=torch.uint8, device=device)
    return t, t.data_ptr() + (-t.data_ptr()) % 256


def wino2d_workspace_bytes(cin, cout):
    fn = lib().mr3d_conv3d_wino2d_workspace_bytes
    fn.restype = ctypes.c_size_t
    return int(fn(int(cin), int(cout)))


def _opt(t):
    return None if t is None else t.data_ptr()


def pack_conv_weight(w):
    """torch Conv3d weight [Cout, Cin, kd, kh, kw] -> [kd*kh*kw, Cin, Cout] contiguous (the layout the kernels read)."""
    co, ci = w.shape[:2]
    return w.reshape(co, ci, -1).permute(2, 1, 0).contiguous()


def conv3d_out_shape(x_shape, cout, k, stride, pad):
    n, d, h, w, _ = x_shape
    return (n, (d + 2 * pad[0] - k[0]) // stride[0] + 1, (h + 2 * pad[1] - k[1]) // stride[1] + 1,
            (w + 2 * pad[2] - k[2]) // stride[2] + 1, cout)


def voxel_liveness_words(positions):
    fn = lib().mr3d_voxel_liveness_words
    fn.restype = ctypes.c_longlong
    fn.argtypes = [ctypes.c_longlong]
    return int(fn(int(positions)))


def voxel_liveness(x, out=None):
    """int32 words of the liveness bitmap of the NDHWC map x (fp32 / bf16): bit p & 31 of word p >> 5 is set iff position p has a
    non-zero channel (mr3d_voxel_liveness) -- the `x_live` argument of conv3d_forward for data gradients of sparse maps"""
    assert x.is_cuda and x.is_contiguous() and x.dim() == 5 and x.dtype in (torch.float32, torch.bfloat16)
    n, d, h, w, c = x.shape
    words = voxel_liveness_words(n * d * h * w)
    bits = out if out is not None else torch.empty((words,), dtype=torch.int32, device=x.device)
    assert bits.numel() >= words and bits.dtype == torch.int32
    with torch.cuda.device(x.device):
        check(lib().mr3d_voxel_liveness(ptr(x), int(x.dtype == torch.bfloat16), n, d, h, w, c, ptr(bits), stream_ptr(x.device)),
              'mr3d_voxel_liveness')
    return bits


def conv3d_forward(x, w_packed, k, stride, pad, shift=None, add1=None, add2=None, up=None, mask=None, relu=False,
                   out=None, out_stride=None, out_offset=0, x_live=None):
    """x: [N,D,H,W,Cin] contiguous (NDHWC); w_packed: [taps,Cin,Cout]; returns y [N,Do,Ho,Wo,Cout].
    x_live: voxel_liveness(x) -- tiles whose input halo holds no live voxel are written from the epilogue alone."""
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 5
    assert w_packed.is_contiguous() and w_packed.dim() == 3 and w_packed.size(1) == x.size(4)
    cout = w_packed.size(2)
    oshape = conv3d_out_shape(x.shape, cout, k, stride, pad)
    y = out if out is not None else torch.empty(oshape, dtype=torch.float32, device=x.device)
    if out_stride is None:
        assert tuple(y.shape) == tuple(oshape)
    assert y.is_contiguous()
    for t in (add1, add2, mask):
        assert t is None or (tuple(t.shape) == tuple(y.shape) and t.is_contiguous())
    d = Conv3dDesc()
    # out_offset (elements): with out_stride, the scatter origin inside `out` (e.g. tap (a,b,c) of a stride-2 transposed conv)
    d.x, d.w, d.y = x.data_ptr(), w_packed.data_ptr(), y.data_ptr() + 4 * int(out_offset)
    d.shift, d.add1, d.add2, d.up, d.mask = _opt(shift), _opt(add1), _opt(add2), _opt(up), _opt(mask)
    d.n, d.d, d.h, d.w_in, d.cin, d.cout = x.size(0), x.size(1), x.size(2), x.size(3), x.size(4), cout
    d.kd, d.kh, d.kw = k
    d.sd, d.sh, d.sw = stride
    d.pd, d.ph, d.pw = pad
    if up is not None:
        assert up.is_contiguous() and up.size(4) == cout and up.size(0) == x.size(0)
        d.up_d, d.up_h, d.up_w = up.size(1), up.size(2), up.size(3)
    d.relu = int(bool(relu))
    if out_stride is not None:
        d.ys_d, d.ys_h, d.ys_w = out_stride
        d.y_d, d.y_h, d.y_w = y.size(1), y.size(2), y.size(3)
    keep = None
    if tuple(k) == (3, 3, 3) and tuple(stride) == (1, 1, 1):
        # scratch for the weight-transforming instances (two-dimensional Winograd): lent per call, from torch's caching
        # allocator (stream-ordered: the next user of the block runs behind this launch)
        nbytes = wino2d_workspace_bytes(d.cin, d.cout)
        keep, d.workspace = _aligned_bytes(nbytes, x.device)
        d.workspace_bytes = nbytes
    if x_live is not None:
        assert x_live.dtype == torch.int32 and x_live.is_contiguous() and x_live.numel() >= voxel_liveness_words(x.numel() // x.size(4))
        d.x_live = x_live.data_ptr()
    with torch.cuda.device(x.device):
        st = lib().mr3d_conv3d_forward(ctypes.byref(d), stream_ptr(x.device))
    check(st, 'mr3d_conv3d_forward')
    del keep
    return y


def pack_conv_weight_bf16(w):
    """torch Conv3d weight [Cout, Cin, kd, kh, kw] -> bf16 [taps, Cin16/8, Cout32, 8]: the B-operand layout of
    mr3d_conv3d_forward_bf16 (8 consecutive input channels of one output channel are one 16-byte fragment; channel counts
    rounded up to 16 / 32 with zeros).  The data-gradient operand is pack_conv_weight_bf16(w.flip(2, 3, 4).transpose(0, 1))."""
    co, ci = w.shape[:2]
    taps = w[0, 0].numel()
    ci16, co32 = -(-ci // 16) * 16, -(-co // 32) * 32
    buf = torch.zeros(taps, ci16, co32, dtype=torch.float32, device=w.device)
    buf[:, :ci, :co] = w.reshape(co, ci, taps).permute(2, 1, 0)
    return buf.view(taps, ci16 // 8, 8, co32).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)


def conv3d_forward_bf16(x, w16, cout, k, stride, pad, shift=None, add1=None, add2=None, up=None, mask=None, relu=False, x_live=None):
    """x: [N,D,H,W,Cin] bf16 (NDHWC); w16: pack_conv_weight_bf16 layout; returns y [N,Do,Ho,Wo,cout] bf16.
    x_live: a voxel_liveness bitmap that covers x's non-zero positions (3x3x3 / stride 1 / pad 1: dead tiles are their epilogue)."""
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.bfloat16 and x.dim() == 5 and w16.dtype == torch.bfloat16
    oshape = conv3d_out_shape(x.shape, cout, k, stride, pad)
    y = torch.empty(oshape, dtype=torch.bfloat16, device=x.device)
    for t in (add1, add2, mask):
        assert t is None or (tuple(t.shape) == tuple(y.shape) and t.is_contiguous() and t.dtype == torch.bfloat16)
    d = Conv3dDesc()
    d.x, d.w, d.y = x.data_ptr(), w16.data_ptr(), y.data_ptr()
    d.shift, d.add1, d.add2, d.up, d.mask = _opt(shift), _opt(add1), _opt(add2), _opt(up), _opt(mask)
    d.n, d.d, d.h, d.w_in, d.cin, d.cout = x.size(0), x.size(1), x.size(2), x.size(3), x.size(4), cout
    d.kd, d.kh, d.kw = k
    d.sd, d.sh, d.sw = stride
    d.pd, d.ph, d.pw = pad
    if up is not None:
        assert up.is_contiguous() and up.size(4) == cout and up.dtype == torch.bfloat16
        d.up_d, d.up_h, d.up_w = up.size(1), up.size(2), up.size(3)
    d.relu = int(bool(relu))
    if x_live is not None:
        assert x_live.dtype == torch.int32 and x_live.is_contiguous() and x_live.numel() >= voxel_liveness_words(x.numel() // x.size(4))
        d.x_live = x_live.data_ptr()
    with torch.cuda.device(x.device):
        st = lib().mr3d_conv3d_forward_bf16(ctypes.byref(d), stream_ptr(x.device))
    check(st, 'mr3d_conv3d_forward_bf16')
    return y


def conv3d_wgrad_bf16(x, dy, dw_packed, k, stride, pad, dbias=None, sparse_dy=False):
    """dw_packed fp32 [taps,Cin,Cout] += wgrad(x bf16 [N,D,H,W,Cin], dy bf16 [N,Do,Ho,Wo,Cout]); dw pre-zeroed.
    sparse_dy: dy is expected to be zero on most position tiles (stride-1 kernels test and skip them)."""
    assert x.is_cuda and x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype == torch.bfloat16
    assert dw_packed.dtype == torch.float32 and dw_packed.is_contiguous()
    cout = dw_packed.size(2)
    assert tuple(dy.shape) == tuple(conv3d_out_shape(x.shape, cout, k, stride, pad))
    d = Conv3dWgradDesc()
    d.x, d.dy, d.dw, d.dbias = x.data_ptr(), dy.data_ptr(), dw_packed.data_ptr(), _opt(dbias)
    d.n, d.d, d.h, d.w_in, d.cin, d.cout = x.size(0), x.size(1), x.size(2), x.size(3), x.size(4), cout
    d.kd, d.kh, d.kw = k
    d.sd, d.sh, d.sw = stride
    d.pd, d.ph, d.pw = pad
    d.flags = 1 if sparse_dy else 0                      # MR3D_WGRAD_SPARSE_DY
    with torch.cuda.device(x.device):
        st = lib().mr3d_conv3d_wgrad_bf16(ctypes.byref(d), stream_ptr(x.device))
    check(st, 'mr3d_conv3d_wgrad_bf16')
    return dw_packed


def conv3d_dgrad_s2k3_bf16(dy, wb16, dx_shape, add1=None, add2=None, mask=None, out=None):
    """bf16 form of conv3d_dgrad_s2k3 (mr3d_conv3d_dgrad_s2k3_bf16).  dy: [N,Do,Ho,Wo,Cout_fwd] bf16; wb16: pack_conv_weight_bf16 of
    the flipped / transposed taps (Cin = Cout_fwd, Cout = Cin_fwd); returns dx [N,D,H,W,Cin_fwd] bf16."""
    assert dy.is_cuda and dy.is_contiguous() and dy.dtype == torch.bfloat16 and dy.dim() == 5 and wb16.dtype == torch.bfloat16
    dx = out if out is not None else torch.empty(tuple(dx_shape), dtype=torch.bfloat16, device=dy.device)
    assert tuple(dx.shape) == tuple(dx_shape) and dx.is_contiguous() and dx.dtype == torch.bfloat16
    for t in (add1, add2, mask):
        assert t is None or (tuple(t.shape) == tuple(dx.shape) and t.is_contiguous() and t.dtype == torch.bfloat16)
    d = Conv3dDesc()
    d.x, d.w, d.y = dy.data_ptr(), wb16.data_ptr(), dx.data_ptr()
    d.add1, d.add2, d.mask = _opt(add1), _opt(add2), _opt(mask)
    d.n, d.d, d.h, d.w_in, d.cin, d.cout = dy.size(0), dy.size(1), dy.size(2), dy.size(3), dy.size(4), dx.size(4)
    d.kd = d.kh = d.kw = 3
    d.sd = d.sh = d.sw = 2
    d.pd = d.ph = d.pw = 1
    d.y_d, d.y_h, d.y_w = dx.size(1), dx.size(2), dx.size(3)
    with torch.cuda.device(dy.device):
        st = lib().mr3d_conv3d_dgrad_s2k3_bf16(ctypes.byref(d), stream_ptr(dy.device))
    check(st, 'mr3d_conv3d_dgrad_s2k3_bf16')
    return dx


def conv3d_dgrad_s2k3(dy, wb_packed, dx_shape, add1=None, add2=None, mask=None, out=None):
    """Data gradient of a kernel-3/stride-2/pad-1 conv without zero-stuffing (mr3d_conv3d_dgrad_s2k3).
    dy: [N,Do,Ho,Wo,Cout_fwd]; wb_packed: [27,Cout_fwd,Cin_fwd] flipped/transposed taps; returns dx [N,D,H,W,Cin_fwd]."""
    assert dy.is_cuda and dy.is_contiguous() and dy.dtype == torch.float32 and dy.dim() == 5
    assert wb_packed.is_contiguous() and wb_packed.size(0) == 27 and wb_packed.size(1) == dy.size(4)
    dx = out if out is not None else torch.empty(tuple(dx_shape), dtype=torch.float32, device=dy.device)
    assert tuple(dx.shape) == tuple(dx_shape) and dx.is_contiguous() and dx.size(4) == wb_packed.size(2)
    for t in (add1, add2, mask):
        assert t is None or (tuple(t.shape) == tuple(dx.shape) and t.is_contiguous())
    d = Conv3dDesc()
    d.x, d.w, d.y = dy.data_ptr(), wb_packed.data_ptr(), dx.data_ptr()
    d.add1, d.add2, d.mask = _opt(add1), _opt(add2), _opt(mask)
    d.n, d.d, d.h, d.w_in, d.cin, d.cout = dy.size(0), dy.size(1), dy.size(2), dy.size(3), dy.size(4), dx.size(4)
    d.kd = d.kh = d.kw = 3
    d.sd = d.sh = d.sw = 2
    d.pd = d.ph = d.pw = 1
    d.y_d, d.y_h, d.y_w = dx.size(1), dx.size(2), dx.size(3)
    with torch.cuda.device(dy.device):
        st = lib().mr3d_conv3d_dgrad_s2k3(ctypes.byref(d), stream_ptr(dy.device))
    check(st, 'mr3d_conv3d_dgrad_s2k3')
    return dx


class Conv3dWgradDesc(ctypes.Structure):
    """mr3d_conv3d_wgrad_desc (include/mr3d.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ('x', 'dy', 'dw', 'dbias')] + \
               [(n, ctypes.c_int) for n in ('n', 'd', 'h', 'w_in', 'cin', 'cout', 'kd', 'kh', 'kw', 'sd', 'sh', 'sw',
                                            'pd', 'ph', 'pw', 'flags')] + \
               [('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_longlong)]


WGRAD_WORKSPACE_BYTES = 64 << 20        # MR3D_WGRAD_WORKSPACE_BYTES
_wgrad_ws = {}


def device_cus(device):
    """compute units of the device (256 on MI355X): the persistent weight-gradient kernels launch two workgroups per CU"""
    return int(torch.cuda.get_device_properties(torch.device(device)).multi_processor_count)


def wgrad_workspace(device):
    """The partial-sum workspace lent to the weight-gradient kernels of PLAN records (one per device): all of them
    run in order on the plan executor's single side stream, so they can share it.  Direct calls on other streams must
    bring their own (or none: atomics)."""
    key = torch.device(device).index or 0
    if key not in _wgrad_ws:
        _wgrad_ws[key] = torch.empty(WGRAD_WORKSPACE_BYTES // 4, dtype=torch.float32, device=torch.device('cuda', key))
    return _wgrad_ws[key]


WGRAD_SPARSE_DY = 1      # mr3d_conv3d_wgrad_desc.flags / i[17] of a CONV_WGRAD record: MR3D_WGRAD_SPARSE_DY


def conv3d_wgrad(x, dy, dw_packed, k, stride, pad, dbias=None, workspace=None, sparse_dy=False):
    """dw_packed [taps,Cin,Cout] += wgrad(x [N,D,H,W,Cin], dy [N,Do,Ho,Wo,Cout]); dw must be pre-zeroed.
    workspace: optional fp32 tensor (>= WGRAD_WORKSPACE_BYTES) for the two-stage reduction; None = atomics.
    sparse_dy: dy is expected to be zero on most position tiles (tested and skipped by the large-map kernels)."""
    assert x.is_cuda and x.is_contiguous() and dy.is_contiguous() and dw_packed.is_contiguous()
    cout = dw_packed.size(2)
    assert tuple(dy.shape) == tuple(conv3d_out_shape(x.shape, cout, k, stride, pad))
    assert dw_packed.size(0) == k[0] * k[1] * k[2] and dw_packed.size(1) == x.size(4)
    d = Conv3dWgradDesc()
    d.x, d.dy, d.dw, d.dbias = x.data_ptr(), dy.data_ptr(), dw_packed.data_ptr(), _opt(dbias)
    d.n, d.d, d.h, d.w_in, d.cin, d.cout = x.size(0), x.size(1), x.size(2), x.size(3), x.size(4), cout
    d.kd, d.kh, d.kw = k
    d.sd, d.sh, d.sw = stride
    d.pd, d.ph, d.pw = pad
    d.flags = WGRAD_SPARSE_DY if sparse_dy else 0
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * 4
    with torch.cuda.device(x.device):
        st = lib().mr3d_conv3d_wgrad(ctypes.byref(d), stream_ptr(x.device))
    check(st, 'mr3d_conv3d_wgrad')
    return dw_packed


# --------------------------------------------------------------------------------------------
# plan records (mr3d_op / mr3d_layer_desc in include/mr3d.h)
OP_CONV_FWD, OP_CONV_WGRAD, OP_MAXPOOL_FWD, OP_MAXPOOL_BWD, OP_ZERO_STUFF2, OP_UPSAMPLE_ADJ = 1, 2, 3, 4, 5, 6
OP_SUBSAMPLE_FWD, OP_SUBSAMPLE_BWD, OP_VOLUME_PACK, OP_MEMSET0, OP_AXPY, OP_PREP_WEIGHTS, OP_FINISH_GRADS = 7, 8, 9, 10, 11, 12, 13
OP_STEM_FWD, OP_STEM_WGRAD, OP_CONV_DGRAD_S2, OP_PREP_WEIGHTS_BF16, OP_STEM_PACK_BF16, OP_SYNC, OP_WINO2D_PREP, OP_PW_PAIR = 14, 15, 16, 17, 18, 19, 20, 21
OP_VOXEL_LIVENESS = 22      # p = {x, bits}, i = {n,d,h,w,c, arm}: arm = 1 hands the bitmap to the next CONV_FWD record as its x_live
SYNC_JOIN, SYNC_MARK, SYNC_WAIT = 0, 1, 2      # mr3d_op.i[0] of an OP_SYNC record
DTYPE_BF16 = 1          # mr3d_op.i[25]: MR3D_DTYPE_BF16
CONV_RELU, CONV_UP_EVEN = 1, 2      # mr3d_conv3d_desc.relu flags (i[18] of a CONV_FWD record)


class Op(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('i', ctypes.c_int32 * 27), ('f', ctypes.c_float * 4), ('p', ctypes.c_void_p * 8)]


class LayerDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_longlong) for n in ('w_off', 'gamma_off', 'beta_off', 'mean_off', 'var_off', 'bias_off',
                                                 'wf_off', 'wb_off', 'shift_off', 'scale_off', 'dwraw_off', 'dbias_off')] + \
               [('co', ctypes.c_int), ('ci', ctypes.c_int), ('ci_pad', ctypes.c_int), ('taps', ctypes.c_int),
                ('eps', ctypes.c_float), ('reserved', ctypes.c_int)]


def make_op(kind, ptrs=(), ints=(), alpha=0.0, side=False):
    """side: the record may run on the plan's side stream (mr3d_op.i[26], see mr3d_plan_set_side_stream)"""
    o = Op()
    o.kind = kind
    if side:
        o.i[26] = 1
    for k, v in enumerate(ptrs):
        o.p[k] = None if v is None else (v.data_ptr() if hasattr(v, 'data_ptr') else int(v))
    for k, v in enumerate(ints):
        o.i[k] = int(v)
    o.f[0] = float(alpha)
    return o


def pw_pair_supported(c0, c1, c2, positions):
    fn = lib().mr3d_pointwise_pair_supported
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong]
    return fn(int(c0), int(c1), int(c2), int(positions)) == 1


def make_pw_pair_op(x, w1, shift1, add1, mask1, y1, w2, shift2, mask2, y2, c0, c1, c2, relu1, relu2, positions):
    """PW_PAIR plan record (mr3d_pointwise_pair): tensors or raw addresses; the shift pointers travel in the integer block"""
    addr = lambda v: 0 if v is None else (v.data_ptr() if hasattr(v, 'data_ptr') else int(v))
    ints = [c0, c1, c2, int(bool(relu1)), int(bool(relu2)), *split64(positions), *split64(addr(shift1)), *split64(addr(shift2))]
    return make_op(OP_PW_PAIR, [x, w1, add1, mask1, y1, w2, mask2, y2], ints)


def plan_stream_wait_mark(slot, stream):
    """mr3d_plan_stream_wait_mark: torch stream `stream` waits for SYNC mark `slot` of the last plan run on its device"""
    with torch.cuda.device(stream.device):
        check(lib().mr3d_plan_stream_wait_mark(int(slot), ctypes.c_void_p(stream.cuda_stream)), 'mr3d_plan_stream_wait_mark')


def plan_stream_wait_mark_checked(slot, stream):
    """mr3d_plan_stream_wait_mark_checked: True when `stream` now waits for SYNC mark `slot` as recorded by a live plan run since
    the last checked wait; False (no wait was issued) when the mark was not recorded -- side stream off, captured plan, no run"""
    waited = ctypes.c_int(0)
    with torch.cuda.device(stream.device):
        check(lib().mr3d_plan_stream_wait_mark_checked(int(slot), ctypes.c_void_p(stream.cuda_stream), ctypes.byref(waited)),
              'mr3d_plan_stream_wait_mark_checked')
    return bool(waited.value)


def split64(v):
    v = int(v)
    lo, hi = v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF
    return (lo - (1 << 32) if lo >= (1 << 31) else lo), (hi - (1 << 32) if hi >= (1 << 31) else hi)


class EventPair(ctypes.Structure):
    """mr3d_event_pair (include/mr3d.h)"""
    _fields_ = [('start', ctypes.c_void_p), ('stop', ctypes.c_void_p)]


class _PairOwner:
    def __init__(self, pair):
        self.pair = EventPair(pair.start, pair.stop)

    def __del__(self):
        if _lib is not None and self.pair.start:
            _lib.mr3d_event_pair_destroy(ctypes.byref(self.pair))


class HipEvent:
    """one end of a library-created event pair; mirrors torch.cuda.Event.elapsed_time for the pair's two ends"""

    def __init__(self, owner, is_start):
        self.owner, self.is_start = owner, is_start

    def elapsed_time(self, other):
        assert self.is_start and other.owner is self.owner and not other.is_start
        ms = ctypes.c_float()
        check(lib().mr3d_event_pair_elapsed_ms(ctypes.byref(self.owner.pair), ctypes.byref(ms)), 'mr3d_event_pair_elapsed_ms')
        return float(ms.value)


class Plan:
    """A list of mr3d_op records executed by one mr3d_plan_run call on the current stream."""

    _variant_cache = {}

    def __init__(self, ops):
        ops = self._with_wino2d_prep(list(ops))
        self.n = len(ops)
        self.array = (Op * self.n)(*ops)
        self._failed = ctypes.c_int(-1)

    def _with_wino2d_prep(self, ops):
        """The two-dimensional Winograd kernels read pre-transformed weights.  The library allocates nothing: the plan owns
        ONE scratch buffer for the transformed weights of all its qualifying convolutions and a WINO2D_PREP record (inserted
        in front of the first of them in every stretch between weight-writing records) fills it in one launch.  The buffer
        lives as long as the plan -- and therefore as long as any hipGraph captured from it."""
        out, self._u_ws = [], []
        k = 0
        while k < len(ops):
            # one stretch: [k, e) up to (not including) the next weight-writing record behind a conv
            e, first, need, seen = k, None, 0, set()
            while e < len(ops) and not (ops[e].kind in (OP_PREP_WEIGHTS, OP_PREP_WEIGHTS_BF16) and first is not None):
                o = ops[e]
                if o.kind == OP_CONV_FWD and o.i[25] != DTYPE_BF16 and o.i[6] == 3 and o.p[1] not in seen:
                    key = tuple(o.i[0:25])
                    if key not in Plan._variant_cache:
                        Plan._variant_cache[key] = op_conv_variant(o).winograd
                    if Plan._variant_cache[key] == 8:
                        first = e if first is None else first
                        seen.add(o.p[1])
                        need += wino2d_workspace_bytes(o.i[4], o.i[5])
                e += 1
            if first is not None:
                dev = torch.device('cuda', torch.cuda.current_device())
                keep, addr = _aligned_bytes(need, dev)
                self._u_ws.append(keep)
                out += ops[k:first] + [make_op(OP_WINO2D_PREP, [addr], split64(need))] + ops[first:e]
            else:
                out += ops[k:e]
            k = e
        return out

    def _run_range(self, lo, hi, device):
        if hi <= lo:
            return
        if hi == lo + 1 and self.array[lo].kind == OP_CONV_FWD and self._u_ws:
            # a conv record run on its own (per-op timing): the transformed weights only live during the run that prepared
            # them, so the nearest WINO2D_PREP record in front of it rides along (it prepares exactly this conv)
            prep = next((j for j in range(lo - 1, -1, -1) if self.array[j].kind == OP_WINO2D_PREP), None)
            if prep is not None:
                pair = (Op * 2)(self.array[prep], self.array[lo])
                st = lib().mr3d_plan_run(pair, ctypes.c_int(2), ctypes.byref(self._failed), stream_ptr(device))
                if st != OK:
                    raise Mr3dError(f'mr3d_plan_run: op #{lo} failed: {lib().mr3d_status_string(int(st)).decode()}')
                return
        base = ctypes.cast(ctypes.byref(self.array, lo * ctypes.sizeof(Op)), ctypes.POINTER(Op))
        st = lib().mr3d_plan_run(base, ctypes.c_int(hi - lo), ctypes.byref(self._failed), stream_ptr(device))
        if st != OK:
            k = lo + self._failed.value
            raise Mr3dError(f'mr3d_plan_run: op #{k} (kind {self.array[k].kind}) failed: '
                            f'{lib().mr3d_status_string(int(st)).decode()}')

    def find_ptr(self, address):
        """[(op index, pointer slot)] of every reference to a device address -- the places to re-point when a caller's
        tensor is bound in place of a plan-owned staging buffer"""
        return [(k, j) for k in range(self.n) for j in range(8) if self.array[k].p[j] == address]

    def repoint(self, where, address):
        assert getattr(self, '_graph', None) is None, 'a captured plan has its addresses baked in'
        for k, j in where:
            self.array[k].p[j] = address

    def capture(self):
        """hipGraph of the whole plan (mr3d_plan_capture); later run() calls without marks replay it"""
        g = ctypes.c_void_p()
        check(lib().mr3d_plan_capture(self.array, self.n, ctypes.byref(g)), 'mr3d_plan_capture')
        self._graph = g
        return self

    def __del__(self):
        g = getattr(self, '_graph', None)
        if g is not None and _lib is not None:
            _lib.mr3d_graph_destroy(g)

    def run_marked(self, marks, device=None):
        """The whole plan in ONE mr3d_plan_run_marked call -- the production launch sequence, side stream and all -- with HIP
        timing events around the records `marks`, each on the stream its record really runs on.  Returns {index: (start,
        stop)} of HipEvent objects; `start.elapsed_time(stop)` is valid once the device has been synchronised."""
        marks = sorted(marks)
        arr = (ctypes.c_int * len(marks))(*marks)
        pairs = (EventPair * len(marks))()
        st = lib().mr3d_plan_run_marked(self.array, self.n, ctypes.byref(self._failed), stream_ptr(device), arr, len(marks), pairs)
        if st != OK:
            raise Mr3dError(f'mr3d_plan_run_marked: op #{self._failed.value} failed: {lib().mr3d_status_string(int(st)).decode()}')
        out = {}
        for k, m in enumerate(marks):
            owner = _PairOwner(pairs[k])
            out[m] = (HipEvent(owner, True), HipEvent(owner, False))
        return out

    def run(self, device=None, marks=None):
        """Enqueue every op on torch's current stream.  marks: optional sorted op indices to bracket with
        torch.cuda.Event pairs (same stream as the kernels); returns {index: (start_event, end_event)}."""
        if not marks:
            if getattr(self, '_graph', None) is not None:
                check(lib().mr3d_graph_launch(self._graph, stream_ptr(device)), 'mr3d_graph_launch')
            else:
                self._run_range(0, self.n, device)
            return {}
        ev, pos = {}, 0
        for k in marks:
            self._run_range(pos, k, device)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self._run_range(k, k + 1, device)
            e.record()
            ev[k] = (s, e)
            pos = k + 1
        self._run_range(pos, self.n, device)
        return ev


class Conv3dVariant(ctypes.Structure):
    """mr3d_conv3d_variant (include/mr3d.h)"""
    _fields_ = [(n, ctypes.c_int) for n in ('streaming_pointwise', 'winograd', 'tile_d', 'tile_h', 'tile_w', 'cin_chunk',
                                            'cout_tile', 'parity_decomposed')] + [('mfma_issue_factor', ctypes.c_float)]


def op_conv_variant(op):
    """the kernel instance a CONV_FWD plan record dispatches to (mr3d_conv3d_forward_variant; nothing is launched)"""
    assert op.kind == OP_CONV_FWD
    d = Conv3dDesc()
    for k, name in enumerate(('x', 'w', 'y', 'shift', 'add1', 'add2', 'up', 'mask')):
        setattr(d, name, op.p[k])
    ctypes.memmove(ctypes.addressof(d) + Conv3dDesc.n.offset, ctypes.addressof(op.i), 25 * 4)
    d.workspace_bytes = -1                 # a plan lends the scratch of the weight-transforming instances (WINO2D_PREP)
    v = Conv3dVariant()
    check(lib().mr3d_conv3d_forward_variant(ctypes.byref(d), ctypes.byref(v)), 'mr3d_conv3d_forward_variant')
    return v


def op_wgrad_issue_factor(op):
    """MFMA FLOPs a CONV_WGRAD plan record issues / its algorithmic FLOPs (mr3d_conv3d_wgrad_issue_factor: 1 direct, 2/3 F(3,2)
    along W, 1/3 the two-dimensional Winograd form; nothing is launched).  bf16 records: direct form (1)."""
    assert op.kind == OP_CONV_WGRAD
    if op.i[25] == DTYPE_BF16:
        return 1.0
    d = Conv3dWgradDesc()
    d.x, d.dy, d.dw, d.dbias = op.p[0], op.p[1], op.p[2], op.p[3]
    ctypes.memmove(ctypes.addressof(d) + Conv3dWgradDesc.n.offset, ctypes.addressof(op.i), 15 * 4)
    d.flags = op.i[17]
    d.workspace, d.workspace_bytes = op.p[4], (op.i[15] & 0xFFFFFFFF) | (op.i[16] << 32) if op.p[4] else 0
    fn = lib().mr3d_conv3d_wgrad_issue_factor
    fn.restype = ctypes.c_float
    return float(fn(ctypes.byref(d)))


def _narrow_nb(n, do, ho, wo, cout, nb):
    """mirror of narrow_nb() in csrc/conv3d_fwd.hip: the Cout tile the forward/dgrad dispatch really picks"""
    def wgs(nb_):
        th = 8 if nb_ == 32 else 4
        return n * ((do + 1) // 2) * (-(-ho // th)) * (-(-wo // 8)) * (cout // nb_)
    while nb > 16 and wgs(nb) < 256:
        nb //= 2
    return nb


def op_kernel_key(op):
    """identity of the device kernel an op launches (template instance for the conv kernels) + its
    algorithmic FLOPs (2*MAC over the REAL channels; the 3-channel stem is stored padded to 4)"""
    i = op.i
    if op.kind in (OP_CONV_FWD, OP_CONV_WGRAD):
        n, d, h, w, cin, cout, k = i[0], i[1], i[2], i[3], i[4], i[5], i[6]
        sd, sh, sw, pd = i[9], i[10], i[11], i[12]
        do, ho, wo = (d + 2 * pd - k) // sd + 1, (h + 2 * pd - k) // sh + 1, (w + 2 * pd - k) // sw + 1
        nb = 64 if cout % 64 == 0 else (32 if cout % 32 == 0 else 16)
        if op.kind == OP_CONV_FWD:
            nb = _narrow_nb(n, do, ho, wo, cout, nb)
        if k == 1:
            cc = 64 if cin % 64 == 0 else (32 if cin % 32 == 0 else 16)
        else:
            cc = 4 if cin == 4 else (8 if (k == 3 and sd == 2 and op.kind == OP_CONV_FWD) else 16)
        real_cin = 3 if (cin == 4 and k == 7) else cin
        flops = 2.0 * n * do * ho * wo * cout * real_cin * k ** 3
        name = 'conv3d_fwd' if op.kind == OP_CONV_FWD else 'conv3d_wgrad'
        if i[25] == DTYPE_BF16:
            return f'{name}_bf16<k{k},s{sd}{sh}{sw},cin{cin},cout{cout}>', flops
        # the Winograd instances are kernels of their own (conv3d_wino2d_wide_kernel / conv3d_wino2d_kernel / the F(4,3)-along-W
        # rows kernels): name them, so that a group of the measurement harness is ONE kernel (round 5 mixed the wide form, 0.63 of
        # the peak, and the 2 x 8 x 8 form, 0.23, under one key).  A data-gradient launch that arms a liveness bitmap runs the
        # 2 x 8 x 8 form whatever the probe says (sparse_pyramid_grads; off in bench.py).
        suffix = ''
        if op.kind == OP_CONV_FWD and k == 3 and (sd, sh, sw) == (1, 1, 1):
            try:
                v = op_conv_variant(op)
                if v.winograd == 8:
                    suffix = ',wino2d-wide' if v.tile_d == 4 else ',wino2d'
                elif v.winograd:
                    suffix = f',winoW{v.winograd}'
            except Exception:
                pass
        return f'{name}<k{k},s{sd}{sh}{sw},cc{cc},nb{nb}{suffix}>', flops
    if op.kind == OP_PW_PAIR:
        c0, c1, c2 = i[0], i[1], i[2]
        pos = (i[5] & 0xFFFFFFFF) | (i[6] << 32)
        return f'pw_pair<{c0},{c1},{c2}>', 2.0 * pos * (c0 * c1 + c1 * c2)
    if op.kind == OP_CONV_DGRAD_S2:      # i[1:4] = dy's (coarse) grid; every dy element meets each of the 27 taps once
        n, d, h, w, cin, cout = i[0], i[1], i[2], i[3], i[4], i[5]
        nb = _narrow_nb(n, d, h, w, cout, 64 if cout % 64 == 0 else (32 if cout % 32 == 0 else 16))
        return f'conv3d_dgrad_s2k3<cc16,nb{nb}>', 2.0 * n * d * h * w * cin * cout * 27
    if op.kind in (OP_STEM_FWD, OP_STEM_WGRAD):
        n, d, h, w = i[0], i[1], i[2], i[3]
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        flops = 2.0 * n * d * ho * wo * 16 * 3 * 343
        if op.kind == OP_STEM_WGRAD and i[7] == 1 and i[25] != DTYPE_BF16 and w % 2 == 0 and not lib().mr3d_tuning_get(3):
            # the gather over the pooled elements (stem_wgrad_gather.hip): one multiply-add row of 1029 per POOLED element -- the
            # zeros of the max-pool adjoint (>= 7/8 of dy) are not multiplied; vector-ALU / LDS work, no MFMA
            return 'stem_wgrad_gather', 2.0 * n * ((d - 1) // 2 + 1) * ((ho - 1) // 2 + 1) * ((wo - 1) // 2 + 1) * 16 * 3 * 343
        return ('stem_fwd' if op.kind == OP_STEM_FWD else 'stem_wgrad'), flops
    names = {OP_MAXPOOL_FWD: 'maxpool_fwd', OP_MAXPOOL_BWD: 'maxpool_bwd', OP_ZERO_STUFF2: 'zero_stuff', OP_UPSAMPLE_ADJ:
             'upsample_adjoint_add', OP_SUBSAMPLE_FWD: 'subsample2_fwd', OP_SUBSAMPLE_BWD: 'subsample2_bwd', OP_VOLUME_PACK:
             'ncdhw3_to_ndhwc4', OP_MEMSET0: 'memset', OP_AXPY: 'axpy', OP_PREP_WEIGHTS: 'prep_weights', OP_FINISH_GRADS:
             'finish_grads'}
    return names.get(op.kind, f'op{op.kind}'), 0.0


def stem_conv_forward(x_ncdhw, w_packed, shift=None, relu=False):
    """x [N,3,D,H,W] planar fp32, w_packed [343,4,16] -> y [N,D,Ho,Wo,16] (mr3d_stem_conv_forward)"""
    assert x_ncdhw.is_cuda and x_ncdhw.is_contiguous() and x_ncdhw.size(1) == 3 and tuple(w_packed.shape) == (343, 4, 16)
    n, _, d, h, w = x_ncdhw.shape
    y = torch.empty((n, d, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 16), dtype=torch.float32, device=x_ncdhw.device)
    with torch.cuda.device(x_ncdhw.device):
        st = lib().mr3d_stem_conv_forward(ptr(x_ncdhw), ptr(w_packed), ctypes.c_void_p(_opt(shift)), ptr(y), n, d, h, w,
                                          int(bool(relu)), stream_ptr(x_ncdhw.device))
    check(st, 'mr3d_stem_conv_forward')
    return y


def stem_conv_forward_bf16(x_ncdhw, w_packed, shift=None, relu=False):
    """bf16-MFMA stem: x [N,3,D,H,W] fp32, w_packed fp32 [343,4,16] (packed to bf16 here) -> y bf16 [N,D,Ho,Wo,16]"""
    assert x_ncdhw.is_cuda and x_ncdhw.is_contiguous() and x_ncdhw.size(1) == 3 and tuple(w_packed.shape) == (343, 4, 16)
    n, _, d, h, w = x_ncdhw.shape
    w16 = torch.empty(49 * 4 * 16 * 8, dtype=torch.bfloat16, device=x_ncdhw.device)
    y = torch.empty((n, d, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 16), dtype=torch.bfloat16, device=x_ncdhw.device)
    with torch.cuda.device(x_ncdhw.device):
        check(lib().mr3d_stem_pack_weights_bf16(ptr(w_packed), ptr(w16), stream_ptr(x_ncdhw.device)), 'mr3d_stem_pack_weights_bf16')
        st = lib().mr3d_stem_conv_forward_bf16(ptr(x_ncdhw), ptr(w16), ctypes.c_void_p(_opt(shift)), ptr(y), n, d, h, w,
                                               int(bool(relu)), stream_ptr(x_ncdhw.device))
    check(st, 'mr3d_stem_conv_forward_bf16')
    return y


def stem_conv_wgrad_bf16(x_ncdhw, dy, dw_packed, dbias=None):
    n, _, d, h, w = x_ncdhw.shape
    assert dy.is_contiguous() and dy.dtype == torch.bfloat16 and tuple(dw_packed.shape) == (343, 4, 16)
    with torch.cuda.device(x_ncdhw.device):
        st = lib().mr3d_stem_conv_wgrad_bf16(ptr(x_ncdhw), ptr(dy), ptr(dw_packed), ctypes.c_void_p(_opt(dbias)), n, d, h, w,
                                             stream_ptr(x_ncdhw.device))
    check(st, 'mr3d_stem_conv_wgrad_bf16')
    return dw_packed


def stem_conv_wgrad_bf16_pooled(x_ncdhw, g_pooled, argmax, pooled, dw_packed, dbias=None):
    """bf16 form of stem_conv_wgrad_pooled (mr3d_stem_conv_wgrad_bf16_pooled)"""
    n, _, d, h, w = x_ncdhw.shape
    assert g_pooled.is_contiguous() and g_pooled.dtype == torch.bfloat16 and tuple(dw_packed.shape) == (343, 4, 16)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    assert tuple(g_pooled.shape) == (n, (d - 1) // 2 + 1, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1, 16)
    assert argmax.is_contiguous() and argmax.numel() * 4 == g_pooled.numel()
    assert pooled is None or (pooled.is_contiguous() and pooled.shape == g_pooled.shape and pooled.dtype == torch.bfloat16)
    with torch.cuda.device(x_ncdhw.device):
        st = lib().mr3d_stem_conv_wgrad_bf16_pooled(ptr(x_ncdhw), ptr(g_pooled), ptr(argmax), ctypes.c_void_p(_opt(pooled)),
                                                    ptr(dw_packed), ctypes.c_void_p(_opt(dbias)), n, d, h, w,
                                                    stream_ptr(x_ncdhw.device))
    check(st, 'mr3d_stem_conv_wgrad_bf16_pooled')
    return dw_packed


def stem_conv_wgrad(x_ncdhw, dy, dw_packed, dbias=None, workspace=None):
    n, _, d, h, w = x_ncdhw.shape
    assert dy.is_contiguous() and tuple(dw_packed.shape) == (343, 4, 16)
    with torch.cuda.device(x_ncdhw.device):
        st = lib().mr3d_stem_conv_wgrad_ws(ptr(x_ncdhw), ptr(dy), ptr(dw_packed), ctypes.c_void_p(_opt(dbias)), n, d, h, w,
                                           ctypes.c_void_p(_opt(workspace)),
                                           ctypes.c_longlong(0 if workspace is None else workspace.numel() * 4),
                                           stream_ptr(x_ncdhw.device))
    check(st, 'mr3d_stem_conv_wgrad')
    return dw_packed


def stem_conv_wgrad_pooled(x_ncdhw, g_pooled, argmax, pooled, dw_packed, dbias=None, workspace=None):
    """stem weight gradient straight from the gradient of the max-pooled map (mr3d_stem_conv_wgrad_pooled): the max-pool
    backward (resnet3d.py:471) and the stem's ReLU mask (pooled > 0; None: no mask) happen in the kernel's dy staging"""
    n, _, d, h, w = x_ncdhw.shape
    assert g_pooled.is_contiguous() and g_pooled.dtype == torch.float32 and tuple(dw_packed.shape) == (343, 4, 16)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    assert tuple(g_pooled.shape) == (n, (d - 1) // 2 + 1, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1, 16)
    assert argmax.is_contiguous() and argmax.numel() * 4 == g_pooled.numel()
    assert pooled is None or (pooled.is_contiguous() and pooled.shape == g_pooled.shape and pooled.dtype == torch.float32)
    with torch.cuda.device(x_ncdhw.device):
        st = lib().mr3d_stem_conv_wgrad_pooled(ptr(x_ncdhw), ptr(g_pooled), ptr(argmax), ctypes.c_void_p(_opt(pooled)),
                                               ptr(dw_packed), ctypes.c_void_p(_opt(dbias)), n, d, h, w,
                                               ctypes.c_void_p(_opt(workspace)),
                                               ctypes.c_longlong(0 if workspace is None else workspace.numel() * 4),
                                               stream_ptr(x_ncdhw.device))
    check(st, 'mr3d_stem_conv_wgrad_pooled')
    return dw_packed


# --------------------------------------------------------------------------------------------
def _f6(v, default):
    v = list(default if v is None else v)
    assert len(v) == 6
    return (ctypes.c_float * 6)(*[float(x) for x in v])


def _hwd(max_shape):
    """max_shape is the reference's img_shape tuple (H, W, 3, D) (coco_3d_2scales.py:253-255) -> int[3] {H, W, D}"""
    if max_shape is None:
        return None
    return (ctypes.c_int * 3)(int(max_shape[0]), int(max_shape[1]), int(max_shape[3]))


def _dev_f32(t, what):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f'{what} must be a float32 CUDAtensor ')
    return t.contiguous()


def delta2bbox3d(rois, deltas, means, stds, max_shape, wh_ratio_clip):
    rois, deltas = _dev_f32(rois, 'rois'), _dev_f32(deltas, 'deltas')
    n, k6 = deltas.shape
    out = torch.empty_like(deltas)
    with torch.cuda.device(rois.device):
        st = lib().mr3d_delta2bbox3d(ptr(rois), ptr(deltas), n, k6 // 6, _f6(means, [0.] * 6), _f6(stds, [1.] * 6),
                                     _hwd(max_shape), ctypes.c_double(wh_ratio_clip), ptr(out), stream_ptr(rois.device))
    check(st, 'mr3d_delta2bbox3d')
    return out


def bbox2delta3d(proposals, gt, means, stds):
    proposals, gt = _dev_f32(proposals, 'proposals'), _dev_f32(gt, 'gt')
    out = torch.empty_like(proposals)
    with torch.cuda.device(proposals.device):
        st = lib().mr3d_bbox2delta3d(ptr(proposals), ptr(gt), proposals.size(0), _f6(means, [0.] * 6), _f6(stds, [1.] * 6),
                                     ptr(out), stream_ptr(proposals.device))
    check(st, 'mr3d_bbox2delta3d')
    return out


def bbox_overlaps3d(b1, b2):
    b1, b2 = _dev_f32(b1, 'bboxes1'), _dev_f32(b2, 'bboxes2')
    out = torch.empty((b1.size(0), b2.size(0)), dtype=torch.float32, device=b1.device)
    with torch.cuda.device(b1.device):
        st = lib().mr3d_bbox_overlaps3d(ptr(b1), ptr(b2), b1.size(0), b2.size(0), ptr(out), stream_ptr(b1.device))
    check(st, 'mr3d_bbox_overlaps3d')
    return out


def grid_anchors3d(base_anchors, featmap_size, stride, depth_stride):
    base = _dev_f32(base_anchors, 'base_anchors')
    fd, fh, fw = (int(v) for v in featmap_size)
    out = torch.empty((fd * fh * fw * base.size(0), 6), dtype=torch.float32, device=base.device)
    with torch.cuda.device(base.device):
        st = lib().mr3d_grid_anchors3d(ptr(base), base.size(0), fd, fh, fw, int(stride), int(depth_stride), ptr(out),
                                       stream_ptr(base.device))
    check(st, 'mr3d_grid_anchors3d')
    return out


def rpn_decode_level(flat_idx, scores, reg_map, reg_layout, base_anchors, featmap_size, stride, depth_stride, means, stds,
                     max_shape, wh_ratio_clip=16 / 1000, out=None):
    k = int(flat_idx.numel())
    dets = out if out is not None else torch.empty((k, 7), dtype=torch.float32, device=scores.device)
    assert dets.is_contiguous() and tuple(dets.shape) == (k, 7)
    if k == 0:
        return dets
    assert flat_idx.dtype == torch.long and flat_idx.is_contiguous() and scores.is_contiguous() and reg_map.is_contiguous()
    fd, fh, fw = (int(v) for v in featmap_size)
    with torch.cuda.device(scores.device):
        st = lib().mr3d_rpn_decode_level(ptr(flat_idx), ptr(scores), k, ptr(reg_map), int(reg_layout), ptr(base_anchors),
                                         base_anchors.size(0), fd, fh, fw, int(stride), int(depth_stride),
                                         _f6(means, [0.] * 6), _f6(stds, [1.] * 6), _hwd(max_shape),
                                         ctypes.c_double(wh_ratio_clip), ptr(dets), stream_ptr(scores.device))
    check(st, 'mr3d_rpn_decode_level')
    return dets


def relu_mask(g, act, want_bias_grad=False):
    """out = act > 0 ? g : 0 for contiguous [.., C] tensors (mr3d_relu_mask); optionally also sum(out) per channel"""
    assert g.is_cuda and g.is_contiguous() and act.is_contiguous() and g.shape == act.shape and g.dtype == torch.float32
    c = g.shape[-1]
    out = torch.empty_like(g)
    db = torch.zeros(c, dtype=torch.float32, device=g.device) if want_bias_grad else None
    with torch.cuda.device(g.device):
        st = lib().mr3d_relu_mask(ptr(g), ptr(act), ptr(out), ctypes.c_longlong(g.numel() // c), c, ctypes.c_void_p(_opt(db)),
                                  stream_ptr(g.device))
    check(st, 'mr3d_relu_mask')
    return (out, db) if want_bias_grad else out


# -------------------------------------------------------------------------------------------- training targets
def _sz(fn, *args):
    fn.restype = ctypes.c_size_t
    return int(fn(*args))


def _i64(t, what):
    if not (t.is_cuda and t.dtype == torch.long and t.is_contiguous()):
        raise RuntimeError(f'{what} must be a contiguous int64 device tensor')
    return t


def max_iou_assign(boxes, gt, pos_iou_thr, neg_iou_thr, min_pos_iou, gt_labels=None, inside=None, gt_max_assign_all=True):
    """mr3d_max_iou_assign -> (assigned_gt_inds i64 [n], max_overlaps f32 [n], assigned_labels i64 [n] | None)"""
    if not (boxes.is_cuda and boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.stride(1) == 1 and boxes.size(1) >= 6):
        raise RuntimeError('boxes must be a float32 device tensor [n, >=6] with unit inner stride')
    gt = _dev_f32(gt[:, :6].contiguous(), 'gt_bboxes')
    n, g = boxes.size(0), gt.size(0)
    if n == 0 or g == 0:
        raise ValueError('No gt or bboxes')                       # max_iou_assigner.py:79-80
    lo, hi = (0.0, float(neg_iou_thr)) if not isinstance(neg_iou_thr, tuple) else (float(neg_iou_thr[0]), float(neg_iou_thr[1]))
    dev = boxes.device
    assigned = torch.empty(n, dtype=torch.long, device=dev)
    max_ov = torch.empty(n, dtype=torch.float32, device=dev)
    labels = torch.empty(n, dtype=torch.long, device=dev) if gt_labels is not None else None
    if gt_labels is not None:
        gt_labels = _i64(gt_labels.contiguous(), 'gt_labels')
    if inside is not None:
        inside = inside.to(torch.uint8).contiguous()
    L = lib()
    wsb = _sz(L.mr3d_max_iou_assign_workspace_bytes, n, g)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = L.mr3d_max_iou_assign(ptr(boxes), int(boxes.stride(0)), n, ctypes.c_void_p(_opt(inside)), ptr(gt), g,
                                   ctypes.c_void_p(_opt(gt_labels)), ctypes.c_float(pos_iou_thr), ctypes.c_float(lo),
                                   ctypes.c_float(hi), ctypes.c_float(min_pos_iou), int(bool(gt_max_assign_all)), ptr(assigned),
                                   ptr(max_ov), ctypes.c_void_p(_opt(labels)), ptr(ws), ctypes.c_size_t(wsb), stream_ptr(dev))
    check(st, 'mr3d_max_iou_assign')
    return assigned, max_ov, labels


def device_draws(count, device, generator=None):
    """`count` uniform 31-bit words on the device (the sampler takes them modulo the candidate count)"""
    return torch.randint(0, 2 ** 31 - 1, (max(int(count), 1),), dtype=torch.int32, device=device, generator=generator)


def _draws(t, need, device, what):
    if t is None:
        return device_draws(need, device)
    t = torch.as_tensor(t).to(device=device, dtype=torch.int32).contiguous()
    if t.numel() < need:          # short draw lists are fine when the candidates fit (the reference then draws nothing)
        t = torch.cat([t, torch.zeros(need - t.numel(), dtype=torch.int32, device=device)])
    return t


def random_sample(assigned_gt_inds, num, pos_fraction, neg_pos_ub=-1, draws_pos=None, draws_neg=None, want_sel=False):
    """mr3d_random_sample -> (pos_inds i64 [int(num*pos_fraction)], neg_inds i64 [num], counts i32 [4], sel i32 [n] | None);
    lists ascending and padded with -1; nothing is synchronised."""
    a = _i64(assigned_gt_inds, 'assigned_gt_inds')
    n, dev = a.numel(), a.device
    cap_pos = int(num * pos_fraction)
    pos = torch.empty(max(cap_pos, 1), dtype=torch.long, device=dev)
    neg = torch.empty(num, dtype=torch.long, device=dev)
    counts = torch.empty(4, dtype=torch.int32, device=dev)
    sel = torch.empty(n, dtype=torch.int32, device=dev) if want_sel else None
    dp, dn = _draws(draws_pos, max(cap_pos, 1), dev, 'draws_pos'), _draws(draws_neg, num, dev, 'draws_neg')
    L = lib()
    wsb = _sz(L.mr3d_random_sample_workspace_bytes, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = L.mr3d_random_sample(ptr(a), n, int(num), ctypes.c_float(pos_fraction), ctypes.c_float(neg_pos_ub), ptr(dp), ptr(dn),
                                  ptr(pos), ptr(neg), ptr(counts), ctypes.c_void_p(_opt(sel)), ptr(ws), ctypes.c_size_t(wsb),
                                  stream_ptr(dev))
    check(st, 'mr3d_random_sample')
    return pos[:cap_pos], neg, counts, sel


def anchor_target(anchors, assigned_gt_inds, sel, gt, means, stds, pos_weight, gt_labels=None):
    """mr3d_anchor_target -> labels i64 [n], label_weights [n], bbox_targets [n,6], bbox_weights [n,6]"""
    anchors, gt = _dev_f32(anchors, 'anchors'), _dev_f32(gt[:, :6].contiguous(), 'gt_bboxes')
    n, dev = anchors.size(0), anchors.device
    labels = torch.empty(n, dtype=torch.long, device=dev)
    lw = torch.empty(n, dtype=torch.float32, device=dev)
    bt = torch.empty((n, 6), dtype=torch.float32, device=dev)
    bw = torch.empty((n, 6), dtype=torch.float32, device=dev)
    if gt_labels is not None:
        gt_labels = _i64(gt_labels.contiguous(), 'gt_labels')
    with torch.cuda.device(dev):
        st = lib().mr3d_anchor_target(ptr(anchors), int(anchors.stride(0)), n, ptr(_i64(assigned_gt_inds, 'assigned')), ptr(sel),
                                      ptr(gt), ctypes.c_void_p(_opt(gt_labels)), _f6(means, [0.] * 6), _f6(stds, [1.] * 6),
                                      ctypes.c_float(pos_weight), ptr(labels), ptr(lw), ptr(bt), ptr(bw), stream_ptr(dev))
    check(st, 'mr3d_anchor_target')
    return labels, lw, bt, bw


def concat_gt_assign(assigned_proposals, num_gt):
    a = _i64(assigned_proposals, 'assigned_gt_inds')
    out = torch.empty(a.numel() + num_gt, dtype=torch.long, device=a.device)
    with torch.cuda.device(a.device):
        st = lib().mr3d_concat_gt_assign(ptr(a), a.numel(), int(num_gt), ptr(out), stream_ptr(a.device))
    check(st, 'mr3d_concat_gt_assign')
    return out


def bbox_target3d(proposals, gt, gt_labels, assigned_all, pos_inds, neg_inds, counts, cap, batch_index, means, stds, pos_weight):
    """mr3d_bbox_target3d -> rois [cap,7], labels [cap], label_weights [cap], bbox_targets [cap,6], bbox_weights [cap,6],
    pos_assigned_gt_inds [cap]"""
    gt = _dev_f32(gt[:, :6].contiguous(), 'gt_bboxes')
    dev = gt.device
    if not (proposals.is_cuda and proposals.dtype == torch.float32 and proposals.stride(1) == 1 and proposals.size(1) >= 6):
        raise RuntimeError('proposals must be a float32 device tensor [p, >=6] with unit inner stride')
    rois = torch.empty((cap, 7), dtype=torch.float32, device=dev)
    labels = torch.empty(cap, dtype=torch.long, device=dev)
    lw = torch.empty(cap, dtype=torch.float32, device=dev)
    bt = torch.empty((cap, 6), dtype=torch.float32, device=dev)
    bw = torch.empty((cap, 6), dtype=torch.float32, device=dev)
    pgi = torch.empty(cap, dtype=torch.long, device=dev)
    with torch.cuda.device(dev):
        st = lib().mr3d_bbox_target3d(ptr(proposals), int(proposals.stride(0)), ptr(gt), gt.size(0),
                                      ptr(_i64(gt_labels.contiguous(), 'gt_labels')), ptr(_i64(assigned_all, 'assigned_all')),
                                      ptr(_i64(pos_inds.contiguous(), 'pos_inds')), ptr(_i64(neg_inds.contiguous(), 'neg_inds')),
                                      ptr(counts), int(cap), int(batch_index), _f6(means, [0.] * 6), _f6(stds, [1.] * 6),
                                      ctypes.c_float(pos_weight), ptr(rois), ptr(labels), ptr(lw), ptr(bt), ptr(bw), ptr(pgi),
                                      stream_ptr(dev))
    check(st, 'mr3d_bbox_target3d')
    return rois, labels, lw, bt, bw, pgi


# -------------------------------------------------------------------------------------------- losses
_loss_ws = {}


def _loss_workspace(device):
    key = (device.type, device.index)
    if key not in _loss_ws:
        _loss_ws[key] = torch.empty(_sz(lib().mr3d_loss_workspace_bytes), dtype=torch.uint8, device=device)
    return _loss_ws[key]


def _avg(avg_factor):
    """avg_factor: python number or a device float tensor -> (dev_ptr_or_None, host_float)"""
    if torch.is_tensor(avg_factor):
        assert avg_factor.is_cuda and avg_factor.dtype == torch.float32 and avg_factor.numel() == 1
        return ctypes.c_void_p(avg_factor.data_ptr()), ctypes.c_float(0.0)
    return ctypes.c_void_p(0), ctypes.c_float(float(avg_factor))


def avg_factor_from_counts(counts, mode):
    """counts: i32 [num_imgs, >=2] device tensor of {pos, neg, ...} records -> device float scalar"""
    counts = counts.reshape(-1, counts.shape[-1]).contiguous()
    out = torch.empty(1, dtype=torch.float32, device=counts.device)
    with torch.cuda.device(counts.device):
        st = lib().mr3d_avg_factor_from_counts(ptr(counts), counts.size(1), counts.size(0), int(mode), ptr(out),
                                               stream_ptr(counts.device))
    check(st, 'mr3d_avg_factor_from_counts')
    return out


def weighted_bce_logits(score, labels, weights, avg_factor, want_grad=True):
    score = _dev_f32(score, 'score')
    n, c = (score.size(0), score.size(1)) if score.dim() == 2 else (score.numel(), 1)
    loss = torch.empty(1, dtype=torch.float32, device=score.device)
    grad = torch.empty_like(score) if want_grad else None
    ap, ah = _avg(avg_factor)
    with torch.cuda.device(score.device):
        st = lib().mr3d_weighted_bce_logits(ptr(score), ptr(_i64(labels.contiguous(), 'labels')), ptr(_dev_f32(weights, 'weights')),
                                            ctypes.c_longlong(n), c, ap, ah, ptr(loss), ctypes.c_void_p(_opt(grad)),
                                            ptr(_loss_workspace(score.device)), stream_ptr(score.device))
    check(st, 'mr3d_weighted_bce_logits')
    return loss, grad


def weighted_smoothl1(pred, target, weights, beta, avg_factor, want_grad=True):
    pred, target, weights = _dev_f32(pred, 'pred'), _dev_f32(target, 'target'), _dev_f32(weights, 'weights')
    assert pred.shape == target.shape == weights.shape
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    ap, ah = _avg(avg_factor)
    with torch.cuda.device(pred.device):
        st = lib().mr3d_weighted_smoothl1(ptr(pred), ptr(target), ptr(weights), ctypes.c_longlong(pred.numel()), ctypes.c_float(beta),
                                          ap, ah, ptr(loss), ctypes.c_void_p(_opt(grad)), ptr(_loss_workspace(pred.device)),
                                          stream_ptr(pred.device))
    check(st, 'mr3d_weighted_smoothl1')
    return loss, grad


def weighted_softmax_ce(score, labels, weights, valid_rows=None, want_grad=True):
    score = _dev_f32(score, 'score')
    loss = torch.empty(1, dtype=torch.float32, device=score.device)
    acc = torch.empty(1, dtype=torch.float32, device=score.device)
    grad = torch.empty_like(score) if want_grad else None
    with torch.cuda.device(score.device):
        st = lib().mr3d_weighted_softmax_ce(ptr(score), ptr(_i64(labels.contiguous(), 'labels')), ptr(_dev_f32(weights, 'weights')),
                                            score.size(0), score.size(1), ctypes.c_void_p(_opt(valid_rows)), ptr(loss), ptr(acc),
                                            ctypes.c_void_p(_opt(grad)), ptr(_loss_workspace(score.device)),
                                            stream_ptr(score.device))
    check(st, 'mr3d_weighted_softmax_ce')
    return loss, acc, grad


def mask_bce(pred, target, labels, valid_rows=None, want_grad=True):
    pred, target = _dev_f32(pred, 'mask_pred'), _dev_f32(target, 'mask_targets')
    p, c = pred.size(0), pred.size(1)
    spatial = pred[0, 0].numel()
    assert target.numel() == p * spatial
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    with torch.cuda.device(pred.device):
        st = lib().mr3d_mask_bce(ptr(pred), ptr(target), ptr(_i64(labels.contiguous(), 'labels')), p, c, ctypes.c_longlong(spatial),
                                 ctypes.c_void_p(_opt(valid_rows)), ptr(loss), ctypes.c_void_p(_opt(grad)),
                                 ptr(_loss_workspace(pred.device)), stream_ptr(pred.device))
    check(st, 'mr3d_mask_bce')
    return loss, grad

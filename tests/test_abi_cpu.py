"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mr3d.h declares; host-side argument checks of the mmdet.ops mirror."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'mr3d.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mr3d_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    import mr3d_amd
    L = mr3d_amd._hip.lib()
    syms = _declared_symbols()
    assert len(syms) >= 7
    for s in syms:
        assert hasattr(L, s), f'{s} declared in include/mr3d.h but not exported'
    assert b'gfx950' in L.mr3d_version()
    assert L.mr3d_status_string(0) == b'ok'


def test_workspace_query_is_host_only_and_monotone():
    import mr3d_amd
    L = mr3d_amd._hip.lib()
    sizes = [int(L.mr3d_nms3d_workspace_bytes(n)) for n in (0, 1, 64, 65, 2000, 4096, 5000)]
    assert all(s % 256 == 0 for s in sizes)
    assert sizes == sorted(sizes)
    # n=2000: order 8 KB + boxes 64 KB + 2000x32 u64 mask 512 KB
    assert 2000 * 32 * 8 <= sizes[4] <= 2000 * 32 * 8 + 128 * 1024


def test_mirror_exports_reference_names():
    from mr3d_amd import mmdet_ops as ops
    for name in ['nms', 'soft_nms', 'RoIAlign', 'roi_align', 'RoIPool', 'roi_pool', 'DeformConv', 'DeformConvPack',
                 'DeformRoIPooling', 'DeformRoIPoolingPack', 'ModulatedDeformRoIPoolingPack', 'ModulatedDeformConv',
                 'ModulatedDeformConvPack', 'deform_conv', 'modulated_deform_conv', 'deform_roi_pooling',
                 'SigmoidFocalLoss', 'sigmoid_focal_loss', 'RoIAlign3D']:          # mmdet/ops/__init__.py:10-16
        assert hasattr(ops, name), name
    layer = ops.RoIAlign3D(out_size=7, out_size_depth=3, spatial_scale=1 / 4, spatial_scale_depth=1 / 2, sample_num=2)
    assert (layer.out_size, layer.out_size_depth, layer.sample_num) == (7, 3, 2)


def test_error_conventions_without_gpu():
    from mr3d_amd.mmdet_ops import nms, soft_nms, RoIAlign3D, RoIAlign
    with pytest.raises(TypeError):
        nms([[0, 0, 1, 1, 0, 1, 0.5]], 0.5)                       # nms_wrapper.py:34-36
    kept, inds = nms(torch.zeros((0, 7)), 0.5)                    # empty input -> empty long (:39-40)
    assert inds.dtype == torch.long and inds.numel() == 0 and kept.shape == (0, 7)
    with pytest.raises(NotImplementedError):
        nms(torch.zeros((3, 7)), 0.5)                             # CPU tensors: no silent 2D fallback
    with pytest.raises(NotImplementedError):
        soft_nms(torch.zeros((3, 5)), 0.5)
    with pytest.raises(NotImplementedError):
        RoIAlign3D(7, 3, 0.25, 0.5, 2)(torch.zeros(1, 2, 4, 8, 8), torch.zeros(1, 7))   # CPU features
    with pytest.raises(TypeError):
        RoIAlign3D('7', 3, 0.25, 0.5, 2)(torch.zeros(1, 2, 4, 8, 8), torch.zeros(1, 7))
    with pytest.raises(NotImplementedError):
        RoIAlign(7, 0.25)


def test_compiled_extension_modules_export_the_reference_entry_points():
    """roi_align_cuda / nms_cuda are compiled PYBIND11 modules (csrc/torch_ext, built by `make ext` / build()) with the
    reference's entry points (roi_align_cuda.cpp:152-157, nms_cuda.cpp:23-26); the ctypes forms offer the same names"""
    import mr3d_amd  # noqa: F401
    from mr3d_amd.mmdet_ops.roi_align import roi_align_cuda, roi_align_ctypes
    from mr3d_amd.mmdet_ops.nms import nms_cuda, nms_ctypes
    assert roi_align_cuda.__file__.endswith('.so') and nms_cuda.__file__.endswith('.so'), 'extension modules not built'
    for m in (roi_align_cuda, roi_align_ctypes):
        assert all(callable(getattr(m, n)) for n in ('forward', 'forward3d', 'backward', 'backward3d'))
    for m in (nms_cuda, nms_ctypes):
        assert all(callable(getattr(m, n)) for n in ('nms', 'nms_3d'))
    with pytest.raises(RuntimeError, match='must be a CUDAtensor'):          # CHECK_CUDA (nms_cuda.cpp:4,17)
        nms_cuda.nms_3d(torch.zeros((3, 7)), 0.5)
    with pytest.raises(RuntimeError, match='must be a CUDAtensor'):          # CHECK_INPUT (roi_align_cuda.cpp:34-39)
        roi_align_cuda.forward3d(torch.zeros(1, 4, 2, 4, 4), torch.zeros(1, 7), 1, 2, 2, 1.0, 1.0, 2, torch.zeros(1, 4, 1, 2, 2))
    with pytest.raises(NotImplementedError):
        nms_cuda.nms(torch.zeros((3, 5)), 0.5)


def _conv_desc(k, s, h, w, cin, cout, d=4, relu=0, up=False):
    from mr3d_amd import _hip
    c = _hip.Conv3dDesc()
    c.n, c.d, c.h, c.w_in, c.cin, c.cout = 1, d, h, w, cin, cout
    c.kd = c.kh = c.kw = k
    c.sd = c.sh = c.sw = s
    c.pd = c.ph = c.pw = k // 2
    c.relu = relu
    c.workspace_bytes = -1
    if up:
        c.up = 64
        c.up_d, c.up_h, c.up_w = (d + 1) // 2, (h + 1) // 2, (w + 1) // 2
    return c


def test_conv3d_size_guards_sit_exactly_at_the_documented_limits():
    """include/mr3d.h 'Size limits': the fp32 conv kernels use 24-bit products / 32-bit byte offsets inside a tile.  Above a
    limit: MR3D_ERR_UNSUPPORTED before anything is launched; just below: a kernel instance is chosen
    (mr3d_conv3d_forward_variant launches nothing).  The streaming pointwise kernels index with 64 bits and stay available."""
    import ctypes
    from mr3d_amd import _hip
    L = _hip.lib()
    OK, UNSUPPORTED = 0, 4
    v = _hip.Conv3dVariant()
    probe = lambda c: L.mr3d_conv3d_forward_variant(ctypes.byref(c), ctypes.byref(v))
    # h * w * cin: 2^24 elements is out, one column less is in
    assert probe(_conv_desc(3, 1, 512, 512, 64, 64)) == UNSUPPORTED
    assert probe(_conv_desc(3, 1, 512, 511, 64, 64)) == OK
    # the real entry point with non-null operands: refused by the guard, nothing launched (no GPU needed)
    c = _conv_desc(3, 1, 512, 512, 64, 64)
    c.x = c.w = c.y = 64
    assert L.mr3d_conv3d_forward(ctypes.byref(c), None) == UNSUPPORTED
    # h * w: 2^20 positions is out
    # (stride-2 instance: the only fp32 3x3x3 form that takes 8 input channels, so h * w * cin = 2^23 stays inside its own limit)
    assert probe(_conv_desc(3, 2, 1024, 1024, 8, 16, d=2)) == UNSUPPORTED
    assert probe(_conv_desc(3, 2, 1024, 1023, 8, 16, d=2)) == OK
    # output plane y_h * y_w * cout * ys_d below 2^25 elements
    assert probe(_conv_desc(3, 1, 256, 256, 16, 512)) == UNSUPPORTED        # 2^16 * 2^9 = 2^25
    assert probe(_conv_desc(3, 1, 256, 255, 16, 512)) == OK
    # weights below 2^29 elements: 27 * 4096 * 8192 > 2^29
    assert probe(_conv_desc(3, 1, 8, 8, 4096, 8192)) == UNSUPPORTED
    assert probe(_conv_desc(3, 1, 8, 8, 2048, 4096)) == OK                  # 27 * 2^23 < 2^29
    # a 1x1x1 convolution on a 512 x 512 x 64 plane takes the streaming kernel (64-bit indexing), guard or not
    assert probe(_conv_desc(1, 1, 512, 512, 64, 64, d=2)) == OK and v.streaming_pointwise == 1
    # strided 3x3x3 data gradient (parity-decomposed): same ranges on the dy plane
    d = _conv_desc(3, 1, 512, 512, 64, 64)
    d.x = d.w = d.y = 64
    d.pd = d.ph = d.pw = 0
    d.y_d, d.y_h, d.y_w = 2 * 4 - 1, 2 * 512 - 1, 2 * 512 - 1
    assert L.mr3d_conv3d_dgrad_s2k3(ctypes.byref(d), None) == UNSUPPORTED


def test_stem_size_guard():
    """stem kernels: byte offsets inside one channel volume stay below 2^30 (d * h * w < 2^28), output below 2^31 elements"""
    import ctypes
    from mr3d_amd import _hip
    L = _hip.lib()
    p = ctypes.c_void_p(64)
    assert L.mr3d_stem_conv_forward(p, p, p, p, 1, 1024, 512, 512, 1, None) == 4         # 2^28 voxels
    assert L.mr3d_stem_conv_wgrad_ws(p, p, p, p, 1, 1024, 512, 512, None, ctypes.c_longlong(0), None) == 4
    assert L.mr3d_stem_conv_forward(p, p, p, p, 9, 1000, 512, 512, 1, None) == 4         # volume fits, output 9 * 1000 * 256 * 256 * 64 B >= 2^31
    # the bf16 stem kernels fetch their slabs with 32-bit buffer offsets: one image (3 channels, fp32) below 2^31 bytes
    assert L.mr3d_stem_conv_forward_bf16(p, p, p, p, 1, 684, 512, 512, 1, None) == 4     # 684 * 2^18 * 12 B > 2^31
    assert L.mr3d_stem_conv_forward_bf16(p, p, p, p, 40, 100, 512, 512, 1, None) == 4    # images fit, the bf16 output (40 x 100 x 2^16 x 32 B) does not
    assert L.mr3d_stem_conv_wgrad_bf16(p, p, p, p, 1, 684, 512, 512, None) == 4
    assert L.mr3d_stem_conv_wgrad_bf16_pooled(p, p, p, p, p, p, 1, 684, 512, 512, None) == 4


def test_winograd_2d_variant_reports_the_kernel_form_per_map():
    """csrc/conv3d_wino2d.hip holds two 64-wide kernels; mr3d_conv3d_forward_variant says which one a launch takes (nothing is
    launched): the wide form (4x8x8 tiles, Cin chunks of 8) on maps with at least two rounds of such tiles -- the P2 map --, the 2x8x8
    kernel on the P3 map, and either one when MR3D_TUNE_WINO2D_FORM forces it."""
    import ctypes
    from mr3d_amd import _hip
    L = _hip.lib()
    v = _hip.Conv3dVariant()
    p2, p3 = _conv_desc(3, 1, 128, 128, 64, 64, d=40), _conv_desc(3, 1, 64, 64, 64, 64, d=20)
    tile = lambda c: (L.mr3d_conv3d_forward_variant(ctypes.byref(c), ctypes.byref(v)), v.winograd, v.tile_d, v.cin_chunk)[1:]
    try:
        assert tile(p2) == (8, 4, 8) and tile(p3) == (8, 2, 16)
        L.mr3d_tuning_set(5, 1)
        assert tile(p2) == (8, 2, 16)
        L.mr3d_tuning_set(5, 2)
        assert tile(p3) == (8, 4, 8)
    finally:
        L.mr3d_tuning_set(5, 0)
    assert abs(v.mfma_issue_factor - 1.0 / 3.0) < 1e-6
    # the wide kernel addresses x with 32-bit byte offsets: a 2.5 GB input keeps the 2x8x8 kernel (whose own offsets are per tile)
    big = _conv_desc(3, 1, 128, 128, 64, 64, d=600)
    assert tile(big) == (8, 2, 16)


def test_up_even_flag_is_refused_where_no_kernel_implements_it():
    """MR3D_CONV_UP_EVEN (the even-position add of a stride-2 adjoint) exists in the generic fp32 kernel and the streaming
    pointwise kernels.  The two-dimensional Winograd instances must not be chosen for it, and the bf16 entry point refuses it."""
    import ctypes
    from mr3d_amd import _hip
    L = _hip.lib()
    v = _hip.Conv3dVariant()
    plain = _conv_desc(3, 1, 128, 128, 64, 64, d=40)
    assert L.mr3d_conv3d_forward_variant(ctypes.byref(plain), ctypes.byref(v)) == 0 and v.winograd == 8
    flagged = _conv_desc(3, 1, 128, 128, 64, 64, d=40, relu=_hip.CONV_UP_EVEN, up=True)
    assert L.mr3d_conv3d_forward_variant(ctypes.byref(flagged), ctypes.byref(v)) == 0 and v.winograd != 8
    narrow = _conv_desc(3, 1, 128, 128, 16, 16, d=40, relu=_hip.CONV_UP_EVEN, up=True)
    assert L.mr3d_conv3d_forward_variant(ctypes.byref(narrow), ctypes.byref(v)) == 0 and v.winograd != 8
    b = _conv_desc(3, 1, 32, 32, 64, 64, relu=_hip.CONV_UP_EVEN, up=True)
    b.x = b.w = b.y = 64
    assert L.mr3d_conv3d_forward_bf16(ctypes.byref(b), None) == 4

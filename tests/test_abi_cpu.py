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

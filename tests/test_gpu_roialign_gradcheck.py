"""The reference's own test of its RoIAlign op is a torch.autograd.gradcheck (mmdet/ops/roi_align/gradcheck.py:14-29: random
rois in the upper half of the image, 16 channels, sample_num 0 and 2, eps = atol = 1e-3 in fp32).  Same check for the 3D op
(functions/roi_align_3d.py, modules/roi_align_3d.py) through the drop-in module, in both memory layouts, in fp32 with the reference's
tolerances and in fp64 (the kernels dispatch double like the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF,
roi_align_kernel.cu:323,674) with gradcheck's defaults.  Cubic pooled sizes: there the reference's top-gradient index
(roi_align_kernel.cu:554-555, reproduced by default) coincides with the correct one, so the check holds in both index modes; a
non-cubic size is checked with bug_compat=False only."""
import numpy as np
import pytest
import torch
from torch.autograd import gradcheck

pytestmark = pytest.mark.gpu


def _inputs(dtype, fmt, seed, num_imgs=2, channels=4, feat=(5, 7, 7), num_rois=6, scale=1.0 / 8, scale_d=1.0 / 4):
    rng = np.random.RandomState(seed)
    d, h, w = feat
    img = np.array([w / scale, h / scale, d / scale_d])
    lo = rng.rand(num_rois, 3) * img * 0.5
    hi = lo + rng.rand(num_rois, 3) * img * 0.5
    rois = np.stack([rng.randint(num_imgs, size=num_rois), lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(num_imgs, channels, d, h, w, generator=g, dtype=torch.float64).to(dtype).cuda()
    f = f.contiguous(memory_format=fmt).requires_grad_(True)
    return f, torch.from_numpy(rois).to(dtype).cuda()


@pytest.mark.parametrize('fmt', [torch.contiguous_format, torch.channels_last_3d])
@pytest.mark.parametrize('sample_num', [0, 2])
@pytest.mark.parametrize('bug_compat', [True, False])
def test_roi_align3d_gradcheck_fp32_reference_tolerances(fmt, sample_num, bug_compat):
    from mr3d_amd.mmdet_ops import RoIAlign3D
    f, rois = _inputs(torch.float32, fmt, 3)
    layer = RoIAlign3D(3, 3, 1.0 / 8, 1.0 / 4, sample_num=sample_num, bug_compat=bug_compat)
    assert gradcheck(lambda x: layer(x, rois), (f,), eps=1e-3, atol=1e-3, rtol=1e-2, nondet_tol=1e-5)


@pytest.mark.parametrize('out', [(3, 3), (2, 4)])
def test_roi_align3d_gradcheck_fp64(out):
    from mr3d_amd.mmdet_ops import RoIAlign3D
    out_size, out_depth = out[1], out[0]
    f, rois = _inputs(torch.float64, torch.contiguous_format, 5)
    layer = RoIAlign3D(out_size, out_depth, 1.0 / 8, 1.0 / 4, sample_num=2, bug_compat=(out_size == out_depth))
    assert gradcheck(lambda x: layer(x, rois), (f,), nondet_tol=1e-10)

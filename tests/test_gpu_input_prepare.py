"""Device-side input preparation (csrc/input_ops.hip) against the numpy restatement of the reference's per-slice
host pipeline (oracle/input_ref.py).  No-resample path: bit-exact fp32.  1.5x path: <= 1e-5 relative (parity
unpinned w.r.t. scikit-image, which the reference delegates to and which is absent here)."""
import numpy as np
import pytest
import torch

from oracle import input_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape,dtype', [((128, 128, 32), np.uint8), ((100, 77, 19), np.float32), ((64, 96, 40), np.uint8)])
def test_prepare_volume_matches_per_slice_pipeline(shape, dtype):
    from mr3d_amd.datasets import prepare_volume, IMG_NORM_CFG
    rng = np.random.default_rng(1)
    g = rng.integers(0, 256, shape).astype(dtype) if dtype == np.uint8 else (rng.random(shape) * 255).astype(np.float32)
    img, meta = prepare_volume(g, layout='HWD')
    ref = R.prepare_volume_ref(g, IMG_NORM_CFG['mean'], IMG_NORM_CFG['std'])
    assert tuple(img.shape) == (1,) + ref.shape
    assert np.array_equal(img[0].cpu().numpy(), ref)                      # bit-exact: (g - mean) / std in fp32
    H, W, D = shape
    assert meta['img_shape'] == (H, W, 3, D) and meta['pad_shape'] == (ref.shape[2], ref.shape[3], 3, D)
    # [D,H,W] input gives the same tensor
    img2, _ = prepare_volume(np.ascontiguousarray(g.transpose(2, 0, 1)), layout='DHW')
    assert torch.equal(img, img2)


def test_prepare_volume_second_pathway_upscale():
    from mr3d_amd.datasets import prepare_volume, IMG_NORM_CFG
    rng = np.random.default_rng(2)
    g = (rng.random((48, 40, 22)) * 255).astype(np.float32)
    img, meta = prepare_volume(g, upscale=1.5)
    ref = R.prepare_volume_ref(g, IMG_NORM_CFG['mean'], IMG_NORM_CFG['std'], upscale=1.5)
    assert tuple(img.shape) == (1,) + ref.shape and meta['img_shape'] == (72, 60, 3, 33)
    np.testing.assert_allclose(img[0].cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    assert (img[0, :, :, 72:, :] == 0).all() and (img[0, :, :, :, 60:] == 0).all()       # padding stays zero


def test_prepare_volume_full_size_properties():
    """512x512x80 (the bench volume): channel c is an affine map of channel 0; the stem accepts the tensor as is"""
    from mr3d_amd.datasets import prepare_volume, IMG_NORM_CFG
    g = torch.randint(0, 256, (512, 512, 80), dtype=torch.uint8, device='cuda')
    img, meta = prepare_volume(g)
    assert tuple(img.shape) == (1, 3, 80, 512, 512) and img.is_contiguous()
    m, s = IMG_NORM_CFG['mean'], IMG_NORM_CFG['std']
    back = img[0, 0] * s[0] + m[0]
    assert torch.allclose(back, g.permute(2, 0, 1).float(), atol=1e-3)
    assert torch.allclose((back - m[2]) / s[2], img[0, 2], atol=1e-5)

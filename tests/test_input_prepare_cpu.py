"""CPU: the restatement of skimage.transform.resize (oracle/input_ref.resize_trilinear_ref) pinned against the routine
scikit-image 0.18 delegates to for 3-D input -- scipy.ndimage.map_coordinates(order=1, mode='mirror') on the coordinates
c = (o + 0.5) * in/out - 0.5 (transform/_warps.py, n-dimensional branch; default mode 'reflect' = ndimage 'mirror').
scikit-image itself is not installed; scipy is."""
import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import input_ref as R


def _skimage_resize_nd(vol, out_shape):
    """the n-dimensional branch of skimage.transform.resize (0.18) for a float volume that is being up-scaled"""
    factors = np.asarray(vol.shape, dtype=float) / np.asarray(out_shape, dtype=float)
    sigma = np.maximum(0, (factors - 1) / 2)                      # anti-aliasing: zero when up-scaling
    assert not sigma.any()
    coords = [factors[i] * (np.arange(d) + 0.5) - 0.5 for i, d in enumerate(out_shape)]
    cmap = np.array(np.meshgrid(*coords, sparse=False, indexing='ij'))
    return ndi.map_coordinates(vol.astype(np.float64), cmap, order=1, mode='mirror', cval=0)


@pytest.mark.parametrize('shape,scale', [((22, 48, 40), 1.5), ((5, 7, 9), 1.5), ((1, 6, 4), 1.5), ((8, 8, 8), 2.0), ((10, 12, 6), 1.25)])
def test_resize_restatement_equals_ndimage_map_coordinates(shape, scale):
    rng = np.random.default_rng(sum(shape))
    vol = rng.random(shape).astype(np.float32) * 255
    out_shape = tuple(int(round(s * scale)) for s in shape)
    got = R.resize_trilinear_ref(vol, out_shape)
    exp = _skimage_resize_nd(vol, out_shape)
    assert got.shape == exp.shape
    # the restatement forms the coordinates in float32 (as the HIP kernel does): 1e-5 of the value range
    np.testing.assert_allclose(got, exp, rtol=0, atol=255 * 1e-5)


def test_resize_border_rule_is_mirror_not_clamp():
    """the first output sample of a 1.5x up-scaling sits at -1/6: 'mirror' mixes voxel 0 with voxel 1 (5/6, 1/6); edge clamping
    would return voxel 0"""
    line = np.array([[[3.0, 9.0, 4.0, 1.0, 7.0, 2.0, 8.0, 5.0]]], np.float32)
    out = R.resize_trilinear_ref(line, (1, 1, 12))[0, 0]
    np.testing.assert_allclose(out[0], (5 / 6) * 3.0 + (1 / 6) * 9.0, rtol=1e-6)
    np.testing.assert_allclose(out[-1], (5 / 6) * 5.0 + (1 / 6) * 8.0, rtol=1e-6)
    np.testing.assert_allclose(out, _skimage_resize_nd(line, (1, 1, 12))[0, 0], rtol=0, atol=1e-5)

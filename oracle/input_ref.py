"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's input preparation.

  prepare_volume_ref   mmdet/datasets/coco_3d_2scales.py:245-263 (per slice: np.repeat -> ImageTransform -> stack ->
                       transpose(1,0,2,3)), mmdet/datasets/transforms.py:33-51 with keep_ratio at scale 1 (the shipped
                       pipeline passes the image's own size, coco_3d_2scales.py:214,232), mmcv.imnormalize =
                       (img.astype(float32) - mean) / std after the BGR->RGB swap (a no-op on replicated grey), and
                       mmcv.impad_to_multiple = zero padding at the bottom/right AFTER normalisation.
  resize_trilinear_ref what skimage.transform.resize (third party, scikit-image==0.18.0, NOT in the reference tree) computes
                       for an up-scaling with its defaults, restated from its published source (transform/_warps.py, the
                       n-dimensional branch): coordinates c = (o + 0.5) * in/out - 0.5, anti-aliasing sigma max(0, (in/out
                       - 1) / 2) = 0, then scipy.ndimage.map_coordinates(order=1, mode='mirror') -- skimage's default mode
                       'reflect' is ndimage's 'mirror' (_shared/utils.py).  Pinned against scipy.ndimage.map_coordinates
                       itself (tests/test_input_prepare_cpu.py); scikit-image is not installed, so the wrapper logic
                       (img_as_float's dtype-dependent scaling, clipping to the input range) stays a restatement.
"""
import numpy as np


def resize_trilinear_ref(vol_dhw, out_shape):
    v = vol_dhw.astype(np.float64)
    for ax, n_out in enumerate(out_shape):
        n = v.shape[ax]
        c = (np.arange(n_out) + 0.5) * (np.float32(n) / np.float32(n_out)) - 0.5
        f = np.floor(c).astype(int)

        def mirror(i):                                      # ndimage 'mirror': about the centre of the edge voxel
            if n == 1:
                return np.zeros_like(i)
            i = np.abs(i)
            i = np.where(i >= n, 2 * (n - 1) - i, i)
            return np.clip(i, 0, n - 1)
        i0, i1 = mirror(f), mirror(f + 1)
        t = (c - f).reshape([-1 if a == ax else 1 for a in range(3)])
        v = np.take(v, i0, axis=ax) * (1 - t) + np.take(v, i1, axis=ax) * t
    return v


def prepare_volume_ref(grey_hwd, mean, std, size_divisor=32, upscale=None):
    """-> [3, D', Hp, Wp] float32"""
    g = np.asarray(grey_hwd)
    vol = g.transpose(2, 0, 1).astype(np.float32)                  # [D, H, W]
    if upscale:
        out_shape = tuple(int(round(s * upscale)) for s in vol.shape)
        vol = resize_trilinear_ref(vol, out_shape).astype(np.float32)
    D, H, W = vol.shape
    Hp, Wp = -(-H // size_divisor) * size_divisor, -(-W // size_divisor) * size_divisor
    out = np.zeros((3, D, Hp, Wp), np.float32)
    for c in range(3):
        out[c, :, :H, :W] = (vol - np.float32(mean[c])) / np.float32(std[c])
    return out

"""numpy restatement of the mask-crop semantics (csrc/mask_ops.hip); TEST INFRASTRUCTURE.

The reference (mmdet/core/mask/mask_target.py:35-44, fcn_mask_head_3d.py:163-189) delegates the resampling to
skimage.transform.resize, which is absent from its tree and from this image.  mask_target() restates the published
algorithm of scikit-image 0.18 (the target rule is checked against scipy.ndimage, the routines it calls); mask_paste()
is the build-defined test-time resampling (parity unpinned)."""
import numpy as np


def _crop(box, W, H, D):
    x1, y1, x2, y2, z1, z2 = (int(v) for v in np.asarray(box[:6], np.float32).astype(np.int32))
    w, h, d = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1), max(z2 - z1 + 1, 1)
    xa, ya, za = max(x1, 0), max(y1, 0), max(z1, 0)
    xb, yb, zb = min(x1 + w, W), min(y1 + h, H), min(z1 + d, D)
    return xa, ya, za, xb - xa, yb - ya, zb - za


def _stencil(out, inn):
    """per output index the source index range [lo, hi] (before mirroring) that skimage.transform.resize touches with its
    defaults: order-1 taps at c = (o + 0.5) in/out - 0.5, widened by the radius of the anti-aliasing Gaussian where the axis
    shrinks (sigma = (in/out - 1) / 2, scipy.ndimage radius int(4 sigma + 0.5))"""
    f = inn / out
    c = f * (np.arange(out) + 0.5) - 0.5
    fl = np.floor(c).astype(int)
    r = int(4 * ((f - 1) / 2) + 0.5) if inn > out else 0
    return fl - r, np.where(c > fl, fl + 1, fl) + r


def _mirror(i, n):
    if n == 1:
        return 0
    p = 2 * (n - 1)
    i = abs(i) % p
    return p - i if i >= n else i


def mask_target(gt_masks, boxes, gt_inds, md, mh, mw):
    """mask_target.py:41-44 with skimage.transform.resize restated from its published source (scikit-image 0.18,
    transform/_warps.py): x/255 -> Gaussian pre-filter on shrinking axes -> order-1 'mirror' sampling -> clip -> *255 ->
    astype(uint8) -> > 0.  Positive weights summing to 1: the result is 1 only where every voxel under both stencils is set.
    (The reference's float64 arithmetic additionally drops voxels whose all-ones sum lands one ulp below 1/255; see
    mask_target_literal_scipy.)"""
    G, D, H, W = gt_masks.shape
    out = np.zeros((len(boxes), md, mh, mw), np.float32)
    for p, box in enumerate(boxes):
        xa, ya, za, w, h, d = _crop(box, W, H, D)
        if w <= 0 or h <= 0 or d <= 0:
            continue
        crop = (gt_masks[gt_inds[p], za:za + d, ya:ya + h, xa:xa + w] > 0)
        (zl, zh), (yl, yh), (xl, xh) = _stencil(md, d), _stencil(mh, h), _stencil(mw, w)
        for oz in range(md):
            zs = [_mirror(i, d) for i in range(zl[oz], zh[oz] + 1)]
            for oy in range(mh):
                ys = [_mirror(i, h) for i in range(yl[oy], yh[oy] + 1)]
                sub = crop[np.ix_(zs, ys)]
                for ox in range(mw):
                    xs = [_mirror(i, w) for i in range(xl[ox], xh[ox] + 1)]
                    out[p, oz, oy, ox] = 1.0 if sub[:, :, xs].all() else 0.0
    return out


def mask_target_literal_scipy(crop_u8, out_shape):
    """the reference's expression evaluated with the routines scikit-image 0.18 calls for 3-D input (scipy.ndimage):
    255 * resize(crop) -> astype(uint8) -> > 0.  Used by the tests to show that the literal result is a rounding-dependent
    subset of mask_target()'s rule."""
    from scipy import ndimage as ndi
    img = crop_u8.astype(np.float64) / 255.0                                   # img_as_float(uint8)
    factors = np.asarray(img.shape, float) / np.asarray(out_shape, float)
    sigma = np.maximum(0, (factors - 1) / 2)
    filt = ndi.gaussian_filter(img, sigma, cval=0, mode='mirror') if sigma.any() else img
    coords = [factors[i] * (np.arange(n) + 0.5) - 0.5 for i, n in enumerate(out_shape)]
    cmap = np.array(np.meshgrid(*coords, sparse=False, indexing='ij'))
    res = ndi.map_coordinates(filt, cmap, order=1, mode='mirror', cval=0)
    res = np.clip(res, img.min(), img.max())                                   # _clip_warp_output
    return (255 * res).astype(np.uint8) > 0


def skimage_resize_float(img, out_shape):
    """scikit-image 0.18 transform.resize(img, out_shape) for a 3-D float32 image with its defaults, through the scipy.ndimage
    routines it calls (transform/_warps.py, n-dimensional branch): Gaussian pre-filter on shrinking axes, order-1 'mirror'
    sampling, clip to the input range; the result has the image's dtype"""
    from scipy import ndimage as ndi
    img = np.asarray(img, np.float32)
    factors = np.asarray(img.shape, float) / np.asarray(out_shape, float)
    sigma = np.maximum(0, (factors - 1) / 2)
    filt = ndi.gaussian_filter(img, sigma, cval=0, mode='mirror') if sigma.any() else img
    coords = [factors[i] * (np.arange(n) + 0.5) - 0.5 for i, n in enumerate(out_shape)]
    cmap = np.array(np.meshgrid(*coords, sparse=False, indexing='ij'))
    res = ndi.map_coordinates(filt, cmap, order=1, mode='mirror', cval=0)
    return np.clip(res, img.min(), img.max())


def mask_paste(prob, boxes, thr, D, H, W):
    """fcn_mask_head_3d.py:163-189: bbox_mask = resize(prob, (d, h, w)) > thr pasted at the integer box; voxels of a box that
    leave the volume are skipped (the reference's slice assignment raises there)"""
    out = np.zeros((len(boxes), D, H, W), np.uint8)
    for p, box in enumerate(boxes):
        x1, y1, x2, y2, z1, z2 = (int(v) for v in np.asarray(box[:6], np.float32).astype(np.int32))
        w, h, d = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1), max(z2 - z1 + 1, 1)
        m = (skimage_resize_float(prob[p], (d, h, w)) > np.float32(thr)).astype(np.uint8)
        za, ya, xa = max(z1, 0), max(y1, 0), max(x1, 0)
        zb, yb, xb = min(z1 + d, D), min(y1 + h, H), min(x1 + w, W)
        if zb > za and yb > ya and xb > xa:
            out[p, za:zb, ya:yb, xa:xb] = m[za - z1:zb - z1, ya - y1:yb - y1, xa - x1:xb - x1]
    return out

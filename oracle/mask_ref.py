"""numpy restatement of the build-defined mask-crop semantics (csrc/mask_ops.hip); TEST INFRASTRUCTURE.

The reference (mmdet/core/mask/mask_target.py:35-44, fcn_mask_head_3d.py:163-189) delegates the resampling to
skimage.transform.resize, which is absent from its tree and from this image: parity with it is unpinned.  This
file pins the GPU kernels to the documented definition instead."""
import numpy as np


def _crop(box, W, H, D):
    x1, y1, x2, y2, z1, z2 = (int(v) for v in np.asarray(box[:6], np.float32).astype(np.int32))
    w, h, d = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1), max(z2 - z1 + 1, 1)
    xa, ya, za = max(x1, 0), max(y1, 0), max(z1, 0)
    xb, yb, zb = min(x1 + w, W), min(y1 + h, H), min(z1 + d, D)
    return xa, ya, za, xb - xa, yb - ya, zb - za


def _axis(out, inn):
    o = np.arange(out, dtype=np.float32)
    s = (o + np.float32(0.5)) * (np.float32(inn) / np.float32(out)) - np.float32(0.5)
    s = np.clip(s, np.float32(0), np.float32(inn - 1)).astype(np.float32)
    lo = s.astype(np.int32)
    hi = np.minimum(lo + 1, inn - 1)
    return lo, hi, (s - lo.astype(np.float32)).astype(np.float32)


def _resample(src, out_shape):
    """trilinear, half-pixel centres, edge clamp; same evaluation order as the kernel (x, then y, then z), fp32"""
    src = src.astype(np.float32)
    zl, zh, fz = _axis(out_shape[0], src.shape[0])
    yl, yh, fy = _axis(out_shape[1], src.shape[1])
    xl, xh, fx = _axis(out_shape[2], src.shape[2])
    one = np.float32(1)

    def lerp_x(z, y):
        a = src[z][:, y][:, :, xl] if False else src[np.ix_(z, y, xl)]
        b = src[np.ix_(z, y, xh)]
        return a * (one - fx)[None, None, :] + b * fx[None, None, :]
    c00, c01, c10, c11 = lerp_x(zl, yl), lerp_x(zl, yh), lerp_x(zh, yl), lerp_x(zh, yh)
    fyb, fzb = fy[None, :, None], fz[:, None, None]
    return ((c00 * (one - fyb) + c01 * fyb) * (one - fzb) + (c10 * (one - fyb) + c11 * fyb) * fzb).astype(np.float32)


def mask_target(gt_masks, boxes, gt_inds, md, mh, mw):
    G, D, H, W = gt_masks.shape
    out = np.zeros((len(boxes), md, mh, mw), np.float32)
    for p, box in enumerate(boxes):
        xa, ya, za, w, h, d = _crop(box, W, H, D)
        if w <= 0 or h <= 0 or d <= 0:
            continue
        crop = (gt_masks[gt_inds[p], za:za + d, ya:ya + h, xa:xa + w] > 0)
        v = _resample(crop, (md, mh, mw))
        # uint8 0/1 crop -> img_as_float (x/255) -> resize -> *255 -> astype(uint8) -> > 0  ==  floor(coverage) > 0
        # (mask_target.py:41-44): only full coverage survives; 2^-16 of slack for the rounding of the lerps
        out[p] = (v >= np.float32(1.0 - 2.0 ** -16)).astype(np.float32)
    return out


def mask_paste(prob, boxes, thr, D, H, W):
    out = np.zeros((len(boxes), D, H, W), np.uint8)
    for p, box in enumerate(boxes):
        xa, ya, za, w, h, d = _crop(box, W, H, D)
        if w <= 0 or h <= 0 or d <= 0:
            continue
        v = _resample(prob[p], (d, h, w))
        out[p, za:za + d, ya:ya + h, xa:xa + w] = (v > np.float32(thr)).astype(np.uint8)
    return out

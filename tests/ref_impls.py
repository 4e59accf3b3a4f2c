"""Independent float64 re-derivations used to cross-check the C oracle (tests only).

Written from the math in SURVEY.md section 8 row A1/A2, not from oracle/mr3d_oracle.c.
"""
import numpy as np
import torch


def _axis_taps(coord, dim):
    """coord: float64 array of sample coordinates along one axis.
    returns (valid, low, high, l, h) following roi_align_kernel.cu:69-103."""
    valid = ~((coord < -1.0) | (coord > dim))
    c = np.where(coord <= 0, 0.0, coord)
    low = np.floor(c).astype(np.int64)
    clampd = low >= dim - 1
    low = np.where(clampd, dim - 1, low)
    high = np.where(clampd, dim - 1, low + 1)
    c = np.where(clampd, low.astype(np.float64), c)
    l = c - low
    return valid, low, high, l, 1.0 - l


def roi_sample_grid(roi, P, scale, sample_num, axis):
    """sample coordinates [P, s] for one axis (axis: 0=x,1=y,2=z)"""
    lo = {0: 1, 1: 2, 2: 5}[axis]
    hi = {0: 3, 1: 4, 2: 6}[axis]
    start = float(roi[lo]) * scale
    end = (float(roi[hi]) + 1.0) * scale
    ext = max(end - start, 0.0)
    b = ext / P
    s = sample_num if sample_num > 0 else int(np.ceil(ext / P))
    p = np.arange(P, dtype=np.float64)[:, None]
    i = np.arange(s, dtype=np.float64)[None, :]
    return start + p * b + (i + 0.5) * b / max(s, 1), s


def roi_align3d_f64(feat, rois, PD, PH, PW, scale, scale_d, sample_num):
    feat = np.asarray(feat, np.float64)
    N, C, D, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, PD, PH, PW), np.float64)
    for r in range(R):
        roi = np.asarray(rois[r], np.float32).astype(np.float64)
        b = int(roi[0])
        xs, sx = roi_sample_grid(roi, PW, float(np.float32(scale)), sample_num, 0)
        ys, sy = roi_sample_grid(roi, PH, float(np.float32(scale)), sample_num, 1)
        zs, sz = roi_sample_grid(roi, PD, float(np.float32(scale_d)), sample_num, 2)
        vx, xl, xh, lx, hx = _axis_taps(xs, W)
        vy, yl, yh, ly, hy = _axis_taps(ys, H)
        vz, zl, zh, lz, hz = _axis_taps(zs, D)
        f = feat[b]                                           # [C,D,H,W]
        acc = np.zeros((C, PD, PH, PW), np.float64)
        for (zi, wz) in ((zl, hz), (zh, lz)):
            for (yi, wy) in ((yl, hy), (yh, ly)):
                for (xi, wx) in ((xl, hx), (xh, lx)):
                    # gather f[:, zi[pd,iz], yi[ph,iy], xi[pw,ix]]
                    g = f[:, zi[:, :, None, None, None, None], yi[None, None, :, :, None, None], xi[None, None, None, None, :, :]]
                    w = (wz * vz)[:, :, None, None, None, None] * (wy * vy)[None, None, :, :, None, None] * (wx * vx)[None, None, None, None, :, :]
                    acc += (g * w).sum(axis=(2, 4, 6))
        out[r] = acc / max(sx * sy * sz, 1)
    return out


def roi_align3d_torch(feat, rois, PD, PH, PW, scale, scale_d, sample_num):
    """differentiable float64 torch restatement (autograd gives the CORRECT gradient)."""
    N, C, D, H, W = feat.shape
    outs = []
    for r in range(rois.shape[0]):
        roi = rois[r].detach().cpu().numpy().astype(np.float32).astype(np.float64)
        b = int(roi[0])
        xs, sx = roi_sample_grid(roi, PW, float(np.float32(scale)), sample_num, 0)
        ys, sy = roi_sample_grid(roi, PH, float(np.float32(scale)), sample_num, 1)
        zs, sz = roi_sample_grid(roi, PD, float(np.float32(scale_d)), sample_num, 2)
        vx, xl, xh, lx, hx = _axis_taps(xs, W)
        vy, yl, yh, ly, hy = _axis_taps(ys, H)
        vz, zl, zh, lz, hz = _axis_taps(zs, D)
        f = feat[b]
        acc = 0
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        for (zi, wz) in ((zl, hz), (zh, lz)):
            for (yi, wy) in ((yl, hy), (yh, ly)):
                for (xi, wx) in ((xl, hx), (xh, lx)):
                    g = f[:, T(zi)[:, :, None, None, None, None], T(yi)[None, None, :, :, None, None], T(xi)[None, None, None, None, :, :]]
                    w = T(wz * vz)[:, :, None, None, None, None] * T(wy * vy)[None, None, :, :, None, None] * T(wx * vx)[None, None, None, None, :, :]
                    acc = acc + (g * w).sum(dim=(2, 4, 6))
        outs.append(acc / max(sx * sy * sz, 1))
    return torch.stack(outs) if outs else feat.new_zeros((0, C, PD, PH, PW))


def bug_permute_top(top):
    """top'[n,c,pd,ph,pw] = top.flat[(n*C+c)*PD*PH*PW + pd*PD*PW + ph*PW + pw]
    (roi_align_kernel.cu:554-555): grad_bug(top) == grad_correct(top')."""
    R, C, PD, PH, PW = top.shape
    flat = top.reshape(R * C, PD * PH * PW)
    pd, ph, pw = np.meshgrid(np.arange(PD), np.arange(PH), np.arange(PW), indexing='ij')
    idx = (pd * PD * PW + ph * PW + pw).reshape(-1)
    return flat[:, idx].reshape(R, C, PD, PH, PW)


def rand_rois(rng, n, N, vol_whd, min_xy=4, max_xy=128, min_z=2, max_z=40):
    W, H, D = vol_whd
    sx = np.exp(rng.uniform(np.log(min_xy), np.log(max_xy), n))
    sy = np.exp(rng.uniform(np.log(min_xy), np.log(max_xy), n))
    sz = np.exp(rng.uniform(np.log(min_z), np.log(max_z), n))
    cx, cy, cz = rng.uniform(0, W, n), rng.uniform(0, H, n), rng.uniform(0, D, n)
    x1, x2 = np.clip(cx - sx / 2, 0, W - 1), np.clip(cx + sx / 2, 0, W - 1)
    y1, y2 = np.clip(cy - sy / 2, 0, H - 1), np.clip(cy + sy / 2, 0, H - 1)
    z1, z2 = np.clip(cz - sz / 2, 0, D - 1), np.clip(cz + sz / 2, 0, D - 1)
    b = rng.integers(0, N, n).astype(np.float64)
    return np.stack([b, x1, y1, x2, y2, z1, z2], 1).astype(np.float32)

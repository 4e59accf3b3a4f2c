"""Restatements for DeformConv3d (TEST INFRASTRUCTURE).

* deform_conv2d_reference_forward: numpy restatement of the REFERENCE's 2D operator, following
  mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:83-113 (deformable_im2col_bilinear) and :189-242
  (deformable_im2col_gpu_kernel) + the GEMM of deform_conv_cuda.cpp:229-234.
* deform_conv3d_torch: differentiable float64 torch restatement of this build's 3D extension (same sampling rule
  per axis); autograd through it gives the gradients w.r.t. input, offsets, weight and mask."""
import numpy as np
import torch


def deform_conv2d_reference_forward(x, offset, weight, stride, pad, dil, dg):
    """x [N,C,H,W], offset [N, dg*2*kh*kw, Ho, Wo] ((dh, dw) interleaved per tap), weight [Co,C,kh,kw] -> [N,Co,Ho,Wo]"""
    x, offset, weight = (np.asarray(a, np.float64) for a in (x, offset, weight))
    N, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    cpg = C // dg
    cols = np.zeros((N, C, kh, kw, Ho, Wo))
    for n in range(N):
        for c in range(C):
            g = c // cpg
            for i in range(kh):
                for j in range(kw):
                    oh = offset[n, g * 2 * kh * kw + 2 * (i * kw + j)]
                    ow = offset[n, g * 2 * kh * kw + 2 * (i * kw + j) + 1]
                    for ho in range(Ho):
                        for wo in range(Wo):
                            h = ho * stride[0] - pad[0] + i * dil[0] + oh[ho, wo]
                            w = wo * stride[1] - pad[1] + j * dil[1] + ow[ho, wo]
                            if not (h > -1 and w > -1 and h < H and w < W):
                                continue
                            hl, wl = int(np.floor(h)), int(np.floor(w))
                            hh_, wh_ = hl + 1, wl + 1
                            lh, lw = h - hl, w - wl
                            v1 = x[n, c, hl, wl] if (hl >= 0 and wl >= 0) else 0.0
                            v2 = x[n, c, hl, wh_] if (hl >= 0 and wh_ <= W - 1) else 0.0
                            v3 = x[n, c, hh_, wl] if (hh_ <= H - 1 and wl >= 0) else 0.0
                            v4 = x[n, c, hh_, wh_] if (hh_ <= H - 1 and wh_ <= W - 1) else 0.0
                            cols[n, c, i, j, ho, wo] = (1 - lh) * (1 - lw) * v1 + (1 - lh) * lw * v2 + lh * (1 - lw) * v3 + lh * lw * v4
    return np.einsum('ocij,ncijhw->nohw', weight, cols)


def deform_conv3d_torch(x, offset, weight, mask=None, bias=None, stride=(1, 1, 1), pad=(0, 0, 0), dil=(1, 1, 1), dg=1):
    """x [N,C,D,H,W], offset [N, dg*3*K, Do,Ho,Wo] ((dd,dh,dw) per tap), weight [Co,C,kd,kh,kw], mask [N, dg*K, ...]"""
    N, C, D, H, W = x.shape
    Co, _, kd, kh, kw = weight.shape
    K = kd * kh * kw
    Do = (D + 2 * pad[0] - (dil[0] * (kd - 1) + 1)) // stride[0] + 1
    Ho = (H + 2 * pad[1] - (dil[1] * (kh - 1) + 1)) // stride[1] + 1
    Wo = (W + 2 * pad[2] - (dil[2] * (kw - 1) + 1)) // stride[2] + 1
    cpg = C // dg
    dt = x.dtype
    od = torch.arange(Do, dtype=dt).view(1, Do, 1, 1) * stride[0] - pad[0]
    oh = torch.arange(Ho, dtype=dt).view(1, 1, Ho, 1) * stride[1] - pad[1]
    ow = torch.arange(Wo, dtype=dt).view(1, 1, 1, Wo) * stride[2] - pad[2]
    nidx = torch.arange(N).view(N, 1, 1, 1, 1)
    out = x.new_zeros((N, Co, Do, Ho, Wo))
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg]                      # [N, cpg, D, H, W]
        for t in range(K):
            a, b, c = t // (kh * kw), (t // kw) % kh, t % kw
            base = (g * K + t) * 3
            pz = od + a * dil[0] + offset[:, base]
            py = oh + b * dil[1] + offset[:, base + 1]
            px = ow + c * dil[2] + offset[:, base + 2]
            inside = (pz > -1) & (py > -1) & (px > -1) & (pz < D) & (py < H) & (px < W)
            z0, y0, x0 = torch.floor(pz), torch.floor(py), torch.floor(px)
            lz, ly, lx = pz - z0, py - y0, px - x0
            val = 0
            for dz_, wz in ((0, 1 - lz), (1, lz)):
                for dy_, wy in ((0, 1 - ly), (1, ly)):
                    for dx_, wx in ((0, 1 - lx), (1, lx)):
                        zi, yi, xi = (z0 + dz_).long(), (y0 + dy_).long(), (x0 + dx_).long()
                        ok = inside & (zi >= 0) & (zi <= D - 1) & (yi >= 0) & (yi <= H - 1) & (xi >= 0) & (xi <= W - 1)
                        zc, yc, xc = zi.clamp(0, D - 1), yi.clamp(0, H - 1), xi.clamp(0, W - 1)
                        v = xg.permute(0, 2, 3, 4, 1)[nidx.view(N, 1, 1, 1), zc, yc, xc]     # [N,Do,Ho,Wo,cpg]
                        val = val + (wz * wy * wx * ok.to(dt)).unsqueeze(-1) * v
            if mask is not None:
                val = val * mask[:, g * K + t].unsqueeze(-1)
            wt = weight[:, g * cpg:(g + 1) * cpg, a, b, c]       # [Co, cpg]
            out = out + torch.einsum('ndhwc,oc->nodhw', val, wt)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1, 1)
    return out

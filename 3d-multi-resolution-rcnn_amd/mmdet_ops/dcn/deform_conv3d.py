"""DeformConv3d / ModulatedDeformConv3d: the 3D extension of the reference's 2D `DeformConv` /
`ModulatedDeformConv` (mmdet/ops/dcn/functions/deform_conv.py:8-181, modules/deform_conv.py:10-56).

Call signatures follow the 2D modules with 3-tuples in place of pairs:
    DeformConv3d(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False)(x, offset)
    offset: [N, deformable_groups * 3 * kd*kh*kw, Do, Ho, Wo], per tap the channels (dd, dh, dw)
            (the 2D layout's (dh, dw) pairs, deform_conv_cuda_kernel.cu:221-223, with a leading depth component)
    ModulatedDeformConv3d(...)(x, offset, mask), mask: [N, deformable_groups * kd*kh*kw, Do, Ho, Wo]
autograd: backward returns (grad_input, grad_offset, grad_weight[, grad_mask, grad_bias]) like the reference
(functions/deform_conv.py:88-89).  `groups > 1` is not built (the reference's 3D path never instantiates it)."""
import ctypes
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _triple

from ... import _hip


class _Desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ('n', 'd', 'h', 'w', 'c', 'kd', 'kh', 'kw', 'sd', 'sh', 'sw', 'pd', 'ph', 'pw',
                                            'dil_d', 'dil_h', 'dil_w', 'dg')] + \
               [('pos0', ctypes.c_longlong), ('npos', ctypes.c_longlong)]


_CHUNK = 32768          # output positions per column chunk (the reference's im2col_step idea)


def _desc(xs, k, stride, pad, dil, dg, pos0, npos):
    d = _Desc()
    d.n, d.d, d.h, d.w, d.c = xs
    d.kd, d.kh, d.kw = k
    d.sd, d.sh, d.sw = stride
    d.pd, d.ph, d.pw = pad
    d.dil_d, d.dil_h, d.dil_w = dil
    d.dg, d.pos0, d.npos = dg, pos0, npos
    return d


def _out_size(xs, k, stride, pad, dil):
    return tuple((xs[1 + i] + 2 * pad[i] - (dil[i] * (k[i] - 1) + 1)) // stride[i] + 1 for i in range(3))


def _cl(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def _fused_desc(xc_shape, k, stride, pad, dil, dg, co, has_mask, has_bias):
    """the descriptor of the fused (column-free) kernels if they take this shape (csrc/deform_conv3d_fused.hip), else None"""
    if has_mask or has_bias:
        return None
    d = _desc(xc_shape, k, stride, pad, dil, dg, 0, 1)
    return d if _hip.lib().mr3d_deform_conv3d_fused_supported(ctypes.byref(d), int(co)) == 1 else None


class DeformConvFunction3D(Function):

    @staticmethod
    def forward(ctx, x, offset, weight, mask=None, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
        if x is not None and x.dim() != 5:
            raise ValueError(f'Expected 5D tensor as input, got {x.dim()}D tensor instead.')      # cf. deform_conv.py:21-24
        if not x.is_cuda:
            raise NotImplementedError                                                                # deform_conv.py:40-41
        if groups != 1:
            raise NotImplementedError('DeformConv3d is built for groups == 1')
        stride, padding, dilation = _triple(stride), _triple(padding), _triple(dilation)
        co, ci = weight.shape[:2]
        k = tuple(weight.shape[2:])
        K = k[0] * k[1] * k[2]
        assert (K * ci) % 16 == 0 and ci % deformable_groups == 0, 'taps * Cin must be a multiple of 16'
        xc = _cl(x.detach())
        N, D, H, W, C = xc.shape
        osz = _out_size(xc.shape, k, stride, padding, dilation)
        assert tuple(offset.shape) == (N, deformable_groups * 3 * K) + osz, 'offset shape'
        oc = _cl(offset.detach())
        mc = _cl(mask.detach()) if mask is not None else None
        fd = _fused_desc(xc.shape, k, stride, padding, dilation, deformable_groups, co, mask is not None, bias is not None)
        if fd is not None:
            L = _hip.lib()
            total = N * osz[0] * osz[1] * osz[2]
            w = weight.detach().contiguous()
            packed = torch.empty((3 * K * ci * co + 1,), dtype=torch.float32, device=x.device)
            y = torch.empty((total, co), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                st = _hip.stream_ptr(x.device)
                _hip.check(L.mr3d_deform_conv3d_pack_weights(_hip.ptr(w), co, ci, K, _hip.ptr(packed), st),
                           'mr3d_deform_conv3d_pack_weights')
                _hip.check(L.mr3d_deform_conv3d_forward_fused(ctypes.byref(fd), co, _hip.ptr(xc), _hip.ptr(oc), _hip.ptr(packed),
                                                              _hip.ptr(y), st), 'mr3d_deform_conv3d_forward_fused')
            ctx.save_for_backward(xc, oc, None, packed)
            ctx.meta = (k, stride, padding, dilation, deformable_groups, co, co, ci, osz, False, False)
            ctx.fused = True
            return y.view(N, *osz, co).permute(0, 4, 1, 2, 3)
        ctx.fused = False
        cop = -(-co // 16) * 16
        wp = weight.detach().reshape(co, ci, K).permute(2, 1, 0).reshape(1, K * ci, co)          # packed [1][K*Cin][Cout]
        if cop != co:
            wp = torch.nn.functional.pad(wp, (0, cop - co))
        wp = wp.contiguous()
        shift = None
        if bias is not None:
            shift = torch.nn.functional.pad(bias.detach(), (0, cop - co)).contiguous()
        total = N * osz[0] * osz[1] * osz[2]
        y = torch.empty((total, cop), dtype=torch.float32, device=x.device)
        L = _hip.lib()
        with torch.cuda.device(x.device):
            for p0 in range(0, total, _CHUNK):
                npos = min(_CHUNK, total - p0)
                rows = -(-npos // 64) * 64
                col = torch.zeros((rows, K * ci), dtype=torch.float32, device=x.device)
                d = _desc(xc.shape, k, stride, padding, dilation, deformable_groups, p0, npos)
                _hip.check(L.mr3d_deform_im2col3d(ctypes.byref(d), _hip.ptr(xc), _hip.ptr(oc), ctypes.c_void_p(_hip._opt(mc)),
                                                  _hip.ptr(col), _hip.stream_ptr(x.device)), 'mr3d_deform_im2col3d')
                yc = _hip.conv3d_forward(col.view(1, rows // 64, 8, 8, K * ci), wp, (1, 1, 1), (1, 1, 1), (0, 0, 0), shift=shift)
                y[p0:p0 + npos] = yc.view(rows, cop)[:npos]
        ctx.save_for_backward(xc, oc, mc, wp)
        ctx.meta = (k, stride, padding, dilation, deformable_groups, co, cop, ci, osz, bias is not None, mask is not None)
        out = y.view(N, *osz, cop).permute(0, 4, 1, 2, 3)
        return out[:, :co] if cop != co else out

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, oc, mc, wp = ctx.saved_tensors
        k, stride, padding, dilation, dg, co, cop, ci, osz, has_bias, has_mask = ctx.meta
        K = k[0] * k[1] * k[2]
        N = xc.size(0)
        total = N * osz[0] * osz[1] * osz[2]
        g = gy.permute(0, 2, 3, 4, 1).reshape(total, co)
        if ctx.fused:
            g = g.contiguous()
            fd = _desc(xc.shape, k, stride, padding, dilation, dg, 0, 1)
            gx = torch.zeros_like(xc)
            goff = torch.zeros_like(oc)
            dwp = torch.zeros((K, ci, co), dtype=torch.float32, device=g.device)
            L = _hip.lib()
            with torch.cuda.device(g.device):
                st = _hip.stream_ptr(g.device)
                _hip.check(L.mr3d_deform_conv3d_wgrad_fused(ctypes.byref(fd), co, _hip.ptr(xc), _hip.ptr(oc), _hip.ptr(g),
                                                            _hip.ptr(dwp), st), 'mr3d_deform_conv3d_wgrad_fused')
                _hip.check(L.mr3d_deform_conv3d_dgrad_fused(ctypes.byref(fd), co, _hip.ptr(xc), _hip.ptr(oc), _hip.ptr(g),
                                                            _hip.ptr(wp), _hip.ptr(gx), _hip.ptr(goff), st),
                           'mr3d_deform_conv3d_dgrad_fused')
            gw = dwp.permute(2, 1, 0).reshape(co, ci, *k).contiguous()
            return (gx.permute(0, 4, 1, 2, 3), goff.permute(0, 4, 1, 2, 3), gw, None, None, None, None, None, None, None)
        if cop != co:
            g = torch.nn.functional.pad(g, (0, cop - co))
        g = g.contiguous()
        gx = torch.zeros_like(xc)
        goff = torch.empty_like(oc)
        gmask = torch.empty_like(mc) if has_mask else None
        dwp = torch.zeros_like(wp)
        db = torch.zeros((cop,), dtype=torch.float32, device=g.device)
        wpt = wp.view(K * ci, cop).t().contiguous().view(1, cop, K * ci)                            # dcol = dy x Wpacked^T
        L = _hip.lib()
        with torch.cuda.device(g.device):
            for p0 in range(0, total, _CHUNK):
                npos = min(_CHUNK, total - p0)
                rows = -(-npos // 64) * 64
                d = _desc(xc.shape, k, stride, padding, dilation, dg, p0, npos)
                col = torch.zeros((rows, K * ci), dtype=torch.float32, device=g.device)
                _hip.check(L.mr3d_deform_im2col3d(ctypes.byref(d), _hip.ptr(xc), _hip.ptr(oc), ctypes.c_void_p(_hip._opt(mc)),
                                                  _hip.ptr(col), _hip.stream_ptr(g.device)), 'mr3d_deform_im2col3d')
                gc = torch.zeros((rows, cop), dtype=torch.float32, device=g.device)
                gc[:npos] = g[p0:p0 + npos]
                g5 = gc.view(1, rows // 64, 8, 8, cop)
                _hip.conv3d_wgrad(col.view(1, rows // 64, 8, 8, K * ci), g5, dwp, (1, 1, 1), (1, 1, 1), (0, 0, 0), dbias=db)
                dcol = _hip.conv3d_forward(g5, wpt, (1, 1, 1), (1, 1, 1), (0, 0, 0))
                _hip.check(L.mr3d_deform_col2im3d(ctypes.byref(d), _hip.ptr(xc), _hip.ptr(oc), ctypes.c_void_p(_hip._opt(mc)),
                                                  _hip.ptr(dcol), _hip.ptr(gx), _hip.ptr(goff),
                                                  ctypes.c_void_p(_hip._opt(gmask)), _hip.stream_ptr(g.device)),
                           'mr3d_deform_col2im3d')
        gw = dwp.view(K, ci, cop).permute(2, 1, 0).reshape(cop, ci, *k)[:co].contiguous()
        return (gx.permute(0, 4, 1, 2, 3), goff.permute(0, 4, 1, 2, 3), gw,
                gmask.permute(0, 4, 1, 2, 3) if has_mask else None, db[:co].contiguous() if has_bias else None,
                None, None, None, None, None)


class DeformConvBf16Function3D(Function):
    """DeformConv3d on bf16 maps (BASELINE configs[4]; csrc/deform_conv3d_bf16.hip): x16 [N,D,H,W,64] and off16 [N,Do,Ho,Wo,S]
    (bf16, NDHWC; the first 3 * taps entries of an offset row are the taps' (dd, dh, dw), S >= 3 * taps -- the offset conv's
    padded output is taken as it is), weight [64, 64, kd, kh, kw] fp32 -> y16 [N,Do,Ho,Wo,64] bf16.  backward returns
    (grad x bf16, grad offsets bf16 with the same row stride and zero padding, grad weight fp32)."""

    @staticmethod
    def forward(ctx, x16, off16, weight, stride, padding, dilation):
        assert x16.is_cuda and x16.dtype == torch.bfloat16 and off16.dtype == torch.bfloat16 and x16.is_contiguous() and off16.is_contiguous()
        stride, padding, dilation = _triple(stride), _triple(padding), _triple(dilation)
        co, ci = weight.shape[:2]
        k = tuple(weight.shape[2:])
        K = k[0] * k[1] * k[2]
        N = x16.size(0)
        osz = _out_size(x16.shape, k, stride, padding, dilation)
        assert tuple(off16.shape[:4]) == (N,) + osz and off16.size(4) >= 3 * K, 'offset shape'
        d = _desc(x16.shape, k, stride, padding, dilation, 1, 0, 1)
        L = _hip.lib()
        if L.mr3d_dcn_bf16_supported(ctypes.byref(d), int(co)) != 1:
            raise NotImplementedError('the bf16 DeformConv3d kernels take Cin = Cout = 64 and a sample box of <= 768 voxels')
        w = weight.detach()
        y16 = torch.empty((N,) + osz + (co,), dtype=torch.bfloat16, device=x16.device)
        wp = _hip.pack_conv_weight_bf16(w)
        with torch.cuda.device(x16.device):
            _hip.check(L.mr3d_dcn_bf16_forward(ctypes.byref(d), co, _hip.ptr(x16), _hip.ptr(off16), int(off16.size(4)), _hip.ptr(wp),
                                               _hip.ptr(y16), _hip.stream_ptr(x16.device)), 'mr3d_dcn_bf16_forward')
        ctx.save_for_backward(x16, off16, w)
        ctx.meta = (k, stride, padding, dilation)
        return y16

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x16, off16, w = ctx.saved_tensors
        k, stride, padding, dilation = ctx.meta
        co, ci = w.shape[:2]
        K = k[0] * k[1] * k[2]
        g16 = gy.contiguous()
        assert g16.dtype == torch.bfloat16
        d = _desc(x16.shape, k, stride, padding, dilation, 1, 0, 1)
        S = int(off16.size(4))
        L = _hip.lib()
        dev = x16.device
        dwp = torch.zeros((K, ci, co), dtype=torch.float32, device=dev)
        goff = torch.zeros_like(off16)
        gx = torch.zeros(x16.shape, dtype=torch.float32, device=dev)
        wt = _hip.pack_conv_weight_bf16(w.transpose(0, 1))
        packed = torch.empty((3 * K * ci * co + 1,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _hip.stream_ptr(dev)
            _hip.check(L.mr3d_dcn_bf16_wgrad(ctypes.byref(d), co, _hip.ptr(x16), _hip.ptr(off16), S, _hip.ptr(g16), _hip.ptr(dwp), st),
                       'mr3d_dcn_bf16_wgrad')
            _hip.check(L.mr3d_dcn_bf16_doffset(ctypes.byref(d), co, _hip.ptr(x16), _hip.ptr(off16), S, _hip.ptr(g16), _hip.ptr(wt),
                                               _hip.ptr(goff), S, st), 'mr3d_dcn_bf16_doffset')
            _hip.check(L.mr3d_deform_conv3d_pack_weights(_hip.ptr(w.contiguous()), co, ci, K, _hip.ptr(packed), st),
                       'mr3d_deform_conv3d_pack_weights')
            _hip.check(L.mr3d_dcn_bf16_dx(ctypes.byref(d), co, _hip.ptr(off16), S, _hip.ptr(g16), _hip.ptr(packed), _hip.ptr(gx), st),
                       'mr3d_dcn_bf16_dx')
        gw = dwp.permute(2, 1, 0).reshape(co, ci, *k).contiguous()
        return gx.to(torch.bfloat16), goff, gw, None, None, None


def deform_conv3d_bf16(x16, off16, weight, stride=1, padding=0, dilation=1):
    return DeformConvBf16Function3D.apply(x16, off16, weight, stride, padding, dilation)


def deform_conv3d(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    return DeformConvFunction3D.apply(x, offset, weight, None, None, stride, padding, dilation, groups, deformable_groups)


class DeformConv3d(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        assert not bias                                                        # modules/deform_conv.py:22
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _triple(kernel_size), _triple(stride)
        self.padding, self.dilation = _triple(padding), _triple(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)                                 # modules/deform_conv.py:44-49

    def forward(self, x, offset):
        return deform_conv3d(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                             self.deformable_groups)


class ModulatedDeformConv3d(DeformConv3d):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, False)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, offset, mask):
        return DeformConvFunction3D.apply(x, offset, self.weight, mask, self.bias, self.stride, self.padding, self.dilation,
                                          self.groups, self.deformable_groups)

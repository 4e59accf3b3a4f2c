"""Autograd-aware layers on the HIP conv engine, used by the heads (RPN / bbox / mask).

`conv3d` = stride-1 Conv3d (+ bias, + optional fused ReLU) whose forward, data gradient and weight gradient all
run on the fp32-MFMA kernels of csrc/conv3d_fwd.hip / conv3d_wgrad.hip (pointwise shapes take the streaming
kernels of csrc/pointwise.hip).  Tensors keep the reference's logical [N, C, D, H, W] shape; physically they are
NDHWC (torch.channels_last_3d), so chaining these layers never transposes.  `linear` and `conv_transpose3d_k2s2`
are expressed through the same kernel: a Linear is a pointwise conv over rows, a kernel-2/stride-2 transposed conv
is a pointwise conv to 8*Cout channels followed by a pixel shuffle."""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import ctypes

from .. import _hip

def _ndhwc(x):
    """[N,C,D,H,W] (any strides) -> contiguous [N,D,H,W,C] (zero-copy when x is channels_last_3d)"""
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _pad_to(n, m):
    return -(-n // m) * m


class _Conv3dFn(Function):

    @staticmethod
    def forward(ctx, x, weight, bias, padding, relu):
        if not x.is_cuda:
            raise NotImplementedError('the HIP conv engine has no CPU fallback')
        co, ci, k = weight.shape[0], weight.shape[1], weight.shape[2]
        assert weight.shape[2] == weight.shape[3] == weight.shape[4] and ci % 16 == 0, 'cubic kernel, Cin % 16 == 0'
        cop = _pad_to(co, 16)
        w = weight.detach()
        if cop != co:
            w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, 0, 0, cop - co))
        shift = None
        if bias is not None:
            shift = F.pad(bias.detach(), (0, cop - co)) if cop != co else bias.detach().contiguous()
        xc = _ndhwc(x.detach())
        y = _hip.conv3d_forward(xc, _hip.pack_conv_weight(w), (k,) * 3, (1, 1, 1), (padding,) * 3, shift=shift, relu=relu)
        ctx.save_for_backward(xc, w, y if relu else None)
        ctx.meta = (co, cop, k, padding, relu, bias is not None)
        out = y.permute(0, 4, 1, 2, 3)
        return out[:, :co] if cop != co else out

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, w, y = ctx.saved_tensors
        co, cop, k, padding, relu, has_bias = ctx.meta
        g = gy.permute(0, 2, 3, 4, 1)
        if cop != co:
            gp = torch.zeros(g.shape[:4] + (cop,), dtype=torch.float32, device=g.device)
            gp[..., :co] = g
            g = gp
        else:
            g = g.contiguous()
        if relu:
            g = _hip.relu_mask(g.contiguous(), y)
        taps, ci = k ** 3, xc.size(4)
        gw = gb = gx = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dwp = torch.zeros((taps, ci, cop), dtype=torch.float32, device=g.device)
            db = torch.zeros((cop,), dtype=torch.float32, device=g.device)
            _hip.conv3d_wgrad(xc, g, dwp, (k,) * 3, (1, 1, 1), (padding,) * 3, dbias=db)
            gw = dwp.permute(2, 1, 0).reshape(cop, ci, k, k, k)[:co].contiguous()
            gb = db[:co].contiguous() if has_bias else None
        if ctx.needs_input_grad[0]:
            # dgrad = stride-1 conv of dy with flipped taps and swapped channel roles: wb[taps-1-t][co][ci]
            wb = w.reshape(cop, ci, taps).flip(2).permute(2, 0, 1).contiguous()
            gxc = _hip.conv3d_forward(g, wb, (k,) * 3, (1, 1, 1), (k - 1 - padding,) * 3)
            gx = gxc.permute(0, 4, 1, 2, 3)
        return gx, gw, gb, None, None


class _Conv3dBf16Fn(Function):
    """stride-1 cubic Conv3d (+bias) with bf16 activation storage (BASELINE configs[4]): x and the result are rounded to bf16,
    products accumulate in fp32 on the bf16 MFMA kernels (csrc/conv3d_bf16.hip, conv3d_wgrad_bf16.hip); parameters and their
    gradients stay fp32.  Output channels are padded to a multiple of 32 inside (the kernels' N tile)."""

    @staticmethod
    def forward(ctx, x, weight, bias, padding):
        if not x.is_cuda:
            raise NotImplementedError('the HIP conv engine has no CPU fallback')
        co, ci, k = weight.shape[0], weight.shape[1], weight.shape[2]
        assert weight.shape[2] == weight.shape[3] == weight.shape[4] and ci % 16 == 0, 'cubic kernel, Cin % 16 == 0'
        cop = _pad_to(co, 32)
        w = weight.detach()
        if cop != co:
            w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, 0, 0, cop - co))
        shift = None
        if bias is not None:
            shift = F.pad(bias.detach().float(), (0, cop - co)).contiguous()
        x16 = _ndhwc(x.detach()).to(torch.bfloat16)
        y16 = _hip.conv3d_forward_bf16(x16, _hip.pack_conv_weight_bf16(w), cop, (k,) * 3, (1, 1, 1), (padding,) * 3, shift=shift)
        ctx.save_for_backward(x16, w)
        ctx.meta = (co, cop, k, padding, bias is not None)
        return y16.permute(0, 4, 1, 2, 3)[:, :co].float()

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x16, w = ctx.saved_tensors
        co, cop, k, padding, has_bias = ctx.meta
        g = gy.permute(0, 2, 3, 4, 1)
        g16 = torch.zeros(g.shape[:4] + (cop,), dtype=torch.bfloat16, device=g.device)
        g16[..., :co] = g
        taps, ci = k ** 3, x16.size(4)
        gw = gb = gx = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dwp = torch.zeros((taps, ci, cop), dtype=torch.float32, device=g.device)
            db = torch.zeros((cop,), dtype=torch.float32, device=g.device)
            _hip.conv3d_wgrad_bf16(x16, g16, dwp, (k,) * 3, (1, 1, 1), (padding,) * 3, dbias=db)
            gw = dwp.permute(2, 1, 0).reshape(cop, ci, k, k, k)[:co].contiguous()
            gb = db[:co].contiguous() if has_bias else None
        if ctx.needs_input_grad[0]:
            wb16 = _hip.pack_conv_weight_bf16(w.flip(2, 3, 4).transpose(0, 1))          # [ci][cop][k,k,k]: dgrad operand
            gx16 = _hip.conv3d_forward_bf16(g16, wb16, _pad_to(ci, 32), (k,) * 3, (1, 1, 1), (k - 1 - padding,) * 3)
            gx = gx16.permute(0, 4, 1, 2, 3)[:, :ci].float()
        return gx, gw, gb, None


def conv3d_bf16(x, weight, bias=None, padding=0):
    """conv3d with bf16 activation storage (configs[4]); fp32 in / out at the Python boundary"""
    return _Conv3dBf16Fn.apply(x, weight, bias, int(padding))


def conv3d(x, weight, bias=None, padding=0, relu=False):
    """stride-1 cubic Conv3d (+bias, +ReLU); x [N,Cin,D,H,W] -> [N,Cout,D',H',W'] (channels_last_3d view)"""
    return _Conv3dFn.apply(x, weight, bias, int(padding), bool(relu))


_LINEAR_WS = {}


def _linear_workspace(device, rows, k, o):
    """scratch for the split-K launches of csrc/linear.hip (fixed-order summation instead of atomics): one buffer per
    (device, stream), grown on demand -- launches on one stream use it one after the other"""
    fn = _hip.lib().mr3d_linear_workspace_bytes
    if fn.restype is not ctypes.c_size_t:
        fn.restype = ctypes.c_size_t
    need = int(fn(int(rows), int(k), int(o)))
    if need == 0:
        return None, 0
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _LINEAR_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=device)
        _LINEAR_WS[key] = ws
    return ws, ws.numel() * 4


class _LinearFn(Function):
    """torch.nn.Linear (+ReLU) on the fp32 MFMA GEMM of csrc/linear.hip: forward, data gradient and weight gradient read x, dy
    and weight in their torch layouts"""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        if not x.is_cuda:
            raise NotImplementedError('the HIP linear layer has no CPU fallback')
        xc = x.detach().contiguous().float()
        w = weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        R, K = xc.shape
        O = w.shape[0]
        y = torch.empty((R, O), dtype=torch.float32, device=x.device)
        if R > 0:
            ws, wsb = _linear_workspace(x.device, R, K, O)
            with torch.cuda.device(x.device):
                _hip.check(_hip.lib().mr3d_linear_forward(_hip.ptr(xc), _hip.ptr(w), ctypes.c_void_p(_hip._opt(b)), _hip.ptr(y),
                                                          R, K, O, int(relu), ctypes.c_void_p(_hip._opt(ws)), ctypes.c_longlong(wsb),
                                                          _hip.stream_ptr(x.device)), 'mr3d_linear_forward')
        ctx.save_for_backward(xc, w, y if relu else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xc, w, y = ctx.saved_tensors
        R, K = xc.shape
        O = w.shape[0]
        g = g.contiguous().float()
        if y is not None:
            g = g * (y > 0)
        gx = gw = gb = None
        L = _hip.lib()
        ws, wsb = _linear_workspace(g.device, R, K, O) if R > 0 else (None, 0)
        wsp, wsl = ctypes.c_void_p(_hip._opt(ws)), ctypes.c_longlong(wsb)
        with torch.cuda.device(g.device):
            st = _hip.stream_ptr(g.device)
            if ctx.needs_input_grad[0]:
                gx = torch.zeros((R, K), dtype=torch.float32, device=g.device) if R == 0 else torch.empty((R, K), dtype=torch.float32, device=g.device)
                if R > 0:
                    _hip.check(L.mr3d_linear_dgrad(_hip.ptr(g), _hip.ptr(w), _hip.ptr(gx), R, K, O, wsp, wsl, st), 'mr3d_linear_dgrad')
            if ctx.needs_input_grad[1]:
                gw = torch.zeros((O, K), dtype=torch.float32, device=g.device)
                if R > 0:
                    _hip.check(L.mr3d_linear_wgrad(_hip.ptr(g), _hip.ptr(xc), _hip.ptr(gw), R, K, O, wsp, wsl, st), 'mr3d_linear_wgrad')
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb, None


def linear(x, weight, bias=None, relu=False, impl=None):
    """y = x @ weight.T + bias for x [R, K] -- torch.nn.Linear of the reference's bbox heads
    (convfc_bbox_head_3d.py:104-140).  Default: the hand-written fp32 MFMA GEMM of csrc/linear.hip (forward / dgrad / wgrad,
    split-K where the output has few tiles).  impl='conv': the same product through the pointwise-conv kernels (tested)."""
    if impl != 'conv':
        return _LinearFn.apply(x, weight, bias, bool(relu))
    R, K = x.shape
    rp = _pad_to(max(R, 1), 64)
    xp = x if rp == R else F.pad(x, (0, 0, 0, rp - R))
    x5 = xp.reshape(1, rp // 64, 8, 8, K).permute(0, 4, 1, 2, 3)          # logical [1, K, D, H, W], physically NDHWC
    y5 = conv3d(x5, weight.reshape(weight.shape[0], K, 1, 1, 1), bias, 0, relu)
    return y5.permute(0, 2, 3, 4, 1).reshape(rp, weight.shape[0])[:R]


class _Deconv2Fn(Function):
    """ConvTranspose3d(kernel 2, stride 2): windows do not overlap, so
       forward  = 8 pointwise GEMMs (one per tap) whose epilogue scatters straight to (2d+a, 2h+b, 2w+c) -- no
                  8*Cout intermediate and no pixel-shuffle copy;
       dgrad    = a kernel-2 / stride-2 convolution of the output gradient (implicit GEMM, K = 8 taps x Cout);
       wgrad    = the weight gradient of that same strided convolution with the roles (input = dy, output grad = x)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        if not x.is_cuda:
            raise NotImplementedError('the HIP conv engine has no CPU fallback')
        ci, co = weight.shape[0], weight.shape[1]
        assert tuple(weight.shape[2:]) == (2, 2, 2) and ci % 16 == 0 and co % 64 == 0
        xc = _ndhwc(x.detach())
        N, D, H, W, _ = xc.shape
        w = weight.detach()
        wf = w.permute(2, 3, 4, 0, 1).reshape(8, ci, co).contiguous()                 # [tap][ci][co]
        shift = bias.detach().contiguous() if bias is not None else None
        y = torch.empty((N, 2 * D, 2 * H, 2 * W, co), dtype=torch.float32, device=x.device)
        for t in range(8):
            a, b, c = t >> 2, (t >> 1) & 1, t & 1
            _hip.conv3d_forward(xc, wf[t:t + 1], (1, 1, 1), (1, 1, 1), (0, 0, 0), shift=shift, relu=relu, out=y,
                                out_stride=(2, 2, 2), out_offset=((a * 2 * H + b) * 2 * W + c) * co)
        ctx.save_for_backward(xc, w, y if relu else None)
        ctx.meta = (relu, bias is not None)
        return y.permute(0, 4, 1, 2, 3)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, w, y = ctx.saved_tensors
        relu, has_bias = ctx.meta
        ci, co = w.shape[0], w.shape[1]
        g = gy.permute(0, 2, 3, 4, 1).contiguous()
        want_gb = has_bias and ctx.needs_input_grad[2]
        gb_fused = None
        if relu:
            if want_gb:
                g, gb_fused = _hip.relu_mask(g, y, want_bias_grad=True)     # mask + bias gradient in one pass
            else:
                g = _hip.relu_mask(g, y)
        gx = gw = gb = None
        if ctx.needs_input_grad[1]:
            dwp = torch.zeros((8, co, ci), dtype=torch.float32, device=g.device)       # [tap][conv-in = co][conv-out = ci]
            _hip.conv3d_wgrad(g, xc, dwp, (2, 2, 2), (2, 2, 2), (0, 0, 0))
            gw = dwp.reshape(2, 2, 2, co, ci).permute(4, 3, 0, 1, 2).contiguous()
        if want_gb:
            gb = gb_fused if gb_fused is not None else g.sum((0, 1, 2, 3))
        if ctx.needs_input_grad[0]:
            wb = w.permute(2, 3, 4, 1, 0).reshape(8, co, ci).contiguous()
            gx = _hip.conv3d_forward(g, wb, (2, 2, 2), (2, 2, 2), (0, 0, 0)).permute(0, 4, 1, 2, 3)
        return gx, gw, gb, None


def conv_transpose3d_k2s2(x, weight, bias=None, relu=False):
    """ConvTranspose3d(kernel 2, stride 2): weight [Cin, Cout, 2, 2, 2] (torch layout).
    out[n, co, 2d+a, 2h+b, 2w+c] = sum_ci x[n,ci,d,h,w] * weight[ci,co,a,b,c] + bias[co]"""
    return _Deconv2Fn.apply(x, weight, bias, bool(relu))

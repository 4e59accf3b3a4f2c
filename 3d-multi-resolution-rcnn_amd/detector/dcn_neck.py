"""Deformable FPN output stage -- BASELINE.json configs[4] "DeformConv3d neck enabled" (build-defined: the reference's
DCN is 2D only and its 3D neck never instantiates it).

Slot: the four FPN output convolutions `fpn_convs[i]` (mmdet/models/necks/fpn3d.py:56-64, 3x3x3 64 -> 64, applied to the
top-down lateral sums at :111-113).  Each becomes the reference's DCN pair (mmdet/models/backbones/resnet.py `conv2_offset`
+ `conv2`, the 2D template): a plain 3x3x3 conv predicting 3 * 27 * deformable_groups offsets from the lateral sum, and a
DeformConv3d (mmdet_ops.dcn, no bias -- modules/deform_conv.py:22) sampling the same map at the offset taps.  The offset
conv is zero-initialised, so a freshly built neck equals a plain 3x3x3 FPN conv with zero bias; P6 stays the stride-2
subsample of P5 (fpn3d.py:121-123).

The conv engine (ExtractFeat3D(neck_dcn=True)) hands out the lateral sums and takes their gradients; everything here is
ordinary autograd over HIP-backed Functions (detector/layers.conv3d on the MFMA kernels, DeformConvFunction3D)."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _hip
from ..mmdet_ops.dcn import DeformConv3d
from ..mmdet_ops.dcn.deform_conv3d import _desc
from . import layers


class _DcnLevelBf16Fn(Function):
    """One level of the deformable neck on bf16 maps, offset conv + DeformConv3d in ONE autograd node (BASELINE configs[4]):
    lat [N,64,d,h,w] bf16 (channels-last storage) -> offsets (csrc/conv3d_bf16.hip, 81 channels padded to the kernel's 96-wide
    tile: the DeformConv3d kernels read that buffer as it is, row stride 96) -> sampled conv (csrc/deform_conv3d_bf16.hip) ->
    [N,64,d,h,w] bf16.  Nothing is converted or re-laid-out between the two, in either direction: the offset gradient leaves the
    sampling kernel in the padded bf16 layout the offset conv's backward kernels take."""

    @staticmethod
    def forward(ctx, lat, w_off, b_off, w_dcn):
        x16 = lat.detach().permute(0, 2, 3, 4, 1).contiguous()
        assert x16.is_cuda and x16.dtype == torch.bfloat16 and x16.size(4) == 64
        co_off = w_off.shape[0]
        cop = -(-co_off // 32) * 32
        wo = F.pad(w_off.detach(), (0, 0, 0, 0, 0, 0, 0, 0, 0, cop - co_off))
        shift = F.pad(b_off.detach().float(), (0, cop - co_off)).contiguous()
        off16 = _hip.conv3d_forward_bf16(x16, _hip.pack_conv_weight_bf16(wo), cop, (3, 3, 3), (1, 1, 1), (1, 1, 1), shift=shift)
        wd = w_dcn.detach()
        y16 = torch.empty_like(x16)
        d = _desc(x16.shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 1, 0, 1)
        L = _hip.lib()
        with torch.cuda.device(x16.device):
            _hip.check(L.mr3d_dcn_bf16_forward(ctypes.byref(d), 64, _hip.ptr(x16), _hip.ptr(off16), cop,
                                               _hip.ptr(_hip.pack_conv_weight_bf16(wd)), _hip.ptr(y16), _hip.stream_ptr(x16.device)),
                       'mr3d_dcn_bf16_forward')
        ctx.save_for_backward(x16, off16, wo, wd)
        ctx.co_off = co_off
        return y16.permute(0, 4, 1, 2, 3)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x16, off16, wo, wd = ctx.saved_tensors
        co_off, cop = ctx.co_off, off16.size(4)
        dev = x16.device
        g16 = gy.permute(0, 2, 3, 4, 1).contiguous()
        if g16.dtype != torch.bfloat16:
            g16 = g16.to(torch.bfloat16)
        d = _desc(x16.shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 1, 0, 1)
        L = _hip.lib()
        dwd = torch.zeros((27, 64, 64), dtype=torch.float32, device=dev)
        goff = torch.zeros_like(off16)
        dx = torch.zeros(x16.shape, dtype=torch.float32, device=dev)
        packed = torch.empty((3 * 27 * 64 * 64 + 1,), dtype=torch.float32, device=dev)
        wt = _hip.pack_conv_weight_bf16(wd.transpose(0, 1))
        with torch.cuda.device(dev):
            st = _hip.stream_ptr(dev)
            _hip.check(L.mr3d_dcn_bf16_wgrad(ctypes.byref(d), 64, _hip.ptr(x16), _hip.ptr(off16), cop, _hip.ptr(g16), _hip.ptr(dwd), st),
                       'mr3d_dcn_bf16_wgrad')
            _hip.check(L.mr3d_dcn_bf16_doffset(ctypes.byref(d), 64, _hip.ptr(x16), _hip.ptr(off16), cop, _hip.ptr(g16), _hip.ptr(wt),
                                               _hip.ptr(goff), cop, st), 'mr3d_dcn_bf16_doffset')
            _hip.check(L.mr3d_deform_conv3d_pack_weights(_hip.ptr(wd.contiguous()), 64, 64, 27, _hip.ptr(packed), st),
                       'mr3d_deform_conv3d_pack_weights')
            _hip.check(L.mr3d_dcn_bf16_dx(ctypes.byref(d), 64, _hip.ptr(off16), cop, _hip.ptr(g16), _hip.ptr(packed), _hip.ptr(dx), st),
                       'mr3d_dcn_bf16_dx')
        gw_dcn = dwd.permute(2, 1, 0).reshape(64, 64, 3, 3, 3).contiguous()
        # offset conv backward on the padded bf16 offset gradient
        dwo = torch.zeros((27, 64, cop), dtype=torch.float32, device=dev)
        dbo = torch.zeros((cop,), dtype=torch.float32, device=dev)
        _hip.conv3d_wgrad_bf16(x16, goff, dwo, (3, 3, 3), (1, 1, 1), (1, 1, 1), dbias=dbo, sparse_dy=True)    # zero where dy was
        gw_off = dwo.permute(2, 1, 0).reshape(cop, 64, 3, 3, 3)[:co_off].contiguous()
        gb_off = dbo[:co_off].contiguous()
        wb = _hip.pack_conv_weight_bf16(wo.flip(2, 3, 4).transpose(0, 1))                # [ci = 64][cop][3,3,3]: data-gradient operand
        # the offset gradient is non-zero only where dy is (position-wise): dy's liveness bitmap lets the data-gradient kernel write
        # the tiles whose halo is dead from the epilogue alone
        live = _hip.voxel_liveness(g16) if x16.shape[1] * x16.shape[2] * x16.shape[3] >= 65536 else None
        gx16 = _hip.conv3d_forward_bf16(goff, wb, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), add1=dx.to(torch.bfloat16), x_live=live)
        return gx16.permute(0, 4, 1, 2, 3), gw_off, gb_off, gw_dcn


class DCNNeck3D(nn.Module):

    def __init__(self, channels=64, num_levels=4, deformable_groups=1, offset_bf16=False):
        """offset_bf16: the offset-predicting convs run with bf16 activation storage (the detector's bf16 mode: they are a third
        of the neck's time in fp32); the sampling itself stays fp32."""
        super().__init__()
        self.channels, self.num_levels, self.dg = channels, num_levels, deformable_groups
        self.offset_bf16 = bool(offset_bf16)
        self.conv_offset = nn.ModuleList([nn.Conv3d(channels, deformable_groups * 3 * 27, 3, padding=1) for _ in range(num_levels)])
        self.conv = nn.ModuleList([DeformConv3d(channels, channels, 3, padding=1, deformable_groups=deformable_groups)
                                   for _ in range(num_levels)])
        self.init_weights()

    def init_weights(self):
        for off, conv in zip(self.conv_offset, self.conv):
            nn.init.zeros_(off.weight); nn.init.zeros_(off.bias)            # DCN convention: start as a regular conv
            nn.init.xavier_uniform_(conv.weight)                            # fpn3d.py:90-93

    def forward(self, laterals):
        """laterals: the 4 top-down lateral sums [N,64,d,h,w] (channels_last_3d; fp32, or bf16 from the bf16 engine) -> (P2..P6)"""
        outs = []
        if laterals[0].dtype == torch.bfloat16:          # the bf16 engine's pyramid: one fused node per level, bf16 throughout
            assert self.dg == 1
            for lat, off, conv in zip(laterals, self.conv_offset, self.conv):
                outs.append(_DcnLevelBf16Fn.apply(lat, off.weight, off.bias, conv.weight))
            outs.append(outs[-1][:, :, ::2, ::2, ::2])
            return tuple(outs)
        for lat, off, conv in zip(laterals, self.conv_offset, self.conv):
            if self.offset_bf16:
                offset = layers.conv3d_bf16(lat, off.weight, off.bias, padding=1)
            else:
                offset = layers.conv3d(lat, off.weight, off.bias, padding=1)
            outs.append(conv(lat, offset))
        outs.append(outs[-1][:, :, ::2, ::2, ::2])                          # F.max_pool3d(x, 1, stride=2) == subsample
        return tuple(outs)

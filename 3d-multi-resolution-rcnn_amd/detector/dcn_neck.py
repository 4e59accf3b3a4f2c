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
import torch
import torch.nn as nn

from ..mmdet_ops.dcn import DeformConv3d
from . import layers


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
        """laterals: the 4 top-down lateral sums [N,64,d,h,w] (channels_last_3d, fp32) -> (P2..P6)"""
        outs = []
        for lat, off, conv in zip(laterals, self.conv_offset, self.conv):
            if self.offset_bf16:
                offset = layers.conv3d_bf16(lat, off.weight, off.bias, padding=1)
            else:
                offset = layers.conv3d(lat, off.weight, off.bias, padding=1)
            outs.append(conv(lat, offset))
        outs.append(outs[-1][:, :, ::2, ::2, ::2])                          # F.max_pool3d(x, 1, stride=2) == subsample
        return tuple(outs)

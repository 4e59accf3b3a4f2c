"""Plain-PyTorch fp32 restatement of the reference's ResNet3D-50 + FPN3D graph.

TEST INFRASTRUCTURE (part of the oracle): imported only by tests/ and by bench.py's cpu_baseline leg.

Follows mmdet/models/backbones/resnet3d.py:89-235,238-285,289-478 and mmdet/models/necks/fpn3d.py:10-134
with the same parameter names, so the closed-form weight fill (tests/golden/weightfill.py) gives the same
weights as on the real reference modules (pinned by tests/golden/backbone_fpn.npz)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _MaskedReLU(torch.autograd.Function):
    """ReLU linearised at a GIVEN activation pattern: y = x * mask, dx = g * mask.  Used by the parity tests to evaluate
    the fp64 restatement at the engine's own ReLU pattern: an fp32 pre-activation within rounding of 0 may land on either
    side, and a flipped mask routes a whole gradient contribution differently -- that is a property of the point the
    function is evaluated at, not an arithmetic error of the backward pass."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


def _relu(x, pattern, key):
    if pattern is None:
        return F.relu(x)
    return _MaskedReLU.apply(x, pattern[key].to(x.dtype))


class Bottleneck(nn.Module):
    pattern, key = None, ''          # optional activation pattern (see _MaskedReLU), set by RefExtractFeat.linearize_at

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv3d(inplanes, planes, 1, bias=False)
        self.bn3d1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn3d2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, 1, bias=False)
        self.bn3d3 = nn.BatchNorm3d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idn = x
        out = _relu(self.bn3d1(self.conv1(x)), self.pattern, self.key + '.relu1')
        out = _relu(self.bn3d2(self.conv2(out)), self.pattern, self.key + '.relu2')
        out = self.bn3d3(self.conv3(out))
        if self.downsample is not None:
            idn = self.downsample(x)
        return _relu(out + idn, self.pattern, self.key + '.relu3')


class ResNet3D50(nn.Module):
    pattern = None

    def __init__(self, blocks=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv3d(3, 16, 7, stride=(1, 2, 2), padding=3, bias=False)
        self.bn3d1 = nn.BatchNorm3d(16)
        inplanes = 16
        for i, nb in enumerate(blocks):
            planes, stride = 16 * 2 ** i, (1 if i == 0 else 2)
            layers = []
            for b in range(nb):
                s = stride if b == 0 else 1
                ds = None
                if b == 0 and (s != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv3d(inplanes, planes * 4, 1, stride=s, bias=False), nn.BatchNorm3d(planes * 4))
                layers.append(Bottleneck(inplanes, planes, s, ds))
                inplanes = planes * 4
            setattr(self, f'layer{i + 1}', nn.Sequential(*layers))

    def forward(self, x):
        x = _relu(self.bn3d1(self.conv1(x)), self.pattern, 'stem')
        if self.pattern is None:
            x = F.max_pool3d(x, 3, 2, 1)
        else:       # max-pool with a GIVEN arg-max (flat D*H*W index per output element): a gather
            idx = self.pattern['pool_index']
            x = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
        outs = []
        for i in range(4):
            x = getattr(self, f'layer{i + 1}')(x)
            outs.append(x)
        return tuple(outs)


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, pad):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, k, padding=pad)

    def forward(self, x):
        return self.conv(x)


class FPN3D(nn.Module):
    def __init__(self, in_channels=(64, 128, 256, 512), out_channels=64):
        super().__init__()
        self.lateral_convs = nn.ModuleList([_Conv(c, out_channels, 1, 0) for c in in_channels])
        self.fpn_convs = nn.ModuleList([_Conv(out_channels, out_channels, 3, 1) for _ in in_channels])

    def forward(self, inputs):
        lat = [l(x) for l, x in zip(self.lateral_convs, inputs)]
        for i in range(3, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
        outs = [c(l) for c, l in zip(self.fpn_convs, lat)]
        outs.append(F.max_pool3d(outs[-1], 1, stride=2))
        return tuple(outs)


class RefExtractFeat(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = ResNet3D50()
        self.neck = FPN3D()
        self.eval()            # BN eval mode (norm_eval=True); parameters still receive gradients

    def linearize_at(self, pattern):
        """pattern: None (plain ReLU / max-pool) or a dict {'stem', 'layer{s}.{b}.relu{1,2,3}': 0/1 masks [N,C,D,H,W],
        'pool_index': int64 [N,16,Do,Ho,Wo]} -- evaluate forward/backward at that activation pattern"""
        self.backbone.pattern = pattern
        for s in range(4):
            for b, blk in enumerate(getattr(self.backbone, f'layer{s + 1}')):
                blk.pattern, blk.key = pattern, f'layer{s + 1}.{b}'
        return self

    def forward(self, x):
        return self.neck(self.backbone(x))

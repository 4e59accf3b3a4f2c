#!/usr/bin/env python3
"""What the REFERENCE's own backbone path costs on this GPU: ResNet3D-50 + FPN3D written with plain torch.nn modules
(Conv3d / BatchNorm3d(eval) / MaxPool3d / F.interpolate -- the module graph of mmdet/models/backbones/resnet3d.py and
necks/fpn3d.py; on ROCm torch dispatches them to MIOpen + ATen kernels), forward + backward on the bench volume.
A measurement aid only: nothing in the product imports this file.

    python tools/torch_rocm_baseline.py [--vol 80,512,512] [--steps 5] [--channels-last]"""
import argparse
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, down):
        super().__init__()
        self.c1, self.b1 = nn.Conv3d(inplanes, planes, 1, bias=False), nn.BatchNorm3d(planes)
        self.c2, self.b2 = nn.Conv3d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm3d(planes)
        self.c3, self.b3 = nn.Conv3d(planes, planes * 4, 1, bias=False), nn.BatchNorm3d(planes * 4)
        self.down = nn.Sequential(nn.Conv3d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm3d(planes * 4)) if down else None

    def forward(self, x):
        idn = x if self.down is None else self.down(x)
        y = F.relu(self.b1(self.c1(x)))
        y = F.relu(self.b2(self.c2(y)))
        return F.relu(self.b3(self.c3(y)) + idn)


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv3d(3, 16, 7, (1, 2, 2), 3, bias=False), nn.BatchNorm3d(16), nn.ReLU(), nn.MaxPool3d(3, 2, 1))
        layers, inpl = [], 16
        for planes, blocks, stride in ((16, 3, 1), (32, 4, 2), (64, 6, 2), (128, 3, 2)):
            seq = [Bottleneck(inpl, planes, stride, True)] + [Bottleneck(planes * 4, planes, 1, False) for _ in range(blocks - 1)]
            layers.append(nn.Sequential(*seq))
            inpl = planes * 4
        self.layers = nn.ModuleList(layers)
        self.lat = nn.ModuleList([nn.Conv3d(c, 64, 1) for c in (64, 128, 256, 512)])
        self.out = nn.ModuleList([nn.Conv3d(64, 64, 3, padding=1) for _ in range(4)])

    def forward(self, x):
        x = self.stem(x)
        cs = []
        for l in self.layers:
            x = l(x)
            cs.append(x)
        lats = [m(c) for m, c in zip(self.lat, cs)]
        for i in range(3, 0, -1):
            lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode='nearest')
        outs = [m(l) for m, l in zip(self.out, lats)]
        outs.append(F.max_pool3d(outs[-1], 1, stride=2))
        return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vol', default='80,512,512')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--channels-last', action='store_true')
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.vol.split(','))
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = Net().to(dev).eval()                       # BatchNorm in eval mode, as the reference trains (norm_eval=True)
    for p in net.parameters():
        p.requires_grad_(True)
    x = torch.randn(1, 3, D, H, W, device=dev)
    if a.channels_last:
        net = net.to(memory_format=torch.channels_last_3d)
        x = x.contiguous(memory_format=torch.channels_last_3d)

    def step():
        net.zero_grad(set_to_none=True)
        outs = net(x)
        sum(o.sum() for o in outs).backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    print(f'torch/MIOpen ResNet3D-50+FPN3D fwd+bwd on 1x3x{D}x{H}x{W} fp32 ({"channels_last_3d" if a.channels_last else "NCDHW"}): '
          f'{ms:.1f} ms/step = {1e3 / ms:.2f} iters/s;  peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB')


if __name__ == '__main__':
    main()

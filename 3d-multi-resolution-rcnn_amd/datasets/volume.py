"""prepare_volume: grey volume -> the `[1, 3, D, Hp, Wp]` fp32 tensor the detector consumes, on the MI355X.

Mirrors the per-slice loop of the reference's dataset (mmdet/datasets/coco_3d_2scales.py:245-263 and :267-289:
np.repeat to 3 channels, ImageTransform = imnormalize + impad_to_multiple (transforms.py:33-51), stack, transpose)
and, with `upscale`, the whole-volume skimage.transform.resize of the second pathway (:219).  Only the ONE-channel
volume crosses PCIe (1/3 of the reference's H2D bytes as fp32, 1/12 as uint8); one kernel does the rest."""
import ctypes
import math

import numpy as np
import torch

from .. import _hip

# configs/3d-multi-resolution-rcnn.py:146-147
IMG_NORM_CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def prepare_volume(grey, mean=None, std=None, size_divisor=32, upscale=None, layout='HWD', device='cuda'):
    """grey: numpy array or tensor, uint8 or float32, [H, W, D] (the dataset's .npy order) or [D, H, W].
    Returns (img [1, 3, D', Hp, Wp] fp32 on `device`, img_meta dict with the reference's img_shape / pad_shape
    = (h, w, 3, d) tuples).  upscale: None or a factor (the shipped second pathway uses 1.5)."""
    assert layout in ('HWD', 'DHW')
    mean = IMG_NORM_CFG['mean'] if mean is None else mean
    std = IMG_NORM_CFG['std'] if std is None else std
    t = torch.as_tensor(grey)
    if t.dtype not in (torch.uint8, torch.float32):
        t = t.float()
    t = t.to(device).contiguous()
    if not t.is_cuda:
        raise NotImplementedError('prepare_volume runs on the MI355X only (no CPU fallback)')
    if layout == 'HWD':
        H, W, D = t.shape
    else:
        D, H, W = t.shape
    if upscale:
        Do, Ho, Wo = (int(round(v * upscale)) for v in (D, H, W))     # skimage.resize rounds the requested output shape
    else:
        Do, Ho, Wo = D, H, W
    div = size_divisor or 1
    Hp, Wp = int(math.ceil(Ho / div)) * div, int(math.ceil(Wo / div)) * div
    out = torch.empty((1, 3, Do, Hp, Wp), dtype=torch.float32, device=t.device)
    m3, s3 = (ctypes.c_float * 3)(*[float(v) for v in mean]), (ctypes.c_float * 3)(*[float(v) for v in std])
    with torch.cuda.device(t.device):
        st = _hip.lib().mr3d_volume_prepare(_hip.ptr(t), int(t.dtype == torch.uint8), int(layout == 'HWD'), D, H, W, Do, Ho, Wo,
                                            Hp, Wp, m3, s3, _hip.ptr(out), _hip.stream_ptr(t.device))
    _hip.check(st, 'mr3d_volume_prepare')
    meta = dict(ori_shape=(Ho, Wo, 3), img_shape=(Ho, Wo, 3, Do), pad_shape=(Hp, Wp, 3, Do), scale_factor=1.0, flip=False)
    return out, meta

#!/usr/bin/env python3
"""Data-parallel training loop of the shipped dual-pathway model on synthetic volumes -- the reference's
`tools/train.py --launcher pytorch` path (mmdet/apis/train.py:37-100: MMDistributedDataParallel + DistOptimizerHook)
reduced to its hot loop: forward_train -> sum of losses -> backward -> gradient all-reduce -> clip -> SGD.

    python tools/train_synthetic.py --iters 20 --vol 32,128,128
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_synthetic.py --iters 50

All parameters of the detector live in ONE flat fp32 arena (dist_utils.DetectorArena: backbone + neck + every head;
BN running statistics behind the trainable range), so the exchange is ONE in-place RCCL all-reduce per iteration --
the reference's single coalesced buffer (mmdet/core/utils/dist_utils.py:22-28) without its flatten / copy-back -- and
the optimizer steps one tensor."""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402
from mr3d_amd.datasets import prepare_volume  # noqa: E402
from mr3d_amd.detector import TwoStageDetector3D2Scales  # noqa: E402
from mr3d_amd.dist_utils import ArenaSGD, DetectorArena, broadcast_params, train_step  # noqa: E402


def synthetic_sample(rng, D, H, W, dev, n_gt=6):
    grey = (rng.random((H, W, D)) * 255).astype(np.float32)
    ctr = rng.uniform([20, 20, 6], [W - 20, H - 20, D - 6], size=(n_gt, 3))
    sz = np.concatenate([rng.uniform(12, 36, (n_gt, 2)), rng.uniform(4, 10, (n_gt, 1))], 1)
    b = np.stack([ctr[:, 0] - sz[:, 0] / 2, ctr[:, 1] - sz[:, 1] / 2, ctr[:, 0] + sz[:, 0] / 2, ctr[:, 1] + sz[:, 1] / 2,
                  ctr[:, 2] - sz[:, 2] / 2, ctr[:, 2] + sz[:, 2] / 2], 1).astype(np.float32)
    masks = torch.zeros((n_gt, D, H, W), dtype=torch.uint8, device=dev)
    for k, q in enumerate(b.astype(int)):
        masks[k, q[4]:q[5] + 1, q[1]:q[3] + 1, q[0]:q[2] + 1] = 1
        grey[q[1]:q[3] + 1, q[0]:q[2] + 1, q[4]:q[5] + 1] += 80.0                  # lesions are brighter: something to learn
    img, meta = prepare_volume(grey, device=dev)
    img_2, meta_2 = prepare_volume(grey, upscale=1.5, device=dev)
    gt = torch.from_numpy(b).to(dev)
    return img, meta, img_2, meta_2, gt, torch.ones(n_gt, dtype=torch.long, device=dev), masks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--vol', default='32,128,128', help='D,H,W of the region (the reference trains on 128x128 crops)')
    ap.add_argument('--lr', type=float, default=0.001)          # configs/...py:235 (SGD lr 0.001, momentum 0.9, wd 1e-4)
    ap.add_argument('--bf16', action='store_true', help='BASELINE configs[4]: bf16 activation storage')
    ap.add_argument('--dcn-neck', action='store_true', help='BASELINE configs[4]: DeformConv3d neck')
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.vol.split(','))
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('LOCAL_RANK', 0), ('WORLD_SIZE', 1)))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
    torch.manual_seed(0)
    det = TwoStageDetector3D2Scales(device=dev, dtype=torch.bfloat16 if a.bf16 else torch.float32, neck_dcn=a.dcn_neck)
    det.extract.init_weights()
    arena = DetectorArena(det)
    broadcast_params(arena.flat_params)                          # identical initial weights (mmcv DDP init)
    opt = ArenaSGD(arena, lr=a.lr, momentum=0.9, weight_decay=1e-4)
    rng = np.random.default_rng(100 + rank)                      # every rank draws its own volumes (DistributedSampler)
    t0 = time.perf_counter()
    for it in range(a.iters):
        img, meta, img_2, meta_2, gt, gl, masks = synthetic_sample(rng, D, H, W, dev)
        batch = ((img, [meta], img_2, [meta_2], [gt], [gt * 1.5], [gl], [gl]), dict(gt_masks=[masks]))
        total, losses = train_step(det, arena, opt, batch, world, grad_clip=35.0, freeze_gc=True)
        if rank == 0 and (it % 5 == 0 or it == a.iters - 1):
            print(f'iter {it:4d}  loss {total.item():8.4f}  ' +
                  '  '.join(f'{k} {float(sum(t.sum() for t in v) if isinstance(v, list) else v.sum()):.4f}'
                            for k, v in losses.items() if 'loss' in k), flush=True)
    torch.cuda.synchronize()
    if rank == 0:
        print(f'{a.iters} iterations in {time.perf_counter() - t0:.2f} s on {world} GPU(s)')
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

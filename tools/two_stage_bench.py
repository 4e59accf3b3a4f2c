#!/usr/bin/env python3
"""Phase timing of the single-pathway training step (BASELINE.json configs[2]) on one MI355X.
usage: python tools/two_stage_bench.py [--steps 5] [--vol 80,512,512] [--gts 12]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mr3d_amd  # noqa: E402
from mr3d_amd.detector import TwoStageDetector3D, TwoStageDetector3D2Scales  # noqa: E402


def synthetic_gt(rng, n, D, H, W, dev):
    ctr = rng.uniform([30, 30, 8], [W - 30, H - 30, D - 8], size=(n, 3))
    sz = np.concatenate([rng.uniform(16, 48, (n, 2)), rng.uniform(6, 14, (n, 1))], 1)
    b = np.stack([ctr[:, 0] - sz[:, 0] / 2, ctr[:, 1] - sz[:, 1] / 2, ctr[:, 0] + sz[:, 0] / 2, ctr[:, 1] + sz[:, 1] / 2,
                  ctr[:, 2] - sz[:, 2] / 2, ctr[:, 2] + sz[:, 2] / 2], 1).astype(np.float32)
    masks = torch.zeros((n, D, H, W), dtype=torch.uint8, device=dev)
    for k, q in enumerate(b.astype(int)):
        masks[k, q[4]:q[5] + 1, q[1]:q[3] + 1, q[0]:q[2] + 1] = 1
    return torch.from_numpy(b).to(dev), masks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--vol', default='80,512,512')
    ap.add_argument('--gts', type=int, default=12)
    ap.add_argument('--repeats', type=int, default=1)
    ap.add_argument('--none-grads', action='store_true')
    ap.add_argument('--two-scales', action='store_true', help='the shipped dual-pathway model (region + its 1.5x copy)')
    ap.add_argument('--phases', action='store_true', help='device-event timing of the phases of one step')
    ap.add_argument('--json', action='store_true', help='print one JSON line with the step time')
    ap.add_argument('--no-mask', action='store_true')
    ap.add_argument('--static-mask-batch', action='store_true',
                    help='mask heads always on the full 128-row batch (no positive-count read-back, no host sync)')
    ap.add_argument('--profile', action='store_true', help='torch profiler table of one step')
    ap.add_argument('--split-backward', action='store_true', help='forward_train + total.backward() instead of forward_backward')
    ap.add_argument('--host-profile', action='store_true', help='cProfile of the host side of 5 steps + enqueue time per step')
    ap.add_argument('--bf16', action='store_true', help='BASELINE configs[4]: bf16 activation storage in backbone + FPN')
    ap.add_argument('--dcn-neck', action='store_true', help='BASELINE configs[4]: DeformConv3d in place of the FPN output convs')
    ap.add_argument('--tune', default='', help='key=value pairs for mr3d_tuning_set, e.g. 7=1 (round-5 RoIAlign3D kernels)')
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.vol.split(','))
    from mr3d_amd import _hip as _h
    for kv in filter(None, a.tune.split(',')):
        _h.lib().mr3d_tuning_set(*(int(v) for v in kv.split('=')))
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    kw = dict(dtype=torch.bfloat16 if a.bf16 else torch.float32, neck_dcn=a.dcn_neck)
    det = TwoStageDetector3D2Scales(device=dev, **kw) if a.two_scales else TwoStageDetector3D(device=dev, with_mask=not a.no_mask, **kw)
    det.extract.init_weights()
    if a.static_mask_batch:
        import copy
        det.train_cfg = copy.deepcopy(det.train_cfg)
        det.train_cfg['rcnn']['mask_batch_buckets'] = None
    rng = np.random.default_rng(0)
    img = torch.randn(1, 3, D, H, W, device=dev)
    gt, masks = synthetic_gt(rng, a.gts, D, H, W, dev)
    gl = torch.ones(a.gts, dtype=torch.long, device=dev)
    meta = [dict(img_shape=(H, W, 3, D), pad_shape=(H, W, 3, D), ori_shape=(H, W, 3))]
    if a.two_scales:
        D2, H2, W2 = int(round(D * 1.5)), int(round(H * 1.5)), int(round(W * 1.5))
        img_2 = torch.randn(1, 3, D2, H2, W2, device=dev)
        meta_2 = [dict(img_shape=(H2, W2, 3, D2), pad_shape=(H2, W2, 3, D2), ori_shape=(H2, W2, 3))]

    def step():
        det.extract.zero_grad()
        if a.none_grads:
            for m in [c for n_, c in det.named_children() if n_ != 'extract']:
                m.zero_grad(set_to_none=True)
        if a.two_scales and not a.split_backward:
            return det.forward_backward(img, meta, img_2, meta_2, [gt], [gt * 1.5], [gl], [gl], gt_masks=[masks])
        elif a.two_scales:
            losses = det.forward_train(img, meta, img_2, meta_2, [gt], [gt * 1.5], [gl], [gl], gt_masks=[masks])
        elif not a.split_backward:
            return det.forward_backward(img, meta, [gt], [gl], gt_masks=[masks])
        else:
            losses = det.forward_train(img, meta, [gt], [gl], gt_masks=[masks])
        tot = det.total_loss(losses)
        tot.backward()
        return losses

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    from mr3d_amd.dist_utils import freeze_gc_after_warmup      # the opt-in collector freeze of a training loop (dist_utils.freeze_gc_after_warmup)
    for _ in range(4):
        freeze_gc_after_warmup()
    if a.phases and not a.two_scales:
        det._phase_events = []
        det.extract.zero_grad()
        if a.split_backward:
            losses = det.forward_train(img, meta, [gt], [gl], gt_masks=[masks])
            tot = det.total_loss(losses)
            det._mark('sum losses')
            tot.backward()
            det._mark('backward (heads + backbone)')
        else:
            losses = det.forward_backward(img, meta, [gt], [gl], gt_masks=[masks])
        torch.cuda.synchronize()
        ev = det._phase_events
        det._phase_events = None
        for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
            print(f'  {n1:36s} {e0.elapsed_time(e1):8.3f} ms')
        print(f'  {"total":36s} {ev[0][1].elapsed_time(ev[-1][1]):8.3f} ms')
    if a.profile:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        rows = [(e.key, e.self_device_time_total / 1e3, e.count) for e in prof.key_averages() if e.self_device_time_total > 0]
        rows.sort(key=lambda r: -r[1])
        tot = sum(r[1] for r in rows)
        print(f'device time total {tot:.2f} ms')
        for k, ms, cnt in rows[:60]:
            print(f'{ms:8.3f} ms {100 * ms / tot:5.1f}% n={cnt:4d}  {k[:150]}')
    if a.host_profile:
        import cProfile
        import pstats
        enq = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            enq.append((round((t1 - t0) * 1e3, 2), round((time.perf_counter() - t0) * 1e3, 2)))
        print('host enqueue ms / step wall ms (GPU idle at start):', enq)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(70)
        pstats.Stats(pr).sort_stats('tottime').print_stats(35)
    for rep in range(a.repeats):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            losses = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        if a.repeats > 1:
            print(f'  repeat {rep}: {dt * 1e3:.2f} ms/step')
    if a.json:
        import json
        print(json.dumps({'two_stage_step_ms': dt * 1e3, 'vol': [D, H, W], 'gts': a.gts, 'steps': a.steps,
                          'sample_counts': det.last_sample_counts.tolist(), 'static_mask_batch': a.static_mask_batch,
                          'bf16': a.bf16, 'dcn_neck': a.dcn_neck}))
        return
    print({k: ([round(float(x), 5) for x in v] if isinstance(v, list) else round(float(v), 5)) for k, v in losses.items()})
    print('sample counts (pos, neg, cand_pos, cand_neg):', det.last_sample_counts.tolist())
    print(f'two-stage step: {dt * 1e3:.2f} ms  ({1 / dt:.2f} iters/s)  vol {D}x{H}x{W}, {a.gts} gts, mask={not a.no_mask}')


if __name__ == '__main__':
    main()

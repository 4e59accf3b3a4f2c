#!/usr/bin/env python3
"""Where do the backbone gradients of the bf16 (+ DeformConv3d neck) dual-pathway model leave the fp32 model's?  Per-parameter
relative L2 error, for neck_dcn in (False, True).  usage: python tools/diag/cfg4_grad_diag.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mr3d_amd  # noqa: E402,F401
from mr3d_amd.detector import TwoStageDetector3D2Scales  # noqa: E402
from mr3d_amd.datasets import prepare_volume  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(3)
    grey = (rng.random((128, 128, 32)) * 255).astype(np.float32)
    img, meta = prepare_volume(grey)
    img_2, meta_2 = prepare_volume(grey, upscale=1.5)
    gt = torch.tensor([[20, 30, 60, 75, 4, 20], [70, 20, 110, 64, 10, 28], [40, 80, 90, 120, 2, 14]], dtype=torch.float32, device=dev)
    masks = torch.zeros((3, 32, 128, 128), dtype=torch.uint8, device=dev)
    for k, b in enumerate(gt.int().tolist()):
        masks[k, b[4]:b[5] + 1, b[1]:b[3] + 1, b[0]:b[2] + 1] = 1
    gl = torch.ones(3, dtype=torch.long, device=dev)
    w = lambda n: rng.integers(0, 2 ** 31 - 1, n)
    draws = {k: [(w(128), w(256))] for k in ('rpn', 'rpn_2')}
    draws.update({k: [(w(128), w(512))] for k in ('rcnn', 'rcnn_2', 'refine')})
    args = (img, [meta], img_2, [meta_2], [gt], [gt * 1.5], [gl], [gl])
    for dcn in (False, True):
        for zir in (True, 0.25, False):
            def build(dtype):
                torch.manual_seed(5)
                det = TwoStageDetector3D2Scales(device=dev, dtype=dtype, neck_dcn=dcn)
                det.extract.zero_init_residual = zir is True
                det.extract.init_weights()
                if zir not in (True, False):          # damped residual branches: every conv gets gradient, activations stay tame
                    with torch.no_grad():
                        for b in det.extract.blocks:
                            det.extract.get_parameter(b['conv3'].bn + '.weight').fill_(float(zir))
                if dcn:
                    with torch.no_grad():
                        for off in det.dcn_neck.conv_offset:
                            off.weight.normal_(0, 0.01)
                return det
            ref = build(torch.float32)
            ref.extract.zero_grad()
            lr = ref.forward_train(*args, gt_masks=[masks], draws=draws)
            ref.total_loss(lr).backward()
            det = build(torch.bfloat16)
            det.load_state_dict(ref.state_dict(), strict=True)
            det.extract.zero_grad()
            lb = det.forward_train(*args, gt_masks=[masks], draws=draws, proposals=ref.last_proposals)
            det.total_loss(lb).backward()
            g, gr = det.extract.flat_grads, ref.extract.flat_grads
            print(f'--- neck_dcn={dcn} zero_init_residual={zir}: whole arena rel L2 {((g - gr).norm() / gr.norm()).item():.3e}  '
                  f'cos {torch.nn.functional.cosine_similarity(g, gr, dim=0).item():.5f}  |g_ref| {gr.norm().item():.3e}')
            rows = []
            for (n, p), (_, pr) in zip(det.extract.named_parameters(), ref.extract.named_parameters()):
                d, r = (p.grad - pr.grad).norm().item(), pr.grad.norm().item()
                rows.append((d, r, n))
            rows.sort(reverse=True)
            for d, r, n in rows[:12]:
                print(f'   |diff| {d:10.3e}  |ref| {r:10.3e}  rel {d / max(r, 1e-30):8.3f}  {n}')
            for k in lr:
                v = lambda x: float(sum(t.sum() for t in x)) if isinstance(x, (list, tuple)) else float(x)
                print(f'   {k}: bf16 {v(lb[k]):.5f} fp32 {v(lr[k]):.5f}')


if __name__ == '__main__':
    main()

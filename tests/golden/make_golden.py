#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container: it imports the reference's own Python from
/root/reference (with empty stand-ins for the third-party packages that are not
installed: mmcv, cv2, skimage, pycocotools, and for the reference's compiled
CUDA ops) and records inputs + the reference's outputs as small .npz files.
Nothing from the reference travels: the fixtures hold numbers only.

Vectors written (each is {inputs..., expected...}):
  nms3d.npz        mmdet/core/evaluation/coco_utils.py:245-282 nms_3d_python
                   (the reference's only CPU NMS3D; true 3D IoU, keep iff iou <= thr)
  iou3d.npz        mmdet/core/bbox/geometry.py:49-60 bbox_overlaps (6-col branch)
                   + the known answers of bbox_overlaps_test :81-101
  codec.npz        mmdet/core/bbox/transforms.py:34-65,105-160,220-239
  anchors.npz      mmdet/core/anchor/anchor_generator_3d.py:6-92
  roi_levels.npz   mmdet/models/roi_extractors/single_level.py:58-82
  backbone_fpn.npz mmdet/models/backbones/resnet3d.py + necks/fpn3d.py forward/backward
                   with a closed-form weight fill (tests/golden/weightfill.py)
  targets.npz      mmdet/core/bbox/assigners/max_iou_assigner.py:47-196, samplers/random_sampler.py + base_sampler.py
                   (numpy's randint draws are RECORDED so that they can be replayed), anchor/anchor_target.py:114-216,
                   bbox/bbox_target.py:139-165
  losses.npz       mmdet/core/loss/losses.py (weighted BCE / CE / smooth-L1, accuracy, mask BCE): values + gradients
  eval3d.npz (+ _input.json, _nms.npz)  pycocotools_local/cocoeval.py COCOeval(is3D=True, 'bbox') precision / recall /
                   stats on a synthetic set; coco_utils.py:245-282 nms_3d_python on a record list
  state_dict_keys.json  checkpoint key -> shape of backbone / neck / rpn_head / bbox_head / mask_head as the shipped
                   config builds them (checkpoint compatibility)

usage: python tests/golden/make_golden.py   (from the repo root)
"""
import ast
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weightfill import fill_module_closed_form, cotangent_like, volume_like  # noqa: E402


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    import torch.nn as nn

    def constant_init(module, val, bias=0):
        nn.init.constant_(module.weight, val)
        if hasattr(module, 'bias') and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def kaiming_init(module, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
        if distribution == 'uniform':
            nn.init.kaiming_uniform_(module.weight, mode=mode, nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(module.weight, mode=mode, nonlinearity=nonlinearity)
        if hasattr(module, 'bias') and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def xavier_init(module, gain=1, bias=0, distribution='normal'):
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
        if hasattr(module, 'bias') and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def normal_init(module, mean=0, std=1, bias=0):
        nn.init.normal_(module.weight, mean, std)
        if hasattr(module, 'bias') and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    _pkg('mmdet', f'{REF}/mmdet')
    _pkg('mmdet.models', f'{REF}/mmdet/models')
    for sub in ('backbones', 'necks', 'roi_extractors'):
        _pkg(f'mmdet.models.{sub}', f'{REF}/mmdet/models/{sub}')
    _stub('mmcv')
    _stub('mmcv.runner', load_checkpoint=None)
    _stub('mmcv.cnn', constant_init=constant_init, kaiming_init=kaiming_init,
          xavier_init=xavier_init, normal_init=normal_init)
    _stub('mmdet.ops', DeformConv=None, ModulatedDeformConv=None, nms=None)
    for n in ('cv2', 'skimage', 'skimage.transform', 'pycocotools', 'pycocotools.mask'):
        _stub(n)
    _pkg('mmdet.core', f'{REF}/mmdet/core')
    _pkg('mmdet.core.bbox', f'{REF}/mmdet/core/bbox')
    _pkg('mmdet.core.anchor', f'{REF}/mmdet/core/anchor')
    ref = types.SimpleNamespace()
    ref.tr = importlib.import_module('mmdet.core.bbox.transforms')
    ref.geo = importlib.import_module('mmdet.core.bbox.geometry')
    ref.ag = importlib.import_module('mmdet.core.anchor.anchor_generator_3d')
    ref.ResNet3D = importlib.import_module('mmdet.models.backbones.resnet3d').ResNet3D
    ref.FPN3D = importlib.import_module('mmdet.models.necks.fpn3d').FPN3D
    ref.SingleRoIExtractor = importlib.import_module('mmdet.models.roi_extractors.single_level').SingleRoIExtractor
    # coco_utils imports pycocotools/tqdm/mmcv at module level: pull the one function out by AST
    src = open(f'{REF}/mmdet/core/evaluation/coco_utils.py').read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'nms_3d_python'][0]
    ns = {'np': np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'coco_utils.nms_3d_python', 'exec'), ns)
    ref.nms_3d_python = ns['nms_3d_python']
    # target assignment / sampling / losses (section 8(f) rank 1)
    sys.modules['mmdet.ops'].sigmoid_focal_loss = None
    bb = sys.modules['mmdet.core.bbox']
    asg = importlib.import_module('mmdet.core.bbox.assigners')
    smp = importlib.import_module('mmdet.core.bbox.samplers')
    sys.modules['mmcv.runner'].obj_from_dict = None
    asmp = importlib.import_module('mmdet.core.bbox.assign_sampling')
    for k in ('build_assigner', 'build_sampler', 'assign_and_sample'):
        setattr(bb, k, getattr(asmp, k))
    bb.PseudoSampler, bb.bbox2delta, bb.bbox2delta3d = smp.PseudoSampler, ref.tr.bbox2delta, ref.tr.bbox2delta3d
    misc_spec = importlib.util.spec_from_file_location('refmisc', f'{REF}/mmdet/core/utils/misc.py')
    misc = importlib.util.module_from_spec(misc_spec)
    misc_spec.loader.exec_module(misc)
    _stub('mmdet.core.utils', multi_apply=misc.multi_apply)
    bt = importlib.import_module('mmdet.core.bbox.bbox_target')
    ref.at = importlib.import_module('mmdet.core.anchor.anchor_target')
    ref.MaxIoUAssigner, ref.RandomSampler, ref.bt = asg.MaxIoUAssigner, smp.RandomSampler, bt
    _pkg('mmdet.core.loss', f'{REF}/mmdet/core/loss')
    ref.losses = importlib.import_module('mmdet.core.loss.losses')
    # heads (state-dict key names / shapes of the shipped single-pathway model): they import names from mmdet.core
    L = ref.losses
    sys.modules['mmdet.core'].__dict__.update(
        AnchorGenerator3D=ref.ag.AnchorGenerator3D, anchor_target=ref.at.anchor_target, delta2bbox=ref.tr.delta2bbox,
        delta2bbox3D=ref.tr.delta2bbox3D, multi_apply=misc.multi_apply, tensor2img3D=misc.tensor2img3D,
        weighted_cross_entropy=L.weighted_cross_entropy, weighted_smoothl1=L.weighted_smoothl1,
        weighted_binary_cross_entropy=L.weighted_binary_cross_entropy, weighted_sigmoid_focal_loss=L.weighted_sigmoid_focal_loss,
        multiclass_nms=None, multiclass_nms_3d=None, bbox_target_3d=bt.bbox_target_3d, accuracy=L.accuracy,
        mask_cross_entropy=L.mask_cross_entropy, mask_target=None)
    sys.modules['skimage.transform'].resize = None
    for sub in ('anchor_heads', 'bbox_heads', 'mask_heads'):
        _pkg(f'mmdet.models.{sub}', f'{REF}/mmdet/models/{sub}')
    ref.RPNHead3D = importlib.import_module('mmdet.models.anchor_heads.rpn_head_3d').RPNHead3D
    ref.SharedFCBBoxHead3D = importlib.import_module('mmdet.models.bbox_heads.convfc_bbox_head_3d').SharedFCBBoxHead3D
    ref.FCNMaskHead3D = importlib.import_module('mmdet.models.mask_heads.fcn_mask_head_3d').FCNMaskHead3D
    ref.SharedFCBBoxHead3DRefinement = importlib.import_module(
        'mmdet.models.bbox_heads.convfc_bbox_head_3d_refinement_head').SharedFCBBoxHead3DRefinement
    return ref


# ----------------------------------------------------------------------------------------------
def rand_boxes(rng, n, vol=(512, 512, 80), size_xy=(6, 60), size_z=(3, 20), cluster=0):
    """[n,6] fp32 boxes x1,y1,x2,y2,z1,z2 (inclusive).  cluster>0: boxes jittered around
    `cluster` centres so that many pairs overlap above typical thresholds."""
    W, H, D = vol
    if cluster:
        cid = rng.integers(0, cluster, n)
        ctr = rng.uniform([20, 20, 5], [W - 20, H - 20, D - 5], size=(cluster, 3))
        csz = np.concatenate([rng.uniform(*size_xy, size=(cluster, 2)), rng.uniform(*size_z, size=(cluster, 1))], 1)
        ctr = ctr[cid] + rng.normal(0, 0.08, size=(n, 3)) * csz[cid]
        sz3 = csz[cid] * (1.0 + rng.normal(0, 0.08, size=(n, 3)))
        sxy, sz = sz3[:, :2], sz3[:, 2:]
    else:
        ctr = rng.uniform([0, 0, 0], [W, H, D], size=(n, 3))
        sxy = rng.uniform(*size_xy, size=(n, 2))
        sz = rng.uniform(*size_z, size=(n, 1))
    half = np.concatenate([sxy, sz], 1) / 2
    lo, hi = ctr - half, ctr + half
    b = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    return b.astype(np.float32)


def iou_matrix_f64(b):
    b = b.astype(np.float64)
    x1, y1, x2, y2, z1, z2 = [b[:, i] for i in range(6)]
    w = np.clip(np.minimum(x2[:, None], x2) - np.maximum(x1[:, None], x1) + 1, 0, None)
    h = np.clip(np.minimum(y2[:, None], y2) - np.maximum(y1[:, None], y1) + 1, 0, None)
    d = np.clip(np.minimum(z2[:, None], z2) - np.maximum(z1[:, None], z1) + 1, 0, None)
    inter = w * h * d
    a = (x2 - x1 + 1) * (y2 - y1 + 1) * (z2 - z1 + 1)
    return inter / (a[:, None] + a - inter)


def make_nms(ref, out):
    rng = np.random.default_rng(20260101)
    cases = {}
    specs = [(1, 0.7, 0), (2, 0.5, 1), (7, 0.3, 2), (63, 0.7, 6), (64, 0.7, 6), (65, 0.5, 6),
             (128, 0.7, 10), (200, 0.1, 20), (333, 0.5, 25), (500, 0.7, 30), (1000, 0.7, 60),
             (2000, 0.7, 150), (2000, 0.5, 0), (4000, 0.5, 300)]
    for k, (n, thr, cl) in enumerate(specs):
        for attempt in range(100):
            boxes = rand_boxes(rng, n, cluster=cl)
            scores = rng.permutation(n).astype(np.float32) / np.float32(n) + np.float32(0.5 / n)  # distinct
            iou = iou_matrix_f64(boxes)
            if np.abs(iou - thr).min() > 1e-5:   # no pair within fp32 noise of the threshold
                break
        else:
            raise RuntimeError('could not draw a threshold-safe case')
        dets = np.concatenate([boxes, scores[:, None]], 1).astype(np.float32)
        kept = ref.nms_3d_python(np.arange(n), dets, thr)      # reference call; fp32 numpy arithmetic
        keep = np.sort(np.asarray(kept, dtype=np.int64))         # nms_kernel.cu:253-256 returns ascending indices
        cases[f'dets_{k}'] = dets
        cases[f'thr_{k}'] = np.float32(thr)
        cases[f'keep_{k}'] = keep
        print(f'  nms case {k}: n={n} thr={thr} kept={len(keep)}')
    cases['num_cases'] = np.int64(len(specs))
    np.savez_compressed(out, **cases)


def make_iou(ref, out):
    rng = np.random.default_rng(7)
    a = rand_boxes(rng, 37, cluster=5)
    b = rand_boxes(rng, 53, cluster=5)
    iou = ref.geo.bbox_overlaps(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    ka1 = np.array([[2, 3, 4, 6, 3, 4], [39, 63, 203, 112, 4, 5], [49, 75, 203, 125, 4, 5], [31, 69, 201, 125, 4, 5]], np.float32)
    ka2 = np.array([[2, 3, 4, 6, 3, 4], [54, 66, 198, 114, 4, 5], [42, 78, 186, 126, 4, 5], [18, 63, 235, 135, 4, 5]], np.float32)
    ref.geo.bbox_overlaps_test()   # the reference's own known-answer test must pass here
    ka = np.array([ref.geo.bbox_overlaps(torch.from_numpy(ka1[i:i + 1]), torch.from_numpy(ka2[i:i + 1]))[0, 0].item()
                   for i in range(4)], np.float64)
    assert [round(v, 4) for v in ka] == [1.0, 0.798, 0.7899, 0.6125]
    np.savez_compressed(out, a=a, b=b, iou=iou.astype(np.float32), ka1=ka1, ka2=ka2, ka=ka.astype(np.float32),
                        ka_rounded=np.array([1.0, 0.798, 0.7899, 0.6125]))


def make_codec(ref, out):
    rng = np.random.default_rng(11)
    n = 257
    rois = rand_boxes(rng, n)
    deltas = rng.normal(0, 0.6, size=(n, 6)).astype(np.float32)
    deltas[:5] *= 20  # exercise the +-4.135 clamp on dw,dh,dz,dd
    means = [0., 0., 0., 0., 0., 0.]
    stds = [0.1, 0.1, 0.2, 0.2, 0.1, 0.1]
    max_shape = (512, 512, 3, 80)
    dec_rpn = ref.tr.delta2bbox3D(torch.from_numpy(rois), torch.from_numpy(deltas), [0.] * 6, [1.] * 6, max_shape).numpy()
    dec_rcnn = ref.tr.delta2bbox3D(torch.from_numpy(rois), torch.from_numpy(deltas), means, stds, max_shape).numpy()
    dec_noclip = ref.tr.delta2bbox3D(torch.from_numpy(rois), torch.from_numpy(deltas), means, stds, None).numpy()
    deltas12 = rng.normal(0, 0.5, size=(n, 12)).astype(np.float32)   # 2 classes x 6 (reg_class_agnostic=False)
    dec12 = ref.tr.delta2bbox3D(torch.from_numpy(rois), torch.from_numpy(deltas12), means, stds, max_shape).numpy()
    gt = rand_boxes(rng, n)
    enc = ref.tr.bbox2delta3d(torch.from_numpy(rois), torch.from_numpy(gt), means, stds).numpy()
    roi_list = [torch.from_numpy(rois[:100]), torch.from_numpy(rois[100:100]), torch.from_numpy(rois[100:])]
    r7 = ref.tr.bbox2roi3D(roi_list).numpy()
    np.savez_compressed(out, rois=rois, deltas=deltas, deltas12=deltas12, gt=gt, means=np.array(means, np.float32),
                        stds=np.array(stds, np.float32), max_shape=np.array(max_shape), dec_rpn=dec_rpn,
                        dec_rcnn=dec_rcnn, dec_noclip=dec_noclip, dec12=dec12, enc=enc, rois7=r7,
                        split=np.array([100, 0, n - 100]))


def make_anchors(ref, out):
    d = {}
    # configs/3d-multi-resolution-rcnn.py:20-24 (rpn_head) and :32-36 (rpn_head_2)
    strides, dstrides = [4, 8, 16, 32, 64], [2, 4, 8, 16, 32]
    for tag, scale in (('s2', 2), ('s3', 3)):
        for lv, (s, ds) in enumerate(zip(strides, dstrides)):
            g = ref.ag.AnchorGenerator3D(s, [scale], [scale], [1.0], ds)
            d[f'base_{tag}_{lv}'] = g.base_anchors.numpy()
    g = ref.ag.AnchorGenerator3D(4, [2], [2], [1.0], 2)
    assert g.base_anchors.tolist() == [[-2, -2, 5, 5, -1, 2]]
    fs = (3, 5, 4)   # (feat_z, feat_h, feat_w)
    d['grid_fs'] = np.array(fs)
    d['grid'] = g.grid_anchors(fs, 4, 2, device='cpu').numpy()
    d['valid'] = g.valid_flags(fs, (2, 4, 3), device='cpu').numpy()
    d['valid_size'] = np.array((2, 4, 3))
    np.savez_compressed(out, **d)


def make_roi_levels(ref, out):
    rng = np.random.default_rng(5)
    n = 400
    boxes = rand_boxes(rng, n, size_xy=(2, 300), size_z=(1, 60))
    rois = np.concatenate([np.zeros((n, 1), np.float32), boxes], 1)
    ext = ref.SingleRoIExtractor.__new__(ref.SingleRoIExtractor)
    ext.finest_scale = 56
    lv = ref.SingleRoIExtractor.map_roi_levels(ext, torch.from_numpy(rois), 4).numpy()
    np.savez_compressed(out, rois=rois, levels=lv.astype(np.int64), finest_scale=np.int64(56), num_levels=np.int64(4))


def make_backbone(ref, out):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    bb = ref.ResNet3D(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, style='pytorch')
    fpn = ref.FPN3D(in_channels=[64, 128, 256, 512], out_channels=64, num_outs=5)
    fill_module_closed_form(bb, 1)
    fill_module_closed_form(fpn, 2)
    bb.train()      # norm_eval=True keeps BN in eval mode (resnet3d.py:480-486)
    fpn.train()
    d = {}
    for tag, shape in (('a', (1, 3, 8, 32, 32)), ('b', (1, 3, 16, 64, 64))):
        x = volume_like(shape)
        bb.zero_grad(); fpn.zero_grad()
        c = bb(x)
        p = fpn(c)
        loss = sum((o * cotangent_like(o.shape, i)).sum() for i, o in enumerate(p))
        loss.backward()
        for i, o in enumerate(c):
            d[f'{tag}_c{i}'] = o.detach().numpy()
        for i, o in enumerate(p):
            d[f'{tag}_p{i}'] = o.detach().numpy()
        names, norms = [], []
        for pref, m in (('backbone.', bb), ('neck.', fpn)):
            for k, v in m.named_parameters():
                names.append(pref + k)
                norms.append(float(v.grad.double().norm()))
        d[f'{tag}_grad_names'] = np.array(names)
        d[f'{tag}_grad_norms'] = np.array(norms)
        for k in ('conv1.weight', 'bn3d1.weight', 'bn3d1.bias', 'layer1.0.conv2.weight', 'layer1.0.bn3d3.weight',
                  'layer2.0.downsample.0.weight', 'layer4.2.conv3.weight', 'layer3.1.bn3d2.bias'):
            d[f'{tag}_g_backbone.{k}'] = dict(bb.named_parameters())[k].grad.numpy().copy()
        for k in ('lateral_convs.0.conv.weight', 'lateral_convs.3.conv.bias', 'fpn_convs.0.conv.weight', 'fpn_convs.2.conv.bias'):
            d[f'{tag}_g_neck.{k}'] = dict(fpn.named_parameters())[k].grad.numpy().copy()
        d[f'{tag}_loss'] = np.float64(loss.item())
        print(f'  backbone case {tag}: shape={shape} loss={loss.item():.6f}')
    np.savez_compressed(out, **d)


class _RecordRandint:
    """stand-in for numpy.random.randint that records every draw (random_sampler.py:36)"""

    def __init__(self, seed):
        self.rng, self.draws = np.random.RandomState(seed), []

    def __call__(self, low=0, high=None, size=None):
        r = self.rng.randint(low=low, high=high, size=size)
        self.draws.append(np.asarray(r, np.int64).copy())
        return r


def make_targets(ref, out):
    rng = np.random.default_rng(77)
    d = {}
    real_randint = np.random.randint
    # ---- RPN stage: anchors of a 128x128x32 volume, shipped rpn train_cfg and a low-threshold variant
    strides, dstrides, scales = [4, 8, 16, 32, 64], [2, 4, 8, 16, 32], [4]
    img_shape = (128, 128, 3, 32)          # (h, w, c, d): anchor_target.py:196-199 reads [0], [1], [3]
    levels = []
    for st, ds in zip(strides, dstrides):
        g = ref.ag.AnchorGenerator3D(st, scales, scales, [1.0], ds)
        fs = (max(32 // ds, 1), 128 // st, 128 // st)
        levels.append(g.grid_anchors(fs, st, ds, device='cpu'))
    anchors = torch.cat(levels)
    for tag, (pos_thr, neg_thr, min_pos, num, frac, pos_weight, border, G) in {
            'rpn_shipped': (0.7, 0.3, 0.3, 256, 0.5, 3, 0, 4),
            'rpn_manypos': (0.25, 0.1, 0.1, 64, 0.5, -1, 0, 6),
            'rpn_border': (0.5, 0.3, 0.3, 256, 0.5, 3, 8, 3)}.items():
        gt = torch.from_numpy(rand_boxes(rng, G, vol=(128, 128, 32), size_xy=(20, 50), size_z=(10, 24)))
        if tag == 'rpn_shipped':            # one gt equal to an anchor: IoU exactly 1, and a tie for step 4
            gt[0] = anchors[5000]
        valid = torch.ones(anchors.size(0), dtype=torch.uint8)
        cfg = types.SimpleNamespace(
            assigner=ref.MaxIoUAssigner(pos_iou_thr=pos_thr, neg_iou_thr=neg_thr, min_pos_iou=min_pos, ignore_iof_thr=-1),
            sampler=ref.RandomSampler(num=num, pos_fraction=frac, neg_pos_ub=-1, add_gt_as_proposals=False),
            allowed_border=border, pos_weight=pos_weight)
        rec = _RecordRandint(1234)
        np.random.randint = rec
        try:
            inside = ref.at.anchor_inside_flags(anchors, valid, img_shape, border)
            ar = cfg.assigner.assign(anchors[inside.bool(), :], gt, None, None)
            rec.draws.clear(); rec.rng = np.random.RandomState(1234)
            res = ref.at.anchor_target_single(anchors, valid, gt, None, None, dict(img_shape=img_shape),
                                              [0.] * 6, [1.] * 6, cfg)
        finally:
            np.random.randint = real_randint
        labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds, inside_flags = res
        d.update({f'{tag}_anchors': anchors.numpy(), f'{tag}_gt': gt.numpy(), f'{tag}_img_shape': np.array(img_shape),
                  f'{tag}_cfg': np.array([pos_thr, neg_thr, min_pos, num, frac, pos_weight, border], np.float64),
                  f'{tag}_assigned_gt_inds': ar.gt_inds.numpy(), f'{tag}_max_overlaps': ar.max_overlaps.numpy(),
                  f'{tag}_labels': labels.numpy(), f'{tag}_label_weights': label_weights.numpy(),
                  f'{tag}_bbox_targets': bbox_targets.numpy(), f'{tag}_bbox_weights': bbox_weights.numpy(),
                  f'{tag}_pos_inds': pos_inds.numpy(), f'{tag}_neg_inds': neg_inds.numpy(),
                  f'{tag}_inside_flags': inside_flags.numpy().astype(np.uint8),
                  f'{tag}_num_draws': np.int64(len(rec.draws))})
        for k, dr in enumerate(rec.draws):
            d[f'{tag}_draw{k}'] = dr
        print(f'  targets {tag}: A={anchors.size(0)} inside={int(inside_flags.sum())} pos={pos_inds.numel()} '
              f'neg={neg_inds.numel()} draws={[len(x) for x in rec.draws]}')
    # ---- RCNN stage: proposals clustered around the gts, add_gt_as_proposals=True
    for tag, (P, G, num, frac, cluster) in {'rcnn_shipped': (2000, 5, 512, 0.25, 12), 'rcnn_fewpos': (300, 3, 512, 0.25, 0),
                                            'rcnn_small': (40, 2, 16, 0.5, 3)}.items():
        gt = torch.from_numpy(rand_boxes(rng, G, size_xy=(20, 60), size_z=(6, 20)))
        props = rand_boxes(rng, P, size_xy=(15, 70), size_z=(5, 24), cluster=cluster)
        k = P // 3                              # a third of the proposals are jittered copies of the gts
        src = gt.numpy()[rng.integers(0, G, k)]
        props[:k] = src + rng.normal(0, 2.0, size=(k, 6)).astype(np.float32)
        props = torch.from_numpy(np.concatenate([props, rng.uniform(0, 1, (P, 1)).astype(np.float32)], 1))   # [P,7] with score
        gt_labels = torch.ones(G, dtype=torch.long)
        asg = ref.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=-1)
        smp = ref.RandomSampler(num=num, pos_fraction=frac, neg_pos_ub=-1, add_gt_as_proposals=True)
        rec = _RecordRandint(4321)
        np.random.randint = rec
        try:
            ar = asg.assign(props, gt, None, gt_labels)
            gt_inds0, max_ov0, labels0 = ar.gt_inds.clone(), ar.max_overlaps.clone(), ar.labels.clone()
            sr = smp.sample(ar, props, gt, gt_labels)
        finally:
            np.random.randint = real_randint
        cfg = types.SimpleNamespace(pos_weight=3)
        lab, lw, bt_, bw = ref.bt.bbox_target_single_3d(sr.pos_bboxes, sr.neg_bboxes, sr.pos_gt_bboxes, sr.pos_gt_labels, cfg,
                                                        1, [0.] * 6, [0.1, 0.1, 0.2, 0.2, 0.1, 0.2])
        rois = ref.tr.bbox2roi3D([sr.bboxes])
        d.update({f'{tag}_proposals': props.numpy(), f'{tag}_gt': gt.numpy(), f'{tag}_gt_labels': gt_labels.numpy(),
                  f'{tag}_cfg': np.array([0.5, 0.5, 0.5, num, frac, 3], np.float64),
                  f'{tag}_assigned_gt_inds': gt_inds0.numpy(), f'{tag}_max_overlaps': max_ov0.numpy(),
                  f'{tag}_assigned_labels': labels0.numpy(),
                  f'{tag}_pos_inds': sr.pos_inds.numpy(), f'{tag}_neg_inds': sr.neg_inds.numpy(),
                  f'{tag}_pos_is_gt': sr.pos_is_gt.numpy(), f'{tag}_pos_assigned_gt_inds': sr.pos_assigned_gt_inds.numpy(),
                  f'{tag}_rois': rois.numpy(), f'{tag}_labels': lab.numpy(), f'{tag}_label_weights': lw.numpy(),
                  f'{tag}_bbox_targets': bt_.numpy(), f'{tag}_bbox_weights': bw.numpy(),
                  f'{tag}_target_stds': np.array([0.1, 0.1, 0.2, 0.2, 0.1, 0.2], np.float32),
                  f'{tag}_num_draws': np.int64(len(rec.draws))})
        for k_, dr in enumerate(rec.draws):
            d[f'{tag}_draw{k_}'] = dr
        print(f'  targets {tag}: P={P} pos={sr.pos_inds.numel()} neg={sr.neg_inds.numel()} draws={[len(x) for x in rec.draws]}')
    np.savez_compressed(out, **d)


def make_losses(ref, out):
    L = ref.losses
    g = torch.Generator().manual_seed(5)
    d = {}
    # RPN: weighted BCE with logits on [A,1] scores (sum / avg_factor) and smooth-L1 (beta 1/9) on [A,6]
    A = 3000
    score = torch.randn(A, 1, generator=g, requires_grad=True)
    labels = (torch.rand(A, generator=g) < 0.05).long()
    weights = torch.where(torch.rand(A, generator=g) < 0.1, torch.tensor(1.0), torch.tensor(0.0))
    weights = torch.where(labels > 0, torch.tensor(3.0), weights)
    loss = L.weighted_binary_cross_entropy(score, labels, weights, avg_factor=256)
    loss.backward()
    d.update(bce_score=score.detach().numpy(), bce_labels=labels.numpy(), bce_weights=weights.numpy(), bce_avg=np.float64(256),
             bce_loss=loss.detach().numpy(), bce_grad=score.grad.numpy())
    pred = torch.randn(A, 6, generator=g, requires_grad=True)
    tgt = torch.randn(A, 6, generator=g) * 0.5
    w6 = (labels > 0).float()[:, None].expand(A, 6).contiguous()
    loss = L.weighted_smoothl1(pred, tgt, w6, beta=1 / 9.0, avg_factor=256)
    loss.backward()
    d.update(sl1_pred=pred.detach().numpy(), sl1_target=tgt.numpy(), sl1_weights=w6.numpy(), sl1_beta=np.float64(1 / 9.0),
             sl1_avg=np.float64(256), sl1_loss=loss.detach().numpy(), sl1_grad=pred.grad.numpy())
    # RCNN: weighted softmax CE on [R,2] (avg = #weights>0), accuracy, smooth-L1 (beta 1) on the positives with avg = R
    R = 512
    cls = torch.randn(R, 2, generator=g, requires_grad=True)
    lab = (torch.rand(R, generator=g) < 0.2).long()
    lw = torch.where(lab > 0, torch.tensor(3.0), torch.tensor(1.0))
    loss = L.weighted_cross_entropy(cls, lab, lw)
    loss.backward()
    acc = L.accuracy(cls.detach(), lab)
    d.update(ce_score=cls.detach().numpy(), ce_labels=lab.numpy(), ce_weights=lw.numpy(), ce_loss=loss.detach().numpy(),
             ce_grad=cls.grad.numpy(), ce_acc=acc.numpy())
    bp = torch.randn(R, 6, generator=g, requires_grad=True)
    btg = torch.randn(R, 6, generator=g)
    bw = (lab > 0).float()[:, None].expand(R, 6).contiguous()
    pos = lab > 0
    loss = L.weighted_smoothl1(bp.view(R, 6)[pos], btg[pos], bw[pos], avg_factor=btg.size(0))
    loss.backward()
    d.update(reg_pred=bp.detach().numpy(), reg_target=btg.numpy(), reg_weights=bw.numpy(), reg_loss=loss.detach().numpy(),
             reg_grad=bp.grad.numpy())
    # mask: BCE with logits, mean over [P, 10, 14, 14] of the label-selected channel
    P = 9
    mp = torch.randn(P, 2, 10, 14, 14, generator=g, requires_grad=True)
    mt = (torch.rand(P, 10, 14, 14, generator=g) < 0.4).float()
    ml = torch.ones(P, dtype=torch.long)
    loss = L.mask_cross_entropy(mp, mt, ml)
    loss.backward()
    d.update(mask_pred=mp.detach().numpy(), mask_target=mt.numpy(), mask_labels=ml.numpy(), mask_loss=loss.detach().numpy(),
             mask_grad=mp.grad.numpy())
    np.savez_compressed(out, **d)
    print('  losses:', {k: float(d[k][0]) for k in ('bce_loss', 'sl1_loss', 'ce_loss', 'reg_loss', 'mask_loss')})


def make_state_dict_keys(ref, out):
    """key -> shape of the single-pathway detector's checkpoint (module names of two_stage_3d.py:33-60, head kwargs of
    configs/3d-multi-resolution-rcnn.py:16-79)"""
    import json
    import torch.nn as nn
    m = nn.Module()
    m.backbone = ref.ResNet3D(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, style='pytorch')
    m.neck = ref.FPN3D(in_channels=[64, 128, 256, 512], out_channels=64, num_outs=5)
    m.rpn_head = ref.RPNHead3D(in_channels=64, feat_channels=64, anchor_scales=[2], anchor_depth_scales=[2], anchor_ratios=[1.0],
                               anchor_strides=[4, 8, 16, 32, 64], anchor_strides_depth=[2, 4, 8, 16, 32],
                               target_means=[.0] * 6, target_stds=[1.0] * 6, use_sigmoid_cls=True)
    m.bbox_head = ref.SharedFCBBoxHead3D(num_fcs=2, in_channels=64, fc_out_channels=1024, roi_feat_size=7, roi_feat_size_depth=3,
                                         num_classes=2, target_means=[0.] * 6, target_stds=[0.1, 0.1, 0.2, 0.2, 0.1, 0.1],
                                         reg_class_agnostic=False)
    m.mask_head = ref.FCNMaskHead3D(num_convs=4, in_channels=64, conv_out_channels=64, num_classes=2)
    sd = {k: list(v.shape) for k, v in m.state_dict().items()}
    json.dump(sd, open(out, 'w'), indent=0, sort_keys=True)
    print(f'  state dict: {len(sd)} entries, {sum(int(np.prod(v)) for v in sd.values())} elements')
    # the shipped dual-pathway model (two_stage_3d_2scales.py:33-86, configs/...py:16-85)
    m.rpn_head_2 = ref.RPNHead3D(in_channels=64, feat_channels=64, anchor_scales=[3], anchor_depth_scales=[3], anchor_ratios=[1.0],
                                 anchor_strides=[4, 8, 16, 32, 64], anchor_strides_depth=[2, 4, 8, 16, 32],
                                 target_means=[.0] * 6, target_stds=[1.0] * 6, use_sigmoid_cls=True)
    m.refinement_head = ref.SharedFCBBoxHead3DRefinement(num_fcs=2, in_channels=64, fc_out_channels=1024, roi_feat_size=7,
                                                         roi_feat_size_depth=3, num_classes=2, target_means=[0.] * 6,
                                                         target_stds=[0.1, 0.1, 0.2, 0.2, 0.1, 0.1], reg_class_agnostic=False)
    m.refinement_mask_head = ref.FCNMaskHead3D(num_convs=4, in_channels=64, conv_out_channels=64, num_classes=2)
    sd2 = {k: list(v.shape) for k, v in m.state_dict().items()}
    json.dump(sd2, open(out.replace('.json', '_2scales.json'), 'w'), indent=0, sort_keys=True)
    print(f'  state dict (2 scales): {len(sd2)} entries')


class _FakeCOCO:
    """the four accessors pycocotools_local/cocoeval.py uses (:90-92,:122-135)"""

    def __init__(self, anns):
        self.anns = {a['id']: a for a in anns}

    def getAnnIds(self, imgIds=(), catIds=()):
        return [a['id'] for a in self.anns.values() if (len(imgIds) == 0 or a['image_id'] in imgIds) and
                (len(catIds) == 0 or a['category_id'] in catIds)]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]

    def getImgIds(self):
        return sorted({a['image_id'] for a in self.anns.values()})

    def getCatIds(self):
        return sorted({a['category_id'] for a in self.anns.values()})


def make_eval(ref, out):
    """pycocotools_local/cocoeval.py COCOeval(is3D=True, iouType='bbox'): evaluate -> accumulate -> summarize on a synthetic
    6-volume / 2-category set; boxes in the evaluator's [x, y, w, h, z, d] format (coco_utils.py:233-242)."""
    import contextlib
    import copy
    import io
    if not hasattr(np, 'float'):
        np.float = float            # the reference targets numpy < 1.24 (cocoeval.py:597)
    spec = importlib.util.spec_from_file_location('ref_cocoeval', f'{REF}/pycocotools_local/cocoeval.py')
    ce = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ce)
    rng = np.random.default_rng(11)
    gts, dts, gid, did = [], [], 1, 1
    for img in range(1, 7):
        for cat in (1, 2):
            n = int(rng.integers(0, 6))
            for _ in range(n):
                x, y, z = rng.uniform(0, 400, 2).tolist() + [float(rng.uniform(0, 60))]
                w, h, d = rng.uniform(6, 120, 2).tolist() + [float(rng.uniform(3, 18))]
                box = [float(x), float(y), float(w), float(h), float(z), float(d)]
                gts.append(dict(id=gid, image_id=img, category_id=cat, bbox=box, area=float(w * h), iscrowd=int(rng.random() < 0.1)))
                gid += 1
                for _ in range(int(rng.integers(0, 3))):        # detections around this gt
                    j = rng.normal(0, 0.12, 6) * np.array([w, h, w, h, d, d])
                    b = [box[0] + j[0], box[1] + j[1], max(box[2] + j[2], 1.0), max(box[3] + j[3], 1.0), box[4] + j[4], max(box[5] + j[5], 1.0)]
                    dts.append(dict(id=did, image_id=img, category_id=cat, bbox=[float(v) for v in b], area=float(b[2] * b[3]),
                                    score=float(rng.random()), iscrowd=0))
                    did += 1
            for _ in range(int(rng.integers(0, 4))):            # false positives
                b = rng.uniform(0, 400, 2).tolist() + rng.uniform(6, 100, 2).tolist() + [float(rng.uniform(0, 60)), float(rng.uniform(3, 18))]
                dts.append(dict(id=did, image_id=img, category_id=cat, bbox=[float(v) for v in b], area=float(b[2] * b[3]),
                                score=float(rng.random()), iscrowd=0))
                did += 1
    E = ce.COCOeval(_FakeCOCO(copy.deepcopy(gts)), _FakeCOCO(copy.deepcopy(dts)), 'bbox', is3D=True)
    with contextlib.redirect_stdout(io.StringIO()):
        E.evaluate(); E.accumulate(); E.summarize()
    import json
    json.dump(dict(gts=gts, dts=dts), open(out.replace('.npz', '_input.json'), 'w'))
    np.savez_compressed(out, precision=E.eval['precision'], recall=E.eval['recall'], scores=E.eval['scores'], stats=np.asarray(E.stats),
                        iou_thrs=E.params.iouThrs, rec_thrs=E.params.recThrs, max_dets=np.asarray(E.params.maxDets),
                        area_rng=np.asarray(E.params.areaRng))
    print(f'  eval: {len(gts)} gts, {len(dts)} dts, AP={E.stats[0]:.4f} AP50={E.stats[10]:.4f} AR={E.stats[25]:.4f}')
    # the reference's greedy CPU NMS on a result list (coco_utils.py:245-282), keeping whole records
    boxes = rand_boxes(rng, 60, cluster=6)
    boxes = np.concatenate([boxes, rng.random((60, 1)).astype(np.float32)], 1)
    recs = [dict(k=i) for i in range(60)]
    kept = ref.nms_3d_python(recs, boxes, 0.1)
    np.savez_compressed(out.replace('.npz', '_nms.npz'), boxes=boxes, kept=np.array([r['k'] for r in kept]))


if __name__ == '__main__':
    ref = import_reference()
    which = sys.argv[1:] or ['nms', 'iou', 'codec', 'anchors', 'roi_levels', 'backbone', 'targets', 'losses', 'state_dict', 'eval']
    if 'nms' in which: make_nms(ref, f'{HERE}/nms3d.npz')
    if 'iou' in which: make_iou(ref, f'{HERE}/iou3d.npz')
    if 'codec' in which: make_codec(ref, f'{HERE}/codec.npz')
    if 'anchors' in which: make_anchors(ref, f'{HERE}/anchors.npz')
    if 'roi_levels' in which: make_roi_levels(ref, f'{HERE}/roi_levels.npz')
    if 'backbone' in which: make_backbone(ref, f'{HERE}/backbone_fpn.npz')
    if 'targets' in which: make_targets(ref, f'{HERE}/targets.npz')
    if 'losses' in which: make_losses(ref, f'{HERE}/losses.npz')
    if 'state_dict' in which: make_state_dict_keys(ref, f'{HERE}/state_dict_keys.json')
    if 'eval' in which: make_eval(ref, f'{HERE}/eval3d.npz')
    print('golden vectors written to', HERE)

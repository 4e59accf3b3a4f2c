"""RPN / bbox / mask head forwards on the HIP conv engine, with the reference's parameter names.

  RPNHead3D            mmdet/models/anchor_heads/rpn_head_3d.py:13-35  (+ proposals: rpn_proposals.py)
  SharedFCBBoxHead3D   mmdet/models/bbox_heads/convfc_bbox_head_3d.py:9-183 (num_shared_fcs=2 configuration)
  FCNMaskHead3D        mmdet/models/mask_heads/fcn_mask_head_3d.py:15-97 (+ get_target -> core.mask.mask_target)
Target assignment, sampling and the losses run as HIP kernels too (core/anchor, core/bbox, core/loss)."""
import numpy as np
import torch
import torch.nn as nn

from . import layers
from ..core.mask import mask_target
from ..core.anchor import AnchorGenerator3D, anchor_target
from ..core.bbox import bbox_target_3d, delta2bbox3D
from ..core.post_processing import multiclass_nms_3d
from ..core.mask import paste_masks_3d
from ..core.loss import (weighted_binary_cross_entropy, weighted_smoothl1, weighted_cross_entropy, accuracy,
                         mask_cross_entropy)
from .rpn_proposals import get_bboxes_single


class _RPNHeadPlanFn(torch.autograd.Function):
    """All pyramid levels of the RPN head as TWO plan runs (forward, backward) instead of ~10 autograd conv nodes:
    the shared 3x3x3 conv (+bias+ReLU fused), the fused cls+reg 1x1x1 conv; in backward the ReLU mask rides in the
    epilogue of the 1x1x1 data gradient (no separate mask pass), the four weight gradients of a level run on the
    plan's side stream beside the data-gradient chain, and the shared weights accumulate over the levels in one scratch
    (the kernels add atomically).  Returns, per level, the padded [N, 16, D, H, W] map (channels-last storage): channel
    0..A-1 = cls, A..7A-1 = reg."""

    @staticmethod
    def forward(ctx, w3, b3, w1, b1, *feats):
        from .. import _hip
        dev = feats[0].device
        # a bf16 pyramid (BASELINE configs[4]) runs the head on the bf16 kernels: bf16 activations / activation gradients,
        # fp32 weights, biases, weight gradients and accumulation; the score / delta maps leave in fp32
        bf = feats[0].dtype == torch.bfloat16
        adt = torch.bfloat16 if bf else torch.float32
        pack = _hip.pack_conv_weight_bf16 if bf else _hip.pack_conv_weight
        co1 = w1.shape[0]
        cop = -(-co1 // 16) * 16
        c = w3.shape[0]
        wf3 = pack(w3.detach())                                                          # [27][64][64]
        w1p = torch.zeros((cop, c, 1, 1, 1), dtype=torch.float32, device=dev)
        w1p[:co1] = w1.detach()
        wf1 = pack(w1p)                                                                  # [1][64][16]
        sh3 = b3.detach().contiguous()
        sh1 = torch.zeros(cop, dtype=torch.float32, device=dev)
        sh1[:co1] = b1.detach()
        xs = [f.detach().permute(0, 2, 3, 4, 1).contiguous() for f in feats]             # NDHWC (zero-copy for the engine's maps)
        hs = [torch.empty(x.shape, dtype=adt, device=dev) for x in xs]
        outs = [torch.empty(x.shape[:4] + (cop,), dtype=adt, device=dev) for x in xs]
        ops = []
        for x, h, o in zip(xs, hs, outs):
            n, d, hh, w, _ = x.shape
            i3 = [n, d, hh, w, c, c, 3, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1]              # ..., up dims, relu = 1
            ops.append(_hip.make_op(_hip.OP_CONV_FWD, [x, wf3, h, sh3, None, None, None, None], i3))
            i1 = [n, d, hh, w, c, cop, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
            ops.append(_hip.make_op(_hip.OP_CONV_FWD, [h, wf1, o, sh1, None, None, None, None], i1))
        if bf:
            for o_ in ops:
                o_.i[25] = _hip.DTYPE_BF16
        keep = (wf3, wf1, sh3, sh1)
        with torch.cuda.device(dev):
            _hip.Plan(ops).run(dev)
        ctx.save_for_backward(w3, w1p, *xs, *hs)
        ctx.meta = (len(xs), co1, cop, c, bf)
        ctx.keep = keep
        if bf:
            return tuple(o.float().permute(0, 4, 1, 2, 3) for o in outs)
        return tuple(o.permute(0, 4, 1, 2, 3) for o in outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        from .. import _hip
        L, co1, cop, c, bf = ctx.meta
        adt = torch.bfloat16 if bf else torch.float32
        saved = ctx.saved_tensors
        w3, w1p = saved[0], saved[1]
        xs, hs = saved[2:2 + L], saved[2 + L:2 + 2 * L]
        dev = xs[0].device
        if bf:
            wb3 = _hip.pack_conv_weight_bf16(w3.detach().flip(2, 3, 4).transpose(0, 1).contiguous())
            wb1 = _hip.pack_conv_weight_bf16(w1p.transpose(0, 1).contiguous())
        else:
            wb3 = w3.detach().reshape(c, c, 27).flip(2).permute(2, 0, 1).contiguous()   # [26-tap][co][ci]
            wb1 = w1p.reshape(cop, c, 1).permute(2, 0, 1).contiguous()                   # [1][16][64]
        dw3 = torch.zeros((27, c, c), dtype=torch.float32, device=dev)
        db3 = torch.zeros(c, dtype=torch.float32, device=dev)
        dw1 = torch.zeros((1, c, cop), dtype=torch.float32, device=dev)
        db1 = torch.zeros(cop, dtype=torch.float32, device=dev)
        gxs, ops, keep = [], [], []
        for x, h, g in zip(xs, hs, gouts):
            n, d, hh, w, _ = x.shape
            if g is None:
                gxs.append(None)
                continue
            g16 = g.permute(0, 2, 3, 4, 1).to(adt).contiguous()
            gh = torch.empty(x.shape, dtype=adt, device=dev)
            gx = torch.empty(x.shape, dtype=adt, device=dev)
            keep += [g16, gh]
            side = [0] * 11 + [1]                                                        # i[26] = 1: side stream
            side[2] = _hip.WGRAD_SPARSE_DY          # i[17]: the loss gradient is non-zero at the sampled anchors only
            ops.append(_hip.make_op(_hip.OP_CONV_WGRAD, [h, g16, dw1, db1], [n, d, hh, w, c, cop, 1, 1, 1, 1, 1, 1, 0, 0, 0] + side))
            ops.append(_hip.make_op(_hip.OP_CONV_FWD, [g16, wb1, gh, None, None, None, None, h],
                                    [n, d, hh, w, cop, c, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]))      # dgrad 1x1, mask = h (ReLU)
            ops.append(_hip.make_op(_hip.OP_CONV_WGRAD, [x, gh, dw3, db3], [n, d, hh, w, c, c, 3, 3, 3, 1, 1, 1, 1, 1, 1] + side))
            if d * hh * w >= 65536:
                # gh is non-zero where the loss gradient g16 is (a 1x1x1 conv + mask): its voxel liveness, from the 16-channel
                # map, lets the large-map data-gradient kernel write the tiles whose halo is dead from its epilogue
                lbits = torch.empty((_hip.voxel_liveness_words(n * d * hh * w),), dtype=torch.int32, device=dev)
                keep.append(lbits)
                ops.append(_hip.make_op(_hip.OP_VOXEL_LIVENESS, [g16, lbits], [n, d, hh, w, cop, 1]))
            ops.append(_hip.make_op(_hip.OP_CONV_FWD, [gh, wb3, gx, None, None, None, None, None],
                                    [n, d, hh, w, c, c, 3, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0]))
            gxs.append(gx.permute(0, 4, 1, 2, 3))
        if bf:
            for o_ in ops:
                o_.i[25] = _hip.DTYPE_BF16
        if ops:
            with torch.cuda.device(dev):
                _hip.Plan(ops).run(dev)
        gw3 = dw3.permute(2, 1, 0).reshape(c, c, 3, 3, 3).contiguous()
        gw1 = dw1.permute(2, 1, 0).reshape(cop, c, 1, 1, 1)[:co1].contiguous()
        return (gw3, db3, gw1, db1[:co1].contiguous()) + tuple(gxs)


class RPNHead3D(nn.Module):

    def __init__(self, in_channels=64, feat_channels=64, anchor_scales=(2,), anchor_depth_scales=(2,), anchor_ratios=(1.0,),
                 anchor_strides=(4, 8, 16, 32, 64), anchor_strides_depth=(2, 4, 8, 16, 32),
                 target_means=(0., 0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1., 1.), use_sigmoid_cls=True):
        super().__init__()
        assert use_sigmoid_cls, 'the shipped config uses sigmoid classification (configs/...py:27)'
        self.anchor_strides, self.anchor_strides_depth = list(anchor_strides), list(anchor_strides_depth)
        self.target_means, self.target_stds = list(target_means), list(target_stds)
        self.anchor_generators = [AnchorGenerator3D(s, list(anchor_scales), list(anchor_depth_scales), list(anchor_ratios), d)
                                  for s, d in zip(anchor_strides, anchor_strides_depth)]       # anchor_head_3d.py:66-70
        self.num_anchors = len(anchor_ratios) * len(anchor_scales)
        self.rpn_conv = nn.Conv3d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv3d(feat_channels, self.num_anchors, 1)
        self.rpn_reg = nn.Conv3d(feat_channels, self.num_anchors * 6, 1)
        self.init_weights()

    def init_weights(self):
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):              # normal_init(std=0.01), rpn_head_3d.py:25-28
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)

    def forward_single(self, x):
        h = layers.conv3d(x, self.rpn_conv.weight, self.rpn_conv.bias, padding=1, relu=True)
        # the two 1x1x1 heads share their input: one 64 -> (A + 6A) conv, split afterwards
        w = torch.cat([self.rpn_cls.weight, self.rpn_reg.weight], 0)
        b = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias], 0)
        out = layers.conv3d(h, w, b)
        return out[:, :self.num_anchors], out[:, self.num_anchors:]

    def forward(self, feats):
        if feats[0].is_cuda and self.rpn_conv.in_channels % 16 == 0 and self.rpn_conv.out_channels % 64 == 0:
            # every level in one plan run (two for backward): see _RPNHeadPlanFn
            w1 = torch.cat([self.rpn_cls.weight, self.rpn_reg.weight], 0)
            b1 = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias], 0)
            outs = _RPNHeadPlanFn.apply(self.rpn_conv.weight, self.rpn_conv.bias, w1, b1, *feats)
            A = self.num_anchors
            return [o[:, :A] for o in outs], [o[:, A:7 * A] for o in outs]
        outs = [self.forward_single(f) for f in feats]
        return [o[0] for o in outs], [o[1] for o in outs]

    # ------------------------------------------------------------------ training (anchor_head_3d.py:83-230)
    def get_anchors(self, featmap_sizes, img_metas, device='cuda'):
        """anchor_head_3d.py:83-123.  Anchors and valid flags are functions of the pyramid shape and of the images' padded
        shapes only, so both are generated once per such geometry and cached (the reference rebuilds them every step:
        ~45 tiny launches for the flags of five levels)."""
        key = (tuple(tuple(int(v) for v in fs) for fs in featmap_sizes), str(device))
        if getattr(self, '_anchor_cache_key', None) != key:
            self._anchor_cache = [g.grid_anchors(fs, s, d, device=device) for g, fs, s, d in
                                  zip(self.anchor_generators, featmap_sizes, self.anchor_strides, self.anchor_strides_depth)]
            self._anchor_cache_key = key
            self._flag_cache = {}
        anchor_list = [list(self._anchor_cache) for _ in img_metas]
        valid_flag_list = []
        for meta in img_metas:
            pad = tuple(int(v) for v in meta['pad_shape'])
            if pad not in self._flag_cache:
                flags = []
                for g, (fz, fh, fw), s, ds in zip(self.anchor_generators, featmap_sizes, self.anchor_strides, self.anchor_strides_depth):
                    h, w, _, z = pad
                    valid = (min(int(np.ceil(z / ds)), fz), min(int(np.ceil(h / s)), fh), min(int(np.ceil(w / s)), fw))
                    flags.append(g.valid_flags((fz, fh, fw), valid, device=device))
                self._flag_cache[pad] = flags
            valid_flag_list.append(list(self._flag_cache[pad]))
        return anchor_list, valid_flag_list

    def anchor_geometry(self, featmap_sizes, img_metas, cfg, device='cuda'):
        """Everything about the anchors that depends on shapes alone, cached per geometry: per image the flat anchors, the
        flat valid flags and the inside flags (anchor_target.py:193-216), and the per-level inside flags in the layout
        anchor_target returns them (what the reference caches as `self.pos_indices`, anchor_head_3d.py:212)."""
        from ..core.anchor.anchor_target import anchor_inside_flags, images_to_levels
        border = cfg['allowed_border'] if isinstance(cfg, dict) else cfg.allowed_border
        key = (tuple(tuple(int(v) for v in fs) for fs in featmap_sizes), str(device), int(border),
               tuple((tuple(int(v) for v in m['pad_shape']), tuple(int(v) for v in m['img_shape'])) for m in img_metas))
        if getattr(self, '_geom_key', None) != key:
            anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, device=device)
            flat_anchors = [torch.cat(a) for a in anchor_list[:1]] * len(img_metas)        # the same tensor for every image
            flat_flags = [torch.cat(v) for v in valid_flag_list]
            inside = [anchor_inside_flags(a, f, m['img_shape'], border) for a, f, m in zip(flat_anchors, flat_flags, img_metas)]
            levels = images_to_levels(inside, [a.size(0) for a in anchor_list[0]])
            self._geom = (flat_anchors, flat_flags, inside, levels)
            self._geom_key = key
        return self._geom

    def loss_single(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_total_samples, cfg):
        """anchor_head_3d.py:125-177 (sigmoid classification, no focal loss)"""
        cls_score = cls_score.permute(0, 3, 4, 2, 1).reshape(-1, 1)            # (h, w, d, a) anchor order
        loss_cls = weighted_binary_cross_entropy(cls_score, labels.reshape(-1), label_weights.reshape(-1),
                                                 avg_factor=num_total_samples)
        bbox_pred = bbox_pred.permute(0, 3, 4, 2, 1).reshape(-1, 6)
        beta = cfg['smoothl1_beta'] if isinstance(cfg, dict) else cfg.smoothl1_beta
        loss_reg = weighted_smoothl1(bbox_pred, bbox_targets.reshape(-1, 6), bbox_weights.reshape(-1, 6), beta=beta,
                                     avg_factor=num_total_samples)
        return loss_cls, loss_reg

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, cfg, gt_bboxes_ignore=None, draws_list=None):
        """rpn_head_3d.py:36-70 -> anchor_head_3d.py:179-230.  `draws_list`: per image (pos_words, neg_words) for a
        reproducible sampler; default = the device generator."""
        featmap_sizes = [tuple(f.shape[-3:]) for f in cls_scores]
        flat_anchors, flat_flags, inside, inside_levels = self.anchor_geometry(featmap_sizes, img_metas, cfg, cls_scores[0].device)
        (labels_list, label_weights_list, bbox_targets_list, bbox_weights_list, num_total_pos, num_total_neg, _, _) = \
            anchor_target(flat_anchors, flat_flags, gt_bboxes, img_metas, self.target_means, self.target_stds,
                          cfg, gt_bboxes_ignore_list=gt_bboxes_ignore, gt_labels_list=None, label_channels=1,
                          sampling=True, draws_list=draws_list, inside_flags_list=inside,
                          num_level_anchors=[int(np.prod(fs)) * self.num_anchors for fs in featmap_sizes])
        self.pos_indices = inside_levels                                         # anchor_head_3d.py:212
        num_total_samples = (num_total_pos + num_total_neg).float()             # device scalar: no read-back
        losses_cls, losses_reg = [], []
        for lvl in range(len(cls_scores)):
            lc, lr = self.loss_single(cls_scores[lvl], bbox_preds[lvl], labels_list[lvl], label_weights_list[lvl],
                                      bbox_targets_list[lvl], bbox_weights_list[lvl], num_total_samples, cfg)
            losses_cls.append(lc)
            losses_reg.append(lr)
        return dict(loss_rpn_cls=losses_cls, loss_rpn_reg=losses_reg)

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg, use_pos_indices=False, with_counts=False):
        """per image proposals (anchor_head_3d.py:232-268 -> rpn_head_3d.py:72-149).  use_pos_indices: apply the
        inside-flags cached by the last loss() call on the levels with more than nms_pre anchors, as the reference does
        implicitly whenever the shapes match (rpn_head_3d.py:97-101, anchor_head_3d.py:212).
        with_counts: static shapes for the detectors -- returns (list of [max_num, 7] blocks whose rows behind the count are
        zero, list of int32 [1] device counts) and never synchronises with the host where the sync-free path applies."""
        from .rpn_proposals import get_bboxes_single_device, _device_path_ok, _as_cfg
        flags = getattr(self, 'pos_indices', None) if use_pos_indices else None
        res, counts = [], []
        for i, meta in enumerate(img_metas):
            vf = None
            if flags is not None:
                vf = [f if f.dim() == 1 else f[i] for f in flags]
            args = ([c[i] for c in cls_scores], [r[i] for r in bbox_preds], self.anchor_generators, self.anchor_strides,
                    self.anchor_strides_depth, meta['img_shape'], cfg, self.target_means, self.target_stds)
            if with_counts and _device_path_ok(args[0], _as_cfg(cfg)):
                p, c = get_bboxes_single_device(*args, valid_flags=vf)
            else:
                p = get_bboxes_single(*args, valid_flags=vf)
                if with_counts:
                    c = torch.full((1,), p.size(0), dtype=torch.int32, device=p.device)
                    cap = int(_as_cfg(cfg).max_num)
                    if p.size(0) < cap:
                        p = torch.cat((p, p.new_zeros((cap - p.size(0), 7))))
            res.append(p)
            if with_counts:
                counts.append(c)
        return (res, counts) if with_counts else res


class SharedFCBBoxHead3D(nn.Module):

    def __init__(self, in_channels=64, fc_out_channels=1024, roi_feat_size=7, roi_feat_size_depth=3, num_classes=2,
                 reg_class_agnostic=False, with_cls=True, target_means=(0., 0., 0., 0., 0., 0.),
                 target_stds=(0.1, 0.1, 0.2, 0.2, 0.1, 0.1)):
        super().__init__()
        self.num_classes, self.reg_class_agnostic = num_classes, reg_class_agnostic
        self.target_means, self.target_stds = list(target_means), list(target_stds)
        flat = in_channels * roi_feat_size_depth * roi_feat_size * roi_feat_size
        self.shared_fcs = nn.ModuleList([nn.Linear(flat, fc_out_channels), nn.Linear(fc_out_channels, fc_out_channels)])
        self.with_cls = with_cls                     # the refinement head has no classifier (..._refinement_head.py)
        if with_cls:
            self.fc_cls = nn.Linear(fc_out_channels, num_classes)
        self.fc_reg = nn.Linear(fc_out_channels, 6 if reg_class_agnostic else 6 * num_classes)
        self.init_weights()

    def init_weights(self):
        for m in self.shared_fcs:                                           # convfc_bbox_head_3d.py:122-128
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0)
        if self.with_cls:
            nn.init.normal_(self.fc_cls.weight, 0, 0.01)                    # bbox_head_3d.py:59-66
            nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_reg.bias, 0)

    def forward(self, x):
        x = x.reshape(x.size(0), -1)                                         # (c, d, h, w) flatten order of RoIAlign3D's output
        for fc in self.shared_fcs:
            x = layers.linear(x, fc.weight, fc.bias, relu=True)
        # cls and reg read the same activations: one GEMM with stacked rows
        if self.with_cls:
            w = torch.cat([self.fc_cls.weight, self.fc_reg.weight], 0)
            b = torch.cat([self.fc_cls.bias, self.fc_reg.bias], 0)
            out = layers.linear(x, w, b)
            nc = self.fc_cls.weight.shape[0]
            return out[:, :nc], out[:, nc:]
        return None, layers.linear(x, self.fc_reg.weight, self.fc_reg.bias)

    def get_target(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg):
        """bbox_head_3d.py:68-84"""
        return bbox_target_3d([r.pos_bboxes for r in sampling_results], [r.neg_bboxes for r in sampling_results],
                              [r.pos_gt_bboxes for r in sampling_results], [r.pos_gt_labels for r in sampling_results],
                              rcnn_train_cfg, 1 if self.reg_class_agnostic else self.num_classes,
                              target_means=self.target_means, target_stds=self.target_stds)

    def get_det_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None, valid_rows=None):
        """bbox_head_3d.py:136-170: softmax scores, per-class delta2bbox3D decode clipped to img_shape, optional
        rescale (all 6 coordinates divided, as the reference does), multiclass NMS3D.  valid_rows (device scalar): rows
        behind it are the zero padding of a fixed-capacity proposal block; their scores are set to -1, below any score_thr."""
        scores = torch.softmax(cls_score, dim=1) if cls_score is not None else None
        if valid_rows is not None and scores is not None:
            live = torch.arange(scores.size(0), device=scores.device) < valid_rows.reshape(())
            scores = torch.where(live[:, None], scores, scores.new_full((), -1.0))
        if bbox_pred is not None:
            bboxes = delta2bbox3D(rois[:, 1:].contiguous(), bbox_pred.contiguous(), self.target_means, self.target_stds, img_shape)
        else:
            bboxes = rois[:, 1:]
        if rescale:
            bboxes = bboxes / scale_factor
        if cfg is None:
            return bboxes, scores
        get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
        return multiclass_nms_3d(bboxes, scores, get('score_thr'), dict(get('nms')), get('max_per_img'))

    def loss(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, valid_rows=None, num_rois=None):
        """bbox_head_3d.py:86-135.  The regression loss is evaluated densely: the reference gathers the positive rows
        (`bbox_pred.view(R,-1,6)[pos_inds, labels[pos_inds]]`), here every row selects its label's slot and the
        negatives / padding rows carry weight 0 -- the same sum.  valid_rows / num_rois (device scalars) describe a
        padded static-shape roi batch: rows >= valid_rows are ignored, avg_factor = num_rois instead of R."""
        losses = dict()
        if cls_score is not None:
            losses['loss_cls'] = weighted_cross_entropy(cls_score, labels, label_weights, valid_rows=valid_rows)
            losses['acc'] = accuracy(cls_score, labels, valid_rows=valid_rows)
        if bbox_pred is not None:
            R = bbox_pred.size(0)
            if self.reg_class_agnostic:
                sel = bbox_pred.view(R, 6)
            else:
                idx = labels.view(R, 1, 1).expand(R, 1, 6)
                sel = bbox_pred.view(R, -1, 6).gather(1, idx).squeeze(1)
            avg = float(bbox_targets.size(0)) if num_rois is None else num_rois
            losses['loss_reg'] = weighted_smoothl1(sel, bbox_targets, bbox_weights, avg_factor=avg)
        return losses


def rpn_get_bboxes_pair(head_a, outs_a, metas_a, head_b, outs_b, metas_b, cfg, use_pos_indices=False):
    """`head_a.get_bboxes(*outs_a, metas_a, cfg, with_counts=True)` and the same for the second pathway's head
    (two_stage_3d_2scales.py:170-176) -- with ONE NMS3D launch sequence over both pathways' level sets when each pathway holds
    one image and the sync-free path applies (rpn_proposals.get_bboxes_pair_device); otherwise the two separate calls.
    Returns ((proposal list, counts), (proposal list 2, counts 2))."""
    from .rpn_proposals import get_bboxes_pair_device, _device_path_ok, _as_cfg
    c = _as_cfg(cfg)
    if len(metas_a) == 1 and len(metas_b) == 1 and _device_path_ok([t[0] for t in outs_a[0]], c) and \
            _device_path_ok([t[0] for t in outs_b[0]], c):
        def pack(head, outs, meta):
            flags = getattr(head, 'pos_indices', None) if use_pos_indices else None
            vf = None if flags is None else [f if f.dim() == 1 else f[0] for f in flags]
            return ([t[0] for t in outs[0]], [t[0] for t in outs[1]], head.anchor_generators, head.anchor_strides,
                    head.anchor_strides_depth, meta['img_shape'], head.target_means, head.target_stds), vf
        args_a, vf_a = pack(head_a, outs_a, metas_a[0])
        args_b, vf_b = pack(head_b, outs_b, metas_b[0])
        (pa, ca), (pb, cb_) = get_bboxes_pair_device(args_a, args_b, c, vf_a, vf_b)
        return ([pa], [ca]), ([pb], [cb_])
    return (head_a.get_bboxes(*outs_a, metas_a, cfg, use_pos_indices=use_pos_indices, with_counts=True),
            head_b.get_bboxes(*outs_b, metas_b, cfg, use_pos_indices=use_pos_indices, with_counts=True))


class _ConvModule(nn.Module):
    """parameter container named like ConvModule3D (`convs.0.conv.weight`, conv_module3d.py:34)"""

    def __init__(self, cin, cout, k, pad):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, k, padding=pad)
        self.pad = pad
        nn.init.kaiming_normal_(self.conv.weight, mode='fan_out', nonlinearity='relu')     # conv_module3d.py:70-74
        nn.init.constant_(self.conv.bias, 0)

    def forward(self, x):
        return layers.conv3d(x, self.conv.weight, self.conv.bias, padding=self.pad, relu=True)


class _MaskHeadPlanFn(torch.autograd.Function):
    """FCNMaskHead3D.forward (fcn_mask_head_3d.py:88-97: 4 x [3x3x3 conv + ReLU] -> ConvTranspose3d(2, stride 2) + ReLU
    -> 1x1x1 logits) as ONE plan run, and its backward as one more: every ReLU mask rides in the epilogue of the data
    gradient that produces the masked tensor, the six weight gradients run on the plan's side stream, and the 8 taps of
    the transposed conv scatter straight into the up-sampled map.  args: x, bf16 flag, then (weight, bias) of the convs, the
    deconv and the logits conv.
    bf16 = True (BASELINE configs[4]): activations and activation gradients of the head are stored in bf16 and every product
    runs on the bf16 MFMA kernels (fp32 accumulation; weights, biases and their gradients stay fp32; the logits leave in
    fp32).  The bf16 kernels have no kernel-2 instance, so the transposed conv's backward is written as 8 stride-2 POINTWISE
    passes: its data gradient accumulates the taps through the residual-add epilogue (the ReLU mask on the last one), its
    weight gradient is one stride-2 pointwise weight gradient per tap with the roles of input and output swapped."""

    @staticmethod
    def forward(ctx, x, bf, *params):
        from .. import _hip
        dev = x.device
        nconv = (len(params) - 4) // 2
        convs = [(params[2 * i], params[2 * i + 1]) for i in range(nconv)]
        wd, bd, wl, bl = params[2 * nconv:]
        c = convs[0][0].shape[0]
        col = wl.shape[0]
        colp = -(-col // 16) * 16
        adt = torch.bfloat16 if bf else torch.float32
        esz = 2 if bf else 4
        pack = _hip.pack_conv_weight_bf16 if bf else _hip.pack_conv_weight
        xc = x.detach().permute(0, 2, 3, 4, 1).to(adt).contiguous()
        N, D, H, W, cin = xc.shape
        acts, ops, keep = [xc], [], []
        for w, b in convs:
            wf, sh = pack(w.detach()), b.detach().contiguous()
            a = torch.empty((N, D, H, W, c), dtype=adt, device=dev)
            ops.append(_hip.make_op(_hip.OP_CONV_FWD, [acts[-1], wf, a, sh, None, None, None, None],
                                    [N, D, H, W, acts[-1].shape[4], c, 3, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1]))
            keep += [wf, sh]
            acts.append(a)
        shd = bd.detach().contiguous()
        up = torch.empty((N, 2 * D, 2 * H, 2 * W, c), dtype=adt, device=dev)
        if bf:
            wdf = _hip.pack_conv_weight_bf16(wd.detach().transpose(0, 1).contiguous())     # [8 taps][ci/8][co][8]
            tap_bytes = wdf[0].numel() * 2
        else:
            wdf = wd.detach().permute(2, 3, 4, 0, 1).reshape(8, c, c).contiguous()          # [tap][ci][co]
            tap_bytes = 4 * c * c
        for t in range(8):
            a_, b_, c_ = t >> 2, (t >> 1) & 1, t & 1
            off = ((a_ * 2 * H + b_) * 2 * W + c_) * c
            ops.append(_hip.make_op(_hip.OP_CONV_FWD, [acts[-1], wdf.data_ptr() + t * tap_bytes, up.data_ptr() + esz * off, shd,
                                                        None, None, None, None],
                                    [N, D, H, W, c, c, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 2, 2, 2, 2 * D, 2 * H, 2 * W]))
        wlp = torch.zeros((colp, c, 1, 1, 1), dtype=torch.float32, device=dev)
        wlp[:col] = wl.detach()
        wlf = pack(wlp)
        shl = torch.zeros(colp, dtype=torch.float32, device=dev)
        shl[:col] = bl.detach()
        out = torch.empty((N, 2 * D, 2 * H, 2 * W, colp), dtype=adt, device=dev)
        ops.append(_hip.make_op(_hip.OP_CONV_FWD, [up, wlf, out, shl, None, None, None, None],
                                [N, 2 * D, 2 * H, 2 * W, c, colp, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]))
        keep += [wdf, shd, wlf, shl]
        if bf:
            for o_ in ops:
                o_.i[25] = _hip.DTYPE_BF16
        with torch.cuda.device(dev):
            _hip.Plan(ops).run(dev)
        ctx.save_for_backward(*params, *acts, up, wlp)
        ctx.meta = (nconv, c, col, colp, bf)
        ctx.keep = keep
        out = out.float() if bf else out
        return out.permute(0, 4, 1, 2, 3)[:, :col]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        from .. import _hip
        nconv, c, col, colp, bf = ctx.meta
        adt = torch.bfloat16 if bf else torch.float32
        esz = 2 if bf else 4
        sv = ctx.saved_tensors
        params = sv[:2 * nconv + 4]
        acts = sv[2 * nconv + 4:2 * nconv + 4 + nconv + 1]
        up, wlp = sv[-2], sv[-1]
        wd = params[2 * nconv]
        dev = up.device
        N, D, H, W, cin = acts[0].shape
        side = [1]
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        g16 = torch.zeros((N, 2 * D, 2 * H, 2 * W, colp), dtype=adt, device=dev)
        g16[..., :col] = gout.permute(0, 2, 3, 4, 1)
        ops, keep = [], [g16]

        def wgrad(xin, dy, dw, db, k, s, p, xdims, cin_, cout_):
            n_, d_, h_, w_ = xdims
            ints = [n_, d_, h_, w_, cin_, cout_, k, k, k, s, s, s, p, p, p] + [0] * 11 + side
            ops.append(_hip.make_op(_hip.OP_CONV_WGRAD, [xin, dy, dw, db], ints))

        def dgrad(dy, wb, dx, mask, k, s, p, dims, cin_, cout_, add1=None):
            n_, d_, h_, w_ = dims
            ops.append(_hip.make_op(_hip.OP_CONV_FWD, [dy, wb, dx, None, add1, None, None, mask],
                                    [n_, d_, h_, w_, cin_, cout_, k, k, k, s, s, s, p, p, p, 0, 0, 0, 0]))

        # logits 1x1x1
        dwl, dbl = z(1, c, colp), z(colp)
        wgrad(up, g16, dwl, dbl, 1, 1, 0, (N, 2 * D, 2 * H, 2 * W), c, colp)
        gup = torch.empty(up.shape, dtype=adt, device=dev)
        if bf:
            wbl = _hip.pack_conv_weight_bf16(wlp.transpose(0, 1).contiguous())
        else:
            wbl = wlp.reshape(colp, c, 1).permute(2, 0, 1).contiguous()
        dgrad(g16, wbl, gup, up, 1, 1, 0, (N, 2 * D, 2 * H, 2 * W), colp, c)             # ReLU mask of the deconv output
        dwd = z(8, c, c)                                                                 # [tap][conv-in = co][conv-out = ci]
        g = torch.empty(acts[-1].shape, dtype=adt, device=dev)
        if bf:
            # transposed conv backward as 8 stride-2 pointwise passes over gup shifted by the tap (no kernel-2 bf16 instance)
            wbd = _hip.pack_conv_weight_bf16(wd.detach().contiguous())                   # conv weight (out = ci, in = co) per tap
            tap_bytes = wbd[0].numel() * 2
            for t in range(8):
                a_, b_, c_ = t >> 2, (t >> 1) & 1, t & 1
                off = ((a_ * 2 * H + b_) * 2 * W + c_) * c
                gsh = gup.data_ptr() + esz * off
                wgrad(gsh, acts[-1], dwd.data_ptr() + 4 * t * c * c, None, 1, 2, 0, (N, 2 * D, 2 * H, 2 * W), c, c)
                dgrad(gsh, wbd.data_ptr() + t * tap_bytes, g, acts[-1] if t == 7 else None, 1, 2, 0, (N, 2 * D, 2 * H, 2 * W), c, c,
                      add1=g if t > 0 else None)
        else:
            # transposed conv = kernel-2 / stride-2 conv of the gradient (roles swapped for the weight gradient)
            wgrad(gup, acts[-1], dwd, None, 2, 2, 0, (N, 2 * D, 2 * H, 2 * W), c, c)
            wbd = wd.detach().permute(2, 3, 4, 1, 0).reshape(8, c, c).contiguous()
            dgrad(gup, wbd, g, acts[-1], 2, 2, 0, (N, 2 * D, 2 * H, 2 * W), c, c)
        keep += [gup, wbl, wbd]
        # the 3x3x3 convs, last to first
        dws, dbs = [], []
        for i in range(nconv - 1, -1, -1):
            w = params[2 * i]
            ci = acts[i].shape[4]
            dw, db = z(27, ci, c), z(c)
            wgrad(acts[i], g, dw, db, 3, 1, 1, (N, D, H, W), ci, c)
            if bf:
                wb = _hip.pack_conv_weight_bf16(w.detach().flip(2, 3, 4).transpose(0, 1).contiguous())
            else:
                wb = w.detach().reshape(c, ci, 27).flip(2).permute(2, 0, 1).contiguous()
            gprev = torch.empty(acts[i].shape, dtype=adt, device=dev)
            dgrad(g, wb, gprev, acts[i] if i > 0 else None, 3, 1, 1, (N, D, H, W), c, ci)
            keep += [wb, g]
            dws.append((dw, ci)); dbs.append(db)
            g = gprev
        if bf:
            for o_ in ops:
                o_.i[25] = _hip.DTYPE_BF16
        with torch.cuda.device(dev):
            _hip.Plan(ops).run(dev)
        grads = []
        for (dw, ci), db in zip(reversed(dws), reversed(dbs)):
            grads += [dw.permute(2, 1, 0).reshape(c, ci, 3, 3, 3).contiguous(), db]
        gwd = dwd.reshape(2, 2, 2, c, c).permute(4, 3, 0, 1, 2).contiguous()
        gbd = gup.float().sum((0, 1, 2, 3))
        gwl = dwl.permute(2, 1, 0).reshape(colp, c, 1, 1, 1)[:col].contiguous()
        return (g.float().permute(0, 4, 1, 2, 3), None) + tuple(grads) + (gwd, gbd, gwl, dbl[:col].contiguous())


class FCNMaskHead3D(nn.Module):

    def __init__(self, num_convs=4, in_channels=64, conv_kernel_size=3, conv_out_channels=64, upsample_ratio=2,
                 num_classes=2, class_agnostic=False):
        super().__init__()
        assert upsample_ratio == 2, 'deconv kernel 2 / stride 2 (the shipped configuration)'
        self.num_classes, self.class_agnostic = num_classes, class_agnostic
        self.convs = nn.ModuleList([_ConvModule(in_channels if i == 0 else conv_out_channels, conv_out_channels,
                                                conv_kernel_size, (conv_kernel_size - 1) // 2) for i in range(num_convs)])
        self.upsample = nn.ConvTranspose3d(conv_out_channels, conv_out_channels, 2, stride=2)
        self.conv_logits = nn.Conv3d(conv_out_channels, 1 if class_agnostic else num_classes, 1)
        self.init_weights()

    def init_weights(self):
        for m in (self.upsample, self.conv_logits):                          # fcn_mask_head_3d.py:79-86
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        c = self.upsample.weight.shape[0]
        if x.is_cuda and c % 64 == 0 and x.shape[1] % 16 == 0 and all(m.conv.kernel_size == (3, 3, 3) for m in self.convs):
            params = []
            for m in self.convs:
                params += [m.conv.weight, m.conv.bias]
            params += [self.upsample.weight, self.upsample.bias, self.conv_logits.weight, self.conv_logits.bias]
            return _MaskHeadPlanFn.apply(x, bool(getattr(self, 'bf16', False)), *params)      # the whole head: one plan run per pass
        for conv in self.convs:
            x = conv(x)
        x = layers.conv_transpose3d_k2s2(x, self.upsample.weight, self.upsample.bias, relu=True)
        return layers.conv3d(x, self.conv_logits.weight, self.conv_logits.bias)

    def get_target(self, sampling_results, gt_masks, rcnn_train_cfg):
        pos_proposals = [res.pos_bboxes for res in sampling_results]
        pos_assigned_gt_inds = [res.pos_assigned_gt_inds for res in sampling_results]
        return mask_target(pos_proposals, pos_assigned_gt_inds, gt_masks, rcnn_train_cfg)

    def get_seg_masks(self, mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape, scale_factor, rescale):
        """fcn_mask_head_3d.py:126-191: sigmoid, the label's channel, resize to the (truncated) box, threshold, paste
        into a [D,H,W] uint8 volume -- all on the device (one kernel); returns per class lists of uint8 numpy volumes
        like the reference.  ori_shape = (h, w, d) as the reference unpacks it (:155)."""
        cls_segms = [[] for _ in range(self.num_classes - 1)]
        if det_bboxes.shape[0] == 0:
            return cls_segms
        assert rescale, 'the reference only supports rescale=True here (breakpoint() otherwise, :157)'
        img_h, img_w, img_d = ori_shape[:3]
        labels = det_labels + 1
        idx = torch.arange(mask_pred.size(0), device=mask_pred.device)
        ch = torch.zeros_like(labels) if self.class_agnostic else labels
        prob = mask_pred[idx, ch].sigmoid()
        thr = rcnn_test_cfg['mask_thr_binary'] if isinstance(rcnn_test_cfg, dict) else rcnn_test_cfg.mask_thr_binary
        masks = paste_masks_3d(prob, det_bboxes[:, :6] / scale_factor, (img_d, img_h, img_w), thr)
        masks_np, labels_np = masks.cpu().numpy(), labels.cpu().numpy()
        for i in range(masks_np.shape[0]):
            cls_segms[labels_np[i] - 1].append(masks_np[i])
        return cls_segms

    def loss(self, mask_pred, mask_targets, labels, valid_rows=None):
        """fcn_mask_head_3d.py:108-124"""
        lab = torch.zeros_like(labels) if self.class_agnostic else labels
        return dict(loss_mask=mask_cross_entropy(mask_pred, mask_targets, lab, valid_rows=valid_rows))

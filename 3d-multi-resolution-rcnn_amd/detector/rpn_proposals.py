"""RPN proposal generation of one image (RPNHead3D.get_bboxes_single, mmdet/models/anchor_heads/
rpn_head_3d.py:72-149, called from AnchorHead3D.get_bboxes anchor_head_3d.py:232-268).

Per level: sigmoid -> top-`nms_pre` -> [anchors + delta2bbox3D + score concat in ONE kernel] -> NMS3D -> first
`nms_post`; across levels: top-`max_num` by score.  The top-k itself is torch.topk (selection plumbing); anchors
are never materialised for the whole level (the reference builds all 655,360 P2 anchors, then gathers 2000).
The reference's stateful `self.pos_indices` filter (rpn_head_3d.py:95-106: the inside-flags cached by the previous
loss() call drop border-crossing anchors before the top-k) is reproduced through the explicit `valid_flags`
argument; RPNHead3D.get_bboxes passes the cached flags when asked to (forward_train does)."""
from types import SimpleNamespace

import torch

from .. import _hip
from ..mmdet_ops import nms


def _as_cfg(cfg):
    return cfg if not isinstance(cfg, dict) else SimpleNamespace(**cfg)


def _device_path_ok(cls_scores, cfg):
    """The sync-free path serves the configurations whose per-level `[:nms_post]` cut can never bite (nms_post >= nms_pre: the
    shipped train and test configs have both at 2000) -- the cut keeps rows in the order the NMS saw them, which the
    reference makes score order or anchor order depending on a data-dependent branch (:103-112)."""
    return (not getattr(cfg, 'nms_across_levels', False) and 0 < cfg.nms_pre <= cfg.nms_post and cfg.nms_pre <= 4096 and
            len(cls_scores) <= 8 and 0 < cfg.max_num <= 2048 and len(cls_scores) * min(cfg.nms_pre, cfg.nms_post) <= 16384 and
            all(c.dtype == torch.float32 for c in cls_scores))


def get_bboxes_single_device(cls_scores, bbox_preds, anchor_generators, anchor_strides, anchor_strides_depth, img_shape, cfg,
                             target_means=(0., 0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1., 1.), valid_flags=None):
    """get_bboxes_single without a host round trip: (proposals [max_num, 7] with zero rows behind the count, count int32 [1] on
    the device).  Launch sequence: radix select of the nms_pre best valid anchors of all levels (mr3d_rpn_select_topk), one
    decode per level, batched NMS3D with device-side row counts, merge of the levels' kept rows into the max_num best."""
    cfg = _as_cfg(cfg)
    caps = _level_caps(cls_scores, cfg)
    dets = torch.empty((sum(caps), 7), dtype=torch.float32, device=cls_scores[0].device)
    counts = torch.empty(len(caps), dtype=torch.int32, device=dets.device)
    _select_and_decode(cls_scores, bbox_preds, anchor_generators, anchor_strides, anchor_strides_depth, img_shape, cfg,
                       target_means, target_stds, valid_flags, dets, counts)
    keep, num_keep = _hip.nms3d_batched_dev(dets, caps, counts, cfg.nms_thr)
    return _hip.rpn_merge_levels(dets, caps, keep, num_keep, cfg.nms_post, cfg.max_num)


def _level_caps(cls_scores, cfg):
    return [min(int(c.numel()), int(cfg.nms_pre)) for c in cls_scores]


def _select_and_decode(cls_scores, bbox_preds, anchor_generators, anchor_strides, anchor_strides_depth, img_shape, cfg,
                       target_means, target_stds, valid_flags, dets, counts):
    """top-`nms_pre` selection of every level + decode into the caller's `dets` rows / `counts` slots"""
    caps, flat, scores, _ = _hip.rpn_select_topk(cls_scores, valid_flags, cfg.nms_pre, counts_out=counts)
    row = 0
    for lvl, (cls, reg) in enumerate(zip(cls_scores, bbox_preds)):
        A, D, H, W = cls.shape
        if reg.is_contiguous():
            reg_map, layout = reg, _hip.LAYOUT_NCDHW
        else:
            reg_map, layout = reg.permute(1, 2, 3, 0).contiguous(), _hip.LAYOUT_NDHWC
        k = caps[lvl]
        _hip.rpn_decode_level(flat[row:row + k], scores[row:row + k], reg_map, layout, anchor_generators[lvl].base_anchors_on(flat.device),
                              (D, H, W), anchor_strides[lvl], anchor_strides_depth[lvl], target_means, target_stds, img_shape,
                              out=dets[row:row + k])
        row += k
    return caps


def get_bboxes_pair_device(args_a, args_b, cfg, valid_flags_a=None, valid_flags_b=None):
    """The two pathways of the dual-pathway detector (two_stage_3d_2scales.py:170-176: `proposal_list`, `proposal_list_2`) with
    ONE NMS3D launch sequence over both pathways' level sets (10 independent sets; the reference issues 10 nms() calls).
    args_*: (cls_scores, bbox_preds, anchor_generators, anchor_strides, anchor_strides_depth, img_shape, target_means,
    target_stds) of one image of a pathway.  Returns ((proposals, count), (proposals_2, count_2)) as get_bboxes_single_device
    does per pathway -- same rows, bit for bit (the sets are independent problems)."""
    cfg = _as_cfg(cfg)
    caps_a, caps_b = _level_caps(args_a[0], cfg), _level_caps(args_b[0], cfg)
    dev = args_a[0][0].device
    ra = sum(caps_a)
    dets = torch.empty((ra + sum(caps_b), 7), dtype=torch.float32, device=dev)
    counts = torch.empty(len(caps_a) + len(caps_b), dtype=torch.int32, device=dev)
    for args, vf, d, c in ((args_a, valid_flags_a, dets[:ra], counts[:len(caps_a)]),
                           (args_b, valid_flags_b, dets[ra:], counts[len(caps_a):])):
        cls, reg, gens, st, std, shape, means, stds = args
        _select_and_decode(cls, reg, gens, st, std, shape, cfg, means, stds, vf, d, c)
    keep, num_keep = _hip.nms3d_batched_dev(dets, caps_a + caps_b, counts, cfg.nms_thr)
    out_a = _hip.rpn_merge_levels(dets[:ra], caps_a, keep[:ra], num_keep[:len(caps_a)], cfg.nms_post, cfg.max_num)
    out_b = _hip.rpn_merge_levels(dets[ra:], caps_b, keep[ra:], num_keep[len(caps_a):], cfg.nms_post, cfg.max_num)
    return out_a, out_b


def get_bboxes_single(cls_scores, bbox_preds, anchor_generators, anchor_strides, anchor_strides_depth, img_shape, cfg,
                      target_means=(0., 0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1., 1.), valid_flags=None):
    """cls_scores[l]: [A, D, H, W] logits (any memory format); bbox_preds[l]: [A*6, D, H, W];
    img_shape = (H, W, 3, D).  Returns proposals [<= max_num, 7] = [x1,y1,x2,y2,z1,z2,score] (exact row count: one 4-byte
    read-back; the detectors use get_bboxes_single_device and never read it)."""
    cfg = _as_cfg(cfg)
    if _device_path_ok(cls_scores, cfg):
        props, count = get_bboxes_single_device(cls_scores, bbox_preds, anchor_generators, anchor_strides, anchor_strides_depth,
                                                img_shape, cfg, target_means, target_stds, valid_flags)
        return props[:int(count.item())]
    mlvl, real = [], {}
    for lvl, (cls, reg) in enumerate(zip(cls_scores, bbox_preds)):
        assert cls.shape[-3:] == reg.shape[-3:]
        A, D, H, W = cls.shape
        gen = anchor_generators[lvl]
        scores = cls.permute(2, 3, 1, 0).reshape(-1).sigmoid()                 # flat (h, w, d, a) order (:87-90)
        if cfg.nms_pre > 0 and scores.shape[0] > cfg.nms_pre:
            # rpn_head_3d.py:95-106: on levels with more anchors than nms_pre the anchors flagged out (the inside-flags
            # cached by the previous loss() call) are dropped BEFORE the top-k.  Masking their score keeps the selection
            # on the device (the reference's boolean gather synchronises once per level).  When fewer than nms_pre
            # anchors are flagged in, the surplus picks carry score -1 and sort last: they ride through the NMS as
            # fillers behind the real rows (they cannot suppress one) and are cut from its kept list on the device
            # (`real_rows`), so the level contributes exactly the reference's rows (:95-112).
            if valid_flags is not None and valid_flags[lvl].shape == scores.shape:
                vf = valid_flags[lvl].bool()
                scores = torch.where(vf, scores, scores.new_full((), -1.0))
                real[lvl] = vf.sum().clamp(max=cfg.nms_pre)
            scores, flat = scores.topk(cfg.nms_pre)
            if lvl in real and cfg.nms_post < cfg.nms_pre:
                # Fewer valid anchors than nms_pre: the reference skips the top-k and runs its NMS on ALL of them in ANCHOR
                # order (:103-107), so its `[:nms_post]` cut (:135) keeps the first rows by anchor index, not by score.  Only
                # observable when nms_post < nms_pre (the shipped config has both at 2000): the real rows are then re-ordered
                # by anchor index whenever they are all that is valid (decided on the device).
                key = torch.where(scores >= 0, flat, torch.full_like(flat, torch.iinfo(flat.dtype).max))
                by_anchor = torch.argsort(key)
                pick = torch.where(real[lvl] < cfg.nms_pre, by_anchor, torch.arange(flat.numel(), device=flat.device))
                scores, flat = scores[pick], flat[pick]
        else:
            flat = torch.arange(scores.shape[0], device=scores.device)
        # regression map of this level, as the kernel reads it
        if reg.is_contiguous():
            reg_map, layout = reg, _hip.LAYOUT_NCDHW
        else:
            reg_map, layout = reg.permute(1, 2, 3, 0).contiguous(), _hip.LAYOUT_NDHWC
        dets = _hip.rpn_decode_level(flat.contiguous(), scores.contiguous(), reg_map, layout,
                                     gen.base_anchors_on(scores.device), (D, H, W), anchor_strides[lvl],
                                     anchor_strides_depth[lvl], target_means, target_stds, img_shape)
        mlvl.append(dets)
    # the reference calls nms() once per level (:134); the levels are independent, so they share one batched launch
    # sequence and one read-back of the kept counts
    keeps = _hip.nms3d_batched(mlvl, cfg.nms_thr, real_rows=[real.get(l) for l in range(len(mlvl))] if real else None)
    mlvl = [d[k][:cfg.nms_post, :] for d, k in zip(mlvl, keeps)]
    proposals = torch.cat(mlvl, 0)
    if getattr(cfg, 'nms_across_levels', False):
        proposals, _ = nms(proposals, cfg.nms_thr)
        proposals = proposals[:cfg.max_num, :]
    else:
        num = min(cfg.max_num, proposals.shape[0])
        _, top = proposals[:, 6].topk(num)
        proposals = proposals[top, :]
    return proposals

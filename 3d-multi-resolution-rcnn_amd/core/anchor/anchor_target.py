"""anchor_target / anchor_inside_flags (mmdet/core/anchor/anchor_target.py:9-216) on the device.

Per image: inside flags -> fused max-IoU assignment over ALL anchors with the flags as a participation mask (the
reference first compacts the inside subset and `unmap`s every output afterwards, :138-144,180-187) -> device sampler
-> one kernel that writes labels / weights / regression targets for the full anchor set.  No host round trip: the
sample counts stay on the device and `num_total_pos` / `num_total_neg` are returned as device scalars (the reference
returns Python ints, :69-70); their sum feeds the loss kernels as `avg_factor` directly."""
import torch

from ... import _hip
from ..bbox import build_assigner, build_sampler


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    """anchor_target.py:193-216; img_shape = (h, w, c, d)"""
    img_h, img_w, img_d = img_shape[0], img_shape[1], img_shape[3]
    valid_flags = valid_flags.bool()
    if allowed_border < 0:
        return valid_flags
    b = allowed_border
    return valid_flags & (flat_anchors[:, 0] >= -b) & (flat_anchors[:, 1] >= -b) & (flat_anchors[:, 4] >= -b) & \
        (flat_anchors[:, 2] < img_w + b) & (flat_anchors[:, 3] < img_h + b) & (flat_anchors[:, 5] < img_d + b)


def _get(cfg, k):
    return cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)


def anchor_target_single(flat_anchors, valid_flags, gt_bboxes, gt_bboxes_ignore, gt_labels, img_meta, target_means, target_stds,
                         cfg, label_channels=1, sampling=True, unmap_outputs=True, draws=None, inside_flags=None):
    """-> labels, label_weights, bbox_targets, bbox_weights (full anchor set), counts (i32 [4] device), inside_flags.
    inside_flags: precomputed anchor_inside_flags of this geometry (they depend on shapes only; RPNHead3D caches them)."""
    if not sampling:
        raise NotImplementedError('sampling=False (focal loss) is not on the shipped path (use_focal_loss=False)')
    assert unmap_outputs, 'targets are produced for the full anchor set'
    if inside_flags is None:
        inside_flags = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'], _get(cfg, 'allowed_border'))
    assigner, sampler = build_assigner(_get(cfg, 'assigner')), build_sampler(_get(cfg, 'sampler'))
    assert not sampler.add_gt_as_proposals, 'RPN sampling never adds the gts (configs/...py:100)'
    gt = gt_bboxes.float()
    ar = assigner.assign(flat_anchors, gt, gt_bboxes_ignore, None, inside_flags=inside_flags)
    dp, dn = draws if draws is not None else (None, None)
    pos, neg, counts, sel = _hip.random_sample(ar.gt_inds, sampler.num, sampler.pos_fraction, sampler.neg_pos_ub, dp, dn,
                                               want_sel=True)
    labels, label_weights, bbox_targets, bbox_weights = _hip.anchor_target(
        flat_anchors, ar.gt_inds, sel, gt, target_means, target_stds, float(_get(cfg, 'pos_weight')), gt_labels=gt_labels)
    return labels, label_weights, bbox_targets, bbox_weights, counts, inside_flags, pos, neg


def images_to_levels(target, num_level_anchors):
    """anchor_target.py:98-111"""
    target = torch.stack(target, 0) if len(target) > 1 else target[0].unsqueeze(0)      # one image: a view, no copy
    level_targets, start = [], 0
    for n in num_level_anchors:
        level_targets.append(target[:, start:start + n].squeeze(0))
        start += n
    return level_targets


def anchor_target(anchor_list, valid_flag_list, gt_bboxes_list, img_metas, target_means, target_stds, cfg,
                  gt_bboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True, unmap_outputs=True,
                  iteration=None, draws_list=None, inside_flags_list=None, num_level_anchors=None):
    """anchor_list[i]: the per-level anchors of image i, or their concatenation (then pass num_level_anchors)"""
    num_imgs = len(img_metas)
    assert len(anchor_list) == len(valid_flag_list) == num_imgs
    if num_level_anchors is None:
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
    if inside_flags_list is None:
        inside_flags_list = [None] * num_imgs
    flat_anchors = [torch.cat(a) if isinstance(a, (list, tuple)) else a for a in anchor_list]
    flat_flags = [torch.cat(v) if isinstance(v, (list, tuple)) else v for v in valid_flag_list]
    if gt_bboxes_ignore_list is None:
        gt_bboxes_ignore_list = [None] * num_imgs
    if gt_labels_list is None:
        gt_labels_list = [None] * num_imgs
    if draws_list is None:
        draws_list = [None] * num_imgs
    res = [anchor_target_single(flat_anchors[i], flat_flags[i], gt_bboxes_list[i], gt_bboxes_ignore_list[i], gt_labels_list[i],
                                img_metas[i], target_means, target_stds, cfg, label_channels, sampling, unmap_outputs,
                                draws=draws_list[i], inside_flags=inside_flags_list[i]) for i in range(num_imgs)]
    all_labels, all_label_weights, all_bbox_targets, all_bbox_weights, counts, inside, _, _ = (list(x) for x in zip(*res))
    counts = torch.stack(counts)                                              # [num_imgs, 4] on the device
    one = counts.new_ones(())
    num_total_pos = torch.maximum(counts[:, 0], one).sum()                    # sum_i max(#pos_i, 1)   (:69)
    num_total_neg = torch.maximum(counts[:, 1], one).sum()
    return (images_to_levels(all_labels, num_level_anchors), images_to_levels(all_label_weights, num_level_anchors),
            images_to_levels(all_bbox_targets, num_level_anchors), images_to_levels(all_bbox_weights, num_level_anchors),
            num_total_pos, num_total_neg, images_to_levels(flat_anchors, num_level_anchors),
            images_to_levels(inside, num_level_anchors))

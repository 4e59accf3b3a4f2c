"""bbox_target_3d (mmdet/core/bbox/bbox_target.py:37-63,139-165): labels / weights / regression targets of the
sampled rois.  List-in, tensor-out like the reference; the encode is the HIP bbox2delta3d kernel.  The detector's
static-shape path uses the fused `_hip.bbox_target3d` instead (one launch, no host sync)."""
import torch

from ... import _hip


def bbox_target_single_3d(pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg, reg_classes=1,
                          target_means=(.0, .0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0)):
    num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
    num_samples = num_pos + num_neg
    labels = pos_bboxes.new_zeros(num_samples, dtype=torch.long)
    label_weights = pos_bboxes.new_zeros(num_samples)
    bbox_targets = pos_bboxes.new_zeros(num_samples, 6)
    bbox_weights = pos_bboxes.new_zeros(num_samples, 6)
    pos_weight = cfg['pos_weight'] if isinstance(cfg, dict) else cfg.pos_weight
    if num_pos > 0:
        labels[:num_pos] = pos_gt_labels
        label_weights[:num_pos] = 1.0 if pos_weight <= 0 else pos_weight
        bbox_targets[:num_pos, :] = _hip.bbox2delta3d(pos_bboxes[:, :6].contiguous(), pos_gt_bboxes[:, :6].contiguous().float(),
                                                      target_means, target_stds)
        bbox_weights[:num_pos, :] = 1
    if num_neg > 0:
        label_weights[-num_neg:] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights


def bbox_target_3d(pos_bboxes_list, neg_bboxes_list, pos_gt_bboxes_list, pos_gt_labels_list, cfg, reg_classes=1,
                   target_means=(.0, .0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0), concat=True):
    res = [bbox_target_single_3d(p, n, g, l, cfg, reg_classes, target_means, target_stds)
           for p, n, g, l in zip(pos_bboxes_list, neg_bboxes_list, pos_gt_bboxes_list, pos_gt_labels_list)]
    labels, label_weights, bbox_targets, bbox_weights = (list(x) for x in zip(*res))
    if concat:
        return torch.cat(labels, 0), torch.cat(label_weights, 0), torch.cat(bbox_targets, 0), torch.cat(bbox_weights, 0)
    return labels, label_weights, bbox_targets, bbox_weights

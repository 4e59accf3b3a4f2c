"""bbox_target_3d (mmdet/core/bbox/bbox_target.py:37-63,139-165): labels / weights / regression targets of the
sampled rois.  List-in, tensor-out like the reference; the encode is the HIP bbox2delta3d kernel.  The detector's
static-shape path uses the fused `_hip.bbox_target3d` instead (one launch, no host sync)."""
import torch

from ... import _hip


def bbox_target_single_3d(pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg, reg_classes=1,
                          target_means=(.0, .0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0)):
    """One image's sampled rois -> (labels, label_weights, bbox_targets, bbox_weights), positives first.  The fused device
    kernel of the detector path (`_hip.bbox_target3d`) does the work: the sampled boxes are handed to it as a candidate
    list in which positive j is assigned to row j of `pos_gt_bboxes`."""
    P, Q = pos_bboxes.size(0), neg_bboxes.size(0)
    dev = pos_bboxes.device
    if P + Q == 0:
        z = pos_bboxes.new_zeros
        return z(0, dtype=torch.long), z(0), z(0, 6), z(0, 6)
    pos_weight = cfg['pos_weight'] if isinstance(cfg, dict) else cfg.pos_weight
    boxes = torch.cat((pos_bboxes[:, :6], neg_bboxes[:, :6])).float().contiguous()
    # the kernel wants at least one gt row; with no positives none of them is ever read
    gt = pos_gt_bboxes[:, :6].float().contiguous() if P else boxes[:1]
    gl = pos_gt_labels.long().contiguous() if P else torch.zeros(1, dtype=torch.long, device=dev)
    G = gt.size(0)
    ar = torch.arange(P + Q, device=dev)
    assigned = torch.cat((torch.zeros(G, dtype=torch.long, device=dev), torch.where(ar < P, ar + 1, torch.zeros_like(ar))))
    counts = torch.tensor([P, Q, P, Q], dtype=torch.int32, device=dev)
    _, labels, label_weights, bbox_targets, bbox_weights, _ = _hip.bbox_target3d(
        boxes, gt, gl, assigned, G + ar[:max(P, 1)], G + ar[P:] if Q else ar[:1], counts, P + Q, 0, target_means, target_stds,
        float(pos_weight))                   # (an empty index list still needs a valid address)
    return labels, label_weights, bbox_targets, bbox_weights


def bbox_target_3d(pos_bboxes_list, neg_bboxes_list, pos_gt_bboxes_list, pos_gt_labels_list, cfg, reg_classes=1,
                   target_means=(.0, .0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0), concat=True):
    res = [bbox_target_single_3d(p, n, g, l, cfg, reg_classes, target_means, target_stds)
           for p, n, g, l in zip(pos_bboxes_list, neg_bboxes_list, pos_gt_bboxes_list, pos_gt_labels_list)]
    labels, label_weights, bbox_targets, bbox_weights = (list(x) for x in zip(*res))
    if concat:
        return torch.cat(labels, 0), torch.cat(label_weights, 0), torch.cat(bbox_targets, 0), torch.cat(bbox_weights, 0)
    return labels, label_weights, bbox_targets, bbox_weights

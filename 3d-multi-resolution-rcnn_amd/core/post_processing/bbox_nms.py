"""multiclass_nms_3d: the test-time detection filter of mmdet/core/post_processing/bbox_nms.py:57-106 as one device op
(csrc/multiclass_nms3d.hip: fixed-size class segments -> one batched NMS3D -> on-device concat / top-k).  Inputs beyond
that op's limits (4096 rows, 16 foreground classes) take a class-by-class walk over the same device NMS3D."""
import torch

from ... import _hip


def _class_by_class(multi_bboxes, multi_scores, score_thr, iou_thr, max_num):
    """any n / class count: one NMS3D launch per foreground class that has a candidate (a host read-back each)"""
    n, num_classes = multi_scores.shape
    shared = multi_bboxes.size(1) == 6
    kept, kept_labels = [], []
    for cls in range(1, num_classes):
        above = (multi_scores[:, cls] > score_thr).nonzero(as_tuple=True)[0]
        if above.numel() == 0:
            continue
        boxes = multi_bboxes[above] if shared else multi_bboxes[above, 6 * cls:6 * cls + 6]
        cand = torch.cat((boxes, multi_scores[above, cls, None]), dim=1).contiguous()
        survivors = cand[_hip.nms3d(cand, iou_thr)]
        kept.append(survivors)
        kept_labels.append(torch.full((survivors.size(0),), cls - 1, dtype=torch.long, device=cand.device))
    if not kept:
        return multi_bboxes.new_zeros((0, 7)), torch.zeros((0,), dtype=torch.long, device=multi_bboxes.device)
    dets, labels = torch.cat(kept), torch.cat(kept_labels)
    if dets.size(0) > max_num:                       # max_num = -1 drops the last row, like the reference's slice
        order = torch.argsort(dets[:, 6], descending=True, stable=True)[:max_num]
        dets, labels = dets[order], labels[order]
    return dets, labels


def multiclass_nms_3d(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    """multi_bboxes [n, 6 | 6*C], multi_scores [n, C] (column 0 = background) -> (dets [k, 7], labels [k]).
    nms_cfg: dict(type='nms', iou_thr=...) as in test_cfg.rcnn.nms (configs/...py:141)."""
    kind = nms_cfg.get('type', 'nms')
    if kind != 'nms':
        raise NotImplementedError(f"nms type '{kind}': only the hard NMS3D of the shipped config is built for gfx950")
    if multi_scores.dim() == 2 and multi_scores.size(0) > 0 and not _hip.multiclass_nms3d_fits(multi_scores.size(0), multi_scores.size(1)):
        return _class_by_class(multi_bboxes.contiguous(), multi_scores.contiguous(), float(score_thr), float(nms_cfg['iou_thr']),
                               int(max_num))
    return _hip.multiclass_nms3d(multi_bboxes, multi_scores, float(score_thr), float(nms_cfg['iou_thr']), int(max_num))

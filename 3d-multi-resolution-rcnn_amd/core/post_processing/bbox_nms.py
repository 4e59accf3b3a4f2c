"""multiclass_nms_3d (mmdet/core/post_processing/bbox_nms.py:57-106): per foreground class score filter ->
NMS3D -> labels -> top-`max_num` by score."""
import torch

from ...mmdet_ops.nms import nms_wrapper


def multiclass_nms_3d(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    num_classes = multi_scores.shape[1]
    bboxes, labels = [], []
    nms_cfg_ = nms_cfg.copy()
    nms_type = nms_cfg_.pop('type', 'nms')
    nms_op = getattr(nms_wrapper, nms_type)          # the reference's getattr dispatch (bbox_nms.py:76-77)
    for i in range(1, num_classes):
        cls_inds = multi_scores[:, i] > score_thr
        if not cls_inds.any():
            continue
        if multi_bboxes.shape[1] == 6:
            _bboxes = multi_bboxes[cls_inds, :]
        else:
            _bboxes = multi_bboxes[cls_inds, i * 6:(i + 1) * 6]
        _scores = multi_scores[cls_inds, i]
        cls_dets = torch.cat([_bboxes, _scores[:, None]], dim=1)
        cls_dets, _ = nms_op(cls_dets, **nms_cfg_)
        bboxes.append(cls_dets)
        labels.append(multi_bboxes.new_full((cls_dets.shape[0],), i - 1, dtype=torch.long))
    if bboxes:
        bboxes = torch.cat(bboxes)
        labels = torch.cat(labels)
        if bboxes.shape[0] > max_num:
            _, inds = bboxes[:, -1].sort(descending=True)
            inds = inds[:max_num]
            bboxes = bboxes[inds]
            labels = labels[inds]
    else:
        bboxes = multi_bboxes.new_zeros((0, 7))
        labels = multi_bboxes.new_zeros((0,), dtype=torch.long)
    return bboxes, labels

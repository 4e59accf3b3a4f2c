"""multiclass_nms_3d: the test-time detection filter of mmdet/core/post_processing/bbox_nms.py:57-106 as one device op
(csrc/multiclass_nms3d.hip: fixed-size class segments -> one batched NMS3D -> on-device concat / top-k)."""
from ... import _hip


def multiclass_nms_3d(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    """multi_bboxes [n, 6 | 6*C], multi_scores [n, C] (column 0 = background) -> (dets [k, 7], labels [k]).
    nms_cfg: dict(type='nms', iou_thr=...) as in test_cfg.rcnn.nms (configs/...py:141)."""
    kind = nms_cfg.get('type', 'nms')
    if kind != 'nms':
        raise NotImplementedError(f"nms type '{kind}': only the hard NMS3D of the shipped config is built for gfx950")
    return _hip.multiclass_nms3d(multi_bboxes, multi_scores, float(score_thr), float(nms_cfg['iou_thr']), int(max_num))

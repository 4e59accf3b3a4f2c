"""MaxIoUAssigner / AssignResult (mmdet/core/bbox/assigners/max_iou_assigner.py:9-196, assign_result.py:4-22) on the
device: one fused two-pass kernel pair (csrc/assign_ops.hip), no [G, N] IoU matrix, no Python loop over the gts."""
import torch

from ... import _hip


class AssignResult(object):

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels

    def add_gt_(self, gt_labels):
        """assign_result.py:13-22: the gts become candidates 0..G-1, each assigned to itself"""
        self.gt_inds = _hip.concat_gt_assign(self.gt_inds, self.num_gts)
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(self.num_gts), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


class MaxIoUAssigner(object):

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True):
        if ignore_iof_thr > 0:
            raise NotImplementedError('gt_bboxes_ignore / ignore_iof_thr > 0 is not on the shipped path (configs/...py:93,116)')
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr, self.ignore_wrt_candidates = ignore_iof_thr, ignore_wrt_candidates

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None, inside_flags=None):
        """bboxes [n, >=6] (a 7th score column is ignored, :90-91); `inside_flags` (build extension): boxes whose flag is
        0 do not take part, exactly as if they had been filtered out before the call (anchor_target.py:138-144)."""
        if isinstance(gt_bboxes, list) and len(gt_bboxes) == 1:
            gt_bboxes = gt_bboxes[0]
        if isinstance(gt_labels, list) and len(gt_labels) == 1:
            gt_labels = gt_labels[0]
        if bboxes.shape[0] == 0 or gt_bboxes.shape[0] == 0:
            raise ValueError('No gt or bboxes')
        if not bboxes.is_cuda:
            raise NotImplementedError('MaxIoUAssigner runs on the MI355X only (no CPU fallback)')
        gt_inds, max_overlaps, labels = _hip.max_iou_assign(bboxes, gt_bboxes.float(), self.pos_iou_thr, self.neg_iou_thr,
                                                            self.min_pos_iou, gt_labels=gt_labels, inside=inside_flags,
                                                            gt_max_assign_all=self.gt_max_assign_all)
        return AssignResult(gt_bboxes.shape[0], gt_inds, max_overlaps, labels=labels)

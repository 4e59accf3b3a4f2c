"""RandomSampler / SamplingResult (mmdet/core/bbox/samplers/random_sampler.py:7-59, base_sampler.py:8-104,
sampling_result.py:4-32) on the device.

The reference samples with numpy on the host (`np.random.randint(0, len(gallery), size=num)` with replacement, then
`.unique()`), i.e. two device->host->device round trips per call.  Here the draws are device words (torch's device
generator by default, or an explicit `draws=(pos_words, neg_words)` for reproducible tests), the candidate
compaction, the picks and the final ascending compaction all run as kernels (csrc/assign_ops.hip), and the result
is kept as fixed-capacity lists + device counts.  The exact-length tensors the reference exposes (`pos_inds`,
`pos_bboxes`, ...) are materialised lazily -- that, and only that, synchronises."""
import torch

from ... import _hip


class SamplingResult(object):

    def __init__(self, pos_list, neg_list, counts, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_list, self.neg_list, self.counts = pos_list, neg_list, counts     # padded with -1; counts i32 [4] on device
        self.all_bboxes, self.gt_bboxes, self.assign_result, self.gt_flags = bboxes, gt_bboxes, assign_result, gt_flags
        self.num_gts = gt_bboxes.shape[0]
        self._exact = None

    def _materialise(self):
        if self._exact is None:
            npos, nneg = self.counts[:2].tolist()                                   # the one host sync
            self._exact = (self.pos_list[:npos], self.neg_list[:nneg])
        return self._exact

    pos_inds = property(lambda self: self._materialise()[0])
    neg_inds = property(lambda self: self._materialise()[1])
    pos_bboxes = property(lambda self: self.all_bboxes[self.pos_inds])
    neg_bboxes = property(lambda self: self.all_bboxes[self.neg_inds])
    pos_is_gt = property(lambda self: self.gt_flags[self.pos_inds])
    pos_assigned_gt_inds = property(lambda self: self.assign_result.gt_inds[self.pos_inds] - 1)
    pos_gt_bboxes = property(lambda self: self.gt_bboxes[self.pos_assigned_gt_inds, :])

    @property
    def pos_gt_labels(self):
        return None if self.assign_result.labels is None else self.assign_result.labels[self.pos_inds]

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


class RandomSampler(object):

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction, self.neg_pos_ub = num, pos_fraction, neg_pos_ub
        self.add_gt_as_proposals = add_gt_as_proposals

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, draws=None, **kwargs):
        if isinstance(gt_bboxes, list) and len(gt_bboxes) == 1:
            gt_bboxes = gt_bboxes[0]
        if isinstance(gt_labels, list) and len(gt_labels) == 1:
            gt_labels = gt_labels[0]
        bboxes = bboxes[:, :6]
        gt_flags = bboxes.new_zeros((bboxes.shape[0],), dtype=torch.uint8)
        if self.add_gt_as_proposals:
            bboxes = torch.cat([gt_bboxes[:, :6].to(bboxes.dtype), bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = torch.cat([bboxes.new_ones(gt_bboxes.shape[0], dtype=torch.uint8), gt_flags])
        dp, dn = draws if draws is not None else (None, None)
        pos, neg, counts, _ = _hip.random_sample(assign_result.gt_inds, self.num, self.pos_fraction, self.neg_pos_ub, dp, dn)
        return SamplingResult(pos, neg, counts, bboxes, gt_bboxes, assign_result, gt_flags)

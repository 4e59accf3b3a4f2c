"""TwoStageDetector3D.forward_train (mmdet/models/detectors/two_stage_3d.py:114-224): the single-pathway training
step -- backbone+FPN -> RPN (+loss, proposals) -> assign/sample 512 -> RoIAlign3D (3,7,7) -> bbox head (+loss) ->
RoIAlign3D (10,14,14) on the positives -> mask head (+loss) -- i.e. pathway 1 of the shipped dual-pathway model
(two_stage_3d_2scales.py:135-330) and BASELINE.json configs[2]/[3].

Static shapes, device-resident counts: the roi batch always has `sampler.num` rows (sampled positives, negatives,
then zero-weight padding) and the mask batch int(num * pos_fraction) rows; how many of them are real stays in device
memory and enters the loss kernels as `valid_rows` / `avg_factor`.  The reference synchronises with the host at every
`.item()`, `nonzero`, numpy sampler call and boolean-mask gather of this path (~40 per iteration)."""
import re
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import _hip
from ..core.bbox import build_assigner, build_sampler, bbox2roi3D
from ..core.mask import mask_target_single
from .extract_feat import ExtractFeat3D
from .heads import RPNHead3D, SharedFCBBoxHead3D, FCNMaskHead3D
from .roi_extractor import SingleRoIExtractor, assert_all_flushed, deferred_backward, pyramid_gradient

# configs/3d-multi-resolution-rcnn.py:87-127
DEFAULT_TRAIN_CFG = dict(
    rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, ignore_iof_thr=-1),
             sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
             allowed_border=0, pos_weight=3, smoothl1_beta=1 / 9.0, debug=False),
    rpn_proposal=dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_num=2000, nms_thr=0.7, min_bbox_size=0),
    rcnn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=-1),
              sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
              mask_size=28, mask_size_depth=20, pos_weight=3, debug=False,
              # not in the reference: row capacities of the mask-head batch (see PosCountReadback); None = always the
              # full int(num * pos_fraction) rows with no host synchronisation at all
              mask_batch_buckets=(16, 32, 64, 128)))


# configs/3d-multi-resolution-rcnn.py:131-144
DEFAULT_TEST_CFG = dict(
    rpn=dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_num=2000, nms_thr=0.7, min_bbox_size=0),
    rcnn=dict(score_thr=0.2, nms=dict(type='nms', iou_thr=0.5), max_per_img=2000, mask_thr_binary=0.25))


def bbox2result3D(bboxes, labels, num_classes):
    """mmdet/core/bbox/transforms.py:274-292"""
    import numpy as np
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 7), dtype=np.float32) for _ in range(num_classes - 1)]
    b, l = bboxes.cpu().numpy(), labels.cpu().numpy()
    return [b[l == i, :] for i in range(num_classes - 1)]


MASK_BATCH_BUCKETS = (16, 32, 64, 128)


class PosCountReadback(object):
    """The reference runs its mask head on exactly #positives rois (two_stage_3d.py:194-199), a number the host only
    learns by synchronising.  Here the roi batch has a static capacity with the real rows first, so the mask head could
    run sync-free on all int(num * pos_fraction) = 128 rows -- but while the RPN is young there are far fewer positives
    than that, and 128 rows of a 4-conv + deconv head are ~9 ms of MFMA work on rows that carry no loss.  Compromise:
    the positive count is copied to pinned host memory right after sampling (asynchronously, BEFORE the bbox head is
    enqueued), and read just before the mask head is enqueued -- by then the copy finished long ago relative to the
    device's queue, so the wait does not drain the device -- and the batch is cut to the next bucket capacity.  Rows
    beyond the count still carry zero weight, so the losses do not depend on the bucket."""

    def __init__(self, counts, buckets, cap_pos):
        self.buckets, self.cap_pos = buckets, cap_pos
        if buckets is None:
            return
        self.host = torch.empty(1, dtype=counts.dtype).pin_memory()
        self.host.copy_(counts[:1], non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def rows(self):
        if self.buckets is None:
            return self.cap_pos
        self.event.synchronize()
        n = int(self.host[0])
        for b in sorted(self.buckets):
            if b >= n:
                return min(b, self.cap_pos)
        return self.cap_pos


def front_pack_order(valid_counts, rows_per_img):
    """imgs_per_gpu > 1: every image contributes a fixed-capacity block whose first `valid_counts[i]` rows are real.
    Returns the permutation that moves the real rows of ALL images to the front (stable, so the per-image order is kept)
    without reading the counts back: the packed batch then has one contiguous valid prefix like the single-image one."""
    n = len(valid_counts)
    dev = valid_counts[0].device
    local = torch.arange(rows_per_img, device=dev).repeat(n)
    limit = torch.stack([c.reshape(()) for c in valid_counts]).repeat_interleave(rows_per_img)
    return torch.argsort((local >= limit).to(torch.int8), stable=True)


class TwoStageDetector3D(nn.Module):

    def __init__(self, device='cuda', train_cfg=None, test_cfg=None, with_mask=True, anchor_scales=(2,),
                 anchor_depth_scales=(2,), dtype=torch.float32, neck_dcn=False):
        """dtype / neck_dcn: BASELINE configs[4] switches -- bf16 activation storage in backbone + FPN, and DeformConv3d in
        place of the FPN output convs (detector/dcn_neck.py).  Defaults = the reference's fp32 model."""
        super().__init__()
        self.device = torch.device(device)
        self.train_cfg = train_cfg or DEFAULT_TRAIN_CFG
        self.test_cfg = test_cfg or DEFAULT_TEST_CFG
        self.extract = ExtractFeat3D(device=device, dtype=dtype, neck_dcn=neck_dcn)   # backbone + neck (module names as the reference)
        self.neck_dcn = bool(neck_dcn)
        self.extract.sparse_pyramid_grads = True          # detector training: see ExtractFeat3D.sparse_pyramid_grads
        self.extract.clone_outputs = False                # a training / test step consumes the pyramid before the engine runs again
        if dtype == torch.bfloat16:
            # RPN head and RoIAlign3D read the bf16 pyramid directly; with the deformable neck the lateral sums leave the engine
            # in bf16 and the neck's bf16 kernels take them as they are (detector/dcn_neck.py)
            self.extract.pyramid_dtype = torch.bfloat16
        if neck_dcn:
            from .dcn_neck import DCNNeck3D
            self.dcn_neck = DCNNeck3D(offset_bf16=(dtype == torch.bfloat16)).to(self.device)
        self.rpn_head = RPNHead3D(anchor_scales=anchor_scales, anchor_depth_scales=anchor_depth_scales).to(self.device)
        strides, dstrides = [4, 8, 16, 32], [2, 4, 8, 16]
        self.bbox_roi_extractor = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=7, out_size_depth=3, sample_num=2), 64,
                                                     strides, dstrides)
        self.bbox_head = SharedFCBBoxHead3D().to(self.device)
        self.with_mask = with_mask
        if with_mask:
            self.mask_roi_extractor = SingleRoIExtractor(dict(type='RoIAlign3D', out_size=14, out_size_depth=10, sample_num=2),
                                                         64, strides, dstrides)
            self.mask_head = FCNMaskHead3D().to(self.device)
            self.mask_head.bf16 = dtype == torch.bfloat16            # configs[4]: the head's activations in bf16 too
        self.iteration = 0
        # checkpoint compatibility (two_stage_3d.py:33-60): the reference names its feature extractor `backbone` / `neck`;
        # here both live in one engine module, so its prefix is dropped on save and added on load
        self._register_state_dict_hook(self._strip_extract_prefix)
        self._register_load_state_dict_pre_hook(self._add_extract_prefix)

    _DCN_KEY = re.compile(r'^dcn_neck\.(conv|conv_offset)\.(\d+)\.(.+)$')
    _DCN_REF_KEY = re.compile(r'^neck\.fpn_convs\.(\d+)\.(conv|conv_offset)\.(.+)$')

    @staticmethod
    def _strip_extract_prefix(module, state_dict, prefix, local_metadata):
        for k in [k for k in state_dict if k.startswith(prefix + 'extract.')]:
            state_dict[prefix + k[len(prefix) + len('extract.'):]] = state_dict.pop(k)
        for k in [k for k in state_dict if k.startswith(prefix + 'dcn_neck.')]:         # -> neck.fpn_convs.<i>.{conv,conv_offset}.*
            m = TwoStageDetector3D._DCN_KEY.match(k[len(prefix):])
            state_dict[f'{prefix}neck.fpn_convs.{m.group(2)}.{m.group(1)}.{m.group(3)}'] = state_dict.pop(k)
        return state_dict

    def _add_extract_prefix(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for k in list(state_dict):
            bare = k[len(prefix):]
            if bare.startswith('module.'):                                   # resnet3d.py:431-436 (DataParallel checkpoints)
                state_dict[prefix + bare[7:]] = state_dict.pop(k)
                k, bare = prefix + bare[7:], bare[7:]
            m = TwoStageDetector3D._DCN_REF_KEY.match(bare)
            if m is not None and (m.group(2) == 'conv_offset' or self.neck_dcn):
                state_dict[f'{prefix}dcn_neck.{m.group(2)}.{m.group(1)}.{m.group(3)}'] = state_dict.pop(k)
            elif bare.startswith(('backbone.', 'neck.')):
                state_dict[prefix + 'extract.' + bare] = state_dict.pop(k)

    def load_checkpoint(self, path_or_dict, strict=False):
        """mmcv `load_checkpoint` semantics for a reference `.pth` ({'state_dict': ...} or a bare dict): keys of heads this
        single-pathway detector does not have (rpn_head_2, refinement heads, ...) are reported, not fatal."""
        sd = torch.load(path_or_dict, map_location='cpu') if isinstance(path_or_dict, str) else path_or_dict
        sd = sd.get('state_dict', sd)
        return self.load_state_dict(dict(sd), strict=strict)

    def _mark(self, name):
        """optional phase timing (tools/two_stage_bench.py --phases): records a device event per phase boundary"""
        t = getattr(self, '_phase_events', None)
        if t is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            t.append((name, e))

    def extract_feat(self, imgs, slot=0, train=True):
        """backbone + neck (two_stage_3d.py:108-112); with neck_dcn the engine hands out the lateral sums and the deformable
        output stage runs here"""
        x = self.extract(imgs, slot) if train else self.extract.forward_features(imgs, slot)
        if not self.neck_dcn:
            return x
        outs = self.dcn_neck(x)
        if self.extract.bf16:            # bf16 mode: the heads read a bf16 pyramid, as without the deformable neck
            outs = tuple(o.to(torch.bfloat16) for o in outs)
        return outs

    def forward_backward(self, *args, **kwargs):
        """forward_train + the backward pass of the summed losses (base.py `_parse_losses` + `loss.backward()`), issued in
        the order that keeps the device fed.  The step has ONE host wait -- the positive count that sizes the mask head's
        batch (PosCountReadback) -- and everything enqueued behind it meets an idle device, so its launch cost is step
        time.  `total.backward()` after forward_train puts the whole head backward (≈ 200 small launches) there.  Here
        the RPN and bbox-head losses are differentiated down to the pyramid BEFORE the wait (their launches hide under the
        backbone forward the device is still running), and only the mask branch and the backbone backward follow it.
        Same gradients: every loss enters the sum with coefficient 1, and a sum's gradient is the sum of the gradients.
        Returns the loss dict with detached values; parameter gradients are accumulated as by backward()."""
        with deferred_backward() as items:
            return self.forward_train(*args, _eager_backward=items, **kwargs)

    def forward_train(self, imgs, img_meta, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, proposals=None,
                      draws=None, _eager_backward=False):
        """Returns the reference's loss dict (loss_rpn_cls / loss_rpn_reg lists per level, loss_cls, acc, loss_reg,
        loss_mask).  draws: optional dict(rpn=[(pos, neg) per image], rcnn=[(pos, neg) per image]) of sampler words."""
        assert imgs.shape[1] == 3                                           # two_stage_3d.py:124
        if _eager_backward is True:      # (a direct call: the recording block belongs to forward_backward)
            return self.forward_backward(imgs, img_meta, gt_bboxes, gt_labels, gt_bboxes_ignore=gt_bboxes_ignore, gt_masks=gt_masks,
                                         proposals=proposals, draws=draws)
        deferred, _eager_backward = _eager_backward, _eager_backward is not False
        draws = draws or {}
        self._mark('start')
        x = self.extract_feat(imgs)
        losses = dict()
        cfg = self.train_cfg
        if _eager_backward:
            # The RPN branch -- head, targets, losses and their whole backward pass (two 3x3x3 convolutions' worth of data and
            # weight gradients on P2: ~2 ms of dense work) -- runs on a second stream next to the proposal / roi-head chain,
            # which is ~300 small launches with an idle device between them.  Each branch differentiates down to its own
            # detached copy of the pyramid; the two gradients are added in front of the backbone backward.
            x_graph = x
            # the roi extractors' backward passes are recorded, not run: they are scattered into the RPN branch's gradient buffers
            # at the end (roi_extractor.deferred_backward / pyramid_gradient) -- no per-extractor zero-filled level buffers, no adds
            x = tuple(t.detach().requires_grad_(True) for t in x_graph)
            deferred.keys.add(x[0].data_ptr())                   # only THIS pyramid's extractor backward passes are recorded
            x_rpn = tuple(t.detach().requires_grad_(True) for t in x_graph)
            self._mark('backbone+fpn fwd')
            main = torch.cuda.current_stream()
            if getattr(self, '_rpn_stream', None) is None:
                self._rpn_stream = torch.cuda.Stream(device=self.device)
            side = self._rpn_stream
            # the inside flags the proposal stage filters with (anchor_head_3d.py:212) depend on shapes only
            self.rpn_head.pos_indices = self.rpn_head.anchor_geometry([tuple(t.shape[-3:]) for t in x], img_meta, cfg['rpn'],
                                                                      x[0].device)[3]
            side.wait_stream(main)
            # (round 4, measured and not kept: enqueueing the branch's loss + backward from a helper thread.  Under rocprofv3 the
            # middle of the step looks launch-bound -- 3.7 ms of device idle in an 18 ms traced step -- but without the tracer the
            # device is busy: 13.996 against 14.08 ms per step.)
            with torch.cuda.stream(side):
                rpn_outs = self.rpn_head(x_rpn)
                head_done = torch.cuda.Event()
                head_done.record(side)
                rpn_losses = self.rpn_head.loss(*rpn_outs, gt_bboxes, img_meta, cfg['rpn'], gt_bboxes_ignore=gt_bboxes_ignore,
                                                draws_list=draws.get('rpn'))
                torch.autograd.backward(self.total_loss(rpn_losses))
            losses.update(rpn_losses)
            main.wait_event(head_done)
            for t in list(rpn_outs[0]) + list(rpn_outs[1]):
                t.record_stream(main)
            self._mark('rpn head fwd (second stream: + loss + backward)')
        else:
            self._mark('backbone+fpn fwd')
            rpn_outs = self.rpn_head(x)
            self._mark('rpn head fwd')
            losses.update(self.rpn_head.loss(*rpn_outs, gt_bboxes, img_meta, cfg['rpn'], gt_bboxes_ignore=gt_bboxes_ignore,
                                             draws_list=draws.get('rpn')))
            self._mark('rpn targets+loss')
        with torch.no_grad():
            # fixed-capacity proposal blocks + device-side row counts: no host round trip (detector/rpn_proposals.py)
            if proposals is None:
                proposal_list, proposal_counts = self.rpn_head.get_bboxes(*rpn_outs, img_meta, SimpleNamespace(**cfg['rpn_proposal']),
                                                                          use_pos_indices=True, with_counts=True)
            else:
                proposal_list, proposal_counts = proposals, [None] * len(proposals)
        self.last_proposals = (proposal_list, proposal_counts)
        self._mark('proposals (decode+nms3d)')

        rc = cfg['rcnn']
        assigner, sampler = build_assigner(rc['assigner']), build_sampler(rc['sampler'])
        assert sampler.add_gt_as_proposals
        cap, cap_pos = sampler.num, int(sampler.num * sampler.pos_fraction)
        num_imgs = imgs.size(0)
        rois, labels, label_weights, bbox_targets, bbox_weights, pos_gt, counts = [], [], [], [], [], [], []
        for i in range(num_imgs):
            gt, gl = gt_bboxes[i].float(), gt_labels[i]
            real = None if proposal_counts[i] is None else \
                torch.arange(proposal_list[i].size(0), device=proposal_list[i].device) < proposal_counts[i]
            ar = assigner.assign(proposal_list[i], gt, None, gl, inside_flags=real)      # rows behind the count take no part
            assigned_all = _hip.concat_gt_assign(ar.gt_inds, gt.size(0))
            dp, dn = draws['rcnn'][i] if 'rcnn' in draws else (None, None)
            pos, neg, cnt, _ = _hip.random_sample(assigned_all, cap, sampler.pos_fraction, sampler.neg_pos_ub, dp, dn)
            r, lab, lw, bt, bw, pgi = _hip.bbox_target3d(proposal_list[i], gt, gl, assigned_all, pos, neg, cnt, cap, i,
                                                         self.bbox_head.target_means, self.bbox_head.target_stds,
                                                         float(rc['pos_weight']))
            rois.append(r); labels.append(lab); label_weights.append(lw); bbox_targets.append(bt); bbox_weights.append(bw)
            pos_gt.append(pgi); counts.append(cnt)
        counts = torch.stack(counts)                                        # [num_imgs, 4]: pos, neg, candidates
        n_pos_total = counts[:, 0].sum().reshape(1).to(counts.dtype)
        readback = PosCountReadback(n_pos_total, rc.get('mask_batch_buckets', MASK_BATCH_BUCKETS), cap_pos * num_imgs) \
            if self.with_mask else None
        self.last_sample_counts = counts
        self._mark('rcnn assign+sample+targets')

        # ---- bbox head (two_stage_3d.py:179-192).  Several images: their real rows are packed to the front of the
        # concatenated batch (bbox2roi3D's layout minus the per-image padding) so that one valid prefix remains.
        cat = lambda ts: torch.cat(ts)
        all_rois, lab, lw, bt, bw = cat(rois), cat(labels), cat(label_weights), cat(bbox_targets), cat(bbox_weights)
        if num_imgs > 1:
            order = front_pack_order([counts[i, 0] + counts[i, 1] for i in range(num_imgs)], cap)
            all_rois, lab, lw, bt, bw = all_rois[order], lab[order], lw[order], bt[order], bw[order]
        bbox_feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], all_rois)
        cls_score, bbox_pred = self.bbox_head(bbox_feats)
        # padding rows have weight 0 everywhere; accuracy / avg_factor need the real row count
        num_rois = _hip.avg_factor_from_counts(counts, 1)
        valid_rows = (counts[:, 0] + counts[:, 1]).sum().reshape(1).to(torch.int32)
        losses.update(self.bbox_head.loss(cls_score, bbox_pred, lab, lw, bt, bw, valid_rows=valid_rows, num_rois=num_rois))
        self._mark('bbox roialign+head+loss')
        if _eager_backward:
            torch.autograd.backward(losses['loss_cls'].sum() + losses['loss_reg'].sum())
            self._mark('bbox head backward')

        # ---- mask head on the positives (two_stage_3d.py:194-218): rows [0, cap_pos) of every image's roi batch, the
        # first #pos of them real; targets are cropped per image (each has its own gt masks), then packed like the rois
        if self.with_mask:
            rows = readback.rows()
            if num_imgs == 1:                  # only the bucket's rows are ever touched
                pos_rois, pos_labels = rois[0][:rows].contiguous(), labels[0][:rows]
                mask_targets = mask_target_single(pos_rois[:, 1:], pos_gt[0][:rows].clamp(min=0), gt_masks[0], SimpleNamespace(**rc))
            else:
                pos_rois = cat([r[:cap_pos] for r in rois])
                pos_labels = cat([l[:cap_pos] for l in labels])
                mask_targets = cat([mask_target_single(rois[i][:cap_pos, 1:].contiguous(), pos_gt[i][:cap_pos].clamp(min=0),
                                                       gt_masks[i], SimpleNamespace(**rc)) for i in range(num_imgs)])
                order = front_pack_order([counts[i, 0] for i in range(num_imgs)], cap_pos)[:rows]
                pos_rois, pos_labels, mask_targets = pos_rois[order].contiguous(), pos_labels[order], mask_targets[order]
            mask_feats = self.mask_roi_extractor(x[:self.mask_roi_extractor.num_inputs], pos_rois)
            mask_pred = self.mask_head(mask_feats)
            losses.update(self.mask_head.loss(mask_pred, mask_targets, pos_labels, valid_rows=n_pos_total))
            self._mark('mask roialign+head+targets+loss')
        if _eager_backward:
            if self.with_mask:
                torch.autograd.backward(losses['loss_mask'].sum())
            main.wait_stream(side)
            assert_all_flushed(deferred, (x,))
            torch.autograd.backward(x_graph, pyramid_gradient(x, x_rpn, deferred))
            deferred.clear()                     # the recorded top gradients are not kept alive until the next step
            losses = {k: ([t.detach() for t in v] if isinstance(v, (list, tuple)) else v.detach()) for k, v in losses.items()}
            self._mark('mask branch + backbone backward')
        self.iteration += 1
        return losses

    @torch.no_grad()
    def simple_test(self, imgs, img_metas, proposals=None, rescale=False):
        """two_stage_3d.py:229-264 (+ test_mixins.py:7-11,33-56,94-119): proposals -> bbox head -> per-class decode +
        NMS3D -> (bbox_results, segm_results); one volume per call (base.py:74-75)."""
        assert imgs.size(0) == 1
        cfg = self.test_cfg
        x = self.extract_feat(imgs, train=False)
        if proposals is None:
            proposal_list, counts = self.rpn_head.get_bboxes(*self.rpn_head(x), img_metas, SimpleNamespace(**cfg['rpn']), with_counts=True)
        else:
            proposal_list, counts = proposals, [None]
        rois = bbox2roi3D(proposal_list)
        roi_feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred = self.bbox_head(roi_feats)
        meta = img_metas[0]
        det_bboxes, det_labels = self.bbox_head.get_det_bboxes(rois, cls_score, bbox_pred, meta['img_shape'],
                                                               meta.get('scale_factor', 1.0), rescale=rescale, cfg=cfg['rcnn'],
                                                               valid_rows=counts[0])
        bbox_results = bbox2result3D(det_bboxes, det_labels, self.bbox_head.num_classes)
        if not self.with_mask:
            return bbox_results
        if det_bboxes.shape[0] == 0:
            return bbox_results, [[] for _ in range(self.mask_head.num_classes - 1)]
        scale = meta.get('scale_factor', 1.0)
        _bboxes = det_bboxes[:, :6] * scale if rescale else det_bboxes[:, :6]
        mask_rois = bbox2roi3D([_bboxes])
        mask_feats = self.mask_roi_extractor(x[:self.mask_roi_extractor.num_inputs], mask_rois)
        mask_pred = self.mask_head(mask_feats)
        h, w, _, d = meta['img_shape']
        ori_shape = meta.get('ori_shape_hwd', (h, w, d))
        segm_results = self.mask_head.get_seg_masks(mask_pred, _bboxes, det_labels, cfg['rcnn'], ori_shape, scale, True)
        return bbox_results, segm_results

    @staticmethod
    def total_loss(losses):
        """mmdet/models/detectors/base.py `_parse_losses`: sum of every entry whose key contains 'loss'"""
        terms = []
        for k, v in losses.items():
            if 'loss' in k:
                terms += [t.reshape(-1) for t in (v if isinstance(v, (list, tuple)) else [v])]
        return torch.cat(terms).sum()                 # one gather + one reduction instead of a sum and an add per entry

"""TwoStageDetector3D2Scales.forward_train -- the SHIPPED model (mmdet/models/detectors/two_stage_3d_2scales.py:135-330,
configs/3d-multi-resolution-rcnn.py): two pathways through ONE backbone+FPN (the region and its 1.5x up-sampled copy),
an RPN head per pathway (anchor scales 2 / 3), a shared bbox head over both roi batches, a regression-only
refinement head fed by pathway-2 predictions mapped back to pathway-1 coordinates, and two mask heads.

Same construction as detector/two_stage.py: static shapes, device-resident counts, no host synchronisation between
the backbone and the loss dict other than SingleRoIExtractor's level histograms.  In particular the reference's
`.cpu().numpy()` / `torch.from_numpy().cuda()` round trip of the refinement proposals (:263-267) stays on the
device, and padding rows of the pathway-2 roi batch are masked out of the refinement assigner."""
from types import SimpleNamespace

import torch

from .. import _hip
from ..core.bbox import build_assigner, build_sampler, delta2bbox3D, bbox2roi3D
from ..core.post_processing import multiclass_nms_3d
from ..core.mask import mask_target_single
from .heads import RPNHead3D, SharedFCBBoxHead3D, FCNMaskHead3D, rpn_get_bboxes_pair
from .roi_extractor import SingleRoIExtractor, assert_all_flushed, deferred_backward, pyramid_gradient
from .two_stage import TwoStageDetector3D, bbox2result3D, PosCountReadback, MASK_BATCH_BUCKETS


class TwoStageDetector3D2Scales(TwoStageDetector3D):

    def __init__(self, device='cuda', train_cfg=None, test_cfg=None, dtype=torch.float32, neck_dcn=False):
        super().__init__(device=device, train_cfg=train_cfg, test_cfg=test_cfg, with_mask=True, dtype=dtype, neck_dcn=neck_dcn)
        strides, dstrides = [4, 8, 16, 32], [2, 4, 8, 16]
        self.rpn_head_2 = RPNHead3D(anchor_scales=(3,), anchor_depth_scales=(3,)).to(self.device)      # configs/...py:28-39
        self.bbox_roi_extractor_refinement = SingleRoIExtractor(
            dict(type='RoIAlign3D', out_size=7, out_size_depth=3, sample_num=2), 64, strides, dstrides)
        self.refinement_head = SharedFCBBoxHead3D(with_cls=False).to(self.device)                      # configs/...py:57-67
        self.refinement_mask_roi_extractor = SingleRoIExtractor(
            dict(type='RoIAlign3D', out_size=14, out_size_depth=10, sample_num=2), 64, strides, dstrides)
        self.refinement_mask_head = FCNMaskHead3D().to(self.device)                                    # configs/...py:80-85
        self.refinement_mask_head.bf16 = dtype == torch.bfloat16

    # ------------------------------------------------------------------ shared pieces
    def _sample(self, proposals, gt, gl, draws, batch_index, inside=None, want_mask=False):
        """assign + add gts + sample + targets for one image -> dict of fixed-capacity tensors (two_stage_3d_2scales.py
        :182-226 / :268-288)"""
        rc = self.train_cfg['rcnn']
        assigner, sampler = build_assigner(rc['assigner']), build_sampler(rc['sampler'])
        cap = sampler.num
        gt = gt.float()
        ar = assigner.assign(proposals, gt, None, gl, inside_flags=inside)
        assigned_all = _hip.concat_gt_assign(ar.gt_inds, gt.size(0))
        dp, dn = draws if draws is not None else (None, None)
        pos, neg, cnt, _ = _hip.random_sample(assigned_all, cap, sampler.pos_fraction, sampler.neg_pos_ub, dp, dn)
        rois, lab, lw, bt, bw, pgi = _hip.bbox_target3d(proposals, gt, gl, assigned_all, pos, neg, cnt, cap, batch_index,
                                                        self.bbox_head.target_means, self.bbox_head.target_stds,
                                                        float(rc['pos_weight']))
        cap_pos = int(cap * sampler.pos_fraction)
        readback = PosCountReadback(cnt, rc.get('mask_batch_buckets', MASK_BATCH_BUCKETS) if want_mask else None, cap_pos)
        return dict(rois=rois, labels=lab, label_weights=lw, bbox_targets=bt, bbox_weights=bw, pos_gt=pgi, counts=cnt,
                    cap_pos=cap_pos, readback=readback)

    def _mask_loss(self, extractor, head, x, s, gt_masks, key):
        rc = self.train_cfg['rcnn']
        rows = s['readback'].rows()                                            # bucketed row capacity (two_stage.PosCountReadback)
        pos_rois = s['rois'][:rows].contiguous()
        mask_pred = head(extractor(x[:extractor.num_inputs], pos_rois))
        gi = s['pos_gt'][:rows].clamp(min=0)
        targets = mask_target_single(pos_rois[:, 1:], gi, gt_masks, SimpleNamespace(**rc))
        loss = head.loss(mask_pred, targets, s['labels'][:rows], valid_rows=s['counts'][0].reshape(1))
        return {key: loss['loss_mask']}

    # ------------------------------------------------------------------ forward_train
    def forward_backward(self, *args, **kwargs):
        """forward_train + backward of the summed losses with both RPN branches (heads, targets, losses, backward) on a second
        stream beside the roi-head chain, and the bbox / refinement heads differentiated in front of the mask heads' host
        waits -- see TwoStageDetector3D.forward_backward.  Returns the loss dict with detached values."""
        with deferred_backward() as items:
            return self.forward_train(*args, _eager_backward=items, **kwargs)

    def forward_train(self, imgs, img_meta, imgs_2, img_meta_2, gt_bboxes, gt_bboxes_2, gt_labels, gt_labels_2,
                      gt_bboxes_ignore=None, gt_masks=None, gt_masks_2=None, proposals=None, draws=None, _eager_backward=False):
        assert imgs.shape[1] == 3 and imgs_2.shape[1] == 3                     # :148
        if _eager_backward is True:      # (a direct call: the recording block belongs to forward_backward)
            return self.forward_backward(imgs, img_meta, imgs_2, img_meta_2, gt_bboxes, gt_bboxes_2, gt_labels, gt_labels_2,
                                         gt_bboxes_ignore=gt_bboxes_ignore, gt_masks=gt_masks, gt_masks_2=gt_masks_2, proposals=proposals,
                                         draws=draws)
        deferred, _eager_backward = _eager_backward, _eager_backward is not False
        # One volume per GPU, like the reference's own dual-pathway model: its refinement stage wraps ALL pathway-2 predictions
        # in a one-element list (`pred_bboxes_2 = [torch.from_numpy(pred_bboxes_2).cuda()]`, two_stage_3d_2scales.py:263-267)
        # and then indexes it per image (:275-283), which raises for a second image; the shipped config sets imgs_per_gpu = 1
        # (configs/...py:191).  The single-pathway detector (two_stage.py) is batch-agnostic.
        assert imgs.size(0) == 1, 'TwoStageDetector3D2Scales trains one volume per GPU (reference: two_stage_3d_2scales.py:263-283)'
        draws = draws or {}
        cfg = self.train_cfg
        x = self.extract_feat(imgs, 0)
        x_2 = self.extract_feat(imgs_2, 1)
        losses = dict()

        # ---- RPN forward, losses, proposals (:155-176); the second head's loss keys carry the _2 suffix (rpn_head_3d.py:63-65)
        if _eager_backward:
            graph = tuple(x) + tuple(x_2)
            leaf = lambda ts: tuple(t.detach().requires_grad_(True) for t in ts)
            x, x_2 = leaf(x), leaf(x_2)                                        # roi heads
            deferred.keys.update((x[0].data_ptr(), x_2[0].data_ptr()))          # only these pyramids' extractor backward passes are recorded
            xr, xr_2 = leaf(x), leaf(x_2)                                      # RPN branches
            main = torch.cuda.current_stream()
            if getattr(self, '_rpn_stream', None) is None:
                self._rpn_stream = torch.cuda.Stream(device=imgs.device)
            side = self._rpn_stream
            sizes = lambda ts: [tuple(t.shape[-3:]) for t in ts]
            self.rpn_head.pos_indices = self.rpn_head.anchor_geometry(sizes(x), img_meta, cfg['rpn'], x[0].device)[3]
            self.rpn_head_2.pos_indices = self.rpn_head_2.anchor_geometry(sizes(x_2), img_meta_2, cfg['rpn'], x_2[0].device)[3]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                rpn_outs, rpn_outs_2 = self.rpn_head(xr), self.rpn_head_2(xr_2)
                heads_done = torch.cuda.Event()
                heads_done.record(side)
                l1 = self.rpn_head.loss(*rpn_outs, gt_bboxes, img_meta, cfg['rpn'], draws_list=draws.get('rpn'))
                l2 = self.rpn_head_2.loss(*rpn_outs_2, gt_bboxes_2, img_meta_2, cfg['rpn'], draws_list=draws.get('rpn_2'))
                torch.autograd.backward(self.total_loss(l1) + self.total_loss(l2))
            main.wait_event(heads_done)
            for t in list(rpn_outs[0]) + list(rpn_outs[1]) + list(rpn_outs_2[0]) + list(rpn_outs_2[1]):
                t.record_stream(main)
        else:
            rpn_outs, rpn_outs_2 = self.rpn_head(x), self.rpn_head_2(x_2)
            l1 = self.rpn_head.loss(*rpn_outs, gt_bboxes, img_meta, cfg['rpn'], draws_list=draws.get('rpn'))
            l2 = self.rpn_head_2.loss(*rpn_outs_2, gt_bboxes_2, img_meta_2, cfg['rpn'], draws_list=draws.get('rpn_2'))
        losses.update(l1)
        losses.update(loss_rpn_cls_2=l2['loss_rpn_cls'], loss_rpn_reg_2=l2['loss_rpn_reg'])
        with torch.no_grad():
            if proposals is None:
                # fixed-capacity blocks + device-side row counts: no host round trip (detector/rpn_proposals.py)
                pcfg = SimpleNamespace(**cfg['rpn_proposal'])
                # both pathways' 10 level sets share ONE NMS3D launch sequence (detector/heads.rpn_get_bboxes_pair)
                (proposal_list, pc1), (proposal_list_2, pc2) = rpn_get_bboxes_pair(
                    self.rpn_head, rpn_outs, img_meta, self.rpn_head_2, rpn_outs_2, img_meta_2, pcfg, use_pos_indices=True)
            else:                       # (proposal lists of pathway 1, of pathway 2 [, their counts]): the `else` branch of :176-177
                proposal_list, proposal_list_2 = proposals[:2]
                pc1, pc2 = proposals[2:4] if len(proposals) >= 4 else ([None] * len(proposal_list), [None] * len(proposal_list_2))
        self.last_proposals = (proposal_list, proposal_list_2, pc1, pc2)
        real_of = lambda p, c: None if c is None else torch.arange(p.size(0), device=p.device) < c

        # ---- assign / sample both pathways (:182-226)
        s1 = self._sample(proposal_list[0], gt_bboxes[0], gt_labels[0], draws.get('rcnn', [None])[0], 0,
                          inside=real_of(proposal_list[0], pc1[0]), want_mask=True)
        s2 = self._sample(proposal_list_2[0], gt_bboxes_2[0], gt_labels_2[0], draws.get('rcnn_2', [None])[0], 0,
                          inside=real_of(proposal_list_2[0], pc2[0]))

        # ---- shared bbox head over both roi batches (:229-257)
        feats = self.bbox_roi_extractor(x[:4], s1['rois'])
        feats_2 = self.bbox_roi_extractor(x_2[:4], s2['rois'])
        cls_score, bbox_pred = self.bbox_head(feats)
        cls_score_2, bbox_pred_2 = self.bbox_head(feats_2)
        cat = lambda k: torch.cat((s1[k], s2[k]), 0)
        counts12 = torch.stack((s1['counts'], s2['counts']))
        num_rois = _hip.avg_factor_from_counts(counts12, 1)                  # bbox_targets.size(0) of the concatenated batch
        lb = self.bbox_head.loss(torch.cat((cls_score, cls_score_2), 0), torch.cat((bbox_pred, bbox_pred_2), 0),
                                 cat('labels'), cat('label_weights'), cat('bbox_targets'), cat('bbox_weights'), num_rois=num_rois)
        # accuracy over the REAL rows of both batches: per-batch accuracies weighted by their row counts
        from ..core.loss import accuracy
        n1 = (s1['counts'][0] + s1['counts'][1]).reshape(1)
        n2 = (s2['counts'][0] + s2['counts'][1]).reshape(1)
        a1 = accuracy(cls_score, s1['labels'], valid_rows=n1.to(torch.int32))
        a2 = accuracy(cls_score_2, s2['labels'], valid_rows=n2.to(torch.int32))
        lb['acc'] = (a1 * n1 + a2 * n2) / (n1 + n2).clamp(min=1)
        losses.update(lb)

        # ---- refinement head (:259-298): pathway-2 class-1 boxes, mapped back to pathway-1 coordinates, + logit as score
        meta1, meta2 = img_meta[0], img_meta_2[0]
        upscaled_factor = meta2['ori_shape'][0] / meta1['ori_shape'][0]
        with torch.no_grad():
            dec = delta2bbox3D(s2['rois'][:, 1:].contiguous(), bbox_pred_2.detach().contiguous(), self.bbox_head.target_means,
                               self.bbox_head.target_stds, meta2['img_shape'])[:, 6:] / upscaled_factor
            pred_2 = torch.cat((dec, cls_score_2.detach()[:, 1, None]), 1).contiguous()
            real_2 = torch.arange(pred_2.size(0), device=pred_2.device) < n2          # padding rows never become candidates
        sr = self._sample(pred_2, gt_bboxes[0], gt_labels[0], draws.get('refine', [None])[0], 0, inside=real_2, want_mask=True)
        feats_r = self.bbox_roi_extractor_refinement(x[:4], sr['rois'])
        _, refined_pred = self.refinement_head(feats_r)
        lr = self.refinement_head.loss(None, refined_pred, sr['labels'], sr['label_weights'], sr['bbox_targets'],
                                       sr['bbox_weights'], num_rois=_hip.avg_factor_from_counts(sr['counts'].reshape(1, -1), 1))
        losses['loss_refinement_reg'] = lr['loss_reg']                                 # bbox_head_3d_refinement.py:113-118
        if _eager_backward:
            torch.autograd.backward(losses['loss_cls'].sum() + losses['loss_reg'].sum() + losses['loss_refinement_reg'].sum())

        # ---- mask heads on the positives of pathway 1 and of the refinement samples (:301-327)
        losses.update(self._mask_loss(self.mask_roi_extractor, self.mask_head, x, s1, gt_masks[0], 'loss_mask'))
        losses.update(self._mask_loss(self.refinement_mask_roi_extractor, self.refinement_mask_head, x, sr, gt_masks[0],
                                      'loss_mask_refinement'))
        self.last_sample_counts = torch.stack((s1['counts'], s2['counts'], sr['counts']))
        if _eager_backward:
            torch.autograd.backward(losses['loss_mask'].sum() + losses['loss_mask_refinement'].sum())
            main.wait_stream(side)
            # the four extractors on pathway 1 and the one on pathway 2 were recorded (roi_extractor.deferred_backward): their
            # scatters go straight into the RPN branches' gradient buffers -- no zero-filled level buffers, no adds
            items = deferred
            assert_all_flushed(items, (x, x_2))
            torch.autograd.backward(graph, pyramid_gradient(x, xr, items) + pyramid_gradient(x_2, xr_2, items))
            items.clear()                     # the recorded top gradients are not kept alive until the next step
            losses = {k: ([t.detach() for t in v] if isinstance(v, (list, tuple)) else v.detach()) for k, v in losses.items()}
        self.iteration += 1
        return losses

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def simple_test(self, imgs, img_metas, imgs_2, img_metas_2, proposals=None, rescale=False):
        """two_stage_3d_2scales.py:335-383 with the shipped `return_bbox_only=True` (configs/...py:144): both pathways'
        per-class boxes (pathway 2 through the refinement head on pathway-1 features), ONE multiclass NMS3D over the
        union.  rescale=True divides pathway-2 boxes by img_metas_2[0]['scale_factor'] (1.5), which brings them into
        pathway-1 coordinates before the refinement RoIAlign, exactly as the reference's get_det_bboxes does."""
        assert imgs.size(0) == 1
        cfg = self.test_cfg
        meta, meta_2 = img_metas[0], img_metas_2[0]
        x = self.extract_feat(imgs, 0, train=False)
        x_2 = self.extract_feat(imgs_2, 1, train=False)
        rcfg = SimpleNamespace(**cfg['rpn'])
        if proposals is None:
            (proposal_list, pc1), (proposal_list_2, pc2) = rpn_get_bboxes_pair(
                self.rpn_head, self.rpn_head(x), img_metas, self.rpn_head_2, self.rpn_head_2(x_2), img_metas_2, rcfg)
        else:
            proposal_list = proposal_list_2 = proposals
            pc1 = pc2 = [None]

        def bboxes_scores(feats, props, m, count):                           # test_mixins_3d.py simple_test_bboxes, cfg=None
            rois = bbox2roi3D(props)
            cls_score, bbox_pred = self.bbox_head(self.bbox_roi_extractor(feats[:4], rois))
            return self.bbox_head.get_det_bboxes(rois, cls_score, bbox_pred, m['img_shape'], m.get('scale_factor', 1.0),
                                                 rescale=rescale, cfg=None, valid_rows=count)
        bboxes, scores = bboxes_scores(x, proposal_list, meta, pc1[0])
        bboxes_2, scores_2 = bboxes_scores(x_2, proposal_list_2, meta_2, pc2[0])
        # refinement (test_mixins_3d.py:102-128): class-1 boxes of pathway 2 + their score -> RoIAlign on PATHWAY-1 features
        ref_in = [torch.cat((bboxes_2[:, 6:], scores_2[:, 1, None]), dim=1)]
        rois_r = bbox2roi3D(ref_in)
        _, pred_r = self.refinement_head(self.bbox_roi_extractor_refinement(x[:4], rois_r))
        bboxes_2_ref = delta2bbox3D(rois_r[:, 1:].contiguous(), pred_r.contiguous(), self.refinement_head.target_means,
                                    self.refinement_head.target_stds, meta['img_shape'])
        if rescale:
            bboxes_2_ref = bboxes_2_ref / meta.get('scale_factor', 1.0)
        rc = cfg['rcnn']
        det_bboxes, det_labels = multiclass_nms_3d(torch.cat((bboxes, bboxes_2_ref), 0), torch.cat((scores, scores_2), 0),
                                                   rc['score_thr'], dict(rc['nms']), rc['max_per_img'])
        self.last_det = (det_bboxes, det_labels)
        return bbox2result3D(det_bboxes, det_labels, self.bbox_head.num_classes)

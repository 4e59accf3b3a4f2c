"""SingleRoIExtractor (mmdet/models/roi_extractors/single_level.py:11-104): RoI -> pyramid level map, per-level
RoIAlign3D, results scattered into one [R, C, PD, PH, PW] tensor.

Kept: the level rule (sqrt of the box VOLUME -- not its cube root -- against finest_scale, :73-81), one
RoIAlign3D per level built through `getattr(ops, layer_type)` with spatial_scale = 1/stride and
spatial_scale_depth = 1/depth_stride (:45-56).  MI355X path: when the pyramid is channels-last on the device the whole
extractor is ONE multi-level kernel launch (forward) / one plan + two kernels (backward) -- the level of every RoI is
computed inside the kernel, so there is no host synchronisation, no per-level boolean mask, no gather / scatter of
RoI subsets (the reference pays a sync per level for `inds.any()` and allocates with torch.cuda.FloatTensor, :93).
Other layouts keep a per-level path: rois grouped by level with one stable sort + one bincount read-back."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _hip
from .. import mmdet_ops as ops


# Deferred backward (detector/two_stage*.py forward_backward): while a `deferred_backward()` block is active the extractors'
# backward passes are RECORDED (top gradient, rois, geometry) instead of run, and `flush_deferred` later scatters all of them with
# the kernels' own fp32 atomics straight into the gradient buffers the RPN branch has already written -- one buffer per pyramid
# level for the whole step.  The autograd form allocates and zero-fills a full set of level buffers per extractor (168 MB for P2
# of a 512x512x80 volume, 570 MB on the 1.5x pathway) and adds them pairwise: 4 fills + 4 adds per step on the shipped model.
_deferred = None          # a plain global: autograd runs the backward of device ops on its own worker thread


class _DeferredItems(list):
    """what a deferred_backward() block records + the pyramids it owns: `keys` holds the data_ptr of the first level of every pyramid
    whose extractor backward passes may be deferred (the detector registers them once its leaf copies exist).  An extractor backward
    on any OTHER feature map -- a second detector, a nested call -- runs normally instead of being recorded and then dropped."""

    def __init__(self):
        super().__init__()
        self.keys = set()


class deferred_backward:
    def __enter__(self):
        global _deferred
        self.prev, self.items = _deferred, _DeferredItems()
        _deferred = self.items
        return self.items

    def __exit__(self, *exc):
        global _deferred
        _deferred = self.prev
        return False


def flush_deferred(items, sinks_by_key):
    """items: what a deferred_backward() block recorded; sinks_by_key: {data_ptr of a pyramid's first level: its per-level
    gradient buffers ([N,C,D,H,W] channels_last_3d fp32, already holding the other branches' gradient)}"""
    for key, top, rois, cfg, n in items:
        out_d, out_h, out_w, scales, scales_d, sample_num, finest, bug = cfg
        _hip.roi_align3d_backward_ml(top, rois, list(sinks_by_key[key][:n]), out_d, out_h, out_w, scales, scales_d, sample_num, finest, bug)


def assert_all_flushed(items, pyramids):
    """every recorded backward pass belongs to one of the pyramids whose gradients are about to be handed to the backbone"""
    keys = {p[0].data_ptr() for p in pyramids}
    stray = [it for it in items if it[0] not in keys]
    assert not stray, f'{len(stray)} deferred RoIAlign3D backward passes refer to feature maps outside this step\'s pyramids'


def pyramid_gradient(x, x_rpn, items):
    """Cotangents of one pyramid for the backbone backward: per level the RPN branch's gradient (x_rpn[l].grad) plus whatever
    autograd accumulated on the roi heads' copies (x[l].grad: extractors that did not defer, e.g. a bf16 pyramid), with the
    recorded extractor backward passes of THIS pyramid scattered on top, in place."""
    key = x[0].data_ptr()
    mine = [it for it in items if it[0] == key]
    G = []
    for a, b in zip(x, x_rpn):
        g = b.grad
        if g is None:
            g = a.grad
        elif a.grad is not None:
            g = g + a.grad
        if mine and (g is None or g.dtype != torch.float32 or not g.is_contiguous(memory_format=torch.channels_last_3d)):
            g = (torch.zeros_like(a, dtype=torch.float32, memory_format=torch.channels_last_3d) if g is None else
                 g.float().contiguous(memory_format=torch.channels_last_3d))
        G.append(g)
    flush_deferred(mine, {key: G})
    return G


class _MultiLevelRoIAlign3D(Function):
    """the whole extractor as one op: level choice inside the kernel (csrc/roi_align3d.hip, multi-level form)"""

    @staticmethod
    def forward(ctx, rois, cfg, *feats):
        out_d, out_h, out_w, scales, scales_d, sample_num, finest, bug = cfg
        ctx.cfg, ctx.shapes, ctx.feat_dtype = cfg, [f.shape for f in feats], feats[0].dtype
        ctx.key = feats[0].data_ptr()
        ctx.save_for_backward(rois)
        return _hip.roi_align3d_forward_ml([f.detach() for f in feats], rois, out_d, out_h, out_w, scales, scales_d, sample_num, finest)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        out_d, out_h, out_w, scales, scales_d, sample_num, finest, bug = ctx.cfg
        if _deferred is not None and ctx.feat_dtype == torch.float32 and ctx.key in _deferred.keys:
            _deferred.append((ctx.key, grad_output.contiguous(), rois, ctx.cfg, len(ctx.shapes)))
            return (None, None) + (None,) * len(ctx.shapes)
        grads = [torch.empty(sh, dtype=torch.float32, device=rois.device, memory_format=torch.channels_last_3d).zero_()
                 for sh in ctx.shapes]
        _hip.roi_align3d_backward_ml(grad_output, rois, grads, out_d, out_h, out_w, scales, scales_d, sample_num, finest, bug)
        if ctx.feat_dtype != torch.float32:       # bf16 pyramid (configs[4]): the scatter accumulates in fp32, the gradient leaves in bf16
            grads = [g.to(ctx.feat_dtype) for g in grads]
        return (None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[2:]))


class SingleRoIExtractor(nn.Module):

    def __init__(self, roi_layer, out_channels, featmap_strides, featmap_strides_depth=None, finest_scale=56):
        super().__init__()
        self.roi_layers = self.build_roi_layers(roi_layer, featmap_strides, featmap_strides_depth)
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.featmap_strides_depth = featmap_strides_depth
        self.finest_scale = finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def build_roi_layers(self, layer_cfg, featmap_strides, featmap_strides_depth=None):
        cfg = layer_cfg.copy()
        layer_type = cfg.pop('type')
        assert hasattr(ops, layer_type)
        layer_cls = getattr(ops, layer_type)
        if featmap_strides_depth is None:
            return nn.ModuleList([layer_cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        return nn.ModuleList([layer_cls(spatial_scale=1 / s, spatial_scale_depth=1 / d, **cfg)
                              for s, d in zip(featmap_strides, featmap_strides_depth)])

    def map_roi_levels(self, rois, num_levels):
        assert rois.shape[1] == 7, 'the 3D path uses [batch, x1, y1, x2, y2, z1, z2] rois'
        scale = torch.sqrt((rois[:, 3] - rois[:, 1] + 1) * (rois[:, 4] - rois[:, 2] + 1) * (rois[:, 6] - rois[:, 5] + 1))
        target_lvls = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return target_lvls.clamp(min=0, max=num_levels - 1).long()

    def forward(self, feats, rois):
        if len(feats) == 1:
            return self.roi_layers[0](feats[0], rois)
        layer0 = self.roi_layers[0]
        num_levels = len(feats)
        # fast path: channels-last pyramid on the device -> ONE multi-level launch, no host synchronisation
        if (num_levels <= 4 and layer0.sample_num > 0 and isinstance(layer0.out_size, int) and rois.is_cuda and
                all(f.is_cuda and f.dtype in (torch.float32, torch.bfloat16) and f.dtype == feats[0].dtype and f.dim() == 5 and
                    f.is_contiguous(memory_format=torch.channels_last_3d) for f in feats)):
            cfg = (layer0.out_size_depth, layer0.out_size, layer0.out_size, tuple(l.spatial_scale for l in self.roi_layers),
                   tuple(l.spatial_scale_depth for l in self.roi_layers), layer0.sample_num, float(self.finest_scale),
                   layer0.bug_compat)
            return _MultiLevelRoIAlign3D.apply(rois.float(), cfg, *feats)
        target_lvls = self.map_roi_levels(rois, num_levels)
        # group the rois by level once (stable sort + one bincount read-back = ONE host sync per call; the
        # reference's boolean-mask loop costs one per level for `inds.any()` plus one per masked gather)
        order = torch.argsort(target_lvls, stable=True)
        counts = torch.bincount(target_lvls, minlength=num_levels).tolist()
        rois_sorted = rois.index_select(0, order)
        parts, start = [], 0
        for i, n in enumerate(counts):
            if n:
                parts.append(self.roi_layers[i](feats[i], rois_sorted[start:start + n].contiguous()))
                start += n
        if not parts:
            return feats[0].new_zeros(0, self.out_channels, layer0.out_size_depth, layer0.out_size, layer0.out_size)
        out_sorted = torch.cat(parts, 0)
        return torch.empty_like(out_sorted).index_copy(0, order, out_sorted)

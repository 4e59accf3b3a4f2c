"""ResNet3D-50 + FPN3D feature extractor (`extract_feat`) on the MI355X conv engine.

Mirrors the reference's `TwoStageDetector3D2Scales.extract_feat` (mmdet/models/detectors/
two_stage_3d_2scales.py:123-127) = `ResNet3D.forward` (mmdet/models/backbones/resnet3d.py:467-478,
Bottleneck :198-233, stem :397-409, make_res_layer :238-285) followed by `FPN3D.forward`
(mmdet/models/necks/fpn3d.py:95-134), with the reference's parameter names
(`backbone.conv1.weight`, `backbone.layer1.0.bn3d1.running_var`, `neck.lateral_convs.0.conv.bias`, ...)
so that a reference checkpoint's state_dict loads unchanged.

MI355X design (not a module-by-module translation):
  * every tensor is NDHWC fp32; the volume comes in as the reference's [N,3,D,H,W] and the pyramid goes
    out as [N,64,D,H,W] tensors with channels_last_3d strides (zero-copy views);
  * all parameters and BN buffers live in ONE flat arena, all gradients in a mirror arena (the buffer the
    data-parallel all-reduce runs on); BatchNorm3d runs in eval mode during training (norm_eval=True,
    resnet3d.py:330,480-486) so it is folded into the conv weights + a per-channel shift by one
    table-driven kernel per step; gamma/beta gradients are recovered from the wgrad results;
  * forward and backward are pre-built *plans* (lists of kernel records) replayed by one native call each;
    conv + BN + ReLU (+ residual add / + nearest-upsample add) is one MFMA implicit-GEMM kernel, and every
    data-gradient kernel applies the ReLU mask of the layer it feeds in its epilogue.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _hip

_ALIGN = 64     # floats; keeps every sub-buffer 256-byte aligned


class _Holder(nn.Module):
    """empty container used to reproduce the reference's module tree (for state_dict key names)"""


def _attach(root, dotted, tensor, is_param):
    parts = dotted.split('.')
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Holder())
        m = getattr(m, p)
    if is_param:
        m.register_parameter(parts[-1], tensor)
    else:
        m.register_buffer(parts[-1], tensor)


class _ConvSpec:
    def __init__(self, name, cin, cout, k, stride, pad, bn=None, bias=False, need_dgrad=True):
        self.name, self.cin, self.cout, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad
        self.bn, self.bias, self.need_dgrad = bn, bias, need_dgrad
        self.cin_pad = 4 if cin == 3 else cin
        self.taps = k ** 3
        self.index = -1


class ExtractFeat3D(nn.Module):
    STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
    MARK_GRADS_BUT_STEM = 1          # SYNC slot of the backward plan (see dist_utils.allreduce_overlapped)
    MARK_GRADS_EARLY = 3             # SYNC slot: the gradients of layer3, layer4 and the neck are final (dist_utils.allreduce_overlapped)
    EARLY_STAGE = 2                  # layer3: the first stage whose parameters belong to the early bucket
    TUNE_LATE_GRAD_FINISH, TUNE_FPN_AFTER_LATERALS = 8, 9      # mr3d_tuning_set keys (include/mr3d.h): the A/B switches of this class
    MARK_COARSE_DGRADS = 2           # SYNC slot: the P3..P5 output-conv data gradients (side stream) are done
    MARK_LATERAL_DGRAD = 4           # SYNC slots 4 / 5: the P2 / P3 lateral data gradients (side stream, fp32 plans) are done

    def __init__(self, depth=50, fpn_out_channels=64, num_outs=5, zero_init_residual=True, device='cuda',
                 dtype=torch.float32, neck_dcn=False):
        """neck_dcn: BASELINE configs[4]'s "DeformConv3d neck" -- the four FPN output convs (fpn3d.py:56-64) are NOT part of
        the engine: it hands out the top-down lateral sums (4 maps) and takes their gradients; detector/dcn_neck.DCNNeck3D
        applies offset conv + DeformConv3d on them (build-defined: the reference never runs DCN on 5-D input).
        dtype: storage type of activations and activation gradients -- torch.float32 (the reference's arithmetic, the
        headline path) or torch.bfloat16 (BASELINE configs[4]: bf16 maps in HBM, v_mfma_f32_32x32x16_bf16 with fp32
        accumulation; parameters, their gradients, the folded BN shifts and the stem's arithmetic stay fp32).  The pyramid
        is handed out in fp32 either way."""
        super().__init__()
        assert dtype in (torch.float32, torch.bfloat16)
        self.act_dtype = dtype
        self.bf16 = dtype == torch.bfloat16
        # dtype of the pyramid handed out: fp32 (any consumer), or bf16 for consumers that read bf16 maps directly (the RPN
        # head's bf16 plan, RoIAlign3D's bf16-feature kernel): no fp32 copy of the five maps, bf16 gradients come straight back
        self.pyramid_dtype = torch.float32
        # set by the detectors BEFORE the first forward pass: the gradients arriving at the pyramid are zero on most position
        # tiles (256 sampled anchors + the sampled rois' footprints), and the FPN output convs' weight gradients test for that
        self.sparse_pyramid_grads = False
        self.neck_dcn = bool(neck_dcn)
        if depth not in self.STAGE_BLOCKS:
            raise KeyError(f'invalid depth {depth} for the bottleneck ResNet3D engine')
        assert num_outs == 5, 'the shipped config uses 4 pyramid levels + 1 max-pooled level (configs/...py:11-15)'
        self.depth = depth
        self.device = torch.device(device)
        # the early gradient bucket (round 5): off = ONE finishing at the end of the backward plan, rounds 1-4 (an A/B knob)
        self.early_bucket = _hip.lib().mr3d_tuning_get(self.TUNE_LATE_GRAD_FINISH) == 0
        self.fpn_out = fpn_out_channels
        self.zero_init_residual = zero_init_residual
        self.eps = 1e-5                                    # norm.py:50 cfg_.setdefault('eps', 1e-5)
        self._build_specs()
        self._build_arena()
        self._plans = {}
        # forward() through autograd hands out COPIES of the plan's pyramid buffers (the next forward pass of the same shape and
        # slot overwrites them).  A caller that consumes the pyramid before it runs the engine again -- the detectors' training
        # step -- sets this to False and saves the 224 MB copy (0.09 ms on a 512x512x80 volume, 0.3 ms on the 1.5x pathway).
        self.clone_outputs = True
        self.grads_touched = False
        self.init_weights()

    # ------------------------------------------------------------------ structure
    def _build_specs(self):
        S = []
        add = S.append
        add(_ConvSpec('backbone.conv1', 3, 16, 7, (1, 2, 2), 3, bn='backbone.bn3d1', need_dgrad=False))
        self.blocks = []          # (stage, block, stride, has_downsample, inplanes, planes)
        inplanes = 16
        for si, nb in enumerate(self.STAGE_BLOCKS[self.depth]):
            planes = 16 * 2 ** si
            for bi in range(nb):
                stride = 2 if (bi == 0 and si > 0) else 1
                pre = f'backbone.layer{si + 1}.{bi}'
                ds = bi == 0 and (stride != 1 or inplanes != planes * 4)
                blk = dict(stage=si, block=bi, stride=stride, ds=ds, inplanes=inplanes, planes=planes, pre=pre)
                blk['conv1'] = _ConvSpec(pre + '.conv1', inplanes, planes, 1, (1, 1, 1), 0, bn=pre + '.bn3d1')
                blk['conv2'] = _ConvSpec(pre + '.conv2', planes, planes, 3, (stride,) * 3, 1, bn=pre + '.bn3d2')
                blk['conv3'] = _ConvSpec(pre + '.conv3', planes, planes * 4, 1, (1, 1, 1), 0, bn=pre + '.bn3d3')
                for key in ('conv1', 'conv2', 'conv3'):
                    add(blk[key])
                if ds:
                    blk['dsconv'] = _ConvSpec(pre + '.downsample.0', inplanes, planes * 4, 1, (stride,) * 3, 0,
                                              bn=pre + '.downsample.1')
                    add(blk['dsconv'])
                self.blocks.append(blk)
                inplanes = planes * 4
        self.stage_out_channels = [16 * 2 ** i * 4 for i in range(4)]
        self.lateral, self.fpnconv = [], []
        for i, c in enumerate(self.stage_out_channels):
            self.lateral.append(_ConvSpec(f'neck.lateral_convs.{i}.conv', c, self.fpn_out, 1, (1, 1, 1), 0, bias=True))
            add(self.lateral[-1])
            if not self.neck_dcn:
                self.fpnconv.append(_ConvSpec(f'neck.fpn_convs.{i}.conv', self.fpn_out, self.fpn_out, 3, (1, 1, 1), 1, bias=True))
                add(self.fpnconv[-1])
        for k, s in enumerate(S):
            s.index = k
        self.specs = S

    def _arena_entries(self):
        """(name, shape) of every trainable tensor, then of every BatchNorm running statistic, in arena order"""
        train, stats = [], []
        for s in self.specs:
            train.append((s.name + '.weight', (s.cout, s.cin, s.k, s.k, s.k)))
            if s.bias:
                train.append((s.name + '.bias', (s.cout,)))
            if s.bn:
                train.append((s.bn + '.weight', (s.cout,)))
                train.append((s.bn + '.bias', (s.cout,)))
                stats.append((s.bn + '.running_mean', (s.cout,)))
                stats.append((s.bn + '.running_var', (s.cout,)))
        return train, stats

    def _build_arena(self):
        """Arena layout: [every trainable tensor | every BN running statistic].  `flat_params` / `flat_grads` are the
        TRAINABLE prefix only -- the range the optimiser steps and the data-parallel all-reduce exchanges; the running
        statistics (frozen: norm_eval=True, resnet3d.py:330,480-486) sit behind it, so weight decay / momentum never touch
        them.  The kernels address both regions through offsets from one base pointer (mr3d_layer_desc)."""
        train, stats = self._arena_entries()
        al = lambda v: -(-v // _ALIGN) * _ALIGN
        self.rel_offsets, off = {}, 0
        for name, shape in train:
            self.rel_offsets[name] = off
            off += al(int(np.prod(shape)))
        self.num_trainable = off
        soff = 0
        for name, shape in stats:
            self.rel_offsets[name] = soff
            soff += al(int(np.prod(shape)))
        self.num_stats = soff
        arena = torch.zeros(self.num_trainable + self.num_stats, dtype=torch.float32, device=self.device)
        grads = torch.zeros(self.num_trainable, dtype=torch.float32, device=self.device)
        self._bind_storage(arena, grads, 0, self.num_trainable, first=True)
        for s in self.specs:
            if s.bn:
                _attach(self, s.bn + '.num_batches_tracked', torch.zeros((), dtype=torch.long, device=self.device), False)
        # packed-weight arena and wgrad scratch arena
        poff = soff = 0
        self._layer_offsets = []
        for s in self.specs:
            nw = s.taps * s.cin_pad * s.cout
            L = dict(wf=poff); poff += al(nw)
            if s.need_dgrad:
                L['wb'] = poff; poff += al(nw)
            else:
                L['wb'] = -1
            L['shift'] = poff; poff += al(s.cout)
            L['scale'] = poff; poff += al(s.cout)
            L['dwraw'] = soff; soff += al(nw)
            L['dbias'] = soff; soff += al(s.cout)
            self._layer_offsets.append(L)
        self.register_buffer('_packed', torch.zeros(poff, dtype=torch.float32, device=self.device), persistent=False)
        self.register_buffer('_scratch', torch.zeros(soff, dtype=torch.float32, device=self.device), persistent=False)
        self.max_cout = max(s.cout for s in self.specs)
        self._build_table()
        if self.bf16:
            # bf16 operand arena (csrc/aux_ops.hip prep_weights_bf16_kernel): every conv but the stem (fp32 arithmetic)
            r16, r32 = (lambda v: -(-v // 16) * 16), (lambda v: -(-v // 32) * 32)
            off16, self._layer_offsets16 = 0, {}
            for s in self.specs:
                if s.k == 7:
                    continue
                o = dict(wf=off16); off16 += -(-s.taps * r16(s.cin_pad) * r32(s.cout) // 128) * 128
                if s.need_dgrad:
                    o['wb'] = off16; off16 += -(-s.taps * r16(s.cout) * r32(s.cin_pad) // 128) * 128
                else:
                    o['wb'] = -1
                self._layer_offsets16[s.index] = o
            self.register_buffer('_packed16', torch.zeros(off16, dtype=torch.bfloat16, device=self.device), persistent=False)
            self.register_buffer('_stem_w16', torch.zeros(49 * 4 * 16 * 8, dtype=torch.bfloat16, device=self.device), persistent=False)
            self._build_table16()

    def _bind_storage(self, arena, grads, param_off, stats_off, first=False):
        """Point every parameter / running statistic (and every parameter's .grad) at `arena` / `grads`: trainable
        tensors from element `param_off`, running statistics from `stats_off`; `grads` mirrors the arena's offsets over
        the trainable range.  Current values are carried over."""
        train, stats = self._arena_entries()
        old = None if first else {n: self._tensor_by_name(n).detach().clone() for n, _ in train + stats}
        self.offsets = {}
        for n, _ in train:
            self.offsets[n] = param_off + self.rel_offsets[n]
        for n, _ in stats:
            self.offsets[n] = stats_off + self.rel_offsets[n]
        self._param_off, self._stats_off = param_off, stats_off
        if first:
            self.register_buffer('_arena', arena, persistent=False)
            self.register_buffer('_grad_arena', grads, persistent=False)
        else:
            self._arena, self._grad_arena = arena, grads
        self._grad_views = {}
        # one persistent view object each (callers attach `.grad` to flat_params and hand it to an optimiser)
        object.__setattr__(self, '_flat_params_view', arena[param_off:param_off + self.num_trainable])
        object.__setattr__(self, '_flat_grads_view', grads[param_off:param_off + self.num_trainable])
        with torch.no_grad():
            for name, shape in train:
                o, n = self.offsets[name], int(np.prod(shape))
                view, gview = arena[o:o + n].view(shape), grads[o:o + n].view(shape)
                if first:
                    p = nn.Parameter(view)
                    _attach(self, name, p, True)
                else:
                    view.copy_(old[name])
                    p = self.get_parameter(name)
                    p.data = view
                p.grad = gview
                self._grad_views[name] = (p, gview)
            for name, shape in stats:
                o, n = self.offsets[name], int(np.prod(shape))
                view = arena[o:o + n].view(shape)
                if first:
                    _attach(self, name, view, False)
                else:
                    view.copy_(old[name])
                    holder, leaf = name.rsplit('.', 1)
                    setattr(self.get_submodule(holder), leaf, view)

    def _tensor_by_name(self, name):
        holder, leaf = name.rsplit('.', 1)
        return getattr(self.get_submodule(holder), leaf)

    def _build_table(self):
        table = (_hip.LayerDesc * len(self.specs))()
        for s in self.specs:
            L, o = table[s.index], self._layer_offsets[s.index]
            L.w_off = self.offsets[s.name + '.weight']
            L.bias_off = self.offsets[s.name + '.bias'] if s.bias else -1
            if s.bn:
                L.gamma_off, L.beta_off = self.offsets[s.bn + '.weight'], self.offsets[s.bn + '.bias']
                L.mean_off, L.var_off = self.offsets[s.bn + '.running_mean'], self.offsets[s.bn + '.running_var']
            else:
                L.gamma_off = L.beta_off = L.mean_off = L.var_off = -1
            L.wf_off, L.wb_off, L.shift_off, L.scale_off = o['wf'], o['wb'], o['shift'], o['scale']
            L.dwraw_off, L.dbias_off = o['dwraw'], o['dbias']
            L.co, L.ci, L.ci_pad, L.taps, L.eps = s.cout, s.cin, s.cin_pad, s.taps, self.eps
        self._table_host = table
        raw = torch.from_numpy(np.frombuffer(bytes(table), dtype=np.uint8).copy()).to(self.device)
        if hasattr(self, '_table'):
            self._table.copy_(raw)            # same address: plans built earlier stay valid as far as the table goes
        else:
            self.register_buffer('_table', raw, persistent=False)

    def _build_table16(self):
        """layer table of mr3d_prepare_weights_bf16: parameter offsets as in the fp32 table, wf_off / wb_off = offsets in
        bf16 elements into the bf16 operand arena"""
        idx = sorted(self._layer_offsets16)
        table = (_hip.LayerDesc * len(idx))()
        for k, li in enumerate(idx):
            L, src, o = table[k], self._table_host[li], self._layer_offsets16[li]
            ctypes.memmove(ctypes.addressof(L), ctypes.addressof(src), ctypes.sizeof(_hip.LayerDesc))
            L.wf_off, L.wb_off, L.shift_off, L.scale_off = o['wf'], o['wb'], -1, -1
        self._table16_len = len(idx)
        raw = torch.from_numpy(np.frombuffer(bytes(table), dtype=np.uint8).copy()).to(self.device)
        if hasattr(self, '_table16'):
            self._table16.copy_(raw)
        else:
            self.register_buffer('_table16', raw, persistent=False)

    def rehome(self, arena, grads, param_off, stats_off):
        """Move the engine's parameters into a caller-owned flat arena (and gradient mirror) so that the WHOLE detector
        -- backbone, neck and every head -- is one buffer and one all-reduce (mmdet/core/utils/dist_utils.py:22-28;
        dist_utils.DetectorArena).  `arena` needs num_trainable floats at param_off and num_stats floats at stats_off
        (both multiples of 64 floats); `grads` mirrors the trainable range.  Plans built before are dropped."""
        assert arena.dtype == torch.float32 and grads.dtype == torch.float32 and arena.device == self._arena.device
        assert param_off % _ALIGN == 0 and stats_off % _ALIGN == 0 and arena.data_ptr() % 256 == 0 and grads.data_ptr() % 256 == 0
        assert arena.numel() >= max(param_off + self.num_trainable, stats_off + self.num_stats)
        assert grads.numel() >= param_off + self.num_trainable
        self._bind_storage(arena, grads, param_off, stats_off)
        self._build_table()
        if self.bf16:
            self._build_table16()
        self._plans = {}

    # ------------------------------------------------------------------ parameters
    def init_weights(self, pretrained=None):
        """resnet3d.py:440-465 (kaiming-normal fan_out convs, BN = 1/0, zero-init of every block's bn3d3) and
        fpn3d.py:90-93 (xavier-uniform convs, zero bias)."""
        if isinstance(pretrained, str):
            sd = torch.load(pretrained, map_location='cpu')
            sd = sd.get('state_dict', sd.get('model_state_dict', sd))
            sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}      # resnet3d.py:431-436
            self.load_state_dict(sd, strict=False)
            return
        with torch.no_grad():
            for s in self.specs:
                w = self.get_parameter(s.name + '.weight')
                if s.name.startswith('backbone.'):
                    nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu')
                else:
                    nn.init.xavier_uniform_(w)
                    self.get_parameter(s.name + '.bias').zero_()
                if s.bn:
                    self.get_parameter(s.bn + '.weight').fill_(1.0)
                    self.get_parameter(s.bn + '.bias').zero_()
                    self.get_buffer(s.bn + '.running_mean').zero_()
                    self.get_buffer(s.bn + '.running_var').fill_(1.0)
            if self.zero_init_residual:
                for b in self.blocks:
                    self.get_parameter(b['conv3'].bn + '.weight').zero_()

    @property
    def stem_elems(self):
        """length of the leading arena range that holds the stem's parameters (conv1 weight + its BN scale / shift): the only
        gradients that are not final when the backward plan's MARK_GRADS_BUT_STEM point is reached"""
        return self.rel_offsets[self.specs[1].name + '.weight']

    def _early_index(self):
        """table index of the first layer of the early bucket (layer3.0.conv1): specs [index:] = layer3, layer4, the neck"""
        return next(b for b in self.blocks if b['stage'] == self.EARLY_STAGE)['conv1'].index

    @property
    def early_elems(self):
        """arena offset of the early bucket: the gradients of [early_elems:] (layer3, layer4, the neck -- and, in a DetectorArena,
        every head behind the engine) are final at the backward plan's MARK_GRADS_EARLY point, in the middle of the backward pass"""
        return self.rel_offsets[self.specs[self._early_index()].name + '.weight']

    @property
    def flat_params(self):
        """the TRAINABLE range of the arena (no BN running statistics): what an optimiser may step"""
        return self._flat_params_view

    @property
    def flat_grads(self):
        return self._flat_grads_view

    def zero_grad(self, set_to_none=False):
        """gradients live in the arena for good: they are zeroed in place, never detached (set_to_none is ignored)"""
        self.flat_grads.zero_()
        self._relink_grads()

    def _relink_grads(self):
        """`optimizer.zero_grad()` / `nn.Module.zero_grad()` of a parent default to set_to_none=True on torch 2.x, which
        would orphan every parameter from the gradient arena (the engine does not hand gradients to autograd).  Returns
        True when some parameter had lost its view -- the caller then treats the arena as freshly zeroed."""
        lost = False
        for p, gview in self._grad_views.values():
            if p.grad is None or p.grad.data_ptr() != gview.data_ptr():
                p.grad = gview
                lost = True
        return lost

    # ------------------------------------------------------------------ plan construction
    def _w(self, spec, kind):
        if self.bf16 and kind in ('wf', 'wb') and spec.k != 7:
            off = self._layer_offsets16[spec.index][kind]
            assert off >= 0
            return self._packed16.data_ptr() + 2 * off
        off = getattr(self._table_host[spec.index], kind + '_off')
        assert off >= 0
        return self._packed.data_ptr() + 4 * off

    def _op(self, kind, ptrs=(), ints=(), alpha=0.0):
        """plan record whose activations have the engine's storage type (mr3d_op.i[25])"""
        o = _hip.make_op(kind, ptrs, ints, alpha)
        if self.bf16:
            o.i[25] = _hip.DTYPE_BF16
        return o

    @staticmethod
    def _side(op):
        """the record may run on the plan's side stream (mr3d_op.i[26]); its readers sit behind a SYNC record"""
        op.i[26] = 1
        return op

    def _s(self, spec, kind):
        return self._scratch.data_ptr() + 4 * getattr(self._table_host[spec.index], kind + '_off')

    @staticmethod
    def _conv_ints(x_shape, cin, cout, k, stride, pad, up=None, relu=0, ys=None, ydims=None):
        n, d, h, w, _ = x_shape
        ints = [n, d, h, w, cin, cout, k, k, k, stride[0], stride[1], stride[2], pad, pad, pad]
        ints += list(up) if up is not None else [0, 0, 0]
        ints += [int(relu)]
        ints += list(ys) if ys is not None else [0, 0, 0]
        ints += list(ydims) if ydims is not None else [0, 0, 0]
        return ints

    def _conv_fwd(self, x, spec, y, add1=None, add2=None, up=None, relu=False):
        ints = self._conv_ints(x.shape, spec.cin_pad, spec.cout, spec.k, spec.stride, spec.pad,
                               up=None if up is None else up.shape[1:4], relu=relu)
        return self._op(_hip.OP_CONV_FWD, [x, self._w(spec, 'wf'), y, self._w(spec, 'shift'), add1, add2, up, None], ints)

    def _conv_dgrad(self, dy, spec, dx, add1=None, add2=None, mask=None, scatter=False):
        """dx = conv_transpose(dy) as a stride-1 conv with flipped/transposed folded weights.  `dy` must already
        be on dx's grid (dense, or zero-stuffed for a strided 3x3x3); scatter=True instead writes a pointwise
        stride-2 result onto the even positions of the pre-zeroed dx."""
        k, pad = spec.k, spec.k - 1 - spec.pad
        if scatter:
            ints = self._conv_ints(dy.shape, spec.cout, spec.cin_pad, k, (1, 1, 1), pad, ys=spec.stride, ydims=dx.shape[1:4])
        else:
            ints = self._conv_ints(dy.shape, spec.cout, spec.cin_pad, k, (1, 1, 1), pad)
        return self._op(_hip.OP_CONV_FWD, [dy, self._w(spec, 'wb'), dx, None, add1, add2, None, mask], ints)

    def _conv_wgrad(self, x, dy, spec, sparse_dy=False):
        n, d, h, w, _ = x.shape
        ints = [n, d, h, w, spec.cin_pad, spec.cout, spec.k, spec.k, spec.k, *spec.stride, spec.pad, spec.pad, spec.pad]
        ints = ints + [0] * (26 - len(ints)) + [1]          # i[26] = 1: may run on the plan's side stream
        if sparse_dy:
            ints[17] = _hip.WGRAD_SPARSE_DY                  # zero dy tiles are tested for and skipped
        return self._op(_hip.OP_CONV_WGRAD, [x, dy, self._s(spec, 'dwraw'), self._s(spec, 'dbias')], ints)

    def _new(self, shape, dtype=None):
        t = torch.empty(tuple(shape), dtype=dtype or self.act_dtype, device=self.device)
        self._building.append(t)      # plans hold raw pointers: every buffer must outlive them
        return t

    def _out_shape(self, x_shape, spec):
        return _hip.conv3d_out_shape(x_shape, spec.cout, (spec.k,) * 3, spec.stride, (spec.pad,) * 3)

    def _build_plan(self, in_shape):
        N, C, D, H, W = in_shape
        assert C == 3, 'the detector asserts 3 input channels (two_stage_3d_2scales.py:148)'
        P = type('PlanState', (), {})()
        self._building = P.buffers = []
        fwd, bwd = [], []
        nL = len(self.specs)
        dsz = ctypes.sizeof(_hip.LayerDesc)
        # Weight preparation (BN fold + repack of all 61 layers, ~60 us): only the stem's descriptor is needed before the stem
        # kernel, so the rest runs on the plan's side stream underneath it and is joined in front of the first other conv
        fwd.append(_hip.make_op(_hip.OP_PREP_WEIGHTS, [self._table.data_ptr() + dsz, self._arena, self._packed], [nL - 1], side=True))
        fwd.append(_hip.make_op(_hip.OP_PREP_WEIGHTS, [self._table, self._arena, self._packed], [1]))
        if self.bf16:
            fwd.append(_hip.make_op(_hip.OP_PREP_WEIGHTS_BF16, [self._table16, self._arena, self._packed16], [self._table16_len], side=True))
        # ---------------- forward
        P.x_in = self._new((N, 3, D, H, W), torch.float32)
        stem = self.specs[0]
        P.stem_out = self._new(self._out_shape((N, D, H, W, 4), stem))
        # the stem kernels read the planar volume directly (no NDHWC repack)
        # bf16 engine: the stem runs on the bf16 matrix cores too (csrc/stem_bf16.hip)
        P.stem_bf16 = self.bf16
        if P.stem_bf16:
            fwd.append(_hip.make_op(_hip.OP_STEM_PACK_BF16, [self._w(stem, 'wf'), self._stem_w16], []))
            fwd.append(self._op(_hip.OP_STEM_FWD, [P.x_in, self._stem_w16, self._w(stem, 'shift'), P.stem_out], [N, D, H, W, 1, 1]))
        else:
            fwd.append(self._op(_hip.OP_STEM_FWD, [P.x_in, self._w(stem, 'wf'), self._w(stem, 'shift'), P.stem_out],
                                [N, D, H, W, 1]))
        _, sd, sh, sw, _ = P.stem_out.shape
        pshape = (N, (sd - 1) // 2 + 1, (sh - 1) // 2 + 1, (sw - 1) // 2 + 1, 16)
        P.pool_out = self._new(pshape)
        P.pool_idx = self._new(pshape[:4] + (4,), torch.int32)
        fwd.append(self._op(_hip.OP_MAXPOOL_FWD, [P.stem_out, P.pool_out, P.pool_idx], [N, sd, sh, sw, 16]))
        fwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_JOIN]))          # the other layers' weights are ready from here on
        x = P.pool_out
        P.blk = []
        stage_outs = []
        # Small maps (layer3 / layer4; fp32): a block's conv3 (+ identity + ReLU) and the NEXT block's conv1 (+ ReLU) are one
        # launch (csrc/pw_pair.hip) -- neither needs a halo, the 4x-wide block output stays in LDS for the second product.
        def pair_ok(b, nb):
            if self.bf16 or nb is None or nb['stage'] != b['stage'] or nb['ds']:
                return False
            pos = int(np.prod(self._out_shape(self._out_shape(x.shape, b['conv1']), b['conv2'])[:4]))
            return _hip.pw_pair_supported(b['planes'], b['planes'] * 4, nb['planes'], pos)
        pending_a1 = None
        for bi, b in enumerate(self.blocks):
            nb = self.blocks[bi + 1] if bi + 1 < len(self.blocks) else None
            st = dict(x=x)
            if b['ds']:
                # the projection shortcut only meets the chain at conv3's residual add: it runs on the side stream beside
                # conv1 / conv2 (a streaming pointwise kernel next to the 3x3x3 matrix work; 147 us of the forward chain)
                st['idn'] = self._new(self._out_shape(x.shape, b['dsconv']))
                fwd.append(self._side(self._conv_fwd(x, b['dsconv'], st['idn'])))
            if pending_a1 is not None:
                st['a1'] = pending_a1                    # already produced by the previous block's pair launch
            else:
                st['a1'] = self._new(self._out_shape(x.shape, b['conv1']))
                fwd.append(self._conv_fwd(x, b['conv1'], st['a1'], relu=True))
            st['a2'] = self._new(self._out_shape(st['a1'].shape, b['conv2']))
            fwd.append(self._conv_fwd(st['a1'], b['conv2'], st['a2'], relu=True))
            oshape = self._out_shape(st['a2'].shape, b['conv3'])
            if b['ds']:
                assert tuple(st['idn'].shape) == tuple(oshape)
                fwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_JOIN]))
                idn = st['idn']
            else:
                idn = x
            st['out'] = self._new(oshape)
            st['pair_next'] = pair_ok(b, nb)
            if st['pair_next']:
                pending_a1 = self._new(self._out_shape(oshape, nb['conv1']))
                fwd.append(_hip.make_pw_pair_op(st['a2'], self._w(b['conv3'], 'wf'), self._w(b['conv3'], 'shift'), idn, None, st['out'],
                                                self._w(nb['conv1'], 'wf'), self._w(nb['conv1'], 'shift'), None, pending_a1,
                                                b['planes'], b['planes'] * 4, nb['planes'], True, True, int(np.prod(oshape[:4]))))
            else:
                pending_a1 = None
                fwd.append(self._conv_fwd(st['a2'], b['conv3'], st['out'], add1=idn, relu=True))
            P.blk.append(st)
            x = st['out']
            last_of_stage = b['block'] == self.STAGE_BLOCKS[self.depth][b['stage']] - 1
            if last_of_stage:
                stage_outs.append(x)
        P.C = stage_outs
        P.lat = [None] * 4
        P.outs = [None] * 4
        p6 = None
        # The P3..P5 output convolutions and P6 (small maps: tens to hundreds of workgroups each) run on the side stream, and each is
        # queued RIGHT BEHIND the lateral convolution that produces its input: they run beside the finer laterals (the P2 one streams
        # 0.3 GB and leaves the matrix cores idle) and are done when the P2 output convolution starts.  Queued behind ALL laterals
        # (rounds 3-4: 11.05 -> 11.01 ms, beside the P2 convolution) they now wait for it: its wide form (conv3d_wino2d.hip) keeps one
        # 512-thread workgroup per CU, and nothing else fits beside that.  Nothing reads them before the plan's final join.
        interleave = not self.neck_dcn and _hip.lib().mr3d_tuning_get(self.TUNE_FPN_AFTER_LATERALS) == 0
        for i in (3, 2, 1, 0):
            P.lat[i] = self._new(self._out_shape(P.C[i].shape, self.lateral[i]))
            fwd.append(self._conv_fwd(P.C[i], self.lateral[i], P.lat[i], up=P.lat[i + 1] if i < 3 else None))
            if interleave and i > 0:
                P.outs[i] = self._new(self._out_shape(P.lat[i].shape, self.fpnconv[i]))
                fwd.append(self._side(self._conv_fwd(P.lat[i], self.fpnconv[i], P.outs[i])))
                if i == 3:
                    _, d5, h5, w5, _ = P.outs[3].shape
                    p6 = self._new((N, (d5 - 1) // 2 + 1, (h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1, self.fpn_out))
                    fwd.append(self._side(self._op(_hip.OP_SUBSAMPLE_FWD, [P.outs[3], p6], [N, d5, h5, w5, self.fpn_out])))
        if self.neck_dcn:
            P.outs = list(P.lat)                    # the deformable neck (detector/dcn_neck.py) takes over from here
        else:
            if not interleave:
                for i in (3, 2, 1):
                    P.outs[i] = self._new(self._out_shape(P.lat[i].shape, self.fpnconv[i]))
                    fwd.append(self._side(self._conv_fwd(P.lat[i], self.fpnconv[i], P.outs[i])))
                    if i == 3:
                        _, d5, h5, w5, _ = P.outs[3].shape
                        p6 = self._new((N, (d5 - 1) // 2 + 1, (h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1, self.fpn_out))
                        fwd.append(self._side(self._op(_hip.OP_SUBSAMPLE_FWD, [P.outs[3], p6], [N, d5, h5, w5, self.fpn_out])))
            P.outs[0] = self._new(self._out_shape(P.lat[0].shape, self.fpnconv[0]))
            fwd.append(self._conv_fwd(P.lat[0], self.fpnconv[0], P.outs[0]))
            P.outs.append(p6)

        # ---------------- backward
        P.gouts = [self._new(o.shape) for o in P.outs]
        # the weight-gradient scratch is zeroed on the side stream (every conv weight gradient queues behind it there); the
        # stem weight gradient, which runs on the main stream, waits for that point of the side queue (slot 0)
        bwd.append(_hip.make_op(_hip.OP_MEMSET0, [self._scratch], _hip.split64(self._scratch.numel() * 4), side=True))
        bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_MARK, 0]))
        if self.neck_dcn:
            glat = P.gouts                          # gradients w.r.t. the lateral sums arrive from the deformable neck
            P.n_direct = 1                          # glat[1..3] also receive the upsample adjoint in place: plan-owned
        else:
            bwd.append(self._op(_hip.OP_SUBSAMPLE_BWD, [P.gouts[4], P.gouts[3]], [N, d5, h5, w5, self.fpn_out]))
            glat = [self._new(l.shape) for l in P.lat]
            P.n_direct = 3                          # P2..P4 gradients are only read; the P5 buffer also takes P6's
        gC_lat = [self._new(c.shape) for c in P.C]
        lat_on_side = not self.bf16          # fp32, 300 steps each way: 10.146 -> 10.066 ms

        def fpn_dgrad(i, side):
            ops = []
            if self.sparse_pyramid_grads and not self.bf16 and P.gouts[i].shape[1] * P.gouts[i].shape[2] * P.gouts[i].shape[3] >= 65536:
                # voxel liveness of the incoming gradient, handed to the data-gradient record behind it (large maps: the
                # two-dimensional Winograd kernel writes tiles whose halo holds no live voxel from its epilogue alone)
                n_, d_, h_, w_, c_ = P.gouts[i].shape
                lbits = torch.empty((_hip.voxel_liveness_words(n_ * d_ * h_ * w_),), dtype=torch.int32, device=self.device)
                P.keep = getattr(P, 'keep', []) + [lbits]
                ops.append(_hip.make_op(_hip.OP_VOXEL_LIVENESS, [P.gouts[i], lbits], [n_, d_, h_, w_, c_, 1]))
            ops.append(self._conv_dgrad(P.gouts[i], self.fpnconv[i], glat[i]))
            return [self._side(o) for o in ops] if side else ops

        # bf16 only: in the fp32 step the side queue (every weight gradient) has no slack -- whatever it still holds when the
        # stem weight gradient starts runs after it (the stem's persistent workgroups leave no room for an LDS-heavy kernel):
        # measured 11.07 -> 11.18 ms with these 0.17 ms moved over; bf16 step, 300 steps each way: 5.57 -> 5.51 ms
        coarse_on_side = self.bf16           # (fp32 re-measured with two side queues: 9.98 -> 10.01 ms)
        if not self.neck_dcn:
            # the data gradients of the P3..P5 output convolutions only meet the main chain at the upsample adjoints: they are
            # queued on the side stream in front of the weight gradients, beside the P2 one
            if coarse_on_side:
                for i in (3, 2, 1):
                    bwd += fpn_dgrad(i, True)
                bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_MARK, self.MARK_COARSE_DGRADS]))
        for i in range(4):
            if not self.neck_dcn:
                # P2: the data gradient FIRST -- the whole backward chain waits for it (the top-down adds carry its result up to P5,
                # where the backbone backward starts), so it runs alone (0.56 instead of 0.78 ms) and its 0.75 ms weight gradient
                # runs beside the small-map kernels that follow: 10.02 -> 9.98 ms
                wg_first = i != 0
                if wg_first:
                    bwd.append(self._conv_wgrad(P.lat[i], P.gouts[i], self.fpnconv[i], sparse_dy=self.sparse_pyramid_grads))
                if i == 0 or not coarse_on_side:
                    bwd += fpn_dgrad(i, False)
                elif i == 1:
                    bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_WAIT, self.MARK_COARSE_DGRADS]))
                if not wg_first:
                    bwd.append(self._conv_wgrad(P.lat[i], P.gouts[i], self.fpnconv[i], sparse_dy=self.sparse_pyramid_grads))
            if i > 0:
                nf, df, hf, wf, c = glat[i - 1].shape
                bwd.append(self._op(_hip.OP_UPSAMPLE_ADJ, [glat[i - 1], glat[i]],
                                    [N, df, hf, wf, glat[i].shape[1], glat[i].shape[2], glat[i].shape[3], c]))
            bwd.append(self._conv_wgrad(P.C[i], glat[i], self.lateral[i]))
            # the gradient of the LAST stage output only comes through its lateral conv: apply its ReLU mask here
            # gC_lat[i] is read where the backbone backward reaches the end of stage i: the P2 / P3 lateral data gradients (0.6 ms of
            # streaming pointwise work while the chain behind them is matrix-bound) leave the main chain for the side queue
            op = self._conv_dgrad(glat[i], self.lateral[i], gC_lat[i], mask=P.C[3] if i == 3 else None)
            if lat_on_side and i < 2 and not self.neck_dcn:
                bwd.append(self._side(op))
                bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_MARK, self.MARK_LATERAL_DGRAD + i]))
            else:
                bwd.append(op)
        g = gC_lat[3]                     # grad w.r.t. the current block's output, ReLU-masked
        pending_g_a2 = None
        # wgrad_defer (A/B knob, default off): weight-gradient records of the listed stages are collected and issued in ONE run at
        # the end of the stage's data-gradient chain (consecutive side records share one fork event of the main queue)
        defer_stages = set(getattr(self, 'wgrad_defer', ()) or ())
        deferred = []

        def emit_wgrad(op, stage):
            (deferred if stage in defer_stages else bwd).append(op)
        for bi in range(len(self.blocks) - 1, -1, -1):
            b, st = self.blocks[bi], P.blk[bi]
            x = st['x']
            if pending_g_a2 is not None:
                g_a2 = pending_g_a2                      # produced by the pair launch that closed the block behind this one
            else:
                g_a2 = self._new(st['a2'].shape)
                bwd.append(self._conv_dgrad(g, b['conv3'], g_a2, mask=st['a2']))
            pending_g_a2 = None
            # conv3's and conv2's weight gradients behind conv3's data gradient: the two side records share ONE fork event of
            # the caller's stream (plan.hip) -- an event record delays the kernel behind it in that queue by ~6 us
            emit_wgrad(self._conv_wgrad(st['a2'], g, b['conv3']), b['stage'])
            emit_wgrad(self._conv_wgrad(st['a1'], g_a2, b['conv2']), b['stage'])
            g_a1 = self._new(st['a1'].shape)
            if b['stride'] == 1:
                bwd.append(self._conv_dgrad(g_a2, b['conv2'], g_a1, mask=st['a1']))
            else:
                # strided 3x3x3: parity-decomposed data gradient straight from the coarse g_a2 (no zero-stuffing; the bf16 plan
                # zero-stuffed until round 4: 8x the positions through the stride-1 kernel, 7/8 of them zeros)
                sp = b['conv2']
                ints = self._conv_ints(g_a2.shape, sp.cout, sp.cin_pad, 3, (2, 2, 2), 1, ydims=g_a1.shape[1:4])
                bwd.append(self._op(_hip.OP_CONV_DGRAD_S2, [g_a2, self._w(sp, 'wb'), g_a1, None, None, None, None, st['a1']], ints))
            emit_wgrad(self._conv_wgrad(x, g_a1, b['conv1']), b['stage'])          # (behind conv1's data gradient instead: 9.99 -> 10.25 ms)
            first_of_stage = b['block'] == 0
            lat_part = gC_lat[b['stage'] - 1] if (first_of_stage and b['stage'] > 0) else None
            if lat_part is not None and lat_on_side and b['stage'] - 1 < 2 and not self.neck_dcn:
                bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_WAIT, self.MARK_LATERAL_DGRAD + b['stage'] - 1]))
            mask = x if bi > 0 else None          # x of the very first block is the max-pool output (no ReLU after it)
            gx = self._new(x.shape)
            if b['ds']:
                emit_wgrad(self._conv_wgrad(x, g, b['dsconv']), b['stage'])
                if b['stride'] == 1:
                    t = self._new(x.shape)
                    bwd.append(self._conv_dgrad(g, b['dsconv'], t))
                    bwd.append(self._conv_dgrad(g_a1, b['conv1'], gx, add1=t, add2=lat_part, mask=mask))
                elif self.bf16:
                    t = self._new(x.shape)
                    bwd.append(_hip.make_op(_hip.OP_MEMSET0, [t], _hip.split64(t.numel() * t.element_size())))
                    bwd.append(self._conv_dgrad(g, b['dsconv'], t, scatter=True))
                    bwd.append(self._conv_dgrad(g_a1, b['conv1'], gx, add1=t, add2=lat_part, mask=mask))
                else:
                    # strided 1x1x1 downsample conv: its data gradient lives on the COARSE grid (one value per 2x2x2 cell of
                    # x, at the even positions).  It is computed there and the conv1 data gradient adds it at the even
                    # positions in its epilogue (MR3D_CONV_UP_EVEN): the zero-stuffed fine map (168 MB at layer2) is neither
                    # zero-filled, nor scattered into, nor read back.
                    tc = self._new(g.shape[:4] + (b['dsconv'].cin_pad,))
                    bwd.append(self._conv_dgrad(g, b['dsconv'], tc))
                    op = self._conv_dgrad(g_a1, b['conv1'], gx, add1=lat_part, mask=mask)
                    op.p[6] = tc.data_ptr()
                    op.i[15], op.i[16], op.i[17] = tc.shape[1:4]
                    op.i[18] |= _hip.CONV_UP_EVEN
                    bwd.append(op)
            else:
                assert lat_part is None
                pst = P.blk[bi - 1]
                if pst['pair_next']:
                    # mirror of the forward pair: gx = mask_x(conv1^T(g_a1) + g), then the previous block's
                    # g_a2 = mask_a2(conv3^T(gx)), in one launch
                    pb = self.blocks[bi - 1]
                    pending_g_a2 = self._new(pst['a2'].shape)
                    bwd.append(_hip.make_pw_pair_op(g_a1, self._w(b['conv1'], 'wb'), None, g, mask, gx,
                                                    self._w(pb['conv3'], 'wb'), None, pst['a2'], pending_g_a2,
                                                    b['planes'], b['planes'] * 4, pb['planes'], False, False, int(np.prod(gx.shape[:4]))))
                else:
                    bwd.append(self._conv_dgrad(g_a1, b['conv1'], gx, add1=g, mask=mask))
            g = gx
            if b['block'] == 0 and deferred:          # the stage's data-gradient chain is complete: its weight gradients, one run
                bwd.extend(deferred)
                deferred = []
            if b['block'] == 0 and b['stage'] == self.EARLY_STAGE and self.early_bucket:
                # Early bucket of the gradient exchange: every weight gradient of layer3, layer4 and the neck has been issued; their
                # finishing (dW = s dW', BN gamma / beta) runs on the side queue NOW, underneath the backward pass of layer2 / layer1,
                # and slot MARK_GRADS_EARLY tells the communication stream that the arena from `early_elems` on -- 99 % of the
                # message once the heads sit behind the engine -- may be exchanged (dist_utils.allreduce_overlapped).  Until round 5
                # ONE finishing ran beside the stem weight gradient; the gather form of that kernel occupies the chip's wave slots
                # and the finishing of all 60 layers ended with it, i.e. with the backward pass: nothing left to overlap.
                k0 = self._early_index()
                bwd.append(_hip.make_op(_hip.OP_FINISH_GRADS, [self._table.data_ptr() + k0 * dsz, self._arena, self._scratch, self._grad_arena],
                                        [nL - k0, max(sp.cout for sp in self.specs[k0:])], side=True))
                bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_MARK, self.MARK_GRADS_EARLY]))
        # Gradient finishing (dW = s dW', BN gamma / beta from the raw sums, ~0.11 ms) of every layer but the stem: queued on
        # the side stream behind the last conv weight gradient -- a side record waits for the main chain only up to the point
        # where it is issued, so it is issued HERE, in front of the max-pool backward and the stem weight gradient, and runs
        # underneath them.  The stem weight gradient (1.8 ms) needs the last data gradient and closes the main chain; only its
        # own 16 channels are finished after the join.
        k0 = self._early_index() if self.early_bucket else nL          # (layer1 / layer2 only: the later stages were finished in the early bucket)
        bwd.append(_hip.make_op(_hip.OP_FINISH_GRADS, [self._table.data_ptr() + dsz, self._arena, self._scratch, self._grad_arena],
                                [k0 - 1, max(sp.cout for sp in self.specs[1:k0])], side=True))
        # slot 1: "every parameter gradient but the stem's is final" -- the point the overlapped gradient exchange waits for
        bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_MARK, self.MARK_GRADS_BUT_STEM]))
        # the max-pool backward and the stem's ReLU mask happen inside the weight gradient's dy staging
        # (mr3d_stem_conv_wgrad_pooled / _bf16_pooled) -- no un-pooled gradient buffer (335 MB fp32 on the 512 volume), one
        # launch less on the main chain
        bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_WAIT, 0]))
        # its own partial-sum workspace (one block of [343][3][16] per workgroup, 34 MB): the shared one belongs to the side queue,
        # whose last weight gradients still run beside this kernel; without one the 512 workgroups add 16,464 values each with
        # same-address atomics (0.3 ms of the gather kernel's 1.1)
        ws_bytes = 0 if self.bf16 else 2 * _hip.device_cus(self.device) * 343 * 3 * 16 * 4
        if ws_bytes and getattr(self, '_stem_ws', None) is None:
            self._stem_ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=self.device)
        bwd.append(self._op(_hip.OP_STEM_WGRAD, [P.x_in, g, self._s(stem, 'dwraw'), self._s(stem, 'dbias'),
                                                 self._stem_ws if ws_bytes else None, P.pool_idx, P.pool_out],
                            [N, D, H, W, *_hip.split64(ws_bytes), 1 if P.stem_bf16 else 0, 1]))
        bwd.append(_hip.make_op(_hip.OP_SYNC, [], [_hip.SYNC_JOIN]))
        bwd.append(_hip.make_op(_hip.OP_FINISH_GRADS, [self._table, self._arena, self._scratch, self._grad_arena], [1, stem.cout]))
        P.fwd_ops, P.bwd_ops = fwd, bwd
        P.fwd, P.bwd = _hip.Plan(fwd), _hip.Plan(bwd)
        # zero-copy binding (see bind_input / bind_grads): where the plans refer to the input volume and to the
        # read-only output gradients (P2..P4; the P5 buffer also receives P6's gradient in place, so it stays owned)
        P.x_refs = (P.fwd.find_ptr(P.x_in.data_ptr()), P.bwd.find_ptr(P.x_in.data_ptr()))
        P.g_refs = [P.bwd.find_ptr(g.data_ptr()) for g in P.gouts[:P.n_direct]]
        P.bound_x, P.bound_g = None, [None] * P.n_direct
        self._building = None
        return P

    def _plan_for(self, shape, slot=0):
        """one plan (kernel records + every activation / gradient buffer) per input shape and slot; two forward passes
        whose backward passes are both still pending (the two pathways of the shipped model,
        two_stage_3d_2scales.py:149-150) must use different slots if they have the same shape"""
        key = tuple(shape) + (int(slot),)
        if key not in self._plans:
            self._plans[key] = self._build_plan(tuple(shape))
        return self._plans[key]

    # ------------------------------------------------------------------ execution
    def forward_features(self, img, slot=0):
        """img: [N,3,D,H,W] fp32 on the engine's device -> (P2..P6) as [N,64,d,h,w] channels_last_3d views."""
        if not img.is_cuda:
            raise NotImplementedError('ExtractFeat3D runs on the MI355X only (no CPU fallback)')
        P = self._plan_for(img.shape, slot)
        self.bind_input(P, img)
        with torch.cuda.device(self.device):
            P.fwd.run(self.device)
        self._last = P
        if self.bf16 and self.pyramid_dtype == torch.bfloat16:
            return tuple(o.permute(0, 4, 1, 2, 3).clone(memory_format=torch.preserve_format) for o in P.outs)
        if self.bf16:       # the pyramid leaves in fp32 (any head, RoIAlign3D and the losses compute in fp32)
            return tuple(o.permute(0, 4, 1, 2, 3).float() for o in P.outs)
        return tuple(o.permute(0, 4, 1, 2, 3) for o in P.outs)

    def capture(self, P):
        """hipGraph replay of both passes of a plan (mr3d_plan_capture).  A graph bakes its addresses in, so the plan
        is first re-pointed at its own staging buffers; from then on inputs and output gradients are copied."""
        P.fwd.repoint(P.x_refs[0], P.x_in.data_ptr())
        P.bwd.repoint(P.x_refs[1], P.x_in.data_ptr())
        for refs, g in zip(P.g_refs, P.gouts):
            P.bwd.repoint(refs, g.data_ptr())
        P.bound_x, P.bound_g = None, [None] * P.n_direct
        P.fwd.capture()
        P.bwd.capture()

    @staticmethod
    def _captured(P):
        return getattr(P.fwd, '_graph', None) is not None or getattr(P.bwd, '_graph', None) is not None

    def bind_input(self, P, img):
        """Point the stem kernels (forward, and the weight gradient of the matching backward) at the caller's volume
        when it is a dense fp32 NCDHW tensor -- no 252 MB staging copy; the plan keeps a reference until the next bind,
        and the caller must leave the tensor unmodified until the backward pass has run (as for any tensor autograd
        saves).  Other layouts / dtypes, and plans captured into a hipGraph, take the copy."""
        direct = (img.dtype == torch.float32 and img.is_contiguous() and img.device == P.x_in.device and
                  img.data_ptr() % 16 == 0 and not self._captured(P))
        src = img if direct else P.x_in
        if not direct:
            P.x_in.copy_(img)
        if P.bound_x is None or P.bound_x.data_ptr() != src.data_ptr():
            if not self._captured(P):
                P.fwd.repoint(P.x_refs[0], src.data_ptr())
                P.bwd.repoint(P.x_refs[1], src.data_ptr())
        P.bound_x, P.bound_x_version = src, src._version

    def bind_grads(self, P, grads):
        """Output gradients: P2..P4 are read in place when they arrive as dense channels-last fp32 tensors (they
        are only read); P5 / P6 (and anything else) are copied into the plan's buffers."""
        for i, (gbuf, g) in enumerate(zip(P.gouts, grads)):
            if g is None:
                gbuf.zero_()
                src = gbuf
            else:
                gl = g.permute(0, 2, 3, 4, 1)
                direct = (i < P.n_direct and g.dtype == gbuf.dtype and gl.is_contiguous() and g.device == gbuf.device and
                          g.data_ptr() % 16 == 0 and not self._captured(P))
                src = gl if direct else gbuf
                if not direct:
                    gbuf.copy_(gl)
            if i < P.n_direct:
                if (P.bound_g[i] is None or P.bound_g[i].data_ptr() != src.data_ptr()) and not self._captured(P):
                    P.bwd.repoint(P.g_refs[i], src.data_ptr())
                P.bound_g[i] = src

    def backward_features(self, grads, plan=None):
        """grads: gradients w.r.t. the 5 pyramid outputs (any memory format; None = zero).  Accumulates parameter
        gradients into the flat gradient arena (== every parameter's .grad).  plan: the forward pass this backward
        belongs to (default: the most recent one)."""
        P = plan if plan is not None else self._last
        if P.bound_x is not P.x_in and P.bound_x._version != P.bound_x_version:
            raise RuntimeError('the input volume bound to this plan was modified in place between forward and backward '
                               '(the stem weight gradient reads it)')
        self.bind_grads(P, grads)
        if self._relink_grads():
            # someone set the parameters' .grad to None (torch's zero_grad default): that means "zero", and the arena
            # must not keep the previous step's sums
            self.flat_grads.zero_()
        with torch.cuda.device(self.device):
            P.bwd.run(self.device)
        self.grads_touched = True           # every engine parameter has had a gradient (dist_utils.ArenaSGD / DetectorArena)

    def forward(self, img, slot=0):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _ExtractFeatFn.apply(self, img, self._autograd_anchor(), slot)
        return self.forward_features(img, slot)

    def _autograd_anchor(self):
        if not hasattr(self, '_anchor'):
            self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        return self._anchor

    def train(self, mode=True):
        # BatchNorm3d stays in eval mode either way (norm_eval=True, resnet3d.py:480-486)
        return super().train(mode)


class _ExtractFeatFn(torch.autograd.Function):
    """autograd bridge: the engine writes parameter gradients straight into the flat gradient arena
    (p.grad of every parameter is a view of it), so backward returns no tensor gradients."""

    @staticmethod
    def forward(ctx, engine, img, anchor, slot=0):
        ctx.engine = engine
        outs = engine.forward_features(img, slot)
        ctx.plan = engine._last
        if engine.bf16 or not engine.clone_outputs:
            return outs                   # bf16: already copies of the plan's buffers (fp32 or bf16, see pyramid_dtype)
        return tuple(o.clone(memory_format=torch.preserve_format) for o in outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        ctx.engine.backward_features(grads, plan=ctx.plan)
        return None, None, None, None

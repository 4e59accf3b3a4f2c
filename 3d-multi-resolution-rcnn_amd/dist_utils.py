"""Data-parallel gradient exchange (the only collective on the reference's hot path).

Mirror of mmdet/core/utils/dist_utils.py:9-57 (`allreduce_grads`, `_allreduce_coalesced`,
`DistOptimizerHook.after_train_iter`) and of the one-time parameter broadcast that mmcv's
MMDistributedDataParallel performs at wrap time (mmdet/apis/train.py:92).

MI355X-first: gradients already live in ONE persistent flat fp32 arena (ExtractFeat3D.flat_grads and
FlatArena below), so the reference's per-iteration flatten + copy-back (`_flatten_dense_tensors`,
`tensor.copy_(synced)`, dist_utils.py:22-28) disappears: one in-place SUM all-reduce over RCCL/xGMI on the
arena, then one in-place 1/world scale.  Semantics kept: SUM then divide (not pre-divide), issued after
backward and before gradient clipping (dist_utils.py:51-57).  Works on any torch.distributed backend
("nccl" = RCCL on ROCm; "gloo" for the CPU tests).
"""
from collections import OrderedDict

import os

import torch
import torch.distributed as dist


def allreduce_flat_grads(flat, world_size=None, group=None):
    """in-place mean of a flat gradient buffer over the data-parallel group"""
    if world_size is None:
        world_size = dist.get_world_size(group)
    if world_size == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    return flat


_COMM_STREAMS = {}


def allreduce_overlapped(flat, stem_elems, world_size=None, group=None, mark_slot=1, force_single_rank=False, early_elems=None,
                         early_slot=3):
    """In-place mean of the flat gradient arena as TWO collectives so that the exchange runs underneath the backward pass.
    early_elems given (round 5, the default of DetectorArena / bench.py): the arena from `early_elems` on -- layer3, layer4, the neck
    and every head: 99 % of the message -- is final when the backward plan reaches its MARK_GRADS_EARLY point (slot `early_slot`: the
    finishing of those layers on the plan's side queue, in the MIDDLE of the backward pass); a communication stream waits for that point
    and carries the all-reduce of flat[early_elems:] while layer2, layer1 and the stem weight gradient still run; the caller's stream
    reduces the small front range flat[:early_elems] after the backward pass and then waits for the first collective.
    early_elems None (rounds 3-4): the split is at `stem_elems` and the first collective waits for slot `mark_slot` ("every gradient
    but the stem's"), which ran beside the stem weight gradient -- the gather form of that kernel leaves no room beside it.
    Semantics are those of allreduce_flat_grads / the reference (SUM, then divide, before the gradient clip:
    mmdet/core/utils/dist_utils.py:22-28,51-57).  CPU tensors (gloo tests): the same two collectives, one after the other.
    Call it after loss.backward() has returned (everything is enqueued by then)."""
    if world_size is None:
        world_size = dist.get_world_size(group)
    if world_size == 1 and not force_single_rank:      # force_single_rank: launch the collectives anyway (1-GPU check of the
        return flat                                     # launch structure under a profiler)
    split, slot = (stem_elems, mark_slot) if early_elems is None else (early_elems, early_slot)
    head, tail = flat[:split], flat[split:]
    if flat.is_cuda:
        from . import _hip
        dev = flat.device
        comm = _COMM_STREAMS.get(dev)
        if comm is None:
            comm = _COMM_STREAMS[dev] = torch.cuda.Stream(device=dev)
        # the communication stream starts behind the plan's mark -- IF this step's backward pass recorded it.  It did not when
        # the side stream is off, when the plan is replayed from a hipGraph, or when no plan ran: the collective then waits for
        # everything the caller's stream holds (no overlap, same result).
        if not _hip.plan_stream_wait_mark_checked(slot, comm):
            comm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(comm):
            work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=group, async_op=True)
        dist.all_reduce(head, op=dist.ReduceOp.SUM, group=group)
        work.wait()                                 # the caller's stream waits for the communication stream's collective
    else:
        dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(head, op=dist.ReduceOp.SUM, group=group)
    if world_size > 1:
        flat.div_(world_size)
    return flat


XGMI_LINKS, XGMI_LINK_GBPS = 7, 153.0      # MI355X: seven point-to-point xGMI links of ~153 GB/s per GPU ("All-reduce")


def grads_checksum(flat):
    """64-bit content checksum of a flat fp32 buffer, exact and order-free: the sum of its bit patterns as unsigned 32-bit
    integers plus 2^32 times the sum of (pattern * (index % 65521 + 1)) -- equal buffers give equal checksums on every rank and
    backend, a swapped or perturbed element changes it (device-side reduction, 8 bytes read back by the caller)"""
    bits = flat.detach().contiguous().view(torch.int32).to(torch.int64) & 0xffffffff
    idx = (torch.arange(bits.numel(), device=bits.device, dtype=torch.int64) % 65521) + 1
    lo = bits.sum() & 0xffffffff
    hi = (bits * idx).sum() & 0x7fffffff
    return (hi << 32) | lo


def ranks_hold_identical(flat, group=None):
    """After the exchange every rank must hold the SAME gradient bits (the reference copies the reduced bucket back into every
    rank's .grad, mmdet/core/utils/dist_utils.py:22-28).  all-gather of grads_checksum; returns (identical, [checksum per rank])."""
    cs = grads_checksum(flat).reshape(1)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return True, [int(cs.item())]
    got = [torch.empty_like(cs) for _ in range(world)]
    dist.all_gather(got, cs, group=group)
    vals = [int(g.item()) for g in got]
    return all(v == vals[0] for v in vals), vals


def allreduce_bus_bandwidth(nbytes, seconds, world_size):
    """bus bandwidth of an all-reduce = 2 (n - 1) / n * bytes / t (GB/s), and its fraction of the GPU's aggregate
    xGMI bandwidth (7 links x 153 GB/s) and of one link (what a ring is bound by)"""
    if seconds <= 0 or world_size < 1:
        return {'busbw_GBps': None, 'frac_of_xgmi': None, 'frac_of_one_link': None}
    bw = 2.0 * (world_size - 1) / world_size * nbytes / seconds / 1e9
    return {'busbw_GBps': bw, 'frac_of_xgmi': bw / (XGMI_LINKS * XGMI_LINK_GBPS), 'frac_of_one_link': bw / XGMI_LINK_GBPS}


def time_allreduce_standalone(flat, world_size, reps=10, group=None, sync=None):
    """seconds per in-place mean of `flat` with nothing else on the device (max over ranks): the cost the step would pay with no
    overlap at all.  sync: callable that drains the device (torch.cuda.synchronize on a GPU, nothing on the gloo tests)."""
    import time
    sync = sync or (lambda: None)
    allreduce_flat_grads(flat, world_size, group) if world_size > 1 else dist.all_reduce(flat, group=group)
    sync()
    if world_size > 1:
        dist.barrier(group=group)
    t0 = time.perf_counter()
    for _ in range(reps):
        allreduce_flat_grads(flat, world_size, group) if world_size > 1 else dist.all_reduce(flat, group=group)
    sync()
    el = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=flat.device)
    if world_size > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX, group=group)
    return float(el.item())


def broadcast_params(flat, src=0, group=None):
    """identical initial weights on every rank (what MMDistributedDataParallel.__init__ does with its
    coalesced state_dict broadcast, mmdet/apis/train.py:92)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def _keep_grad_in_arena(p, gview, on_touch=None):
    """torch's zero_grad (set_to_none=True by default on 2.x) detaches a parameter from its arena slot and autograd then
    allocates a fresh gradient.  This hook moves such a gradient back: `None` meant zero, so the slot is overwritten.
    on_touch(): called every time autograd has accumulated a gradient into the parameter (see FlatArena.touched)."""
    def hook(param):
        if param.grad is not None and param.grad.data_ptr() != gview.data_ptr():
            gview.copy_(param.grad)
            param.grad = gview
        if on_touch is not None:
            on_touch()
    p.register_post_accumulate_grad_hook(hook)


class FlatArena:
    """Re-homes the parameters (and their .grad) of arbitrary nn.Modules into two persistent flat fp32
    buffers, so that heads written in plain PyTorch join the same single all-reduce as the conv engine.
    storage = (flat_params, flat_grads, offset): use a slice of caller-owned buffers instead of allocating."""

    def __init__(self, params, storage=None):
        params = [p for p in params if p.requires_grad]
        assert params, 'no trainable parameters'
        dev = params[0].device
        n = sum(p.numel() for p in params)
        if storage is None:
            self.flat_params = torch.empty(n, dtype=torch.float32, device=dev)
            self.flat_grads = torch.zeros(n, dtype=torch.float32, device=dev)
        else:
            fp, fg, at = storage
            self.flat_params, self.flat_grads = fp[at:at + n], fg[at:at + n]
        off = 0
        # touched[i]: parameter i has received a gradient at least once.  In the reference a parameter whose .grad is still None
        # is neither exchanged nor stepped (mmdet/core/utils/dist_utils.py:32-35; torch.optim.SGD skips `p.grad is None`), so
        # weight decay does not reach e.g. the mask head before its first positive roi.  Here every .grad is an arena view
        # from the start; the flags carry the difference (ArenaSGD reads them).
        self.touched = [False] * len(params)
        self.ranges = []
        self.version = 0                     # bumped whenever a flag flips (ArenaSGD caches its mask per version)
        with torch.no_grad():
            for i, p in enumerate(params):
                k = p.numel()
                self.flat_params[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_params[off:off + k].view(p.shape)
                gview = self.flat_grads[off:off + k].view(p.shape)
                p.grad = gview
                _keep_grad_in_arena(p, gview, on_touch=lambda i=i: self._touch(i))
                self.ranges.append((off, off + k))
                off += k
        self.params = params
        self.numel = n

    def _touch(self, i):
        if not self.touched[i]:
            self.touched[i] = True
            self.version += 1

    def zero_grad(self):
        self.flat_grads.zero_()
        for p in self.params:
            if p.grad is None:
                off = p.data.storage_offset() - self.flat_params.storage_offset()
                p.grad = self.flat_grads[off:off + p.numel()].view(p.shape)


class DetectorArena:
    """ONE flat fp32 buffer for the whole detector -- conv engine (backbone + neck) first, then every head -- and one
    mirror buffer for its gradients: the single ~26 M-element message of the reference's `_allreduce_coalesced`
    (mmdet/core/utils/dist_utils.py:22-28; 103.9 MB for the shipped model) without its per-iteration flatten and
    copy-back.  Layout: [engine trainables | head parameters | engine BN running statistics]; `flat_params` /
    `flat_grads` are the trainable prefix (what the optimiser steps and the all-reduce exchanges), so weight decay and
    momentum never reach the frozen BN statistics."""

    def __init__(self, detector, engine_attr='extract'):
        eng = getattr(detector, engine_attr, None) if engine_attr else None
        prefix = (engine_attr + '.') if eng is not None else '\0'
        heads = [p for n, p in detector.named_parameters() if not n.startswith(prefix) and p.requires_grad]
        n_heads = -(-sum(p.numel() for p in heads) // 64) * 64
        n_eng, n_stats = (eng.num_trainable, eng.num_stats) if eng is not None else (0, 0)
        n_train = n_eng + n_heads
        dev = eng.flat_params.device if eng is not None else heads[0].device
        self.storage = torch.zeros(n_train + n_stats, dtype=torch.float32, device=dev)
        self.grad_storage = torch.zeros(n_train, dtype=torch.float32, device=dev)
        if eng is not None:
            eng.rehome(self.storage, self.grad_storage, 0, n_train)
        self.heads = FlatArena(heads, storage=(self.storage, self.grad_storage, n_eng)) if heads else None
        self.flat_params = self.storage[:n_train]
        self.flat_grads = self.grad_storage
        self.flat_params.grad = self.flat_grads          # an optimiser over [flat_params] steps the whole detector
        self.engine = eng

    def zero_grad(self):
        self.flat_grads.zero_()
        if self.engine is not None:
            self.engine._relink_grads()
        if self.heads is not None:
            self.heads.zero_grad()

    def untouched_ranges(self):
        """[(lo, hi)] element ranges of flat_params whose parameters have never received a gradient (none, after the first
        iteration that produced every loss): the reference does not step those (see FlatArena.touched)"""
        out = []
        if self.engine is not None and not getattr(self.engine, 'grads_touched', True):
            out.append((0, self.engine.num_trainable))
        if self.heads is not None:
            base = self.engine.num_trainable if self.engine is not None else 0
            out += [(base + lo, base + hi) for (lo, hi), t in zip(self.heads.ranges, self.heads.touched) if not t]
        return out

    def touch_version(self):
        return (self.heads.version if self.heads is not None else 0,
                bool(getattr(self.engine, 'grads_touched', True)) if self.engine is not None else True)

    def allreduce(self, world_size=None, group=None, overlap=True):
        """the one gradient exchange of a step.  overlap (default): two collectives, the first one -- layer3, layer4, the neck and
        every head -- started in the middle of the backward pass (allreduce_overlapped); False: ONE in-place all-reduce after the
        backward pass, as the reference issues it"""
        if overlap and self.engine is not None:
            # (an engine built without the early bucket never records its slot: split at the stem instead of waiting on a slot
            #  nobody arms, which would serialise the whole exchange behind the backward pass)
            early = self.engine.early_elems if getattr(self.engine, 'early_bucket', True) else None
            return allreduce_overlapped(self.flat_grads, self.engine.stem_elems, world_size, group,
                                        mark_slot=self.engine.MARK_GRADS_BUT_STEM, early_elems=early,
                                        early_slot=self.engine.MARK_GRADS_EARLY)
        return allreduce_flat_grads(self.flat_grads, world_size, group)


class ArenaSGD:
    """torch.optim.SGD (momentum, weight decay, no dampening / Nesterov: configs/3d-multi-resolution-rcnn.py:235) over a
    DetectorArena as ONE fused update of the flat buffer -- with the reference's treatment of parameters that have never
    received a gradient: torch.optim.SGD skips `p.grad is None` (no weight decay, no momentum buffer), and the reference's
    zero_grad (torch 1.0: in-place zero) keeps a gradient non-None for good once it has existed.  So a slice is stepped from
    its first gradient onwards, exactly like the per-parameter optimiser; until then it stays at its initial value.
    Interface: step(), zero_grad(), param_groups[0]['lr'] (the lr schedule hooks of the runner write it)."""

    def __init__(self, arena, lr, momentum=0.0, weight_decay=0.0):
        self.arena = arena
        self.param_groups = [dict(params=[arena.flat_params], lr=lr, momentum=momentum, weight_decay=weight_decay)]
        self.momentum_buffer = torch.zeros_like(arena.flat_params)
        self._d = torch.empty_like(arena.flat_params)
        self._mask, self._mask_version = None, None

    def _active_mask(self):
        """None when every slice is live, else a 0/1 float tensor over the arena (rebuilt only when a flag flips)"""
        version = self.arena.touch_version()
        if version != self._mask_version:
            ranges = self.arena.untouched_ranges()
            if not ranges:
                self._mask = None
            else:
                self._mask = torch.ones_like(self.arena.flat_params)
                for lo, hi in ranges:
                    self._mask[lo:hi] = 0
            self._mask_version = version
        return self._mask

    @torch.no_grad()
    def step(self):
        g = self.param_groups[0]
        p, grad, buf, d = self.arena.flat_params, self.arena.flat_grads, self.momentum_buffer, self._d
        torch.add(grad, p, alpha=g['weight_decay'], out=d)          # d = grad + wd * p
        mask = self._active_mask()
        if mask is not None:
            d.mul_(mask)                                            # a never-touched slice: no decay, buffer stays zero
        if g['momentum'] != 0:
            buf.mul_(g['momentum']).add_(d)                         # first live step: buf = d (torch clones d there)
            p.add_(buf, alpha=-g['lr'])
        else:
            p.add_(d, alpha=-g['lr'])

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def state_dict(self):
        return {'momentum_buffer': self.momentum_buffer, 'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]}

    def load_state_dict(self, sd):
        self.momentum_buffer.copy_(sd['momentum_buffer'])
        self.param_groups[0].update(sd['param_groups'][0])


def _allreduce_coalesced(tensors, world_size, bucket_size_mb=-1, group=None):
    """reference-compatible path for gradients that are NOT in a flat arena (dist_utils.py:9-28): bucket by
    tensor type (or by size when bucket_size_mb > 0), flatten, SUM all-reduce, divide, copy back."""
    if bucket_size_mb > 0:
        limit, buckets, cur, size = bucket_size_mb * 1024 * 1024, [], [], 0
        for t in tensors:
            nbytes = t.numel() * t.element_size()
            if cur and (size + nbytes > limit or t.type() != cur[0].type()):
                buckets.append(cur); cur, size = [], 0
            cur.append(t); size += nbytes
        if cur:
            buckets.append(cur)
    else:
        by_type = OrderedDict()
        for t in tensors:
            by_type.setdefault(t.type(), []).append(t)
        buckets = list(by_type.values())
    for bucket in buckets:
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, group=group)
        flat.div_(world_size)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def _arena_of(grads):
    """if every gradient is a view into one contiguous storage range return that range as a flat tensor"""
    if not grads:
        return None
    base = grads[0].untyped_storage().data_ptr()
    if any(g.untyped_storage().data_ptr() != base or g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    lo = min(g.storage_offset() for g in grads)
    hi = max(g.storage_offset() + g.numel() for g in grads)
    flat = torch.empty(0, dtype=torch.float32, device=grads[0].device)
    flat.set_(grads[0].untyped_storage(), lo, (hi - lo,))
    return flat


def allreduce_grads(model, coalesce=True, bucket_size_mb=-1, group=None):
    """same call signature and result as the reference (dist_utils.py:31-41): every rank ends with the mean
    gradient.  Gradients that already share one flat arena are reduced in place with a single collective."""
    grads = [p.grad.data for p in model.parameters() if p.requires_grad and p.grad is not None]
    world_size = dist.get_world_size(group)
    if coalesce:
        flat = _arena_of(grads) if bucket_size_mb <= 0 else None
        if flat is not None:
            allreduce_flat_grads(flat, world_size, group)      # padding between views is reduced too: harmless zeros
        else:
            _allreduce_coalesced(grads, world_size, bucket_size_mb, group)
    else:
        for t in grads:
            dist.all_reduce(t.div_(world_size), group=group)   # dist_utils.py:40-41


def clip_grad_norm_flat(flat_grads_list, max_norm, norm_type=2):
    """L2 clip over one or more flat gradient arenas (grad_clip=dict(max_norm=35, norm_type=2),
    configs/3d-multi-resolution-rcnn.py:236) without a host sync"""
    assert norm_type == 2
    total = torch.sqrt(sum((g * g).sum() for g in flat_grads_list))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in flat_grads_list:
        g.mul_(coef)
    return total


class DistOptimizerStep:
    """`DistOptimizerHook.after_train_iter` (dist_utils.py:44-57) as a callable:
    zero_grad -> loss.backward() -> allreduce_grads -> clip -> optimizer.step().
    Gradients that live in a flat arena (the conv engine's, FlatArena's, DetectorArena's) are zeroed IN PLACE and stay
    attached to their parameters -- `optimizer.zero_grad()` with torch 2.x's set_to_none=True default would orphan
    them from the buffer the all-reduce, the clip and the optimiser read."""

    def __init__(self, model, optimizer, grad_clip=None, coalesce=True, bucket_size_mb=-1):
        self.model, self.optimizer, self.grad_clip = model, optimizer, grad_clip
        self.coalesce, self.bucket_size_mb = coalesce, bucket_size_mb

    def __call__(self, loss):
        self.optimizer.zero_grad(set_to_none=False)
        for m in self.model.modules():
            if hasattr(m, 'flat_grads'):
                m.zero_grad()
        loss.backward()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            allreduce_grads(self.model, self.coalesce, self.bucket_size_mb)
        if self.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad], **self.grad_clip)
        self.optimizer.step()


_gc_state = {'steps': 0, 'frozen': False}


def freeze_gc_after_warmup(warmup_steps=3):
    """OPT-IN helper for a training loop (bench.py, tools/two_stage_bench.py and tools/train_synthetic.py call it once per step;
    train_step does so only when asked, `freeze_gc=True`): after `warmup_steps` calls, one full collection, then `gc.freeze()` --
    every object alive at that point moves to the permanent generation and later collections only look at what the steps themselves
    allocate.  It is a process-wide, irreversible-until-`gc.unfreeze()` change to the HOST application's collector, which is why the
    library never makes it on its own (round 5 did, from inside train_step).  Why a loop wants it: CPython runs a full generation-2
    pass every 10-20 steps of the dual-pathway model and that pass takes 30-85 ms while the GPU drains (per-step times 45, 45, ...,
    130, 45, ...; 44.9 ms per step over 8 steps with the freeze, 50-54 without).  tools/diag/gc_census.py: of the 215 k objects
    the collector tracks once the two-stage detector, its plans and three steps exist, 170 k are PyTorch's own modules, functions
    and descriptors -- the detectors add 44 k -- so flattening the plans into one ctypes array would not take the pass away.
    Returns True in the call that froze; logs one line when it does."""
    import gc
    _gc_state['steps'] += 1
    if _gc_state['frozen'] or _gc_state['steps'] <= warmup_steps:
        return False
    gc.collect()
    gc.freeze()
    _gc_state['frozen'] = True
    import sys
    print(f'mr3d: gc.freeze() after {warmup_steps} warm-up steps ({gc.get_freeze_count()} objects made permanent; gc.unfreeze() undoes it)',
          file=sys.stderr)
    return True


def train_step(detector, arena, optimizer, batch, world_size=1, grad_clip=35.0, group=None, overlap=True, freeze_gc=False):
    """One iteration of the reference's hot loop (mmdet/apis/train.py:37-48 batch_processor + dist_utils.py:51-57):
    zero -> forward_train -> sum of the losses -> backward -> ONE all-reduce of the detector's gradient arena -> L2 clip
    (configs/...py:236) -> optimizer.step().  `batch` = (args, kwargs) of detector.forward_train.  Returns
    (total loss, loss dict) as device tensors: nothing is synchronised here.  freeze_gc=True: freeze_gc_after_warmup() per call (a
    process-wide change to the host's cyclic collector: off unless the loop's owner asks for it)."""
    if freeze_gc:
        freeze_gc_after_warmup()
    arena.zero_grad()
    args, kwargs = batch
    if hasattr(detector, 'forward_backward'):     # same gradients, head backward issued ahead of the step's one host wait
        losses = detector.forward_backward(*args, **kwargs)
        total = detector.total_loss(losses)
    else:
        losses = detector.forward_train(*args, **kwargs)
        total = detector.total_loss(losses)
        total.backward()
    if world_size > 1:
        arena.allreduce(world_size, group, overlap=overlap)
    if grad_clip is not None:
        clip_grad_norm_flat([arena.flat_grads], grad_clip)
    optimizer.step()
    return total, losses

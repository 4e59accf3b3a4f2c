"""world_size-2 gloo tests (CPU) of the data-parallel gradient exchange."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import mr3d_amd
        from mr3d_amd.dist_utils import (allreduce_flat_grads, allreduce_grads, broadcast_params, FlatArena,
                                         _allreduce_coalesced)
        torch.manual_seed(100 + rank)
        # 1) flat arena: every rank ends with the exact fp32 mean computed in rank order
        flat = torch.randn(10007)
        mine = flat.clone()
        allreduce_flat_grads(flat, world)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        expect = gathered[0].clone()
        for g in gathered[1:]:
            expect += g
        expect /= world
        ok1 = torch.equal(flat, expect)
        # 2) reference-signature allreduce_grads on a model whose grads are re-homed into a FlatArena
        model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        arena = FlatArena(model.parameters())
        broadcast_params(arena.flat_params)
        w0 = [torch.empty_like(arena.flat_params) for _ in range(world)]
        dist.all_gather(w0, arena.flat_params)
        ok2 = all(torch.equal(w0[0], w) for w in w0)
        x = torch.randn(4, 7)
        model(x).pow(2).sum().backward()
        local = arena.flat_grads.clone()
        allreduce_grads(model)
        g2 = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(g2, local)
        ok3 = torch.allclose(arena.flat_grads, sum(g2) / world, rtol=0, atol=1e-7)
        ok3 = ok3 and all(p.grad.data_ptr() >= arena.flat_grads.data_ptr() for p in model.parameters())
        # 3) non-arena path (bucketed copy-back) and the non-coalesced path agree with the mean too
        ts = [torch.full((3,), float(rank + 1)), torch.full((2, 2), float(10 * (rank + 1)))]
        _allreduce_coalesced(ts, world, bucket_size_mb=1)
        ok4 = torch.allclose(ts[0], torch.full((3,), (1 + world) / 2)) and torch.allclose(ts[1], torch.full((2, 2), 10 * (1 + world) / 2))
        m2 = torch.nn.Linear(3, 2)
        for p in m2.parameters():
            p.grad = torch.full_like(p, float(rank))
        allreduce_grads(m2, coalesce=False)
        ok5 = all(torch.allclose(p.grad, torch.full_like(p, (world - 1) / 2)) for p in m2.parameters())
        q.put((rank, ok1, ok2, ok3, ok4, ok5))
    finally:
        dist.destroy_process_group()


def test_allreduce_paths_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_clip_grad_norm_flat_matches_torch():
    from mr3d_amd.dist_utils import clip_grad_norm_flat
    torch.manual_seed(0)
    a, b = torch.randn(1000) * 3, torch.randn(77) * 3
    pa, pb = torch.nn.Parameter(torch.zeros(1000)), torch.nn.Parameter(torch.zeros(77))
    pa.grad, pb.grad = a.clone(), b.clone()
    n_ref = torch.nn.utils.clip_grad_norm_([pa, pb], max_norm=35, norm_type=2)
    n = clip_grad_norm_flat([a, b], 35)
    assert torch.allclose(n, n_ref) and torch.allclose(a, pa.grad, atol=1e-6) and torch.allclose(b, pb.grad, atol=1e-6)


class _StubDetector(torch.nn.Module):
    """CPU stand-in with the detector's training interface (forward_train -> loss dict, total_loss)"""

    def __init__(self):
        super().__init__()
        self.backbone = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 8))
        self.rpn_head = torch.nn.Linear(8, 1)
        self.bbox_head = torch.nn.Linear(8, 6)

    def forward_train(self, x, cls_t, reg_t):
        f = self.backbone(x)
        return dict(loss_rpn_cls=[torch.nn.functional.binary_cross_entropy_with_logits(self.rpn_head(f), cls_t)],
                    loss_reg=torch.nn.functional.smooth_l1_loss(self.bbox_head(f), reg_t), acc=torch.tensor(50.0))

    @staticmethod
    def total_loss(losses):
        from mr3d_amd.detector.two_stage import TwoStageDetector3D
        return TwoStageDetector3D.total_loss(losses)


def _train_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import mr3d_amd
        from mr3d_amd.dist_utils import DetectorArena, broadcast_params, train_step
        torch.manual_seed(7 + rank)                      # different initial weights per rank until the broadcast
        det = _StubDetector()
        arena = DetectorArena(det, engine_attr=None)
        broadcast_params(arena.flat_params)
        opt = torch.optim.SGD([arena.flat_params], lr=0.05, momentum=0.9, weight_decay=1e-4)
        one_storage = all(p.data.untyped_storage().data_ptr() == arena.storage.untyped_storage().data_ptr() and
                          p.grad.untyped_storage().data_ptr() == arena.grad_storage.untyped_storage().data_ptr()
                          for p in det.parameters())
        gen = torch.Generator().manual_seed(1000 + rank)  # every rank sees its own samples (DistributedSampler)
        w_start = arena.flat_params.clone()
        local_grads = []
        for it in range(3):
            x = torch.randn(5, 6, generator=gen)
            batch = ((x, (torch.rand(5, 1, generator=gen) > 0.5).float(), torch.randn(5, 6, generator=gen)), {})
            if it == 1:
                det.zero_grad()                          # torch's set_to_none=True default: grads must find their way back
            total, losses = train_step(det, arena, opt, batch, world, grad_clip=35.0)
            local_grads.append(arena.flat_grads.clone())
        still_views = all(p.grad is not None and p.grad.untyped_storage().data_ptr() == arena.grad_storage.untyped_storage().data_ptr()
                          for p in det.parameters())
        ws = [torch.empty_like(arena.flat_params) for _ in range(world)]
        dist.all_gather(ws, arena.flat_params)
        same = all(torch.equal(ws[0], w) for w in ws)
        moved = not torch.equal(arena.flat_params, w_start)
        gs = [torch.empty_like(local_grads[-1]) for _ in range(world)]
        dist.all_gather(gs, local_grads[-1])
        same_g = all(torch.equal(gs[0], g) for g in gs)        # after the exchange every rank holds the same mean gradient
        q.put((rank, one_storage, still_views, same, moved, same_g, bool(torch.isfinite(total))))
    finally:
        dist.destroy_process_group()


def test_train_step_world2_identical_parameters():
    """drive the training step (zero -> forward_train -> backward -> ONE arena all-reduce -> clip -> SGD) on
    two gloo ranks with different data; after 3 steps both ranks hold bit-identical parameters."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def _overlap_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import mr3d_amd  # noqa: F401
        from mr3d_amd.dist_utils import DetectorArena, broadcast_params, allreduce_flat_grads, allreduce_overlapped, clip_grad_norm_flat
        dets, arenas, opts = [], [], []
        for _ in range(3):                               # [serial exchange, split at the stem, split at the early bucket], identical starts
            torch.manual_seed(3)
            d = _StubDetector()
            a = DetectorArena(d, engine_attr=None)
            broadcast_params(a.flat_params)
            dets.append(d); arenas.append(a)
            opts.append(torch.optim.SGD([a.flat_params], lr=0.05, momentum=0.9, weight_decay=1e-4))
        gen = torch.Generator().manual_seed(500 + rank)
        split = 37                                       # any split point: the leading range plays the stem's part
        for it in range(3):
            x = torch.randn(5, 6, generator=gen)
            args = (x, (torch.rand(5, 1, generator=gen) > 0.5).float(), torch.randn(5, 6, generator=gen))
            for k in range(3):
                arenas[k].zero_grad()
                dets[k].total_loss(dets[k].forward_train(*args)).backward()
                if k == 0:
                    allreduce_flat_grads(arenas[k].flat_grads, world)
                elif k == 1:
                    allreduce_overlapped(arenas[k].flat_grads, split, world)
                else:                                    # round 5: [early_elems:] first (mid-backward on the GPU), the front range last
                    allreduce_overlapped(arenas[k].flat_grads, split, world, early_elems=61)
                clip_grad_norm_flat([arenas[k].flat_grads], 35.0)
                opts[k].step()
        same_paths = all(torch.equal(arenas[0].flat_params, arenas[k].flat_params) and torch.equal(arenas[0].flat_grads, arenas[k].flat_grads)
                         for k in (1, 2))
        ws = [torch.empty_like(arenas[1].flat_params) for _ in range(world)]
        dist.all_gather(ws, arenas[1].flat_params)
        q.put((rank, same_paths, all(torch.equal(ws[0], w) for w in ws)))
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_equals_serial_exchange_world2():
    """allreduce_overlapped (two collectives: everything behind the stem's range first, on the GPU underneath the stem weight
    gradient, then the stem's range) leaves bit-identical gradients and, after 3 SGD steps, bit-identical parameters to the
    reference's single all-reduce (dist_utils.py:51-57) -- and identical parameters on both ranks"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_dist_optimizer_step_keeps_arena_grads_attached():
    """DistOptimizerStep must not orphan arena-backed gradients (torch 2.x zero_grad sets them to None)."""
    from mr3d_amd.dist_utils import DistOptimizerStep, FlatArena
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.ReLU(), torch.nn.Linear(3, 2))
    arena = FlatArena(model.parameters())
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    step = DistOptimizerStep(model, opt, grad_clip=dict(max_norm=35, norm_type=2))
    w0 = arena.flat_params.clone()
    for _ in range(2):
        step(model(torch.randn(8, 4)).pow(2).sum())
    assert not torch.equal(arena.flat_params, w0)
    base = arena.flat_grads.untyped_storage().data_ptr()
    assert all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in model.parameters())
    # and the plain torch idiom (set_to_none=True) finds its way back into the arena through the hook
    opt.zero_grad()
    model(torch.randn(8, 4)).pow(2).sum().backward()
    assert all(p.grad.untyped_storage().data_ptr() == base for p in model.parameters())
    assert arena.flat_grads.abs().sum() > 0


class _LateHeadDetector(torch.nn.Module):
    """a detector whose mask head receives no gradient before iteration `first_mask_iter` (no positive roi yet): the
    reference neither exchanges nor steps a parameter whose .grad is None (mmdet/core/utils/dist_utils.py:32-35)"""

    def __init__(self):
        super().__init__()
        self.backbone = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 8))
        self.bbox_head = torch.nn.Linear(8, 6)
        self.mask_head = torch.nn.Linear(8, 4)
        self.never_used = torch.nn.Linear(3, 3)
        self.with_mask = False

    def forward_train(self, x, reg_t, mask_t):
        f = self.backbone(x)
        losses = dict(loss_reg=torch.nn.functional.smooth_l1_loss(self.bbox_head(f), reg_t))
        if self.with_mask:
            losses['loss_mask'] = torch.nn.functional.mse_loss(self.mask_head(f), mask_t)
        return losses

    @staticmethod
    def total_loss(losses):
        return sum(v for k, v in losses.items() if 'loss' in k)


def test_arena_sgd_equals_per_parameter_sgd_with_grads_none():
    """ArenaSGD over the flat arena == torch.optim.SGD over the separate parameters, where parameters that have not yet received a
    gradient have .grad None (skipped: no weight decay, no momentum) and, once they have had one, keep being stepped after an
    in-place zero (torch 1.0's zero_grad, which the reference runs under)."""
    import copy
    import mr3d_amd  # noqa: F401
    from mr3d_amd.dist_utils import ArenaSGD, DetectorArena, train_step
    torch.manual_seed(3)
    ref = _LateHeadDetector()
    det = copy.deepcopy(ref)
    arena = DetectorArena(det, engine_attr=None)
    opt = ArenaSGD(arena, lr=0.05, momentum=0.9, weight_decay=0.01)
    ref_opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=0.01)
    gen = torch.Generator().manual_seed(9)
    for it in range(6):
        ref.with_mask = det.with_mask = it in (2, 3)           # the mask loss exists in iterations 2 and 3 only
        x, reg_t, mask_t = torch.randn(5, 6, generator=gen), torch.randn(5, 6, generator=gen), torch.randn(5, 4, generator=gen)
        # reference order: zero (in place: .grad stays a tensor once it exists), backward, step
        for p in ref.parameters():
            if p.grad is not None:
                p.grad.zero_()
        ref.total_loss(ref.forward_train(x, reg_t, mask_t)).backward()
        ref_opt.step()
        train_step(det, arena, opt, ((x, reg_t, mask_t), {}), world_size=1, grad_clip=None)
        for (n, a), b in zip(ref.named_parameters(), det.parameters()):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (it, n)
        if it < 2:
            assert len(arena.untouched_ranges()) == 4           # mask head + the unused layer: weight and bias each
    # the never-used layer was never decayed; the mask head was decayed from iteration 2 on, also in 4 and 5 (zero gradient)
    assert torch.equal(ref.never_used.weight, det.never_used.weight)
    init = _LateHeadDetector
    torch.manual_seed(3)
    fresh = init()
    assert torch.equal(fresh.never_used.weight, det.never_used.weight)
    assert not torch.equal(fresh.mask_head.weight, det.mask_head.weight)
    assert len(arena.untouched_ranges()) == 2


def test_freeze_gc_after_warmup_is_opt_in_and_freezes_once(monkeypatch):
    """dist_utils.freeze_gc_after_warmup: nothing during the warm-up steps, then ONE full collection + gc.freeze(); train_step calls it
    only with freeze_gc=True (a library call must not change the host's collector on its own)"""
    import gc
    import inspect
    from mr3d_amd import dist_utils
    assert inspect.signature(dist_utils.train_step).parameters['freeze_gc'].default is False
    monkeypatch.setattr(dist_utils, '_gc_state', {'steps': 0, 'frozen': False})
    before = gc.get_freeze_count()
    try:
        assert [dist_utils.freeze_gc_after_warmup(2) for _ in range(5)] == [False, False, True, False, False]
        assert gc.get_freeze_count() > before
    finally:
        gc.unfreeze()


def _checksum_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import mr3d_amd  # noqa: F401
        from mr3d_amd.dist_utils import (allreduce_flat_grads, allreduce_overlapped, grads_checksum, ranks_hold_identical,
                                         time_allreduce_standalone)
        torch.manual_seed(7 + rank)
        flat = torch.randn(40009)
        before, sums0 = ranks_hold_identical(flat)                 # different seeds: different gradients, different checksums
        allreduce_overlapped(flat, 1000, world, early_elems=5000)
        after, sums1 = ranks_hold_identical(flat)                  # the exchange leaves every rank with the same bits
        serial = flat.clone()
        allreduce_flat_grads(serial, world)                        # a mean of equal buffers is the buffer (x + x) / 2 == x
        same_again, _ = ranks_hold_identical(serial)
        bad = flat.clone()
        if rank == 1:
            bad[12345] = torch.nextafter(bad[12345], torch.tensor(float('inf')))     # one ulp on one rank
        caught, _ = ranks_hold_identical(bad)
        swapped = flat.clone()
        if rank == 1:
            swapped[[3, 4]] = swapped[[4, 3]]                      # same multiset of values, other order
        caught_swap, _ = ranks_hold_identical(swapped)
        sec = time_allreduce_standalone(flat.clone(), world, reps=2)
        q.put((rank, before, sums0, after, sums1, same_again, caught, caught_swap, sec, int(grads_checksum(flat))))
    finally:
        dist.destroy_process_group()


def test_post_exchange_checksum_world2():
    """bench.py's N > 1 line: after the exchange every rank must hold the same gradient bits (the reference copies the reduced
    bucket back on every rank, mmdet/core/utils/dist_utils.py:22-28); the all-gathered 64-bit checksum sees a one-ulp
    difference and a swap of two elements"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_checksum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, before, sums0, after, sums1, same_again, caught, caught_swap, sec, mine in res:
        assert before is False and len(set(sums0)) == 2
        assert after is True and len(set(sums1)) == 1 and sums1[rank] == mine
        assert same_again is True
        assert caught is False and caught_swap is False
        assert sec > 0
    assert res[0][4] == res[1][4]                                   # both ranks gathered the same list


def test_allreduce_bus_bandwidth_formula():
    from mr3d_amd.dist_utils import allreduce_bus_bandwidth, grads_checksum
    # 2 (n - 1) / n * bytes / t; the 103.9 MB arena in 1 ms on 8 ranks = 181.8 GB/s = 0.17 of 7 x 153 GB/s
    r = allreduce_bus_bandwidth(103.9e6, 1e-3, 8)
    assert abs(r['busbw_GBps'] - 2 * 7 / 8 * 103.9) < 1e-6
    assert abs(r['frac_of_xgmi'] - r['busbw_GBps'] / (7 * 153.0)) < 1e-12 and abs(r['frac_of_one_link'] - r['busbw_GBps'] / 153.0) < 1e-12
    assert allreduce_bus_bandwidth(1e6, 1e-3, 1)['busbw_GBps'] == 0.0          # one rank moves nothing
    assert allreduce_bus_bandwidth(1e6, 0.0, 8)['busbw_GBps'] is None
    a = torch.arange(1000, dtype=torch.float32)
    assert int(grads_checksum(a)) == int(grads_checksum(a.clone())) != int(grads_checksum(a.flip(0)))

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

#!/usr/bin/env python3
"""One rank of BASELINE.md's C4: mean of a flat fp32 gradient buffer over gloo (CPU), as mmdet/core/utils/dist_utils.py:22-28 does
it -- all_reduce(SUM) then divide by the world size.  RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment
(bench.py starts one child per rank).  usage: gloo_mean_bench.py <numel> <repetitions>; rank 0 prints `ms_per_exchange <ms>`."""
import os
import sys
import time

import torch
import torch.distributed as dist


def main():
    numel, reps = int(sys.argv[1]), int(sys.argv[2])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    buf = torch.full((numel,), float(rank + 1))
    dist.all_reduce(buf)
    buf.div_(world)                                  # warm-up: connection set-up
    assert abs(float(buf[0]) - (world + 1) / 2) < 1e-5
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
        buf.div_(world)
    dist.barrier()
    el = (time.perf_counter() - t0) / reps
    if rank == 0:
        print(f'ms_per_exchange {el * 1e3:.3f}', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

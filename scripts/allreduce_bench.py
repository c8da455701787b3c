#!/usr/bin/env python3
"""All-reduce bus-bandwidth microbench over RCCL/xGMI: run on the multi-GPU node to choose the DDP
bucket size.  busbw = 2*(N-1)/N * bytes / time.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/allreduce_bench.py
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.parallel import comm  # noqa: E402


def main():
    world, rank, lr = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    cuda = torch.cuda.is_available()
    dev = torch.device("cuda", lr) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(lr)
    if world > 1:
        comm.init_process_group(rank, world, dev)
    rows = []
    for mb in (0.25, 1, 4, 8, 16, 25, 32, 64, 128, 256):
        n = int(mb * 2 ** 20 / 4)
        t = torch.ones(n, device=dev)
        for _ in range(5):
            if world > 1:
                dist.all_reduce(t)
        if cuda:
            torch.cuda.synchronize()
        iters = 20
        t0 = time.perf_counter()
        for _ in range(iters):
            if world > 1:
                dist.all_reduce(t)
        if cuda:
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        algbw = n * 4 / dt / 1e9
        rows.append(dict(mbytes=mb, us=round(dt * 1e6, 1), algbw_GBs=round(algbw, 2),
                         busbw_GBs=round(algbw * 2 * (world - 1) / max(world, 1), 2)))
    if rank == 0:
        print(json.dumps({"world": world, "backend": "rccl" if cuda else "gloo", "results": rows}))
    comm.destroy_process_group()


if __name__ == "__main__":
    main()

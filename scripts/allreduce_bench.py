#!/usr/bin/env python3
"""All-reduce bus-bandwidth microbench for choosing the DDP bucket size on an 8 x MI355X node.
Measures the NATIVE communicator (csrc/comm: the path training uses) and, with --also-torch, torch.distributed's RCCL
group on the same sizes.  busbw = 2*(N-1)/N * bytes / time.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/allreduce_bench.py
  ... scripts/allreduce_bench.py --sweep        # also: a ResNet-50-sized gradient (102 MB) cut into buckets of 4..128 MiB,
                                                # all in flight - the time the reducer would see per step and bucket size
  ... scripts/allreduce_bench.py --transport mesh   # the IPC mesh transport (own two-shot kernels over hipIpc peer buffers)
  scripts/comm_first_contact.sh                 # everything above for N = 2, 4, 8 and both transports, one JSON line each

Nothing here has been run with more than one GPU by the builder (one-GPU boxes only): `profiles/r3_allreduce_mesh_shared_gpu.txt`
is N processes SHARING one MI355X (DPA_SHARE_GPU=1: no xGMI involved, it times the kernels + the host handshake).

What to read off it (xGMI is point-to-point, 7 links x ~153 GB/s per GPU): a single ring is bound by ONE link
(~87 GB/s algbw at N=8), the mesh algorithms by all seven (~600 GB/s); the bucket cap should be the smallest size whose
busbw is within ~10 % of the plateau, so that the last bucket's exposed latency stays small.
On a machine without GPUs the TCP transport is measured instead (plumbing check only)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.parallel import comm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--also-torch", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--transport", default=None, choices=["rccl", "mesh", "tcp"],
                    help="default: rccl on GPUs, tcp on CPUs (DPA_COMM_TRANSPORT overrides)")
    ap.add_argument("--max-mb", type=float, default=256)
    a = ap.parse_args()
    world, rank, lr = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    cuda = torch.cuda.is_available()
    if cuda and os.environ.get("DPA_SHARE_GPU", "0") == "1":      # rehearsal: several ranks on one GPU (DPA_COMM_TRANSPORT=tcp)
        lr = lr % torch.cuda.device_count()
    dev = torch.device("cuda", lr) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(lr)
        torch.cuda.set_stream(torch.cuda.Stream())
    nc = comm.init_native(rank, world, dev, create_for_single_rank=True, transport=a.transport,
                          timeout_s=float(os.environ.get("DPA_COLLECTIVE_TIMEOUT_S", "120")))
    pre = comm.preflight(nc, dev, world, rank)        # known-answer all-reduce + broadcast before anything is timed

    def sync():
        if cuda:
            torch.cuda.synchronize()
        else:
            nc.sync()

    def timed(fn):
        for _ in range(3):
            fn()
        sync()
        nc.barrier()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        sync()
        return (time.perf_counter() - t0) / a.iters

    rows = []
    for mb in (0.25, 1, 4, 8, 16, 25, 32, 64, 128, 256):
        if mb > a.max_mb:
            break
        n = int(mb * 2 ** 20 / 4)
        t = torch.ones(n, device=dev)

        def one():
            nc.allreduce_async(t, 0)
            nc.wait(0)
        dt = nc.all_reduce_scalars([timed(one)], "max")[0]
        algbw = n * 4 / dt / 1e9
        rows.append(dict(mbytes=mb, us=round(dt * 1e6, 1), algbw_GBs=round(algbw, 2),
                         busbw_GBs=round(algbw * 2 * (world - 1) / max(world, 1), 2)))
    out = {"world": world, "comm": nc.name, "transport_ranks": pre["ranks"], "preflight_ok": pre["ok"],
           "shared_gpu": os.environ.get("DPA_SHARE_GPU", "0") == "1", "results": rows}
    if a.sweep:
        total = 25_557_032 * 4          # ResNet-50 gradient bytes
        sweep = []
        for cap_mb in (4, 8, 16, 25, 32, 64, 128):
            per = int(cap_mb * 2 ** 20 / 4)
            sizes = [per] * (total // 4 // per) + ([total // 4 % per] if total // 4 % per else [])
            bufs = [torch.ones(s, device=dev) for s in sizes]

            def step():
                for b, buf in enumerate(bufs):
                    nc.allreduce_async(buf, b)
                for b in range(len(bufs)):
                    nc.wait(b)
            dt = nc.all_reduce_scalars([timed(step)], "max")[0]
            sweep.append(dict(bucket_mb=cap_mb, buckets=len(sizes), ms_per_step=round(dt * 1e3, 3),
                              busbw_GBs=round(total / dt / 1e9 * 2 * (world - 1) / max(world, 1), 2)))
            del bufs
        out["resnet50_gradient_sweep"] = sweep
    if a.also_torch and world > 1 and cuda:
        import torch.distributed as dist
        os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")))
        comm.init_process_group(rank, world, dev)
        trows = []
        for mb in (1, 8, 25, 64, 256):
            t = torch.ones(int(mb * 2 ** 20 / 4), device=dev)
            dt = nc.all_reduce_scalars([timed(lambda: dist.all_reduce(t))], "max")[0]
            algbw = t.numel() * 4 / dt / 1e9
            trows.append(dict(mbytes=mb, us=round(dt * 1e6, 1), busbw_GBs=round(algbw * 2 * (world - 1) / world, 2)))
        out["torch_process_group"] = trows
    if rank == 0:
        print(json.dumps(out))
    comm.shutdown()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One rank of a two-rank DDP run that SHARES one MI355X through the IPC mesh transport - started once per rank as an
ordinary program, so that one of the ranks can run under `rocprofv3 --kernel-trace` (no launcher, no exec from a process
that has touched the GPU):

  export HSA_ENABLE_IPC_MODE_LEGACY=0
  python3 scripts/diag/mesh_trace_rank.py 1 2 29733 &                       # rank 1, plain
  rocprofv3 --kernel-trace -d gpurun_out/trace -- python3 scripts/diag/mesh_trace_rank.py 0 2 29733
  python3 scripts/diag/comm_overlap.py gpurun_out/trace/*/*_results.db > profiles/r3_forced_collectives_trace.txt

The kernel trace of rank 0 then shows the communicator's kernels (mesh_reduce_kernel / mesh_gather_kernel on the
transport's stream) between the backward kernels of the compute stream: the overlap the reducer is built for."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.models.factory import get_model  # noqa: E402
from distributedpytorch_amd.nn.loss import Criterion  # noqa: E402
from distributedpytorch_amd.nn.optim import ArenaOptimizer  # noqa: E402
from distributedpytorch_amd.parallel import comm  # noqa: E402
from distributedpytorch_amd.parallel.ddp import DistributedModel  # noqa: E402

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
batch = int(os.environ.get("BATCH", "64"))
steps = int(os.environ.get("STEPS", "6"))
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
dev = torch.device("cuda", 0)
nc = comm.init_native(rank, world, dev, "127.0.0.1", port, timeout_s=120, rendezvous_timeout_s=120, transport="mesh")
pre = comm.preflight(nc, dev, world, rank)
torch.manual_seed(1234)
net, _ = get_model("resnet50", 1000, device=dev, dtype=torch.float32)
dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3), native_comm=nc, activation_arena=True)
g = torch.Generator(device=dev).manual_seed(7 + rank)
x = torch.zeros(batch, 224, 224, net.in_channels, device=dev)
x[..., :3] = torch.randn(batch, 224, 224, 3, device=dev, generator=g)
y = torch.randint(0, 1000, (batch,), device=dev, generator=g)
for _ in range(steps):
    dm.train_step(x, y)
torch.cuda.synchronize()
assert dm.reducer.check_in_sync() == 0.0
if rank == 0:
    print("mesh trace run ok:", pre, "buckets MiB:", [round(b["mbytes"], 2) for b in dm.reducer.describe()], flush=True)
comm.shutdown()

"""Device-side multi-rank behaviour of the native communicator on ONE GPU:

  * the IPC mesh transport (csrc/comm/mesh_backend.hip): 2-4 processes share the device, exchange hipIpc memory handles
    through the C++ store, and run the two-shot all-reduce as our own HIP kernels over each other's buffers - sums must
    be bit-identical on every rank and equal to the rank-order sums the socket transport computes;
  * the RCCL transport's watchdog: a collective that never finishes on the comm stream (test hook: a host function that
    blocks the stream) is diagnosed inside the collective timeout, the stream is released, later calls fail loudly.
RCCL itself refuses two ranks on one device - its world > 1 behaviour can only be observed on the driver's 8-GPU node."""
import os
import subprocess
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mesh_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["DPA_MESH_BUFFER_MB"] = "1"              # 1 MiB staging: the 700 k-float bucket below takes three chunks
    os.environ["DPA_COMM_EVENT_RING"] = "8"
    from distributedpytorch_amd.parallel import comm
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    dev = torch.device("cuda", 0)
    nc = comm.init_native(rank, world, dev, "127.0.0.1", port, timeout_s=60, rendezvous_timeout_s=60, transport="mesh")
    assert nc.name == "mesh_native" and nc.peer_count() == world
    info = comm.preflight(nc, dev, world, rank)
    assert info["ok"] and info["ranks"] == world
    # broadcast of an odd byte count from the last rank
    t = (torch.arange(100_003, dtype=torch.int32, device=dev) * (rank + 1)).to(torch.uint8)
    nc.broadcast(t, root=world - 1)
    want = (torch.arange(100_003, dtype=torch.int32, device=dev) * world).to(torch.uint8)
    assert torch.equal(t, want)
    g = torch.Generator().manual_seed(1000 + rank)
    sizes = [1, 77, 100_003, 700_001]
    results = []
    for it in range(6):
        host = [torch.randn(n, generator=g) for n in sizes]
        bufs = [h.to(dev) for h in host]
        # the producers run on the compute stream right before the collective: the ready event must order them
        for b in bufs:
            b.mul_(2.0)
        for b, buf in enumerate(bufs):
            nc.allreduce_async(buf, b)
        for b in range(len(sizes)):
            nc.wait(b)
        # consumers on the compute stream right after the wait: the done event must order them
        outs = [(b * 0.5).cpu() for b in bufs]
        results.append((host, outs))
    torch.cuda.synchronize()
    torch.save(results, os.path.join(out_dir, f"mesh{rank}.pt"))
    assert nc.async_error() == 0 and nc.inflight() == 0
    issued, _ = nc.progress()
    assert issued >= 6 * 4 + 3
    comm.shutdown()


@pytest.mark.parametrize("world", [2, 4])
def test_mesh_transport_collectives_bit_identical_to_rank_order_sums(tmp_path, world):
    import torch.multiprocessing as mp
    mp.spawn(_mesh_worker, nprocs=world, args=(world, 29650 + world, str(tmp_path)), join=True)
    res = [torch.load(tmp_path / f"mesh{r}.pt") for r in range(world)]
    for it in range(6):
        for b in range(4):
            contributions = [res[r][it][0][b] * 2.0 for r in range(world)]
            expect = contributions[0].clone()
            for c in contributions[1:]:
                expect += c                                 # rank order 0, 1, ..., world-1: the transports' order
            expect *= 0.5
            for r in range(world):
                assert torch.equal(res[r][it][1][b], expect), (it, b, r)


def _mesh_ddp_worker(rank, world, port, out_dir, transport):
    sys.path.insert(0, ROOT)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.parallel import comm
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    nc = comm.init_native(rank, world, torch.device("cuda", 0), "127.0.0.1", port, timeout_s=120, transport=transport)
    torch.manual_seed(100 + rank)                               # wrap-time broadcast must make the ranks identical
    net, _ = get_model("resnet", 10, device="cuda", dtype=torch.bfloat16)
    dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3), first_bucket_mb=0.25,
                          bucket_cap_mb=4.0, native_comm=nc)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.zeros(16, 64, 64, net.in_channels, device="cuda", dtype=torch.bfloat16)
    x[..., :3] = torch.randn(16, 64, 64, 3, device="cuda", generator=g).to(torch.bfloat16)
    y = torch.randint(0, 10, (16,), device="cuda", generator=g)
    grads = []
    for _ in range(3):
        dm.train_step(x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8])
        grads.append(net.store.g.clone().cpu())
    dm.eval_step(x[:8], y[:8])                                  # first eval after training: one more buffer broadcast
    torch.cuda.synchronize()
    assert dm.reducer.check_in_sync() == 0.0
    torch.save({"p": net.store.p.cpu(), "g": grads, "b": net.store.buffer_arena.cpu()}, os.path.join(out_dir, f"{transport}{rank}.pt"))
    comm.shutdown()


def test_ddp_training_over_the_mesh_equals_the_socket_transport(tmp_path):
    """Two ranks train ResNet-18 on ONE GPU, once over the IPC mesh (device-side reduce kernels over peer buffers) and once
    over the staged socket transport (host adds in rank order): parameters, gradients and BatchNorm buffers must agree
    bit for bit between the ranks AND between the transports."""
    import torch.multiprocessing as mp
    for i, transport in enumerate(("mesh", "tcp")):
        mp.spawn(_mesh_ddp_worker, nprocs=2, args=(2, 29670 + 4 * i, str(tmp_path), transport), join=True)
    m0, m1 = torch.load(tmp_path / "mesh0.pt"), torch.load(tmp_path / "mesh1.pt")
    t0 = torch.load(tmp_path / "tcp0.pt")
    assert torch.equal(m0["p"], m1["p"]) and torch.equal(m0["b"], m1["b"])
    for a, b, c in zip(m0["g"], m1["g"], t0["g"]):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(m0["p"], t0["p"]) and torch.equal(m0["b"], t0["b"])


_STALL = r'''
import os, sys, time
sys.path.insert(0, os.environ["DPA_ROOT"])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
from distributedpytorch_amd.parallel import comm
from distributedpytorch_amd.parallel.native_comm import CommError
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
dev = torch.device("cuda", 0)
nc = comm.init_native(0, 1, dev, "127.0.0.1", 29690, timeout_s=2.0, create_for_single_rank=True, exit_grace_s=float(os.environ["GRACE"]))
assert nc.name == "rccl_native" and nc.peer_count() == 1
t = torch.ones(1 << 20, device=dev)
t0 = time.time()
for i in range(2):                  # collectives #1, #2 complete
    nc.allreduce_async(t, i)
    nc.wait(i)
torch.cuda.synchronize()
assert nc.quiesce() and nc.progress() == (2, 2)
nc.allreduce_async(t, 5)            # collective #3: DPA_TEST_STALL_COLLECTIVE=3 parks a blocking host function on the comm stream
nc.wait(5)                          # the compute stream now waits for a collective that will not finish
t.add_(1.0)
torch.cuda.synchronize()            # the trainer's position when a peer is gone: blocked in a device synchronisation ...
took = time.time() - t0             # ... until the watchdog trips, aborts the communicator and releases the stream
assert nc.watchdog_tripped() and nc.async_error() == 110, "watchdog did not trip"
try:
    nc.allreduce_async(t, 6)
    msg = ""
except CommError as e:
    msg = str(e)
print("TOOK", round(took, 2))
print("MSG", msg)
comm.shutdown(clean=False)
'''


def test_rccl_watchdog_diagnoses_a_collective_that_never_finishes(tmp_path):
    path = tmp_path / "stall.py"
    path.write_text(_STALL)
    env = dict(os.environ, DPA_ROOT=ROOT, DPA_TEST_STALL_COLLECTIVE="3", GRACE="-1")
    r = subprocess.run([sys.executable, str(path)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    took = float([ln for ln in r.stdout.splitlines() if ln.startswith("TOOK")][0].split()[1])
    assert 1.5 < took < 8.0, took                    # the 2 s collective timeout, not forever
    msg = [ln for ln in r.stdout.splitlines() if ln.startswith("MSG")][0]
    assert "aborted by its watchdog" in msg and "collective #3 (all-reduce, bucket 5, 4194304 bytes)" in msg, msg
    assert "transport rccl" in msg and "issued #3, finished #2" in msg, msg
    assert "[dpa comm watchdog]" in r.stderr


def _mesh_failure_worker(rank, world, port, out_dir, scenario):
    sys.path.insert(0, ROOT)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    from distributedpytorch_amd.parallel import comm
    from distributedpytorch_amd.parallel.native_comm import CommError
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    dev = torch.device("cuda", 0)
    nc = comm.init_native(rank, world, dev, "127.0.0.1", port, timeout_s=3, rendezvous_timeout_s=60, transport="mesh", exit_grace_s=-1)
    msg, t0 = "", time.time()
    try:
        t = torch.ones(4096, device=dev)
        nc.allreduce_async(t, 0)                 # collective #1: everyone
        nc.wait(0)
        if scenario == "mismatch":
            t = torch.ones(1000 if rank == 0 else 2000, device=dev)      # the ranks disagree about the bucket size
            nc.allreduce_async(t, 0)
            nc.wait(0)
        elif rank != world - 1:                  # "straggler": the last rank never issues collective #2
            nc.allreduce_async(t, 0)
            nc.wait(0)
        else:
            time.sleep(8)
    except CommError as e:
        msg = str(e)
    with open(os.path.join(out_dir, f"mf{rank}.txt"), "w") as f:
        f.write(f"{time.time() - t0:.2f}\n{msg}")
    comm.shutdown(clean=False)


def test_mesh_transport_names_the_rank_that_is_behind(tmp_path):
    """Three ranks on one GPU over the IPC mesh; rank 2 never joins collective #2: ranks 0 and 1 fail inside the 3 s
    collective timeout with an error that names rank 2 (the transport's stage-counter wait or the front's watchdog,
    whichever fires first - both name it)."""
    import torch.multiprocessing as mp
    mp.spawn(_mesh_failure_worker, nprocs=3, args=(3, 29710, str(tmp_path), "straggler"), join=True)
    for r in (0, 1):
        secs, m = open(tmp_path / f"mf{r}.txt").read().split("\n", 1)
        assert float(secs) < 7.5, secs
        assert "rank 2" in m and ("timed out" in m or "BEHIND: [2]" in m or "aborted by rank" in m), m
    msgs = [open(tmp_path / f"mf{r}.txt").read() for r in (0, 1)]
    assert any("waiting for rank 2" in m or "BEHIND: [2]" in m for m in msgs), msgs


def test_mesh_transport_refuses_mismatched_collectives(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_mesh_failure_worker, nprocs=2, args=(2, 29720, str(tmp_path), "mismatch"), join=True)
    msgs = [open(tmp_path / f"mf{r}.txt").read() for r in range(2)]
    assert any("collective mismatch with rank" in m for m in msgs), msgs      # nobody adds buffers of different lengths
    assert all(m.split("\n", 1)[1] for m in msgs), msgs                       # and nobody hangs

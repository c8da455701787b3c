"""The native communicator's C ABI (csrc/comm: comm_api.cpp front + tcp_backend.cpp) with 2-4 real ranks on CPU: the slot
ring / bucket_slot / wait protocol that the RCCL transport shares, the two-shot mesh all-reduce, and the failure modes -
mismatched collectives, a rank that never shows up, a full slot ring (SURVEY.md §5.2, §5.3, §5.8 'Test backend')."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _up(rank, world, port, timeout_s=60, ring=None):
    sys.path.insert(0, ROOT)
    if ring:
        os.environ["DPA_COMM_EVENT_RING"] = str(ring)
    from distributedpytorch_amd.parallel import comm
    torch.set_num_threads(1)
    return comm, comm.init_native(rank, world, torch.device("cpu"), "127.0.0.1", port, timeout_s=timeout_s,
                                  rendezvous_timeout_s=60)


def _collectives_worker(rank, world, port, out_dir):
    comm, nc = _up(rank, world, port, ring=4)          # tiny ring: 10 iterations x 3 buckets wrap it many times
    assert nc.name == "tcp_native" and nc.world == world
    # broadcast (bytes of any dtype)
    t = torch.arange(1000, dtype=torch.int64) * (rank + 1)
    nc.broadcast(t, root=world - 1)
    assert torch.equal(t, torch.arange(1000, dtype=torch.int64) * world)
    # three buckets of awkward sizes in flight at once, waited in order, repeated: sums must be exact and bit-identical
    g = torch.Generator().manual_seed(1000 + rank)
    sizes = [1, 77, 100_003]
    results = []
    for it in range(10):
        bufs = [torch.randn(n, generator=g) for n in sizes]
        mine = [b.clone() for b in bufs]
        for b, buf in enumerate(bufs):
            nc.allreduce_async(buf, b)
        assert nc.inflight() == 3
        for b in range(3):
            nc.wait(b)
        assert nc.inflight() == 0
        results.append((mine, bufs))
    torch.save(results, os.path.join(out_dir, f"coll{rank}.pt"))
    # protocol errors raised by the front, before anything is sent
    from distributedpytorch_amd.parallel.native_comm import CommError
    a, b = torch.ones(8), torch.ones(8)
    nc.allreduce_async(a, 0)
    with pytest.raises(CommError, match="already has a collective in flight"):
        nc.allreduce_async(b, 0)
    nc.wait(0)
    live = [torch.ones(8) for _ in range(5)]
    for i in range(4):
        nc.allreduce_async(live[i], i)
    with pytest.raises(CommError, match="slot ring exhausted"):
        nc.allreduce_async(live[4], 4)
    for i in range(4):
        nc.wait(i)
    assert all(float(v[0]) == world for v in live[:4])
    # control plane
    assert nc.all_reduce_scalars([rank, 1.0], "sum") == [sum(range(world)), float(world)]
    assert nc.all_reduce_scalars([rank], "max") == [world - 1.0]
    big = bytes([rank]) * 200_000                       # larger than the store client's first buffer
    assert nc.all_gather_bytes(big) == [bytes([r]) * 200_000 for r in range(world)]
    comm.shutdown()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_tcp_transport_collectives_and_protocol(tmp_path, world):
    mp.spawn(_collectives_worker, nprocs=world, args=(world, _free_port(), str(tmp_path)), join=True)
    res = [torch.load(tmp_path / f"coll{r}.pt") for r in range(world)]
    for it in range(10):
        for b in range(3):
            contributions = [res[r][it][0][b] for r in range(world)]
            expect = contributions[0].clone()
            for c in contributions[1:]:
                expect += c                                 # rank order 0, 1, ..., world-1: the transport's order
            for r in range(world):
                assert torch.equal(res[r][it][1][b], expect), (it, b, r)      # exact, and identical on every rank


def _mismatch_worker(rank, world, port, out_dir):
    comm, nc = _up(rank, world, port, timeout_s=20)
    from distributedpytorch_amd.parallel.native_comm import CommError
    t = torch.ones(1000 if rank == 0 else 2000)         # the ranks disagree about the bucket size
    msg = ""
    try:
        nc.allreduce_async(t, 0)
        nc.wait(0)
    except CommError as e:
        msg = str(e)
    with open(os.path.join(out_dir, f"mm{rank}.txt"), "w") as f:
        f.write(msg)
    assert nc.async_error() != 0
    comm.shutdown(clean=False)


def test_mismatched_collective_fails_loudly(tmp_path):
    mp.spawn(_mismatch_worker, nprocs=2, args=(2, _free_port(), str(tmp_path)), join=True)
    msgs = [open(tmp_path / f"mm{r}.txt").read() for r in range(2)]
    assert any("collective mismatch with rank" in m for m in msgs), msgs
    assert all(m for m in msgs), msgs                   # nobody hangs, nobody adds buffers of different lengths


def _straggler_worker(rank, world, port, out_dir):
    comm, nc = _up(rank, world, port, timeout_s=3)
    from distributedpytorch_amd.parallel.native_comm import CommError
    msg = ""
    if rank != 2:                                       # rank 2 never joins the collective
        try:
            nc.allreduce_async(torch.ones(4096), 0)
            nc.wait(0)
        except CommError as e:
            msg = str(e)
    else:
        import time
        time.sleep(6)
    with open(os.path.join(out_dir, f"st{rank}.txt"), "w") as f:
        f.write(msg)
    comm.shutdown(clean=False)


def test_collective_timeout_names_the_missing_rank(tmp_path):
    mp.spawn(_straggler_worker, nprocs=3, args=(3, _free_port(), str(tmp_path)), join=True)
    for r in (0, 1):
        m = open(tmp_path / f"st{r}.txt").read()
        assert "timed out" in m and "rank 2" in m, m


def _absent_rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from distributedpytorch_amd.parallel import comm
    from distributedpytorch_amd.parallel.store import StoreTimeout
    msg = ""
    try:
        comm.init_native(rank, world, torch.device("cpu"), "127.0.0.1", port, timeout_s=5, rendezvous_timeout_s=2)
    except StoreTimeout as e:
        msg = str(e)
    with open(os.path.join(out_dir, f"ab{rank}.txt"), "w") as f:
        f.write(msg)


def test_rendezvous_names_ranks_that_never_arrive(tmp_path):
    """World size 3, only ranks 0 and 1 start: the bring-up fails within the rendezvous timeout and says who is missing
    (the reference blocks in init_process_group forever, README.md:47-49 of the reference)."""
    mp.spawn(_absent_rank_worker, nprocs=2, args=(3, _free_port(), str(tmp_path)), join=True)
    for r in (0, 1):
        assert "missing ranks [2]" in open(tmp_path / f"ab{r}.txt").read()


def test_single_rank_communicator_is_identity():
    sys.path.insert(0, ROOT)
    from distributedpytorch_amd.parallel.native_comm import NativeComm
    nc = NativeComm(None, 0, 1, torch.device("cpu"))
    t = torch.randn(1000)
    ref = t.clone()
    nc.allreduce_async(t, 3)
    nc.wait(3)
    nc.broadcast(t, 0)
    assert torch.equal(t, ref) and nc.all_reduce_scalars([2.5], "max") == [2.5]
    nc.destroy()

"""The communicator's watchdog (csrc/comm/comm_api.cpp) with real ranks on CPU.  RCCL and the IPC mesh
have no notion of time: the watchdog is the ONLY bound on a stuck collective there.  The socket transport's own
per-transfer timeout is pushed out of the way (DPA_TCP_SOCKET_TIMEOUT_MS) so that these tests see the same code path the
GPU transports rely on: progress records in the store, a named-rank diagnosis, abort, and the process-ending backstop.
The reference simply blocks (/root/reference/README.md:47-49, classif.py:86-87)."""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _up(rank, world, port, timeout_s, addr="127.0.0.1", exit_grace_s=-1):
    sys.path.insert(0, ROOT)
    os.environ["DPA_TCP_SOCKET_TIMEOUT_MS"] = "60000"
    from distributedpytorch_amd.parallel import comm
    torch.set_num_threads(1)
    return comm, comm.init_native(rank, world, torch.device("cpu"), addr, port, timeout_s=timeout_s, rendezvous_timeout_s=60,
                                  exit_grace_s=exit_grace_s)


def _behind_worker(rank, world, port, out_dir):
    comm, nc = _up(rank, world, port, timeout_s=3)
    from distributedpytorch_amd.parallel.native_comm import CommError
    msg, t0 = "", time.time()
    for it in range(2):                                  # two collectives everyone takes part in
        t = torch.ones(1000)
        nc.allreduce_async(t, 0)
        nc.wait(0)
    if rank != 2:                                        # rank 2 never issues collective #3
        try:
            nc.allreduce_async(torch.ones(4096), 0)
            nc.wait(0)
        except CommError as e:
            msg = str(e)
        assert nc.watchdog_tripped() and nc.async_error() == 110
        try:                                             # the communicator is dead: every later call fails with the diagnosis
            nc.allreduce_async(torch.ones(8), 1)
            later = ""
        except CommError as e:
            later = str(e)
        assert "aborted by its watchdog" in later, later
    else:
        time.sleep(7)
    with open(os.path.join(out_dir, f"wd{rank}.txt"), "w") as f:
        f.write(f"{time.time() - t0:.2f}\n{msg}")
    comm.shutdown(clean=False)


def test_watchdog_names_the_rank_that_is_behind(tmp_path):
    mp.spawn(_behind_worker, nprocs=3, args=(3, _free_port(), str(tmp_path)), join=True)
    for r in (0, 1):
        secs, m = open(tmp_path / f"wd{r}.txt").read().split("\n", 1)
        assert float(secs) < 6.5, secs                                          # bounded by the collective timeout, not by luck
        assert "collective #3 (all-reduce, bucket 0, 16384 bytes)" in m, m      # WHICH collective
        assert "ranks BEHIND: [2]" in m and "rank 2: has only issued #2" in m, m   # WHO
        other = 1 - r
        assert f"rank {other}: issued #3" in m, m                               # and who is not to blame


_BLOCKED = r'''
import os, sys, time
sys.path.insert(0, os.environ["DPA_ROOT"])
os.environ["DPA_TCP_SOCKET_TIMEOUT_MS"] = "60000"
import torch
from distributedpytorch_amd.parallel.native_comm import NativeComm
# one rank, a collective that hangs INSIDE the transport (test hook), and a caller that never checks anything: it just
# blocks, the way a trainer blocks in a device synchronisation when a peer is gone
nc = NativeComm(None, 0, 1, torch.device("cpu"), timeout_s=1.0, exit_grace_s=1.0)
t = torch.ones(100)
nc.allreduce_async(t, 0)            # collective #1: fine
nc.wait(0)
nc.allreduce_async(t, 0)            # collective #2: DPA_TEST_STALL_COLLECTIVE=2
time.sleep(60)                      # "blocked"; never reaches the next line in time
print("NOT REACHED")
'''


def test_watchdog_ends_a_process_that_stays_blocked(tmp_path):
    path = tmp_path / "blocked.py"
    path.write_text(_BLOCKED)
    env = dict(os.environ, DPA_ROOT=ROOT, DPA_TEST_STALL_COLLECTIVE="2")
    t0 = time.time()
    r = subprocess.run([sys.executable, str(path)], env=env, capture_output=True, text=True, timeout=50)
    took = time.time() - t0
    assert r.returncode == 110, (r.returncode, r.stderr[-1500:])
    assert "NOT REACHED" not in r.stdout
    assert "collective #2 (all-reduce, bucket 0, 400 bytes)" in r.stderr, r.stderr[-1500:]
    assert "ending the process (exit 110)" in r.stderr
    assert took < 30, took              # timeout 1 s + grace 1 s + interpreter start-up; nowhere near the 60 s sleep


def _stall_inside_worker(rank, world, port, out_dir):
    if rank == 1:
        os.environ["DPA_TEST_STALL_COLLECTIVE"] = "2"      # rank 1's transport swallows collective #2 after it was issued
    comm, nc = _up(rank, world, port, timeout_s=2)
    from distributedpytorch_amd.parallel.native_comm import CommError
    msg = ""
    try:
        for it in range(2):
            t = torch.ones(64)
            nc.allreduce_async(t, 0)
            nc.wait(0)
    except CommError as e:
        msg = str(e)
    with open(os.path.join(out_dir, f"si{rank}.txt"), "w") as f:
        f.write(msg)
    comm.shutdown(clean=False)


def test_watchdog_says_when_every_rank_issued_the_collective(tmp_path):
    """Both ranks ISSUED collective #2; it hangs inside one rank's transport.  Nobody is behind - the diagnosis must say so
    (on RCCL: a link / firmware problem rather than a straggling rank)."""
    mp.spawn(_stall_inside_worker, nprocs=2, args=(2, _free_port(), str(tmp_path)), join=True)
    msgs = [open(tmp_path / f"si{r}.txt").read() for r in range(2)]
    assert all(msgs), msgs             # nobody hangs (the rank that trips second may see the first one's abort as a closed socket)
    assert any("collective #2" in m and "every rank has issued it: the stall is inside the transport" in m for m in msgs), msgs


def _hostname_worker(rank, world, port, out_dir):
    comm, nc = _up(rank, world, port, timeout_s=20, addr="localhost")       # a NAME, as torchrun's rendezvous may export
    t = torch.full((10,), float(rank + 1))
    nc.allreduce_async(t, 0)
    nc.wait(0)
    assert float(t[0]) == 3.0
    # the control plane must not grow with the job: many exchanges, at most a handful of keys left in the store
    for _ in range(50):
        nc.barrier()
        nc.all_reduce_scalars([1.0], "sum")
    n = nc.store.num_keys()
    with open(os.path.join(out_dir, f"hn{rank}.txt"), "w") as f:
        f.write(str(n))
    comm.shutdown()


def test_master_addr_may_be_a_host_name_and_control_keys_are_deleted(tmp_path):
    mp.spawn(_hostname_worker, nprocs=2, args=(2, _free_port(), str(tmp_path)), join=True)
    for r in range(2):
        n = int(open(tmp_path / f"hn{r}.txt").read())
        assert 0 < n < 20, n            # join/announce keys, heartbeats, addresses, <= 2 live control keys per rank - not 200+


def test_unresolvable_master_addr_fails_at_once():
    sys.path.insert(0, ROOT)
    from distributedpytorch_amd.parallel.store import TCPStore
    t0 = time.time()
    try:
        TCPStore("no-such-host.invalid", 1, is_master=False, timeout_s=30)
        raised = ""
    except RuntimeError as e:
        raised = str(e)
    assert "cannot resolve" in raised and time.time() - t0 < 10


_BRINGUP = r'''
import os, sys, time
sys.path.insert(0, os.environ["DPA_ROOT"])
import torch
from distributedpytorch_amd.parallel.store import TCPStore
from distributedpytorch_amd.parallel.native_comm import _BringupGuard
rank, port = int(sys.argv[1]), int(sys.argv[2])
store = TCPStore("127.0.0.1", port, is_master=(rank == 0), timeout_s=30)
store.announce_and_wait(rank, 2, tag="join")
with _BringupGuard("ncclCommInitRank", store, rank, 2, 1.5, "up/1"):
    if rank == 1:
        time.sleep(60)          # "blocked inside the transport's bring-up": the peer is there, this call never returns
time.sleep(6)                   # rank 0: finished its bring-up, stays around (it hosts the store)
print("RANK0 DONE")
'''


def test_transport_bringup_is_bounded_and_says_who_finished(tmp_path):
    """ncclCommInitRank blocks until every rank has called it and has no timeout of its own; the collective watchdog does not
    exist yet at that point.  The bring-up guard ends a process stuck there with status 111 and names the ranks whose
    bring-up has / has not finished."""
    path = tmp_path / "bringup.py"
    path.write_text(_BRINGUP)
    port = _free_port()
    env = dict(os.environ, DPA_ROOT=ROOT)
    procs = [subprocess.Popen([sys.executable, str(path), str(r), str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    t0 = time.time()
    outs = [p.communicate(timeout=60) for p in procs]
    assert procs[1].returncode == 111, (procs[1].returncode, outs[1][1][-800:])
    assert time.time() - t0 < 30
    assert "ncclCommInitRank on rank 1 of 2 did not return within 2 s" in outs[1][1] or "did not return within 1 s" in outs[1][1], outs[1][1]
    assert "finished: [0], not finished: [1]" in outs[1][1], outs[1][1]
    assert procs[0].returncode == 0 and "RANK0 DONE" in outs[0][0]

"""Process-group bring-up (X1 of SURVEY.md §2.6; /root/reference/classif.py:86-87): one process per GPU,
`env://` rendezvous on MASTER_ADDR/MASTER_PORT, backend "nccl" (= RCCL over xGMI on ROCm) on GPUs and "gloo" on
CPUs (the plumbing config of BASELINE.json).  Adds what the reference lacks: bounded rendezvous with named missing
ranks, collective timeouts, destroy at exit (Q17)."""
from __future__ import annotations

import datetime
import logging
import os

import torch
import torch.distributed as dist


def warm_kernel_library(device):
    """Load the gfx950 code object (one trivial launch + sync) before any RCCL communicator exists.  Defensive only:
    round 1 suspected that a module first loaded AFTER ncclCommInitRank caused its intermittent hang, but the round-2
    experiment (scripts/diag/rccl_module_order.py, docs/COMM.md) ran both orders, at normal and at high comm-stream
    priority, without a hang - the order is not the cause.  Kept because it costs one launch and makes the first
    training step's latency independent of the code-object load."""
    from .. import ops
    warm = torch.zeros(2, dtype=torch.int64, device=device)
    ops.rng_advance(warm)
    torch.cuda.synchronize(device)


def init_process_group(rank, world_size, device, timeout_s=600.0, master_addr=None, master_port=None):
    """Initialise torch.distributed for this rank.  device: torch.device('cuda', i) or cpu."""
    if master_addr:
        os.environ["MASTER_ADDR"] = str(master_addr)
    if master_port:
        os.environ["MASTER_PORT"] = str(master_port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver stack
    backend = "nccl" if device.type == "cuda" else "gloo"
    # DPA_DIST_BACKEND=gloo: rehearse multi-rank runs with several processes sharing ONE GPU (RCCL refuses two ranks
    # on one device; gloo stages CUDA tensors through the host) - plumbing check only, not a performance path
    backend = os.environ.get("DPA_DIST_BACKEND", backend)
    if device.type == "cuda":
        warm_kernel_library(device)
    kwargs = dict(backend=backend, init_method="env://", world_size=world_size, rank=rank,
                  timeout=datetime.timedelta(seconds=timeout_s))
    if device.type == "cuda" and backend == "nccl":
        kwargs["device_id"] = device
    try:
        dist.init_process_group(**kwargs)
    except TypeError:                       # older signature without device_id
        kwargs.pop("device_id", None)
        dist.init_process_group(**kwargs)
    logging.debug(f"process group up: backend={backend} rank={rank}/{world_size}")
    return backend


def destroy_process_group():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# The native path (default): ONE communicator per rank (parallel/native_comm.py), rendezvous and control plane on the
# C++ TCP store - no torch.distributed process group.  COMM_BACKEND='torch' keeps the process-group path as fallback.
_NATIVE = None
_STORE = None
STORE_PORT_OFFSET = 2          # the store listens on MASTER_PORT + 2 (MASTER_PORT itself belongs to torchrun's agent
                               # when bench.py runs under torch.distributed.run; + 1 is the torch path's arrival check)


def init_native(rank, world_size, device, master_addr=None, master_port=None, timeout_s=600.0, rendezvous_timeout_s=300.0,
                transport=None, create_for_single_rank=False, watchdog=None, exit_grace_s=None, bringup_timeout_s=None):
    """Bring up the native communicator.  Returns it (None at world size 1 unless `create_for_single_rank`)."""
    global _NATIVE, _STORE
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver stack
    addr = str(master_addr or os.environ.get("MASTER_ADDR", "127.0.0.1"))
    port = int(master_port or os.environ.get("MASTER_PORT", "6779")) + STORE_PORT_OFFSET
    if world_size == 1 and not create_for_single_rank:
        return None
    from .native_comm import NativeComm
    store = None
    if world_size > 1:
        from .store import TCPStore
        store = TCPStore(addr, port, is_master=(rank == 0), timeout_s=rendezvous_timeout_s)
        # every rank announces itself; on timeout the error NAMES the ranks that never arrived (the reference blocks
        # forever in init_process_group, README.md:47-49)
        store.announce_and_wait(rank, world_size, tag="join")
        store.timeout_s = timeout_s
    nc = NativeComm(store, rank, world_size, device, transport=transport, timeout_s=timeout_s, master_addr=addr,
                    watchdog=watchdog, exit_grace_s=exit_grace_s, bringup_timeout_s=bringup_timeout_s)
    _NATIVE, _STORE = nc, store
    logging.debug(f"native communicator up: {nc.name} rank={rank}/{world_size}")
    return nc


def native():
    return _NATIVE


def preflight(nc, device, world, rank):
    """Known-answer check of the data plane, on whichever backend is up (nc: the NativeComm, or None for the torch
    process group).  Raises on a wrong answer; returns what bench.py records next to its number:
    ranks = how many ranks the TRANSPORT connects (RCCL: ncclCommCount), the sums seen, the time it took."""
    import time
    t0 = time.perf_counter()
    n = 4096 + 3                                        # not a multiple of anything the transports chunk by
    a = torch.full((n,), float(rank + 1), device=device)
    b = (torch.arange(1024, dtype=torch.int64, device=device) * (rank + 7)).contiguous()
    root = world - 1
    if nc is not None:
        nc.allreduce_async(a, 0)
        nc.wait(0)
        nc.broadcast(b, root)
        ranks = nc.peer_count()
    else:
        dist.all_reduce(a)
        dist.broadcast(b, root)
        ranks = dist.get_world_size()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    want = world * (world + 1) / 2.0
    lo, hi = float(a.min()), float(a.max())
    bc_ok = bool(torch.equal(b, torch.arange(1024, dtype=torch.int64, device=device) * (root + 7)))
    ok = lo == want and hi == want and bc_ok and ranks == world
    info = dict(ok=ok, ranks=int(ranks), expected_ranks=world, allreduce_min=lo, allreduce_max=hi, allreduce_expected=want,
                broadcast_ok=bc_ok, ms=round(1000 * (time.perf_counter() - t0), 2))
    if not ok:
        raise RuntimeError(f"communication preflight failed on rank {rank}: {info}")
    return info


def shutdown(clean=True):
    """Tear down whichever backend is up (Q17: the reference never calls destroy_process_group).  clean=False (an
    exception is propagating): no farewell barrier - the other ranks may be stuck in a collective this rank left."""
    global _NATIVE, _STORE
    if _NATIVE is not None:
        try:
            if _NATIVE.async_error() != 0 or not clean:
                _NATIVE.abort()
            elif _STORE is not None:
                _NATIVE.barrier("shutdown")        # nobody closes the store while others still use it
        except Exception:       # noqa: BLE001 - shutdown path
            pass
        try:
            _NATIVE.destroy()
        except Exception:       # noqa: BLE001
            pass
        _NATIVE = None
    if _STORE is not None:
        try:
            _STORE.close()
        except Exception:       # noqa: BLE001
            pass
        _STORE = None
    destroy_process_group()


def is_distributed():
    return world_size() > 1


def world_size():
    if _NATIVE is not None:
        return _NATIVE.world
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def get_rank():
    if _NATIVE is not None:
        return _NATIVE.rank
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def all_reduce_scalars(values, op="sum", device=None):
    """Reduce a few Python floats over all ranks (metrics, timing, checksums) on whichever backend is up."""
    if _NATIVE is not None:
        return _NATIVE.all_reduce_scalars(values, op)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(v) for v in values], dtype=torch.float64,
                         device=device if device is not None and dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
        return t.tolist()
    return [float(v) for v in values]


def barrier():
    if _NATIVE is not None:
        _NATIVE.barrier()
    elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

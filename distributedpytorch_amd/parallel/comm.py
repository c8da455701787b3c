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
    """Load the gfx950 code object (one trivial launch + sync) before any RCCL communicator exists: on the ROCm 7.0
    runtime a module first loaded AFTER ncclCommInitRank deadlocked later cross-stream event waits (measured with the
    native communicator, scripts/debug/dbg_rccl5.py vs dbg_rccl6.py); loading it first costs ~ms once."""
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


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def get_rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0

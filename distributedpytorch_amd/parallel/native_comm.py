"""Python face of the native communicator (csrc/comm: comm_api.cpp front, rccl_backend.cpp, tcp_backend.cpp) - what
`init_process_group('nccl')` + ProcessGroupNCCL are for the reference (/root/reference/classif.py:86-87), owned here:

  * rendezvous and every control-plane exchange (layout hash, parameter checksums, barriers, the bench's MAX-reduce) go
    through the C++ TCP store (csrc/host/tcpstore.cpp) - no torch.distributed process group exists in this mode;
  * the data plane (bucketed gradient all-reduce, parameter / BatchNorm-buffer broadcasts) goes through ONE communicator
    per rank: RCCL over xGMI on GPUs, the TCP mesh transport on CPUs (and, opt-in, for several ranks sharing one GPU).
"""
from __future__ import annotations

import ctypes
import os
import socket
import struct

import torch

from ..ops import native

_LIB = None


def comm_lib():
    """libdpamd_comm.so: native communicator (RCCL + TCP transports) and the gradient reducer core (csrc/comm/*.cpp)."""
    global _LIB
    if _LIB is None:
        path = native.lib_path("libdpamd_comm.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m distributedpytorch_amd.csrc.build`")
        lib = ctypes.CDLL(path)
        lib.dpa_comm_init.restype = ctypes.c_void_p
        lib.dpa_comm_init_tcp.restype = ctypes.c_void_p
        lib.dpa_comm_last_error.restype = ctypes.c_char_p
        lib.dpa_comm_create_error.restype = ctypes.c_char_p
        lib.dpa_comm_backend_name.restype = ctypes.c_char_p
        lib.dpa_reducer_create.restype = ctypes.c_void_p
        lib.dpa_reducer_launched_bytes.restype = ctypes.c_longlong
        _LIB = lib
    return _LIB


class CommError(RuntimeError):
    pass


def _local_addr_towards(host):
    """The address of the interface that routes to `host` (what peers should connect to)."""
    try:
        s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        s.connect((host, 9))
        addr = s.getsockname()[0]
        s.close()
        return addr
    except OSError:
        return "127.0.0.1"


class NativeComm:
    """One communicator per rank.  transport: 'rccl' (GPU default), 'tcp' (CPU default; on a GPU: device buffers staged
    through pinned memory, several ranks may share the device).  `store`: parallel.store.TCPStore shared by all ranks
    (None at world size 1)."""

    def __init__(self, store, rank, world, device, transport=None, timeout_s=600.0, master_addr="127.0.0.1"):
        self.lib = comm_lib()
        self.rank, self.world, self.store = rank, world, store
        self.device = torch.device(device)
        on_gpu = self.device.type == "cuda"
        self.transport = transport or os.environ.get("DPA_COMM_TRANSPORT") or ("rccl" if on_gpu else "tcp")
        if self.transport == "rccl" and not on_gpu:
            raise ValueError("the RCCL transport needs a GPU device")
        if world > 1 and store is None:
            raise ValueError("a rendezvous store is required for world size > 1")
        self._ctl_seq = 0
        dev_index = (self.device.index or 0) if on_gpu else -1
        if on_gpu:
            # Load the kernel library's code object before the communicator exists (defensive; docs/COMM.md: the order
            # was suspected in round 1 and cleared by the round-2 experiment).
            from .comm import warm_kernel_library
            if os.environ.get("DPA_DIAG_SKIP_WARM") != "1":       # (scripts/diag/rccl_module_order.py only)
                warm_kernel_library(self.device)
        if self.transport == "rccl":
            n = self.lib.dpa_comm_unique_id_size()
            buf = ctypes.create_string_buffer(n)
            if rank == 0:
                rc = self.lib.dpa_comm_unique_id(buf)
                if rc:
                    raise CommError(f"ncclGetUniqueId failed rc={rc}")
                if store is not None:
                    store.set("rccl/unique_id", buf.raw)
            else:
                buf.raw = store.get("rccl/unique_id")
            self.h = self.lib.dpa_comm_init(buf, world, rank, dev_index)
            if not self.h:
                raise CommError(f"RCCL communicator bring-up failed: {self.lib.dpa_comm_create_error().decode()}")
        elif self.transport == "tcp":
            self.h = self.lib.dpa_comm_init_tcp(world, rank, dev_index)
            if not self.h:
                raise CommError(f"TCP communicator bring-up failed: {self.lib.dpa_comm_create_error().decode()}")
            self.lib.dpa_comm_set_timeout_ms(ctypes.c_void_p(self.h), int(timeout_s * 1000))
            if world > 1:
                me = f"{_local_addr_towards(master_addr)}:{self.lib.dpa_comm_tcp_port(ctypes.c_void_p(self.h))}"
                store.set(f"tcpcomm/addr{rank}", me.encode())
                peers = ",".join(store.get(f"tcpcomm/addr{r}").decode() for r in range(world))
                self._check(self.lib.dpa_comm_tcp_connect(ctypes.c_void_p(self.h), peers.encode(), int(timeout_s * 1000)),
                            "mesh bring-up")
        else:
            raise ValueError(f"unknown transport {self.transport!r}")
        if store is not None:
            self.barrier("comm_up")

    # ------------------------------------------------------------------ data plane
    def _stream(self):
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        return ctypes.c_void_p(0)

    def _check(self, rc, what):
        if rc:
            msg = self.lib.dpa_comm_last_error(ctypes.c_void_p(self.h)).decode()
            raise CommError(f"{what} on rank {self.rank}: {msg} (rc={rc})")

    def allreduce_async(self, t: torch.Tensor, bucket: int):
        assert t.dtype == torch.float32 and t.is_contiguous()
        self._check(self.lib.dpa_comm_allreduce_async(ctypes.c_void_p(self.h), ctypes.c_void_p(t.data_ptr()),
                                                      ctypes.c_longlong(t.numel()), bucket, self._stream()), "all-reduce")

    def wait(self, bucket: int):
        self._check(self.lib.dpa_comm_wait(ctypes.c_void_p(self.h), bucket, self._stream()), "wait")

    def broadcast(self, t: torch.Tensor, root=0):
        assert t.is_contiguous()
        self._check(self.lib.dpa_comm_broadcast(ctypes.c_void_p(self.h), ctypes.c_void_p(t.data_ptr()),
                                                ctypes.c_longlong(t.numel() * t.element_size()), root, self._stream()),
                    "broadcast")

    def inflight(self) -> int:
        return int(self.lib.dpa_comm_inflight(ctypes.c_void_p(self.h)))

    def last_error(self) -> str:
        return self.lib.dpa_comm_last_error(ctypes.c_void_p(self.h)).decode()

    def async_error(self) -> int:
        return int(self.lib.dpa_comm_async_error(ctypes.c_void_p(self.h)))

    def sync(self):
        return int(self.lib.dpa_comm_sync(ctypes.c_void_p(self.h)))

    @property
    def name(self):
        return f"{self.transport}_native"

    # ------------------------------------------------------------------ control plane (through the store)
    def all_gather_bytes(self, payload: bytes, tag="ag"):
        """Every rank contributes a small blob; returns the list indexed by rank.  Calls must be made in the same
        order on every rank (a sequence number keys them)."""
        if self.world == 1:
            return [payload]
        self._ctl_seq += 1
        key = f"ctl/{tag}/{self._ctl_seq}"
        self.store.set(f"{key}/{self.rank}", payload)
        return [payload if r == self.rank else self.store.get(f"{key}/{r}") for r in range(self.world)]

    def barrier(self, tag="barrier"):
        self.all_gather_bytes(b"\x01", tag)

    def all_reduce_scalars(self, values, op="sum"):
        """Reduce a short list of Python floats over all ranks (max / min / sum); host-side, for metrics and checks."""
        vals = [float(v) for v in values]
        blobs = self.all_gather_bytes(struct.pack(f"{len(vals)}d", *vals), "scalars")
        rows = [struct.unpack(f"{len(vals)}d", b) for b in blobs]
        f = {"sum": sum, "max": max, "min": min}[op]
        return [f(r[i] for r in rows) for i in range(len(vals))]

    # ------------------------------------------------------------------ teardown
    def abort(self):
        if self.h:
            self.lib.dpa_comm_abort(ctypes.c_void_p(self.h))

    def destroy(self):
        if self.h:
            self.lib.dpa_comm_destroy(ctypes.c_void_p(self.h))
            self.h = None

"""Python face of the native RCCL communicator (csrc/comm/rccl_comm.cpp).  Opt-in: DPA_COMM_BACKEND=rccl_native.
The ncclUniqueId is created by rank 0 and published through the C++ TCP store."""
from __future__ import annotations

import ctypes
import os

import torch

from ..ops import native


_LIB = None


def comm_lib():
    """libdpamd_comm.so: native RCCL communicator + gradient reducer core (csrc/comm/*.cpp)."""
    global _LIB
    if _LIB is None:
        path = native.lib_path("libdpamd_comm.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m distributedpytorch_amd.csrc.build`")
        lib = ctypes.CDLL(path)
        lib.dpa_comm_init.restype = ctypes.c_void_p
        lib.dpa_comm_last_error.restype = ctypes.c_char_p
        lib.dpa_reducer_create.restype = ctypes.c_void_p
        lib.dpa_reducer_launched_bytes.restype = ctypes.c_longlong
        _LIB = lib
    return _LIB


_lib = comm_lib


class NativeRcclComm:
    def __init__(self, store, rank, world, device_index):
        self.lib = _lib()
        self.rank, self.world = rank, world
        # Load the kernel library's code object BEFORE ncclCommInitRank.  Measured on MI355X (ROCm 7.0 runtime):
        # when the gfx950 module is first loaded after the communicator exists, later cross-stream event waits
        # deadlock (scripts/debug/dbg_rccl5.py hangs, dbg_rccl6.py - same steps, comm created after a warm step - passes).
        from .comm import warm_kernel_library
        warm_kernel_library(torch.device("cuda", device_index))
        n = self.lib.dpa_comm_unique_id_size()
        buf = ctypes.create_string_buffer(n)
        if rank == 0:
            rc = self.lib.dpa_comm_unique_id(buf)
            if rc:
                raise RuntimeError(f"ncclGetUniqueId failed rc={rc}")
            if store is not None:
                store.set("rccl/unique_id", buf.raw)
        else:
            buf.raw = store.get("rccl/unique_id")
        self.h = self.lib.dpa_comm_init(buf, world, rank, device_index)
        if not self.h:
            raise RuntimeError("ncclCommInitRank failed")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _check(self, rc, what):
        if rc:
            msg = self.lib.dpa_comm_last_error(ctypes.c_void_p(self.h)).decode()
            raise RuntimeError(f"{what}: {msg} (rc={rc})")

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

    def async_error(self) -> int:
        return int(self.lib.dpa_comm_async_error(ctypes.c_void_p(self.h)))

    def abort(self):
        if self.h:
            self.lib.dpa_comm_abort(ctypes.c_void_p(self.h))

    def destroy(self):
        if self.h:
            self.lib.dpa_comm_destroy(ctypes.c_void_p(self.h))
            self.h = None

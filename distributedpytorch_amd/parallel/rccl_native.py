"""Compatibility alias: the native communicator lives in native_comm.py (RCCL and TCP transports behind one C ABI)."""
from .native_comm import CommError, NativeComm, comm_lib  # noqa: F401

_lib = comm_lib


class NativeRcclComm(NativeComm):
    """NativeComm pinned to the RCCL transport, with round 1's (store, rank, world, device_index) signature."""

    def __init__(self, store, rank, world, device_index):
        import torch
        super().__init__(store, rank, world, torch.device("cuda", device_index), transport="rccl")

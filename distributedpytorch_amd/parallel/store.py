"""Python face of the C++ TCP rendezvous store (csrc/host/tcpstore.cpp): rank arrival check with NAMED missing ranks,
barriers outside the collective library, small blob exchange.  The reference blocks forever in init_process_group
when a node of DDTNodes never shows up (README.md:47-49 of the reference); here the wait is bounded and explicit."""
from __future__ import annotations

import ctypes
import socket
import struct
import time

from ..ops import native


class StoreTimeout(RuntimeError):
    pass


class TCPStore:
    def __init__(self, host, port, is_master, timeout_s=300.0):
        lib = native.host_lib()
        lib.dpa_store_server_start.restype = ctypes.c_void_p
        lib.dpa_store_connect.restype = ctypes.c_void_p
        lib.dpa_store_add.restype = ctypes.c_longlong
        lib.dpa_store_last_size.restype = ctypes.c_longlong
        lib.dpa_store_last_copy.restype = ctypes.c_longlong
        lib.dpa_store_num_keys.restype = ctypes.c_longlong
        self._lib, self._server = lib, None
        self.timeout_s = timeout_s
        # The C++ store speaks dotted IPv4 only: resolve a name (MASTER_ADDR=localhost, or the host name torchrun's
        # multi-node rendezvous exports) HERE, once, and fail at once with a clear message when it cannot be resolved
        # instead of spinning for the whole rendezvous timeout.
        try:
            host = socket.gethostbyname(host)
        except OSError as e:
            raise RuntimeError(f"cannot resolve the rendezvous address MASTER_ADDR={host!r}: {e}") from e
        self.host = host
        if is_master:
            self._server = lib.dpa_store_server_start(host.encode(), int(port))
            if not self._server:
                raise RuntimeError(f"cannot bind rendezvous store on port {port}")
            port = lib.dpa_store_server_port(ctypes.c_void_p(self._server))
        self.port = int(port)
        self._client = lib.dpa_store_connect(host.encode(), self.port, int(timeout_s * 1000))
        if not self._client:
            raise StoreTimeout(f"could not reach the rendezvous store at {host}:{self.port} within {timeout_s}s")

    def new_client(self, timeout_s=10.0):
        """A second raw connection to the same store (for a thread of its own, e.g. the communicator's watchdog).  The
        caller owns the handle and closes it with dpa_store_close."""
        h = self._lib.dpa_store_connect(self.host.encode(), self.port, int(timeout_s * 1000))
        if not h:
            raise StoreTimeout(f"could not open a second connection to the rendezvous store at {self.host}:{self.port}")
        return h

    def set(self, key, value: bytes):
        rc = self._lib.dpa_store_set(ctypes.c_void_p(self._client), key.encode(), value, len(value))
        if rc:
            raise RuntimeError(f"store set({key}) failed rc={rc}")

    def get(self, key, timeout_s=None) -> bytes:
        buf = ctypes.create_string_buffer(1 << 16)
        ms = int((self.timeout_s if timeout_s is None else timeout_s) * 1000)
        n = self._lib.dpa_store_get(ctypes.c_void_p(self._client), key.encode(), buf, len(buf), ctypes.c_longlong(ms))
        if n == -1:
            raise StoreTimeout(f"key {key!r} not set within {ms / 1000:.1f}s")
        if n == -5:         # larger than the first buffer: the client kept it
            size = int(self._lib.dpa_store_last_size(ctypes.c_void_p(self._client)))
            big = ctypes.create_string_buffer(size)
            got = int(self._lib.dpa_store_last_copy(ctypes.c_void_p(self._client), big, ctypes.c_longlong(size)))
            if got != size:
                raise RuntimeError(f"store get({key}) failed while fetching a {size}-byte value")
            return big.raw[:size]
        if n < 0:
            raise RuntimeError(f"store get({key}) failed rc={n}")
        return buf.raw[:n]

    def add(self, key, delta=1) -> int:
        v = self._lib.dpa_store_add(ctypes.c_void_p(self._client), key.encode(), ctypes.c_longlong(delta))
        if v == -(2 ** 63):
            raise RuntimeError(f"store add({key}) failed")
        return int(v)

    def delete(self, key) -> bool:
        return self._lib.dpa_store_delete(ctypes.c_void_p(self._client), key.encode()) == 0

    def num_keys(self) -> int:
        return int(self._lib.dpa_store_num_keys(ctypes.c_void_p(self._client)))

    def check(self, key) -> bool:
        return self._lib.dpa_store_check(ctypes.c_void_p(self._client), key.encode()) == 0

    def announce_and_wait(self, rank, world_size, tag="join", timeout_s=None):
        """Every rank announces itself; on timeout the error lists exactly which ranks never arrived."""
        timeout_s = self.timeout_s if timeout_s is None else timeout_s
        self.set(f"{tag}/rank{rank}", struct.pack("d", time.time()))
        deadline = time.time() + timeout_s
        missing = list(range(world_size))
        while missing:
            missing = [r for r in missing if not self.check(f"{tag}/rank{r}")]
            if not missing:
                break
            if time.time() > deadline:
                raise StoreTimeout(f"rendezvous '{tag}' timed out after {timeout_s:.0f}s: missing ranks {missing} "
                                   f"of world size {world_size}")
            time.sleep(0.05)

    def barrier(self, rank, world_size, tag, timeout_s=None):
        self.announce_and_wait(rank, world_size, f"barrier/{tag}", timeout_s)

    def close(self):
        if self._client:
            self._lib.dpa_store_close(ctypes.c_void_p(self._client))
            self._client = None
        if self._server:
            self._lib.dpa_store_server_stop(ctypes.c_void_p(self._server))
            self._server = None

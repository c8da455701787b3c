"""Planned activation arena.

Every tensor a training / evaluation step obtains through `Context.empty` - activations, gradients, argmax maps, slabs -
has the same shape, is requested at the same point and dies at the same point in every step of the same kind.  So the
first step of a kind is RECORDED (tensors come from torch's allocator, their lifetimes are observed through weak
references on their storages), `csrc/host/arena_planner.cpp` packs the lifetimes into one buffer, and every later step
of that kind is REPLAYED: the k-th request returns a view of the arena at the planned offset.  No allocator call, no
fragmentation (ResNet-50, batch 1024, fp32: arena = the live-set peak, where the caching allocator reserves ~8 GB more),
identical addresses in every step.

Rules that keep it safe:
 * a step is identified by a key (mode, input shape, dtype); a request that does not match the recording (other shape,
   dtype, or one request too many) drops the plan for that key - the rest of the step is served by torch's allocator
   and the next step records again;
 * tensors still alive when the NEXT step begins ("escaping": anything the caller keeps) are never placed in the arena;
 * reuse is stream-ordered like a caching allocator's: memory released between requests k-1 and k may back request k,
   which is written by kernels queued after every kernel that used the released tensor (one compute stream; the
   side-stream wgrad of `DPA_BWD_STREAMS=2` keeps its operands referenced until the streams have joined);
 * hipGraph capture bypasses the arena (a graph has its own static pool).

`DPA_ACT_ARENA=0` disables it; `DPA_ACT_ARENA_VERIFY=1` never replays but re-records every step and raises if a step's
request sequence or lifetimes differ from the plan (the determinism the replay relies on; used by the tests)."""
from __future__ import annotations

import ctypes
import math
import os
import weakref
from typing import Dict, List, Optional

import torch

ALIGN = 256


class _Item:
    __slots__ = ("shape", "dtype", "nbytes", "free_at", "offset", "ref")

    def __init__(self, shape, dtype, nbytes):
        self.shape, self.dtype, self.nbytes = shape, dtype, nbytes
        self.free_at = None        # number of requests that had been made when the storage died (None: still alive)
        self.offset = -1           # byte offset in the arena (-1: not in the arena)
        self.ref = None


class _Plan:
    def __init__(self, items: List[_Item], peak: int, lower_bound: int):
        self.items, self.peak, self.lower_bound = items, peak, lower_bound


def _plan(items: List[_Item]) -> _Plan:
    from ..ops import native
    lib = native.host_lib()
    idx = [i for i, it in enumerate(items) if it.free_at is not None and it.nbytes > 0]
    n = len(idx)
    arr = ctypes.c_longlong * max(n, 1)
    size, start, end, off = arr(), arr(), arr(), arr()
    for k, i in enumerate(idx):
        size[k], start[k], end[k] = items[i].nbytes, i, max(items[i].free_at, i + 1)
    lb = ctypes.c_longlong(0)
    lib.dpa_plan_arena.restype = ctypes.c_longlong
    peak = lib.dpa_plan_arena(n, size, start, end, ctypes.c_longlong(ALIGN), off, ctypes.byref(lb))
    if peak < 0:
        raise RuntimeError("dpa_plan_arena rejected the recorded lifetimes")
    for k, i in enumerate(idx):
        items[i].offset = int(off[k])
    return _Plan(items, int(peak), int(lb.value))


class ActivationArena:
    def __init__(self, device):
        self.device = torch.device(device)
        self.enabled = os.environ.get("DPA_ACT_ARENA", "1") != "0"
        self.verify = os.environ.get("DPA_ACT_ARENA_VERIFY", "0") == "1"
        self.plans: Dict[tuple, _Plan] = {}
        self.buf: Optional[torch.Tensor] = None
        self._mode = None              # None | "record" | "replay"
        self._key = None
        self._items: List[_Item] = []
        self._pos = 0
        self._plan_cur: Optional[_Plan] = None
        self.stats = {"recorded": 0, "replayed": 0, "dropped": 0}

    # ------------------------------------------------------------------ step boundaries
    def begin(self, key):
        """Start of a step of kind `key`.  Finishes the recording of the previous step (its survivors are escaping)."""
        self._finish()
        if not self.enabled or (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            self._mode = None
            return
        self._key, self._pos = key, 0
        plan = self.plans.get(key)
        if plan is None or self.verify:
            self._mode, self._items = "record", []
        else:
            self._mode, self._plan_cur = "replay", plan
            if self.buf is None or self.buf.numel() < plan.peak:
                self.buf = None
                if self.device.type == "cuda":
                    # the recorded step's blocks are still cached by torch's allocator: hand them back first, or the
                    # device holds the recording's footprint AND the arena from now on (one-time, synchronises)
                    torch.cuda.synchronize(self.device)
                    torch.cuda.empty_cache()
                self.buf = torch.empty(plan.peak, device=self.device, dtype=torch.uint8)

    def end(self):
        """Optional: marks the end of the step's requests (requests between steps go to torch's allocator)."""
        if self._mode == "replay":
            if self._pos != len(self._plan_cur.items):
                self._drop("fewer requests than recorded")
            else:
                self.stats["replayed"] += 1
            self._mode = None
        elif self._mode == "record":
            self._mode = "recorded"          # stop recording new requests, keep observing releases until the next begin()

    def _finish(self):
        if self._mode in ("record", "recorded") and self._items:
            items, key = self._items, self._key
            for it in items:
                it.ref = None                # survivors: free_at stays None -> escaping
            new = _plan(items)
            old = self.plans.get(key)
            if self.verify and old is not None:
                same = len(old.items) == len(new.items) and all(
                    a.shape == b.shape and a.dtype == b.dtype and a.free_at == b.free_at for a, b in zip(old.items, new.items))
                if not same:
                    raise RuntimeError(f"activation arena: step {key} is not reproducible (request sequence or lifetimes changed)")
            self.plans[key] = new
            self.stats["recorded"] += 1
        self._items, self._mode = [], None

    def _drop(self, why):
        self.plans.pop(self._key, None)
        self.stats["dropped"] += 1
        self._mode = None
        if os.environ.get("DPA_ACT_ARENA_DEBUG"):
            print(f"[arena] plan for {self._key} dropped: {why}", flush=True)

    # ------------------------------------------------------------------ requests
    def alloc(self, shape, dtype):
        shape = tuple(int(s) for s in shape)
        if self._mode == "record":
            t = torch.empty(shape, device=self.device, dtype=dtype)
            it = _Item(shape, dtype, t.numel() * t.element_size())
            if it.nbytes > 0:
                items = self._items
                it.ref = weakref.ref(t.untyped_storage(), lambda _w, it=it, items=items: self._released(it, items))
            self._items.append(it)
            return t
        if self._mode == "replay":
            k, plan = self._pos, self._plan_cur
            self._pos += 1
            if k >= len(plan.items) or plan.items[k].shape != shape or plan.items[k].dtype != dtype:
                self._drop("request does not match the recording")
                return torch.empty(shape, device=self.device, dtype=dtype)
            it = plan.items[k]
            if it.offset < 0:
                return torch.empty(shape, device=self.device, dtype=dtype)
            return self.buf[it.offset: it.offset + it.nbytes].view(dtype).view(shape)
        return torch.empty(shape, device=self.device, dtype=dtype)

    def _released(self, it, items):
        if items is self._items and it.free_at is None:      # (a release after the next begin() is an escape: ignored)
            it.free_at = len(items)

    # ------------------------------------------------------------------ reporting
    def summary(self):
        out = {}
        for key, p in self.plans.items():
            planned = [it for it in p.items if it.offset >= 0]
            out[str(key)] = {"requests": len(p.items), "in_arena": len(planned), "arena_bytes": p.peak,
                             "live_set_lower_bound": p.lower_bound,
                             "escaping_bytes": sum(it.nbytes for it in p.items if it.offset < 0)}
        return out

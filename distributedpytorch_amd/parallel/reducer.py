"""DDP gradient path (N3/N4/N5, X2-X6 of SURVEY.md §2): what c10d::Reducer + DistributedDataParallel do for the
reference (/root/reference/classif.py:138), rebuilt around the flat gradient arena.

  * gradients already live in ONE arena in registration order, so a bucket is a contiguous slice: no flatten /
    unflatten copies and no per-parameter views to rebuild;
  * buckets are planned once, statically, in backward order by the C++ planner (small first bucket, then a cap
    sized for xGMI: per-peer chunk = bucket/8 should stay >= 1-2 MB);
  * when the last gradient of a bucket is written the bucket's all-reduce is issued asynchronously on RCCL's stream
    (torch.distributed orders it after the producing kernels with an event and returns a Work handle), so it
    overlaps the rest of backward;
  * after backward each bucket is waited for in order and its fused optimizer update is launched immediately, so the
    update of bucket k overlaps the all-reduce of bucket k+1; the 1/world_size averaging is folded into the
    optimizer kernel (no separate divide pass);
  * wrap-time parameter/buffer broadcast from rank 0 and the per-step BatchNorm buffer broadcast reproduce DDP's
    semantics (rank 0's running statistics win).
"""
from __future__ import annotations

import ctypes
import hashlib
import logging
import os

import torch
import torch.distributed as dist

from ..nn.engine import ALIGN, ParamStore, pad_to
from ..ops import native

DEFAULT_FIRST_BUCKET_MB = 1.0
DEFAULT_BUCKET_CAP_MB = 25.0


def plan_buckets(store: ParamStore, first_mb=DEFAULT_FIRST_BUCKET_MB, cap_mb=DEFAULT_BUCKET_CAP_MB):
    """Returns a list of buckets in ready order: dict(lo, hi, params=[spec...])."""
    n = len(store.params)
    sizes = (ctypes.c_longlong * n)(*[pad_to(s.numel, ALIGN) * 4 for s in store.params])
    train = (ctypes.c_int * n)(*[int(s.trainable) for s in store.params])
    out = (ctypes.c_int * n)()
    nb = native.host_lib().dpa_plan_buckets(sizes, train, n, ctypes.c_longlong(int(first_mb * 2 ** 20)),
                                            ctypes.c_longlong(int(cap_mb * 2 ** 20)), out)
    buckets = [dict(lo=None, hi=None, params=[]) for _ in range(nb)]
    for i, spec in enumerate(store.params):
        b = out[i]
        if b < 0:
            continue
        bk = buckets[b]
        lo, hi = spec.offset, spec.offset + pad_to(spec.numel, ALIGN)
        bk["lo"] = lo if bk["lo"] is None else min(bk["lo"], lo)
        bk["hi"] = hi if bk["hi"] is None else max(bk["hi"], hi)
        bk["params"].append(spec)
    for bk in buckets:
        covered = sum(pad_to(s.numel, ALIGN) for s in bk["params"])
        assert covered == bk["hi"] - bk["lo"], "bucket is not a contiguous arena range"
    return buckets


class GradReducer:
    def __init__(self, store: ParamStore, optimizer, first_mb=DEFAULT_FIRST_BUCKET_MB, cap_mb=DEFAULT_BUCKET_CAP_MB,
                 broadcast_buffers=True, sync_mode=False, process_group=None, native_comm=None):
        self.store, self.opt = store, optimizer
        self.pg = process_group
        self.native = native_comm            # NativeComm: one communicator per rank (csrc/comm), no process group
        if native_comm is not None:
            self.world, self.rank = native_comm.world, native_comm.rank
        else:
            self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
            self.rank = dist.get_rank(self.pg) if dist.is_initialized() else 0
        self.buckets = plan_buckets(store, first_mb, cap_mb)
        self.bucket_of = {}
        for bi, bk in enumerate(self.buckets):
            for s in bk["params"]:
                self.bucket_of[s.index] = bi
        self.broadcast_buffers = broadcast_buffers
        self.sync_mode = sync_mode            # debug: run every collective synchronously (race detection aid)
        # DPA_FORCE_COLLECTIVES=1: issue every collective even at world_size 1 (single-GPU rehearsal of the RCCL
        # stream/event path; a 1-rank all-reduce is the identity)
        self.force = os.environ.get("DPA_FORCE_COLLECTIVES", "0") == "1" and (native_comm is not None or dist.is_initialized())
        self.active = self.world > 1 or self.force          # are collectives issued at all?
        self._core_lib, self._core = self._make_core()
        self._pending = None
        self._works = None
        self.comm_bytes = sum((b["hi"] - b["lo"]) * 4 for b in self.buckets)
        optimizer.set_grad_scale(1.0 / self.world)
        if self.active:
            self._verify_and_broadcast()

    def _make_core(self):
        """The C++ reducer core (csrc/comm/reducer.cpp): pending counters, ready detection, launch of the bucket's
        all-reduce on the native communicator's stream.  With the torch.distributed backend (RCCL process group or
        gloo) the core does the bookkeeping and this class issues the collective for the bucket it returns."""
        from .native_comm import comm_lib
        lib = comm_lib()
        n = len(self.store.params)
        bo = (ctypes.c_int * n)(*[self.bucket_of.get(s.index, -1) for s in self.store.params])
        assert all(s.index == i for i, s in enumerate(self.store.params))
        nb = len(self.buckets)
        lo = (ctypes.c_longlong * max(nb, 1))(*[b["lo"] for b in self.buckets])
        hi = (ctypes.c_longlong * max(nb, 1))(*[b["hi"] for b in self.buckets])
        comm_h = ctypes.c_void_p(self.native.h) if (self.native is not None and self.active) else ctypes.c_void_p(0)
        core = lib.dpa_reducer_create(comm_h, ctypes.c_void_p(self.store.g.data_ptr()), n, bo, nb, lo, hi)
        if not core:
            raise RuntimeError("dpa_reducer_create failed (bucket plan out of range)")
        return lib, ctypes.c_void_p(core)

    def _stream(self):
        if self.store.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        return ctypes.c_void_p(0)

    def __del__(self):
        try:
            if getattr(self, "_core", None):
                self._core_lib.dpa_reducer_destroy(self._core)
                self._core = None
        except Exception:
            pass

    # ------------------------------------------------------------------ wrap-time sync (X2, X3)
    def _verify_and_broadcast(self):
        sig = hashlib.sha1(repr([(s.name, s.logical_shape, s.trainable) for s in self.store.params]).encode()).digest()[:8]
        if self.native is not None:
            for r, other in enumerate(self.native.all_gather_bytes(sig, "layout")):
                if other != sig:
                    raise RuntimeError(f"parameter layout differs between rank {self.rank} and rank {r}")
            for t in (self.store.p, self.store.buffer_arena):
                self.native.broadcast(t, 0)
            self.store.refresh_compute_copy()
            return
        mine = torch.tensor(list(sig), dtype=torch.int64, device=self.store.device)
        allsig = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allsig, mine, group=self.pg)
        for r, t in enumerate(allsig):
            if not torch.equal(t, mine):
                raise RuntimeError(f"parameter layout differs between rank {self.rank} and rank {r}")
        dist.broadcast(self.store.p, 0, group=self.pg)
        dist.broadcast(self.store.buffer_arena, 0, group=self.pg)
        self.store.refresh_compute_copy()

    # ------------------------------------------------------------------ per step
    def sync_buffers(self):
        """DDP's per-forward buffer broadcast (X4): rank 0's BatchNorm running statistics overwrite everyone's.
        ONE collective per step (statistics and num_batches_tracked share a byte arena), issued here at the start of
        the step and waited for by the FIRST BatchNorm that touches the statistics (ParamStore.fence_buffers), so the
        stem convolution runs beside it instead of behind it."""
        if self.active and self.broadcast_buffers and self.store.b_size > 0:
            st = self.store
            st.fence_buffers()                  # a previous step's broadcast nobody waited for (a model pass without BatchNorm)
            if self.native is not None:
                ticket = self.native.broadcast_async(st.buffer_arena, 0)
                st.buffer_fence = lambda: self.native.broadcast_wait(ticket)
                return
            dist.broadcast(st.buffer_arena, 0, group=self.pg)

    def begin_backward(self):
        self._core_lib.dpa_reducer_begin(self._core)
        self._works = [None] * len(self.buckets)

    def param_ready(self, spec):
        """Backward-tape hook: the gradient kernel of `spec` has been queued on the current stream."""
        bi = self._core_lib.dpa_reducer_param_ready(self._core, spec.index, self._stream())
        if bi >= 0:
            if self.native is not None:
                if self.active:
                    self._works[bi] = "native"    # the core enqueued the all-reduce on the communicator's stream
                    if self.sync_mode:            # debug: the collective finishes before backward continues
                        self.native.sync()
            else:
                self._launch(bi)
        elif bi == -2:
            raise RuntimeError(f"reducer protocol error: gradient of {spec.name} reported ready twice in one backward")
        elif bi == -3:
            detail = self.native.last_error() if self.native is not None else ""
            raise RuntimeError(f"all-reduce could not be enqueued for the bucket of {spec.name}: {detail}")

    def _launch(self, bi):
        """torch.distributed backend (RCCL process group on GPUs, gloo on CPUs): async all-reduce of bucket bi."""
        bk = self.buckets[bi]
        sl = self.store.g[bk["lo"]: bk["hi"]]
        if self.world == 1 and not self.force:
            return
        if self.sync_mode:
            dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            self._works[bi] = dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def finish_and_step(self, mark=None):
        """Wait for each bucket in ready order and launch its optimizer update right away.  mark(name): optional phase
        callback (utils/steplog.py): 'wait' after each bucket's wait, 'opt' after each optimizer launch - so the step log
        can tell exposed all-reduce time from optimizer time."""
        self.store.fence_buffers()          # (no-op unless no BatchNorm ran this step)
        stream = self._stream()
        for bi, bk in enumerate(self.buckets):
            rc = self._core_lib.dpa_reducer_wait_bucket(self._core, bi, stream)   # native: hipStreamWaitEvent(done[b])
            if rc == 1:
                missing = self._core_lib.dpa_reducer_pending(self._core, bi)
                raise RuntimeError(f"bucket {bi} never became ready: {missing} gradients missing "
                                   f"(a trainable parameter did not take part in backward)")
            if rc != 0:
                detail = self.native.last_error() if self.native is not None else ""
                raise RuntimeError(f"waiting for the all-reduce of bucket {bi} failed on rank {self.rank}: {detail}")
            w = self._works[bi]
            if w is not None and w != "native":
                w.wait()                       # stream-level wait on GPU; blocks on gloo
            if mark is not None:
                mark("wait")
            self.opt.step_range(bk["lo"], bk["hi"])
            if mark is not None:
                mark("opt")
        self.opt.finish_step()

    # ------------------------------------------------------------------ diagnostics
    def check_in_sync(self):
        """All ranks must hold bit-identical parameters; returns the max deviation of the checksum (0.0 if in sync)."""
        if not self.active:
            return 0.0
        cs = torch.stack([self.store.p.double().sum(), self.store.p.double().abs().sum()])
        if self.native is not None:
            vals = cs.tolist()
            lo = self.native.all_reduce_scalars(vals, "min")
            hi = self.native.all_reduce_scalars(vals, "max")
            return max(abs(h - l) for h, l in zip(hi, lo))
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        return float((hi - lo).abs().max())

    def describe(self):
        return [dict(bucket=i, mbytes=(b["hi"] - b["lo"]) * 4 / 2 ** 20, params=len(b["params"])) for i, b in
                enumerate(self.buckets)]

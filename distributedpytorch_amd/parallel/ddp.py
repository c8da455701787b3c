"""One training / evaluation iteration of the reference's hot loop (/root/reference/classif.py:41-62) on the engine:
forward -> fused loss+accuracy -> hand-written backward with bucketed all-reduce overlapped -> fused optimizer.
No host synchronisation happens inside: loss/accuracy accumulate on the device (K16) and are read at log points.

`graph=True` captures the whole iteration (forward, backward, optimizer) into a hipGraph after a warm-up, which
removes the per-kernel launch cost that dominates small-batch MNIST steps.  Collectives are
captured too when they go through the native RCCL communicator (`can_capture`).
"""
from __future__ import annotations

import torch

from ..nn.arena import ActivationArena
from ..nn.engine import Tape, set_current_arena
from ..utils.trace import PhaseTimer, rng
from ..nn.loss import Criterion
from ..nn.optim import ArenaOptimizer
from .reducer import GradReducer


class DistributedModel:
    """The DDP-wrapped model + optimizer: owns the reducer and the device-side metric accumulators."""

    def __init__(self, net, criterion: Criterion, optimizer: ArenaOptimizer = None, first_bucket_mb=1.0,
                 bucket_cap_mb=25.0, broadcast_buffers=True, sync_comm=False, native_comm=None, activation_arena=False):
        """activation_arena: serve the step's activation / gradient tensors from a planned arena (nn/arena.py).  Opt-in
        because it assumes that the caller keeps no tensor of a step beyond the start of the next one unless it kept
        it in the recorded step too - true for the trainer and bench.py, not for arbitrary API users."""
        self.net, self.criterion, self.opt = net, criterion, optimizer
        dev = net.store.device
        self.device = dev
        self.arena = ActivationArena(dev) if activation_arena else None
        self.reducer = None
        if optimizer is not None:
            self.reducer = GradReducer(net.store, optimizer, first_bucket_mb, bucket_cap_mb, broadcast_buffers,
                                       sync_mode=sync_comm, native_comm=native_comm)
        self.metrics = torch.zeros(2, device=dev)          # last-iteration (loss, acc)
        self.accum = torch.zeros(3, device=dev)            # epoch sums: loss, acc, iterations
        self._graph = None
        self._static_x = self._static_y = None
        self.steplog = None            # utils/steplog.StepLog, set by the trainer on global rank 0
        # DDP's `require_forward_param_sync` (torch/nn/parallel/distributed.py:1557-1558,1616-1617): True at wrap time
        # and after every grad-enabled forward, dropped by a no-grad forward - i.e. the FIRST eval forward after a
        # training step still broadcasts rank 0's BatchNorm statistics, later ones do not (X4 of SURVEY.md §2.6).
        self._need_buffer_sync = True

    # ------------------------------------------------------------------ eager iteration
    def _forward_loss(self, x, labels, tape):
        ctx, out = self.net.forward(x, tape)
        if isinstance(out, tuple):                          # inception in train mode: loss = main + 0.4 * aux
            main, aux = out
            self.criterion(ctx, main, labels, self.net.num_classes, self.metrics, accum=self.accum)
            self.criterion(ctx, aux, labels, self.net.num_classes, self.metrics, accum=self.accum, scale=0.4, aux=True)
        else:
            self.criterion(ctx, out, labels, self.net.num_classes, self.metrics, accum=self.accum)
        return ctx

    def train_step(self, x, labels, mark=None):
        """x: NHWC compute-dtype batch on the device; labels: int64 on the device.  mark(name): optional callback at
        the end of each phase (utils/steplog.py records a HIP event there; never a host sync)."""
        net, red = self.net, self.reducer
        opt_mark = mark
        mark = mark or (lambda name: None)
        net.train()
        with rng("sync_buffers"):
            red.sync_buffers()
        mark("sync")
        tape = Tape()
        tape.param_ready_hook = red.param_ready
        red.begin_backward()
        prev = self._arena_begin(("train", tuple(x.shape), x.dtype))
        try:
            with rng("forward+loss"):
                self._forward_loss(x, labels, tape)
            mark("fwd")
            with rng("backward+allreduce"):
                tape.backward()
        finally:
            self._arena_end(prev)
        mark("bwd")
        with rng("wait+optimizer"):
            red.finish_and_step(opt_mark)
        from .. import ops
        ops.rng_advance(net.rng)
        mark("opt")
        self._need_buffer_sync = True

    def timed_train_step(self, x, labels):
        """One iteration with HIP-event timing per phase (synchronises): {'forward': ms, 'backward': ms, 'optimizer': ms}."""
        net, red = self.net, self.reducer
        net.train()
        t = PhaseTimer()
        with t.phase("sync_buffers"):
            red.sync_buffers()
        tape = Tape()
        tape.param_ready_hook = red.param_ready
        red.begin_backward()
        with t.phase("forward"):
            self._forward_loss(x, labels, tape)
        with t.phase("backward"):
            tape.backward()
        with t.phase("optimizer"):
            red.finish_and_step()
        from .. import ops
        ops.rng_advance(net.rng)
        self._need_buffer_sync = True
        return t.result()

    def eval_step(self, x, labels):
        self.net.eval()
        if self._need_buffer_sync:
            if self.reducer is not None:
                self.reducer.sync_buffers()
            self._need_buffer_sync = False
        prev = self._arena_begin(("eval", tuple(x.shape), x.dtype))
        try:
            self._forward_loss(x, labels, None)
        finally:
            self._arena_end(prev)

    def _arena_begin(self, key):
        if self.arena is None:
            return None
        self.arena.begin(key)
        return (set_current_arena(self.arena),)

    def _arena_end(self, prev):
        if prev is not None:
            self.arena.end()
            set_current_arena(prev[0])

    # ------------------------------------------------------------------ hipGraph iteration (single-rank body)
    def capture_train_step(self, x, labels, warmup=3):
        """Capture the whole iteration into a hipGraph.  Returns False if unsupported.
        Without collectives (world size 1) that is forward + backward + optimizer.  With collectives the native RCCL
        communicator's work is captured too: the ready/done event fencing forks the capture onto the comm stream,
        ncclAllReduce / ncclBroadcast are recorded there (RCCL supports stream capture) and the per-bucket waits join it
        back before the optimizer kernels.  The torch process-group path and the staged TCP transport (host-side
        waits) cannot be captured."""
        if not self.can_capture():
            return False
        if self.arena is not None:
            # a captured graph owns a static pool of its own: recording / replaying the arena during the warm-up would
            # only leave a second copy of the activation footprint allocated next to it
            self.arena.enabled = False
        self._static_x, self._static_y = x.clone(), labels.clone()
        # warm-up iterations must not perturb training: snapshot every piece of mutable state and restore it
        st, opt = self.net.store, self.opt
        state = [st.p, st.buffer_arena, opt.m, opt.hyper, self.net.rng, self.accum, self.metrics]
        if opt.v is not None:
            state.append(opt.v)
        if st.p16 is not None:
            state.append(st.p16)
        state.extend(st.aux_tensors)       # BatchNorm's statistics shift (previous batch mean) is carried state too
        saved = [t.clone() for t in state]
        host_step = opt._host_step
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.train_step(self._static_x, self._static_y)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        nc = self.reducer.native
        if nc is not None and self.reducer.active:
            nc.quiesce()            # the watchdog must have retired the warm-up's collectives: no hipEventQuery beside a capture
        from ..nn.engine import freeze_workspaces
        freeze_workspaces()         # the graph's kernels hold the scratch buffers' addresses from here on
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.train_step(self._static_x, self._static_y)
        self._graph = g
        for t, sv in zip(state, saved):
            t.copy_(sv)
        opt._host_step = host_step
        return True

    def can_capture(self):
        """May the whole iteration be captured into a hipGraph?  Without collectives: yes.  With collectives only through
        the native RCCL transport AND only when asked for (DPA_GRAPH_COLLECTIVES=1): replaying captured ncclAllReduce /
        ncclBroadcast nodes interleaved with eager collectives (partial last batch, validation's buffer broadcast) has
        been seen bit-for-bit at world size 1 only - until a multi-rank replay has been observed it is not a default
        (VERDICT r2 missing #5)."""
        if self.device.type != "cuda" or self.reducer is None:
            return False
        if not self.reducer.active:
            return True
        import os
        if os.environ.get("DPA_GRAPH_COLLECTIVES", "0") != "1":
            return False
        nc = self.reducer.native
        return nc is not None and nc.transport == "rccl" and not self.reducer.sync_mode

    def replay_train_step(self, x, labels):
        self._static_x.copy_(x, non_blocking=True)
        self._static_y.copy_(labels, non_blocking=True)
        self._graph.replay()
        self._need_buffer_sync = True

    # ------------------------------------------------------------------ metrics (host sync only here)
    def reset_epoch_metrics(self):
        self.accum.zero_()

    def epoch_metrics(self):
        """(mean loss, mean acc, iterations) - one D2H sync."""
        a = self.accum.tolist()
        n = max(a[2], 1.0)
        return a[0] / n, a[1] / n, int(a[2])

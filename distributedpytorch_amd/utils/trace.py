"""Tracing / profiling aids (the reference only has wall-clock epoch timers).

* `rng(name)`: a roctx range (torch.cuda.nvtx is roctx on ROCm) around forward / backward / comm / optimizer, visible
  in `rocprofv3 --marker-trace`; a no-op unless DPA_TRACE=1 so the hot loop pays nothing by default.
* `PhaseTimer`: HIP-event timing of the phases of one iteration, for the JSONL metrics and for scripts/profile_step.py.
Recipes for counters (gfx950: 8 SQ + 4 TCC slots per pass, FETCH_SIZE reads 1/2 of streamed bytes) are in
profiles/README.md.
"""
from __future__ import annotations

import contextlib
import os

import torch

ENABLED = os.environ.get("DPA_TRACE", "0") == "1"


@contextlib.contextmanager
def rng(name):
    if ENABLED and torch.cuda.is_available():
        torch.cuda.nvtx.range_push(name)
        try:
            yield
        finally:
            torch.cuda.nvtx.range_pop()
    else:
        yield


class PhaseTimer:
    """with timer.phase('fwd'): ...   -> timer.result() = {'fwd': ms, ...} (synchronises once, at result())."""

    def __init__(self, enabled=True):
        self.enabled = enabled and torch.cuda.is_available()
        self.marks = []

    @contextlib.contextmanager
    def phase(self, name):
        if not self.enabled:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        with rng(name):
            yield
        b.record()
        self.marks.append((name, a, b))

    def result(self):
        if not self.enabled:
            return {}
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.marks:
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        self.marks = []
        return out

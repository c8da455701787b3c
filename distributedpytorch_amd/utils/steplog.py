"""Step-level machine-readable metrics: `<RSL_PATH>/steps.jsonl`, one line per training step, written
by global rank 0 - step, loss, acc, images/s and where the time went (data pipeline, BatchNorm-buffer broadcast,
forward, backward with its overlapped all-reduces, bucket waits + fused optimizer).

The hot loop never synchronises for this: phase boundaries are HIP events recorded on the compute stream, the step's
(loss, acc) is copied into a device-side log row, and everything is read back in one go at the points where the
trainer synchronises anyway (the reference's 10 % progress lines, classif.py:63-68, and the end of the epoch).
The reference logs nothing per step and never computes images/s (classif.py:68,178-180)."""
from __future__ import annotations

import json
import os
import time

import torch

PHASES = ("data", "sync", "fwd", "bwd", "wait", "opt")


class StepLog:
    def __init__(self, path, device, world_size, per_rank_batch, capacity=4096, enabled=True):
        self.enabled = enabled and path is not None
        self.path, self.device, self.world, self.batch = path, torch.device(device), world_size, per_rank_batch
        self.cuda = self.device.type == "cuda"
        self.capacity = capacity
        self.rows = torch.zeros(capacity, 2, device=self.device) if self.enabled else None
        self.pending = []           # (global step, epoch, n images, marks[list of events / timestamps], names)
        self._pool = []
        self._cur = None
        if self.enabled:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    # ------------------------------------------------------------------ recording (no host sync)
    def _stamp(self):
        if not self.cuda:
            return time.perf_counter()
        ev = self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def begin(self, step, epoch):
        if not self.enabled:
            return
        self._cur = dict(step=step, epoch=epoch, marks=[self._stamp()], names=[])

    def mark(self, name):
        """Phase `name` ended here (called by the trainer / DistributedModel between phases)."""
        if not self.enabled or self._cur is None:
            return
        self._cur["marks"].append(self._stamp())
        self._cur["names"].append(name)

    def end(self, metrics, n_images):
        """metrics: the device tensor (loss, acc) of this step."""
        if not self.enabled or self._cur is None:
            return
        if len(self.pending) >= self.capacity:
            self.flush()
        self.rows[len(self.pending)].copy_(metrics, non_blocking=True)
        self._cur["n"] = int(n_images)
        self.pending.append(self._cur)
        self._cur = None

    # ------------------------------------------------------------------ read-back (at the trainer's own sync points)
    def flush(self):
        if not self.enabled or not self.pending:
            return
        if self.cuda:
            self.pending[-1]["marks"][-1].synchronize()
        vals = self.rows[: len(self.pending)].tolist()
        with open(self.path, "a") as f:
            for rec, (loss, acc) in zip(self.pending, vals):
                m = rec["marks"]
                if self.cuda:
                    ms = [m[i].elapsed_time(m[i + 1]) for i in range(len(m) - 1)]
                    self._pool.extend(m)
                else:
                    ms = [(m[i + 1] - m[i]) * 1e3 for i in range(len(m) - 1)]
                total = sum(ms)
                out = dict(step=rec["step"], epoch=rec["epoch"], loss=round(loss, 6), acc=round(acc, 6),
                           step_ms=round(total, 4),
                           images_per_s=round(rec["n"] * self.world / max(total, 1e-9) * 1e3, 1), world_size=self.world)
                for name, v in zip(rec["names"], ms):      # "wait" and "opt" occur once per gradient bucket: summed
                    out[f"{name}_ms"] = round(out.get(f"{name}_ms", 0.0) + v, 4)
                # the two the survey names explicitly: time in the input pipeline, and time on communication that
                # backward did not hide = the buffer broadcast's issue + the per-bucket waits for the all-reduces.  The
                # fused optimizer launches between the waits are timed separately (opt_ms), so comm_ms is exposed
                # all-reduce time only.
                out["data_ms"] = out.get("data_ms", 0.0)
                out["comm_ms"] = round(out.get("sync_ms", 0.0) + out.get("wait_ms", 0.0), 4)
                f.write(json.dumps(out) + "\n")
        self.pending = []

"""Loss selection of the reference (/root/reference/classif.py:106-120) on the fused CE kernel.

`cross_entropy` -> mode 0, `weighted_cross_entropy` -> mode 1 (torch's weighted mean), `focal_loss` -> mode 2
(FocalLossN, utils.py:142-156: gamma 2, plain mean over the batch).  One launch yields loss, accuracy, the logits
gradient and adds loss/accuracy to device-resident epoch accumulators (no per-iteration .item(), Q13).
"""
from __future__ import annotations

import torch

from .. import ops
from .engine import Act, Context

MODES = {"cross_entropy": 0, "weighted_cross_entropy": 1, "focal_loss": 2}


class Criterion:
    def __init__(self, kind="cross_entropy", class_weights=None, gamma=2.0):
        if kind not in MODES:
            raise ValueError(f"Invalid loss {kind!r}")
        self.kind, self.mode, self.gamma = kind, MODES[kind], gamma
        self.class_weights = class_weights
        if self.mode != 0 and class_weights is None:
            raise ValueError(f"{kind} needs class weights")

    def to(self, device):
        if self.class_weights is not None:
            self.class_weights = self.class_weights.to(device=device, dtype=torch.float32)
        return self

    def __call__(self, ctx: Context, logits: Act, labels, num_classes, out, accum=None, scale=1.0, aux=False):
        """out: fp32[2] (loss, acc) on the device.  Sets logits.grad when the tape is recording."""
        z = logits.data.reshape(logits.data.shape[0], -1)
        dl = None
        if ctx.grad_enabled and logits.needs_grad:
            dl = torch.empty_like(z)
        ops.ce_loss(z, labels, num_classes, out, accum=accum, dlogits=dl, class_w=self.class_weights if self.mode else None,
                    mode=self.mode, gamma=self.gamma, scale=scale, aux=aux)
        if dl is not None:
            logits.grad = dl.view(logits.data.shape)
        return out

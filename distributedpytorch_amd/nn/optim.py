"""Fused arena optimizers: Adam(lr=1e-3) and SGD(lr=1e-3, momentum=0.9)+StepLR(1, 0.1) as selected by the reference
(/root/reference/classif.py:122-131,168-169).  One launch updates a contiguous arena range (the whole model, or one
DDP bucket so the update of bucket k overlaps the all-reduce of bucket k+1).  lr / step / 1/world live in a device
vector so a captured hipGraph stays valid when they change.  `state_dict()` reproduces torch.optim's layout
(per-parameter `step` 0-d fp32 tensor, `exp_avg`, `exp_avg_sq` / `momentum_buffer`) in torchvision's
OIHW layout so the reference's `optimizer.load_state_dict` accepts it.
"""
from __future__ import annotations

import torch

from .. import ops
from .engine import ParamStore

# hyper vector slots (csrc/hip/optim.hip)
LR, B1, B2, EPS, WD, STEP, MOM, GSCALE, OMB1, OMB2, B1LO, B2LO = range(12)


def make_hyper(lr, beta1, beta2, eps, weight_decay, step, momentum, grad_scale, device=None):
    """The 12-float hyper-parameter vector of the optimizer kernels.  Slots 8-11 carry what torch.optim keeps in Python
    doubles: float(1 - beta) for the moment increments and the low parts beta - float(beta), from which the kernel
    rebuilds the double betas for the bias corrections (csrc/hip/optim.hip)."""
    import struct

    def f32(x):
        return struct.unpack("f", struct.pack("f", x))[0]
    return torch.tensor([lr, beta1, beta2, eps, weight_decay, step, momentum, grad_scale, 1.0 - beta1, 1.0 - beta2,
                         beta1 - f32(beta1), beta2 - f32(beta2)], dtype=torch.float32, device=device)


class ArenaOptimizer:
    def __init__(self, store: ParamStore, kind="adam", lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 momentum=0.9, grad_scale=1.0):
        if kind not in ("adam", "SGD"):
            raise ValueError(f"Invalid optimizer {kind!r}")
        self.store, self.kind = store, kind
        dev = store.device
        self.hyper = make_hyper(lr, betas[0], betas[1], eps, weight_decay, 0.0, momentum if kind == "SGD" else 0.0,
                                grad_scale, device=dev)
        self.m = torch.zeros_like(store.p)
        self.v = torch.zeros_like(store.p) if kind == "adam" else None
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, momentum=momentum)
        self.base_lr = lr
        self._host_step = 0
        self._host_lr = lr

    # ------------------------------------------------------------------ stepping
    def step_range(self, lo, hi):
        s = self.store
        if self.kind == "adam":
            ops.adam_step(s.p, s.g, self.m, self.v, s.p16, self.hyper, lo, hi)
        else:
            ops.sgd_step(s.p, s.g, self.m, s.p16, self.hyper, lo, hi)

    def finish_step(self):
        ops.advance_step(self.hyper)
        self._host_step += 1

    def step(self):
        for lo, hi in self.store.trainable_ranges():
            self.step_range(lo, hi)
        self.finish_step()

    def set_lr(self, lr):
        self._host_lr = lr
        self.hyper[LR] = lr

    def get_lr(self):
        return self._host_lr

    def set_grad_scale(self, s):
        self.hyper[GSCALE] = s

    # ------------------------------------------------------------------ torch.optim-compatible state
    def state_dict(self):
        # The reference builds the optimizer over model.parameters() - ALL of them, frozen ones included
        # (/root/reference/classif.py:124, utils.py:107-110) - so `params` indexes every parameter in registration
        # order and `state` holds entries only for those that received gradients (the trainable ones).
        st = self.store
        state = {}
        step = float(self.hyper[STEP].item())
        if step > 0:
            for i, spec in enumerate(st.params):
                if not spec.trainable:
                    continue
                sl = slice(spec.offset, spec.offset + spec.numel)
                if self.kind == "adam":
                    state[i] = {"step": torch.tensor(step, dtype=torch.float32),
                                "exp_avg": st.storage_to_logical(spec, self.m[sl].view(spec.storage_shape)),
                                "exp_avg_sq": st.storage_to_logical(spec, self.v[sl].view(spec.storage_shape))}
                else:
                    state[i] = {"momentum_buffer": st.storage_to_logical(spec, self.m[sl].view(spec.storage_shape))}
        lr = float(self.get_lr())
        if self.kind == "adam":
            group = {"lr": lr, "betas": tuple(self.defaults["betas"]), "eps": self.defaults["eps"],
                     "weight_decay": self.defaults["weight_decay"], "amsgrad": False, "maximize": False, "foreach": None,
                     "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False}
        else:
            group = {"lr": lr, "momentum": self.defaults["momentum"], "dampening": 0, "weight_decay":
                     self.defaults["weight_decay"], "nesterov": False, "maximize": False, "foreach": None,
                     "differentiable": False, "fused": None, "initial_lr": self.base_lr}
        group["params"] = list(range(len(st.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        st = self.store
        group = sd["param_groups"][0]
        if len(group["params"]) != len(st.params):
            raise ValueError(f"optimizer state_dict covers {len(group['params'])} parameters, the model has "
                             f"{len(st.params)}")
        self.set_lr(float(group["lr"]))
        step = 0.0
        for i, spec in enumerate(st.params):
            s = sd["state"].get(i)
            if s is None or not spec.trainable:
                continue
            sl = slice(spec.offset, spec.offset + spec.numel)
            if self.kind == "adam":
                step = float(s["step"])
                st.logical_to_storage(spec, s["exp_avg"].to(st.device, torch.float32), self.m[sl].view(spec.storage_shape))
                st.logical_to_storage(spec, s["exp_avg_sq"].to(st.device, torch.float32), self.v[sl].view(spec.storage_shape))
            else:
                step = max(step, 1.0)
                st.logical_to_storage(spec, s["momentum_buffer"].to(st.device, torch.float32), self.m[sl].view(spec.storage_shape))
        self.hyper[STEP] = step
        self._host_step = int(step)


class StepLR:
    """lr = base_lr * gamma ** (epoch // step_size); the reference steps it once per epoch for SGD only."""

    def __init__(self, optimizer: ArenaOptimizer, step_size=1, gamma=0.1, last_epoch=0):
        self.opt, self.step_size, self.gamma, self.epoch = optimizer, step_size, gamma, last_epoch
        self.opt.set_lr(self.opt.base_lr * self.gamma ** (self.epoch // self.step_size))

    def step(self):
        self.epoch += 1
        self.opt.set_lr(self.opt.base_lr * self.gamma ** (self.epoch // self.step_size))

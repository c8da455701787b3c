#!/usr/bin/env python3
"""Phase breakdown of one training iteration with HIP events (forward / backward / optimizer), plus the recipe to get a
kernel trace with roctx ranges:

  python scripts/profile_step.py --model resnet50 --batch 256 --dtype fp32
  DPA_TRACE=1 rocprofv3 --kernel-trace --marker-trace --stats -d gpurun_out/prof -- python3 scripts/profile_step.py ...
  python scripts/rocpd_stats.py gpurun_out/prof/*_results.db          # per-kernel CSV
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models.factory import get_model, get_model_input_size  # noqa: E402
from distributedpytorch_amd.nn.loss import Criterion  # noqa: E402
from distributedpytorch_amd.nn.optim import ArenaOptimizer  # noqa: E402
from distributedpytorch_amd.parallel.ddp import DistributedModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="resnet50")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--classes", type=int, default=1000)
a = ap.parse_args()
dtype = torch.float32 if a.dtype == "fp32" else torch.bfloat16
side = 224 if a.model == "resnet50" else get_model_input_size(a.model)
net, _ = get_model(a.model, a.classes, device="cuda", dtype=dtype)
dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3))
x = torch.zeros(a.batch, side, side, net.in_channels, device="cuda", dtype=dtype)
x[..., :3] = torch.randn(a.batch, side, side, 3, device="cuda").to(dtype)
y = torch.randint(0, a.classes, (a.batch,), device="cuda")
for _ in range(2):
    dm.train_step(x, y)
tot = {}
for _ in range(a.steps):
    for k, v in dm.timed_train_step(x, y).items():
        tot[k] = tot.get(k, 0.0) + v
res = {k: round(v / a.steps, 3) for k, v in tot.items()}
res["total_ms"] = round(sum(res.values()), 3)
res["images_per_s"] = round(a.batch / res["total_ms"] * 1e3, 1)
print(json.dumps({"model": a.model, "batch": a.batch, "dtype": a.dtype, "phase_ms": res}))

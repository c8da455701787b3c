"""Achievable HBM bandwidth on this GPU with plain streaming kernels (context for the BatchNorm kernels' GB/s):
torch copy (R+W), torch add (2R+W), read-only sum, and our bn_apply on the same sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd import ops  # noqa: E402


def t_ms(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for shape in ((1024, 56, 56, 256), (1024, 28, 28, 512), (1024, 56, 56, 64), (1024, 14, 14, 1024)):
    a = torch.randn(shape, device="cuda"); b = torch.randn(shape, device="cuda"); c = torch.empty_like(a)
    gb = a.numel() * 4 / 1e9
    sc = torch.randn(shape[-1], device="cuda"); sh = torch.randn(shape[-1], device="cuda")
    coef = torch.randn(3, shape[-1], device="cuda"); sums = torch.zeros(2, shape[-1], device="cuda")
    dg = torch.zeros(shape[-1], device="cuda"); db = torch.zeros(shape[-1], device="cuda")
    r = {"copy (R+W)": (t_ms(lambda: c.copy_(a)), 2), "add (2R+W)": (t_ms(lambda: torch.add(a, b, out=c)), 3),
         "sum (R)": (t_ms(lambda: a.sum()), 1),
         "bn_apply relu (R+W)": (t_ms(lambda: ops.bn_apply(a, sc, sh, c, relu=True)), 2),
         "bn_apply +res relu (2R+W)": (t_ms(lambda: ops.bn_apply(a, sc, sh, c, residual=b, relu=True)), 3),
         "bn_bwd_dx (2R+W)": (t_ms(lambda: ops.bn_bwd_dx(a, b, coef, c, mask_scale=sc, mask_shift=sh)), 3),
         "bn_bwd_reduce mask y (3R+W)": (t_ms(lambda: ops.bn_bwd_reduce(c, a, b, sc, sh, sc, dg, db, sums, coef)), 4),
         "bn_bwd_reduce mask x (2R)": (t_ms(lambda: ops.bn_bwd_reduce(c, None, b, sc, sh, sc, dg, db, sums, coef,
                                                                    mask_scale=sc, mask_shift=sh)), 2)}
    print(f"{shape}  {gb:.2f} GB per pass")
    for k, (ms, passes) in r.items():
        print(f"   {k:28s} {ms:7.3f} ms  {passes * gb / ms:7.2f} TB/s".replace("TB/s", "GB/ms = TB/s"))

"""A/B of the fp32 x6 convolution kernels on the ResNet-50 layer shapes: general x6 kernel (operands split after every
fragment read) vs the pre-split kernel (csrc/hip/conv_x6p.hip), interleaved in one process.  TFLOP/s are fp32-equivalent
(2 * M * N * K / time).  python scripts/bench_x6p.py [--batch 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import hip_conv  # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode  # noqa: E402

LAYERS = [(56, 64, 64, 1, 1), (56, 64, 64, 3, 1), (56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (56, 256, 128, 1, 1),
          (56, 128, 128, 3, 2), (28, 128, 512, 1, 1), (28, 512, 128, 1, 1), (28, 128, 128, 3, 1), (28, 256, 256, 3, 2),
          (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1), (14, 256, 256, 3, 1), (14, 512, 512, 3, 2), (7, 512, 2048, 1, 1),
          (7, 2048, 512, 1, 1), (7, 512, 512, 3, 1)]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    a = ap.parse_args()
    set_fp32_mode("x6")
    print(f"{'layer':28s} | fprop x6 us TF | fprop x6p us TF | dgrad x6 us TF | dgrad x6p us TF")
    tot = [0.0] * 4
    for hw, ci, co, k, s in LAYERS:
        pd = k // 2
        N = a.batch
        x = torch.randn(N, hw, hw, ci, device="cuda")
        w = torch.randn(co, k, k, ci, device="cuda") * (k * k * ci) ** -0.5
        ho = hip_conv.out_size(hw, k, s, pd)
        y = torch.empty(N, ho, ho, co, device="cuda")
        dy = torch.randn(N, ho, ho, co, device="cuda")
        dx = torch.empty_like(x)
        wt = w.permute(3, 1, 2, 0).contiguous()
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
        wt3 = hip_conv.split3_planes(wt, torch.empty(3 * wt.numel(), device="cuda", dtype=torch.bfloat16))
        flop = 2.0 * N * ho * ho * co * ci * k * k
        ts = []
        for rep in range(2):          # interleaved rounds, keep the best of each
            t = [timeit(lambda: hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd))),
                 timeit(lambda: hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd), w3=w3)),
                 timeit(lambda: hip_conv.conv_dgrad(dy, wt, dx, (s, s), (pd, pd))),
                 timeit(lambda: hip_conv.conv_dgrad(dy, wt, dx, (s, s), (pd, pd), wt3=wt3))]
            ts = t if not ts else [min(u, v) for u, v in zip(ts, t)]
        for i in range(4):
            tot[i] += ts[i]
        print(f"{hw}x{hw} {ci}->{co} k{k} s{s}".ljust(28) + " | " + " | ".join(f"{t:8.1f} {flop / t / 1e6:5.0f}" for t in ts), flush=True)
    print("totals (ms, one call per distinct layer): " + ", ".join(f"{t / 1e3:.2f}" for t in tot))


if __name__ == "__main__":
    main()

"""Experiment (docs/KERNELS.md §2): how fast would the fp32 x6 forward convolution be if the ACTIVATIONS arrived pre-split
into three bf16 planes as well ("producer-side split": BatchNorm kernels would write 6 B per element instead of 4)?
conv_x6p.hip has an experiment instantiation (APL) that stages both operands as planes and has no VALU split in the K
loop.  Compares it with the production pre-split-filter kernel on ResNet-50 layers and checks the result bit for bit
(same arithmetic).   python scripts/bench_x6pp.py   [DPA_X6PP_TN=4: 128 x 64 tiles = two blocks per CU]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import hip_conv as H  # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode  # noqa: E402

LAYERS = [(56, 64, 64, 3), (56, 256, 64, 1), (56, 64, 256, 1), (28, 128, 128, 3), (28, 512, 128, 1), (28, 128, 512, 1),
          (14, 256, 256, 3), (14, 1024, 256, 1), (14, 256, 1024, 1), (7, 512, 512, 3), (7, 2048, 512, 1), (7, 512, 2048, 1)]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    set_fp32_mode("x6")
    N = 1024
    tot = [0.0, 0.0]
    print(f"tile width with activation planes: {os.environ.get('DPA_X6PP_TN', '8')} x 16 columns")
    print(f"{'layer':24s} | x6p us  TF | x6pp us  TF | ratio")
    for hw, ci, co, k in LAYERS:
        pd = k // 2
        x = torch.randn(N, hw, hw, ci, device="cuda")
        w = torch.randn(co, k, k, ci, device="cuda") * (k * k * ci) ** -0.5
        y0 = torch.empty(N, hw, hw, co, device="cuda")
        y1 = torch.full_like(y0, float("nan"))
        w3 = H.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
        x3 = H.split3_planes(x, torch.empty(3 * x.numel(), device="cuda", dtype=torch.bfloat16))
        t0 = timeit(lambda: H.conv_fprop(x, w, y0, (1, 1), (pd, pd), w3=w3))
        t1 = timeit(lambda: H.conv_fprop(x, w, y1, (1, 1), (pd, pd), w3=w3, _x3=x3))
        assert torch.equal(y0, y1), (hw, ci, co, k, (y0 - y1).abs().max().item())
        fl = 2.0 * N * hw * hw * co * ci * k * k
        tot[0] += t0; tot[1] += t1
        print(f"{hw}x{hw} {ci}->{co} k{k}".ljust(24) + f" | {t0:7.1f} {fl / t0 / 1e6:4.0f} | {t1:7.1f} {fl / t1 / 1e6:4.0f} | {t1 / t0:.3f}")
    print(f"totals (ms): x6p {tot[0] / 1e3:.2f}, x6pp {tot[1] / 1e3:.2f}  ({tot[1] / tot[0]:.3f})")


if __name__ == "__main__":
    main()

"""A/B of the 8-wave double-buffered fp32 (x6) weight-gradient kernel (conv_wgrad_x6q_kernel) against the 4-wave kernel on
the ResNet-50 layer shapes, interleaved in one process; the results must agree to rounding (same products, another
summation order across split-K slabs).  python scripts/diag/wgrad_x6q_ab.py [--batch 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.ops import hip_conv as H  # noqa: E402
from distributedpytorch_amd.ops.native import hip_lib, set_fp32_mode  # noqa: E402

# (H, Cin, Cout, k, stride, count per ResNet-50 step)
LAYERS = [(56, 64, 64, 1, 1, 1), (56, 64, 64, 3, 1, 3), (56, 64, 256, 1, 1, 4), (56, 256, 64, 1, 1, 2), (56, 256, 128, 1, 1, 1),
          (56, 128, 128, 3, 2, 1), (28, 128, 512, 1, 1, 4), (56, 256, 512, 1, 2, 1), (28, 512, 128, 1, 1, 3), (28, 128, 128, 3, 1, 3),
          (28, 512, 256, 1, 1, 1), (28, 256, 256, 3, 2, 1), (14, 256, 1024, 1, 1, 6), (28, 512, 1024, 1, 2, 1),
          (14, 1024, 256, 1, 1, 5), (14, 256, 256, 3, 1, 5), (14, 1024, 512, 1, 1, 1), (14, 512, 512, 3, 2, 1),
          (7, 512, 2048, 1, 1, 3), (14, 1024, 2048, 1, 2, 1), (7, 2048, 512, 1, 1, 2), (7, 512, 512, 3, 1, 2)]


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
    L = hip_lib()
    N = a.batch
    tot = [0.0, 0.0]
    print(f"{'layer':28s} | 4-wave us TF | 8-wave x6q us TF | max rel diff")
    for hw, ci, co, k, s, cnt in LAYERS:
        pd = k // 2
        ho = H.out_size(hw, k, s, pd)
        x = torch.randn(N, hw, hw, ci, device="cuda")
        dy = torch.randn(N, ho, ho, co, device="cuda")
        dw = torch.empty(co, k, k, ci, device="cuda")
        ws = torch.empty(H.wgrad_workspace_numel(co, k * k * ci), device="cuda")
        flop = 2.0 * N * ho * ho * co * ci * k * k
        f = lambda: H.conv_wgrad(dy, x, dw, (k, k), (s, s), (pd, pd), workspace=ws)      # noqa: E731
        L.dpa_wgrad_knob(1, 0); f(); d0 = dw.clone()
        L.dpa_wgrad_knob(1, 1); dw.zero_(); f()
        rel = float((dw - d0).abs().max() / d0.abs().max())
        best = [1e30, 1e30]
        for rep in range(3):
            for v in (0, 1):
                L.dpa_wgrad_knob(1, v)
                best[v] = min(best[v], timeit(f))
        for v in (0, 1):
            tot[v] += best[v] * cnt
        print(f"{hw}x{hw} {ci}->{co} k{k} s{s}".ljust(28) + " | " + " | ".join(f"{t:8.1f} {flop / t / 1e6:5.0f}" for t in best) + f" | {rel:.1e}", flush=True)
    L.dpa_wgrad_knob(1, 1)
    print("totals per ResNet-50 step (ms): 4-wave %.2f, 8-wave x6q %.2f" % (tot[0] / 1e3, tot[1] / 1e3))


if __name__ == "__main__":
    main()

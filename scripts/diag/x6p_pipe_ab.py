"""A/B of the software-pipelined K loop of the pre-split fp32 kernel (csrc/hip/conv_x6p.hip, PIPE) against the plain
double-buffered loop, on the ResNet-50 layer shapes, interleaved in one process; also checks that both produce the SAME
bits (per accumulator the order of the MFMA products is unchanged).  python scripts/diag/x6p_pipe_ab.py [--batch 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.ops import hip_conv  # noqa: E402
from distributedpytorch_amd.ops.native import hip_lib as _lib, set_fp32_mode  # noqa: E402

LAYERS = [(56, 64, 64, 1, 1), (56, 64, 64, 3, 1), (56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (56, 256, 128, 1, 1),
          (56, 128, 128, 3, 2), (28, 128, 512, 1, 1), (28, 512, 128, 1, 1), (28, 128, 128, 3, 1), (28, 256, 256, 3, 2),
          (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1), (14, 256, 256, 3, 1), (14, 512, 512, 3, 2), (7, 512, 2048, 1, 1),
          (7, 2048, 512, 1, 1), (7, 512, 512, 3, 1)]
COUNT = {(56, 64, 64, 1, 1): 1, (56, 64, 64, 3, 1): 3, (56, 64, 256, 1, 1): 4, (56, 256, 64, 1, 1): 2, (28, 128, 512, 1, 1): 4,
         (28, 512, 128, 1, 1): 3, (28, 128, 128, 3, 1): 3, (14, 256, 1024, 1, 1): 6, (14, 1024, 256, 1, 1): 5,
         (14, 256, 256, 3, 1): 5, (7, 512, 2048, 1, 1): 3, (7, 2048, 512, 1, 1): 2, (7, 512, 512, 3, 1): 2}


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
    ap.add_argument("--pipe-min", type=int, default=4)
    a = ap.parse_args()
    set_fp32_mode("x6")
    L = _lib()
    print(f"{'layer':26s} | fprop: plain us TF | pipe us TF | dgrad: plain us TF | pipe us TF | same bits")
    tot = [0.0] * 4
    CFG = [0, a.pipe_min]
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
        f = lambda: hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd), w3=w3)          # noqa: E731
        d = lambda: hip_conv.conv_dgrad(dy, wt, dx, (s, s), (pd, pd), wt3=wt3)     # noqa: E731
        # same bits?
        L.dpa_x6p_knob(3, 0); f(); d(); y0, dx0 = y.clone(), dx.clone()
        same = True
        for pm in CFG[1:]:
            L.dpa_x6p_knob(3, pm); y.zero_(); dx.zero_(); f(); d()
            same = same and bool(torch.equal(y, y0)) and bool(torch.equal(dx, dx0))
        ts = []
        for rep in range(3):
            t = []
            for fn in (f, d):
                for pm in CFG:
                    L.dpa_x6p_knob(3, pm)
                    t.append(timeit(fn))
            ts = t if not ts else [min(u, v) for u, v in zip(ts, t)]
        c = COUNT.get((hw, ci, co, k, s), 1)
        for i in range(4):
            tot[i] += ts[i] * c
        print(f"{hw}x{hw} {ci}->{co} k{k} s{s}".ljust(26) + " | " + " | ".join(f"{t:8.1f} {flop / t / 1e6:5.0f}" for t in ts)
              + f" | {same}", flush=True)
    L.dpa_x6p_knob(3, 4)
    print("totals per ResNet-50 step (ms): fprop plain %.2f, pipe %.2f; dgrad plain %.2f, pipe %.2f" % tuple(t / 1e3 for t in tot))


if __name__ == "__main__":
    main()

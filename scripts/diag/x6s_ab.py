"""A/B of the persistent streaming kernel (conv_x6s.hip) on its layers, interleaved rounds in ONE process:
conv_x6p.hip's kernel (DPA_X6S = 0) and the streaming kernel with two / one resident blocks per CU, with and without the
BatchNorm-statistics epilogue and with an addend.  python scripts/diag/x6s_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.ops import hip_conv  # noqa: E402
from distributedpytorch_amd.ops.native import hip_lib, set_fp32_mode  # noqa: E402


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
    set_fp32_mode("x6")
    lib = hip_lib()
    N = int(os.environ.get("BATCH", "1024"))
    for hw, ci, co in [(56, 64, 256), (56, 64, 64)]:
        x = torch.randn(N, hw, hw, ci, device="cuda")
        w = torch.randn(co, 1, 1, ci, device="cuda") * ci ** -0.5
        y = torch.empty(N, hw, hw, co, device="cuda")
        add = torch.randn(N, hw, hw, co, device="cuda")
        stats = torch.empty((N * hw * hw + 63) // 64, 2, co, device="cuda")
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
        variants = {"x6p": (0, 512), "x6s": (1, 512), "x6s 256 blocks": (1, 256)}
        best = {}
        for rep in range(3):
            for name, (on, blocks) in variants.items():
                lib.dpa_x6s_knob(1, on); lib.dpa_x6s_knob(2, blocks)
                for tag, kw in (("stats", dict(stats=stats)), ("plain", {}), ("addend", dict(addend=add)), ("relu", dict(relu=True))):
                    t = timeit(lambda: hip_conv.conv_fprop(x, w, y, (1, 1), (0, 0), w3=w3, **kw))
                    best[(name, tag)] = min(best.get((name, tag), 1e30), t)
        lib.dpa_x6s_knob(1, 1); lib.dpa_x6s_knob(2, 512)
        gb = (x.numel() + y.numel()) * 4 / 1e9
        print(f"{hw}x{hw} {ci}->{co} 1x1, batch {N}: {gb:.2f} GB in + out")
        for (name, tag), t in best.items():
            extra = gb + (y.numel() * 4 / 1e9 if tag == "addend" else 0)
            print(f"   {name:24s} {tag:7s} {t:8.1f} us   {extra / t * 1e3:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()

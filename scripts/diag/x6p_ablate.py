"""Where does a K-tile of the pre-split fp32 (x6p) forward kernel spend its time?  Ablation in ONE process, interleaved
rounds (cdna_hip_programming.md §7 'The diagnostic loop'): needs a diagnostic build of the kernels,

  python -m distributedpytorch_amd.csrc.build --libdir /tmp/lib_diag -D DPA_X6P_DIAG
  DPA_LIB_DIR=/tmp/lib_diag python scripts/diag/x6p_ablate.py

DPA_X6P_ABL bits: 1 = no operand split (VALU), 2 = no LDS-DMA after the second K-tile, 4 = no MFMAs.  The ablated
variants compute WRONG results; only their times mean anything."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.ops import hip_conv  # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode  # noqa: E402

LAYERS = [(14, 256, 256, 3, 1), (28, 128, 512, 1, 1), (56, 64, 256, 1, 1), (7, 512, 2048, 1, 1)]
NAMES = {0: "full", 1: "no split", 2: "no DMA", 3: "no split, no DMA", 4: "no MFMA", 5: "no split, no MFMA",
         6: "no DMA, no MFMA", 7: "LDS reads + barriers only"}


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
    from distributedpytorch_amd.ops.native import hip_lib
    hip_lib().dpa_x6p_knob(3, 0)        # ABL 0..7: the plain double-buffered loop; 8..15: the software-pipelined loop
    N = int(os.environ.get("BATCH", "1024"))
    for hw, ci, co, k, s in LAYERS:
        pd = k // 2
        x = torch.randn(N, hw, hw, ci, device="cuda")
        w = torch.randn(co, k, k, ci, device="cuda") * (k * k * ci) ** -0.5
        ho = hip_conv.out_size(hw, k, s, pd)
        y = torch.empty(N, ho, ho, co, device="cuda")
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
        flop = 2.0 * N * ho * ho * co * ci * k * k
        best = {}
        for rep in range(3):
            for abl in range(16):
                os.environ["DPA_X6P_ABL"] = str(abl)
                t = timeit(lambda: hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd), w3=w3))
                best[abl] = min(best.get(abl, 1e30), t)
        os.environ["DPA_X6P_ABL"] = "0"
        print(f"{hw}x{hw} {ci}->{co} k{k} s{s}  ({flop / 1e9:.1f} GFLOP)")
        for abl in range(16):
            print(f"   ABL={abl:2d} {('pipelined loop, ' if abl >= 8 else '') + NAMES[abl & 7]:44s} {best[abl]:9.1f} us   {best[abl] / best[abl & 8] * 100:5.1f} %   "
                  f"{flop / best[abl] / 1e6:6.0f} TF-eq", flush=True)


if __name__ == "__main__":
    main()

"""Where does a K-tile of the fp32 (x6) weight-gradient kernel spend its time?  Ablation in ONE process, interleaved rounds;
needs a diagnostic build:  python -m distributedpytorch_amd.csrc.build --libdir build_ab/diag -D DPA_X6P_DIAG -D DPA_WG_DIAG
then DPA_LIB_DIR=build_ab/diag python scripts/diag/wgrad_ablate.py.  DPA_WG_ABL bits: 1 = no operand split, 2 = no global
loads after the first tile, 4 = no MFMAs, 8 = no LDS stores after the first tile.  Ablated variants compute WRONG results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.ops import hip_conv as H  # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode  # noqa: E402

LAYERS = [(14, 256, 256, 3, 1), (28, 128, 128, 3, 1), (14, 1024, 256, 1, 1), (56, 256, 128, 1, 1)]
NAMES = {0: "full", 1: "no split", 2: "no global loads", 3: "no split, no loads", 4: "no MFMA", 8: "no LDS stores",
         9: "no split, no LDS stores", 12: "no MFMA, no LDS stores", 14: "barriers + fragment reads only (+ split)", 15: "barriers + fragment reads only"}


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
    N = int(os.environ.get("BATCH", "1024"))
    for hw, ci, co, k, s in LAYERS:
        pd = k // 2
        ho = H.out_size(hw, k, s, pd)
        x = torch.randn(N, hw, hw, ci, device="cuda")
        dy = torch.randn(N, ho, ho, co, device="cuda")
        dw = torch.empty(co, k, k, ci, device="cuda")
        ws = torch.empty(H.wgrad_workspace_numel(co, k * k * ci), device="cuda")
        flop = 2.0 * N * ho * ho * co * ci * k * k
        best = {}
        for rep in range(3):
            for abl in NAMES:
                os.environ["DPA_WG_ABL"] = str(abl)
                t = timeit(lambda: H.conv_wgrad(dy, x, dw, (k, k), (s, s), (pd, pd), workspace=ws))
                best[abl] = min(best.get(abl, 1e30), t)
        os.environ["DPA_WG_ABL"] = "0"
        print(f"{hw}x{hw} {ci}->{co} k{k} s{s}  ({flop / 1e9:.1f} GFLOP)")
        for abl in NAMES:
            print(f"   ABL={abl:2d} {NAMES[abl]:44s} {best[abl]:9.1f} us   {best[abl] / best[0] * 100:5.1f} %   {flop / best[abl] / 1e6:6.0f} TF-eq", flush=True)


if __name__ == "__main__":
    main()

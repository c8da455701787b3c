"""Forward time of the short-reduction 1x1 layers of ResNet-50 (and two 3x3 layers for contrast) through the pre-split
fp32 kernel - the harness behind the short-K paragraph of docs/KERNELS.md §2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.ops import hip_conv as H
from distributedpytorch_amd.ops.native import set_fp32_mode
set_fp32_mode("x6")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
N = 1024
out = []
for hw, ci, co, k in ((56, 64, 256, 1), (56, 256, 64, 1), (28, 128, 512, 1), (28, 512, 128, 1), (14, 256, 256, 3), (56, 64, 64, 3)):
    x = torch.randn(N, hw, hw, ci, device="cuda"); w = torch.randn(co, k, k, ci, device="cuda") * 0.1
    y = torch.empty(N, hw, hw, co, device="cuda")
    w3 = H.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
    out.append(f"{hw}x{hw} {ci}->{co} k{k}: {timeit(lambda: H.conv_fprop(x, w, y, (1, 1), (k // 2, k // 2), w3=w3)):.0f}")
print("us: " + " | ".join(out))

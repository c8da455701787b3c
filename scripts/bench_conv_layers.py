"""Per-layer conv micro-benchmark on ResNet-50 shapes (batch 256): TFLOP/s of fprop / dgrad / wgrad kernels."""
import argparse, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import hip_conv as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--only", default="")
ap.add_argument("--bnr", action="store_true", help="dgrad as the training step runs it for a BatchNorm+residual+ReLU input: in-place "
                "accumulation (addend) + the fused BatchNorm-backward reduction with the mask from y; adds a column with its time and traffic")
ap.add_argument("--net", default="resnet50", choices=["resnet50", "resnet18_cifar", "resnet18"],
                help="resnet18_cifar: BASELINE config 4 (ResNet-18 on 32x32 inputs: 3x3 stem, no max-pool; maps 32/16/8/4)")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
# (H, Cin, Cout, k, stride, pad, count)
LAYERS = [(224, 8, 64, 7, 2, 3, 1),
          (56, 64, 64, 1, 1, 0, 1), (56, 64, 64, 3, 1, 1, 3), (56, 64, 256, 1, 1, 0, 4), (56, 256, 64, 1, 1, 0, 2),
          (56, 256, 128, 1, 1, 0, 1), (56, 128, 128, 3, 2, 1, 1), (28, 128, 512, 1, 1, 0, 4), (56, 256, 512, 1, 2, 0, 1),
          (28, 512, 128, 1, 1, 0, 3), (28, 128, 128, 3, 1, 1, 3),
          (28, 512, 256, 1, 1, 0, 1), (28, 256, 256, 3, 2, 1, 1), (14, 256, 1024, 1, 1, 0, 6), (28, 512, 1024, 1, 2, 0, 1),
          (14, 1024, 256, 1, 1, 0, 5), (14, 256, 256, 3, 1, 1, 5),
          (14, 1024, 512, 1, 1, 0, 1), (14, 512, 512, 3, 2, 1, 1), (7, 512, 2048, 1, 1, 0, 3), (14, 1024, 2048, 1, 2, 0, 1),
          (7, 2048, 512, 1, 1, 0, 2), (7, 512, 512, 3, 1, 1, 2)]
if a.net == "resnet18_cifar":
    LAYERS = [(32, 8, 64, 3, 1, 1, 1), (32, 64, 64, 3, 1, 1, 4),
              (32, 64, 128, 3, 2, 1, 1), (16, 128, 128, 3, 1, 1, 3), (32, 64, 128, 1, 2, 0, 1),
              (16, 128, 256, 3, 2, 1, 1), (8, 256, 256, 3, 1, 1, 3), (16, 128, 256, 1, 2, 0, 1),
              (8, 256, 512, 3, 2, 1, 1), (4, 512, 512, 3, 1, 1, 3), (8, 256, 512, 1, 2, 0, 1)]
if a.net == "resnet18":          # the reference's default model at 224x224 (SURVEY.md §2.3 shape table)
    LAYERS = [(224, 8, 64, 7, 2, 3, 1), (56, 64, 64, 3, 1, 1, 4),
              (56, 64, 128, 3, 2, 1, 1), (28, 128, 128, 3, 1, 1, 3), (56, 64, 128, 1, 2, 0, 1),
              (28, 128, 256, 3, 2, 1, 1), (14, 256, 256, 3, 1, 1, 3), (28, 128, 256, 1, 2, 0, 1),
              (14, 256, 512, 3, 2, 1, 1), (7, 512, 512, 3, 1, 1, 3), (14, 256, 512, 1, 2, 0, 1)]
N = a.batch
tot = {"fprop": 0.0, "dgrad": 0.0, "wgrad": 0.0}
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / a.iters
print(f"{'layer':34s} {'GFLOP':>8s} | {'fprop us':>9s} {'TF':>6s} | {'dgrad us':>9s} {'TF':>6s} | {'wgrad us':>9s} {'TF':>6s}")
for (h, ci, co, k, s, p, cnt) in LAYERS:
    name = f"{h}x{h} {ci}->{co} k{k} s{s}"
    if a.only and a.only not in name: continue
    ho = H.out_size(h, k, s, p)
    x = torch.randn(N, h, h, ci, device="cuda").to(dt)
    w = (torch.randn(co, k, k, ci, device="cuda") * 0.05).to(dt)
    wt = w.permute(3, 1, 2, 0).contiguous()
    y = torch.empty(N, ho, ho, co, device="cuda", dtype=dt)
    dy = torch.randn(N, ho, ho, co, device="cuda").to(dt)
    dx = torch.empty_like(x)
    dw = torch.empty(co, k, k, ci, device="cuda")
    ws = torch.empty(H.wgrad_workspace_numel(co, k * k * ci), device="cuda")
    stats = torch.empty((N * ho * ho + 63) // 64, 2, co, device="cuda")
    w3 = wt3 = None
    if dt == torch.float32 and ci % 32 == 0 and co % 32 == 0:      # the pre-split planes the training step passes in x6 mode
        w3 = H.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
        wt3 = H.split3_planes(wt, torch.empty(3 * wt.numel(), device="cuda", dtype=torch.bfloat16))
    fl = 2.0 * N * ho * ho * co * k * k * ci
    tf = timeit(lambda: H.conv_fprop(x, w, y, (s, s), (p, p), stats=stats, w3=w3))
    td = timeit(lambda: H.conv_dgrad(dy, wt, dx, (s, s), (p, p), wt3=wt3))
    tw = timeit(lambda: H.conv_wgrad(dy, x, dw, (k, k), (s, s), (p, p), workspace=ws))
    extra = ""
    if a.bnr and k >= s:
        z = torch.randn_like(x); yy = torch.randn_like(x)
        mean = torch.zeros(ci, device="cuda"); inv = torch.ones(ci, device="cuda")
        slab = torch.empty(H.bn_reduce_rows(N * h * h, (s, s)), 2, ci, device="cuda")
        ymask = None if os.environ.get("DPA_BENCH_BNR_NO_Y") == "1" else yy      # (upper bound of a 1-bit mask: y not read at all)
        tb = timeit(lambda: H.conv_dgrad(dy, wt, dx, (s, s), (p, p), addend=dx, wt3=wt3, bn_reduce=(z, ymask, mean, inv, None, None, slab)))
        gb = (dy.numel() + 4 * dx.numel()) * dx.element_size() / 1e9
        extra = f" | fused dgrad {tb*1e6:8.1f} us {gb / tb / 1e3:5.2f} TB/s (+{(tb - td)*1e6:6.1f} us)"
    tot["fprop"] += tf * cnt; tot["dgrad"] += td * cnt; tot["wgrad"] += tw * cnt
    print(f"{name:34s} {fl/1e9:8.1f} | {tf*1e6:9.1f} {fl/tf/1e12:6.0f} | {td*1e6:9.1f} {fl/td/1e12:6.0f} | {tw*1e6:9.1f} {fl/tw/1e12:6.0f}   x{cnt}{extra}")
print("totals per step (ms): " + ", ".join(f"{k} {v*1e3:.2f}" for k, v in tot.items()))

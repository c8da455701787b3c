"""TFLOP/s of the fprop kernel vs problem size (tile-quantisation / tail study): python scripts/conv_size_sweep.py fp32"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import hip_conv as H
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
def run(N, h, ci, co, k, s, p):
    ho = H.out_size(h, k, s, p)
    x = torch.randn(N, h, h, ci, device="cuda").to(dt); w = (torch.randn(co, k, k, ci, device="cuda") * 0.05).to(dt)
    y = torch.empty(N, ho, ho, co, device="cuda", dtype=dt)
    for _ in range(2): H.conv_fprop(x, w, y, (s, s), (p, p))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): H.conv_fprop(x, w, y, (s, s), (p, p))
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    fl = 2.0 * N * ho * ho * co * k * k * ci
    tiles = ((N * ho * ho + 127) // 128) * ((co + 127) // 128)
    print(f"N={N:5d} {h}x{h} {ci}->{co} k{k}: tiles(128x128)={tiles:6d} = {tiles / 256:.2f}/CU  {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF", flush=True)
for N in (64, 128, 167, 256, 334, 512, 1024, 2048):
    run(N, 14, 256, 256, 3, 1, 1)
for N in (64, 256, 1024):
    run(N, 28, 128, 512, 1, 1, 0)
for N in (32, 64, 256):
    run(N, 56, 128, 128, 3, 1, 1)

"""Achievable HBM bandwidth on this GPU with plain torch kernels (copy, add, scale) - the ceiling for the BN kernels."""
import time, torch
n = 1 << 30                                  # 4 GiB fp32
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda"); z = torch.empty(n, device="cuda")
def t(fn, bytes_, name):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:28s} {bytes_ / dt / 1e12:5.2f} TB/s", flush=True)
t(lambda: y.copy_(x), 8 * n, "copy (1R+1W)")
t(lambda: torch.add(x, y, out=z), 12 * n, "add (2R+1W)")
t(lambda: x.mul_(1.0001), 8 * n, "scale in place (1R+1W)")
t(lambda: x.sum(), 4 * n, "sum (1R)")
xb = x.view(torch.bfloat16); yb = y.view(torch.bfloat16)
t(lambda: yb.copy_(xb), 8 * n, "bf16 copy (1R+1W)")

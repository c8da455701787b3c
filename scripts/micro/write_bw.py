"""Pure-write and mixed read / write HBM bandwidth with plain torch kernels (context for the write-heavy 1x1 layers):
python scripts/micro/write_bw.py"""
import torch


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


y = torch.empty(1024, 56, 56, 256, device="cuda")
x = torch.randn(1024, 56, 56, 64, device="cuda")
gb = y.numel() * 4 / 1e9
t = timeit(lambda: y.fill_(1.0)); print(f"fill 3.29 GB (W)                    {t:8.1f} us  {gb / t * 1e3:5.2f} TB/s")
t = timeit(lambda: y.zero_()); print(f"zero 3.29 GB (W, memset path)       {t:8.1f} us  {gb / t * 1e3:5.2f} TB/s")
y4 = y.view(1024, 56, 56, 4, 64)
t = timeit(lambda: torch.add(x.unsqueeze(3), 1.0, out=y4) if False else y4.copy_(x.unsqueeze(3)))
print(f"broadcast copy 0.82 GB -> 3.29 GB   {t:8.1f} us  {(gb * 1.25) / t * 1e3:5.2f} TB/s (R 20 % / W 80 %)")
z = torch.empty_like(y)
t = timeit(lambda: z.copy_(y)); print(f"copy 3.29 GB (R 50 % / W 50 %)      {t:8.1f} us  {2 * gb / t * 1e3:5.2f} TB/s")
t = timeit(lambda: torch.add(y, z, out=z)); print(f"add in place (R 67 % / W 33 %)      {t:8.1f} us  {3 * gb / t * 1e3:5.2f} TB/s")

import ctypes, torch, time, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "distributedpytorch_amd", "lib", "libdpamd_hip.so"))
print("cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0))
n = 11_181_642
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda")
m = torch.zeros_like(p); v = torch.zeros_like(p)
from distributedpytorch_amd.nn.optim import make_hyper
hyper = make_hyper(1e-3, .9, .999, 1e-8, 0, 0, 0.9, 1.0, device="cuda")
pr = torch.nn.Parameter(p.clone()); pr.grad = g.clone()
opt = torch.optim.Adam([pr], lr=1e-3)
s = torch.cuda.current_stream().cuda_stream
V = ctypes.c_void_p
for it in range(3):
    lib.dpa_adam_step(V(p.data_ptr()), V(g.data_ptr()), V(m.data_ptr()), V(v.data_ptr()), V(0), V(hyper.data_ptr()), ctypes.c_longlong(n), V(s))
    lib.dpa_advance_step(V(hyper.data_ptr()), V(s))
    opt.step()
torch.cuda.synchronize()
print("maxdiff", (p - pr.data).abs().max().item(), "step", hyper[5].item())
t0 = time.perf_counter()
for it in range(50):
    lib.dpa_adam_step(V(p.data_ptr()), V(g.data_ptr()), V(m.data_ptr()), V(v.data_ptr()), V(0), V(hyper.data_ptr()), ctypes.c_longlong(n), V(s))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print(f"adam {dt*1e6:.1f} us  {n*28/dt/1e12:.2f} TB/s")

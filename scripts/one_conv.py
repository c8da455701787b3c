"""Run one conv kernel a few times (for rocprofv3 --pmc): python scripts/one_conv.py H Cin Cout k stride pad [mode] [dtype]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import hip_conv as H
h, ci, co, k, s, p = [int(v) for v in sys.argv[1:7]]
mode = sys.argv[7] if len(sys.argv) > 7 else "fprop"
dt = torch.bfloat16 if (len(sys.argv) <= 8 or sys.argv[8] == "bf16") else torch.float32
N = int(os.environ.get("DPA_ONE_CONV_BATCH", "256"))
ho = H.out_size(h, k, s, p)
x = torch.randn(N, h, h, ci, device="cuda").to(dt)
w = (torch.randn(co, k, k, ci, device="cuda") * 0.05).to(dt)
wt = w.permute(3, 1, 2, 0).contiguous()
y = torch.empty(N, ho, ho, co, device="cuda", dtype=dt)
dy = torch.randn(N, ho, ho, co, device="cuda").to(dt)
dx = torch.empty_like(x)
dw = torch.empty(co, k, k, ci, device="cuda")
ws = torch.empty(H.wgrad_workspace_numel(co, k * k * ci), device="cuda")
pre = os.environ.get("DPA_ONE_CONV_PRESPLIT", "0") == "1" and dt == torch.float32      # the pre-split x6 kernel (conv_x6p.hip)
w3 = H.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16)) if pre else None
wt3 = H.split3_planes(wt, torch.empty(3 * wt.numel(), device="cuda", dtype=torch.bfloat16)) if pre else None
for _ in range(3):
    if mode == "fprop": H.conv_fprop(x, w, y, (s, s), (p, p), w3=w3)
    elif mode == "dgrad": H.conv_dgrad(dy, wt, dx, (s, s), (p, p), wt3=wt3)
    else: H.conv_wgrad(dy, x, dw, (k, k), (s, s), (p, p), workspace=ws)
torch.cuda.synchronize()

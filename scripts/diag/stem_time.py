import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from distributedpytorch_amd.ops import hip_conv as H
N=1024
x = torch.zeros(N,224,224,4, device="cuda"); x[...,:3]=torch.randn(N,224,224,3, device="cuda")
w = torch.randn(64,7,7,4, device="cuda")*0.05
y = torch.empty(N,112,112,64, device="cuda")
st = torch.empty((N*112*112+63)//64, 2, 64, device="cuda")
for _ in range(3): H.conv_fprop(x,w,y,(2,2),(3,3),stats=st)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): H.conv_fprop(x,w,y,(2,2),(3,3),stats=st)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
ref = torch.nn.functional.conv2d(x[:8].permute(0,3,1,2), w.permute(0,3,1,2), stride=2, padding=3).permute(0,2,3,1)
print(os.environ.get("DPA_CONV_STAGING","auto"), "stem fprop ms", round(dt*1e3,3), "max err", (y[:8]-ref).abs().max().item())

import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from distributedpytorch_amd.ops import hip_conv as H
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/20*1e6
for dt in (torch.float32, torch.bfloat16):
    N,h,ci,co=256,56,64,256
    x=torch.randn(N,h,h,ci,device="cuda").to(dt); w=(torch.randn(co,1,1,ci,device="cuda")*0.05).to(dt)
    y=torch.empty(N,h,h,co,device="cuda",dtype=dt)
    st=torch.empty(H.stats_rows(N*h*h,co,dt),2,co,device="cuda")
    b=torch.randn(co,device="cuda")
    print(dt, "stats", round(t(lambda: H.conv_fprop(x,w,y,(1,1),(0,0),stats=st))), "nostats", round(t(lambda: H.conv_fprop(x,w,y,(1,1),(0,0)))), "bias+relu", round(t(lambda: H.conv_fprop(x,w,y,(1,1),(0,0),bias=b,relu=True))),
          "copy-equivalent", round(t(lambda: y.copy_(y))), flush=True)

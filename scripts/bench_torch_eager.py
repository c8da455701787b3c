"""Measured stand-in for the reference on MI355X: a torch-eager (MIOpen/hipBLASLt, autograd, nn.DataParallel-free)
restatement of the reference's training step - torchvision-architecture ResNet-50, CrossEntropyLoss, Adam(1e-3),
synthetic device-resident batch - at the same per-GPU batch as bench.py.  The reference itself is fp32 with no AMP
(`--dtype fp32`); `--dtype bf16` adds torch.autocast as the strongest eager configuration."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import torch_ref_models as ref

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--model", default="resnet50")
ap.add_argument("--side", type=int, default=224)
ap.add_argument("--classes", type=int, default=1000)
ap.add_argument("--channels-last", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda")
model = getattr(ref, "resnet50" if a.model == "resnet50" else "resnet18")(a.classes).to(dev)
if a.channels_last:
    model = model.to(memory_format=torch.channels_last)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
crit = torch.nn.CrossEntropyLoss()
x = torch.randn(a.batch, 3, a.side, a.side, device=dev)
if a.channels_last:
    x = x.contiguous(memory_format=torch.channels_last)
y = torch.randint(0, a.classes, (a.batch,), device=dev)
def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(a.dtype == "bf16")):
        loss = crit(model(x), y)
    loss.backward()
    opt.step()
for _ in range(a.warmup): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"torch_eager_images_per_s": round(a.batch * a.steps / dt, 1), "ms_per_step": round(1000 * dt / a.steps, 2),
                  "dtype": a.dtype, "channels_last": a.channels_last, "model": a.model, "batch": a.batch}))

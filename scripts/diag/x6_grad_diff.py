"""Per-parameter gradient difference between the exact and the x6 fp32 modes after one backward pass (ResNet-50)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.models.factory import get_model
from distributedpytorch_amd.nn.loss import Criterion
from distributedpytorch_amd.nn.optim import ArenaOptimizer
from distributedpytorch_amd.ops.native import set_fp32_mode
from distributedpytorch_amd.parallel.ddp import DistributedModel
dev = torch.device("cuda", 0)
model = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
grads = {}
for mode in ("exact", "x6", "exact_perm"):
    set_fp32_mode("x6" if mode == "x6" else "exact")
    torch.manual_seed(5)
    net, _ = get_model(model, 10, device=dev, dtype=torch.float32)
    dm = DistributedModel(net, Criterion("cross_entropy"), ArenaOptimizer(net.store, "SGD", lr=0.0))
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.zeros(32, 224, 224, net.in_channels, device=dev)
    x[..., :3] = torch.randn(32, 224, 224, 3, device=dev, generator=g)
    y = torch.randint(0, 10, (32,), device=dev, generator=g)
    if mode == "exact_perm":      # same samples in another order: mathematically identical gradients, different summation order
        perm = torch.randperm(32, device=dev, generator=g)
        x, y = x[perm].contiguous(), y[perm].contiguous()
    dm.train_step(x, y)
    torch.cuda.synchronize()
    grads[mode] = {s.name: net.store.gview(s).clone() for s in net.store.params}
    print(mode, "loss", dm.epoch_metrics()[0])
rows = []
for k in grads["exact"]:
    a, b = grads["exact"][k].double(), grads["x6"][k].double()
    rows.append(((a - b).norm().item() / (a.norm().item() + 1e-30), k, a.norm().item()))
for r in rows[-14:]:
    print(f"{r[0]:.3e}  {r[1]}  |g|={r[2]:.3e}")
print("--- exact vs exact with the batch permuted (rounding-order noise floor)")
for k in list(grads["exact"])[-14:]:
    a, b = grads["exact"][k].double(), grads["exact_perm"][k].double()
    print(f"{(a - b).norm().item() / (a.norm().item() + 1e-30):.3e}  {k}")

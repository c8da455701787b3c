"""Compare BatchNorm running statistics and eval outputs, engine vs torch eager, after train steps on full (64) and
partial (48) batches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch_ref_models as ref
from distributedpytorch_amd.models.factory import get_model
from distributedpytorch_amd.nn.loss import Criterion
from distributedpytorch_amd.nn.optim import ArenaOptimizer
from distributedpytorch_amd.parallel.ddp import DistributedModel

dev = torch.device("cuda", 0)
torch.manual_seed(3)
net, _ = get_model("resnet", 10, device=dev, dtype=torch.float32)
net.seed_rng(3)
sd0 = {k: v.clone() for k, v in net.store.state_dict().items()}
dm = DistributedModel(net, Criterion("cross_entropy"), ArenaOptimizer(net.store, "adam", lr=1e-3))
model = ref.resnet18(10).cuda()
model.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in sd0.items()}, strict=True)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
crit = torch.nn.CrossEntropyLoss()
g = torch.Generator(device=dev).manual_seed(5)


def batch(n):
    x = torch.zeros(n, 224, 224, net.in_channels, device=dev)
    x[..., :3] = torch.randn(n, 224, 224, 3, device=dev, generator=g) * (0.5 + torch.rand(n, 1, 1, 1, device=dev, generator=g))
    y = torch.randint(0, 10, (n,), device=dev, generator=g)
    return x, y


def compare(tag):
    sd = net.store.state_dict()
    worst = {}
    for k, v in model.state_dict().items():
        e = sd["module." + k] if ("module." + k) in sd else sd[k]
        if "running" in k or "num_batches" in k:
            d = (e.float() - v.float()).abs().max().item() / (v.float().abs().max().item() + 1e-6)
            kind = k.split(".")[-1]
            worst[kind] = max(worst.get(kind, 0.0), d)
    xe, ye = batch(32)
    model.eval()
    with torch.no_grad():
        ot = model(xe[..., :3].permute(0, 3, 1, 2).contiguous())
    model.train()
    dm.reset_epoch_metrics()
    oe = dm.eval_step(xe, ye)
    le = dm.epoch_metrics()[0]
    lt = float(crit(ot, ye))
    print(f"{tag}: worst relative diff of buffers {worst} | eval loss engine {le:.5f} torch {lt:.5f}", flush=True)


for step, n in enumerate([64, 64, 64, 48, 64, 48, 17, 64]):
    x, y = batch(n)
    dm.train_step(x, y)
    opt.zero_grad()
    out = model(x[..., :3].permute(0, 3, 1, 2).contiguous())
    crit(out, y).backward()
    opt.step()
    compare(f"after step {step} (batch {n})")

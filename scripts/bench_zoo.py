#!/usr/bin/env python3
"""Training-step throughput of every zoo model (reference defaults: batch 64, 224x224 / 299x299, Adam) on one MI355X:
this engine (fp32 and bf16) vs the plain-torch restatement (tests/torch_ref_models.py, fp32 eager).
  python scripts/bench_zoo.py [--batch 64] [--models resnet,alexnet,...] [--no-torch]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from distributedpytorch_amd.models.factory import get_model, get_model_input_size
from distributedpytorch_amd.nn.loss import Criterion
from distributedpytorch_amd.nn.optim import ArenaOptimizer
from distributedpytorch_amd.parallel.ddp import DistributedModel

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--models", default="resnet,alexnet,vgg,squeezenet,densenet,inception")
ap.add_argument("--no-torch", action="store_true")
ap.add_argument("--fp32-only", action="store_true", help="only the default fp32 eager number (knob sweeps)")
a = ap.parse_args()

def timeit(step, n):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def engine(name, dtype, graph):
    side = 224 if name == 'resnet50' else get_model_input_size(name)
    net, _ = get_model(name, 10, device="cuda", dtype=dtype)
    dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3))
    x = torch.zeros(a.batch, side, side, net.in_channels, device="cuda", dtype=dtype)
    x[..., :3] = torch.randn(a.batch, side, side, 3, device="cuda").to(dtype)
    y = torch.randint(0, 10, (a.batch,), device="cuda")
    if graph:
        dm.capture_train_step(x, y)
        t = timeit(lambda: dm.replay_train_step(x, y), a.steps)
    else:
        t = timeit(lambda: dm.train_step(x, y), a.steps)
    return a.batch / t

def torch_eager(name):
    import torch_ref_models as ref
    side = get_model_input_size(name)
    ctor = {"resnet": ref.resnet18, "alexnet": ref.AlexNet, "vgg": ref.VGG11BN, "squeezenet": ref.SqueezeNet10,
            "densenet": ref.DenseNet121, "inception": ref.Inception3}[name]
    m = ctor(10).cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    x = torch.randn(a.batch, 3, side, side, device="cuda"); y = torch.randint(0, 10, (a.batch,), device="cuda")
    crit = torch.nn.CrossEntropyLoss()
    def step():
        opt.zero_grad()
        out = m(x)
        loss = crit(out[0], y) + 0.4 * crit(out[1], y) if isinstance(out, tuple) else crit(out, y)
        loss.backward(); opt.step()
    return a.batch / timeit(step, a.steps)

for name in a.models.split(","):
    rec = {"model": name, "batch": a.batch}
    rec["engine_fp32"] = round(engine(name, torch.float32, False), 1)
    if a.fp32_only:
        print(json.dumps(rec), flush=True)
        torch.cuda.empty_cache()
        continue
    rec["engine_fp32_graph"] = round(engine(name, torch.float32, True), 1)
    from distributedpytorch_amd.ops.native import set_fp32_mode
    prev = set_fp32_mode("x6")          # fp32 products on the bf16 matrix pipe (opt-in mode)
    try:
        rec["engine_fp32_x6_graph"] = round(engine(name, torch.float32, True), 1)
    finally:
        set_fp32_mode(prev)
    rec["engine_bf16_graph"] = round(engine(name, torch.bfloat16, True), 1)
    if not a.no_torch:
        rec["torch_eager_fp32"] = round(torch_eager(name), 1)
    print(json.dumps(rec), flush=True)
    torch.cuda.empty_cache()

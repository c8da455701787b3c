"""ResNet-50, 16 images of 64x64: per-parameter gradient distance to an fp64 torch oracle for (a) this engine (fp32 x6 /
exact), (b) torch's own fp32 on the same GPU.  Context for scripts/diag/x6s_amplification.py: is the 1-2 % sensitivity
of conv1.weight's gradient to last-bit changes of the BatchNorm statistics a property of the network or of the engine?
python scripts/diag/resnet50_grad_vs_fp64.py [resnet50|resnet18] [batch] [side]   (defaults: resnet50 16 64; the fp64 oracle
runs on the GPU when the problem is larger than that)"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.models.factory import get_model  # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode  # noqa: E402
from tests import torch_ref_models as ref  # noqa: E402
from tests.helpers import framework_step, grads_by_name, oracle_step  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
side = int(sys.argv[3]) if len(sys.argv) > 3 else 64
big = batch * side * side > 16 * 64 * 64
torch.manual_seed(5)
oracle = ref.resnet50(10) if which == "resnet50" else ref.resnet18(10)
x = torch.randn(batch, 3, side, side); y = torch.randint(0, 10, (batch,))
o64 = copy.deepcopy(oracle).double()
if big:
    o64 = o64.cuda()
    oracle_step(o64, x.double().cuda(), y.cuda())
else:
    oracle_step(o64, x.double(), y)
g64 = {k: p.grad.float().cpu() for k, p in o64.named_parameters()}
del o64
gmax = max(g.norm().item() for g in g64.values())
o32 = copy.deepcopy(oracle).float().cuda()
oracle_step(o32, x.cuda(), y.cuda())
g32 = {k: p.grad.detach().cpu() for k, p in o32.named_parameters()}
g32c = None
if not big:
    o32c = copy.deepcopy(oracle).float()
    oracle_step(o32c, x, y)
    g32c = {k: p.grad.detach() for k, p in o32c.named_parameters()}


def top(tag, g):
    rows = sorted((((g[k] - v).norm().item() / gmax, k, v.norm().item() / gmax) for k, v in g64.items()), reverse=True)[:3]
    ga = torch.cat([g[k].float().cpu().flatten() for k in g64]); gb = torch.cat([v.flatten() for v in g64.values()])
    print(f"{tag}: all gradients |g - g64| / |g64| = {(ga - gb).norm().item() / gb.norm().item():.3e}")
    for d, k, n in rows:
        print(f"      {k:32s} |g - g64| / gmax {d:.2e}   |g64| / gmax {n:.2e}")


top("torch fp32 (GPU) vs fp64", g32)
if g32c is not None:
    top("torch fp32 (CPU) vs fp64", g32c)
for mode in ("x6", "exact"):
    set_fp32_mode(mode)
    net, _ = get_model("resnet50" if which == "resnet50" else "resnet", 10, device="cuda", dtype=torch.float32)
    net.load_state_dict(oracle.state_dict())
    framework_step(net, x, y)
    top(f"engine fp32 {mode} vs fp64", grads_by_name(net))

"""ResNet-50, 16 images of 64x64: per-parameter gradient distance to an fp64 torch oracle for (a) this engine (fp32 x6 /
exact), (b) torch's own fp32 on the same GPU.  Context for scripts/diag/x6s_amplification.py: is the 1-2 % sensitivity
of conv1.weight's gradient to last-bit changes of the BatchNorm statistics a property of the network or of the engine?
python scripts/diag/resnet50_grad_vs_fp64.py"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.models.factory import get_model  # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode  # noqa: E402
from tests import torch_ref_models as ref  # noqa: E402
from tests.helpers import framework_step, grads_by_name, oracle_step  # noqa: E402

torch.manual_seed(5)
oracle = ref.resnet50(10)
x = torch.randn(16, 3, 64, 64); y = torch.randint(0, 10, (16,))
o64 = copy.deepcopy(oracle).double()
oracle_step(o64, x.double(), y)
g64 = {k: p.grad.float() for k, p in o64.named_parameters()}
gmax = max(g.norm().item() for g in g64.values())
o32 = copy.deepcopy(oracle).float().cuda()
oracle_step(o32, x.cuda(), y.cuda())
g32 = {k: p.grad.detach().cpu() for k, p in o32.named_parameters()}
o32c = copy.deepcopy(oracle).float()
oracle_step(o32c, x, y)
g32c = {k: p.grad.detach() for k, p in o32c.named_parameters()}


def top(tag, g):
    rows = sorted((((g[k] - v).norm().item() / gmax, k, v.norm().item() / gmax) for k, v in g64.items()), reverse=True)[:3]
    print(tag)
    for d, k, n in rows:
        print(f"      {k:32s} |g - g64| / gmax {d:.2e}   |g64| / gmax {n:.2e}")


top("torch fp32 (GPU) vs fp64", g32)
top("torch fp32 (CPU) vs fp64", g32c)
for mode in ("x6", "exact"):
    set_fp32_mode(mode)
    net, _ = get_model("resnet50", 10, device="cuda", dtype=torch.float32)
    net.load_state_dict(oracle.state_dict())
    framework_step(net, x, y)
    top(f"engine fp32 {mode} vs fp64", grads_by_name(net))

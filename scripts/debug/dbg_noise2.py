import sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.models.factory import get_model
from tests import torch_ref_models as ref
from tests.helpers import framework_step, grads_by_name, oracle_step
oracle = ref.resnet18(10)
torch.manual_seed(0)
net, _ = get_model("resnet", 10, device="cuda")
net.load_state_dict(oracle.state_dict())
x = torch.randn(16, 3, 64, 64); y = torch.randint(0, 10, (16,))
o64 = copy.deepcopy(oracle).double(); oracle_step(o64, x.double(), y)
ref_g = {k: p.grad for k, p in o64.named_parameters()}
prev = None
for it in range(3):
    logits, loss, acc = framework_step(net, x, y)
    g = grads_by_name(net)
    errs = {k: (g[k].double() - v).abs().max().item() / max(v.abs().max().item(), 1e-9) for k, v in ref_g.items()}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(it, loss, [(k, f"{e:.1e}") for k, e in worst])
    if prev is not None:
        print("  run-to-run max diff", max((g[k] - prev[k]).abs().max().item() for k in g))
    prev = g

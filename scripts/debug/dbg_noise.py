import sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.models.factory import get_model
from tests import torch_ref_models as ref
from tests.helpers import framework_step, grads_by_name, oracle_step
torch.manual_seed(0)
oracle = ref.resnet18(10)
net, _ = get_model("resnet", 10, device="cuda")
net.load_state_dict(oracle.state_dict())
x = torch.randn(16, 3, 64, 64); y = torch.randint(0, 10, (16,))
framework_step(net, x, y); oracle_step(oracle, x, y)
o64 = copy.deepcopy(oracle).double(); oracle_step(o64, x.double(), y)
ogpu = copy.deepcopy(oracle).cuda(); oracle_step(ogpu, x.cuda(), y.cuda())
g = grads_by_name(net)
for k, p in list(o64.named_parameters())[::6]:
    v = p.grad; s = max(v.abs().max().item(), 1e-9)
    e1 = (g[k].double() - v).abs().max().item() / s
    e2 = (dict(oracle.named_parameters())[k].grad.double() - v).abs().max().item() / s
    e3 = (dict(ogpu.named_parameters())[k].grad.double().cpu() - v).abs().max().item() / s
    print(f"{k:36s} mine {e1:.2e} torch-cpu32 {e2:.2e} torch-gpu32 {e3:.2e}")

"""Per-parameter gradient error of the bf16 engine against an fp64 torch oracle, next to the error of torch's OWN bf16
execution (GPU autocast) of the same model on the same batch - the noise floor any bf16 implementation has.
Prints one line per parameter; used to set the bound of tests/test_gpu_models.py (bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.models.factory import get_model      # noqa: E402
from tests import torch_ref_models as ref                        # noqa: E402
from tests.helpers import framework_step, grads_by_name, oracle_step   # noqa: E402

n, hw = int(os.environ.get("N", 16)), int(os.environ.get("HW", 64))
torch.manual_seed(0)
oracle = ref.resnet18(10)
net, _ = get_model("resnet", 10, device="cuda", dtype=torch.bfloat16)
net.load_state_dict(oracle.state_dict())
x = torch.randn(n, 3, hw, hw)
y = torch.randint(0, 10, (n,))
framework_step(net, x, y)
g = grads_by_name(net)
# torch bf16 autocast on the GPU
import copy
tb = copy.deepcopy(oracle).cuda()
tb.train()
with torch.autocast("cuda", dtype=torch.bfloat16):
    loss = torch.nn.functional.cross_entropy(tb(x.cuda()), y.cuda())
loss.backward()
gt = {k: p.grad.float().cpu() for k, p in tb.named_parameters()}
o64 = oracle.double()
oracle_step(o64, x.double(), y)
worst = (0, None)
for k, p in o64.named_parameters():
    v = p.grad.float()
    e1 = (g[k] - v).norm().item() / max(v.norm().item(), 1e-12)
    e2 = (gt[k] - v).norm().item() / max(v.norm().item(), 1e-12)
    pj = lambda a: (a.flatten().double() @ v.flatten().double()).item() / (v.flatten().double() @ v.flatten().double()).item()   # noqa: E731
    print(f"{k:34s} |g|={v.norm().item():.3e}  engine-bf16 {e1:.3f} (proj {pj(g[k]):.3f})   torch-bf16-autocast {e2:.3f} (proj {pj(gt[k]):.3f})")

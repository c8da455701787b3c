import sys, os, copy, torch
sys.path.insert(0, os.getcwd())
from distributedpytorch_amd.models.factory import get_model
from distributedpytorch_amd.ops.native import set_fp32_mode
from tests import torch_ref_models as ref
from tests.helpers import framework_step, grads_by_name, oracle_step
from tests.test_models_cpu import _no_dropout, _zero_dropout
for seed in (1,):
    for mode in ("x6", "exact"):
        set_fp32_mode(mode)
        torch.manual_seed(seed)
        oracle = _no_dropout(ref.DenseNet121(10))
        net, _ = get_model("densenet", 10, device="cuda", dtype=torch.float32)
        _zero_dropout(net)
        net.load_state_dict(oracle.state_dict())
        x = torch.randn(8, 3, 64, 64); y = torch.randint(0, 10, (8,))
        logits, loss, acc = framework_step(net, x, y)
        o64 = copy.deepcopy(oracle).double()
        rl, rloss = oracle_step(o64, x.double(), y)
        g = grads_by_name(net)
        o32 = copy.deepcopy(oracle).float().cuda()
        oracle_step(o32, x.cuda(), y.cuda())
        g32 = {k: q.grad.detach().cpu() for k, q in o32.named_parameters()}
        worst = []
        for k, p in o64.named_parameters():
            v = p.grad.float()
            e = (g[k] - v).norm().item() / max(v.norm().item(), 1e-12)
            e32 = (g32[k] - v).norm().item() / max(v.norm().item(), 1e-12)
            worst.append((e, e32, k))
        worst.sort(reverse=True)
        print(mode, "loss", loss, float(rloss), "logit err", (logits - rl.float()).abs().max().item())
        for w in worst[:6]: print("   ", w)
        # element-wise look at the worst parameter
        k = worst[0][2]
        d = (g[k] - dict(o64.named_parameters())[k].grad.float()).abs()
        print("   worst param elementwise: max", d.max().item(), "n>10%max", int((d > 0.1 * d.max()).sum()), "of", d.numel())

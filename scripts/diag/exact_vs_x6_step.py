"""One ResNet training step in both fp32 arithmetic modes (exact: conv_igemm.hip everywhere; x6: conv_x6p / conv_x6s /
conv_wgrad_x6q): the two kernel families share no epilogue code, so agreement of every gradient to fp32 rounding level
cross-checks both.  python scripts/diag/exact_vs_x6_step.py [model] [batch] [side]"""
import sys

import torch

sys.path.insert(0, ".")
from distributedpytorch_amd.models.factory import get_model      # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode        # noqa: E402
from tests.helpers import framework_step, grads_by_name            # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "resnet"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
side = int(sys.argv[3]) if len(sys.argv) > 3 else 224
torch.manual_seed(3)
x = torch.randn(batch, 3, side, side)
y = torch.randint(0, 10, (batch,))
res = {}
for mode in ("exact", "x6"):
    prev = set_fp32_mode(mode)
    torch.manual_seed(5)
    net, _ = get_model(model, 10, device="cuda", dtype=torch.float32)
    logits, loss, _ = framework_step(net, x, y)
    res[mode] = (logits.double(), loss, {k: v.double().clone() for k, v in grads_by_name(net).items()})
    set_fp32_mode(prev)
    del net
a, b = res["exact"], res["x6"]
print(f"loss exact {a[1]!r} x6 {b[1]!r}; logits max |d| {(a[0] - b[0]).abs().max().item():.3e} of {a[0].abs().max().item():.3e}")
ga = torch.cat([v.flatten() for v in a[2].values()]); gb = torch.cat([b[2][k].flatten() for k in a[2]])
print(f"all gradients: 1 - cos = {1 - torch.dot(ga, gb).item() / (ga.norm().item() * gb.norm().item()):.3e}, |d| / |g| = {(ga - gb).norm().item() / ga.norm().item():.3e}")
worst = sorted(((a[2][k] - b[2][k]).norm().item() / max(a[2][k].norm().item(), 1e-30), k) for k in a[2])[-6:]
for w, k in worst:
    print(f"   {k:40s} |d| / |g| = {w:.3e}")

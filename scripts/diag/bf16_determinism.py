"""Is one bf16 ResNet-50 training step bit-reproducible at bench scale?  Two fresh models (same seed, same batch), one
step each; lists the parameters whose gradients differ.  python scripts/diag/bf16_determinism.py [batch] [side] [dtype] [model]"""
import sys

import torch

sys.path.insert(0, ".")
from distributedpytorch_amd.models.factory import get_model      # noqa: E402
from tests.helpers import framework_step, grads_by_name            # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
side = int(sys.argv[2]) if len(sys.argv) > 2 else 224
dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
model = sys.argv[4] if len(sys.argv) > 4 else "resnet50"
torch.manual_seed(3)
x = torch.randn(batch, 3, side, side)
y = torch.randint(0, 1000, (batch,))
runs = []
for r in range(3):
    torch.manual_seed(5)
    net, _ = get_model(model, 1000, device="cuda", dtype=dtype)
    logits, loss, _ = framework_step(net, x, y)
    runs.append((logits.clone(), loss, {k: v.clone() for k, v in grads_by_name(net).items()}))
    del net
    torch.cuda.synchronize()
for r in (1, 2):
    same_logits = torch.equal(runs[0][0], runs[r][0])
    diff = [k for k in runs[0][2] if not torch.equal(runs[0][2][k], runs[r][2][k])]
    print(f"run {r} vs 0: logits equal {same_logits}, loss {runs[0][1]!r} / {runs[r][1]!r}, {len(diff)} of {len(runs[0][2])} gradients differ")
    for k in diff[-12:]:
        a, b = runs[0][2][k].float(), runs[r][2][k].float()
        print(f"   {k:40s} max |d| {(a - b).abs().max().item():.3e} of {a.abs().max().item():.3e}")

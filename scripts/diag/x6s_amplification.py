"""How far do two ResNet-50 training steps on 16 images drift apart when only the ORDER in which the BatchNorm statistics
are summed changes (conv_x6s.hip: per block; conv_x6p.hip: per 128-row tile, or per 64-row tile with the short tile)?
Fresh model with the same seed for every run.  python scripts/diag/x6s_amplification.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.models.factory import get_model  # noqa: E402
from distributedpytorch_amd.ops.native import hip_lib, set_fp32_mode  # noqa: E402
from tests.helpers import framework_step, grads_by_name  # noqa: E402

lib = hip_lib()
set_fp32_mode("x6")
lib.dpa_x6s_knob(2, 16); lib.dpa_x6p_min_tiles(1)


def run(x6s, narrow, short, steps=1):
    torch.manual_seed(5)
    net, _ = get_model("resnet50", 10, device="cuda", dtype=torch.float32)
    x = torch.randn(16, 3, 64, 64); y = torch.randint(0, 10, (16,))
    lib.dpa_x6s_knob(1, x6s); lib.dpa_x6p_knob(1, narrow); lib.dpa_x6p_knob(2, short)
    for _ in range(steps):
        net.store.g.zero_()
        logits, loss, _ = framework_step(net, x, y)
    return logits, loss, grads_by_name(net)


def report(tag, a, b):
    gmax = max(g.norm().item() for g in b[2].values())
    rows = sorted((((a[2][k] - g).norm().item() / gmax, k, g.norm().item() / gmax) for k, g in b[2].items()), reverse=True)[:3]
    ga = torch.cat([a[2][k].double().flatten() for k in b[2]]); gb = torch.cat([g.double().flatten() for g in b[2].values()])
    cos = torch.dot(ga, gb).item() / (ga.norm().item() * gb.norm().item())
    print(f"{tag:44s} loss {a[1]:.6f} / {b[1]:.6f}  max logit diff {(a[0] - b[0]).abs().max().item():.2e}  1 - cos {1 - cos:.2e}")
    for d, k, n in rows:
        print(f"      {k:32s} |diff| / gmax {d:.2e}   |g| / gmax {n:.2e}")


base = run(0, 0, 0)
report("x6p wide, same run again", run(0, 0, 0), base)
xs = run(1, 0, 0)
sh = run(0, 1 << 30, 1 << 30)
st2 = run(0, 0, 0, steps=2)
report("x6s vs x6p wide", xs, base)
report("x6p short tiles vs x6p wide", sh, base)
report("x6p wide, second step vs first step", st2, base)
report("x6s vs x6p short tiles", xs, sh)
report("x6s vs x6p wide second step", xs, st2)
report("x6s second step vs x6p wide second step", run(1, 0, 0, steps=2), st2)

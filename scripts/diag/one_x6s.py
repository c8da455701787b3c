"""One 1x1 convolution on 64 input channels a few times (for rocprofv3 --pmc): python scripts/diag/one_x6s.py [Cout] [mode]
mode: fprop (default) | copy (a plain device copy of the same number of output bytes, for comparison)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributedpytorch_amd.ops import hip_conv  # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode  # noqa: E402

co = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "fprop"
N = int(os.environ.get("BATCH", "1024"))
set_fp32_mode("x6")
x = torch.randn(N, 56, 56, 64, device="cuda")
w = torch.randn(co, 1, 1, 64, device="cuda") * 0.125
y = torch.empty(N, 56, 56, co, device="cuda")
w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
for _ in range(3):
    if mode == "fprop":
        hip_conv.conv_fprop(x, w, y, (1, 1), (0, 0), w3=w3)
    else:
        y.fill_(1.0)
torch.cuda.synchronize()

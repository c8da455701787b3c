"""Repro for the run-to-run difference of the fused BatchNorm statistics of the bf16 register-staged convolution kernel:
the same convolution N times, slabs compared with the first.  python scripts/diag/stats_race.py [reps] [dtype]"""
import sys

import torch

sys.path.insert(0, ".")
from distributedpytorch_amd.ops import hip_conv      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
g = torch.Generator(device="cuda").manual_seed(1)
for (n, h, c, k, ks, st, pd) in ((64, 56, 64, 256, 1, 1, 0), (64, 224, 8, 64, 7, 2, 3), (64, 56, 64, 64, 3, 1, 1), (64, 28, 512, 128, 1, 1, 0)):
    x = torch.randn(n, h, h, c, device="cuda", generator=g).to(dtype)
    w = (torch.randn(k, ks, ks, c, device="cuda", generator=g) * (ks * ks * c) ** -0.5).to(dtype)
    ho = hip_conv.out_size(h, ks, st, pd)
    rows = hip_conv.stats_rows(n * ho * ho, k, dtype)
    w3 = None
    if dtype == torch.float32 and c % 32 == 0:        # the pre-split kernels (conv_x6p.hip / conv_x6s.hip) in x6 mode
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
    first, bad, bad_rows = None, 0, []
    for r in range(reps):
        y = torch.empty(n, ho, ho, k, device="cuda", dtype=dtype)
        slab = torch.zeros(rows, 2, k, device="cuda")
        hip_conv.conv_fprop(x, w, y, (st, st), (pd, pd), stats=slab, w3=w3)
        torch.cuda.synchronize()
        if first is None:
            first = (y.clone(), slab.clone())
        else:
            assert torch.equal(y, first[0])
            if not torch.equal(slab, first[1]):
                bad += 1
                d = (slab - first[1]).nonzero()
                bad_rows.append((d[0].tolist(), d.shape[0]))
    print(f"{h}x{h} {c}->{k} k{ks} s{st}: {bad} of {reps - 1} repeats differ from the first", bad_rows[:6])

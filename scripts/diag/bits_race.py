"""Launch-to-launch check of the other shuffle + masked-store loops (see conv_igemm.hip's statistics epilogue for why):
bn_apply's ReLU mask bits against (y > 0) recomputed from the stored y, at training sizes, many launches; and the
max-pool backward's fused BatchNorm slab.  python scripts/diag/bits_race.py [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from distributedpytorch_amd.ops import hip_ops      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator(device="cuda").manual_seed(2)
for dtype in (torch.float32, torch.bfloat16):
    for (m, c) in ((64 * 56 * 56, 256), (64 * 28 * 28, 512), (64 * 7 * 7, 2048)):
        x = torch.randn(m, 1, 1, c, device="cuda", generator=g).to(dtype)
        res = torch.randn(m, 1, 1, c, device="cuda", generator=g).to(dtype)
        scale = torch.rand(c, device="cuda", generator=g) + 0.5
        shift = torch.randn(c, device="cuda", generator=g) * 0.3
        bad = 0
        for r in range(reps):
            y = torch.empty_like(x)
            bits = torch.full((m, c // 32), -1, device="cuda", dtype=torch.int32)
            hip_ops.bn_apply(x, scale, shift, y, residual=res, relu=True, mask_bits=bits)
            pos = (y.reshape(m, c // 32, 32) > 0).to(torch.int64)
            words = (pos << torch.arange(32, dtype=torch.int64, device="cuda")).sum(-1)
            want = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
            bad += int(not torch.equal(bits, want))
        print(f"bn_apply bits {str(dtype)[6:]} [{m}, {c}]: {bad} of {reps} launches wrong")
# max-pool backward + stem BatchNorm slab
n, h, c = 64, 112, 64
for dtype in (torch.float32, torch.bfloat16):
    z = torch.randn(n, h, h, c, device="cuda", generator=g).to(dtype)
    scale = torch.rand(c, device="cuda", generator=g) + 0.5
    shift = torch.randn(c, device="cuda", generator=g) * 0.3
    mean = torch.randn(c, device="cuda", generator=g) * 0.1
    invstd = torch.rand(c, device="cuda", generator=g) + 0.5
    yp = torch.empty(n, h // 2, h // 2, c, device="cuda", dtype=dtype)
    idx = torch.empty(n, h // 2, h // 2, c, device="cuda", dtype=torch.uint8)
    hip_ops.maxpool_fwd(z, yp, idx, (3, 3), (2, 2), (1, 1), scale=scale, shift=shift)
    dyp = torch.randn(n, h // 2, h // 2, c, device="cuda", generator=g).to(dtype)
    first, bad = None, 0
    for r in range(reps):
        dx = torch.empty_like(z)
        slab = torch.zeros(2048, 2, c, device="cuda")
        rows = hip_ops.maxpool_bwd_bn(dyp, idx, dx, (3, 3), (2, 2), (1, 1), z, scale, shift, mean, invstd, slab)
        if first is None:
            first = (dx.clone(), slab.clone())
        else:
            bad += int(not (torch.equal(dx, first[0]) and torch.equal(slab, first[1])))
    print(f"maxpool_bwd_bn {str(dtype)[6:]}: {bad} of {reps - 1} repeats differ ({rows} slab rows)")

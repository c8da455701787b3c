"""Compare two dumps of scripts/diag/bf16_first_divergence.py (5th argument): first record whose output checksum or
statistics slab differs.  python scripts/diag/compare_dumps.py a.pt b.pt"""
import sys

import torch

a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for i, (x, y) in enumerate(zip(a, b)):
    same_out = x[1] == y[1] and x[2] == y[2]
    same_aux = x[3] is None or torch.equal(x[3], y[3])
    if not (same_out and same_aux):
        print(f"record {i} ({x[0]}): output checksum equal {same_out}, aux equal {same_aux}")
        if not same_aux:
            d = (x[3].double() - y[3].double())
            nz = d.nonzero()
            print("   differing aux entries:", nz.shape[0], "of", d.numel(), "first", nz[:5].tolist(), "max rel",
                  (d.abs() / x[3].double().abs().clamp_min(1e-30)).max().item())
            for ix in nz[:5].tolist():
                print("     ", ix, repr(x[3][tuple(ix)].item()), repr(y[3][tuple(ix)].item()))
        break
else:
    print("all", len(a), "records equal")

"""Which layer output differs first between two identical bf16 ResNet-50 forward passes (fresh model, same seed, same
batch)?  Records the output of every ConvBNAct / ConvBNReLUPool and the raw convolution outputs (forward_raw).
python scripts/diag/bf16_first_divergence.py [batch] [side] [runs]"""
import sys

import torch

sys.path.insert(0, ".")
from distributedpytorch_amd.models.factory import get_model      # noqa: E402
from distributedpytorch_amd.nn import layers                      # noqa: E402
from tests.helpers import framework_step                          # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
side = int(sys.argv[2]) if len(sys.argv) > 2 else 224
nruns = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.manual_seed(3)
x = torch.randn(batch, 3, side, side)
y = torch.randint(0, 1000, (batch,))
rec = None
orig_raw = layers.Conv2d.forward_raw
orig_apply = layers.ops.bn_apply
orig_pool = layers.ops.maxpool_fwd


def raw(self, ctx, xin, *a, **kw):
    out = orig_raw(self, ctx, xin, *a, **kw)
    st = kw.get("stats")
    rec.append(("conv %dx%d %d->%d s%d in%s" % (self.ksize[0], self.ksize[1], self.cin, self.cout, self.stride[0], tuple(xin.shape)),
                out.clone(), None if st is None else st[: self.last_stats_rows].clone()))
    return out


def apply(z, scale, shift, out, **kw):
    r = orig_apply(z, scale, shift, out, **kw)
    rec.append(("bn_apply", out.clone(), torch.stack([scale.clone(), shift.clone()])))
    return r


def pool(xin, yout, idx, *a, **kw):
    r = orig_pool(xin, yout, idx, *a, **kw)
    rec.append(("maxpool", yout.clone(), idx.clone()))
    return r


layers.Conv2d.forward_raw = raw
layers.ops.bn_apply = apply
layers.ops.maxpool_fwd = pool
runs = []
for r in range(nruns):
    torch.manual_seed(5)
    net, _ = get_model("resnet50", 1000, device="cuda", dtype=torch.float32 if len(sys.argv) > 4 and sys.argv[4] == "fp32" else torch.bfloat16)
    rec = []
    _, loss, _ = framework_step(net, x, y)
    torch.cuda.synchronize()
    from tests.helpers import grads_by_name
    for k, gk in grads_by_name(net).items():      # (as records too: checksums in a dump, tensors in the in-process comparison)
        rec.append(("grad " + k, gk.clone(), None))
    rec.append(("loss", torch.tensor([loss], dtype=torch.float64), None))
    runs.append(rec)
    del net
if len(sys.argv) > 5:        # cross-build comparison: dump run 0's slabs and output checksums, compare with compare_dumps()
    torch.save([(a[0], a[1].double().sum().item(), a[1].double().abs().sum().item(), None if a[2] is None else a[2].cpu())
                for a in runs[0]], sys.argv[5])
    print("dumped", len(runs[0]), "records to", sys.argv[5])
for r in range(1, nruns):
    for i, (a, b) in enumerate(zip(runs[0], runs[r])):
        same_out = torch.equal(a[1], b[1])
        same_aux = a[2] is None or torch.equal(a[2], b[2])
        if not (same_out and same_aux):
            d = (a[1].float() - b[1].float())
            nz = d.nonzero()
            print(f"run {r}: first difference at record {i} ({a[0]}): output equal {same_out}, aux equal {same_aux}; "
                  f"{nz.shape[0]} of {d.numel()} elements differ, max |d| {d.abs().max().item():.3e}")
            if nz.shape[0]:
                print("   first / last differing index:", nz[0].tolist(), nz[-1].tolist())
                rows = torch.unique(nz[:, :-1], dim=0)
                print("   differing pixels:", rows.shape[0], " channels:", torch.unique(nz[:, -1]).tolist()[:40])
            if a[2] is not None and not same_aux:
                da = (a[2].float() - b[2].float())
                print("   aux: differing entries", da.nonzero().shape[0], "of", da.numel(), "first", da.nonzero()[:4].tolist())
                for ix in da.nonzero()[:6].tolist():
                    print("      ", ix, repr(a[2][tuple(ix)].item()), repr(b[2][tuple(ix)].item()), " sum entry:",
                          repr(a[2][ix[0], 0, ix[2]].item()) if a[2].dim() == 3 else "")
            break
    else:
        print(f"run {r}: all {len(runs[0])} records equal")

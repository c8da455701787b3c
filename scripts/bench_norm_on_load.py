"""Experiment behind docs/KERNELS.md §4: what does folding BatchNorm-apply + ReLU into the consumer convolution's operand
path ("normalise-on-load") cost in the fp32 x6p forward kernel?  Times the ResNet-50 layers that follow a BN+ReLU with a
single consumer (conv2 after bn1, conv3 after bn2) with and without the transform, next to the bn_apply pass the fold
would remove.  The 1x1 case is also checked numerically (padding is not handled by the experiment kernel).
  python scripts/bench_norm_on_load.py [--batch 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd import ops  # noqa: E402
from distributedpytorch_amd.ops import hip_conv  # noqa: E402
from distributedpytorch_amd.ops.native import set_fp32_mode  # noqa: E402

# (hw_in, cin, cout, k, stride, count per step)   conv2 (3x3) and conv3 (1x1) of every bottleneck
LAYERS = [(56, 64, 64, 3, 1, 3), (56, 64, 256, 1, 1, 3), (56, 128, 128, 3, 2, 1), (28, 128, 128, 3, 1, 3), (28, 128, 512, 1, 1, 4),
          (28, 256, 256, 3, 2, 1), (14, 256, 256, 3, 1, 5), (14, 256, 1024, 1, 1, 6), (14, 512, 512, 3, 2, 1), (7, 512, 512, 3, 1, 2),
          (7, 512, 2048, 1, 1, 3)]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    a = ap.parse_args()
    set_fp32_mode("x6")
    N = a.batch
    tot = [0.0, 0.0, 0.0]
    print(f"{'layer':26s} | conv us | conv+fold us | bn_apply us (removed by the fold) | net per step us")
    for hw, ci, co, k, s, cnt in LAYERS:
        pd = k // 2
        z = torch.randn(N, hw, hw, ci, device="cuda")
        w = torch.randn(co, k, k, ci, device="cuda") * (k * k * ci) ** -0.5
        ho = hip_conv.out_size(hw, k, s, pd)
        y = torch.empty(N, ho, ho, co, device="cuda")
        act = torch.empty_like(z)
        sc = torch.rand(ci, device="cuda") + 0.5
        sh = torch.randn(ci, device="cuda") * 0.3
        w3 = hip_conv.split3_planes(w, torch.empty(3 * w.numel(), device="cuda", dtype=torch.bfloat16))
        t0 = timeit(lambda: hip_conv.conv_fprop(z, w, y, (s, s), (pd, pd), w3=w3))
        t1 = timeit(lambda: hip_conv.conv_fprop(z, w, y, (s, s), (pd, pd), w3=w3, _nol=(sc, sh)))
        t2 = timeit(lambda: ops.bn_apply(z, sc, sh, act, relu=True))
        if k == 1:
            ops.bn_apply(z, sc, sh, act, relu=True)
            ref = torch.empty_like(y)
            hip_conv.conv_fprop(act, w, ref, (s, s), (pd, pd), w3=w3)
            hip_conv.conv_fprop(z, w, y, (s, s), (pd, pd), w3=w3, _nol=(sc, sh))
            err = (y - ref).abs().max().item()
            assert err < 1e-4 * ref.abs().max().item(), err
        net = (t1 - t0 - t2) * cnt
        tot[0] += t0 * cnt; tot[1] += t1 * cnt; tot[2] += t2 * cnt
        print(f"{hw}x{hw} {ci}->{co} k{k} s{s} x{cnt:<2d}     | {t0:8.1f} | {t1:8.1f} | {t2:8.1f} | {net:+9.1f}")
    print(f"per step (ms): conv {tot[0] / 1e3:.2f}, conv with fold {tot[1] / 1e3:.2f} (+{(tot[1] - tot[0]) / 1e3:.2f}), "
          f"bn_apply removed {tot[2] / 1e3:.2f}  -> net {(tot[1] - tot[0] - tot[2]) / 1e3:+.2f} ms in the forward pass; wgrad reads the "
          f"same activation and would need the same transform (or the activation materialised anyway)")


if __name__ == "__main__":
    main()

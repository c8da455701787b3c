"""Accuracy of the fp32 conv paths against an fp64 reference: run once per DPA_FP32_MODE (exact | x6)."""
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import hip_conv

SHAPES = [(64, 56, 56, 64, 64, 3, 1, 1), (64, 14, 14, 256, 256, 3, 1, 1), (64, 14, 14, 1024, 256, 1, 1, 0),
          (64, 28, 28, 128, 128, 3, 2, 1), (32, 7, 7, 512, 512, 3, 1, 1)]
print("mode", os.environ.get("DPA_FP32_MODE", "exact"))
for (N, H, W, C, K, k, s, pd) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(N, H, W, C, device="cuda", generator=g)
    w = torch.randn(K, k, k, C, device="cuda", generator=g) * (k * k * C) ** -0.5
    Ho, Wo = hip_conv.out_size(H, k, s, pd), hip_conv.out_size(W, k, s, pd)
    y = torch.empty(N, Ho, Wo, K, device="cuda")
    hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd))
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), stride=s, padding=pd).permute(0, 2, 3, 1)
    err = (y.double() - ref)
    dy = torch.randn(N, Ho, Wo, K, device="cuda", generator=g)
    wt = w.permute(3, 1, 2, 0).contiguous()
    dx = torch.empty(N, H, W, C, device="cuda")
    hip_conv.conv_dgrad(dy, wt, dx, (s, s), (pd, pd))
    refdx = torch.nn.grad.conv2d_input((N, C, H, W), w.permute(0, 3, 1, 2).double(), dy.permute(0, 3, 1, 2).double(),
                                       stride=s, padding=pd).permute(0, 2, 3, 1)
    e2 = dx.double() - refdx
    dw = torch.empty(K, k, k, C, device="cuda")
    ws = torch.empty(hip_conv.wgrad_workspace_numel(K, k * k * C), device="cuda")
    hip_conv.conv_wgrad(dy, x, dw, (k, k), (s, s), (pd, pd), workspace=ws)
    refdw = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (K, C, k, k), dy.permute(0, 3, 1, 2).double(),
                                        stride=s, padding=pd).permute(0, 2, 3, 1)
    e3 = dw.double() - refdw
    print(f"{H}x{W} {C}->{K} k{k} s{s}: fprop max|err| {err.abs().max().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e} "
          f"(ref rms {ref.pow(2).mean().sqrt().item():.3f}) | dgrad max {e2.abs().max().item():.3e} rms {e2.pow(2).mean().sqrt().item():.3e}")
    print(f"   wgrad max|err| {e3.abs().max().item():.3e} rms {e3.pow(2).mean().sqrt().item():.3e} (ref rms {refdw.pow(2).mean().sqrt().item():.3f})")

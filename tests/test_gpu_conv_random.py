"""Randomised geometry sweep of conv fprop / dgrad / wgrad against torch (T2): odd sizes, strides, paddings, channel counts
around the tile boundaries, row counts around the tail-split / split-K thresholds."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cases(seed, n):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([1, 1, 3, 3, 5, 7])
        s = rng.choice([1, 1, 2])
        p = rng.choice([0, k // 2])
        h = rng.randint(max(k, 5), 40)
        w = rng.randint(max(k, 5), 40)
        c = 8 * rng.randint(1, 40)
        kk = 8 * rng.randint(1, 40)
        nb = rng.choice([1, 2, 3, 8, 17, 64, 130])
        if nb * h * w * max(c, kk) > 6e7 or (h + 2 * p - k) // s + 1 < 1:
            continue
        out.append((nb, h, w, c, kk, k, s, p))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_random_geometries(dtype, fp32_mode):
    from distributedpytorch_amd.ops import hip_conv
    g = torch.Generator(device="cuda").manual_seed(99)
    for (N, H, W, C, K, k, s, pd) in _cases(1234, 40):
        x = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
        w = (torch.randn(K, k, k, C, device="cuda", generator=g) * (k * k * C) ** -0.5).to(dtype)
        Ho, Wo = hip_conv.out_size(H, k, s, pd), hip_conv.out_size(W, k, s, pd)
        y = torch.full((N, Ho, Wo, K), float("nan"), device="cuda", dtype=dtype)
        stats = torch.zeros(hip_conv.stats_rows(N * Ho * Wo, K, dtype), 2, K, device="cuda")
        hip_conv.conv_fprop(x, w, y, (s, s), (pd, pd), stats=stats)
        xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
        yr = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), stride=s, padding=pd)
        tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
        msg = f"shape {(N, H, W, C, K, k, s, pd)}"
        torch.testing.assert_close(y.float().permute(0, 3, 1, 2), yr.detach(), msg=lambda m: f"fprop {msg}: {m}", **tol)
        yf = y.float().reshape(-1, K)
        torch.testing.assert_close(stats.sum(0)[0], yf.sum(0), rtol=2e-3, atol=5e-2, msg=lambda m: f"stats {msg}: {m}")
        dy = torch.randn(N, Ho, Wo, K, device="cuda", generator=g).to(dtype)
        yr.backward(dy.float().permute(0, 3, 1, 2))
        dx = torch.full((N, H, W, C), float("nan"), device="cuda", dtype=dtype)
        hip_conv.conv_dgrad(dy, w.permute(3, 1, 2, 0).contiguous(), dx, (s, s), (pd, pd))
        torch.testing.assert_close(dx.float(), xr.grad, msg=lambda m: f"dgrad {msg}: {m}", **tol)
        dw = torch.full((K, k, k, C), float("nan"), device="cuda")
        ws = torch.empty(hip_conv.wgrad_workspace_numel(K, k * k * C), device="cuda")
        hip_conv.conv_wgrad(dy, x, dw, (k, k), (s, s), (pd, pd), workspace=ws)
        scale = wr.grad.abs().max().item()
        torch.testing.assert_close(dw, wr.grad, rtol=tol["rtol"], atol=max(tol["atol"], scale * (3e-2 if dtype == torch.bfloat16 else 1e-4)),
                                   msg=lambda m: f"wgrad {msg}: {m}")

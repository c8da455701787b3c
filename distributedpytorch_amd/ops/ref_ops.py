"""Plain-PyTorch implementation of every op with the SAME signatures as hip_ops.

Two roles: (1) the CPU execution path (config 1 of BASELINE.json: gloo, world_size=2, no GPU) and (2) the
numerics oracle the HIP kernels are tested against.  Everything is NHWC; tensors may be channel slices.
Nothing here runs when tensors live on the GPU - the dispatcher in ops/__init__.py sends those to hip_ops.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _nchw(t):
    return t.permute(0, 3, 1, 2)


def _nhwc(t):
    return t.permute(0, 2, 3, 1)


def out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


# ----------------------------------------------------------------------------------------- convolution
def stats_rows(m, cout):
    return 1


def conv_fprop(x, w, y, stride, pad, bias=None, relu=False, stats=None, addend=None, stats_shift=None, w3=None, oscale=None):
    out = F.conv2d(_nchw(x).float(), _nchw(w).float(), None, stride=stride, padding=pad)
    out = _nhwc(out)
    if oscale is not None:
        out = out * oscale.float()[: w.shape[0]]
    if bias is not None:
        out = out + bias.float()[: w.shape[0]]
    stored = out.to(y.dtype)
    if stats is not None:
        sf = stored.float().reshape(-1, stored.shape[-1])
        if stats_shift is not None:
            sf = sf - stats_shift
        stats.zero_()
        stats[0, 0] = sf.sum(0)
        stats[0, 1] = (sf * sf).sum(0)
    if addend is not None:
        out = out + addend.float()
    if relu:
        out = out.clamp_min(0)
    y.copy_(out.to(y.dtype))
    return y


class SecondBnUnsupported(RuntimeError):
    """conv_dgrad(bn_reduce=(..., bits, second)) on a launch that does not go to the pre-split fp32 kernel: nothing was launched."""


class CompactAddendUnsupported(RuntimeError):
    """conv_dgrad(addend2=...) on a launch that does not go to the pre-split fp32 kernel: nothing was launched."""


def conv_dgrad(dy, wt, dx, stride, pad, addend=None, wt3=None, bn_reduce=None, addend2=None):
    # wt: [Cin, kh, kw, Cout]  -> torch weight [Cout, Cin, kh, kw]
    w = wt.permute(3, 0, 1, 2).float()
    n, h, wd, c = dx.shape
    g = torch.nn.grad.conv2d_input((n, c, h, wd), w, _nchw(dy).float(), stride=stride, padding=pad)
    g = _nhwc(g)
    if addend is not None:
        g = g + addend.float()
    if addend2 is not None:         # compact gradient of a 1x1 / stride-2 convolution over the same input
        g = g.clone()
        g[:, ::2, ::2, :] += addend2.float()
    if bn_reduce is not None:       # fused BatchNorm-backward reduction (csrc/hip/conv_common.h: BnReduceLane)
        x, y, mean, invstd, mscale, mshift, slab = bn_reduce[:7]      # (an eighth element: y's mask as bits, same mask)
        second = bn_reduce[8] if len(bn_reduce) > 8 else None         # (a ninth: (x2, mean2, invstd2, slab2), see hip_conv)
        if y is not None:
            g = g * (y > 0)
        elif mscale is not None:
            g = g * ((x.float() * mscale + mshift) > 0)
        gs = g.to(dx.dtype).float().reshape(-1, c)
        slab.zero_()
        slab[0, 0] = gs.sum(0)
        slab[0, 1] = (gs * ((x.float().reshape(-1, c) - mean) * invstd)).sum(0)
        if second is not None:
            x2, mean2, invstd2, slab2 = second
            slab2.zero_()
            slab2[0, 0] = gs.sum(0)
            slab2[0, 1] = (gs * ((x2.float().reshape(-1, c) - mean2) * invstd2)).sum(0)
    dx.copy_(g.to(dx.dtype))
    return 1


def wgrad_workspace_numel(cout, rsc):
    return 1


def conv_wgrad(dy, x, dw, ksize, stride, pad, workspace=None, accumulate=False):
    k, c = dy.shape[3], x.shape[3]
    g = torch.nn.grad.conv2d_weight(_nchw(x).float(), (k, c, ksize[0], ksize[1]), _nchw(dy).float(), stride=stride,
                                    padding=pad)
    g = g.permute(0, 2, 3, 1).reshape(dw.shape)
    if accumulate:
        dw.add_(g)
    else:
        dw.copy_(g)
    return dw


def transpose_krsc(w, wt):
    wt.copy_(w.permute(3, 1, 2, 0))
    return wt


# ----------------------------------------------------------------------------------------- batch norm
def bn_slab_rows(m):
    return 1


def bn_stats(x, slab, shift=None):
    xf = x.float().reshape(-1, x.shape[-1])
    if shift is not None:
        xf = xf - shift
    slab.zero_()
    slab[0, 0] = xf.sum(0)
    slab[0, 1] = (xf * xf).sum(0)
    return slab


def bn_finalize(slab, count, gamma, beta, running_mean, running_var, nbt, momentum, eps, mean, invstd, scale, shift,
                scratch=None, stat_shift=None):
    """The slab holds sum(x - c), sum((x - c)^2) about the per-channel shift c = stat_shift (0 if None); stat_shift is
    refreshed to the new mean (shifted-data variance, csrc/hip/bn.hip: bn_finalize_kernel)."""
    s = slab.sum(0)
    dm = s[0] / count
    m = dm if stat_shift is None else stat_shift + dm
    var = (s[1] / count - dm * dm).clamp_min(0)
    if stat_shift is not None:
        stat_shift.copy_(m)
    inv = torch.rsqrt(var + eps)
    mean.copy_(m)
    invstd.copy_(inv)
    g = gamma if gamma is not None else torch.ones_like(m)
    b = beta if beta is not None else torch.zeros_like(m)
    scale.copy_(g * inv)
    shift.copy_(b - m * g * inv)
    if running_mean is not None:
        unbiased = var * count / (count - 1) if count > 1 else var
        running_mean.mul_(1 - momentum).add_(momentum * m)
        running_var.mul_(1 - momentum).add_(momentum * unbiased)
    if nbt is not None:
        nbt.add_(1)


def bn_eval_coeffs(gamma, beta, rm, rv, eps, scale, shift):
    inv = torch.rsqrt(rv + eps)
    g = gamma if gamma is not None else torch.ones_like(rm)
    b = beta if beta is not None else torch.zeros_like(rm)
    scale.copy_(g * inv)
    shift.copy_(b - rm * g * inv)


def bn_apply(x, scale, shift, y, residual=None, relu=False, mask_bits=None, res_scale=None, res_shift=None):
    out = x.float() * scale + shift
    if residual is not None:
        out = out + (residual.float() if res_scale is None else residual.float() * res_scale + res_shift)
    if relu:
        out = out.clamp_min(0)
    y.copy_(out.to(y.dtype))
    if mask_bits is not None:       # bit b of word w = (y[row, 32 w + b] > 0)
        c = y.shape[-1]
        pos = (y.reshape(-1, c // 32, 32) > 0).to(torch.int64)
        words = (pos << torch.arange(32, dtype=torch.int64, device=y.device)).sum(-1)
        mask_bits.copy_(torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).reshape(mask_bits.shape))
    return y


def bn_bwd_reduce(dy, y, x, mean, invstd, gamma, dgamma, dbeta, sums, coef, slab=None, scratch=None, accumulate=False,
                  mask_scale=None, mask_shift=None):
    """Masks dy in place by (y > 0) when y is given; sums[0]=sum dz, sums[1]=sum dz*xhat; dgamma/dbeta; coef = the
    folded per-channel (A, B, D) of dx = A*dz + B*x + D consumed by bn_bwd_dx.  With mask_scale/mask_shift the ReLU mask
    is recomputed as scale*x+shift > 0 and dy is left untouched (bn_bwd_dx gets the same coefficients)."""
    if mask_scale is not None:
        dz = dy.float() * ((x.float() * mask_scale + mask_shift) > 0)
        dz = dz.reshape(-1, dy.shape[-1])
    else:
        if y is not None:
            dy.mul_((y > 0).to(dy.dtype))
        dz = dy.float().reshape(-1, dy.shape[-1])
    m = dz.shape[0]
    xh = (x.float().reshape(-1, x.shape[-1]) - mean) * invstd
    s = dz.sum(0)
    q = (dz * xh).sum(0)
    sums[0].copy_(s)
    sums[1].copy_(q)
    if dgamma is not None:
        if accumulate:
            dgamma.add_(q)
        else:
            dgamma.copy_(q)
    if dbeta is not None:
        if accumulate:
            dbeta.add_(s)
        else:
            dbeta.copy_(s)
    g = gamma if gamma is not None else torch.ones_like(mean)
    a = g * invstd
    b = -g * invstd * invstd * q / m
    coef[0].copy_(a)
    coef[1].copy_(b)
    coef[2].copy_(-a * s / m - b * mean)


def bn_bwd_from_slab(slab, m, mean, invstd, gamma, dgamma, dbeta, sums, coef, scratch=None, accumulate=False):
    t = slab.sum(0)
    s, q = t[0], t[1]
    sums[0].copy_(s)
    sums[1].copy_(q)
    if dgamma is not None:
        dgamma.copy_(dgamma + q if accumulate else q)
    if dbeta is not None:
        dbeta.copy_(dbeta + s if accumulate else s)
    g = gamma if gamma is not None else torch.ones_like(invstd)
    a = g * invstd
    b = -g * invstd * invstd * q / m
    coef[0].copy_(a)
    coef[1].copy_(b)
    coef[2].copy_(-a * s / m - b * mean)


def bn_bwd_dx(dz, x, coef, dx, mask_scale=None, mask_shift=None, accumulate=False):
    g = dz.float()
    if mask_scale is not None:
        g = g * ((x.float() * mask_scale + mask_shift) > 0)
    out = coef[0] * g + coef[1] * x.float() + coef[2]
    if accumulate:
        out = out + dx.float()
    dx.copy_(out.to(dx.dtype))
    return dx


def bn_bwd_dx2(dz, x1, coef1, dx1, x2, coef2, dx2):
    g = dz.float()
    o1 = coef1[0] * g + coef1[1] * x1.float() + coef1[2]
    o2 = coef2[0] * g + coef2[1] * x2.float() + coef2[2]
    dx1.copy_(o1.to(dx1.dtype))
    dx2.copy_(o2.to(dx2.dtype))
    return dx1, dx2


# ----------------------------------------------------------------------------------------- elementwise
def relu_fwd(x, y):
    y.copy_(x.clamp_min(0))
    return y


def relu_bwd(dy, y):
    dy.mul_((y > 0).to(dy.dtype))
    return dy


def add(a, b, out):
    out.copy_((a.float() + b.float()).to(out.dtype))
    return out


def copy(a, out):
    out.copy_(a)
    return out


def _philox_mask(shape, p, rng, layer_id, device):
    # CPU masks come from a torch generator keyed like the device Philox stream (seed, step, layer): deterministic
    # and identical between forward and backward, not bit-identical to the device stream.
    g = torch.Generator(device="cpu")
    g.manual_seed((int(rng[0]) * 1000003 + int(rng[1]) * 8191 + int(layer_id)) % (2 ** 63 - 1))
    return (torch.rand(shape, generator=g) >= p).to(device)


def dropout(x, y, p, rng, layer_id):
    keep = _philox_mask(tuple(x.shape), p, rng, layer_id, x.device)
    y.copy_((x.float() * keep / (1.0 - p)).to(y.dtype))
    return y


def rng_advance(rng):
    rng[1] += 1


def colsum(x, out, accumulate=False):
    s = x.float().reshape(-1, x.shape[-1]).sum(0)
    if accumulate:
        out.add_(s)
    else:
        out.copy_(s)
    return out


def relu_bwd_colsum(dy, y, out, accumulate=False):
    relu_bwd(dy, y)
    return colsum(dy, out, accumulate)


# ----------------------------------------------------------------------------------------- pooling
def pool_out_size(h, k, s, p, ceil_mode=False):
    if ceil_mode:
        o = -(-(h + 2 * p - k) // s) + 1
        if (o - 1) * s >= h + p:
            o -= 1
        return o
    return (h + 2 * p - k) // s + 1


def maxpool_fwd(x, y, idx, ksize, stride, pad, scale=None, shift=None):
    if scale is not None:
        x = (x.float() * scale + shift).clamp_min(0).to(x.dtype)
    n, h, w, c = x.shape
    ho, wo = y.shape[1], y.shape[2]
    # emulate ceil_mode by padding -inf on the bottom/right as needed
    need_h = (ho - 1) * stride[0] + ksize[0] - (h + 2 * pad[0])
    need_w = (wo - 1) * stride[1] + ksize[1] - (w + 2 * pad[1])
    xp = F.pad(_nchw(x).float(), (pad[1], pad[1] + max(need_w, 0), pad[0], pad[0] + max(need_h, 0)), value=float("-inf"))
    cols = F.unfold(xp, ksize, stride=stride).reshape(n, c, ksize[0] * ksize[1], ho, wo)
    val, arg = cols.max(dim=2)
    y.copy_(_nhwc(val).to(y.dtype))
    if idx is not None:
        idx.copy_(_nhwc(arg).to(torch.uint8))
    return y


def maxpool_bwd(dy, idx, dx, ksize, stride, pad):
    n, h, w, c = dx.shape
    ho, wo = dy.shape[1], dy.shape[2]
    need_h = max((ho - 1) * stride[0] + ksize[0] - (h + 2 * pad[0]), 0)
    need_w = max((wo - 1) * stride[1] + ksize[1] - (w + 2 * pad[1]), 0)
    kk = ksize[0] * ksize[1]
    onehot = F.one_hot(_nchw(idx).long(), kk).permute(0, 1, 4, 2, 3).float()      # n c kk ho wo
    cols = (onehot * _nchw(dy).float().unsqueeze(2)).reshape(n, c * kk, ho * wo)
    g = F.fold(cols, (h + 2 * pad[0] + need_h, w + 2 * pad[1] + need_w), ksize, stride=stride)
    g = g[:, :, pad[0]:pad[0] + h, pad[1]:pad[1] + w]
    dx.copy_(_nhwc(g).to(dx.dtype))
    return dx


def maxpool_bwd_bn(dy, idx, dx, ksize, stride, pad, z, scale, shift, mean, invstd, slab):
    maxpool_bwd(dy, idx, dx, ksize, stride, pad)
    c = dx.shape[3]
    d = (dx.float() * ((z.float() * scale + shift) > 0)).reshape(-1, c)
    slab.zero_()
    slab[0, 0] = d.sum(0)
    slab[0, 1] = (d * ((z.float().reshape(-1, c) - mean) * invstd)).sum(0)
    return 1


def avgpool_fwd(x, y, ksize, stride, pad, adaptive=False):
    if adaptive:
        out = F.adaptive_avg_pool2d(_nchw(x).float(), (y.shape[1], y.shape[2]))
    else:
        out = F.avg_pool2d(_nchw(x).float(), ksize, stride=stride, padding=pad)
    y.copy_(_nhwc(out).to(y.dtype))
    return y


def avgpool_bwd(dy, dx, ksize, stride, pad, adaptive=False):
    xin = torch.zeros(dx.shape[0], dx.shape[3], dx.shape[1], dx.shape[2], requires_grad=True)
    if adaptive:
        out = F.adaptive_avg_pool2d(xin, (dy.shape[1], dy.shape[2]))
    else:
        out = F.avg_pool2d(xin, ksize, stride=stride, padding=pad)
    (g,) = torch.autograd.grad(out, xin, _nchw(dy).float())
    dx.copy_(_nhwc(g).to(dx.dtype))
    return dx


# ----------------------------------------------------------------------------------------- loss
def ce_loss(logits, labels, num_classes, out, accum=None, dlogits=None, class_w=None, mode=0, gamma=2.0, scale=1.0,
            aux=False):
    z = logits[:, :num_classes].float()
    n = z.shape[0]
    logp = F.log_softmax(z, dim=-1)
    p = logp.exp()
    w = class_w[labels] if class_w is not None else torch.ones(n)
    lp_y = logp.gather(1, labels.view(-1, 1)).squeeze(1)
    p_y = lp_y.exp()
    if mode == 2:
        per = -w * (1 - p_y).clamp_min(0) ** gamma * lp_y
        denom = float(n)
        f = w * ((1 - p_y).clamp_min(0) ** gamma - gamma * (1 - p_y).clamp_min(0) ** (gamma - 1) * p_y * lp_y)
    else:
        per = -w * lp_y
        denom = float(w.sum()) if mode == 1 else float(n)
        f = w
    loss = per.sum() / denom
    acc = (z.argmax(1) == labels).float().mean()
    if aux:
        out[0] += scale * loss
        if accum is not None:
            accum[0] += scale * loss
    else:
        out[0] = scale * loss
        out[1] = acc
        if accum is not None:
            accum[0] += scale * loss
            accum[1] += acc
            accum[2] += 1
    if dlogits is not None:
        onehot = F.one_hot(labels, num_classes).float()
        g = (f / denom * scale).unsqueeze(1) * (p - onehot)
        dlogits.zero_()
        dlogits[:, :num_classes] = g.to(dlogits.dtype)
    return out


# ----------------------------------------------------------------------------------------- optimizer
def adam_step(p, g, m, v, p16, hyper, lo=0, hi=None):
    hi = p.numel() if hi is None else hi
    h = [float(x) for x in hyper]
    lr, b1, b2, eps, wd, step, _, gs, omb1, omb2 = h[:10]
    b1d, b2d = b1 + h[10], b2 + h[11]          # the double betas (nn/optim.py: make_hyper)
    step += 1
    gr = g[lo:hi] * gs
    if wd:
        gr = gr + wd * p[lo:hi]
    m[lo:hi].lerp_(gr, omb1)
    v[lo:hi].mul_(b2).addcmul_(gr, gr, value=omb2)
    bc1 = 1 - b1d ** step
    bc2s = math.sqrt(1 - b2d ** step)
    denom = v[lo:hi].sqrt() / bc2s + eps
    p[lo:hi].addcdiv_(m[lo:hi], denom, value=-lr / bc1)
    if p16 is not None:
        p16[lo:hi].copy_(p[lo:hi])


def sgd_step(p, g, buf, p16, hyper, lo=0, hi=None):
    hi = p.numel() if hi is None else hi
    lr, _, _, _, wd, step, mu, gs = [float(h) for h in hyper[:8]]
    gr = g[lo:hi] * gs
    if wd:
        gr = gr + wd * p[lo:hi]
    if mu != 0:
        if step < 0.5:
            buf[lo:hi].copy_(gr)
        else:
            buf[lo:hi].mul_(mu).add_(gr)
        gr = buf[lo:hi]
    p[lo:hi].add_(gr, alpha=-lr)
    if p16 is not None:
        p16[lo:hi].copy_(p[lo:hi])


def advance_step(hyper):
    hyper[5] += 1


def cast(src, dst):
    dst.copy_(src)
    return dst


# ----------------------------------------------------------------------------------------- input pipeline
def augment_mnist(images, labels, index, out, labels_out, train, mean, std, seed, epoch):
    """CPU stand-in for csrc/hip/augment.hip: same transform family (rotate +-5 deg nearest, random-resized-crop,
    bilinear resize, normalise, 3-channel repeat), parameters from a torch generator keyed by (seed, epoch, index)."""
    b = index.numel()
    s = images.shape[1]
    d = out.shape[1]
    idx = index.long()
    src = images[idx].float()                      # b s s
    labels_out.copy_(labels[idx])
    if not train:        # eval transform has no per-sample parameters: one batched bilinear resize
        res = F.interpolate(src[:, None], size=(d, d), mode="bilinear", align_corners=False)[:, 0]
        out.zero_()
        out[..., :3] = ((res / 255.0 - mean) / std).unsqueeze(-1).to(out.dtype)
        return out
    res = torch.empty(b, d, d)
    for i in range(b):
        img = src[i]
        top, left, h, w = 0, 0, s, s
        if train:
            g = torch.Generator().manual_seed((int(seed) * 2654435761 + int(epoch) * 97 + int(idx[i])) % (2 ** 63 - 1))
            r = torch.rand(64, generator=g)
            ang = float(r[0] * 10 - 5) * math.pi / 180
            ca, sa = math.cos(ang), math.sin(ang)
            ys, xs = torch.meshgrid(torch.arange(s) + 0.5 - s / 2, torch.arange(s) + 0.5 - s / 2, indexing="ij")
            sx = torch.floor(ca * xs - sa * ys + s / 2).long()
            sy = torch.floor(sa * xs + ca * ys + s / 2).long()
            ok = (sx >= 0) & (sx < s) & (sy >= 0) & (sy < s)
            img = torch.where(ok, img[sy.clamp(0, s - 1), sx.clamp(0, s - 1)], torch.zeros(()))
            for a in range(10):
                ta = s * s * (0.08 + 0.92 * float(r[1 + 4 * a]))
                ar = math.exp(math.log(3 / 4) + (math.log(4 / 3) - math.log(3 / 4)) * float(r[2 + 4 * a]))
                cw, ch = int(round(math.sqrt(ta * ar))), int(round(math.sqrt(ta / ar)))
                if 0 < cw <= s and 0 < ch <= s:
                    top = int(float(r[3 + 4 * a]) * (s - ch + 1))
                    left = int(float(r[4 + 4 * a]) * (s - cw + 1))
                    h, w = ch, cw
                    break
        crop = img[top:top + h, left:left + w]
        res[i] = F.interpolate(crop[None, None], size=(d, d), mode="bilinear", align_corners=False)[0, 0]
    res = (res / 255.0 - mean) / std
    out.zero_()
    out[..., :3] = res.unsqueeze(-1).to(out.dtype)
    return out

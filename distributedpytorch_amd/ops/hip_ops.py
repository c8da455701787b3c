"""HIP (gfx950) implementation of every op: thin ctypes launches of the kernels in csrc/hip/*.hip on the current
torch HIP stream.  Same signatures as ref_ops.  No eager fallback: a missing library raises (native.hip_lib)."""
from __future__ import annotations

import ctypes

import torch

from . import native
from .hip_conv import (conv_dgrad, conv_fprop, conv_wgrad, out_size, stats_rows,  # noqa: F401
                       wgrad_workspace_numel, _check_nhwc, split3_planes, last_stats_rows, bn_reduce_rows, weight_prep, CompactAddendUnsupported, SecondBnUnsupported)
from .native import F32, I32, I64, VP, ptr, stream_ptr

_DT = {torch.float32: 0, torch.bfloat16: 1}
ROWS_L2 = 256


class PoolP(ctypes.Structure):
    _fields_ = [("N", I32), ("H", I32), ("W", I32), ("C", I32), ("Ho", I32), ("Wo", I32),
                ("kh", I32), ("kw", I32), ("sh", I32), ("sw", I32), ("ph", I32), ("pw", I32),
                ("x_cs", I32), ("y_cs", I32)]


class AugP(ctypes.Structure):
    _fields_ = [("B", I32), ("S", I32), ("D", I32), ("Cpad", I32), ("train", I32), ("mean", F32), ("std", F32),
                ("seed", ctypes.c_ulonglong), ("epoch", ctypes.c_ulonglong)]


def _lib():
    return native.hip_lib()


def _mc(t):
    """(M rows, C channels, channel stride) of an NHWC (or [M, C]) tensor."""
    if t.dim() == 2:
        assert t.stride(1) == 1
        return t.shape[0], t.shape[1], t.stride(0)
    cs = _check_nhwc(t, "tensor")
    return t.shape[0] * t.shape[1] * t.shape[2], t.shape[3], cs


_scratch = {}


def _scratch_f32(dev, n, tag):
    key = (dev, tag)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 16), device=dev, dtype=torch.float32)
        _scratch[key] = buf
    return buf


def transpose_krsc(w, wt):
    k, r, s, c = w.shape
    _lib().dpa_transpose_krsc(ptr(w), ptr(wt), k, r * s, c, _DT[w.dtype], stream_ptr())
    return wt


# ----------------------------------------------------------------------------------------- batch norm
def bn_slab_rows(m):
    return int(_lib().dpa_bn_slab_rows(I64(m)))


def bn_stats(x, slab, shift=None):
    m, c, cs = _mc(x)
    assert slab.numel() >= bn_slab_rows(m) * 2 * c
    _lib().dpa_bn_stats(ptr(x), _DT[x.dtype], I64(m), c, cs, ptr(slab), ptr(shift), stream_ptr())
    return slab


def bn_finalize(slab, count, gamma, beta, running_mean, running_var, nbt, momentum, eps, mean, invstd, scale, shift,
                scratch=None, stat_shift=None):
    rows, c = slab.shape[0], slab.shape[2]
    if scratch is None:
        scratch = _scratch_f32(slab.device, ROWS_L2 * 2 * c, "bn")
    _lib().dpa_bn_finalize(ptr(slab), rows, c, I64(count), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                           ptr(nbt), F32(momentum), F32(eps), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                           ptr(scratch), ptr(stat_shift), stream_ptr())


def bn_eval_coeffs(gamma, beta, rm, rv, eps, scale, shift):
    _lib().dpa_bn_eval_coeffs(rm.numel(), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), F32(eps), ptr(scale), ptr(shift),
                              stream_ptr())


def bn_apply(x, scale, shift, y, residual=None, relu=False, mask_bits=None, res_scale=None, res_shift=None):
    """res_scale / res_shift: the residual is the input of a second BatchNorm, residual * res_scale + res_shift is added.
    mask_bits: int32 [rows, C / 32] or None - written with bit b of word w = (y[row, 32 w + b] > 0) (C % 32 == 0): the
    ReLU mask for a fused dgrad epilogue (conv_dgrad(bn_reduce=(..., bits))) at 1 / 32 of y's bytes."""
    m, c, xcs = _mc(x)
    _, _, ycs = _mc(y)
    rcs = _mc(residual)[2] if residual is not None else 0
    if mask_bits is not None:
        assert c % 32 == 0 and mask_bits.dtype == torch.int32 and mask_bits.is_contiguous() and mask_bits.numel() == m * (c // 32)
    _lib().dpa_bn_apply(ptr(x), xcs, ptr(scale), ptr(shift), ptr(residual), rcs, ptr(y), ycs, I64(m), c, int(relu),
                        _DT[x.dtype], stream_ptr(), ptr(mask_bits), ptr(res_scale), ptr(res_shift))
    return y


def bn_bwd_reduce(dy, y, x, mean, invstd, gamma, dgamma, dbeta, sums, coef, slab=None, scratch=None, accumulate=False,
                  mask_scale=None, mask_shift=None):
    m, c, dcs = _mc(dy)
    xcs = _mc(x)[2]
    ycs = _mc(y)[2] if y is not None else 0
    rows = bn_slab_rows(m)
    if slab is None:
        slab = _scratch_f32(dy.device, rows * 2 * c, "bnbwd_slab")
    if scratch is None:
        scratch = _scratch_f32(dy.device, ROWS_L2 * 2 * c, "bn")
    _lib().dpa_bn_bwd_reduce(ptr(dy), dcs, ptr(y), ycs, ptr(x), xcs, ptr(mean), ptr(invstd), ptr(gamma), I64(m), c,
                             ptr(slab), ptr(scratch), ptr(dgamma), ptr(dbeta), ptr(sums), ptr(coef), int(accumulate),
                             _DT[dy.dtype], stream_ptr(), ptr(mask_scale), ptr(mask_shift))


def bn_bwd_from_slab(slab, m, mean, invstd, gamma, dgamma, dbeta, sums, coef, scratch=None, accumulate=False):
    """Finish a BatchNorm-backward reduction whose per-tile partial sums a dgrad epilogue wrote (conv_dgrad(bn_reduce=))."""
    rows, c = slab.shape[0], slab.shape[2]
    if scratch is None:
        scratch = _scratch_f32(slab.device, ROWS_L2 * 2 * c, "bn")
    _lib().dpa_bn_bwd_from_slab(ptr(slab), rows, ptr(mean), ptr(invstd), ptr(gamma), I64(m), c, ptr(scratch), ptr(dgamma),
                                ptr(dbeta), ptr(sums), ptr(coef), int(accumulate), stream_ptr())


def bn_bwd_dx(dz, x, coef, dx, mask_scale=None, mask_shift=None, accumulate=False):
    m, c, zcs = _mc(dz)
    _lib().dpa_bn_bwd_dx(ptr(dz), zcs, ptr(x), _mc(x)[2], ptr(coef), ptr(dx), _mc(dx)[2], I64(m), c, _DT[dz.dtype],
                         stream_ptr(), ptr(mask_scale), ptr(mask_shift), int(bool(accumulate)))
    return dx


def bn_bwd_dx2(dz, x1, coef1, dx1, x2, coef2, dx2):
    """Two bn_bwd_dx (no mask) over one gradient dz in one pass: dx_i = A_i dz + B_i x_i + D_i."""
    m, c, zcs = _mc(dz)
    for t in (x1, dx1, x2, dx2):
        assert _mc(t)[:2] == (m, c) and t.dtype == dz.dtype
    _lib().dpa_bn_bwd_dx2(ptr(dz), zcs, ptr(x1), _mc(x1)[2], ptr(coef1), ptr(dx1), _mc(dx1)[2], ptr(x2), _mc(x2)[2], ptr(coef2),
                          ptr(dx2), _mc(dx2)[2], I64(m), c, _DT[dz.dtype], stream_ptr())
    return dx1, dx2


# ----------------------------------------------------------------------------------------- elementwise
def relu_fwd(x, y):
    m, c, xcs = _mc(x)
    _lib().dpa_relu_fwd(ptr(x), xcs, ptr(y), _mc(y)[2], I64(m), c, _DT[x.dtype], stream_ptr())
    return y


def relu_bwd(dy, y):
    m, c, dcs = _mc(dy)
    _lib().dpa_relu_bwd(ptr(dy), dcs, ptr(y), _mc(y)[2], I64(m), c, _DT[dy.dtype], stream_ptr())
    return dy


def add(a, b, out):
    m, c, acs = _mc(a)
    _lib().dpa_add(ptr(a), acs, ptr(b), _mc(b)[2], ptr(out), _mc(out)[2], I64(m), c, _DT[a.dtype], stream_ptr())
    return out


def copy(a, out):
    m, c, acs = _mc(a)
    _lib().dpa_copy(ptr(a), acs, ptr(out), _mc(out)[2], I64(m), c, _DT[a.dtype], stream_ptr())
    return out


def dropout(x, y, p, rng, layer_id):
    m, c, xcs = _mc(x)
    _lib().dpa_dropout(ptr(x), xcs, ptr(y), _mc(y)[2], I64(m), c, F32(p), ptr(rng), ctypes.c_ulonglong(layer_id),
                       _DT[x.dtype], stream_ptr())
    return y


def rng_advance(rng):
    _lib().dpa_rng_advance(ptr(rng), stream_ptr())


def colsum(x, out, accumulate=False):
    m, c, cs = _mc(x)
    if m > 2048:
        rows = bn_slab_rows(m)
        slab = _scratch_f32(x.device, rows * 2 * c, "colsum_slab")
        scratch = _scratch_f32(x.device, ROWS_L2 * 2 * c, "bn")
        _lib().dpa_colsum_tall(ptr(x), _DT[x.dtype], I64(m), c, cs, ptr(slab), ptr(scratch), ptr(out), int(accumulate),
                               stream_ptr())
    else:
        _lib().dpa_colsum(ptr(x), cs, I64(m), c, ptr(out), int(accumulate), _DT[x.dtype], stream_ptr())
    return out


def relu_bwd_colsum(dy, y, out, accumulate=False):
    """dy *= (y > 0) in place and out (+)= the column sums of the masked dy: ReLU backward + bias gradient in one pass."""
    m, c, cs = _mc(dy)
    if m <= 2048:
        relu_bwd(dy, y)
        return colsum(dy, out, accumulate)
    rows = bn_slab_rows(m)
    slab = _scratch_f32(dy.device, rows * 2 * c, "colsum_slab")
    scratch = _scratch_f32(dy.device, ROWS_L2 * 2 * c, "bn")
    _lib().dpa_relu_bwd_colsum(ptr(dy), cs, ptr(y), _mc(y)[2], _DT[dy.dtype], I64(m), c, ptr(slab), ptr(scratch), ptr(out),
                               int(accumulate), stream_ptr())
    return out


# ----------------------------------------------------------------------------------------- pooling
from .ref_ops import pool_out_size  # noqa: E402,F401  (pure arithmetic)


def _poolp(x, y, ksize, stride, pad):
    p = PoolP()
    p.N, p.H, p.W, p.C = x.shape
    p.Ho, p.Wo = y.shape[1], y.shape[2]
    p.kh, p.kw = ksize
    p.sh, p.sw = stride
    p.ph, p.pw = pad
    p.x_cs = _check_nhwc(x, "x")
    p.y_cs = _check_nhwc(y, "y")
    return p


def maxpool_fwd(x, y, idx, ksize, stride, pad, scale=None, shift=None):
    """scale / shift (per channel): pool relu(x * scale + shift) instead of x (BatchNorm-apply + ReLU folded in)."""
    p = _poolp(x, y, ksize, stride, pad)
    assert ksize[0] * ksize[1] <= 255 and (scale is None) == (shift is None)
    _lib().dpa_maxpool_fwd(ptr(x), ptr(y), ptr(idx), ctypes.byref(p), _DT[x.dtype], stream_ptr(), ptr(scale), ptr(shift))
    return y


def maxpool_bwd(dy, idx, dx, ksize, stride, pad):
    p = _poolp(dx, dy, ksize, stride, pad)
    _lib().dpa_maxpool_bwd(ptr(dy), ptr(idx), ptr(dx), ctypes.byref(p), p.y_cs, p.x_cs, _DT[dy.dtype], stream_ptr())
    return dx


def maxpool_bwd_bn(dy, idx, dx, ksize, stride, pad, z, scale, shift, mean, invstd, slab):
    """maxpool_bwd + the partial sums of the backward reduction of the BatchNorm(+ReLU) in front of the pool (input z, mask
    scale * z + shift > 0) into slab [>= 2048, 2, C].  Returns the slab rows written, or 0 when the geometry is not taken
    (nothing launched)."""
    p = _poolp(dx, dy, ksize, stride, pad)
    assert slab.dtype == torch.float32 and slab.is_contiguous() and slab.shape[0] >= 2048 and tuple(slab.shape[1:]) == (2, dx.shape[3])
    assert z.shape == dx.shape and z.dtype == dx.dtype
    rows = _lib().dpa_maxpool_bwd_bn(ptr(dy), ptr(idx), ptr(dx), ctypes.byref(p), p.y_cs, p.x_cs, _DT[dy.dtype], stream_ptr(),
                                     ptr(z), _mc(z)[2], ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(slab))
    if rows < 0:
        raise RuntimeError(f"dpa_maxpool_bwd_bn: launch error {-rows}")
    return int(rows)


def avgpool_fwd(x, y, ksize, stride, pad, adaptive=False):
    p = _poolp(x, y, ksize or (1, 1), stride or (1, 1), pad or (0, 0))
    _lib().dpa_avgpool_fwd(ptr(x), ptr(y), ctypes.byref(p), int(adaptive), _DT[x.dtype], stream_ptr())
    return y


def avgpool_bwd(dy, dx, ksize, stride, pad, adaptive=False):
    p = _poolp(dx, dy, ksize or (1, 1), stride or (1, 1), pad or (0, 0))
    _lib().dpa_avgpool_bwd(ptr(dy), ptr(dx), ctypes.byref(p), p.y_cs, p.x_cs, int(adaptive), _DT[dy.dtype], stream_ptr())
    return dx


# ----------------------------------------------------------------------------------------- loss
def ce_loss(logits, labels, num_classes, out, accum=None, dlogits=None, class_w=None, mode=0, gamma=2.0, scale=1.0,
            aux=False):
    assert logits.dim() == 2 and logits.stride(1) == 1 and labels.dtype == torch.int64
    if dlogits is not None:
        assert dlogits.stride() == logits.stride()
    _lib().dpa_ce_loss(ptr(logits), logits.stride(0), ptr(labels), logits.shape[0], num_classes, ptr(class_w), int(mode),
                       F32(gamma), F32(scale), int(aux), ptr(dlogits), ptr(out), ptr(accum), _DT[logits.dtype],
                       stream_ptr())
    return out


# ----------------------------------------------------------------------------------------- optimizer
def _off(t, lo, esize):
    return VP(t.data_ptr() + lo * esize) if t is not None else VP(0)


def adam_step(p, g, m, v, p16, hyper, lo=0, hi=None):
    hi = p.numel() if hi is None else hi
    _lib().dpa_adam_step(_off(p, lo, 4), _off(g, lo, 4), _off(m, lo, 4), _off(v, lo, 4), _off(p16, lo, 2), ptr(hyper),
                         I64(hi - lo), stream_ptr())


def sgd_step(p, g, buf, p16, hyper, lo=0, hi=None):
    hi = p.numel() if hi is None else hi
    _lib().dpa_sgd_step(_off(p, lo, 4), _off(g, lo, 4), _off(buf, lo, 4), _off(p16, lo, 2), ptr(hyper), I64(hi - lo),
                        stream_ptr())


def advance_step(hyper):
    _lib().dpa_advance_step(ptr(hyper), stream_ptr())


def cast(src, dst):
    n = src.numel()
    if src.dtype == torch.float32 and dst.dtype == torch.bfloat16:
        _lib().dpa_cast_f32_bf16(ptr(src), ptr(dst), I64(n), stream_ptr())
    elif src.dtype == torch.bfloat16 and dst.dtype == torch.float32:
        _lib().dpa_cast_bf16_f32(ptr(src), ptr(dst), I64(n), stream_ptr())
    else:
        dst.copy_(src)
    return dst


# ----------------------------------------------------------------------------------------- input pipeline
def augment_mnist(images, labels, index, out, labels_out, train, mean, std, seed, epoch):
    p = AugP()
    p.B, p.S, p.D, p.Cpad = index.numel(), images.shape[1], out.shape[1], out.shape[3]
    p.train, p.mean, p.std = int(train), float(mean), float(std)
    p.seed, p.epoch = int(seed), int(epoch)
    assert images.dtype == torch.uint8 and index.dtype == torch.int32 and images.shape[1] <= 64
    assert out.is_contiguous() and labels.dtype == torch.int64
    _lib().dpa_augment_mnist(ptr(images), ptr(labels), ptr(index), ptr(out), ptr(labels_out), ctypes.byref(p),
                             _DT[out.dtype], stream_ptr())
    return out

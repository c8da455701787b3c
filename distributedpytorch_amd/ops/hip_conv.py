"""HIP implicit-GEMM convolution entry points (fprop / dgrad / wgrad) operating on NHWC torch tensors.

Tensors are memory handles only: `x[N,H,W,C]` may be a channel slice of a wider NHWC buffer (stride(2) is the
channel stride, stride(3) == 1), which is how concat is done without copies.  Filters are KRSC
`[Cout, kh, kw, Cin]` contiguous.  See csrc/hip/conv_igemm.hip and conv_wgrad.hip.
"""
from __future__ import annotations

import ctypes

import torch

from . import native
from .native import ConvParams, WgradParams, VP
from .ref_ops import CompactAddendUnsupported, SecondBnUnsupported   # noqa: F401  (launches that cannot take these extras)

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _check_nhwc(t, name):
    if t.dim() != 4 or t.stride(3) != 1:
        raise ValueError(f"{name}: expected NHWC tensor with contiguous channels, got strides {t.stride()}")
    n, h, w, c = t.shape
    cs = t.stride(2)
    if (w > 1 and t.stride(1) != w * cs) or (h > 1 and t.stride(0) != h * w * cs and n > 1):
        raise ValueError(f"{name}: only channel-sliced NHWC views are supported, got strides {t.stride()}")
    if t.data_ptr() % 16:
        raise ValueError(f"{name}: data pointer must be 16-byte aligned")
    return cs


def out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


last_stats_rows = [0]       # rows of the stats slab the most recent conv_fprop wrote (depends on the kernel / tile plan)


def split3_planes(w, out):
    """fp32 tensor (numel % 32 == 0) -> three bf16 planes in element order (csrc/hip/conv_x6p.hip): the B operand of the
    pre-split fp32 convolution kernel.  out: bf16 buffer of 3 * w.numel() elements."""
    assert w.dtype == torch.float32 and out.dtype == torch.bfloat16 and w.is_contiguous() and out.numel() >= 3 * w.numel()
    native.hip_lib().dpa_split3_planes(VP(w.data_ptr()), VP(out.data_ptr()), ctypes.c_longlong(w.numel()), native.stream_ptr())
    return out


def weight_prep(table, nsegs, total_chunks, dtype):
    """One launch for every registered per-step filter transformation (nn/wprep.py; csrc/hip/conv_x6p.hip:
    weight_prep_kernel).  table: int64 device tensor [nsegs, 7] in the layout of struct WPrepSeg."""
    lib = native.hip_lib()
    assert table.dtype == torch.int64 and table.is_contiguous() and lib.dpa_weight_prep_seg_size() == 56
    lib.dpa_weight_prep(VP(table.data_ptr()), ctypes.c_int(nsegs), ctypes.c_longlong(total_chunks), ctypes.c_int(_DT[dtype]),
                        native.stream_ptr())


def conv_fprop(x, w, y, stride, pad, bias=None, relu=False, stats=None, addend=None, stats_shift=None, w3=None, oscale=None):
    """y = conv(x, w) (+bias) (+addend) (relu).  stats: fp32 [rows, 2, Cout] slab written with per-block
    (sum, sumsq) of the stored outputs minus the per-channel `stats_shift` (0 if None); rows = stats_rows(M, Cout).
    oscale: fp32 per-output-channel multiplier applied to the accumulator before the bias (folded inference BatchNorm)."""
    lib = native.hip_lib()
    sh, sw = stride
    ph, pw = pad
    p = ConvParams()
    p.src, p.wgt, p.dst = VP(x.data_ptr()), VP(w.data_ptr()), VP(y.data_ptr())
    p.bias = native.ptr(bias)
    p.stats = native.ptr(stats)
    p.addend = native.ptr(addend)
    p.stats_shift = native.ptr(stats_shift)
    p.oscale = native.ptr(oscale)
    p.wgt3, p.wgt3_plane = native.ptr(w3), w.numel()
    p.sN, p.sH, p.sW, p.sC = x.shape
    p.s_cs, p.s_co = _check_nhwc(x, "x"), 0
    p.dH, p.dW, p.dC = y.shape[1], y.shape[2], y.shape[3]
    p.d_cs, p.d_co = _check_nhwc(y, "y"), 0
    p.R, p.S = w.shape[1], w.shape[2]
    p.sh, p.sw, p.ph, p.pw = sh, sw, ph, pw
    p.relu = int(relu)
    assert w.is_contiguous() and w.shape[0] == p.dC and w.shape[3] == p.sC, (w.shape, x.shape, y.shape)
    assert x.dtype == w.dtype == y.dtype
    if addend is not None:
        assert addend.stride() == y.stride()
    rc = lib.dpa_conv_igemm(ctypes.byref(p), _DT[x.dtype], 0, native.stream_ptr())
    if rc:
        raise RuntimeError(f"dpa_conv_igemm(fprop) rejected geometry rc={rc}: x{tuple(x.shape)} w{tuple(w.shape)}")
    last_stats_rows[0] = int(p.stats_rows_used)
    return y


def conv_dgrad(dy, wt, dx, stride, pad, addend=None, wt3=None, bn_reduce=None, addend2=None):
    """dx = conv_transpose(dy, w).  wt is the filter transposed to [Cin, kh, kw, Cout] (reduction-contiguous).
    addend (same geometry as dx) is added in the epilogue (gradient accumulation for fan-out).
    addend2: a COMPACT second gradient [N, ceil(H / 2), ceil(W / 2), C], contiguous - the gradient of a 1x1 / stride-2
    convolution over the same input without its zeros: added to the pixels with even coordinates.  Only the pre-split
    fp32 kernel implements it; otherwise CompactAddendUnsupported is raised and nothing has been launched.
    bn_reduce = (x, y, mean, invstd, mask_scale, mask_shift, slab[, bits[, (x2, mean2, invstd2, slab2)]]): dx is the COMPLETE gradient w.r.t. the output of a
    BatchNorm(+ReLU) with input x; the epilogue masks the values it stores (y > 0, or mask_scale * x + mask_shift > 0)
    and writes the per-tile partial sums (sum dz, sum dz * xhat) into slab [rows >= ceil(M / 64), 2, C].
    Returns the number of slab rows written (0 without bn_reduce)."""
    lib = native.hip_lib()
    sh, sw = stride
    ph, pw = pad
    p = ConvParams()
    p.src, p.wgt, p.dst = VP(dy.data_ptr()), VP(wt.data_ptr()), VP(dx.data_ptr())
    p.bias, p.stats, p.addend, p.stats_shift = VP(0), VP(0), native.ptr(addend), VP(0)
    p.wgt3, p.wgt3_plane = native.ptr(wt3), wt.numel()
    p.sN, p.sH, p.sW, p.sC = dy.shape
    p.s_cs, p.s_co = _check_nhwc(dy, "dy"), 0
    p.dH, p.dW, p.dC = dx.shape[1], dx.shape[2], dx.shape[3]
    p.d_cs, p.d_co = _check_nhwc(dx, "dx"), 0
    p.R, p.S = wt.shape[1], wt.shape[2]
    p.sh, p.sw, p.ph, p.pw = sh, sw, ph, pw
    p.relu = 0
    assert wt.is_contiguous() and wt.shape[0] == p.dC and wt.shape[3] == p.sC
    if addend is not None:
        assert addend.stride() == dx.stride()
    if addend2 is not None:
        assert addend is None and addend2.is_contiguous() and addend2.dtype == dx.dtype and addend2.dim() == 4
        assert tuple(addend2.shape) == (dx.shape[0], (dx.shape[1] + 1) // 2, (dx.shape[2] + 1) // 2, dx.shape[3])
        p.addend2 = VP(addend2.data_ptr())
        p.a2_H, p.a2_W, p.a2_cs = addend2.shape[1], addend2.shape[2], addend2.shape[3]
    if bn_reduce is not None:
        x, y, mean, invstd, mscale, mshift, slab = bn_reduce[:7]
        bits = bn_reduce[7] if len(bn_reduce) > 7 else None      # y's ReLU mask as bits (bn_apply(mask_bits=)): read instead of y
        if bits is not None:                                      # by the kernels that know them
            assert y is not None and dx.shape[3] % 32 == 0 and bits.dtype == torch.int32 and bits.is_contiguous() \
                and bits.numel() == dx.shape[0] * dx.shape[1] * dx.shape[2] * (dx.shape[3] // 32)
            p.bnr_bits = VP(bits.data_ptr())
        # a second BatchNorm whose output was summed with the first one's before the ReLU (same gradient, same mask): its
        # sum dz * xhat2 goes to slab2 [same rows, 2, C] (sum dz repeated in column 0).  Pre-split fp32 kernel + bits only:
        # SecondBnUnsupported otherwise, and nothing has been launched
        second = bn_reduce[8] if len(bn_reduce) > 8 else None
        if second is not None:
            x2, mean2, invstd2, slab2 = second
            assert bits is not None and x2.shape == dx.shape and x2.dtype == dx.dtype and slab2.shape == slab.shape \
                and slab2.dtype == torch.float32 and slab2.is_contiguous()
            p.bnr2_x, p.bnr2_cs = VP(x2.data_ptr()), _check_nhwc(x2, "bn2 x")
            p.bnr2_mean, p.bnr2_invstd, p.bnr2_slab = VP(mean2.data_ptr()), VP(invstd2.data_ptr()), VP(slab2.data_ptr())
        assert p.R >= sh and p.S >= sw, "fused BN reduction: the dgrad has to write every pixel"
        assert x.shape == dx.shape and x.dtype == dx.dtype and (y is None or (y.shape == dx.shape and y.dtype == dx.dtype))
        n_px = dx.shape[0] * dx.shape[1] * dx.shape[2]
        assert slab.dtype == torch.float32 and slab.is_contiguous() and slab.shape[0] >= bn_reduce_rows(n_px, stride) \
            and tuple(slab.shape[1:]) == (2, p.dC)
        p.bnr_x, p.bnr_cs = VP(x.data_ptr()), _check_nhwc(x, "bn x")
        p.bnr_y, p.bnr_y_cs = (VP(y.data_ptr()), _check_nhwc(y, "bn y")) if y is not None else (VP(0), 0)
        p.bnr_mean, p.bnr_invstd = VP(mean.data_ptr()), VP(invstd.data_ptr())
        p.bnr_mscale, p.bnr_mshift = native.ptr(mscale), native.ptr(mshift)
        p.bnr_slab = VP(slab.data_ptr())
    rc = lib.dpa_conv_igemm(ctypes.byref(p), _DT[dy.dtype], 1, native.stream_ptr())
    if rc == 4 and addend2 is not None:
        raise CompactAddendUnsupported("this launch does not take the pre-split fp32 kernel")
    if rc == 5:
        raise SecondBnUnsupported("this launch does not take the pre-split fp32 kernel")
    if rc:
        raise RuntimeError(f"dpa_conv_igemm(dgrad) rejected geometry rc={rc}")
    return int(p.stats_rows_used) if bn_reduce is not None else 0


def bn_reduce_rows(n_pixels, stride):
    """Upper bound on the slab rows a conv_dgrad(bn_reduce=) over n_pixels writes: one per M-tile (>= 64 rows) of each of
    the stride[0] * stride[1] parity classes."""
    classes = stride[0] * stride[1]
    return (n_pixels + 63) // 64 + classes


def wgrad_workspace_numel(cout, rsc):
    return int(native.hip_lib().dpa_conv_wgrad_workspace(cout, rsc))


def conv_wgrad(dy, x, dw, ksize, stride, pad, workspace=None, accumulate=False):
    """dw[Cout,kh,kw,Cin] (fp32) = sum over pixels of dy (x) im2col(x).  workspace: fp32 scratch for split-K."""
    lib = native.hip_lib()
    p = WgradParams()
    p.dy, p.x, p.out = VP(dy.data_ptr()), VP(x.data_ptr()), VP(0)
    p.N, p.H, p.W, p.C = x.shape
    p.c_cs, p.c_co = _check_nhwc(x, "x"), 0
    p.Ho, p.Wo, p.K = dy.shape[1], dy.shape[2], dy.shape[3]
    p.k_cs, p.k_co = _check_nhwc(dy, "dy"), 0
    p.R, p.S = ksize
    p.sh, p.sw = stride
    p.ph, p.pw = pad
    p.accumulate = int(accumulate)
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == p.K * p.R * p.S * p.C
    if workspace is not None:
        assert workspace.dtype == torch.float32 and workspace.numel() >= wgrad_workspace_numel(p.K, p.R * p.S * p.C)
    rc = lib.dpa_conv_wgrad(ctypes.byref(p), VP(dw.data_ptr()), native.ptr(workspace), _DT[x.dtype],
                            native.stream_ptr())
    if rc:
        raise RuntimeError(f"dpa_conv_wgrad rejected geometry rc={rc}")
    return dw


def stats_rows(m, cout, dtype=torch.float32):
    """Rows of the BN-statistics slab a conv_fprop over m rows x cout columns writes (depends on the tile plan, which
    depends on the dtype)."""
    return int(native.hip_lib().dpa_conv_stats_rows(int(m), int(cout), 1 if dtype == torch.bfloat16 else 0))

"""Op dispatch: tensors on the GPU run the hand-written gfx950 kernels (hip_ops); tensors on the CPU run the
plain-PyTorch reference (ref_ops, also the test oracle).  The first positional argument decides."""
from __future__ import annotations

import ctypes

import torch

from . import ref_ops

_hip_ops = None


def _hip():
    global _hip_ops
    if _hip_ops is None:
        from . import hip_ops as _h   # raises loudly if the native library is missing
        _h._lib()
        _hip_ops = _h
    return _hip_ops


_TENSOR_OPS = [
    "conv_fprop", "conv_dgrad", "conv_wgrad", "transpose_krsc",
    "bn_stats", "bn_finalize", "bn_eval_coeffs", "bn_apply", "bn_bwd_reduce", "bn_bwd_dx", "bn_bwd_dx2", "bn_bwd_from_slab",
    "relu_fwd", "relu_bwd", "relu_bwd_colsum", "add", "copy", "dropout", "rng_advance", "colsum",
    "maxpool_fwd", "maxpool_bwd", "maxpool_bwd_bn", "avgpool_fwd", "avgpool_bwd",
    "ce_loss", "adam_step", "sgd_step", "advance_step", "cast", "augment_mnist",
]


def _make(name):
    ref_fn = getattr(ref_ops, name)

    def op(t, *args, **kwargs):
        if t.is_cuda:
            return getattr(_hip(), name)(t, *args, **kwargs)
        return ref_fn(t, *args, **kwargs)

    op.__name__ = name
    op.__doc__ = ref_fn.__doc__
    return op


for _n in _TENSOR_OPS:
    globals()[_n] = _make(_n)

CompactAddendUnsupported = ref_ops.CompactAddendUnsupported
SecondBnUnsupported = ref_ops.SecondBnUnsupported
out_size = ref_ops.out_size
pool_out_size = ref_ops.pool_out_size


def stats_rows(m, cout, cuda, dtype=torch.float32):
    """Upper bound on the rows of the BN-statistics slab a conv_fprop over m rows writes (one row per M-tile; the
    smallest M-tile of any kernel variant is 64 rows).  The launch reports the rows it actually wrote:
    last_conv_stats_rows()."""
    return (m + 63) // 64 if cuda else 1


def last_conv_stats_rows(cuda):
    return _hip().last_stats_rows[0] if cuda else 1


def split3_planes(w, out):
    return _hip().split3_planes(w, out)


def weight_prep(table, nsegs, total_chunks, dtype):
    return _hip().weight_prep(table, nsegs, total_chunks, dtype)


def fp32_x6_active(cuda, dtype):
    """Is the pre-split-filter fast path of the fp32 convolutions worth preparing (GPU, fp32, x6 arithmetic)?"""
    if not cuda or dtype != torch.float32:
        return False
    from . import native
    return native.get_fp32_mode() == "x6"


def x6p_plausible(m_rows, cout):
    """Could a convolution with m_rows output rows and cout output channels take the pre-split fp32 kernel (size part of
    its eligibility rule, csrc/hip/conv_x6p.hip)?  GPU only - callers have checked fp32_x6_active."""
    from . import native
    return bool(native.hip_lib().dpa_x6p_plausible(ctypes.c_longlong(int(m_rows)), ctypes.c_int(int(cout))))


def bn_reduce_rows(n_pixels, stride, cuda):
    """Upper bound on the slab rows conv_dgrad(bn_reduce=...) writes."""
    return _hip().bn_reduce_rows(n_pixels, stride) if cuda else 1


def bn_slab_rows(m, cuda):
    return _hip().bn_slab_rows(m) if cuda else 1


def wgrad_workspace_numel(cout, rsc, cuda):
    return _hip().wgrad_workspace_numel(cout, rsc) if cuda else 1

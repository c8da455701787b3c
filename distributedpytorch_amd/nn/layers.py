"""Layers with hand-written backward passes over the ops in distributedpytorch_amd.ops (HIP on GPU).

Every layer follows the same protocol: `layer(ctx, x: Act, ...) -> Act`; when `ctx.tape` is set and the output
needs a gradient, the layer pushes ONE closure that consumes `y.grad`, writes parameter gradients straight into the
gradient arena (no zero_grad needed: gradients are written, not accumulated), announces them to the DDP reducer via
`tape.ready(spec)` and accumulates the input gradient.
Names of parameters follow torchvision so checkpoints interoperate with the reference (/root/reference/utils.py:112-136).
"""
from __future__ import annotations

import math
import os

import torch

from .. import ops
from .engine import BWD_LAG, Act, Context, ParamStore, accumulate_grad, pad_to, side_stream, workspace


# fewest GFLOP of a weight gradient that is worth the side stream's fork / join events (profiles/r2_side_stream_threshold.txt)
SIDE_MIN_GFLOP = float(os.environ.get("DPA_BWD_SIDE_MIN_GFLOP", "40"))

# DPA_FOLD_EVAL_BN=0: evaluation keeps the separate BatchNorm-apply pass (A/B switch)
_FOLD_EVAL_BN = os.environ.get("DPA_FOLD_EVAL_BN", "1") != "0"

# DPA_FUSE_BN_BWD=0: never fuse the BatchNorm-backward reduction into the dgrad epilogue (A/B switch)
_FUSE_BN_BWD = os.environ.get("DPA_FUSE_BN_BWD", "1") != "0"
# DPA_DEFER_DOWN_BN=0: the downsample branch of a ResNet block applies its BatchNorm itself (a pass over the activations of
# its own) instead of leaving it to the bn_apply of the block's last BatchNorm + add + ReLU
_DEFER_DOWN_BN = os.environ.get("DPA_DEFER_DOWN_BN", "1") != "0"
# DPA_BN_DX2=0: bn3 and the downsample BatchNorm of a ResNet block compute their dx in separate passes over the gradient
_BN_DX2 = os.environ.get("DPA_BN_DX2", "1") != "0"
# DPA_COMPACT_DOWN=0: the data gradient of a 1x1 / stride-2 downsample convolution is always a full-resolution tensor
# (A/B switch; see Conv2d._dgrad)
_COMPACT_DOWN = os.environ.get("DPA_COMPACT_DOWN", "1") != "0"
# DPA_RELU_BITS=0: no bit masks next to BatchNorm + residual + ReLU outputs; the fused dgrad epilogue reads y (A/B switch)
_RELU_BITS = os.environ.get("DPA_RELU_BITS", "1") != "0"
# DPA_FUSE_BN2=0: the BatchNorm of a downsample branch always runs its own backward reduction (A/B switch; see ConvBNAct)
_FUSE_BN2 = os.environ.get("DPA_FUSE_BN2", "1") != "0"


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# ------------------------------------------------------------------------------------------------ init helpers
def kaiming_normal_fan_out(t):
    torch.nn.init.kaiming_normal_(t, mode="fan_out", nonlinearity="relu")


def kaiming_uniform_default(t):          # nn.Conv2d / nn.Linear default
    torch.nn.init.kaiming_uniform_(t.view(t.shape[0], -1), a=math.sqrt(5))


def bias_default(fan_in):
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0

    def init(t):
        torch.nn.init.uniform_(t, -bound, bound)
    return init


def normal_(std):
    def init(t):
        torch.nn.init.normal_(t, 0.0, std)
    return init


def trunc_normal_(std):
    def init(t):
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std)
    return init


def zeros_(t):
    t.zero_()


def ones_(t):
    t.fill_(1.0)


# ------------------------------------------------------------------------------------------------ convolution
class Conv2d:
    """Convolution (also Linear: a kh x kw 'valid' conv over the whole map, or 1x1 on [N,1,1,C])."""

    def __init__(self, store: ParamStore, name, cin, cout, ksize, stride=1, padding=0, bias=True,
                 w_init=kaiming_uniform_default, b_init=None, trainable=True, linear=False):
        self.store = store
        self.cin, self.cout = cin, cout
        self.ksize, self.stride, self.padding = _pair(ksize), _pair(stride), _pair(padding)
        kh, kw = self.ksize
        self.w = store.add_conv_weight(f"{name}.weight", cout, cin, kh, kw, w_init, trainable, flat_logical=linear)
        self.b = None
        if bias:
            self.b = store.add_vec(f"{name}.bias", cout, b_init or bias_default(cin * kh * kw), trainable)
        self.trainable = trainable
        self._wt = None
        self._ev = None
        self._no_compact = False         # set when the dgrad after this layer's could not take a compact gradient
        self._no_bn2 = False             # set when this layer's dgrad could not take a second BatchNorm reduction
        self._w3 = self._wt3 = None      # the filter / transposed filter as three bf16 planes (fp32 x6 fast path)
        self.last_stats_rows = 1

    def _planes(self, ctx, w, which):
        """w (fp32, KRSC) split into three bf16 planes for the pre-split x6 kernel (csrc/hip/conv_x6p.hip); recomputed
        on every call because the optimizer has moved the weights (25 M parameters: ~0.1 ms per step in total)."""
        if w.shape[3] % 32 or not ops.fp32_x6_active(ctx.cuda, w.dtype):
            return None
        buf = getattr(self, which)
        if buf is None or buf.numel() != 3 * w.numel():
            buf = torch.empty(3 * w.numel(), device=w.device, dtype=torch.bfloat16)
            setattr(self, which, buf)
        return ops.split3_planes(w, buf)

    def _fwd_planes(self, ctx, w, m_rows):
        """The filter planes of this forward pass: from the batched per-step preparation (nn/wprep.py) when this pass ran
        it, else through the per-layer kernel.  None when the pre-split kernel cannot be taken anyway (a problem of m_rows
        output rows is too small for it: Linear layers at small batches, whose planes would be most of the preparation)."""
        if w.shape[3] % 32 or not ops.fp32_x6_active(ctx.cuda, w.dtype) or not ops.x6p_plausible(m_rows, w.shape[0]):
            return None
        r = self.store.wprep.get(ctx, (self.w.index, "f"), self.w, (w.shape[0], 1, w.numel() // w.shape[0]), False, True)
        if r is not None:
            self._w3 = None            # (the per-layer buffer of the registration step is no longer needed)
            return r["pl"]
        return self._planes(ctx, w, "_w3")

    def out_hw(self, h, w):
        return (ops.out_size(h, self.ksize[0], self.stride[0], self.padding[0]),
                ops.out_size(w, self.ksize[1], self.stride[1], self.padding[1]))

    def forward_raw(self, ctx: Context, x: torch.Tensor, out=None, relu=False, stats=None, stats_shift=None, fold=None):
        """fold = (scale, shift, addend): out = conv * scale[c] + shift[c] (+ addend) - an inference BatchNorm (and the
        residual connection) folded into the epilogue; only for convolutions without a bias of their own."""
        n, h, w, _ = x.shape
        ho, wo = self.out_hw(h, w)
        if out is None:
            out = ctx.empty((n, ho, wo, pad_to(self.cout)))
        bias = self.store.view(self.b) if self.b is not None else None
        kw = {}
        if fold is not None:
            assert bias is None
            bias = fold[1]
            kw = dict(oscale=fold[0], addend=fold[2])
        w = self.store.cview(self.w)
        if self._as_gemm(x):      # whole-map 'valid' kernel = plain GEMM over the flattened (h, w, c) features
            w = w.view(w.shape[0], 1, 1, -1)
            ops.conv_fprop(x.view(n, 1, 1, -1), w, out, (1, 1), (0, 0), bias=bias, relu=relu,
                           stats=stats, stats_shift=stats_shift, w3=self._fwd_planes(ctx, w, n), **kw)
        else:
            ops.conv_fprop(x, w, out, self.stride, self.padding, bias=bias, relu=relu, stats=stats, stats_shift=stats_shift,
                           w3=self._fwd_planes(ctx, w, n * ho * wo), **kw)
        if stats is not None:
            self.last_stats_rows = ops.last_conv_stats_rows(ctx.cuda)
        return out

    def _as_gemm(self, x: torch.Tensor) -> bool:
        """True when this is a Linear over a spatial map (kernel == map, no padding): NHWC activations and KRSC weights
        flatten to [N, h*w*c] and [Cout, h*w*c] without a copy, so fprop / dgrad / wgrad run as a 1x1 GEMM instead of a
        kh x kw convolution whose dgrad would visit kh*kw taps per input pixel with all but one masked (AlexNet fc1:
        4.2 ms -> one 64 x 9216 x 4096 GEMM)."""
        return (self.ksize != (1, 1) and tuple(x.shape[1:3]) == self.ksize and self.padding == (0, 0)
                and x.is_contiguous() and x.shape[3] == self.store.cview(self.w).shape[3])

    def backward_raw(self, ctx: Context, x: Act, dy: torch.Tensor, last_use=False, bias_done=False):
        """dy is the gradient w.r.t. the raw conv output (pre-activation).  Writes dW/db, accumulates dx.
        last_use: this dgrad is the LAST contribution to x.grad (every other consumer of x ran later in the forward
        pass) - it may then finish x's gradient for the BatchNorm that produced x (see _dgrad)."""
        st = self.store
        side = side_stream(ctx.device) if (self.trainable and x.needs_grad) else None
        if side is not None and SIDE_MIN_GFLOP > 0.0:
            # the fork / join events cost more than the overlap gains when the kernels are small (batch 64: DenseNet -23 %,
            # AlexNet -8 % with the fork on every layer): only weight gradients above a FLOP count take the side stream
            gflop = 2e-9 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * self.ksize[0] * self.ksize[1] * x.data.shape[3]
            if gflop < SIDE_MIN_GFLOP:
                side = None
        if side is not None:
            # fork: wgrad on the side stream, dgrad (and everything after it) on the main stream.  The join is deferred:
            # the main stream waits for this wgrad only BWD_LAG conv layers further down the tape (or at the end of
            # backward), so it overlaps the MFMA-bound dgrads AND the memory-bound BatchNorm-backward kernels in between.
            tape = ctx.tape
            tape.join_side(max(BWD_LAG - 1, 0))
            if self.b is not None and not bias_done:
                ops.colsum(dy, st.gview(self.b))      # (shares scratch with main-stream kernels: stays on the main stream)
            if self._ev is None:
                self._ev = (torch.cuda.Event(), torch.cuda.Event())
            main = torch.cuda.current_stream()
            self._ev[0].record(main)
            side.wait_event(self._ev[0])
            with torch.cuda.stream(side):
                self._wgrad(ctx, x, dy, bias=False)
                self._ev[1].record(side)
            tape.defer_side(self._ev[1], lambda: self._ready(ctx), (dy, x.data))
            self._dgrad(ctx, x, dy, last_use)
            return
        if self.trainable:
            if ctx.tape.side_pending:
                ctx.tape.join_side(0)        # this wgrad runs on the main stream and shares the split-K workspace
            self._wgrad(ctx, x, dy, bias=not bias_done)
            self._ready(ctx)
        if x.needs_grad:
            self._dgrad(ctx, x, dy, last_use)

    def _ready(self, ctx):
        if self.b is not None:
            ctx.tape.ready(self.b)
        ctx.tape.ready(self.w)

    def _wgrad(self, ctx, x, dy, bias=True):
        st = self.store
        if bias and self.b is not None:
            ops.colsum(dy, st.gview(self.b))
        rsc = self.ksize[0] * self.ksize[1] * x.data.shape[3]
        ws = workspace(ctx.device, ops.wgrad_workspace_numel(pad_to(self.cout), rsc, ctx.cuda), "wgrad") if ctx.cuda else None
        gw = st.gview(self.w)
        if self._as_gemm(x.data):
            ops.conv_wgrad(dy, x.data.view(x.data.shape[0], 1, 1, -1), gw.view(gw.shape[0], 1, 1, -1), (1, 1), (1, 1), (0, 0),
                           workspace=ws)
        else:
            ops.conv_wgrad(dy, x.data, gw, self.ksize, self.stride, self.padding, workspace=ws)

    def _bn_fusion(self, ctx, x, gemm):
        """x is the output of a BatchNorm+ReLU and this dgrad completes its gradient: let the epilogue apply the ReLU mask
        and write the partial sums of that BatchNorm's backward reduction (one memory pass over dy, y and z saved;
        csrc/hip/conv_common.h: BnReduceLane).  Needs a dgrad that writes every pixel (kernel >= stride)."""
        if not (_FUSE_BN_BWD and x.bn_src is not None and not gemm and x.grad_buf is None
                and self.ksize[0] >= self.stride[0] and self.ksize[1] >= self.stride[1]):
            return None
        # measured on ResNet-50, batch 1024 (profiles/r2_bn_bwd_fusion.txt): fp32 -6.5 ms per step (the removed kernel:
        # -17.6 ms, the dgrad epilogues: +12 ms - their reads are exposed at the end of each tile); bf16 -1.3 ms (92.9 ->
        # 91.6 ms; it was +2.5 ms before the finalize kernels were rewritten).  DPA_FUSE_BN_BWD=fp32 keeps bf16 unfused.
        if x.data.dtype != torch.float32 and os.environ.get("DPA_FUSE_BN_BWD") == "fp32":
            return None
        bn, z, mask_from_x = x.bn_src[:3]
        bits = x.bn_src[3] if len(x.bn_src) > 3 else None
        if z.shape != x.data.shape:
            return None
        a = bn.aux(ctx.device)
        n, h, w, c = x.data.shape
        rows = ops.bn_reduce_rows(n * h * w, self.stride, ctx.cuda)
        slab = ctx.empty((rows, 2, c), dtype=torch.float32)
        if mask_from_x:
            return (z, None, a[0], a[1], a[2], a[3], slab)
        fuse = (z, x.data, a[0], a[1], None, None, slab) + ((bits,) if bits is not None else ())
        if bits is not None and x.bn_src2 is not None and _FUSE_BN2 and not self._no_bn2:
            bn2, z2 = x.bn_src2
            if z2.shape == x.data.shape:
                a2 = bn2.aux(ctx.device)
                fuse = fuse + ((z2, a2[0], a2[1], ctx.empty((rows, 2, c), dtype=torch.float32)),)
        return fuse

    def _conv_dgrad(self, *args, bn_reduce=None, **kw):
        """ops.conv_dgrad; a launch that cannot take the second BatchNorm reduction (nothing was launched) is repeated
        without it, and the layer stops asking.  Returns (slab rows, second reduction done)."""
        if bn_reduce is not None and len(bn_reduce) > 8:
            try:
                return ops.conv_dgrad(*args, bn_reduce=bn_reduce, **kw), True
            except ops.SecondBnUnsupported:
                self._no_bn2 = True
                bn_reduce = bn_reduce[:8]
        return ops.conv_dgrad(*args, bn_reduce=bn_reduce, **kw), False

    def _dgrad(self, ctx, x, dy, last_use=False):
        w = self.store.cview(self.w)
        gemm = self._as_gemm(x.data) and (x.grad is None or x.grad.is_contiguous()) \
            and (x.grad_buf is None or x.grad_buf.is_contiguous())
        if gemm:
            w = w.view(w.shape[0], 1, 1, -1)
        k, r, s, c = w.shape
        rows = x.data.shape[0] if gemm else x.data.shape[0] * x.data.shape[1] * x.data.shape[2]
        want_planes = k % 32 == 0 and ops.fp32_x6_active(ctx.cuda, w.dtype) and ops.x6p_plausible(rows, c)
        prep = self.store.wprep.get(ctx, (self.w.index, "t"), self.w, (k, r * s, c), True, want_planes)
        if prep is not None:          # transposed filter (+ planes) made by this pass's batched preparation (nn/wprep.py)
            wt, wt3 = prep["dst"].view(c, r, s, k), prep["pl"]
            self._wt = self._wt3 = None
        else:
            if self._wt is None or self._wt.shape != (c, r, s, k):
                self._wt = torch.empty((c, r, s, k), device=w.device, dtype=w.dtype)
            ops.transpose_krsc(w, self._wt)
            wt, wt3 = self._wt, (self._planes(ctx, self._wt, "_wt3") if want_planes else None)
        stride, padding = ((1, 1), (0, 0)) if gemm else (self.stride, self.padding)
        flat = (lambda t: t.view(t.shape[0], 1, 1, -1)) if gemm else (lambda t: t)
        if (x.compact_ok and x.grad is None and x.grad_buf is None and x.grad_compact is None and not last_use and not gemm
                and _COMPACT_DOWN and not self._no_compact and wt3 is not None
                and self.ksize == (1, 1) and self.stride == (2, 2) and self.padding == (0, 0)):
            # a 1x1 / stride-2 convolution only reads the pixels with even coordinates: its data gradient is a plain GEMM
            # over those, [N, H/2, W/2, C].  Left compact for the dgrad that follows (the block's conv1), whose epilogue
            # adds it where it belongs - instead of writing a full-resolution tensor that is three quarters zeros and
            # reading it back as that dgrad's addend (ResNet-50 at batch 1024: 8.6 GB per step)
            g = ctx.empty((x.data.shape[0], dy.shape[1], dy.shape[2], x.data.shape[3]))
            ops.conv_dgrad(dy, wt, g, (1, 1), (0, 0), wt3=wt3)
            x.grad_compact = (g, self)
            return
        fuse = self._bn_fusion(ctx, x, gemm) if last_use else None
        if x.grad is None and x.grad_compact is not None:
            g, src = x.grad_compact
            x.grad_compact = None
            dx = ctx.empty(tuple(x.data.shape))
            try:
                if gemm:
                    raise ops.CompactAddendUnsupported("flattened whole-map convolution")
                rows, did2 = self._conv_dgrad(dy, wt, dx, stride, padding, wt3=wt3, bn_reduce=fuse, addend2=g)
            except ops.CompactAddendUnsupported:
                # (this dgrad does not take the pre-split fp32 kernel: small problem, other arithmetic.  Nothing was
                # launched; the compact gradient becomes the dense addend it replaces, and the layer stops making it)
                src._no_compact = True
                dx.zero_()
                dx[:, ::2, ::2, :] = g
                rows, did2 = self._conv_dgrad(dy, wt, flat(dx), stride, padding, addend=flat(dx), wt3=wt3, bn_reduce=fuse)
            x.grad = dx
        elif x.grad is None:
            dx = x.grad_buf if x.grad_buf is not None else ctx.empty(tuple(x.data.shape))
            rows, did2 = self._conv_dgrad(dy, wt, flat(dx), stride, padding, wt3=wt3, bn_reduce=fuse)
            x.grad = dx
        else:
            rows, did2 = self._conv_dgrad(dy, wt, flat(x.grad), stride, padding, addend=flat(x.grad), wt3=wt3, bn_reduce=fuse)
        if fuse is not None:
            x.bn_slab = fuse[6][:rows]
            if did2:
                x.bn_slab2 = fuse[8][3][:rows]

    def __call__(self, ctx: Context, x: Act, relu=False, out=None, last_use=False) -> Act:
        y = Act(self.forward_raw(ctx, x.data, out=out, relu=relu), x.needs_grad or self.trainable)
        if ctx.grad_enabled and y.needs_grad:
            def bw():
                dy = y.grad
                y.grad = None
                bias_done = False
                if relu and self.b is not None and self.trainable:
                    ops.relu_bwd_colsum(dy, y.data, self.store.gview(self.b))    # mask + bias gradient in one pass over dy
                    bias_done = True
                elif relu:
                    ops.relu_bwd(dy, y.data)
                self.backward_raw(ctx, x, dy, last_use=last_use, bias_done=bias_done)
            ctx.tape.push(bw)
        return y


class BatchNorm2d:
    def __init__(self, store: ParamStore, name, c, eps=1e-5, momentum=0.1, trainable=True):
        self.store, self.c, self.eps, self.momentum = store, c, eps, momentum
        self.gamma = store.add_vec(f"{name}.weight", c, ones_, trainable)
        self.beta = store.add_vec(f"{name}.bias", c, zeros_, trainable)
        self.rm = store.add_buffer(f"{name}.running_mean", c, 0.0)
        self.rv = store.add_buffer(f"{name}.running_var", c, 1.0)
        self.nbt = store.add_nbt(f"{name}.num_batches_tracked")
        self.trainable = trainable
        # persistent per-layer fp32 vectors: mean, invstd, scale, shift, sums[2], coef[3], statistics shift
        self._aux = torch.zeros(10, pad_to(c), device=store.device)
        store.aux_tensors.append(self._aux)      # (hipGraph capture snapshots / restores these around its warm-up)

    def aux(self, device):
        return self._aux

    def stat_shift(self, ctx: Context):
        """Per-channel shift c of the training statistics (a[9] = the previous step's batch mean; 0 before the first
        step): the conv epilogue / bn_stats accumulate sum(x-c), sum((x-c)^2), so from the second step on the variance
        does not cancel when |mean| >> std (E[x^2] - mean^2 in fp32)."""
        return self.aux(ctx.device)[9] if ctx.training else None

    def coeffs(self, ctx: Context, slab, count):
        """Finalize batch statistics (train) or fold running statistics (eval) into (scale, shift)."""
        st = self.store
        st.fence_buffers()          # the step's buffer broadcast (DDP X4) must have landed before the statistics are touched
        a = self.aux(ctx.device)
        if ctx.training:
            ops.bn_finalize(slab, count, st.view(self.gamma), st.view(self.beta), st.bview(self.rm), st.bview(self.rv),
                            st.nbt[self.nbt: self.nbt + 1], self.momentum, self.eps, a[0], a[1], a[2], a[3],
                            stat_shift=a[9])
        else:
            ops.bn_eval_coeffs(st.view(self.gamma), st.view(self.beta), st.bview(self.rm), st.bview(self.rv), self.eps,
                               a[2], a[3])
        return a

    def backward(self, ctx: Context, dy, y_for_mask, z, dx_out, mask_from_x=False, accumulate=False):
        """dz = dy*(y>0) in place; dgamma/dbeta -> arena; returns dx (written into dx_out).  mask_from_x (BN+ReLU with no
        residual, nobody else needs the masked dy): the mask is recomputed from z with the forward's folded
        coefficients a[2], a[3] - y is not read and dy is not rewritten."""
        st = self.store
        a = self.aux(ctx.device)
        dg = st.gview(self.gamma) if self.trainable else None
        db = st.gview(self.beta) if self.trainable else None
        ms, mh = (a[2], a[3]) if mask_from_x else (None, None)
        ops.bn_bwd_reduce(dy, None if mask_from_x else y_for_mask, z, a[0], a[1], st.view(self.gamma), dg, db, a[4:6], a[6:9],
                          mask_scale=ms, mask_shift=mh)
        if self.trainable:
            ctx.tape.ready(self.gamma)
            ctx.tape.ready(self.beta)
        ops.bn_bwd_dx(dy, z, a[6:9], dx_out, mask_scale=ms, mask_shift=mh, accumulate=accumulate)
        return dx_out


def _bn_finish_slab(bn, ctx, slab, m):
    """Parameter gradients and the dx coefficients (aux rows 6-8) of a BatchNorm from its partial sums."""
    st = bn.store
    a = bn.aux(ctx.device)
    dg = st.gview(bn.gamma) if bn.trainable else None
    db = st.gview(bn.beta) if bn.trainable else None
    ops.bn_bwd_from_slab(slab, m, a[0], a[1], st.view(bn.gamma), dg, db, a[4:6], a[6:9])
    if bn.trainable:
        ctx.tape.ready(bn.gamma)
        ctx.tape.ready(bn.beta)
    return a[6:9]


def _bn_backward_from_slab(bn, ctx, dy, z, dx_out, slab, m, accumulate=False, mask_from_x=False):
    """BatchNorm2d.backward when a dgrad epilogue already masked dy and reduced it (Act.bn_slab).  mask_from_x: the sums are
    of the masked gradient but dy itself is not masked (ConvBNReLUPool): bn_bwd_dx masks from z."""
    st = bn.store
    a = bn.aux(ctx.device)
    dg = st.gview(bn.gamma) if bn.trainable else None
    db = st.gview(bn.beta) if bn.trainable else None
    ops.bn_bwd_from_slab(slab, m, a[0], a[1], st.view(bn.gamma), dg, db, a[4:6], a[6:9])
    if bn.trainable:
        ctx.tape.ready(bn.gamma)
        ctx.tape.ready(bn.beta)
    ms, mh = (a[2], a[3]) if mask_from_x else (None, None)
    ops.bn_bwd_dx(dy, z, a[6:9], dx_out, mask_scale=ms, mask_shift=mh, accumulate=accumulate)
    return dx_out


def _stats_slab(ctx, m, c, conv_fused):
    rows = ops.stats_rows(m, c, ctx.cuda, ctx.dtype) if conv_fused else ops.bn_slab_rows(m, ctx.cuda)
    return workspace(ctx.device, rows * 2 * c, "bn_slab")[: rows * 2 * c].view(rows, 2, c)


class ConvBNAct:
    """conv -> BatchNorm -> (+ residual) -> (ReLU).  BN batch statistics come out of the conv epilogue."""

    def __init__(self, conv: Conv2d, bn: BatchNorm2d):
        self.conv, self.bn = conv, bn

    def __call__(self, ctx: Context, x: Act, relu=True, residual: Act = None, out=None, last_use=False,
                 defer_apply=False) -> Act:
        """last_use: no consumer of x comes after this layer in the backward pass, i.e. every other consumer of x (a
        residual connection, a parallel branch) is called LATER in the forward pass - see Conv2d.backward_raw.
        defer_apply (no ReLU, no residual): the caller hands the result to ONE ConvBNAct as its residual and nowhere else;
        the result may then be the convolution's output with Act.bn_defer = (scale, shift), and that consumer's bn_apply
        applies this BatchNorm on the fly - its output is never written or read."""
        conv, bn = self.conv, self.bn
        defer_apply = defer_apply and _DEFER_DOWN_BN and not relu and residual is None and out is None
        res_defer = residual.bn_defer if residual is not None else None
        n, h, w, _ = x.data.shape
        ho, wo = conv.out_hw(h, w)
        m, cp = n * ho * wo, pad_to(conv.cout)
        if _FOLD_EVAL_BN and not ctx.training and not ctx.grad_enabled and conv.b is None:
            # inference: the BatchNorm is a per-channel affine map of the conv output - applied in the conv epilogue together
            # with the residual connection and the ReLU, so evaluation runs no BatchNorm kernel over the activations at all
            if res_defer is not None:
                raise RuntimeError("a deferred BatchNorm output reached the folded evaluation path")
            a = bn.coeffs(ctx, None, m)
            if out is None:
                out = ctx.empty((n, ho, wo, cp))
            conv.forward_raw(ctx, x.data, out=out, relu=relu,
                             fold=(a[2], a[3], None if residual is None else residual.data))
            return Act(out, False)
        slab = _stats_slab(ctx, m, cp, True) if ctx.training else None
        z = conv.forward_raw(ctx, x.data, stats=slab, stats_shift=bn.stat_shift(ctx))
        if slab is not None:
            slab = slab[: conv.last_stats_rows]       # the rows the chosen kernel / tile plan actually wrote
        a = bn.coeffs(ctx, slab, m)
        if defer_apply:
            out = z
        elif out is None:
            out = ctx.empty((n, ho, wo, cp))
        needs = x.needs_grad or conv.trainable or bn.trainable or (residual is not None and residual.needs_grad)
        # BatchNorm + residual + ReLU in training: the ReLU mask also as bits (1 / 32 of y's bytes) for the dgrad that
        # will complete this activation's gradient and mask it in its epilogue (Conv2d._bn_fusion)
        bits = None
        if (_RELU_BITS and relu and residual is not None and ctx.training and ctx.grad_enabled and needs and _FUSE_BN_BWD
                and ctx.cuda and out.dtype in (torch.float32, torch.bfloat16) and cp % 32 == 0 and out.is_contiguous()):
            bits = ctx.empty((m, cp // 32), dtype=torch.int32)
        if not defer_apply:
            rs, rh = res_defer if res_defer is not None else (None, None)
            ops.bn_apply(z, a[2], a[3], out, residual=None if residual is None else residual.data, relu=relu, mask_bits=bits,
                         res_scale=rs, res_shift=rh)
        y = Act(out, needs)
        if defer_apply:
            y.bn_defer = (a[2], a[3])
        if ctx.grad_enabled and needs:
            mask_from_x = relu and residual is None and ctx.training
            if relu and ctx.training:
                # a consumer's dgrad may do this BatchNorm's backward reduction
                y.bn_src = (bn, z, mask_from_x) + ((bits,) if bits is not None else ())
                # ... and the one of a second BatchNorm under the same ReLU (the residual came straight out of one)
                if bits is not None and residual.bn_lin is not None and residual.needs_grad and residual.data.shape == out.shape:
                    y.bn_src2 = residual.bn_lin
            elif ctx.training and residual is None and not relu:
                y.bn_lin = (bn, z)

            def bw():
                dy = y.grad
                y.grad = None
                slab, y.bn_slab, y.bn_src = y.bn_slab, None, None
                slab2, y.bn_slab2, y.bn_src2, y.bn_lin = y.bn_slab2, None, None, None
                if y.bn_dx_done:           # the consumer ran this BatchNorm's backward together with its own: z holds dx
                    y.bn_dx_done = False
                    conv.backward_raw(ctx, x, z, last_use=last_use)
                    return
                lin = residual.bn_lin if residual is not None else None
                if (slab is not None and slab2 is not None and _BN_DX2 and lin is not None and residual.grad is None
                        and residual.grad_buf is None and residual.needs_grad and lin[1].shape == z.shape):
                    # both BatchNorms under this ReLU take dy as it is and both reductions are done: one pass for both dx
                    c1 = _bn_finish_slab(bn, ctx, slab, m)
                    c2 = _bn_finish_slab(lin[0], ctx, slab2, m)
                    ops.bn_bwd_dx2(dy, z, c1, z, lin[1], c2, lin[1])
                    dz, slab2 = z, None
                    residual.bn_dx_done = True
                elif slab is not None:       # dy arrived masked, its partial sums are in the slab
                    dz = _bn_backward_from_slab(bn, ctx, dy, z, z, slab, m)
                else:
                    dz = bn.backward(ctx, dy, y.data if relu else None, z, z,   # dx overwrites z (dead after this)
                                     mask_from_x=mask_from_x)
                if residual is not None and residual.needs_grad:
                    if slab2 is not None and residual.grad is None:
                        residual.bn_slab = slab2                             # the residual's BatchNorm: reduction done too
                    accumulate_grad(residual, dy)                            # dy now holds dy*(y>0)
                conv.backward_raw(ctx, x, dz, last_use=last_use)
            ctx.tape.push(bw)
        return y


class ConvBNReLUPool:
    """conv -> BatchNorm -> ReLU -> MaxPool as one unit (the ResNet / DenseNet stem).  Forward: the full-resolution
    BatchNorm output is never materialised - the pool reads the conv output through the folded BatchNorm coefficients +
    ReLU (`maxpool_fwd(scale=, shift=)`): one write and one read of a 3.3 GB tensor less per ResNet-50 step at batch 1024.
    Backward: max-pool backward into a full-resolution gradient, then the usual BatchNorm backward with the ReLU mask
    recomputed from the conv output.  (Gathering the pooled gradient through the argmax indices inside the two
    BatchNorm-backward kernels was measured too: 5.9 ms against 4.85 ms for the separate kernels - the gather runs
    twice - so it is not done.)"""

    def __init__(self, conv: Conv2d, bn: BatchNorm2d, pool: "MaxPool2d"):
        self.conv, self.bn, self.pool = conv, bn, pool

    def __call__(self, ctx: Context, x: Act, out_fn=None) -> Act:
        """out_fn(n, hp, wp) -> the tensor to pool into (e.g. a channel slice of a concat buffer); default: a new one."""
        conv, bn, pool = self.conv, self.bn, self.pool
        n, h, w, _ = x.data.shape
        ho, wo = conv.out_hw(h, w)
        m, cp = n * ho * wo, pad_to(conv.cout)
        slab = _stats_slab(ctx, m, cp, True) if ctx.training else None
        z = conv.forward_raw(ctx, x.data, stats=slab, stats_shift=bn.stat_shift(ctx))
        if slab is not None:
            slab = slab[: conv.last_stats_rows]
        a = bn.coeffs(ctx, slab, m)
        hp = ops.pool_out_size(ho, pool.ksize[0], pool.stride[0], pool.padding[0], pool.ceil_mode)
        wp = ops.pool_out_size(wo, pool.ksize[1], pool.stride[1], pool.padding[1], pool.ceil_mode)
        needs = x.needs_grad or conv.trainable or bn.trainable
        track = ctx.grad_enabled and needs
        out = ctx.empty((n, hp, wp, cp)) if out_fn is None else out_fn(n, hp, wp)
        idx = ctx.empty((n, hp, wp, cp), dtype=torch.uint8) if track else None
        ops.maxpool_fwd(z, out, idx, pool.ksize, pool.stride, pool.padding, scale=a[2], shift=a[3])
        y = Act(out, needs)
        if track:
            def bw():
                dyp = y.grad
                y.grad = None
                dy = ctx.empty((n, ho, wo, cp))
                # the pool's backward also makes the BatchNorm's backward reduction where its kernel can (3x3 / stride 2):
                # one read of the full-resolution gradient less
                rows = 0
                if _FUSE_BN_BWD and ctx.training:
                    a = bn.aux(ctx.device)
                    slab = ctx.empty((2048, 2, cp), dtype=torch.float32)
                    rows = ops.maxpool_bwd_bn(dyp, idx, dy, pool.ksize, pool.stride, pool.padding, z, a[2], a[3], a[0], a[1], slab)
                if rows:
                    dz = _bn_backward_from_slab(bn, ctx, dy, z, z, slab[:rows], m, mask_from_x=True)
                else:
                    ops.maxpool_bwd(dyp, idx, dy, pool.ksize, pool.stride, pool.padding)
                    dz = bn.backward(ctx, dy, None, z, z, mask_from_x=True)      # dx overwrites z (dead after this)
                conv.backward_raw(ctx, x, dz)
            ctx.tape.push(bw)
        return y


class BNAct:
    """Stand-alone BatchNorm (+ReLU) on an arbitrary (possibly channel-sliced) activation: DenseNet pre-activation."""

    def __init__(self, bn: BatchNorm2d):
        self.bn = bn

    def __call__(self, ctx: Context, x: Act, relu=True, out=None) -> Act:
        bn = self.bn
        n, h, w, c = x.data.shape
        m = n * h * w
        slab = None
        if ctx.training:
            slab = _stats_slab(ctx, m, c, False)
            ops.bn_stats(x.data, slab, shift=bn.stat_shift(ctx))
        a = bn.coeffs(ctx, slab, m)
        if out is None:
            out = ctx.empty((n, h, w, c))
        ops.bn_apply(x.data, a[2], a[3], out, relu=relu)
        needs = x.needs_grad or bn.trainable
        y = Act(out, needs)
        if ctx.grad_enabled and needs:
            if relu and ctx.training:
                y.bn_src = (bn, x.data, True)        # a consumer's dgrad may do this BatchNorm's backward reduction

            def bw():
                dy = y.grad
                y.grad = None
                slab, y.bn_slab, y.bn_src = y.bn_slab, None, None
                # dx goes straight into x's gradient: added to what later consumers already left there (a DenseNet concat
                # prefix), or written into its preassigned slot - no temporary, no separate add / copy kernel
                acc = False
                if x.needs_grad and x.grad is not None:
                    dx, acc = x.grad, True
                elif x.needs_grad and x.grad_buf is not None:
                    dx = x.grad_buf
                else:
                    dx = ctx.empty((n, h, w, c))
                if slab is not None:
                    _bn_backward_from_slab(bn, ctx, dy, x.data, dx, slab, m, accumulate=acc)
                else:
                    bn.backward(ctx, dy, y.data if relu else None, x.data, dx, mask_from_x=relu and ctx.training,
                                accumulate=acc)
                if x.needs_grad and x.grad is None:
                    x.grad = dx
            ctx.tape.push(bw)
        return y


class Linear(Conv2d):
    """nn.Linear as a convolution: [N,1,1,in] x 1x1, or (after a spatial map) [N,H,W,C] x (H x W) 'valid'."""

    def __init__(self, store, name, in_features, out_features, spatial=None, **kw):
        if spatial is None:
            super().__init__(store, name, in_features, out_features, 1, linear=True, **kw)
        else:
            c, h, w = spatial
            assert c * h * w == in_features
            super().__init__(store, name, c, out_features, (h, w), linear=True, **kw)


# ------------------------------------------------------------------------------------------------ pooling etc.
class MaxPool2d:
    def __init__(self, ksize, stride, padding=0, ceil_mode=False):
        self.ksize, self.stride, self.padding, self.ceil_mode = _pair(ksize), _pair(stride), _pair(padding), ceil_mode

    def __call__(self, ctx: Context, x: Act, out=None) -> Act:
        n, h, w, c = x.data.shape
        ho = ops.pool_out_size(h, self.ksize[0], self.stride[0], self.padding[0], self.ceil_mode)
        wo = ops.pool_out_size(w, self.ksize[1], self.stride[1], self.padding[1], self.ceil_mode)
        if out is None:
            out = ctx.empty((n, ho, wo, c))
        track = ctx.grad_enabled and x.needs_grad
        idx = ctx.empty((n, ho, wo, c), dtype=torch.uint8) if track else None
        ops.maxpool_fwd(x.data, out, idx, self.ksize, self.stride, self.padding)
        y = Act(out, x.needs_grad)
        if track:
            def bw():
                dy = y.grad
                y.grad = None
                dx = ctx.empty((n, h, w, c))
                ops.maxpool_bwd(dy, idx, dx, self.ksize, self.stride, self.padding)
                accumulate_grad(x, dx)
            ctx.tape.push(bw)
        return y


class AvgPool2d:
    """AvgPool2d(k, s, p) (count_include_pad=True) or AdaptiveAvgPool2d(output_size) when `adaptive` is given."""

    def __init__(self, ksize=None, stride=None, padding=0, adaptive=None):
        self.adaptive = _pair(adaptive) if adaptive is not None else None
        self.ksize = _pair(ksize) if ksize is not None else None
        self.stride = _pair(stride if stride is not None else ksize) if ksize is not None else None
        self.padding = _pair(padding)

    def __call__(self, ctx: Context, x: Act, out=None) -> Act:
        n, h, w, c = x.data.shape
        if self.adaptive is not None:
            ho, wo = self.adaptive
        else:
            ho = ops.pool_out_size(h, self.ksize[0], self.stride[0], self.padding[0])
            wo = ops.pool_out_size(w, self.ksize[1], self.stride[1], self.padding[1])
        if out is None:
            out = ctx.empty((n, ho, wo, c))
        ad = self.adaptive is not None
        ops.avgpool_fwd(x.data, out, self.ksize, self.stride, self.padding, adaptive=ad)
        y = Act(out, x.needs_grad)
        if ctx.grad_enabled and x.needs_grad:
            def bw():
                dy = y.grad
                y.grad = None
                dx = ctx.empty((n, h, w, c))
                ops.avgpool_bwd(dy, dx, self.ksize, self.stride, self.padding, adaptive=ad)
                accumulate_grad(x, dx)
            ctx.tape.push(bw)
        return y


class ReLU:
    def __call__(self, ctx: Context, x: Act, out=None) -> Act:
        if out is None:
            out = ctx.empty(tuple(x.data.shape))
        ops.relu_fwd(x.data, out)
        y = Act(out, x.needs_grad)
        if ctx.grad_enabled and x.needs_grad:
            def bw():
                dy = y.grad
                y.grad = None
                ops.relu_bwd(dy, y.data)
                accumulate_grad(x, dy)
            ctx.tape.push(bw)
        return y


_dropout_ids = [0]


class Dropout:
    """Inverted dropout with a Philox counter mask keyed by (seed, step, layer id); the backward pass regenerates
    the mask instead of storing it (K13 of SURVEY.md §2.5)."""

    def __init__(self, p=0.5):
        self.p = p
        _dropout_ids[0] += 1
        self.layer_id = _dropout_ids[0]

    def __call__(self, ctx: Context, x: Act) -> Act:
        if not ctx.training or self.p == 0.0:
            return x
        out = ctx.empty(tuple(x.data.shape))
        ops.dropout(x.data, out, self.p, ctx.rng, self.layer_id)
        y = Act(out, x.needs_grad)
        if ctx.grad_enabled and x.needs_grad:
            def bw():
                dy = y.grad
                y.grad = None
                ops.dropout(dy, dy, self.p, ctx.rng, self.layer_id)     # same mask, same scale
                accumulate_grad(x, dy)
            ctx.tape.push(bw)
        return y


def concat_buffer(ctx: Context, n, h, w, channels):
    """Allocate one wide NHWC buffer; branch i writes its output straight into slices[i] (zero-copy concat)."""
    total = sum(channels)
    buf = ctx.empty((n, h, w, total))
    slices, off = [], 0
    for c in channels:
        slices.append(buf[..., off: off + c])
        off += c
    return buf, slices


def concat(ctx: Context, buf: torch.Tensor, branches) -> Act:
    """branches: Acts whose .data are consecutive channel slices of `buf`."""
    needs = any(b.needs_grad for b in branches)
    y = Act(buf, needs)
    if ctx.grad_enabled and needs:
        def bw():
            g = y.grad
            y.grad = None
            off = 0
            for b in branches:
                c = b.data.shape[3]
                if b.needs_grad:
                    accumulate_grad(b, g[..., off: off + c])
                off += c
        ctx.tape.push(bw)
    return y

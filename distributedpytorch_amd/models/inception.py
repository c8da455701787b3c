"""Inception-v3 (/root/reference/utils.py:87-99 -> torchvision.models.inception_v3 with AuxLogits.fc and fc replaced;
299x299 input, aux logits on): SURVEY.md A.6.  In train mode forward returns (logits, aux_logits) and the trainer uses
loss = main + 0.4 * aux (classif.py:49-53); in eval mode it returns logits only (Q21).  Every BasicConv2d is
conv(no bias) -> BN(eps=1e-3) -> ReLU with the statistics fused into the conv epilogue; every branch writes straight
into its channel slice of the block's concat buffer (K12); 1x7 / 7x1 / 1x3 / 3x1 kernels use the same implicit-GEMM
kernel with asymmetric taps and padding."""
from __future__ import annotations

from .. import ops
from ..nn.engine import Act, Context, ParamStore
from ..nn.layers import (AvgPool2d, BatchNorm2d, Conv2d, ConvBNAct, Dropout, Linear, MaxPool2d, concat, concat_buffer,
                         trunc_normal_)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def BC(store, name, cin, cout, k, stride=1, pad=0, t=True, std=0.1):
    conv = Conv2d(store, f"{name}.conv", cin, cout, k, stride, pad, bias=False, w_init=trunc_normal_(std), trainable=t)
    bn = BatchNorm2d(store, f"{name}.bn", cout, eps=0.001, trainable=t)
    m = ConvBNAct(conv, bn)
    m.cout = cout
    return m


def _out_hw(h, w, k, s, p):
    k, s, p = _pair(k), _pair(s), _pair(p)
    return ops.out_size(h, k[0], s[0], p[0]), ops.out_size(w, k[1], s[1], p[1])


class InceptionA:
    def __init__(self, store, p, cin, pf, t):
        self.b1 = BC(store, f"{p}.branch1x1", cin, 64, 1, t=t)
        self.b5_1 = BC(store, f"{p}.branch5x5_1", cin, 48, 1, t=t)
        self.b5_2 = BC(store, f"{p}.branch5x5_2", 48, 64, 5, pad=2, t=t)
        self.d1 = BC(store, f"{p}.branch3x3dbl_1", cin, 64, 1, t=t)
        self.d2 = BC(store, f"{p}.branch3x3dbl_2", 64, 96, 3, pad=1, t=t)
        self.d3 = BC(store, f"{p}.branch3x3dbl_3", 96, 96, 3, pad=1, t=t)
        self.bp = BC(store, f"{p}.branch_pool", cin, pf, 1, t=t)
        self.pool = AvgPool2d(3, 1, 1)
        self.widths = (64, 64, 96, pf)

    def __call__(self, ctx, x):
        n, h, w, _ = x.data.shape
        buf, s = concat_buffer(ctx, n, h, w, self.widths)
        a = self.b1(ctx, x, out=s[0])
        b = self.b5_2(ctx, self.b5_1(ctx, x), out=s[1], last_use=True)
        c = self.d3(ctx, self.d2(ctx, self.d1(ctx, x), last_use=True), out=s[2], last_use=True)
        d = self.bp(ctx, self.pool(ctx, x), out=s[3])
        return concat(ctx, buf, [a, b, c, d])


class InceptionB:
    def __init__(self, store, p, cin, t):
        self.b3 = BC(store, f"{p}.branch3x3", cin, 384, 3, stride=2, t=t)
        self.d1 = BC(store, f"{p}.branch3x3dbl_1", cin, 64, 1, t=t)
        self.d2 = BC(store, f"{p}.branch3x3dbl_2", 64, 96, 3, pad=1, t=t)
        self.d3 = BC(store, f"{p}.branch3x3dbl_3", 96, 96, 3, stride=2, t=t)
        self.pool = MaxPool2d(3, 2)
        self.widths = (384, 96, cin)

    def __call__(self, ctx, x):
        n, h, w, _ = x.data.shape
        ho, wo = _out_hw(h, w, 3, 2, 0)
        buf, s = concat_buffer(ctx, n, ho, wo, self.widths)
        a = self.b3(ctx, x, out=s[0])
        b = self.d3(ctx, self.d2(ctx, self.d1(ctx, x), last_use=True), out=s[1], last_use=True)
        c = self.pool(ctx, x, out=s[2])
        return concat(ctx, buf, [a, b, c])


class InceptionC:
    def __init__(self, store, p, cin, c7, t):
        self.b1 = BC(store, f"{p}.branch1x1", cin, 192, 1, t=t)
        self.s1 = BC(store, f"{p}.branch7x7_1", cin, c7, 1, t=t)
        self.s2 = BC(store, f"{p}.branch7x7_2", c7, c7, (1, 7), pad=(0, 3), t=t)
        self.s3 = BC(store, f"{p}.branch7x7_3", c7, 192, (7, 1), pad=(3, 0), t=t)
        self.d1 = BC(store, f"{p}.branch7x7dbl_1", cin, c7, 1, t=t)
        self.d2 = BC(store, f"{p}.branch7x7dbl_2", c7, c7, (7, 1), pad=(3, 0), t=t)
        self.d3 = BC(store, f"{p}.branch7x7dbl_3", c7, c7, (1, 7), pad=(0, 3), t=t)
        self.d4 = BC(store, f"{p}.branch7x7dbl_4", c7, c7, (7, 1), pad=(3, 0), t=t)
        self.d5 = BC(store, f"{p}.branch7x7dbl_5", c7, 192, (1, 7), pad=(0, 3), t=t)
        self.bp = BC(store, f"{p}.branch_pool", cin, 192, 1, t=t)
        self.pool = AvgPool2d(3, 1, 1)
        self.widths = (192, 192, 192, 192)

    def __call__(self, ctx, x):
        n, h, w, _ = x.data.shape
        buf, s = concat_buffer(ctx, n, h, w, self.widths)
        a = self.b1(ctx, x, out=s[0])
        b = self.s3(ctx, self.s2(ctx, self.s1(ctx, x), last_use=True), out=s[1], last_use=True)
        c = self.d5(ctx, self.d4(ctx, self.d3(ctx, self.d2(ctx, self.d1(ctx, x), last_use=True), last_use=True), last_use=True),
                    out=s[2], last_use=True)
        d = self.bp(ctx, self.pool(ctx, x), out=s[3])
        return concat(ctx, buf, [a, b, c, d])


class InceptionD:
    def __init__(self, store, p, cin, t):
        self.a1 = BC(store, f"{p}.branch3x3_1", cin, 192, 1, t=t)
        self.a2 = BC(store, f"{p}.branch3x3_2", 192, 320, 3, stride=2, t=t)
        self.b1 = BC(store, f"{p}.branch7x7x3_1", cin, 192, 1, t=t)
        self.b2 = BC(store, f"{p}.branch7x7x3_2", 192, 192, (1, 7), pad=(0, 3), t=t)
        self.b3 = BC(store, f"{p}.branch7x7x3_3", 192, 192, (7, 1), pad=(3, 0), t=t)
        self.b4 = BC(store, f"{p}.branch7x7x3_4", 192, 192, 3, stride=2, t=t)
        self.pool = MaxPool2d(3, 2)
        self.widths = (320, 192, cin)

    def __call__(self, ctx, x):
        n, h, w, _ = x.data.shape
        ho, wo = _out_hw(h, w, 3, 2, 0)
        buf, s = concat_buffer(ctx, n, ho, wo, self.widths)
        a = self.a2(ctx, self.a1(ctx, x), out=s[0], last_use=True)
        b = self.b4(ctx, self.b3(ctx, self.b2(ctx, self.b1(ctx, x), last_use=True), last_use=True), out=s[1], last_use=True)
        c = self.pool(ctx, x, out=s[2])
        return concat(ctx, buf, [a, b, c])


class InceptionE:
    def __init__(self, store, p, cin, t):
        self.b1 = BC(store, f"{p}.branch1x1", cin, 320, 1, t=t)
        self.a1 = BC(store, f"{p}.branch3x3_1", cin, 384, 1, t=t)
        self.a2a = BC(store, f"{p}.branch3x3_2a", 384, 384, (1, 3), pad=(0, 1), t=t)
        self.a2b = BC(store, f"{p}.branch3x3_2b", 384, 384, (3, 1), pad=(1, 0), t=t)
        self.d1 = BC(store, f"{p}.branch3x3dbl_1", cin, 448, 1, t=t)
        self.d2 = BC(store, f"{p}.branch3x3dbl_2", 448, 384, 3, pad=1, t=t)
        self.d3a = BC(store, f"{p}.branch3x3dbl_3a", 384, 384, (1, 3), pad=(0, 1), t=t)
        self.d3b = BC(store, f"{p}.branch3x3dbl_3b", 384, 384, (3, 1), pad=(1, 0), t=t)
        self.bp = BC(store, f"{p}.branch_pool", cin, 192, 1, t=t)
        self.pool = AvgPool2d(3, 1, 1)
        self.widths = (320, 384, 384, 384, 384, 192)

    def __call__(self, ctx, x):
        n, h, w, _ = x.data.shape
        buf, s = concat_buffer(ctx, n, h, w, self.widths)
        o0 = self.b1(ctx, x, out=s[0])
        a = self.a1(ctx, x)
        o1 = self.a2a(ctx, a, out=s[1], last_use=True)      # a2b (called later) contributes to a.grad first in backward
        o2 = self.a2b(ctx, a, out=s[2])
        d = self.d2(ctx, self.d1(ctx, x), last_use=True)
        o3 = self.d3a(ctx, d, out=s[3], last_use=True)
        o4 = self.d3b(ctx, d, out=s[4])
        o5 = self.bp(ctx, self.pool(ctx, x), out=s[5])
        return concat(ctx, buf, [o0, o1, o2, o3, o4, o5])


class InceptionAux:
    def __init__(self, store, p, cin, num_classes, t):
        self.pool = AvgPool2d(5, 3)
        self.conv0 = BC(store, f"{p}.conv0", cin, 128, 1, t=t)
        self.conv1 = BC(store, f"{p}.conv1", 128, 768, 5, t=t, std=0.01)
        self.avg = AvgPool2d(adaptive=1)
        self.fc = Linear(store, f"{p}.fc", 768, num_classes, trainable=True)

    def __call__(self, ctx, x):
        x = self.conv1(ctx, self.conv0(ctx, self.pool(ctx, x)), last_use=True)
        return self.fc(ctx, self.avg(ctx, x))


class Inception3:
    def __init__(self, store: ParamStore, num_classes, feature_extract=False):
        t = not feature_extract
        self.c1a = BC(store, "Conv2d_1a_3x3", 3, 32, 3, stride=2, t=t)
        self.c2a = BC(store, "Conv2d_2a_3x3", 32, 32, 3, t=t)
        self.c2b = BC(store, "Conv2d_2b_3x3", 32, 64, 3, pad=1, t=t)
        self.c3b = BC(store, "Conv2d_3b_1x1", 64, 80, 1, t=t)
        self.c4a = BC(store, "Conv2d_4a_3x3", 80, 192, 3, t=t)
        self.mp = MaxPool2d(3, 2)
        self.m5b = InceptionA(store, "Mixed_5b", 192, 32, t)
        self.m5c = InceptionA(store, "Mixed_5c", 256, 64, t)
        self.m5d = InceptionA(store, "Mixed_5d", 288, 64, t)
        self.m6a = InceptionB(store, "Mixed_6a", 288, t)
        self.m6b = InceptionC(store, "Mixed_6b", 768, 128, t)
        self.m6c = InceptionC(store, "Mixed_6c", 768, 160, t)
        self.m6d = InceptionC(store, "Mixed_6d", 768, 160, t)
        self.m6e = InceptionC(store, "Mixed_6e", 768, 192, t)
        self.aux = InceptionAux(store, "AuxLogits", 768, num_classes, t)
        self.m7a = InceptionD(store, "Mixed_7a", 768, t)
        self.m7b = InceptionE(store, "Mixed_7b", 1280, t)
        self.m7c = InceptionE(store, "Mixed_7c", 2048, t)
        self.avg = AvgPool2d(adaptive=1)
        self.drop = Dropout(0.5)
        self.fc = Linear(store, "fc", 2048, num_classes, trainable=True)

    def forward(self, ctx: Context, x: Act):
        x = self.c2b(ctx, self.c2a(ctx, self.c1a(ctx, x), last_use=True), last_use=True)
        x = self.mp(ctx, x)
        x = self.c4a(ctx, self.c3b(ctx, x), last_use=True)
        x = self.mp(ctx, x)
        x = self.m5d(ctx, self.m5c(ctx, self.m5b(ctx, x)))
        x = self.m6a(ctx, x)
        x = self.m6e(ctx, self.m6d(ctx, self.m6c(ctx, self.m6b(ctx, x))))
        aux = self.aux(ctx, x) if ctx.training else None
        x = self.m7c(ctx, self.m7b(ctx, self.m7a(ctx, x)))
        x = self.fc(ctx, self.drop(ctx, self.avg(ctx, x)))
        return (x, aux) if aux is not None else x


def inception_v3(store, num_classes, feature_extract=False):
    return Inception3(store, num_classes, feature_extract)

"""DenseNet-121 (/root/reference/utils.py:78-85 -> torchvision.models.densenet121, classifier replaced): growth 32,
blocks (6, 12, 24, 16), bn_size 4, SURVEY.md A.5.

MI355X-first concat: each dense block owns ONE NHWC buffer of the block's final width; layer L reads the channel
prefix [0, c_L) and its 3x3 conv writes its 32 new channels at [c_L, c_L+32) in place - the torch.cat copies of the
reference implementation (quadratic traffic in the block depth) do not exist (K12).  The backward pass mirrors it
with one gradient buffer per block that every layer accumulates its input gradient into.
"""
from __future__ import annotations

from .. import ops
from ..nn.engine import Act, Context, ParamStore, accumulate_grad
from ..nn.layers import (AvgPool2d, BatchNorm2d, BNAct, Conv2d, ConvBNAct, ConvBNReLUPool, Linear, MaxPool2d,
                         kaiming_normal_fan_out, zeros_)


def _kn(t):
    import torch
    torch.nn.init.kaiming_normal_(t)


class DenseLayer:
    def __init__(self, store, prefix, cin, growth, bn_size, t):
        self.norm1 = BNAct(BatchNorm2d(store, f"{prefix}.norm1", cin, trainable=t))
        conv1 = Conv2d(store, f"{prefix}.conv1", cin, bn_size * growth, 1, bias=False, w_init=_kn, trainable=t)
        norm2 = BatchNorm2d(store, f"{prefix}.norm2", bn_size * growth, trainable=t)
        self.conv1_norm2 = ConvBNAct(conv1, norm2)
        self.conv2 = Conv2d(store, f"{prefix}.conv2", bn_size * growth, growth, 3, 1, 1, bias=False, w_init=_kn, trainable=t)


class DenseBlock:
    def __init__(self, store, prefix, n_layers, cin, growth, bn_size, t):
        self.cin, self.growth = cin, growth
        self.layers = [DenseLayer(store, f"{prefix}.denselayer{i + 1}", cin + i * growth, growth, bn_size, t)
                       for i in range(n_layers)]
        self.cout = cin + n_layers * growth
        self.trainable = t

    def alloc(self, ctx, n, h, w):
        return ctx.empty((n, h, w, self.cout))

    def __call__(self, ctx: Context, y0: Act, buf) -> Act:
        """y0.data must be buf[..., :cin] (written there by the producer).  Returns the Act of the full buffer."""
        st = {}
        rec = ctx.grad_enabled and (y0.needs_grad or self.trainable)
        c0 = self.cin
        if rec and y0.needs_grad:
            ctx.tape.push(lambda: accumulate_grad(y0, st["G"][..., :c0]))           # runs last
        c = c0
        for layer in self.layers:
            xl = Act(buf[..., :c], rec)
            if rec:
                def release(xl=xl):
                    xl.grad = None
                ctx.tape.push(release)                                                 # after this layer's norm1 backward
            h = layer.norm1(ctx, xl, relu=True)
            if rec:
                def preset(xl=xl, c=c):
                    # norm1's backward ADDS its input gradient straight into the block gradient's channel prefix
                    # (bn_bwd_dx accumulate mode): no temporary and no separate add kernel per layer
                    xl.grad = st["G"][..., :c]
                ctx.tape.push(preset)                                                  # before this layer's norm1 backward
            h = layer.conv1_norm2(ctx, h, relu=True, last_use=True)      # sole reader of norm1's output
            yl = layer.conv2(ctx, h, out=buf[..., c: c + self.growth], last_use=True)
            if rec:
                def set_grad(yl=yl, c=c):
                    yl.grad = st["G"][..., c: c + self.growth]
                ctx.tape.push(set_grad)                                                # before conv2's backward
            c += self.growth
        xt = Act(buf, rec)
        if rec:
            def adopt():
                st["G"], xt.grad = xt.grad, None       # the consumer's input gradient becomes the block gradient buffer
            ctx.tape.push(adopt)
        return xt


class Transition:
    def __init__(self, store, prefix, cin, cout, t):
        self.norm = BNAct(BatchNorm2d(store, f"{prefix}.norm", cin, trainable=t))
        self.conv = Conv2d(store, f"{prefix}.conv", cin, cout, 1, bias=False, w_init=_kn, trainable=t)
        self.pool = AvgPool2d(2, 2)

    def __call__(self, ctx, x: Act, out) -> Act:
        h = self.norm(ctx, x, relu=True)
        h = self.conv(ctx, h, last_use=True)
        return self.pool(ctx, h, out=out)


class DenseNet:
    def __init__(self, store: ParamStore, num_classes, feature_extract=False, growth=32, block_config=(6, 12, 24, 16),
                 init_features=64, bn_size=4):
        t = not feature_extract
        conv0 = Conv2d(store, "features.conv0", 3, init_features, 7, 2, 3, bias=False, w_init=_kn, trainable=t)
        self.stem = ConvBNAct(conv0, BatchNorm2d(store, "features.norm0", init_features, trainable=t))
        self.pool0 = MaxPool2d(3, 2, 1)
        self.stem_pool = ConvBNReLUPool(conv0, self.stem.bn, self.pool0)      # fused conv0-norm0-relu0-pool0 (nn/layers.py)
        self.blocks, self.transitions = [], []
        c = init_features
        for i, n in enumerate(block_config):
            blk = DenseBlock(store, f"features.denseblock{i + 1}", n, c, growth, bn_size, t)
            self.blocks.append(blk)
            c = blk.cout
            if i != len(block_config) - 1:
                self.transitions.append(Transition(store, f"features.transition{i + 1}", c, c // 2, t))
                c //= 2
        self.norm5 = BNAct(BatchNorm2d(store, "features.norm5", c, trainable=t))
        self.avg = AvgPool2d(adaptive=1)
        self.classifier = Linear(store, "classifier", c, num_classes, b_init=zeros_, trainable=True)

    def forward(self, ctx: Context, x: Act):
        bufs = []

        def first_slice(n_, hp, wp):
            bufs.append(self.blocks[0].alloc(ctx, n_, hp, wp))
            return bufs[0][..., : self.blocks[0].cin]
        y = self.stem_pool(ctx, x, out_fn=first_slice)
        buf = bufs[0]
        n, h, w = buf.shape[0], buf.shape[1], buf.shape[2]
        for i, blk in enumerate(self.blocks):
            xt = blk(ctx, y, buf)
            if i < len(self.transitions):
                h, w = h // 2, w // 2
                nxt = self.blocks[i + 1]
                buf = nxt.alloc(ctx, n, h, w)
                y = self.transitions[i](ctx, xt, buf[..., : nxt.cin])
        x = self.norm5(ctx, xt, relu=True)
        x = self.avg(ctx, x)
        return self.classifier(ctx, x)


def densenet121(store, num_classes, feature_extract=False):
    return DenseNet(store, num_classes, feature_extract)

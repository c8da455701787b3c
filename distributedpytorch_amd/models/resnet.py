"""ResNet-18 (the reference default, /root/reference/utils.py:42-49 -> torchvision.models.resnet18 with fc replaced)
and ResNet-50 (BASELINE.json config 5).  Parameter names, registration order and initialisation follow torchvision
 so `state_dict` checkpoints load `strict=True` into the reference's model.
"""
from __future__ import annotations

import os

from ..nn.engine import Act, Context, ParamStore
from ..nn.layers import (AvgPool2d, BatchNorm2d, Conv2d, ConvBNAct, ConvBNReLUPool, Linear, MaxPool2d,
                         kaiming_normal_fan_out)


# DPA_FUSE_STEM_POOL=0: stem as conv+BN+ReLU followed by a separate max-pool (A/B switch for nn/layers.py:ConvBNReLUPool)
_FUSE_STEM_POOL = os.environ.get("DPA_FUSE_STEM_POOL", "1") != "0"
# DPA_RESNET_DOWN_LAST=0: in a block with a downsample branch run that branch FIRST (its stride-2 1x1 dgrad then only
# accumulates into a quarter of the pixels of x.grad).  Default: conv1 first whenever the block's input came out of a
# BatchNorm + ReLU, so that conv1's dgrad is the LAST contribution to x.grad and carries that BatchNorm's backward
# reduction in its epilogue (nn/layers.py: Conv2d._bn_fusion) - one bn_bwd_reduce pass over dy, y and z of the previous
# block's output less, against a full-tensor write + read of the downsample gradient.
_DOWN_LAST = os.environ.get("DPA_RESNET_DOWN_LAST", "1") != "0"


def _cb(store, prefix, conv_name, bn_name, cin, cout, k, stride, pad, trainable):
    conv = Conv2d(store, f"{prefix}{conv_name}", cin, cout, k, stride, pad, bias=False, w_init=kaiming_normal_fan_out,
                  trainable=trainable)
    bn = BatchNorm2d(store, f"{prefix}{bn_name}", cout, trainable=trainable)
    return ConvBNAct(conv, bn)


class BasicBlock:
    expansion = 1

    def __init__(self, store, prefix, cin, planes, stride, trainable):
        self.c1 = _cb(store, prefix, "conv1", "bn1", cin, planes, 3, stride, 1, trainable)
        self.c2 = _cb(store, prefix, "conv2", "bn2", planes, planes, 3, 1, 1, trainable)
        self.down = None
        if stride != 1 or cin != planes:
            self.down = _cb(store, prefix, "downsample.0", "downsample.1", cin, planes, 1, stride, 0, trainable)

    def __call__(self, ctx: Context, x: Act, sole_consumer=False) -> Act:
        """sole_consumer: nothing but this block reads x (the caller's promise).  Then conv1's dgrad completes x.grad
        (the identity connection contributes earlier in the backward pass) and may carry the BatchNorm-backward
        reduction of the layer that produced x (ConvBNAct: last_use); conv2 is the only reader of conv1's output."""
        if self.down is not None and _DOWN_LAST and sole_consumer and getattr(x, "bn_src", None) is not None:
            out = self.c1(ctx, x, relu=True, last_use=True)       # see _DOWN_LAST
            x.compact_ok = True     # conv1's dgrad is the next (and last) contribution to x.grad after the downsample's
            identity = self.down(ctx, x, relu=False, defer_apply=True)
            return self.c2(ctx, out, relu=True, residual=identity, last_use=True)
        # the downsample branch runs FIRST so that, in the reversed backward order, its dgrad comes after conv1's and
        # accumulates in place into x.grad: the stride-2 1x1 dgrad then touches only 1/4 of the pixels
        identity = x if self.down is None else self.down(ctx, x, relu=False, defer_apply=True)
        out = self.c1(ctx, x, relu=True, last_use=sole_consumer and self.down is None)
        return self.c2(ctx, out, relu=True, residual=identity, last_use=True)


class Bottleneck:
    expansion = 4

    def __init__(self, store, prefix, cin, planes, stride, trainable):
        self.c1 = _cb(store, prefix, "conv1", "bn1", cin, planes, 1, 1, 0, trainable)
        self.c2 = _cb(store, prefix, "conv2", "bn2", planes, planes, 3, stride, 1, trainable)   # v1.5: stride on 3x3
        self.c3 = _cb(store, prefix, "conv3", "bn3", planes, planes * 4, 1, 1, 0, trainable)
        self.down = None
        if stride != 1 or cin != planes * 4:
            self.down = _cb(store, prefix, "downsample.0", "downsample.1", cin, planes * 4, 1, stride, 0, trainable)

    def __call__(self, ctx: Context, x: Act, sole_consumer=False) -> Act:
        if self.down is not None and _DOWN_LAST and sole_consumer and getattr(x, "bn_src", None) is not None:
            out = self.c1(ctx, x, relu=True, last_use=True)       # see _DOWN_LAST
            out = self.c2(ctx, out, relu=True, last_use=True)
            x.compact_ok = True     # conv1's dgrad is the next (and last) contribution to x.grad after the downsample's
            identity = self.down(ctx, x, relu=False, defer_apply=True)
            return self.c3(ctx, out, relu=True, residual=identity, last_use=True)
        identity = x if self.down is None else self.down(ctx, x, relu=False, defer_apply=True)      # first: see BasicBlock
        out = self.c1(ctx, x, relu=True, last_use=sole_consumer and self.down is None)
        out = self.c2(ctx, out, relu=True, last_use=True)
        return self.c3(ctx, out, relu=True, residual=identity, last_use=True)


class ResNet:
    def __init__(self, store: ParamStore, block, layers, num_classes, feature_extract=False, small_input=False):
        t = not feature_extract
        self.store = store
        self.num_classes = num_classes
        # small_input: CIFAR-style stem (3x3 stride 1, no max-pool) for 32x32 data (BASELINE.json config 4)
        if small_input:
            self.stem = _cb(store, "", "conv1", "bn1", 3, 64, 3, 1, 1, t)
            self.maxpool = None
        else:
            self.stem = _cb(store, "", "conv1", "bn1", 3, 64, 7, 2, 3, t)
            self.maxpool = MaxPool2d(3, 2, 1)
            self.stem_pool = ConvBNReLUPool(self.stem.conv, self.stem.bn, self.maxpool)      # fused stem (nn/layers.py)
        self.blocks = []
        cin = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            for bi in range(n):
                stride = 2 if (bi == 0 and li > 0) else 1
                self.blocks.append(block(store, f"layer{li + 1}.{bi}.", cin, planes, stride, t))
                cin = planes * block.expansion
        self.avgpool = AvgPool2d(adaptive=1)
        self.fc = Linear(store, "fc", cin, num_classes, bias=True, trainable=True)

    def forward(self, ctx: Context, x: Act):
        if self.maxpool is not None and _FUSE_STEM_POOL:
            x = self.stem_pool(ctx, x)
        else:
            x = self.stem(ctx, x, relu=True)
            if self.maxpool is not None:
                x = self.maxpool(ctx, x)
        for b in self.blocks:
            x = b(ctx, x, sole_consumer=True)
        x = self.avgpool(ctx, x)
        return self.fc(ctx, x)


def resnet18(store, num_classes, feature_extract=False, small_input=False):
    return ResNet(store, BasicBlock, (2, 2, 2, 2), num_classes, feature_extract, small_input)


def resnet50(store, num_classes, feature_extract=False, small_input=False):
    return ResNet(store, Bottleneck, (3, 4, 6, 3), num_classes, feature_extract, small_input)

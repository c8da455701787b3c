"""SqueezeNet 1.0 (/root/reference/utils.py:69-76 -> torchvision.models.squeezenet1_0, classifier[1] replaced by
Conv2d(512, classes, 1)): SURVEY.md A.4.  Fire's channel concat is zero-copy: expand1x1 and expand3x3 write into
channel slices of one buffer (K12).  Max-pools use ceil_mode=True."""
from __future__ import annotations

from ..nn.engine import Act, Context, ParamStore
from ..nn.layers import (AvgPool2d, Conv2d, Dropout, MaxPool2d, concat, concat_buffer, kaiming_uniform_default, normal_,
                         zeros_)


def _ku(t):
    import torch
    torch.nn.init.kaiming_uniform_(t)


class Fire:
    def __init__(self, store, prefix, cin, s, e1, e3, t):
        self.squeeze = Conv2d(store, f"{prefix}.squeeze", cin, s, 1, w_init=_ku, b_init=zeros_, trainable=t)
        self.e1 = Conv2d(store, f"{prefix}.expand1x1", s, e1, 1, w_init=_ku, b_init=zeros_, trainable=t)
        self.e3 = Conv2d(store, f"{prefix}.expand3x3", s, e3, 3, 1, 1, w_init=_ku, b_init=zeros_, trainable=t)
        self.e1c, self.e3c = e1, e3

    def __call__(self, ctx: Context, x: Act) -> Act:
        s = self.squeeze(ctx, x, relu=True)
        n, h, w, _ = s.data.shape
        buf, (a, b) = concat_buffer(ctx, n, h, w, (self.e1c, self.e3c))
        ya = self.e1(ctx, s, relu=True, out=a)
        yb = self.e3(ctx, s, relu=True, out=b)
        return concat(ctx, buf, [ya, yb])


class SqueezeNet:
    def __init__(self, store: ParamStore, num_classes, feature_extract=False):
        t = not feature_extract
        self.num_classes = num_classes
        self.conv0 = Conv2d(store, "features.0", 3, 96, 7, 2, w_init=_ku, b_init=zeros_, trainable=t)
        self.pool = MaxPool2d(3, 2, ceil_mode=True)
        fire = [(3, 96, 16, 64, 64), (4, 128, 16, 64, 64), (5, 128, 32, 128, 128), (7, 256, 32, 128, 128),
                (8, 256, 48, 192, 192), (9, 384, 48, 192, 192), (10, 384, 64, 256, 256), (12, 512, 64, 256, 256)]
        self.fires = {i: Fire(store, f"features.{i}", cin, s, e1, e3, t) for i, cin, s, e1, e3 in fire}
        self.drop = Dropout(0.5)
        self.final = Conv2d(store, "classifier.1", 512, num_classes, 1, w_init=kaiming_uniform_default, trainable=True)
        self.avg = AvgPool2d(adaptive=1)

    def forward(self, ctx: Context, x: Act):
        x = self.pool(ctx, self.conv0(ctx, x, relu=True))
        for i in (3, 4, 5):
            x = self.fires[i](ctx, x)
        x = self.pool(ctx, x)
        for i in (7, 8, 9, 10):
            x = self.fires[i](ctx, x)
        x = self.pool(ctx, x)
        x = self.fires[12](ctx, x)
        x = self.final(ctx, self.drop(ctx, x), relu=True)
        return self.avg(ctx, x)


def squeezenet1_0(store, num_classes, feature_extract=False):
    return SqueezeNet(store, num_classes, feature_extract)

"""VGG11-bn (/root/reference/utils.py:60-67 -> torchvision.models.vgg11_bn, classifier[6] replaced): SURVEY.md A.3.
features = (Conv3x3(bias) -> BN -> ReLU) with MaxPool(2,2) at each 'M'; torchvision indices are kept in the names."""
from __future__ import annotations

from ..nn.engine import Act, Context, ParamStore
from ..nn.layers import (AvgPool2d, BatchNorm2d, Conv2d, ConvBNAct, Dropout, Linear, MaxPool2d, kaiming_normal_fan_out,
                         normal_, zeros_)

CFG_A = [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"]


class VGG:
    def __init__(self, store: ParamStore, num_classes, feature_extract=False):
        t = not feature_extract
        self.features = []
        idx, cin = 0, 3
        for v in CFG_A:
            if v == "M":
                self.features.append(MaxPool2d(2, 2))
                idx += 1
            else:
                conv = Conv2d(store, f"features.{idx}", cin, v, 3, 1, 1, bias=True, w_init=kaiming_normal_fan_out,
                              b_init=zeros_, trainable=t)
                bn = BatchNorm2d(store, f"features.{idx + 1}", v, trainable=t)
                self.features.append(ConvBNAct(conv, bn))
                idx += 3
                cin = v
        self.avgpool = AvgPool2d(adaptive=(7, 7))
        self.c0 = Linear(store, "classifier.0", 512 * 7 * 7, 4096, spatial=(512, 7, 7), w_init=normal_(0.01), b_init=zeros_,
                         trainable=t)
        self.c3 = Linear(store, "classifier.3", 4096, 4096, w_init=normal_(0.01), b_init=zeros_, trainable=t)
        self.c6 = Linear(store, "classifier.6", 4096, num_classes, trainable=True)
        self.d2, self.d5 = Dropout(0.5), Dropout(0.5)

    def forward(self, ctx: Context, x: Act):
        for f in self.features:      # a plain chain: every conv is the sole reader of its input
            x = f(ctx, x, last_use=True) if isinstance(f, ConvBNAct) else f(ctx, x)
        x = self.avgpool(ctx, x)
        x = self.d2(ctx, self.c0(ctx, x, relu=True))
        x = self.d5(ctx, self.c3(ctx, x, relu=True))
        return self.c6(ctx, x)


def vgg11_bn(store, num_classes, feature_extract=False):
    return VGG(store, num_classes, feature_extract)

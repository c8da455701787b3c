"""AlexNet (/root/reference/utils.py:51-58 -> torchvision.models.alexnet, classifier[6] replaced): SURVEY.md A.2.
All convs have bias, no BN; ReLU is fused into the conv/linear epilogues; dropout is Philox-regenerated."""
from __future__ import annotations

from ..nn.engine import Act, Context, ParamStore
from ..nn.layers import AvgPool2d, Conv2d, Dropout, Linear, MaxPool2d


class AlexNet:
    def __init__(self, store: ParamStore, num_classes, feature_extract=False):
        t = not feature_extract
        self.f0 = Conv2d(store, "features.0", 3, 64, 11, 4, 2, trainable=t)
        self.f3 = Conv2d(store, "features.3", 64, 192, 5, 1, 2, trainable=t)
        self.f6 = Conv2d(store, "features.6", 192, 384, 3, 1, 1, trainable=t)
        self.f8 = Conv2d(store, "features.8", 384, 256, 3, 1, 1, trainable=t)
        self.f10 = Conv2d(store, "features.10", 256, 256, 3, 1, 1, trainable=t)
        self.pool = MaxPool2d(3, 2)
        self.avgpool = AvgPool2d(adaptive=(6, 6))
        self.d0, self.d3 = Dropout(0.5), Dropout(0.5)
        self.c1 = Linear(store, "classifier.1", 256 * 6 * 6, 4096, spatial=(256, 6, 6), trainable=t)
        self.c4 = Linear(store, "classifier.4", 4096, 4096, trainable=t)
        self.c6 = Linear(store, "classifier.6", 4096, num_classes, trainable=True)

    def forward(self, ctx: Context, x: Act):
        x = self.pool(ctx, self.f0(ctx, x, relu=True))
        x = self.pool(ctx, self.f3(ctx, x, relu=True))
        x = self.f6(ctx, x, relu=True)
        x = self.f8(ctx, x, relu=True)
        x = self.pool(ctx, self.f10(ctx, x, relu=True))
        x = self.avgpool(ctx, x)
        x = self.c1(ctx, self.d0(ctx, x), relu=True)
        x = self.c4(ctx, self.d3(ctx, x), relu=True)
        return self.c6(ctx, x)


def alexnet(store, num_classes, feature_extract=False):
    return AlexNet(store, num_classes, feature_extract)

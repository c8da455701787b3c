"""Bring-up CNN (`MODEL_NAME='basic'`; not in the reference zoo): conv-conv-pool-dropout-fc-fc
on 28x28 inputs, the classic PyTorch MNIST example.  Exercises conv+bias+ReLU epilogues, max-pool, Philox dropout and
Linear-after-spatial-map in seconds; takes the pipeline's 3 replicated channels."""
from __future__ import annotations

from ..nn.engine import Act, Context, ParamStore
from ..nn.layers import Conv2d, Dropout, Linear, MaxPool2d


class BasicCNN:
    in_channels = 3

    def __init__(self, store: ParamStore, num_classes):
        self.conv1 = Conv2d(store, "conv1", 3, 32, 3)
        self.conv2 = Conv2d(store, "conv2", 32, 64, 3)
        self.pool = MaxPool2d(2, 2)
        self.drop1, self.drop2 = Dropout(0.25), Dropout(0.5)
        self.fc1 = Linear(store, "fc1", 64 * 12 * 12, 128, spatial=(64, 12, 12))
        self.fc2 = Linear(store, "fc2", 128, num_classes)

    def forward(self, ctx: Context, x: Act):
        x = self.conv1(ctx, x, relu=True)
        x = self.conv2(ctx, x, relu=True)
        x = self.pool(ctx, x)
        x = self.drop1(ctx, x)
        x = self.fc1(ctx, x, relu=True)
        x = self.drop2(ctx, x)
        return self.fc2(ctx, x)


def basic_cnn(store, num_classes):
    return BasicCNN(store, num_classes)

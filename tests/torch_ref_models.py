"""Plain torch.nn restatements of the torchvision architectures the reference instantiates (torchvision itself is not
installed here).  Used only as test oracles: same module names => same state_dict keys."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return F.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return F.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=10, small_input=False):
        super().__init__()
        if small_input:
            self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        else:
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.small_input = small_input
        cin = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            blocks = []
            for bi in range(n):
                blocks.append(block(cin, planes, 2 if (bi == 0 and li > 0) else 1))
                cin = planes * block.expansion
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        self.fc = nn.Linear(cin, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        if not self.small_input:
            x = F.max_pool2d(x, 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = F.adaptive_avg_pool2d(x, 1).flatten(1)
        return self.fc(x)


def resnet18(num_classes=10, **kw):
    return ResNet(BasicBlock, (2, 2, 2, 2), num_classes, **kw)


def resnet50(num_classes=10, **kw):
    return ResNet(Bottleneck, (3, 4, 6, 3), num_classes, **kw)


# ------------------------------------------------------------------------------------------------ AlexNet / VGG / SqueezeNet
class AlexNet(nn.Module):
    def __init__(self, num_classes=10):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2))
        self.avgpool = nn.AdaptiveAvgPool2d((6, 6))
        self.classifier = nn.Sequential(nn.Dropout(), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(inplace=True), nn.Dropout(),
                                        nn.Linear(4096, 4096), nn.ReLU(inplace=True), nn.Linear(4096, num_classes))

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


class VGG11BN(nn.Module):
    def __init__(self, num_classes=10):
        super().__init__()
        layers, cin = [], 3
        for v in [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"]:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096),
                                        nn.ReLU(True), nn.Dropout(), nn.Linear(4096, num_classes))

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


class Fire(nn.Module):
    def __init__(self, cin, s, e1, e3):
        super().__init__()
        self.squeeze = nn.Conv2d(cin, s, 1)
        self.squeeze_activation = nn.ReLU(inplace=True)
        self.expand1x1 = nn.Conv2d(s, e1, 1)
        self.expand1x1_activation = nn.ReLU(inplace=True)
        self.expand3x3 = nn.Conv2d(s, e3, 3, padding=1)
        self.expand3x3_activation = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.squeeze_activation(self.squeeze(x))
        return torch.cat([self.expand1x1_activation(self.expand1x1(x)), self.expand3x3_activation(self.expand3x3(x))], 1)


class SqueezeNet10(nn.Module):
    def __init__(self, num_classes=10):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 96, 7, 2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, ceil_mode=True),
            Fire(96, 16, 64, 64), Fire(128, 16, 64, 64), Fire(128, 32, 128, 128), nn.MaxPool2d(3, 2, ceil_mode=True),
            Fire(256, 32, 128, 128), Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256),
            nn.MaxPool2d(3, 2, ceil_mode=True), Fire(512, 64, 256, 256))
        self.classifier = nn.Sequential(nn.Dropout(0.5), nn.Conv2d(512, num_classes, 1), nn.ReLU(inplace=True),
                                        nn.AdaptiveAvgPool2d((1, 1)))

    def forward(self, x):
        return torch.flatten(self.classifier(self.features(x)), 1)


# ------------------------------------------------------------------------------------------------ DenseNet-121
class _DenseLayer(nn.Module):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1)
        return self.conv2(self.relu2(self.norm2(self.conv1(self.relu1(self.norm1(x))))))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, n, cin, growth, bn_size):
        super().__init__()
        for i in range(n):
            self.add_module(f"denselayer{i + 1}", _DenseLayer(cin + i * growth, growth, bn_size))

    def forward(self, x):
        feats = [x]
        for _, layer in self.items():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = nn.BatchNorm2d(cin)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.pool = nn.AvgPool2d(2, 2)


class DenseNet121(nn.Module):
    def __init__(self, num_classes=10, growth=32, cfg=(6, 12, 24, 16), init=64, bn_size=4):
        super().__init__()
        from collections import OrderedDict
        self.features = nn.Sequential(OrderedDict([
            ("conv0", nn.Conv2d(3, init, 7, 2, 3, bias=False)), ("norm0", nn.BatchNorm2d(init)),
            ("relu0", nn.ReLU(inplace=True)), ("pool0", nn.MaxPool2d(3, 2, 1))]))
        c = init
        for i, n in enumerate(cfg):
            self.features.add_module(f"denseblock{i + 1}", _DenseBlock(n, c, growth, bn_size))
            c += n * growth
            if i != len(cfg) - 1:
                self.features.add_module(f"transition{i + 1}", _Transition(c, c // 2))
                c //= 2
        self.features.add_module("norm5", nn.BatchNorm2d(c))
        self.classifier = nn.Linear(c, num_classes)

    def forward(self, x):
        out = F.relu(self.features(x), inplace=True)
        return self.classifier(torch.flatten(F.adaptive_avg_pool2d(out, (1, 1)), 1))


# ------------------------------------------------------------------------------------------------ Inception-v3
class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class IncA(nn.Module):
    def __init__(self, cin, pf):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1 = BasicConv2d(cin, 48, kernel_size=1)
        self.branch5x5_2 = BasicConv2d(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, padding=1)
        self.branch_pool = BasicConv2d(cin, pf, kernel_size=1)

    def forward(self, x):
        return torch.cat([self.branch1x1(x), self.branch5x5_2(self.branch5x5_1(x)),
                          self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                          self.branch_pool(F.avg_pool2d(x, 3, 1, 1))], 1)


class IncB(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        return torch.cat([self.branch3x3(x), self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                          F.max_pool2d(x, 3, 2)], 1)


class IncC(nn.Module):
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7_2 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        d = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))))
        return torch.cat([self.branch1x1(x), b, d, self.branch_pool(F.avg_pool2d(x, 3, 1, 1))], 1)


class IncD(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch3x3_2 = BasicConv2d(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        return torch.cat([self.branch3x3_2(self.branch3x3_1(x)), b, F.max_pool2d(x, 3, 2)], 1)


class IncE(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        a = self.branch3x3_1(x)
        a = torch.cat([self.branch3x3_2a(a), self.branch3x3_2b(a)], 1)
        d = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        d = torch.cat([self.branch3x3dbl_3a(d), self.branch3x3dbl_3b(d)], 1)
        return torch.cat([self.branch1x1(x), a, d, self.branch_pool(F.avg_pool2d(x, 3, 1, 1))], 1)


class IncAux(nn.Module):
    def __init__(self, cin, num_classes):
        super().__init__()
        self.conv0 = BasicConv2d(cin, 128, kernel_size=1)
        self.conv1 = BasicConv2d(128, 768, kernel_size=5)
        self.fc = nn.Linear(768, num_classes)

    def forward(self, x):
        x = self.conv1(self.conv0(F.avg_pool2d(x, 5, 3)))
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1))


class Inception3(nn.Module):
    def __init__(self, num_classes=10):
        super().__init__()
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.Mixed_5b, self.Mixed_5c, self.Mixed_5d = IncA(192, 32), IncA(256, 64), IncA(288, 64)
        self.Mixed_6a = IncB(288)
        self.Mixed_6b, self.Mixed_6c, self.Mixed_6d, self.Mixed_6e = IncC(768, 128), IncC(768, 160), IncC(768, 160), IncC(768, 192)
        self.AuxLogits = IncAux(768, num_classes)
        self.Mixed_7a = IncD(768)
        self.Mixed_7b, self.Mixed_7c = IncE(1280), IncE(2048)
        self.dropout = nn.Dropout(0.5)
        self.fc = nn.Linear(2048, num_classes)

    def forward(self, x):
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x)))
        x = F.max_pool2d(x, 3, 2)
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x))
        x = F.max_pool2d(x, 3, 2)
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(x)))))
        aux = self.AuxLogits(x) if self.training else None
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))
        x = self.fc(self.dropout(torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)))
        return (x, aux) if self.training else x

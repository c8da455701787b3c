"""Model zoo factory with the reference's interface (/root/reference/utils.py:24-105): six torchvision architectures
with the classifier head swapped to `num_classes`, plus `resnet50` (BASELINE.json config 5) and `basic`
(bring-up CNN)."""
from __future__ import annotations

import logging

import torch

from ..nn.engine import Act, Context, ParamStore, Tape, input_pad, pad_to

MODEL_NAMES = ("resnet", "alexnet", "vgg", "squeezenet", "densenet", "inception", "resnet50", "basic")


def get_model_input_size(model_name):
    """utils.getModelInputSize: 224 for everything but inception (299)."""
    if model_name == "inception":
        return 299
    if model_name == "basic":
        return 28
    if model_name in MODEL_NAMES:
        return 224
    return None


class Network:
    """A built model: parameter store + forward definition + bookkeeping the trainer needs."""

    def __init__(self, name, store: ParamStore, net, num_classes, input_size):
        self.name, self.store, self.net = name, store, net
        self.num_classes, self.input_size = num_classes, input_size
        self.training = True
        self.rng = torch.zeros(2, dtype=torch.int64, device=store.device)   # (seed, step) for dropout masks

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    @property
    def in_channels(self):
        return input_pad(getattr(self.net, "in_channels", 3), self.store.compute_dtype)

    def forward(self, x: torch.Tensor, tape: Tape = None):
        """x: NHWC activation in the compute dtype with channels padded to `in_channels`."""
        ctx = Context(self.store, self.training, tape, rng=self.rng)
        self.store.wprep.run(ctx)       # one launch for every filter transpose / plane split the layers registered
        out = self.net.forward(ctx, Act(x, False))
        return ctx, out

    def state_dict(self, prefix="module."):
        """Keys carry the DDP 'module.' prefix by default, as the reference saves them (utils.py:116, Q16)."""
        return self.store.state_dict(prefix)

    def load_state_dict(self, sd, strict=True):
        self.store.load_state_dict(sd, strict)

    def seed_rng(self, seed):
        self.rng[0] = int(seed)
        self.rng[1] = 0


def get_model(model_name, num_classes, feature_extract=False, use_pretrained=False, device="cpu",
              dtype=torch.float32, small_input=False):
    """utils.getModel equivalent.  Returns (Network, input_size).  `use_pretrained` needs torchvision weights, which
    cannot be downloaded here (no network): it is accepted for interface parity and rejected if True."""
    if use_pretrained:
        raise RuntimeError("USE_PRETRAINED=True needs downloadable torchvision weights; load a checkpoint with -f instead")
    store = ParamStore(device, dtype)
    from ..nn import layers as _layers
    _layers._dropout_ids[0] = 0          # dropout stream ids are per model: two identical builds draw identical masks
    if model_name == "resnet":
        from .resnet import resnet18
        net = resnet18(store, num_classes, feature_extract, small_input)
    elif model_name == "resnet50":
        from .resnet import resnet50
        net = resnet50(store, num_classes, feature_extract, small_input)
    elif model_name == "alexnet":
        from .alexnet import alexnet
        net = alexnet(store, num_classes, feature_extract)
    elif model_name == "vgg":
        from .vgg import vgg11_bn
        net = vgg11_bn(store, num_classes, feature_extract)
    elif model_name == "squeezenet":
        from .squeezenet import squeezenet1_0
        net = squeezenet1_0(store, num_classes, feature_extract)
    elif model_name == "densenet":
        from .densenet import densenet121
        net = densenet121(store, num_classes, feature_extract)
    elif model_name == "inception":
        from .inception import inception_v3
        net = inception_v3(store, num_classes, feature_extract)
    elif model_name == "basic":
        from .basic import basic_cnn
        net = basic_cnn(store, num_classes)
    else:
        logging.error("Invalid model name, exiting...")
        raise SystemExit(1)
    store.finalize()
    size = get_model_input_size(model_name)
    if small_input:
        size = 32
    return Network(model_name, store, net, num_classes, size), size

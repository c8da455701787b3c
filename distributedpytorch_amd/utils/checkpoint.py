"""Checkpoint save / load with the reference's on-disk contract (/root/reference/utils.py:112-140):
torch.save of {'model_name', 'model_state_dict', 'optimizer_state_dict', 'epoch', 'loss'}; model keys are torchvision
names with the DDP 'module.' prefix, fp32 OIHW tensors (+ int64 num_batches_tracked); the optimizer dict is in
torch.optim layout.  Internally everything lives in flat arenas in the kernels' layout; this module materialises the
per-tensor views.  Loading accepts keys with or without 'module.' (Q16) and always maps to CPU first (Q18)."""
from __future__ import annotations

import logging
from collections import OrderedDict

import torch


def saveCheckpoint(path, modelName, net, optimizer, epoch, bestValidLoss):
    sd = OrderedDict((k, v.detach().cpu()) for k, v in net.state_dict(prefix="module.").items())
    osd = optimizer.state_dict() if optimizer is not None else {"state": {}, "param_groups": []}
    for st in osd["state"].values():
        for k, v in list(st.items()):
            if torch.is_tensor(v):
                st[k] = v.detach().cpu()
    torch.save({
        "model_name": modelName,
        "model_state_dict": sd,
        "optimizer_state_dict": osd,
        "epoch": epoch,
        "loss": bestValidLoss,
    }, path)
    logging.info(f"epoch:{epoch:04d}: model saved to {path}")


def _load(path):
    return torch.load(path, map_location="cpu", weights_only=True)


def loadCheckpoint(gpu, path, net, optimizer):
    """Returns (start_epoch, best_valid_loss) like utils.loadCheckpoint."""
    ckpt = _load(path)
    net.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None and ckpt.get("optimizer_state_dict", {}).get("param_groups"):
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    epoch = ckpt["epoch"] + 1
    best = ckpt["loss"]
    logging.info(f"epoch:{epoch:04d}: model loaded from {path}")
    return epoch, best


def getCheckpointModelName(path):
    return _load(path)["model_name"]

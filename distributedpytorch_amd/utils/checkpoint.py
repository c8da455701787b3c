"""Checkpoint save / load with the reference's on-disk contract (/root/reference/utils.py:112-140):
torch.save of {'model_name', 'model_state_dict', 'optimizer_state_dict', 'epoch', 'loss'}; model keys are torchvision
names with the DDP 'module.' prefix, fp32 OIHW tensors (+ int64 num_batches_tracked); the optimizer dict is in
torch.optim layout.  Internally everything lives in flat arenas in the kernels' layout; this module materialises the
per-tensor views.  Loading accepts keys with or without 'module.' (Q16) and always maps to CPU first (Q18).

Two pieces of carried state that torch has no slot for travel as EXTRA entries of optimizer_state_dict['param_groups'][0]
(torch.optim.Optimizer.load_state_dict keeps unknown group keys, so the reference's loader - utils.py:123-136 - still
accepts the file, and the five top-level keys stay exactly the reference's):
  dpa_rng_state       [seed, step] of the dropout counter RNG - without it a resumed AlexNet / VGG / SqueezeNet /
                      Inception run would replay epoch 0's dropout masks
  dpa_bn_stat_shift   every BatchNorm's statistics shift (the previous step's batch mean), one flat fp32 tensor -
                      without it the first resumed step computes its variance about 0 and is not bit-identical
A resumed run then equals the uninterrupted one bit for bit (tests/test_gpu_train.py)."""
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
    if optimizer is not None and osd["param_groups"]:
        g = osd["param_groups"][0]
        g["dpa_rng_state"] = [int(v) for v in net.rng.tolist()]
        aux = getattr(net.store, "aux_tensors", [])
        if aux:
            g["dpa_bn_stat_shift"] = torch.cat([a[9].detach().reshape(-1) for a in aux]).cpu()
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
        g = ckpt["optimizer_state_dict"]["param_groups"][0]
        if "dpa_rng_state" in g:
            net.rng.copy_(torch.tensor([int(v) for v in g["dpa_rng_state"]], dtype=torch.int64))
        aux = getattr(net.store, "aux_tensors", [])
        shift = g.get("dpa_bn_stat_shift")
        if shift is not None and aux and shift.numel() == sum(a[9].numel() for a in aux):
            off = 0
            for a in aux:
                n = a[9].numel()
                a[9].copy_(shift[off: off + n].to(a.device))
                off += n
    epoch = ckpt["epoch"] + 1
    best = ckpt["loss"]
    logging.info(f"epoch:{epoch:04d}: model loaded from {path}")
    return epoch, best


def getCheckpointModelName(path):
    return _load(path)["model_name"]

"""Service functions of the reference's utils.py (/root/reference/utils.py:164-202): logging, device report, timing,
seeding - with its defects fixed (Q12: the result directory is created and only the writer rank truncates the log)."""
from __future__ import annotations

import logging
import os
import random
import sys

import numpy as np
import torch


def initializeLogging(rsl_path="./rsl", log_file="test.log", writer=True, debug=False):
    """INFO, '%(message)s', file + stdout like utils.initializeLogging; non-writer ranks log to stdout only."""
    handlers = [logging.StreamHandler(sys.stdout)]
    if writer:
        os.makedirs(rsl_path, exist_ok=True)
        handlers.insert(0, logging.FileHandler(os.path.join(rsl_path, log_file), mode="w"))
    root = logging.getLogger()
    for h in list(root.handlers):
        root.removeHandler(h)
    logging.basicConfig(level=logging.DEBUG if debug else logging.INFO, format="%(message)s", handlers=handlers)


def checkCuda():
    """utils.checkCuda: print versions + GPU count; 1 when a GPU (MI355X via HIP) is usable."""
    print(f"PyTorch {torch.__version__}")
    print(f"distributedpytorch_amd (MI355X-native engine; torchvision not required)")
    if torch.cuda.is_available():
        ids = list(range(torch.cuda.device_count()))
        print(f"HIP {torch.version.hip}")
        print(f"{len(ids)} GPU detected: {ids}")
        return 1
    print("No HIP device, no GPU")
    return 0


def getDuration(start_time, end_time):
    elapsed = end_time - start_time
    mins = int(elapsed / 60)
    return mins, int(elapsed - mins * 60)


def setRandomSeed(seed):
    """utils.setRandomSeed: every RNG the run touches.  Kernel-level determinism needs no flags here: the HIP kernels
    use fixed-order reductions (no float atomics), so a seeded run is bitwise reproducible for a fixed world size."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def printNetworkInfo(net):
    for spec in net.store.params:
        logging.info(f"{net.store.device} {spec.name} {spec.logical_shape}")

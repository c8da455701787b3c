"""MNIST data pipeline with the reference's structure (/root/reference/dataloader.py:47-180) and an MI355X-first
execution model: the uint8 dataset (47 MB) is resident in HBM and every batch is produced by ONE fused HIP kernel
from the sampler's indices (csrc/hip/augment.hip), so no DataLoader workers, no pinned staging and no 38.5 MB H2D
copy per step (K19/K20 of SURVEY.md).  Preserved semantics: normalisation statistics of the 60 000 training images,
54 000/6 000 seeded random split (same partition on every rank), train transform = rotate +-5 deg ->
random-resized-crop -> resize -> 3-channel repeat -> normalise, eval transform = resize -> ... , debug subset of 200
training samples, one DistributedSampler per phase (shuffled, padded), per-process batch size.
"""
from __future__ import annotations

import ctypes
import logging
import math
import os

import numpy as np
import torch

from .. import ops
from ..nn.engine import input_pad, pad_to
from ..ops import native
from .sampler import DistributedSampler

VALID_RATIO = 0.9
FILES = {"train": ("train-images-idx3-ubyte", "train-labels-idx1-ubyte"),
         "test": ("t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")}


# ------------------------------------------------------------------------------------------------ IDX files
def read_idx(path) -> np.ndarray:
    lib = native.host_lib()
    dims = (ctypes.c_int * 4)()
    nd = lib.dpa_idx_header(path.encode(), dims)
    if nd < 0:
        raise FileNotFoundError(f"cannot read IDX file {path} (rc={nd})")
    shape = tuple(dims[i] for i in range(nd))
    out = np.empty(shape, dtype=np.uint8)
    lib.dpa_idx_read.restype = ctypes.c_longlong
    n = lib.dpa_idx_read(path.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(out.size))
    if n != out.size:
        raise IOError(f"short read on {path}: rc={n}")
    return out


def write_idx(path, arr: np.ndarray):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    dims = (ctypes.c_int * arr.ndim)(*arr.shape)
    rc = native.host_lib().dpa_idx_write(path.encode(), arr.ctypes.data_as(ctypes.c_void_p), dims, arr.ndim)
    if rc:
        raise IOError(f"cannot write {path} (rc={rc})")


def raw_dir(root):
    """Where torchvision.datasets.MNIST keeps the REAL files (/root/reference/dataloader.py:92-93)."""
    return os.path.join(root, "MNIST", "raw")


def synthetic_dir(root):
    """Where the synthetic stand-in lives: never under MNIST/raw, so it cannot be taken for the real dataset by a later
    run or by torchvision."""
    return os.path.join(root, "MNIST-synthetic", "raw")


def _have(d):
    return all(os.path.exists(os.path.join(d, f)) for pair in FILES.values() for f in pair)


def have_mnist(root):
    return _have(raw_dir(root))


def have_synthetic(root):
    """Complete synthetic set: the four files AND the marker the writer creates last."""
    return _have(synthetic_dir(root)) and os.path.exists(os.path.join(synthetic_dir(root), ".complete"))


def _render_strokes(ctrl, width, T=20):
    """ctrl: [B, P, 2] polyline control points (y, x) in pixels; width: [B] stroke sigma.  Returns [B, 28, 28] in 0..1."""
    B, P, _ = ctrl.shape
    t = np.linspace(0, 1, T, dtype=np.float32)[None, None, :, None]
    pts = (ctrl[:, :-1, None, :] * (1 - t) + ctrl[:, 1:, None, :] * t).reshape(B, -1, 2)        # [B, (P-1)*T, 2]
    yy, xx = np.mgrid[0:28, 0:28].astype(np.float32)
    d2 = (yy[None, None] - pts[:, :, 0, None, None]) ** 2 + (xx[None, None] - pts[:, :, 1, None, None]) ** 2
    img = np.exp(-d2 / (2 * width[:, None, None, None] ** 2)).max(1)       # max, not sum: uniform stroke intensity
    return img


def generate_synthetic_mnist(root, n_train=60000, n_test=10000, seed=1234, hard=False, jitter=1.0):
    """Deterministic MNIST-shaped stand-in (there is no network to fetch the real files).

    hard=False: ten smooth stroke prototypes, one per class, with per-sample shift, scale jitter and noise - separable
    to 100 % by every implementation after one epoch (good for plumbing, useless for parity).
    hard=True: every SAMPLE is drawn anew - the class's control points are jittered (sigma `jitter` px, comparable to the
    distance between classes, whose prototypes share a common skeleton and differ in two control points), stroke
    width, affine shear / scale / shift, additive clutter strokes and pixel noise vary per sample.  Classes overlap, so
    test accuracy saturates below 100 % (a torch-eager ResNet-18 lands in the high 90s) and differences between
    implementations show.  Written under <root>/MNIST-synthetic/raw (NOT MNIST/raw: see synthetic_dir); each file is
    written under a temporary name and renamed; the `.complete` marker comes last, and readers gate on it."""
    out_dir = synthetic_dir(root)
    os.makedirs(out_dir, exist_ok=True)
    marker = os.path.join(out_dir, ".complete")
    if os.path.exists(marker):
        os.remove(marker)
    rng = np.random.RandomState(seed)
    if hard:
        skeleton = rng.uniform(7, 21, size=(5, 2)).astype(np.float32)
        protos_ctrl = np.repeat(skeleton[None], 10, 0)
        for c in range(10):                  # each class moves two of the five control points of the shared skeleton
            which = rng.choice(5, size=2, replace=False)
            protos_ctrl[c, which] += rng.uniform(-6.0, 6.0, size=(2, 2)).astype(np.float32)
        protos_ctrl = np.clip(protos_ctrl, 4, 24)

        def make(n, rs):
            labels = rs.randint(0, 10, size=n).astype(np.uint8)
            imgs = np.empty((n, 28, 28), np.uint8)
            for s0 in range(0, n, 256):
                lab = labels[s0: s0 + 256]
                b = lab.shape[0]
                ctrl = protos_ctrl[lab] + rs.normal(0, jitter, size=(b, 5, 2)).astype(np.float32)
                # per-sample affine about the image centre: scale, shear, shift
                sc = rs.uniform(0.8, 1.15, size=(b, 1, 1)).astype(np.float32)
                sh = rs.uniform(-0.25, 0.25, size=(b, 1)).astype(np.float32)
                ctr = ctrl - 14.0
                ctr = ctr * sc
                ctr[:, :, 1] += sh * ctr[:, :, 0]
                ctrl = ctr + 14.0 + rs.uniform(-2.5, 2.5, size=(b, 1, 2)).astype(np.float32)
                img = _render_strokes(ctrl, rs.uniform(0.9, 1.7, size=b).astype(np.float32))
                # clutter: one short random stroke of lower intensity
                cl = rs.uniform(3, 25, size=(b, 2, 2)).astype(np.float32)
                cl[:, 1] = cl[:, 0] + rs.uniform(-6, 6, size=(b, 2)).astype(np.float32)
                img = np.maximum(img, 0.55 * _render_strokes(cl, np.full(b, 1.0, np.float32), T=8))
                img = img * rs.uniform(0.6, 1.0, size=(b, 1, 1)).astype(np.float32)
                img = img + rs.normal(0, 0.12, size=img.shape).astype(np.float32)
                imgs[s0: s0 + b] = (np.clip(img, 0, 1) * 255).astype(np.uint8)
            return imgs, labels
    else:
        protos = []
        yy, xx = np.mgrid[0:28, 0:28].astype(np.float32)
        for c in range(10):
            img = np.zeros((28, 28), np.float32)
            pts = rng.uniform(6, 22, size=(4 + c % 3, 2))
            for a, b in zip(pts[:-1], pts[1:]):
                for t in np.linspace(0, 1, 24):
                    cy, cx = a + t * (b - a)
                    img += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.3 ** 2))
            protos.append(np.clip(img / img.max(), 0, 1))
        protos = np.stack(protos)

        def make(n, rs):
            labels = rs.randint(0, 10, size=n).astype(np.uint8)
            imgs = np.empty((n, 28, 28), np.uint8)
            shifts = rs.randint(-2, 3, size=(n, 2))
            gains = rs.uniform(0.7, 1.0, size=n).astype(np.float32)
            for i in range(n):
                p = np.roll(protos[labels[i]], tuple(shifts[i]), axis=(0, 1)) * gains[i]
                p = p + rs.normal(0, 0.08, size=(28, 28)).astype(np.float32) * (p > 0.02)
                imgs[i] = (np.clip(p, 0, 1) * 255).astype(np.uint8)
            return imgs, labels

    tr_x, tr_y = make(n_train, np.random.RandomState(seed + 1))
    te_x, te_y = make(n_test, np.random.RandomState(seed + 2))
    for name, arr in ((FILES["train"][0], tr_x), (FILES["train"][1], tr_y), (FILES["test"][0], te_x), (FILES["test"][1], te_y)):
        tmp = os.path.join(out_dir, f".{name}.tmp{os.getpid()}")
        write_idx(tmp, arr)
        os.replace(tmp, os.path.join(out_dir, name))
    open(marker, "w").close()
    logging.warning(f"SYNTHETIC MNIST written to {out_dir} ({n_train} train / {n_test} test): NOT the real dataset")


# ------------------------------------------------------------------------------------------------ dataset
class Subset:
    """Index view of a split, like torch.utils.data.Subset; carries the class weights the reference assumed (Q10)."""

    def __init__(self, split, indices: torch.Tensor, labels: torch.Tensor, nb_classes):
        self.split, self.indices = split, indices
        counts = torch.bincount(labels[indices], minlength=nb_classes).float().clamp_min(1)
        self.classWeights = len(indices) / (nb_classes * counts)

    def __len__(self):
        return len(self.indices)


class BatchIterator:
    """One phase's loader: sampler indices -> device index list -> fused augment kernel per batch."""

    def __init__(self, ds, phase):
        self.ds, self.phase = ds, phase

    def __len__(self):
        return math.ceil(len(self.ds.sampler[self.phase]) / self.ds.batchSize)

    def __iter__(self):
        ds, phase = self.ds, self.phase
        sub = ds.data[phase]
        local = ds.sampler[phase].indices()
        gidx = sub.indices[local].to(torch.int32)
        gidx_dev = gidx.to(ds.device, non_blocking=True)
        split = sub.split
        images, labels = ds.images[split], ds.labels[split]
        train = phase == "train"
        epoch = ds.sampler[phase].epoch
        b = ds.batchSize
        for s in range(0, gidx.numel(), b):
            idx = gidx_dev[s: s + b]
            n = idx.numel()
            x = torch.empty((n, ds.dataDim, ds.dataDim, ds.cpad), device=ds.device, dtype=ds.dtype)
            y = torch.empty(n, dtype=torch.int64, device=ds.device)
            ops.augment_mnist(images, labels, idx, x, y, train, ds.mean, ds.std, ds.seed, epoch)
            yield x, y


class MNIST:
    def __init__(self, dataPath, batchSize, dataDim, worldSize, rank, gpu, device=None, dtype=torch.float32, debug=False,
                 seed=1234, synthetic_ok=False):
        self.dataPath, self.batchSize, self.dataDim = dataPath, batchSize, dataDim
        self.worldSize, self.rank, self.gpu = worldSize, rank, gpu
        self.device = device if device is not None else torch.device("cuda", gpu) if gpu >= 0 else torch.device("cpu")
        self.dtype, self.debug, self.seed = dtype, debug, seed
        self.nbClasses = 10
        self.cpad = input_pad(3, dtype)
        self.data = {"train": None, "valid": None, "test": None}
        self.sampler = {"train": None, "valid": None, "test": None}
        self.iterator = {"train": None, "valid": None, "test": None}
        self.mean = self.std = 0
        self.synthetic_ok = synthetic_ok
        self.loadDataset()
        self.initIterator()
        if self.gpu <= 0:
            logging.info(f"Number of training examples: {len(self.data['train'])}")
            logging.info(f"Number of validation examples: {len(self.data['valid'])}")
            logging.info(f"Number of testing examples: {len(self.data['test'])}")

    def loadDataset(self):
        if have_mnist(self.dataPath):
            raw = raw_dir(self.dataPath)
        elif not self.synthetic_ok:
            # the reference's behaviour (download=False): a mistyped -d must not silently train on something else
            raise FileNotFoundError(f"MNIST IDX files not found under {raw_dir(self.dataPath)} "
                                    f"(DPA_SYNTHETIC_DATA=1 trains on a generated stand-in instead)")
        else:
            # opted in: one writer per node (local rank 0), EVERY rank - the writer included - gates on the marker
            if not have_synthetic(self.dataPath) and self.gpu <= 0:
                generate_synthetic_mnist(self.dataPath, seed=self.seed)
            import time
            t0 = time.time()
            while not have_synthetic(self.dataPath):
                if time.time() - t0 > 600:
                    raise TimeoutError("timed out waiting for the synthetic MNIST files")
                time.sleep(0.2)
            raw = synthetic_dir(self.dataPath)
            if self.gpu <= 0:
                logging.warning(f"training on SYNTHETIC digits from {raw}: accuracies are not MNIST accuracies")
        tr_x = torch.from_numpy(read_idx(os.path.join(raw, FILES["train"][0])))
        tr_y = torch.from_numpy(read_idx(os.path.join(raw, FILES["train"][1]))).long()
        te_x = torch.from_numpy(read_idx(os.path.join(raw, FILES["test"][0])))
        te_y = torch.from_numpy(read_idx(os.path.join(raw, FILES["test"][1]))).long()
        # normalisation statistics over all training images (dataloader.py:94-95)
        f = tr_x.float()
        self.mean = float(f.mean() / 255)
        self.std = float(f.std() / 255)
        del f
        self.images = {"train": tr_x.to(self.device), "test": te_x.to(self.device)}
        self.labels = {"train": tr_y.to(self.device), "test": te_y.to(self.device)}
        self._labels_cpu = {"train": tr_y, "test": te_y}
        # random_split([n*0.9, rest]) with the global (seeded) torch RNG: identical partition on every rank
        n = tr_x.shape[0]
        n_train = int(n * VALID_RATIO)
        perm = torch.randperm(n)
        self.data["train"] = Subset("train", perm[:n_train], tr_y, self.nbClasses)
        self.data["valid"] = Subset("train", perm[n_train:], tr_y, self.nbClasses)
        self.data["test"] = Subset("test", torch.arange(te_x.shape[0]), te_y, self.nbClasses)

    def initIterator(self):
        if self.debug:   # quick test using only 200 training samples (dataloader.py:139-144)
            sub = self.data["train"]
            self.data["train"] = Subset(sub.split, sub.indices[:200], self._labels_cpu[sub.split], self.nbClasses)
        distributed = self.gpu >= 0 or self.worldSize > 1
        for phase in ("train", "valid", "test"):
            n = len(self.data[phase])
            if distributed:
                self.sampler[phase] = DistributedSampler(n, self.worldSize, self.rank, shuffle=True)
            else:   # reference CPU path: plain sequential loaders (dataloader.py:171-180)
                self.sampler[phase] = DistributedSampler(n, 1, 0, shuffle=False)
            self.iterator[phase] = BatchIterator(self, phase)

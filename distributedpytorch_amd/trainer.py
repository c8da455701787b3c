"""Training / evaluation drivers: the reference's classif.py (train :75-192, test :197-243, processData :28-71) on the
MI355X-native engine.  Same outward behaviour (log lines, checkpoint files, per-process batch size, rank-local
metrics) with the reference's defects fixed as decided in SURVEY.md §2.8 (Q3 resume works, Q4 -d honoured,
Q5 --debug propagated, Q7 set_epoch before the epoch, Q8 stale checkpoint removed, Q9 updated best loss stored,
Q11 global rank 0 writes, Q13 no per-iteration host sync, Q14 progress mean, Q17 destroy_process_group,
Q21 inception eval returns one output).
"""
from __future__ import annotations

import logging
import os
import sys
import time

import torch

from . import ops
from .data.mnist import MNIST
from .models.factory import get_model, get_model_input_size
from .nn.loss import Criterion
from .nn.optim import ArenaOptimizer, StepLR
from .parallel import comm
from .parallel.ddp import DistributedModel
from .utils import checkpoint as ckpt
from .utils import misc


def _cfg():
    import config
    return config


_COMPUTE_STREAMS = {}


def _device_for(gpu):
    if gpu >= 0 and torch.cuda.is_available():
        torch.cuda.set_device(gpu)
        dev = torch.device("cuda", gpu)
        if gpu not in _COMPUTE_STREAMS and os.environ.get("DPA_COMPUTE_STREAM", "own") == "own":
            # explicit non-blocking compute stream (the legacy null stream synchronises implicitly with every
            # blocking stream a library creates)
            _COMPUTE_STREAMS[gpu] = torch.cuda.Stream(device=dev)
        if gpu in _COMPUTE_STREAMS:
            torch.cuda.set_stream(_COMPUTE_STREAMS[gpu])
        return dev
    return torch.device("cpu")


def _dtype(cfg, device):
    if device.type == "cpu":
        return torch.float32
    return torch.bfloat16 if getattr(cfg, "COMPUTE_DTYPE", "fp32") == "bf16" else torch.float32


def _maybe_fault(cfg, rank, step):
    spec = getattr(cfg, "FAULT_INJECT", "")
    if spec:
        r, s = spec.split(":")
        if int(r) == rank and int(s) == step:
            logging.error(f"fault injection: rank {rank} exits at step {step}")
            sys.stdout.flush()
            os._exit(17)


def processData(gpu, phase, epoch, dmodel: DistributedModel, modelName, dataset, rank=0, cfg=None, step0=0):
    """One pass over `phase`.  Returns (mean loss, mean accuracy) = mean of per-batch means, rank-local, exactly what
    the reference reports (classif.py:71); the sums live on the device and are read at the 10 % log points only."""
    train = phase == "train"
    it = dataset.iterator[phase]
    nbIters = len(it)
    lastLog = 0
    dmodel.reset_epoch_metrics()
    graph_ok = train and cfg is not None and getattr(cfg, "USE_HIPGRAPH", False) and dmodel.can_capture()
    steplog = dmodel.steplog if train else None
    batches = iter(it)
    i = -1
    while True:
        if steplog is not None:
            steplog.begin(step0 + i + 1, epoch)
        try:
            inputs, labels = next(batches)
        except StopIteration:
            break
        i += 1
        if train:
            if steplog is not None:
                steplog.mark("data")
            if cfg is not None:
                _maybe_fault(cfg, rank, step0 + i)
            if graph_ok and inputs.shape[0] == dataset.batchSize:
                if dmodel._graph is None:
                    dmodel.capture_train_step(inputs, labels)
                dmodel.replay_train_step(inputs, labels)
                if steplog is not None:
                    steplog.mark("graph")           # the phases live inside the replayed hipGraph
            else:
                dmodel.train_step(inputs, labels, mark=steplog.mark if steplog is not None else None)
            if steplog is not None:
                steplog.end(dmodel.metrics, inputs.shape[0])
            if _NATIVE and i % 50 == 49 and _NATIVE[-1].async_error() != 0:      # ncclCommGetAsyncError
                _NATIVE[-1].abort()
                raise RuntimeError(f"communicator reported an asynchronous error on rank {rank}; aborted")
        else:
            dmodel.eval_step(inputs, labels)
        if gpu <= 0 and train:
            n = i / nbIters * 100
            if sys.stdout.isatty():
                print(f"\r{epoch:03d} {n:.0f}%", end="\r")
            if i and n // 10 > lastLog:
                lastLog = n // 10
                loss, _, cnt = dmodel.epoch_metrics()
                logging.info(f"\repoch:{epoch:03d} nb batches:{i + 1:04d} mean train loss:{loss:.5f}")
                if steplog is not None:
                    steplog.flush()
    loss, acc, cnt = dmodel.epoch_metrics()
    if steplog is not None:
        steplog._cur = None
        steplog.flush()
    if cfg is not None and getattr(cfg, "GLOBAL_METRICS", False) and comm.is_distributed():
        loss, acc = (v / comm.world_size() for v in comm.all_reduce_scalars([loss, acc], "sum", dmodel.device))
    return loss, acc


def _build(args, gpu, rank, modelName, cfg, with_optimizer):
    device = _device_for(gpu)
    dtype = _dtype(cfg, device)
    if device.type == "cuda" and dtype == torch.float32:
        from .ops.native import set_fp32_mode
        set_fp32_mode(getattr(cfg, "FP32_MODE", "x6"))
        if rank == 0:
            logging.debug(f"fp32 convolution arithmetic: {getattr(cfg, 'FP32_MODE', 'x6')}")
    dataset = MNIST(args.dataPath, args.batchSize, get_model_input_size(modelName), worldSize=args.worldSize, rank=rank,
                    gpu=gpu, device=device, dtype=dtype, debug=bool(args.debug), seed=cfg.SEED,
                    synthetic_ok=getattr(cfg, "SYNTHETIC_DATA", False))
    net, _ = get_model(modelName, dataset.nbClasses, cfg.FEATURE_EXTRACT, cfg.USE_PRETRAINED, device=device, dtype=dtype)
    net.seed_rng(cfg.SEED)
    loss_kind = cfg.LOSS if with_optimizer else "cross_entropy"
    if loss_kind == "cross_entropy":
        criterion = Criterion("cross_entropy")
    elif loss_kind in ("weighted_cross_entropy", "focal_loss"):
        criterion = Criterion(loss_kind, dataset.data["train"].classWeights).to(device)
    else:
        logging.error("Invalid loss, exiting...")
        raise SystemExit(1)
    optimizer = None
    scheduler = None
    if with_optimizer:
        if cfg.OPTIMIZER == "adam":
            optimizer = ArenaOptimizer(net.store, "adam", lr=0.001)
        elif cfg.OPTIMIZER == "SGD":
            optimizer = ArenaOptimizer(net.store, "SGD", lr=0.001, momentum=0.9)
        else:
            logging.error("Invalid optimizer, exiting...")
            raise SystemExit(1)
    else:
        optimizer = ArenaOptimizer(net.store, "adam", lr=0.0)     # never stepped; gives the reducer its interface
    dmodel = DistributedModel(net, criterion, optimizer, first_bucket_mb=getattr(cfg, "FIRST_BUCKET_MB", 1.0),
                              bucket_cap_mb=getattr(cfg, "BUCKET_CAP_MB", 25.0),
                              broadcast_buffers=getattr(cfg, "BROADCAST_BUFFERS", True),
                              sync_comm=getattr(cfg, "SYNC_COMM", False), native_comm=_NATIVE[-1] if _NATIVE else None,
                              activation_arena=getattr(cfg, "ACTIVATION_ARENA", True))
    from .utils.steplog import StepLog
    dmodel.steplog = StepLog(os.path.join(cfg.RSL_PATH, "steps.jsonl"), device, args.worldSize, args.batchSize,
                             enabled=with_optimizer and rank == 0 and getattr(cfg, "STEP_METRICS", True))
    return device, dataset, net, dmodel, optimizer, scheduler


def _comm_backend(cfg):
    b = getattr(cfg, "COMM_BACKEND", "native")
    return "native" if b in ("native", "rccl_native", "tcp_native") else b


def _init_dist(args, gpu, rank, cfg):
    """Bring-up of the communication layer (X1 of SURVEY.md §2.6; /root/reference/classif.py:86-87).

    COMM_BACKEND='native' (default): the C++ TCP store on MASTER_PORT+2 is the rendezvous (arrival check that NAMES
    missing ranks - the reference just blocks, README.md:47-49 -, ncclUniqueId / mesh addresses) and ONE native
    communicator per rank carries the data: RCCL over xGMI on GPUs, the TCP mesh on CPUs.  No process group exists.
    COMM_BACKEND='torch': torch.distributed's env:// bring-up of RCCL (GPU) or gloo (CPU), with the store's arrival check
    in front."""
    device = _device_for(gpu)
    if not (args.worldSize > 1 or gpu >= 0):
        return
    addr = os.environ.get("MASTER_ADDR", cfg.MASTER_ADDR)
    port = int(os.environ.get("MASTER_PORT", cfg.MASTER_PORT))
    if _comm_backend(cfg) == "native":
        force = os.environ.get("DPA_FORCE_COLLECTIVES", "0") == "1"
        nc = comm.init_native(rank, args.worldSize, device, addr, port, timeout_s=getattr(cfg, "COLLECTIVE_TIMEOUT_S", 600.0),
                              rendezvous_timeout_s=getattr(cfg, "RENDEZVOUS_TIMEOUT_S", 300.0) or 300.0,
                              create_for_single_rank=force)
        if nc is not None:
            _NATIVE.append(nc)
        return
    if args.worldSize > 1 and getattr(cfg, "RENDEZVOUS_TIMEOUT_S", 0) > 0:
        from .parallel.store import TCPStore
        store = TCPStore(addr, port + 1, is_master=(rank == 0), timeout_s=cfg.RENDEZVOUS_TIMEOUT_S)
        try:
            store.announce_and_wait(rank, args.worldSize, tag="join")
            store.barrier(rank, args.worldSize, "joined")     # nobody tears the server down while others still poll
        finally:
            if rank != 0:
                store.close()
        _STORES.append(store)
    comm.init_process_group(rank, args.worldSize, device, timeout_s=getattr(cfg, "COLLECTIVE_TIMEOUT_S", 600.0))


def make_native_comm(rank, world, device, addr, port, timeout_s=300.0):
    """The native communicator for callers outside the trainer (bench.py, tests): store on `port` + the offset."""
    return comm.init_native(rank, world, device, addr, port, timeout_s=timeout_s, create_for_single_rank=True)


_STORES = []
_NATIVE = []


def train(gpu, args):
    """Per-process entry (spawned once per GPU; gpu = -1 means CPU).  Mirrors classif.train."""
    cfg = _cfg()
    rank = args.firstLocalRank + max(gpu, 0) if gpu >= 0 else getattr(args, "cpuRank", 0)
    misc.initializeLogging(cfg.RSL_PATH, cfg.LOG_FILE, writer=(rank == 0), debug=bool(args.debug))
    if gpu >= 0 or args.worldSize > 1:
        logging.info(f"local rank: {gpu}, overall rank: {rank}, world size: {args.worldSize}")
        logging.info(f"batch size: {args.batchSize}, number of workers: {cfg.NUM_WORKERS}")
    _init_dist(args, gpu, rank, cfg)
    try:
        misc.setRandomSeed(cfg.SEED)
        modelName = ckpt.getCheckpointModelName(args.checkpointFile) if args.checkpointFile else cfg.MODEL_NAME
        device, dataset, net, dmodel, optimizer, scheduler = _build(args, gpu, rank, modelName, cfg, True)
        if args.checkpointFile:
            startEpoch, bestValidLoss = ckpt.loadCheckpoint(gpu, args.checkpointFile, net, optimizer)
        else:
            startEpoch, bestValidLoss = 0, float("inf")
        if cfg.OPTIMIZER == "SGD":
            scheduler = StepLR(optimizer, step_size=1, gamma=0.1, last_epoch=startEpoch)
        compat = getattr(cfg, "COMPAT_EPOCH_ORDER", False)
        startTime = time.monotonic()
        step0 = 0
        for epoch in range(startEpoch, args.nbEpochs):
            if gpu <= 0:
                print(f"====================== epoch{epoch + 1:4d} ======================")
            startEpochTime = time.monotonic()
            if not compat:
                dataset.sampler["train"].set_epoch(epoch)       # documented contract: before the epoch (Q7)
            trainLoss, trainAcc = processData(gpu, "train", epoch, dmodel, modelName, dataset, rank, cfg, step0)
            step0 += len(dataset.iterator["train"])
            validLoss, validAcc = processData(gpu, "valid", epoch, dmodel, modelName, dataset, rank, cfg)
            if compat:
                dataset.sampler["train"].set_epoch(epoch)       # the reference's order (classif.py:164-165)
            if scheduler is not None:
                scheduler.step()
            if device.type == "cuda":
                torch.cuda.synchronize()
            endTime = time.monotonic()
            epochMins, epochSecs = misc.getDuration(startEpochTime, endTime)
            mins, secs = misc.getDuration(startTime, endTime)
            nimg = len(dataset.sampler["train"]) * args.worldSize
            ips = nimg / max(endTime - startEpochTime, 1e-9)
            every = getattr(cfg, "CHECK_SYNC_EVERY", 0)
            if every and (epoch + 1) % every == 0:
                dev = dmodel.reducer.check_in_sync()
                if dev != 0.0:
                    raise RuntimeError(f"ranks diverged: parameter checksum spread {dev}")
            if gpu <= 0:
                logging.info(f"{'*' if validLoss < bestValidLoss else ' '} Epoch: {epoch + 1:03}  | Duration: {epochMins:03d}m "
                             f"{epochSecs:02d}s  | Overall duration: {mins / 60:.2f}h")
                logging.info(f"  Train       | Loss: {trainLoss:.5f}       | Acc: {trainAcc * 100:.2f}%")
                logging.info(f"  Validation  | Loss: {validLoss:.5f}       | Acc: {validAcc * 100:.2f}%")
                logging.info(f"  Throughput  | {ips:.0f} images/s (epoch incl. validation, {args.worldSize} ranks)")
            improved = validLoss < bestValidLoss
            if improved:
                bestValidLoss = validLoss
            if rank == 0:                                        # one writer for the whole job (Q11)
                os.makedirs(cfg.RSL_PATH, exist_ok=True)
                previousCPFile = f"{cfg.RSL_PATH}/checkpoint-mnist-{modelName}-{epoch - 1:03d}.pt.tar"
                if os.path.exists(previousCPFile):
                    os.remove(previousCPFile)
                ckpt.saveCheckpoint(f"{cfg.RSL_PATH}/checkpoint-mnist-{modelName}-{epoch:03d}.pt.tar", modelName, net,
                                    optimizer, epoch, bestValidLoss)
                if improved:
                    ckpt.saveCheckpoint(f"{cfg.RSL_PATH}/bestmodel-mnist-{modelName}.pt.tar", modelName, net, optimizer,
                                        epoch, bestValidLoss)
                _write_metrics(cfg, dict(epoch=epoch, train_loss=trainLoss, train_acc=trainAcc, valid_loss=validLoss,
                                         valid_acc=validAcc, images_per_s=ips, seconds=endTime - startEpochTime,
                                         world_size=args.worldSize))
    finally:
        _close_stores()


def _close_stores():
    _NATIVE.clear()
    # native communicator + its store (with a farewell barrier unless an exception is on its way out), or the process group
    comm.shutdown(clean=sys.exc_info()[0] is None)
    while _STORES:
        try:
            _STORES.pop().close()
        except Exception:       # noqa: BLE001 - shutdown path
            pass


def _write_metrics(cfg, rec):
    import json
    with open(os.path.join(cfg.RSL_PATH, "metrics.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")


def test(gpu, args):
    """Mirrors classif.test: model name from the checkpoint, one pass over the test split, rank-local accuracy."""
    cfg = _cfg()
    rank = args.firstLocalRank + max(gpu, 0) if gpu >= 0 else getattr(args, "cpuRank", 0)
    misc.initializeLogging(cfg.RSL_PATH, cfg.LOG_FILE, writer=(rank == 0), debug=bool(args.debug))
    if gpu >= 0 or args.worldSize > 1:
        logging.info(f"local rank: {gpu}, overall rank: {rank}, world size: {args.worldSize}")
    _init_dist(args, gpu, rank, cfg)
    try:
        misc.setRandomSeed(cfg.SEED)
        modelName = ckpt.getCheckpointModelName(args.checkpointFile)
        device, dataset, net, dmodel, _, _ = _build(args, gpu, rank, modelName, cfg, False)
        ckpt.loadCheckpoint(gpu, args.checkpointFile, net, None)
        startTime = time.monotonic()
        loss, acc = processData(gpu, "test", 0, dmodel, modelName, dataset, rank, cfg)
        if device.type == "cuda":
            torch.cuda.synchronize()
        mins, secs = misc.getDuration(startTime, time.monotonic())
        if gpu <= 0:
            logging.info(f"Time: {mins}m {secs}s, Acc: {acc * 100:.2f}%")
        return acc
    finally:
        _close_stores()

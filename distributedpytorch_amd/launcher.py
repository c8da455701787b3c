"""CLI + launcher (C1-C4 of SURVEY.md §2.1): argparse identical to the reference (/root/reference/main.py:20-58),
local interface discovery with the SIOCGIFCONF ioctl (:60-90), node -> rank topology from config.DDTNodes (:92-110),
then one spawned process per GPU with MASTER_ADDR / MASTER_PORT / HIP_VISIBLE_DEVICES exported (:127-135).
Fixed relative to the reference: the CPU path works (Q1/Q2), an unmatched host gives a clear error instead of a
TypeError, --debug and -d reach the workers (Q4/Q5)."""
from __future__ import annotations

import argparse
import array
import fcntl
import os
import socket
import struct
import sys


def _cfg():
    import config
    return config


def getArgs(argv=None):
    cfg = _cfg()
    parserCommon = argparse.ArgumentParser(add_help=False)
    parserCommon.add_argument('--debug', action='store_true', dest='debug', default=cfg.DEBUG, help='debug mode')
    parserCommon.add_argument('-d', '--data_path', metavar='data_path', action='store', type=str, dest='dataPath',
                              default=None, required=True, help='data path')
    parserCommon.add_argument('-b', '--batchSize', metavar='N', type=int, dest='batchSize', default=cfg.BATCH_SIZE,
                              help=f'batch size (default: {cfg.BATCH_SIZE})')
    parser = argparse.ArgumentParser(description='Et aïe!')
    subparsers = parser.add_subparsers(dest='action', help='action to execute', required=True)
    parserTrain = subparsers.add_parser('train', parents=[parserCommon], help="train model")
    parserTrain.add_argument('-e', '--epochs', metavar='N', type=int, dest='nbEpochs', default=cfg.NB_EPOCHS,
                             help=f'number of training epochs (default: {cfg.NB_EPOCHS})')
    parserTrain.add_argument('-f', '--file', metavar='file_path', action='store', type=str, dest='checkpointFile',
                             default=None, help='training checkpoint file')
    parserTest = subparsers.add_parser('test', parents=[parserCommon], help="test model")
    parserTest.add_argument('-f', '--file', metavar='file_path', action='store', type=str, dest='checkpointFile',
                            default=None, required=True, help='model file')
    return parser.parse_args(argv)


def getLocalInterfaces():
    """{interface name: IPv4 address} via ioctl(SIOCGIFCONF) on a UDP socket (Linux, 40-byte ifreq records)."""
    MAX_BYTES = 4096
    SIOCGIFCONF = 0x8912
    sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    try:
        names = array.array('B', b'\0' * MAX_BYTES)
        addr, _ = names.buffer_info()
        req = struct.pack('iL', MAX_BYTES, addr)
        out_bytes, _ = struct.unpack('iL', fcntl.ioctl(sock.fileno(), SIOCGIFCONF, req))
    finally:
        sock.close()
    raw = names.tobytes()
    ips = {}
    for i in range(0, out_bytes, 40):
        name = raw[i:i + 16].split(b'\0', 1)[0].decode('utf-8')
        ips[name] = '.'.join(str(b) for b in raw[i + 20:i + 24])
    return ips


def _gpu_list(spec):
    """'0,1' -> ['0','1'];  'auto' -> every visible GPU (or one CPU worker when there is none)."""
    if spec == 'auto':
        import torch
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        return [str(i) for i in range(n)] if n else ['0']
    return [g for g in spec.split(',') if g != '']


def getDDTInfo(nodes=None, local_ips=None):
    """(this node's GPU list as CSV, firstLocalRank, worldSize) from DDTNodes, like main.getDDTInfo."""
    nodes = _cfg().DDTNodes if nodes is None else nodes
    forced = getattr(_cfg(), "NODE_ADDR", "")
    if local_ips is None and forced:
        ips = {forced}
    else:
        ips = set((getLocalInterfaces() if local_ips is None else local_ips).values())
    node = None
    worldSize = 0
    firstLocalRank = 0
    for n in nodes:
        gpus = _gpu_list(n['gpus'])
        if node is None and n['address'] in ips:
            node = dict(n, gpus=','.join(gpus))
        if node is None:
            firstLocalRank += len(gpus)
        worldSize += len(gpus)
    if node is None:
        raise RuntimeError(f"this host ({sorted(ips)}) is not listed in config.DDTNodes "
                           f"({[n['address'] for n in nodes]}): add its address or run on a listed node")
    return node['gpus'], firstLocalRank, worldSize


def _worker(local_index, args):
    # spawn target: must be importable at module level (args are pickled)
    from . import trainer
    if args.action == 'train':
        trainer.train(local_index, args)
    else:
        trainer.test(local_index, args)


def main(argv=None):
    import torch
    import torch.multiprocessing as PTmp
    from .utils import misc
    cfg = _cfg()
    args = getArgs(argv)
    args.gpus, args.firstLocalRank, args.worldSize = getDDTInfo()
    misc.initializeLogging(cfg.RSL_PATH, cfg.LOG_FILE, writer=(args.firstLocalRank == 0), debug=bool(args.debug))
    cuda = bool(misc.checkCuda())
    if not cuda:
        torch.set_num_threads(max(1, cfg.NUM_THREADS // max(1, len(args.gpus.split(',')))))
    print('========================= start =========================')
    os.environ['MASTER_ADDR'] = str(cfg.MASTER_ADDR)
    os.environ['MASTER_PORT'] = str(cfg.MASTER_PORT)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for kv in str(getattr(cfg, "NCCL_ENV", "")).split(','):          # transport knobs for multi-node RCCL
        if '=' in kv:
            k, v = kv.split('=', 1)
            os.environ[k.strip()] = v.strip()
    nprocs = len(args.gpus.split(','))
    if cuda:
        os.environ['HIP_VISIBLE_DEVICES'] = args.gpus
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpus       # torch-ROCm honours both; keep them consistent
        PTmp.spawn(_worker, nprocs=nprocs, args=(args,))
    elif args.worldSize > 1 or nprocs > 1:
        # CPU / gloo data parallelism (plumbing configuration): same process model, one rank per listed worker
        PTmp.spawn(_worker, nprocs=nprocs, args=(args,))
    else:
        _worker(-1, args)
    print('========================= end ==========================')
    return 0

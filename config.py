"""
 Run-time constants: same names and defaults as the reference's config.py (/root/reference/config.py:9-54), plus the
 knobs the MI355X-native framework adds.  Precedence: command line > environment (DPA_<NAME>) > this file.
"""
import os as _os


def _env(name, default, cast=str):
    v = _os.environ.get("DPA_" + name)
    if v is None:
        return default
    if cast is bool:
        return v.lower() in ("1", "true", "yes", "on")
    return cast(v)


DEBUG = _env("DEBUG", False, bool)

# node addresses and gpu lists used for distributed learning; the first node is the master node.
# The reference ships a 2-node example (172.16.12.8 / 172.16.12.7, 2 GPUs each); the default here is one MI355X node.
# 'gpus': 'auto' means every GPU visible on that node.
DDTNodes = [
  {'address': '127.0.0.1', 'gpus': _env("GPUS", 'auto')},
]
# DPA_DDT_NODES="addr:gpus;addr:gpus" overrides the list without editing this file, e.g.
#   DPA_DDT_NODES="172.16.12.8:0,1;172.16.12.7:0,1"       (the reference's 2-node example)
# DPA_NODE_ADDR=<address> tells the launcher which entry is this host when its interfaces do not show the listed
# address (NAT, containers) or when several "nodes" are emulated on one machine.
if _os.environ.get("DPA_DDT_NODES"):
    DDTNodes = [{'address': e.split(':', 1)[0], 'gpus': (e.split(':', 1)[1] if ':' in e else 'auto')}
                for e in _os.environ["DPA_DDT_NODES"].split(';') if e]
NODE_ADDR = _env("NODE_ADDR", '')
# two nodes, two GPUs each (the reference's example):
# DDTNodes=[ {'address':'172.16.12.8', 'gpus':'0,1'}, {'address':'172.16.12.7', 'gpus':'0,1'} ]

# the master node ip and the port used for communication with the master node
MASTER_ADDR = _env("MASTER_ADDR", DDTNodes[0]['address'])
MASTER_PORT = _env("MASTER_PORT", '6779')

MODEL_NAME = _env("MODEL_NAME", 'resnet')  # resnet | alexnet | vgg | squeezenet | densenet | inception (| resnet50 | basic)

OPTIMIZER = _env("OPTIMIZER", 'adam')  # adam | SGD

LOSS = _env("LOSS", 'cross_entropy')  # cross_entropy | weighted_cross_entropy | focal_loss

DATA_PATH = _env("DATA_PATH", './data')

RSL_PATH = _env("RSL_PATH", './rsl')

LOG_FILE = _env("LOG_FILE", 'test.log')

NB_EPOCHS = _env("NB_EPOCHS", 2, int)

BATCH_SIZE = _env("BATCH_SIZE", 64 * 1, int)

NUM_WORKERS = _env("NUM_WORKERS", 2, int)   # kept for interface parity: the input pipeline runs on the GPU

SEED = _env("SEED", 1234, int)

# torchvision model : Flag for feature extracting.
# When False, finetune the whole model, when True, only update the reshaped layer params
FEATURE_EXTRACT = _env("FEATURE_EXTRACT", False, bool)

# torchvision model: use pretrained (needs downloadable weights: rejected when True, there is no network)
USE_PRETRAINED = _env("USE_PRETRAINED", False, bool)

# if no GPU: number of threads used
NUM_THREADS = _env("NUM_THREADS", 32, int)

# ----------------------------------------------------------------------------- MI355X-native additions
COMPUTE_DTYPE = _env("COMPUTE_DTYPE", 'fp32')     # fp32 (fp32 in / out / accumulate: the reference's precision; arithmetic per FP32_MODE) | bf16 (bf16 MFMA, fp32 accumulate + fp32 master weights; ~3x faster)
FP32_MODE = _env("FP32_MODE", 'x6')               # x6 (default: each fp32 product as 6 bf16 MFMA products of EXACT 3-way bf16 operand splits, fp32 accumulate; per-element error vs fp64 not above the exact path's on every kernel variant, tests/test_gpu_x6.py) | exact (v_mfma_f32_16x16x4_f32: bit-for-bit an fp32 fma chain, ~1.3x slower convolutions)
FIRST_BUCKET_MB = _env("FIRST_BUCKET_MB", 1.0, float)    # DDP first (last-layer) bucket
BUCKET_CAP_MB = _env("BUCKET_CAP_MB", 25.0, float)       # DDP bucket cap: torch DDP's default, NOT yet swept on an 8-GPU node (scripts/allreduce_bench.py --sweep does it; SURVEY.md §5.8 expects 16-32 MiB to keep each of the 7 xGMI links bandwidth-bound)
BROADCAST_BUFFERS = _env("BROADCAST_BUFFERS", True, bool)  # DDP semantics: rank 0's BatchNorm statistics win every step
COMM_BACKEND = _env("COMM_BACKEND", 'native')      # native (default: C++ store rendezvous + ONE native communicator per rank - RCCL over xGMI on GPUs, TCP mesh on CPUs; csrc/comm) | torch (fallback: torch.distributed RCCL process group / gloo)
USE_HIPGRAPH = _env("USE_HIPGRAPH", True, bool)   # replay the training iteration from a hipGraph when world_size == 1
SYNC_COMM = _env("SYNC_COMM", False, bool)        # debug: synchronous collectives (race-detection aid)
GLOBAL_METRICS = _env("GLOBAL_METRICS", False, bool)  # all-reduce (loss, acc) over ranks; False = reference behaviour
COMPAT_EPOCH_ORDER = _env("COMPAT_EPOCH_ORDER", False, bool)  # reproduce the reference's set_epoch-after-epoch quirk (Q7)
RENDEZVOUS_TIMEOUT_S = _env("RENDEZVOUS_TIMEOUT_S", 300.0, float)
COLLECTIVE_TIMEOUT_S = _env("COLLECTIVE_TIMEOUT_S", 600.0, float)
CHECK_SYNC_EVERY = _env("CHECK_SYNC_EVERY", 0, int)   # >0: all-rank parameter checksum every N epochs
SYNTHETIC_DATA = _env("SYNTHETIC_DATA", False, bool)  # opt-in: when the real IDX files are absent under -d/MNIST/raw, train on a deterministic synthetic stand-in kept under -d/MNIST-synthetic/raw (default: fail like the reference's download=False)
# RCCL/NCCL transport knobs for multi-node runs, exported by the launcher when set (the reference keeps a commented
# NCCL_IB_DISABLE line, main.py:131): e.g. DPA_NCCL_ENV="NCCL_IB_DISABLE=1,NCCL_SOCKET_IFNAME=eth0"
NCCL_ENV = _env("NCCL_ENV", '')
ACTIVATION_ARENA = _env("ACTIVATION_ARENA", True, bool)   # planned activation arena (nn/arena.py): the step's tensors are views of ONE buffer at offsets planned from recorded lifetimes
STEP_METRICS = _env("STEP_METRICS", True, bool)   # global rank 0 writes <RSL_PATH>/steps.jsonl: step, loss, acc, images/s, data ms, comm ms, phase ms (HIP events, no host sync in the loop)
FAULT_INJECT = _env("FAULT_INJECT", '')          # test hook "rank:step": that rank exits(17) at that global step

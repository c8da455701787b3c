"""T3 end-to-end on one MI355X: training steps through DistributedModel (world_size 1), hipGraph replay equivalence,
the MNIST device pipeline + CLI, checkpoint round trip into a torch model."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(dtype, model="resnet", seed=0, **kw):
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    torch.manual_seed(seed)
    net, _ = get_model(model, 10, device="cuda", dtype=dtype, **kw)
    net.seed_rng(1234)
    opt = ArenaOptimizer(net.store, "adam", lr=1e-3)
    return net, opt, DistributedModel(net, Criterion(), opt)


def _batch(n, side, cin, dtype, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.zeros(n, side, side, cin, device="cuda", dtype=dtype)
    x[..., :3] = torch.randn(n, side, side, 3, device="cuda", generator=g).to(dtype)
    return x, torch.randint(0, 10, (n,), device="cuda", generator=g)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_reduces_loss_and_is_deterministic(dtype):
    runs = []
    for _ in range(2):
        net, opt, dm = _make(dtype)
        x, y = _batch(32, 64, net.in_channels, dtype)
        losses = []
        for it in range(12):
            dm.reset_epoch_metrics()
            dm.train_step(x, y)
            losses.append(dm.epoch_metrics()[0])
        runs.append((losses, net.store.p.clone()))
    assert runs[0][0][-1] < 0.5 * runs[0][0][0], runs[0][0]
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])      # bitwise reproducible (no float atomics)


@pytest.mark.parametrize("model", ["resnet", "alexnet"])
def test_hipgraph_replay_matches_eager(model):
    dtype = torch.bfloat16
    net_e, _, dm_e = _make(dtype, model)
    net_g, _, dm_g = _make(dtype, model)
    assert torch.equal(net_e.store.p, net_g.store.p)
    side = 64
    batches = [_batch(16, side, net_e.in_channels, dtype, seed=s) for s in range(4)]
    assert dm_g.capture_train_step(*batches[0])
    assert torch.equal(net_e.store.p, net_g.store.p)           # capture warm-up left no trace
    for x, y in batches:
        dm_e.train_step(x, y)
        dm_g.replay_train_step(x, y)
    torch.cuda.synchronize()
    assert torch.equal(net_e.store.p, net_g.store.p)
    assert dm_e.epoch_metrics() == dm_g.epoch_metrics()


def test_feature_extract_only_updates_head():
    net, opt, dm = _make(torch.float32, feature_extract=True)
    x, y = _batch(8, 64, net.in_channels, torch.float32)
    before = {s.name: net.store.get_logical(s).clone() for s in net.store.params}
    rm0 = net.state_dict("")["bn1.running_mean"].clone()
    dm.train_step(x, y)
    after = {s.name: net.store.get_logical(s) for s in net.store.params}
    changed = sorted(k for k in before if not torch.equal(before[k], after[k]))
    assert changed == ["fc.bias", "fc.weight"]
    assert not torch.equal(rm0, net.state_dict("")["bn1.running_mean"])          # BN still tracks statistics in train mode


def test_cli_mnist_train_and_test_on_gpu(tmp_path):
    """python main.py train -d ... -e 1 --debug on one MI355X, then main.py test; checkpoint loads into torch."""
    env = dict(os.environ, DPA_GPUS="0", DPA_MASTER_PORT="29533", PYTHONPATH=ROOT, DPA_MODEL_NAME="resnet", DPA_SYNTHETIC_DATA="1")
    data = str(tmp_path / "data")
    sys.path.insert(0, ROOT)
    from distributedpytorch_amd.data import mnist
    mnist.generate_synthetic_mnist(data, n_train=3000, n_test=500, seed=3)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "train", "-d", data, "-e", "3", "-b", "64"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Epoch: 003" in r.stdout
    best = tmp_path / "rsl" / "bestmodel-mnist-resnet.pt.tar"
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "test", "-d", data, "-f", str(best), "-b", "100"],
                        cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    acc = float(r2.stdout.split("Acc:")[-1].split("%")[0])
    assert acc > 70.0, r2.stdout[-1500:]      # 3 short epochs on 2.7k synthetic digits (chance = 10 %)
    from tests import torch_ref_models as ref
    ck = torch.load(best)
    m = ref.resnet18(10)
    m.load_state_dict({k[len("module."):]: v for k, v in ck["model_state_dict"].items()}, strict=True)


def test_graft_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.smoke()


def test_native_rccl_comm_world1_matches_plain():
    """The native RCCL communicator (own HIP comm stream, hipEvent fencing) at world_size 1: all-reduce over one rank
    is the identity, so training must be bit-identical to the plain path - this exercises ncclCommInitRank,
    the bucket launch order and the stream/event protocol on real hardware."""
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    from distributedpytorch_amd.parallel.rccl_native import NativeRcclComm
    comm = NativeRcclComm(None, 0, 1, 0)
    try:
        nets = []
        for native in (None, comm):
            torch.manual_seed(0)
            net, _ = get_model("resnet", 10, device="cuda", dtype=torch.bfloat16)
            dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3), first_bucket_mb=0.25,
                                  bucket_cap_mb=4.0, native_comm=native)
            assert len(dm.reducer.buckets) > 4
            x, y = _batch(16, 64, net.in_channels, torch.bfloat16)
            for _ in range(3):
                dm.train_step(x, y)
            torch.cuda.synchronize()
            nets.append(net.store.p.clone())
        assert comm.async_error() == 0
        assert torch.equal(nets[0], nets[1])
    finally:
        comm.destroy()


def test_torch_rccl_collectives_world1_match_plain():
    """The default DDP backend on real hardware: a 1-rank RCCL process group with DPA_FORCE_COLLECTIVES=1 issues every
    collective of the multi-GPU path (layout all-gather, wrap-time and per-step broadcasts, async bucket all-reduce,
    wait + per-bucket optimizer).  Over one rank they are identities, so training must match the plain path bit for bit."""
    script = r'''
import os, sys, torch
sys.path.insert(0, os.environ["DPA_ROOT"])
from distributedpytorch_amd.models.factory import get_model
from distributedpytorch_amd.nn.loss import Criterion
from distributedpytorch_amd.nn.optim import ArenaOptimizer
from distributedpytorch_amd.parallel import comm
from distributedpytorch_amd.parallel.ddp import DistributedModel
force = os.environ.get("DPA_FORCE_COLLECTIVES") == "1"
torch.cuda.set_device(0)
if force:
    comm.init_process_group(0, 1, torch.device("cuda", 0), timeout_s=60)
torch.manual_seed(0)
net, _ = get_model("resnet", 10, device="cuda", dtype=torch.bfloat16)
dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3), first_bucket_mb=0.25, bucket_cap_mb=4.0)
assert dm.reducer.force == force and len(dm.reducer.buckets) > 4
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.zeros(16, 64, 64, net.in_channels, device="cuda", dtype=torch.bfloat16)
x[..., :3] = torch.randn(16, 64, 64, 3, device="cuda", generator=g).to(torch.bfloat16)
y = torch.randint(0, 10, (16,), device="cuda", generator=g)
for _ in range(3):
    dm.train_step(x, y)
torch.cuda.synchronize()
torch.save(net.store.p.cpu(), sys.argv[1])
if force:
    assert dm.reducer.check_in_sync() == 0.0
    comm.destroy_process_group()
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w1.py")
        open(path, "w").write(script)
        outs = []
        for force in ("0", "1"):
            out = os.path.join(d, f"p{force}.pt")
            env = dict(os.environ, DPA_ROOT=ROOT, DPA_FORCE_COLLECTIVES=force, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
            r = subprocess.run([sys.executable, path, out], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
            outs.append(torch.load(out))
        assert torch.equal(outs[0], outs[1])

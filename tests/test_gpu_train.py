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


@pytest.mark.parametrize("model", ["alexnet", "resnet"])
def test_resumed_run_equals_the_uninterrupted_one_bit_for_bit(model, tmp_path):
    """Resume is state-complete (VERDICT r2 missing #7): besides parameters, BatchNorm buffers and Adam moments the
    checkpoint carries the dropout RNG step (AlexNet's masks must continue, not restart) and every BatchNorm's statistics
    shift - as extra entries of optimizer_state_dict['param_groups'][0], so the five top-level keys stay the reference's
    (utils.py:115-119) and the reference-style loader still accepts the file (test_units_cpu.py)."""
    from distributedpytorch_amd.utils import checkpoint as ckpt
    side = 224 if model == "alexnet" else 64
    net, opt, dm = _make(torch.float32, model)
    x, y = _batch(8, side, net.in_channels, torch.float32)
    for _ in range(3):
        dm.train_step(x, y)
    path = str(tmp_path / "mid.pt.tar")
    ckpt.saveCheckpoint(path, model, net, opt, 0, 1.0)
    for _ in range(3):
        dm.train_step(x, y)
    torch.cuda.synchronize()
    full = net.store.p.clone()
    saved = torch.load(path, weights_only=True)
    assert set(saved) == {"model_name", "model_state_dict", "optimizer_state_dict", "epoch", "loss"}
    assert saved["optimizer_state_dict"]["param_groups"][0]["dpa_rng_state"] == [1234, 3]
    net2, opt2, dm2 = _make(torch.float32, model, seed=99)           # different initial weights: everything must come from the file
    net2.seed_rng(77)
    assert ckpt.loadCheckpoint(0, path, net2, opt2) == (1, 1.0)
    for _ in range(3):
        dm2.train_step(x, y)
    torch.cuda.synchronize()
    assert torch.equal(net2.store.p, full)


@pytest.mark.parametrize("model", ["resnet", "densenet"])
def test_side_stream_wgrad_is_bitwise_the_single_stream_result(model, monkeypatch):
    """Weight gradients forked to the side stream on EVERY layer (threshold 0; training only forks the large ones), with
    the deferred join, against everything on one stream: same kernels, so the parameters must be bit-identical."""
    from distributedpytorch_amd.nn import engine, layers
    res = []
    for dual in (True, False):
        monkeypatch.setattr(engine, "DUAL_STREAM_BACKWARD", dual)
        monkeypatch.setattr(layers, "SIDE_MIN_GFLOP", 0.0)
        net, opt, dm = _make(torch.float32, model)
        x, y = _batch(16, 64, net.in_channels, torch.float32)
        for _ in range(4):
            dm.train_step(x, y)
        torch.cuda.synchronize()
        res.append(net.store.p.clone())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("model,dtype", [("resnet", torch.float32), ("alexnet", torch.float32), ("densenet", torch.float32),
                                         ("resnet", torch.bfloat16)])
def test_batched_weight_prep_is_bitwise_the_per_layer_path(model, dtype, monkeypatch):
    """nn/wprep.py: all filter transposes / bf16 plane splits of a step in ONE launch (from the second step on) against
    the per-layer kernels: same values, so training must be bit-identical - including Linear-as-GEMM layers (AlexNet)
    and an evaluation pass between training steps."""
    from distributedpytorch_amd.nn import wprep
    res = []
    for enabled in (True, False):
        monkeypatch.setattr(wprep, "ENABLED", enabled)
        net, opt, dm = _make(dtype, model)
        x, y = _batch(8, 64, net.in_channels, dtype)
        for it in range(4):
            dm.train_step(x, y)
            if it == 1:
                net.eval()
                _, out = net.forward(x)
                ev = out.data.clone() if hasattr(out, "data") else out[0].data.clone()
                net.train()
        torch.cuda.synchronize()
        assert (len(net.store.wprep.ready) > 0) == enabled
        res.append((net.store.p.clone(), ev))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("model,dtype", [("resnet", torch.float32), ("resnet50", torch.float32), ("inception", torch.float32),
                                         ("resnet", torch.bfloat16)])
def test_eval_batchnorm_folded_into_conv_epilogue(model, dtype, monkeypatch):
    """Evaluation applies BatchNorm (+ residual + ReLU) in the conv epilogue (ConvParams::oscale) instead of a separate
    pass over the activations: same logits as the unfolded path, after a few training steps so that the running statistics
    and the affine parameters are not the initial ones."""
    from distributedpytorch_amd.nn import layers
    net, opt, dm = _make(dtype, model)
    side = 299 if model == "inception" else 64
    x, y = _batch(4, side, net.in_channels, dtype)
    for _ in range(3):
        dm.train_step(x, y)
    net.eval()
    outs = []
    for fold in (True, False):
        monkeypatch.setattr(layers, "_FOLD_EVAL_BN", fold)
        _, out = net.forward(x)
        outs.append((out[0] if isinstance(out, tuple) else out).data.float().clone())
    torch.cuda.synchronize()
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(outs[0], outs[1], **tol)


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


_W1_SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, os.environ["DPA_ROOT"])
from distributedpytorch_amd.models.factory import get_model
from distributedpytorch_amd.nn.loss import Criterion
from distributedpytorch_amd.nn.optim import ArenaOptimizer
from distributedpytorch_amd.parallel import comm
from distributedpytorch_amd.parallel.ddp import DistributedModel
force = os.environ.get("DPA_FORCE_COLLECTIVES") == "1"
backend = os.environ["BACKEND"]
graph = os.environ.get("GRAPH") == "1"
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
nc = None
if force and backend == "native":
    nc = comm.init_native(0, 1, torch.device("cuda", 0), "127.0.0.1", 29577, create_for_single_rank=True)
    assert nc.name == "rccl_native" and not torch.distributed.is_initialized()      # ONE communicator, no process group
elif force:
    comm.init_process_group(0, 1, torch.device("cuda", 0), timeout_s=60)
torch.manual_seed(0)
net, _ = get_model("resnet", 10, device="cuda", dtype=torch.bfloat16)
dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3), first_bucket_mb=0.25, bucket_cap_mb=4.0,
                      native_comm=nc)
assert dm.reducer.force == force and dm.reducer.active == force and len(dm.reducer.buckets) > 4
# hipGraph capture of a step WITH collectives is opt-in (never seen with more than one rank)
assert dm.can_capture() == (not force or (backend == "native" and os.environ.get("DPA_GRAPH_COLLECTIVES") == "1"))
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.zeros(16, 64, 64, net.in_channels, device="cuda", dtype=torch.bfloat16)
x[..., :3] = torch.randn(16, 64, 64, 3, device="cuda", generator=g).to(torch.bfloat16)
y = torch.randint(0, 10, (16,), device="cuda", generator=g)
if graph:
    assert dm.capture_train_step(x, y), "capture refused"
    for _ in range(3):
        dm.replay_train_step(x, y)
else:
    for _ in range(3):
        dm.train_step(x, y)
torch.cuda.synchronize()
torch.save(net.store.p.cpu(), sys.argv[1])
if force:
    assert dm.reducer.check_in_sync() == 0.0
    if nc is not None:
        assert nc.async_error() == 0 and nc.inflight() == 0
        launched = dm.reducer._core_lib.dpa_reducer_launched_bytes(dm.reducer._core)
        assert launched >= 3 * dm.reducer.comm_bytes
    comm.shutdown()
'''


def _run_w1(tmpdir, tag, **env_extra):
    path = os.path.join(tmpdir, "w1.py")
    open(path, "w").write(_W1_SCRIPT)
    out = os.path.join(tmpdir, f"p_{tag}.pt")
    env = dict(os.environ, DPA_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", **env_extra)
    r = subprocess.run([sys.executable, path, out], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return torch.load(out)


@pytest.mark.parametrize("backend", ["native", "torch"])
def test_forced_collectives_world1_match_plain(backend, tmp_path):
    """DPA_FORCE_COLLECTIVES=1 issues every collective of the multi-GPU path on ONE rank of real RCCL (layout exchange,
    wrap-time and per-step broadcasts, async bucket all-reduce on the comm stream with hipEvent fencing, wait + per-bucket
    optimizer).  Over one rank they are identities, so training must match the plain path bit for bit.
    backend='native' (the default path): the C++ communicator, and NO torch.distributed process group underneath;
    backend='torch': the fallback path through ProcessGroupNCCL."""
    plain = _run_w1(str(tmp_path), "plain", DPA_FORCE_COLLECTIVES="0", BACKEND=backend)
    forced = _run_w1(str(tmp_path), "forced", DPA_FORCE_COLLECTIVES="1", BACKEND=backend)
    assert torch.equal(plain, forced)


def test_hipgraph_capture_with_native_collectives_matches_eager(tmp_path):
    """hipGraph for world > 1: the whole iteration INCLUDING the communicator's event fencing
    and ncclAllReduce / ncclBroadcast on the comm stream is captured (RCCL supports stream capture) and replayed; at
    world size 1 with forced collectives the replay must equal the eager run bit for bit."""
    eager = _run_w1(str(tmp_path), "eager", DPA_FORCE_COLLECTIVES="1", BACKEND="native", GRAPH="0")
    replay = _run_w1(str(tmp_path), "graph", DPA_FORCE_COLLECTIVES="1", BACKEND="native", GRAPH="1", DPA_GRAPH_COLLECTIVES="1")
    assert torch.equal(eager, replay)


def _shared_gpu_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.parallel import comm
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    nc = comm.init_native(rank, world, torch.device("cuda", 0), "127.0.0.1", port, timeout_s=120, transport="tcp")
    assert nc.name == "tcp_native"
    torch.manual_seed(100 + rank)                               # wrap-time broadcast must make the ranks identical
    net, _ = get_model("resnet", 10, device="cuda", dtype=torch.bfloat16)
    dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3), first_bucket_mb=0.25,
                          bucket_cap_mb=4.0, native_comm=nc)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.zeros(16, 64, 64, net.in_channels, device="cuda", dtype=torch.bfloat16)
    x[..., :3] = torch.randn(16, 64, 64, 3, device="cuda", generator=g).to(torch.bfloat16)
    y = torch.randint(0, 10, (16,), device="cuda", generator=g)
    grads = []
    for _ in range(3):
        dm.train_step(x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8])
        grads.append(net.store.g.clone().cpu())
    torch.cuda.synchronize()
    assert dm.reducer.check_in_sync() == 0.0
    torch.save({"p": net.store.p.cpu(), "g": grads}, os.path.join(out_dir, f"sg{rank}.pt"))
    comm.shutdown()


def test_two_ranks_share_one_gpu_through_the_staged_tcp_transport(tmp_path):
    """World size 2 ON the GPU: RCCL refuses two ranks on one device, so the rehearsal uses the communicator's TCP
    transport with device buffers staged through pinned memory - the same front (slot ring, bucket map), the same
    reducer core, real ready/done hipEvents between the compute stream and the transport's stream, real kernels."""
    import torch.multiprocessing as mp
    mp.spawn(_shared_gpu_worker, nprocs=2, args=(2, 29611, str(tmp_path)), join=True)
    r0, r1 = torch.load(tmp_path / "sg0.pt"), torch.load(tmp_path / "sg1.pt")
    assert torch.equal(r0["p"], r1["p"])
    for a, b in zip(r0["g"], r1["g"]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_bf16_resnet50_step_is_bit_reproducible_at_imagenet_size():
    """Three fresh bf16 ResNet-50 models (same seed, same 224 x 224 batch): logits and every gradient bit-identical.  The
    small-input determinism test above did not see the statistics race of the bf16 register-staged convolution kernel
    (tests/test_gpu_conv.py::test_fused_statistics_are_the_same_from_launch_to_launch): it hit about one tile in 1500."""
    from distributedpytorch_amd.models.factory import get_model
    from tests.helpers import framework_step, grads_by_name
    torch.manual_seed(3)
    x = torch.randn(48, 3, 224, 224)
    y = torch.randint(0, 1000, (48,))
    runs = []
    for _ in range(3):
        torch.manual_seed(5)
        net, _ = get_model("resnet50", 1000, device="cuda", dtype=torch.bfloat16)
        logits, loss, _ = framework_step(net, x, y)
        runs.append((logits.clone(), loss, {k: v.clone() for k, v in grads_by_name(net).items()}))
        del net
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and r[1] == runs[0][1]
        for k, gk in runs[0][2].items():
            assert torch.equal(r[2][k], gk), k

"""T0 unit tests (pure CPU): sampler parity with torch, topology resolver, CLI, IDX files, bucket planner, optimizer
maths vs torch.optim, losses vs torch, TCP store, checkpoint contract."""
import math
import os
import threading

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import given, settings, strategies as st

from distributedpytorch_amd.data.sampler import DistributedSampler


# ------------------------------------------------------------------------------------------------ sampler
@settings(max_examples=60, deadline=None)
@given(n=st.integers(1, 500), world=st.integers(1, 9), epoch=st.integers(0, 5), seed=st.integers(0, 3), data=st.data())
def test_sampler_matches_torch(n, world, epoch, seed, data):
    rank = data.draw(st.integers(0, world - 1))
    ref = torch.utils.data.distributed.DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=True, seed=seed)
    ref.set_epoch(epoch)
    mine = DistributedSampler(n, world, rank, shuffle=True, seed=seed)
    mine.set_epoch(epoch)
    assert list(ref) == mine.indices().tolist()
    assert len(ref) == len(mine)


def test_sampler_set_epoch_reshuffles_and_shards_cover():
    n, world = 103, 4
    shards = []
    for r in range(world):
        s = DistributedSampler(n, world, r)
        e0 = s.indices().tolist()
        s.set_epoch(1)
        assert s.indices().tolist() != e0
        s.set_epoch(0)
        assert s.indices().tolist() == e0
        shards.append(e0)
    flat = [i for sh in shards for i in sh]
    assert len(flat) == math.ceil(n / world) * world and set(flat) == set(range(n))


def test_sampler_unshuffled_is_sequential():
    s = DistributedSampler(10, 1, 0, shuffle=False)
    assert s.indices().tolist() == list(range(10))


# ------------------------------------------------------------------------------------------------ launcher
def test_cli_matches_reference_interface():
    from distributedpytorch_amd.launcher import getArgs
    a = getArgs(["train", "-d", "/data", "-b", "32", "-e", "5", "-f", "ck.pt", "--debug"])
    assert (a.action, a.dataPath, a.batchSize, a.nbEpochs, a.checkpointFile, a.debug) == ("train", "/data", 32, 5, "ck.pt", True)
    a = getArgs(["train", "--data_path", "/d"])
    assert a.batchSize == 64 and a.nbEpochs == 2 and a.checkpointFile is None and a.debug is False
    a = getArgs(["test", "-d", "/d", "--file", "m.pt"])
    assert a.action == "test" and a.checkpointFile == "m.pt"
    with pytest.raises(SystemExit):
        getArgs(["test", "-d", "/d"])             # -f is required for test
    with pytest.raises(SystemExit):
        getArgs(["train"])                        # -d is required


def test_topology_resolver():
    from distributedpytorch_amd.launcher import getDDTInfo
    nodes = [{"address": "10.0.0.1", "gpus": "0,1"}, {"address": "10.0.0.2", "gpus": "0,1,2"}, {"address": "10.0.0.3", "gpus": "0"}]
    assert getDDTInfo(nodes, {"lo": "127.0.0.1", "eth0": "10.0.0.1"}) == ("0,1", 0, 6)
    assert getDDTInfo(nodes, {"eth0": "10.0.0.2"}) == ("0,1,2", 2, 6)
    assert getDDTInfo(nodes, {"eth0": "10.0.0.3"}) == ("0", 5, 6)
    with pytest.raises(RuntimeError, match="not listed"):
        getDDTInfo(nodes, {"eth0": "10.9.9.9"})


def test_local_interfaces_has_loopback():
    from distributedpytorch_amd.launcher import getLocalInterfaces
    assert "127.0.0.1" in getLocalInterfaces().values()


# ------------------------------------------------------------------------------------------------ IDX + dataset
def test_idx_roundtrip_and_magic(tmp_path):
    from distributedpytorch_amd.data.mnist import read_idx, write_idx
    imgs = np.random.RandomState(0).randint(0, 256, size=(7, 28, 28)).astype(np.uint8)
    labels = np.arange(7, dtype=np.uint8)
    write_idx(str(tmp_path / "i"), imgs)
    write_idx(str(tmp_path / "l"), labels)
    raw = open(tmp_path / "i", "rb").read()
    assert raw[:4] == bytes([0, 0, 8, 3]) and int.from_bytes(raw[4:8], "big") == 7      # magic 2051, big-endian dims
    assert open(tmp_path / "l", "rb").read()[:4] == bytes([0, 0, 8, 1])                # magic 2049
    assert np.array_equal(read_idx(str(tmp_path / "i")), imgs)
    assert np.array_equal(read_idx(str(tmp_path / "l")), labels)
    with pytest.raises(FileNotFoundError):
        read_idx(str(tmp_path / "missing"))


@pytest.fixture(scope="module")
def small_mnist(tmp_path_factory):
    from distributedpytorch_amd.data import mnist
    root = str(tmp_path_factory.mktemp("data"))
    mnist.generate_synthetic_mnist(root, n_train=600, n_test=100, seed=1)
    return root


def test_mnist_split_sizes_stats_and_determinism(small_mnist):
    from distributedpytorch_amd.data.mnist import MNIST
    from distributedpytorch_amd.utils.misc import setRandomSeed
    sets = []
    for rank in (0, 1):
        setRandomSeed(1234)
        ds = MNIST(small_mnist, 32, 28, worldSize=2, rank=rank, gpu=0, device=torch.device("cpu"), synthetic_ok=True)
        sets.append(ds)
    a, b = sets
    assert len(a.data["train"]) == 540 and len(a.data["valid"]) == 60 and len(a.data["test"]) == 100
    assert torch.equal(a.data["train"].indices, b.data["train"].indices)          # same partition on every rank
    assert 0 < a.mean < 1 and 0 < a.std < 1
    ia = a.sampler["train"].indices().tolist()
    ib = b.sampler["train"].indices().tolist()
    assert len(ia) == 270 and not set(ia) & set(ib)
    x, y = next(iter(a.iterator["train"]))
    assert x.shape == (32, 28, 28, 4) and y.shape == (32,) and x[..., 3:].abs().max() == 0   # fp32: 3 -> 4 channels
    x2, _ = next(iter(a.iterator["train"]))
    assert torch.equal(x, x2)                                                       # seeded augmentation
    a.sampler["train"].set_epoch(1)
    x3, _ = next(iter(a.iterator["train"]))
    assert not torch.equal(x, x3)
    w = a.data["train"].classWeights
    assert w.shape == (10,) and torch.all(w > 0)


def test_missing_mnist_fails_like_the_reference_unless_opted_in(tmp_path):
    """A mistyped -d must not silently train on generated data (the reference: torchvision MNIST(download=False) raises);
    opted in, the stand-in lives under MNIST-synthetic/raw - never under the real dataset's MNIST/raw."""
    from distributedpytorch_amd.data import mnist
    with pytest.raises(FileNotFoundError):
        mnist.MNIST(str(tmp_path), 32, 28, worldSize=1, rank=0, gpu=-1, device=torch.device("cpu"))
    assert not os.path.exists(mnist.raw_dir(str(tmp_path)))
    mnist.generate_synthetic_mnist(str(tmp_path), n_train=200, n_test=50, seed=1)
    assert mnist.have_synthetic(str(tmp_path)) and not mnist.have_mnist(str(tmp_path))
    ds = mnist.MNIST(str(tmp_path), 32, 28, worldSize=1, rank=0, gpu=-1, device=torch.device("cpu"), synthetic_ok=True)
    assert len(ds.data["train"]) == 180


def test_mnist_debug_subset(small_mnist):
    from distributedpytorch_amd.data.mnist import MNIST
    ds = MNIST(small_mnist, 32, 28, worldSize=1, rank=0, gpu=0, device=torch.device("cpu"), debug=True, synthetic_ok=True)
    assert len(ds.data["train"]) == 200 and len(ds.data["valid"]) == 60


# ------------------------------------------------------------------------------------------------ bucket planner
def test_bucket_planner_contiguous_reverse_order():
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.parallel.reducer import plan_buckets
    net, _ = get_model("resnet", 10, device="cpu")
    buckets = plan_buckets(net.store, first_mb=1.0, cap_mb=25.0)
    assert len(buckets) >= 2
    assert buckets[0]["params"][-1].name == "fc.bias"                              # last layer is ready first
    assert (buckets[0]["hi"] - buckets[0]["lo"]) * 4 <= 1.1 * 2 ** 20 or len(buckets[0]["params"]) == 1
    his = [b["hi"] for b in buckets]
    assert his == sorted(his, reverse=True)                                         # backward order = descending offsets
    total = sum(b["hi"] - b["lo"] for b in buckets)
    assert total == net.store.p_size
    for b in buckets[1:]:
        assert (b["hi"] - b["lo"]) * 4 <= 25 * 2 ** 20 or len(b["params"]) == 1


def test_bucket_planner_skips_frozen():
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.parallel.reducer import plan_buckets
    net, _ = get_model("resnet", 10, feature_extract=True, device="cpu")
    buckets = plan_buckets(net.store)
    names = [s.name for b in buckets for s in b["params"]]
    assert sorted(names) == ["fc.bias", "fc.weight"]
    assert net.store.trainable_ranges() == [(buckets[0]["lo"], buckets[0]["hi"])]


# ------------------------------------------------------------------------------------------------ optimizers
@pytest.mark.parametrize("kind", ["adam", "SGD"])
def test_arena_optimizer_matches_torch_and_state_dict(kind):
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.optim import ArenaOptimizer, StepLR
    from tests import torch_ref_models as ref
    torch.manual_seed(0)
    oracle = ref.resnet18(10, small_input=True)
    net, _ = get_model("resnet", 10, device="cpu", small_input=True)
    net.load_state_dict(oracle.state_dict())
    opt = ArenaOptimizer(net.store, kind, lr=1e-3)
    topt = torch.optim.Adam(oracle.parameters(), lr=1e-3) if kind == "adam" else torch.optim.SGD(oracle.parameters(), lr=1e-3, momentum=0.9)
    sched = StepLR(opt, 1, 0.1) if kind == "SGD" else None
    tsched = torch.optim.lr_scheduler.StepLR(topt, 1, 0.1) if kind == "SGD" else None
    st = net.store
    for it in range(3):
        for spec, p in zip(st.params, oracle.parameters()):
            g = torch.randn_like(p)
            p.grad = g
            st.logical_to_storage(spec, g, st.gview(spec))
        opt.step()
        topt.step()
        if sched is not None and it == 1:
            sched.step()
            tsched.step()
    mine = net.state_dict(prefix="")
    for k, v in oracle.state_dict().items():
        if v.dtype.is_floating_point:
            torch.testing.assert_close(mine[k], v, rtol=1e-5, atol=1e-6)
    # torch.optim must accept our optimizer state_dict and continue identically
    sd = opt.state_dict()
    topt2 = torch.optim.Adam(oracle.parameters(), lr=1e-3) if kind == "adam" else torch.optim.SGD(oracle.parameters(), lr=1e-3, momentum=0.9)
    topt2.load_state_dict(sd)
    ref_sd = topt.state_dict()
    assert sd["param_groups"][0]["params"] == ref_sd["param_groups"][0]["params"]
    assert abs(sd["param_groups"][0]["lr"] - ref_sd["param_groups"][0]["lr"]) < 1e-12
    for i in ref_sd["state"]:
        for k, v in ref_sd["state"][i].items():
            torch.testing.assert_close(sd["state"][i][k], v, rtol=1e-4, atol=1e-6)
    # and our optimizer resumes from torch's state_dict
    opt2 = ArenaOptimizer(net.store, kind, lr=1e-3)
    opt2.load_state_dict(ref_sd)
    torch.testing.assert_close(opt2.m, opt.m, rtol=1e-4, atol=1e-6)
    assert float(opt2.hyper[5]) == float(opt.hyper[5]) or kind == "SGD"


def test_feature_extract_optimizer_state_dict_loads_into_reference_adam():
    """FEATURE_EXTRACT=True: the reference still builds Adam over model.parameters() - all 62 of them, frozen ones
    included (/root/reference/classif.py:124, utils.py:107-110) - so `params` must index every parameter and only the
    trainable head carries state.  torch.optim.Adam(oracle.parameters()).load_state_dict must accept ours."""
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from tests import torch_ref_models as ref
    from tests.helpers import framework_step
    torch.manual_seed(0)
    net, _ = get_model("resnet", 10, feature_extract=True, device="cpu", small_input=True)
    opt = ArenaOptimizer(net.store, "adam", lr=1e-3)
    framework_step(net, torch.randn(4, 3, 16, 16), torch.randint(0, 10, (4,)))
    opt.step()
    sd = opt.state_dict()
    n_all = len(net.store.params)
    assert n_all == 62 and sd["param_groups"][0]["params"] == list(range(n_all))
    assert sorted(sd["state"]) == [n_all - 2, n_all - 1]               # fc.weight, fc.bias: the only trainable ones
    oracle = ref.resnet18(10, small_input=True)
    for name, p in oracle.named_parameters():                          # utils.setParameterRequiresGrad + new head
        p.requires_grad = name.startswith("fc.")
    topt = torch.optim.Adam(oracle.parameters(), lr=1e-3)              # the reference's construction
    topt.load_state_dict(sd)
    got = topt.state_dict()["state"]
    assert sorted(got) == [n_all - 2, n_all - 1]
    torch.testing.assert_close(got[n_all - 2]["exp_avg"], sd["state"][n_all - 2]["exp_avg"])
    # round trip through our loader
    opt2 = ArenaOptimizer(net.store, "adam", lr=1e-3)
    opt2.load_state_dict(topt.state_dict())
    torch.testing.assert_close(opt2.m, opt.m)
    torch.testing.assert_close(opt2.v, opt.v)


# ------------------------------------------------------------------------------------------------ losses
def test_losses_match_reference_definitions():
    from distributedpytorch_amd.nn.engine import Act, Context, ParamStore, Tape
    from distributedpytorch_amd.nn.loss import Criterion
    torch.manual_seed(0)
    store = ParamStore("cpu").finalize()
    z = torch.randn(32, 1, 1, 16)
    z[..., 10:] = 0
    y = torch.randint(0, 10, (32,))
    w = torch.rand(10) + 0.5
    zz = z.reshape(32, 16)[:, :10].clone().requires_grad_(True)
    lp = F.log_softmax(zz, -1)
    focal = F.nll_loss(((1 - lp.exp()) ** 2.0) * lp, y, weight=w, reduction="none").mean()       # utils.FocalLossN
    cases = {"cross_entropy": F.cross_entropy(zz, y), "weighted_cross_entropy": F.cross_entropy(zz, y, weight=w),
             "focal_loss": focal}
    for kind, expect in cases.items():
        (gref,) = torch.autograd.grad(expect, zz, retain_graph=True)
        crit = Criterion(kind, None if kind == "cross_entropy" else w)
        ctx = Context(store, True, Tape())
        a = Act(z.clone(), True)
        out = torch.zeros(2)
        crit(ctx, a, y, 10, out)
        assert abs(float(out[0]) - float(expect)) < 1e-5
        assert abs(float(out[1]) - float((zz.argmax(1) == y).float().mean())) < 1e-6
        torch.testing.assert_close(a.grad.reshape(32, 16)[:, :10], gref, rtol=1e-4, atol=1e-6)
        assert a.grad.reshape(32, 16)[:, 10:].abs().max() == 0
    with pytest.raises(ValueError):
        Criterion("hinge")


# ------------------------------------------------------------------------------------------------ TCP store
def test_tcp_store_set_get_add_and_named_timeout():
    from distributedpytorch_amd.parallel.store import StoreTimeout, TCPStore
    master = TCPStore("127.0.0.1", 0, True, timeout_s=5)
    try:
        client = TCPStore("127.0.0.1", master.port, False, timeout_s=5)
        master.set("id", b"\x01\x02\x03")
        assert client.get("id") == b"\x01\x02\x03"
        assert client.add("ctr", 2) == 2 and master.add("ctr", 3) == 5
        got = []
        t = threading.Thread(target=lambda: got.append(client.get("late", timeout_s=5)))
        t.start()
        master.set("late", b"x")
        t.join()
        assert got == [b"x"]
        with pytest.raises(StoreTimeout):
            client.get("never", timeout_s=0.2)
        # rank 1 never arrives: the error names it
        with pytest.raises(StoreTimeout, match=r"missing ranks \[1\]"):
            master.announce_and_wait(0, 2, tag="j", timeout_s=0.5)
        th = threading.Thread(target=lambda: client.announce_and_wait(1, 2, tag="k", timeout_s=5))
        th.start()
        master.announce_and_wait(0, 2, tag="k", timeout_s=5)
        th.join()
        client.close()
    finally:
        master.close()


# ------------------------------------------------------------------------------------------------ checkpoint contract (T5)
def test_checkpoint_loads_into_reference_style_torch_model(tmp_path):
    """Write a checkpoint, then run the reference's loadCheckpoint logic (/root/reference/utils.py:123-136) against a
    plain torch ResNet-18 wrapped in a `module.` shim (what DDP does) with strict=True, and an Adam optimizer."""
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.utils import checkpoint as ck
    from tests import torch_ref_models as ref
    from tests.helpers import framework_step
    torch.manual_seed(0)
    net, _ = get_model("resnet", 10, device="cpu", small_input=True)
    opt = ArenaOptimizer(net.store, "adam", lr=1e-3)
    x, y = torch.randn(8, 3, 16, 16), torch.randint(0, 10, (8,))
    framework_step(net, x, y)
    opt.step()
    path = str(tmp_path / "checkpoint-mnist-resnet-000.pt.tar")
    ck.saveCheckpoint(path, "resnet", net, opt, 0, 1.25)

    class Shim(torch.nn.Module):                      # state_dict keys 'module.*' like DistributedDataParallel
        def __init__(self, m):
            super().__init__()
            self.module = m

    model = Shim(ref.resnet18(10, small_input=True))
    optimizer = torch.optim.Adam(model.parameters(), lr=0.001)
    checkpoint = torch.load(path)                                     # reference code path, verbatim semantics
    assert set(checkpoint) == {"model_name", "model_state_dict", "optimizer_state_dict", "epoch", "loss"}
    model.load_state_dict(checkpoint["model_state_dict"])             # strict=True
    optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
    # the two extra entries (dropout RNG state, BatchNorm statistics shifts) ride along without upsetting torch's loader
    assert len(optimizer.param_groups[0]["dpa_rng_state"]) == 2
    assert optimizer.param_groups[0]["dpa_bn_stat_shift"].numel() == sum(a[9].numel() for a in net.store.aux_tensors)
    assert checkpoint["epoch"] + 1 == 1 and checkpoint["loss"] == 1.25 and checkpoint["model_name"] == "resnet"
    assert len(checkpoint["model_state_dict"]) == 122
    assert checkpoint["model_state_dict"]["module.bn1.num_batches_tracked"].dtype == torch.int64
    # the torch model now computes what ours computes
    model.eval()
    from tests.helpers import framework_step as fs
    mine, _, _ = fs(net, x, y, train=False)
    torch.testing.assert_close(model.module(x).detach(), mine, rtol=1e-3, atol=1e-4)
    # and our loader accepts both prefixed and un-prefixed keys, restoring epoch+1 / best loss
    net2, _ = get_model("resnet", 10, device="cpu", small_input=True)
    opt2 = ArenaOptimizer(net2.store, "adam", lr=1e-3)
    assert ck.loadCheckpoint(-1, path, net2, opt2) == (1, 1.25)
    assert ck.getCheckpointModelName(path) == "resnet"
    torch.testing.assert_close(net2.store.p, net.store.p)
    torch.testing.assert_close(opt2.m, opt.m)
    net2.load_state_dict(model.module.state_dict())                   # un-prefixed keys


def test_native_reducer_core_protocol():
    """csrc/comm/reducer.cpp bookkeeping (no communicator): ready detection, double-ready and missing-gradient errors,
    plan-order diagnostics - the stream/event-free half of the T1 'reducer discipline' tests."""
    import ctypes
    from distributedpytorch_amd.parallel.native_comm import comm_lib
    lib = comm_lib()
    # 5 params: p0 frozen, buckets in backward order: b0 = {p4}, b1 = {p3, p2}, b2 = {p1}
    bucket_of = (ctypes.c_int * 5)(-1, 2, 1, 1, 0)
    lo = (ctypes.c_longlong * 3)(300, 100, 0)
    hi = (ctypes.c_longlong * 3)(400, 300, 100)
    g = torch.zeros(400)
    core = ctypes.c_void_p(lib.dpa_reducer_create(None, ctypes.c_void_p(g.data_ptr()), 5, bucket_of, 3, lo, hi))
    assert core
    lib.dpa_reducer_begin(core)
    assert lib.dpa_reducer_param_ready(core, 0, None) == -1            # frozen
    assert lib.dpa_reducer_param_ready(core, 4, None) == 0             # bucket 0 complete
    assert lib.dpa_reducer_param_ready(core, 3, None) == -1
    assert lib.dpa_reducer_param_ready(core, 3, None) == -2            # reported twice
    assert lib.dpa_reducer_wait_bucket(core, 1, None) == 1             # p2 still missing
    assert lib.dpa_reducer_pending(core, 1) == 1
    assert lib.dpa_reducer_param_ready(core, 2, None) == 1
    assert lib.dpa_reducer_param_ready(core, 1, None) == 2
    assert [lib.dpa_reducer_wait_bucket(core, b, None) for b in range(3)] == [0, 0, 0]
    assert lib.dpa_reducer_order_violations(core) == 0
    assert lib.dpa_reducer_launched_bytes(core) == 400 * 4
    lib.dpa_reducer_begin(core)                                        # next iteration, out-of-plan order
    assert lib.dpa_reducer_param_ready(core, 1, None) == 2
    assert lib.dpa_reducer_order_violations(core) == 1
    assert lib.dpa_reducer_param_ready(core, 9, None) == -2            # unknown parameter
    lib.dpa_reducer_destroy(core)
    bad = (ctypes.c_int * 1)(7)
    assert not lib.dpa_reducer_create(None, None, 1, bad, 3, lo, hi)   # bucket index out of range


def test_fp32_mode_switch_roundtrip():
    """ops.native.set_fp32_mode / get_fp32_mode (host-side state of the kernel library; no GPU call involved)."""
    from distributedpytorch_amd.ops import native
    if not native.have_hip_lib():
        pytest.skip("kernel library not built")
    try:
        native.hip_lib()
    except OSError:
        pytest.skip("HIP runtime not loadable on this host")
    prev = native.get_fp32_mode()
    try:
        assert native.set_fp32_mode("x6") == prev
        assert native.get_fp32_mode() == "x6"
        assert native.set_fp32_mode("exact") == "x6"
        assert native.get_fp32_mode() == "exact"
        with pytest.raises(ValueError):
            native.set_fp32_mode("tf32")
    finally:
        native.set_fp32_mode(prev)


def test_activation_arena_plan_replay_and_bitwise_training():
    """nn/arena.py: the planner never overlaps live buffers, replayed steps are bit-identical to allocator-served steps,
    verify mode accepts a reproducible step, and a step that deviates from the recording drops its plan."""
    import ctypes
    import random
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.ops import native
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    # --- the C++ planner on random lifetimes
    lib = native.host_lib()
    lib.dpa_plan_arena.restype = ctypes.c_longlong
    rnd = random.Random(3)
    n = 300
    arr = ctypes.c_longlong * n
    size, start, end, off = arr(), arr(), arr(), arr()
    for i in range(n):
        size[i], start[i], end[i] = rnd.randint(1, 5000), i, i + rnd.randint(1, 60)
    lb = ctypes.c_longlong(0)
    peak = lib.dpa_plan_arena(n, size, start, end, ctypes.c_longlong(256), off, ctypes.byref(lb))
    assert peak >= lb.value > 0 and peak <= 1.3 * lb.value
    for i in range(n):
        assert off[i] % 256 == 0 and off[i] + size[i] <= peak
        for j in range(i):
            if start[j] < end[i] and start[i] < end[j]:
                assert off[i] + size[i] <= off[j] or off[j] + size[j] <= off[i], (i, j)

    # --- training with and without the arena: identical parameters after 4 steps
    def run(arena, env=None):
        for k, v in (env or {}).items():
            os.environ[k] = v
        try:
            torch.manual_seed(0)
            net, _ = get_model("resnet", 10, device="cpu", dtype=torch.float32)
            dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3), activation_arena=arena)
            g = torch.Generator().manual_seed(1)
            x = torch.randn(2, 32, 32, net.in_channels, generator=g)
            y = torch.randint(0, 10, (2,), generator=g)
            for _ in range(4):
                dm.train_step(x, y)
            dm.eval_step(x, y)
            dm.train_step(x, y)
            return net.store.p.clone(), dm
        finally:
            for k in (env or {}):
                os.environ.pop(k, None)
    p0, _ = run(False)
    p1, dm = run(True)
    assert torch.equal(p0, p1)
    st = dm.arena.stats
    assert st["recorded"] == 2 and st["replayed"] >= 3 and st["dropped"] == 0, st       # one recording per step kind
    summ = dm.arena.summary()
    train = [v for k, v in summ.items() if "train" in k][0]
    assert train["in_arena"] > 50 and train["arena_bytes"] <= 1.15 * train["live_set_lower_bound"], train
    p2, dm2 = run(True, {"DPA_ACT_ARENA_VERIFY": "1"})          # every step re-recorded and compared with the plan
    assert torch.equal(p0, p2) and dm2.arena.stats["replayed"] == 0
    # --- a different batch shape is another key; a request that deviates from the recording drops the plan
    x2 = torch.randn(3, 32, 32, dm.net.in_channels)
    dm.train_step(x2, torch.randint(0, 10, (3,)))
    assert len(dm.arena.plans) == 3 or dm.arena.stats["recorded"] >= 2
    a = dm.arena
    key = next(k for k in a.plans if k[0] == "train" and k[1][0] == 2)
    a.begin(key)
    t = a.alloc((7, 7), torch.float32)                          # not what the recorded step asked for first
    assert t.shape == (7, 7) and key not in a.plans and a.stats["dropped"] == 1

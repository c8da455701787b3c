"""T1 plumbing (BASELINE.json config 1): real 2-process data parallelism on CPU over gloo - the same reducer, bucket
planner and trainer code that runs over RCCL on MI355X."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bring_up(backend, rank, world, port):
    """'gloo': torch.distributed process group (fallback path).  'native': C++ store + ONE native communicator (TCP mesh
    transport on CPUs) - the default path; no process group exists.  Returns the native communicator or None."""
    from distributedpytorch_amd.parallel import comm
    if backend == "native":
        return comm.init_native(rank, world, torch.device("cpu"), "127.0.0.1", port, timeout_s=120, rendezvous_timeout_s=120)
    comm.init_process_group(rank, world, torch.device("cpu"), timeout_s=120)
    return None


def _ddp_worker(rank, world, port, out_dir, sync_comm, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.parallel import comm
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    from tests.helpers import to_nhwc_padded
    from tests.test_models_cpu import _zero_dropout
    torch.set_num_threads(2)
    nc = _bring_up(backend, rank, world, port)
    torch.manual_seed(100 + rank)                     # different init per rank: the wrap-time broadcast must fix it
    net, _ = get_model("resnet", 10, device="cpu", small_input=True)
    _zero_dropout(net)
    opt = ArenaOptimizer(net.store, "adam", lr=1e-3)
    dm = DistributedModel(net, Criterion(), opt, first_bucket_mb=0.05, bucket_cap_mb=2.0, sync_comm=sync_comm, native_comm=nc)
    assert len(dm.reducer.buckets) > 3
    p0 = net.store.p.clone()
    g = torch.Generator().manual_seed(7)
    xs = torch.randn(3, 8, 3, 16, 16, generator=g)
    ys = torch.randint(0, 10, (3, 8), generator=g)
    grads = []
    for it in range(3):
        x = to_nhwc_padded(xs[it, rank * 4:(rank + 1) * 4], torch.float32, "cpu")
        dm.train_step(x, ys[it, rank * 4:(rank + 1) * 4])
        grads.append(net.store.g.clone())
    assert dm.reducer.check_in_sync() == 0.0
    launched = dm.reducer._core_lib.dpa_reducer_launched_bytes(dm.reducer._core)
    torch.save({"p0": p0, "p": net.store.p.clone(), "b": net.store.b.clone(), "g": grads, "loss": dm.epoch_metrics(),
                "launched": launched, "comm_bytes": dm.reducer.comm_bytes,
                "inflight": nc.inflight() if nc is not None else 0},
               os.path.join(out_dir, f"rank{rank}.pt"))
    comm.shutdown()


@pytest.mark.parametrize("backend,sync_comm", [("gloo", False), ("gloo", True), ("native", False), ("native", True)])
def test_ddp_two_ranks(tmp_path, backend, sync_comm):
    """2 ranks, ResNet-18, 3 Adam steps.  backend='native' runs csrc/comm/reducer.cpp -> dpa_comm_allreduce_async ->
    slot ring -> TCP transport -> dpa_comm_wait: the C++ launch path the RCCL transport shares (VERDICT r1 missing #1)."""
    world = 2
    mp.spawn(_ddp_worker, nprocs=world, args=(world, _free_port(), str(tmp_path), sync_comm, backend), join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["launched"] == 3 * r0["comm_bytes"] and r0["inflight"] == 0   # every bucket launched and waited, 3 iterations
    assert torch.equal(r0["p0"], r1["p0"])                  # X3: rank 0's parameters were broadcast at wrap time
    assert torch.equal(r0["p"], r1["p"])                    # identical parameters after 3 steps
    assert not torch.equal(r0["p"], r0["p0"])
    for a, b in zip(r0["g"], r1["g"]):
        assert torch.equal(a, b)                            # all-reduced gradient buckets are identical (SUM; 1/world in Adam)
    assert r0["loss"][2] == 3


def _eval_sync_worker(rank, world, port, out_dir, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.parallel import comm
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    from tests.helpers import to_nhwc_padded
    torch.set_num_threads(2)
    nc = _bring_up(backend, rank, world, port)
    torch.manual_seed(3)
    net, _ = get_model("resnet", 10, device="cpu", small_input=True)
    dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3), native_comm=nc)
    g = torch.Generator().manual_seed(50 + rank)                # DIFFERENT training data per rank -> different BN statistics
    for _ in range(2):
        dm.train_step(to_nhwc_padded(torch.randn(4, 3, 16, 16, generator=g), torch.float32, "cpu"),
                      torch.randint(0, 10, (4,), generator=g))
    b_before = net.store.b.clone()
    ge = torch.Generator().manual_seed(99)                      # SAME evaluation batch on both ranks
    xe = to_nhwc_padded(torch.randn(4, 3, 16, 16, generator=ge), torch.float32, "cpu")
    ye = torch.randint(0, 10, (4,), generator=ge)
    dm.reset_epoch_metrics()
    dm.eval_step(xe, ye)
    first = dm.epoch_metrics()
    flag_after_first = dm._need_buffer_sync
    dm.eval_step(xe, ye)
    torch.save({"b_before": b_before, "b": net.store.b.clone(), "nbt": net.store.nbt.clone(), "loss": first[0],
                "flag": flag_after_first}, os.path.join(out_dir, f"eval{rank}.pt"))
    comm.shutdown()


@pytest.mark.parametrize("backend", ["gloo", "native"])
def test_first_eval_forward_after_training_broadcasts_rank0_buffers(tmp_path, backend):
    """X4 of SURVEY.md §2.6 (torch DDP behind /root/reference/classif.py:159): after a training phase the ranks' BatchNorm
    running statistics differ (each saw its own last batch); the first evaluation forward broadcasts rank 0's, so every
    rank evaluates the same model and reports the same validation loss on the same inputs."""
    mp.spawn(_eval_sync_worker, nprocs=2, args=(2, _free_port(), str(tmp_path), backend), join=True)
    r0 = torch.load(tmp_path / "eval0.pt")
    r1 = torch.load(tmp_path / "eval1.pt")
    assert not torch.equal(r0["b_before"], r1["b_before"])      # the last training step left rank-local statistics
    assert torch.equal(r0["b"], r0["b_before"])                 # rank 0's statistics win ...
    assert torch.equal(r1["b"], r0["b"]) and torch.equal(r1["nbt"], r0["nbt"])
    assert r0["loss"] == r1["loss"]                             # ... and rank 1's validation loss equals rank 0's
    assert r0["flag"] is False and r1["flag"] is False          # later eval forwards do not broadcast again


def _single_vs_ddp_worker(rank, world, port, out_dir, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.parallel import comm
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    from tests.helpers import to_nhwc_padded
    from tests.test_models_cpu import _zero_dropout
    torch.set_num_threads(2)
    nc = _bring_up(backend, rank, world, port)
    torch.manual_seed(5)
    net, _ = get_model("basic", 10, device="cpu")
    _zero_dropout(net)
    dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "SGD", lr=1e-2), native_comm=nc)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 3, 28, 28, generator=g)
    y = torch.randint(0, 10, (8,), generator=g)
    per = 8 // world
    dm.train_step(to_nhwc_padded(x[rank * per:(rank + 1) * per], torch.float32, "cpu"), y[rank * per:(rank + 1) * per])
    if rank == 0:
        torch.save(net.store.p.clone(), os.path.join(out_dir, "ddp.pt"))
    comm.shutdown()


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("native", 2), ("native", 4)])
def test_ddp_equals_single_process_full_batch(tmp_path, backend, world):
    """Without BatchNorm, W ranks x batch 8/W with gradient averaging == 1 process x batch 8 (the DDP contract)."""
    mp.spawn(_single_vs_ddp_worker, nprocs=world, args=(world, _free_port(), str(tmp_path), backend), join=True)
    sys.path.insert(0, ROOT)
    from distributedpytorch_amd.models.factory import get_model
    from distributedpytorch_amd.nn.loss import Criterion
    from distributedpytorch_amd.nn.optim import ArenaOptimizer
    from distributedpytorch_amd.parallel.ddp import DistributedModel
    from tests.helpers import to_nhwc_padded
    from tests.test_models_cpu import _zero_dropout
    torch.manual_seed(5)
    net, _ = get_model("basic", 10, device="cpu")
    _zero_dropout(net)
    dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "SGD", lr=1e-2))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 3, 28, 28, generator=g)
    y = torch.randint(0, 10, (8,), generator=g)
    dm.train_step(to_nhwc_padded(x, torch.float32, "cpu"), y)
    torch.testing.assert_close(net.store.p, torch.load(tmp_path / "ddp.pt"), rtol=1e-5, atol=1e-6)


def _run_cli(tmp_path, args, env_extra, timeout=600):
    env = dict(os.environ, DPA_MODEL_NAME="basic", DPA_GPUS="0,1", DPA_MASTER_PORT=str(_free_port()), DPA_NUM_THREADS="4",
               DPA_SYNTHETIC_DATA="1", PYTHONPATH=ROOT, **env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "main.py")] + args, cwd=tmp_path, env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.fixture(scope="module")
def tiny_data(tmp_path_factory):
    sys.path.insert(0, ROOT)
    from distributedpytorch_amd.data import mnist
    root = str(tmp_path_factory.mktemp("cli_data"))
    mnist.generate_synthetic_mnist(root, n_train=1000, n_test=200, seed=3)
    return root


def test_cli_train_resume_test_world2(tmp_path, tiny_data):
    """python main.py train / train -f (resume) / test with 2 spawned gloo ranks; seeded rerun is bit-identical."""
    r = _run_cli(tmp_path, ["train", "-d", tiny_data, "-e", "1", "-b", "50", "--debug"], {})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "world size: 2" in r.stdout and "Epoch: 001" in r.stdout and "model saved to" in r.stdout
    ck0 = tmp_path / "rsl" / "checkpoint-mnist-basic-000.pt.tar"
    assert ck0.exists() and (tmp_path / "rsl" / "bestmodel-mnist-basic.pt.tar").exists()
    first = torch.load(ck0)
    # step-level metrics from global rank 0: one JSON line per training step
    import json
    steps = [json.loads(ln) for ln in open(tmp_path / "rsl" / "steps.jsonl")]
    assert [s["step"] for s in steps] == [0, 1] and all(s["world_size"] == 2 and s["epoch"] == 0 for s in steps)
    for key in ("loss", "acc", "images_per_s", "step_ms", "data_ms", "comm_ms", "sync_ms", "fwd_ms", "bwd_ms", "wait_ms", "opt_ms"):
        assert key in steps[0], key
    assert steps[0]["loss"] > 0 and steps[0]["images_per_s"] > 0 and steps[0]["comm_ms"] >= steps[0]["sync_ms"]
    # seeded determinism: an identical second run writes an identical checkpoint
    run2 = tmp_path / "again"
    run2.mkdir()
    r2 = _run_cli(run2, ["train", "-d", tiny_data, "-e", "1", "-b", "50", "--debug"], {})
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    second = torch.load(run2 / "rsl" / "checkpoint-mnist-basic-000.pt.tar")
    for k, v in first["model_state_dict"].items():
        assert torch.equal(v, second["model_state_dict"][k]), k
    # resume: starts at epoch 2 and removes the stale per-epoch checkpoint (Q3, Q8)
    r3 = _run_cli(tmp_path, ["train", "-d", tiny_data, "-e", "2", "-b", "50", "--debug", "-f", str(ck0)], {})
    assert r3.returncode == 0, r3.stdout[-3000:] + r3.stderr[-3000:]
    assert "epoch:0001: model loaded from" in r3.stdout and "Epoch: 002" in r3.stdout and "Epoch: 001" not in r3.stdout
    assert (tmp_path / "rsl" / "checkpoint-mnist-basic-001.pt.tar").exists() and not ck0.exists()
    # test sub-command
    r4 = _run_cli(tmp_path, ["test", "-d", tiny_data, "-b", "100", "-f", str(tmp_path / "rsl" / "bestmodel-mnist-basic.pt.tar")], {})
    assert r4.returncode == 0, r4.stdout[-3000:] + r4.stderr[-3000:]
    assert "Acc:" in r4.stdout and "Time:" in r4.stdout


def test_cli_fault_injection_tears_down_all_ranks(tmp_path, tiny_data):
    """A rank that dies mid-training makes the launcher exit non-zero instead of hanging."""
    r = _run_cli(tmp_path, ["train", "-d", tiny_data, "-e", "1", "-b", "50", "--debug"],
                 {"DPA_FAULT_INJECT": "1:1", "DPA_COLLECTIVE_TIMEOUT_S": "30"}, timeout=300)
    assert r.returncode != 0
    assert "fault injection: rank 1" in r.stdout + r.stderr


def test_cli_two_nodes_two_launchers(tmp_path, tiny_data):
    """Multi-node process model of the reference (README: 'run the same command on every node of DDTNodes'): two
    launcher invocations emulate two hosts (127.0.0.1 and 127.0.0.2, one CPU rank each); ranks come from the node
    order, rendezvous goes through the C++ store and gloo, and only global rank 0 writes the checkpoint."""
    port = str(_free_port())
    common = dict(os.environ, DPA_MODEL_NAME="basic", DPA_DDT_NODES="127.0.0.1:0;127.0.0.2:0", DPA_MASTER_ADDR="127.0.0.1",
                  DPA_MASTER_PORT=port, DPA_NUM_THREADS="2", DPA_SYNTHETIC_DATA="1", PYTHONPATH=ROOT)
    procs = []
    for i, addr in enumerate(("127.0.0.1", "127.0.0.2")):
        wd = tmp_path / f"node{i}"
        wd.mkdir()
        env = dict(common, DPA_NODE_ADDR=addr)
        procs.append((wd, subprocess.Popen([sys.executable, os.path.join(ROOT, "main.py"), "train", "-d", tiny_data, "-e", "1",
                                            "-b", "50", "--debug"], cwd=wd, env=env, stdout=subprocess.PIPE,
                                           stderr=subprocess.STDOUT, text=True)))
    outs = []
    for wd, p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
        assert p.returncode == 0, out[-3000:]
    assert "overall rank: 0, world size: 2" in outs[0] and "overall rank: 1, world size: 2" in outs[1]
    assert (tmp_path / "node0" / "rsl" / "checkpoint-mnist-basic-000.pt.tar").exists()
    assert not (tmp_path / "node1" / "rsl" / "checkpoint-mnist-basic-000.pt.tar").exists()


@pytest.mark.parametrize("loss,opt", [("focal_loss", "SGD"), ("weighted_cross_entropy", "adam")])
def test_cli_other_losses_and_sgd_steplr(tmp_path, tiny_data, loss, opt):
    """The reference's dead branches made live (Q10): weighted CE / focal loss with dataset class weights, and
    SGD(momentum .9) + StepLR(1, .1) stepped once per epoch (classif.py:106-131,168-169)."""
    r = _run_cli(tmp_path, ["train", "-d", tiny_data, "-e", "2", "-b", "50", "--debug"], {"DPA_LOSS": loss, "DPA_OPTIMIZER": opt})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Epoch: 002" in r.stdout
    ck = torch.load(tmp_path / "rsl" / "checkpoint-mnist-basic-001.pt.tar")
    group = ck["optimizer_state_dict"]["param_groups"][0]
    if opt == "SGD":
        assert group["momentum"] == 0.9 and abs(group["lr"] - 1e-5) < 1e-12      # 1e-3 * 0.1 ** 2 after two epochs
    else:
        assert abs(group["lr"] - 1e-3) < 1e-12

"""bench.py's per-rank supervisor (native communicator first, torch process group as fallback, each in a FRESH child
process) with a stub in place of the GPU worker: when the native attempt fails on ONE rank every rank must converge on
the fallback quickly - not after burning its own rendezvous / collective timeouts - and exactly one JSON line may reach
stdout, from an attempt every rank finished."""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_STUB = r'''
import json, os, sys, time
plan = json.loads(os.environ["STUB_PLAN"])
attempt, rank = os.environ["DPA_BENCH_ATTEMPT"], os.environ.get("RANK", "0")
backend = sys.argv[sys.argv.index("--comm") + 1] if "--comm" in sys.argv else "?"
backend = [a for i, a in enumerate(sys.argv) if i and sys.argv[i - 1] == "--comm"][-1]      # the supervisor's own --comm is last
act = plan.get(f"{attempt}:{rank}", plan.get(f"{attempt}:*", "ok"))
if act.startswith("fail"):
    time.sleep(float(act.split(":")[2]) if act.count(":") > 1 else 0.2)
    sys.exit(int(act.split(":")[1]))
if act == "hang":
    time.sleep(300)
time.sleep(0.5)
if rank == "0":
    print(json.dumps({"value": 1.0, "backend": backend, "attempt": int(attempt)}), flush=True)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(tmp_path, world, plan, extra_env=None, limit_s=20):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(_STUB)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DPA_BENCH_STUB_WORKER=str(stub), STUB_PLAN=json.dumps(plan),
                   DPA_BENCH_ATTEMPT_S=str(limit_s), DPA_BENCH_SUPERVISOR_WAIT_S="30", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.time()
    outs = [p.communicate(timeout=120) for p in procs]
    return [p.returncode for p in procs], outs, time.time() - t0


def test_all_ranks_converge_on_the_fallback_when_one_native_worker_dies(tmp_path):
    # attempt 0 (native): rank 1 dies at once, ranks 0 and 2 would hang for ever (a collective with a dead peer)
    plan = {"0:1": "fail:17", "0:0": "hang", "0:2": "hang"}
    rcs, outs, took = _run(tmp_path, 3, plan, limit_s=60)
    assert rcs == [0, 0, 0], (rcs, [o[1][-800:] for o in outs])
    # not the 60 s attempt limit, let alone the 300 s hang (a few seconds on an idle machine; the bound leaves room for six
    # interpreter starts with `import torch` on a machine that is busy building)
    assert took < 45, took
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0]                           # ONE JSON line, from the attempt everyone finished
    d = json.loads(lines[0])
    assert d["backend"] == "torch" and d["attempt"] == 1
    assert outs[1][0].strip() == "" and outs[2][0].strip() == ""
    assert "another rank's attempt failed" in outs[0][1] and "exit 17" in outs[1][1]


def test_a_late_failure_on_another_rank_withholds_rank_zeros_number(tmp_path):
    # rank 0's native worker finishes and prints; rank 1's fails afterwards (e.g. at shutdown): that number must not be
    # reported - the fallback's is
    plan = {"0:1": "fail:3:2.0"}
    rcs, outs, took = _run(tmp_path, 2, plan)
    assert rcs == [0, 0], (rcs, [o[1][-800:] for o in outs])
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["backend"] == "torch", outs[0][0]


def test_native_success_prints_once_and_never_starts_the_fallback(tmp_path):
    rcs, outs, took = _run(tmp_path, 2, {})
    assert rcs == [0, 0]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"value": 1.0, "backend": "native", "attempt": 0}
    assert "re-running" not in outs[0][1] + outs[1][1]


def test_both_attempts_failing_exits_non_zero_without_a_number(tmp_path):
    rcs, outs, took = _run(tmp_path, 2, {"0:*": "fail:5", "1:*": "fail:6"})
    assert all(rc != 0 for rc in rcs), rcs
    assert not any(ln.startswith("{") for ln in outs[0][0].splitlines())


def test_attempt_time_limit_kills_a_hung_job(tmp_path):
    rcs, outs, took = _run(tmp_path, 2, {"0:*": "hang"}, limit_s=3)
    assert rcs == [0, 0], (rcs, [o[1][-800:] for o in outs])
    assert took < 40
    assert json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])["backend"] == "torch"

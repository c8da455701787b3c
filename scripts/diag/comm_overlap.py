#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (rocpd sqlite) of one DDP rank: where the communicator's kernels sit relative to the
backward pass.  Prints, for the last training step in the trace, every communicator kernel with the compute kernels that
ran WHILE it ran, and the share of communicator time that was overlapped by compute.
usage: python3 scripts/diag/comm_overlap.py <results.db>"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
sel = f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
rows = [(re.sub(r"^void ", "", re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))), s, e, q) for n, s, e, q in cur.execute(sel)]
is_comm = lambda n: "mesh_" in n or "nccl" in n.lower() or "rccl" in n.lower()      # noqa: E731
comm = [r for r in rows if is_comm(r[0])]
if not comm:
    print("no communicator kernels in this trace")
    sys.exit(1)
# steps are delimited by the optimizer kernels: take the window between the last two bursts of adam_step kernels
adam = [r for r in rows if "adam" in r[0]]
bursts = []
for r in adam:
    if not bursts or r[1] - bursts[-1][1] > 5e6:          # > 5 ms apart: a new step
        bursts.append([r[1], r[2]])
    else:
        bursts[-1][1] = r[2]
lo, hi = (bursts[-2][1], bursts[-1][1]) if len(bursts) >= 2 else (rows[0][1], rows[-1][2])
step = [r for r in rows if lo <= r[1] <= hi]
comp = [r for r in step if not is_comm(r[0])]
t0 = step[0][1]
print(f"# kernel trace: {len(rows)} kernels, {len(comm)} communicator kernels; last step = {len(step)} kernels, {(hi - lo) / 1e6:.2f} ms")
print(f"# queue column: {qcol}; queues used by compute kernels: {sorted({r[3] for r in comp})}, by communicator kernels: "
      f"{sorted({r[3] for r in step if is_comm(r[0])})}")
first_bwd = next((r for r in step if "dgrad" in r[0] or "wgrad" in r[0] or "bn_bwd" in r[0]), None)
last_bwd = next((r for r in reversed(step) if "wgrad" in r[0] or "bn_bwd" in r[0]), None)
if first_bwd and last_bwd:
    print(f"# backward pass: {(first_bwd[1] - t0) / 1e6:.2f} .. {(last_bwd[2] - t0) / 1e6:.2f} ms after the step's first kernel")
tot = ov = 0
print("# comm kernel                     queue  start_ms   dur_us  overlapped_us  compute kernels running meanwhile")
for n, s, e, q in [r for r in step if is_comm(r[0])]:
    inside = [(c[0], max(s, c[1]), min(e, c[2])) for c in comp if c[1] < e and c[2] > s]
    o = sum(b - a for _, a, b in inside)
    o = min(o, e - s)
    tot += e - s
    ov += o
    names = sorted({re.sub(r"<.*", "", c[0]) for c in inside})
    print(f"{n[:32]:32s} {str(q):>6s} {(s - t0) / 1e6:9.3f} {(e - s) / 1e3:8.1f} {o / 1e3:14.1f}  {', '.join(names)[:120]}")
print(f"# communicator kernel time in the step: {tot / 1e3:.1f} us, of which {ov / 1e3:.1f} us ({100.0 * ov / max(tot, 1):.1f} %) ran while "
      f"compute kernels were running")

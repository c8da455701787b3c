#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg us, % of GPU time) from a rocprofv3 rocpd sqlite database -> CSV on stdout.
usage: python scripts/rocpd_stats.py gpurun_out/prof/x_results.db [skip_first_fraction]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    if len(name) > 110:
        name = name[:107] + "..."
    a = agg.setdefault(name, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(a[1] for a in agg.values())
print("kernel,calls,total_us,avg_us,pct")
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"\"{name}\",{n},{t / 1e3:.1f},{t / 1e3 / n:.2f},{100.0 * t / tot:.2f}")
print(f"\"TOTAL\",{sum(a[0] for a in agg.values())},{tot / 1e3:.1f},,100.00")

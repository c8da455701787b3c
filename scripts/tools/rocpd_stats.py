#!/usr/bin/env python3
"""Per-kernel statistics (the table `rocprofv3 --stats` prints) out of a rocpd SQLite database, for rocprofv3 builds
whose default output is `*_results.db`:  python scripts/tools/rocpd_stats.py results.db [steps] > kernel_stats.csv
`steps` divides the totals so the last column reads ms per training step."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name if len(name) < 140 else name[:137] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col} "
                      "order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print("kernel,calls,total_ms,avg_us,min_us,max_us,percent,ms_per_step")
    for n, c, t, lo, hi in rows:
        print(f"\"{short(n)}\",{c},{t / 1e6:.3f},{t / c / 1e3:.1f},{lo / 1e3:.1f},{hi / 1e3:.1f},{100 * t / total:.2f},{t / 1e6 / steps:.3f}")
    print(f"\"TOTAL\",{sum(r[1] for r in rows)},{total / 1e6:.3f},,,,100.00,{total / 1e6 / steps:.3f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration, share) from a rocprofv3 rocpd sqlite database
(what `rocprofv3 --kernel-trace --stats` writes on ROCm 7.2).  Usage: rocpd_stats.py results.db [last_ms] [> summary.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name if len(name) < 150 else name[:147] + "..."


def main(path, last_ms=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    where = ""
    if last_ms is not None:      # steady state only: kernels that started within the last `last_ms` milliseconds of the trace
        tmax = db.execute("select max(end) from kernels").fetchone()[0]
        where = f"where start >= {int(tmax - last_ms * 1e6)}"
    rows = db.execute(f"select {namecol}, count(*), sum(end - start), min(end - start), max(end - start) from kernels {where} group by {namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_ms,avg_us,min_us,max_us,percent")
    for n, c, t, mn, mx in rows:
        print(f"\"{short(n)}\",{c},{t / 1e6:.3f},{t / c / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * t / total:.2f}")
    print(f"\"TOTAL\",{sum(r[1] for r in rows)},{total / 1e6:.3f},,,,100.00")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else None)

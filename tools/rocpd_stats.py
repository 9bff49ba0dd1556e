#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (the default output format of rocprofv3 --kernel-trace on ROCm 7.2):
    python3 tools/rocpd_stats.py <results.db> [max rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc"))
tot = sum(r[5] for r in rows) or 1
print("%-100s %6s %10s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-100s %6d %10.1f %10.1f %10.1f %6.1f" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot))

#!/usr/bin/env python3
"""Kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from the rocpd sqlite file rocprofv3 writes when no
--output-format is given: tools/rocpd_stats.py <results.db> -> name, calls, total / average / min / max duration."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else cols[0])
dur = "duration" if "duration" in cols else '("end" - start)'
rows = db.execute(f"select {name}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels group by {name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"# {sys.argv[1]}: kernel dispatches grouped by name (ns)")
print(f"{'kernel':60s} {'calls':>7s} {'total_ns':>15s} {'avg_ns':>14s} {'min_ns':>12s} {'max_ns':>12s} {'%':>6s}")
for r in rows:
    print(f"{str(r[0])[:60]:60s} {r[1]:7d} {r[2]:15d} {r[3]:14.1f} {r[4]:12d} {r[5]:12d} {100.0 * r[2] / tot:6.2f}")

#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd .db (kernel-trace --stats) into a small text table for profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    lines = ["# rocprofv3 --kernel-trace --stats (durations in microseconds)", f"# source: {db}",
             f"{'kernel':60s} {'calls':>8s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}"]
    for n, calls, tot, avg, pct in rows:
        lines.append(f"{n[:60]:60s} {calls:8d} {tot:14.3f} {avg:12.3f} {pct:7.2f}")
    try:
        pm = list(c.execute("select * from counters_collection limit 0"))
    except Exception:
        pm = None
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])

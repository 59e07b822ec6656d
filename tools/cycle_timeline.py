#!/usr/bin/env python3
"""One steady cycle of the pending loop as a timeline: every kernel of the cycle with its start offset, duration and the gap to the
previous kernel's end (rocprofv3 --kernel-trace --output-format csv). Usage: cycle_timeline.py <dir> [anchor-kernel] [cycle-index]"""
import csv, glob, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_pend_pop"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(anchor)]
a, b = starts[which], starts[which + 1]
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
busy = 0
print(f"# {f}: cycle {which} ({b - a} launches)")
print(f"{'kernel':44s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:44]:44s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:8.1f}")
    busy += e - s; prev = max(prev, e)
total = int(rows[b]["Start_Timestamp"]) - t0
print(f"cycle {total / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(total - busy) / 1e3:.1f} us")

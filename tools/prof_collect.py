#!/usr/bin/env python
"""Turn the rocprofv3 CSV output of tools/prof_all.sh (gpurun_out/p_<workload>_{stats,fetch,write}) into the files kept under
profiles/: <tag>_<workload>_rocprof_summary.txt, <tag>_<workload>_kernel_stats.csv and pmc_traffic_<workload>.json
(the per-launch HBM traffic bench.py reports in roofline.traffic).

usage: python tools/prof_collect.py <tag> <workload> [<kernel> ...]      e.g.  r01k cfg3 k_nominate k_process
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)  # gpurun merges into existing directories: newest wins
    if not hits:
        raise SystemExit(f"nothing matches {pattern}")
    return hits[-1]


def short(name):
    return name.split("(")[0].strip('"')


def pmc(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            a = acc[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def main(tag, workload, kernels):
    out = os.path.join(ROOT, "profiles")
    base = os.environ.get("KQ_PROF_BASE", os.path.join(ROOT, "gpurun_out"))
    stats = one(f"{base}/p_{workload}_stats/**/*_kernel_stats.csv")
    shutil.copy(stats, os.path.join(out, f"{tag}_{workload}_kernel_stats.csv"))
    fetch = pmc(one(f"{base}/p_{workload}_fetch/**/*_counter_collection.csv"))
    write = pmc(one(f"{base}/p_{workload}_write/**/*_counter_collection.csv"))
    cmd = {"cfg3": "python bench.py --no-cpu-baseline --no-parity-gate --full-run 0 --no-host-leg   (the default bench: workload cfg3, pending loop, 100 steps + 5 warm-up, 1 x MI355X)",
           "cfg3-batch": "python bench.py --workload cfg3-batch --steps 10 --warmup 1 --no-cpu-baseline   (nominate-all-pending, 100 000 heads per launch, 1 x MI355X)",
           "cfg3f": "python bench.py --workload cfg3f --steps 20 --no-cpu-baseline --no-parity-gate --full-run 0 --no-host-leg   (fair sharing, pending loop, 1 x MI355X)",
           "cfg4c": "python bench.py --workload cfg4c --start feasible --steps 24 --warmup 4 --series-cycles 0 --no-cpu-baseline --no-parity-gate   (classical preemption as the CLOSED loop, feasible start, 1000 heads per cycle, 28 cycles, 1 x MI355X; round 6 - until round 5: one open-loop cycle)",
           "cfg4f": "python bench.py --workload cfg4f --start feasible --steps 12 --warmup 4 --series-cycles 0 --no-cpu-baseline --no-parity-gate   (fair sharing + fair preemption as the CLOSED loop, feasible start, 1000 heads per cycle, 16 cycles, 1 x MI355X; round 6 - until round 5: one open-loop cycle)",
           "cfg2": "python bench.py --workload cfg2 --no-cpu-baseline --no-parity-gate --full-run 0 --no-host-leg   (128 ClusterQueues, 110 heads per cycle, pending loop, 1 x MI355X)",
           "cfg5": "python bench.py --workload cfg5 --steps 5 --warmup 1 --no-cpu-baseline   (1 x MI355X)",
           "cfg5-cycle": "python bench.py --workload cfg5-cycle --steps 5 --warmup 1 --no-cpu-baseline --no-parity-gate   (TAS inside the cycle, 1000 heads per cycle, 1 x MI355X)"}.get(workload, workload)
    lines = [f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}"]
    lines += [l.rstrip("\n") for l in open(stats)]
    lines += ["", "# PMC passes (separate runs, --kernel-trace --pmc <counter>, same command): mean per launch, unit = KB as reported by rocprofv3"]
    for cname, tab in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        for k in sorted(tab):
            n, mean = tab[k]
            lines.append(f"{cname:11s} {k:30s} launches {n:4d}  mean_KB {mean:14.3f}")
    lines += ["", "# gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE halves 16 B/lane streaming reads; these kernels read 8-byte cells and",
              "# int32 state, an access width the guide calls uncalibrated, so raw values are kept. Infinity-Cache hits are counted.",
              "# traffic = FETCH + WRITE per launch."]
    traffic = {}
    for k in kernels:
        f = fetch.get(k, (0, 0.0))[1] * 1024
        w = write.get(k, (0, 0.0))[1] * 1024
        traffic[k] = f + w
        lines.append(f"traffic_bytes_per_launch {k}: fetch {f:.0f} + write {w:.0f} = {f + w:.0f}")
    summary = os.path.join(out, f"{tag}_{workload}_rocprof_summary.txt")
    open(summary, "w").write("\n".join(lines) + "\n")
    json.dump({"workload": workload, "source": os.path.relpath(summary, ROOT), "traffic_bytes_per_launch": traffic},
              open(os.path.join(out, f"pmc_traffic_{workload}.json"), "w"), indent=1)
    print("\n".join(lines[:12]))
    print("...")
    print("\n".join(lines[-len(kernels):]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])

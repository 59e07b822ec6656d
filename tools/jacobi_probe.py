"""Dependency depth of processEntry on the CPU oracle: oracle/kq_oracle.cpp kqo_jacobi_probe.
usage: python tools/jacobi_probe.py cfg4c|cfg4f|cfg3p [n_cq] [max_rounds]      (writes a text report to stdout)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
from kueue_amd.api import make_config
from kueue_amd.population import generate
from oracle import kqo

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4c"
ncq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
max_rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 40
fair = name.endswith("f")
pop = generate(4, n_cq=ncq, fair_sharing=fair)
cfg = make_config(fair_sharing=fair)
heads = pop.heads_for_cycle(0, cycle=1)
n = heads.n
per = np.full((n, 12), -1, np.int32)
rounds = np.zeros((max_rounds, 3), np.int32)
nr, conv = C.c_int32(0), C.c_int32(0)
l = kqo.lib()
l.kqo_jacobi_probe.restype = C.c_int
t0 = time.perf_counter()
rc = l.kqo_jacobi_probe(C.byref(cfg), C.byref(pop.snapshot.struct()), C.byref(heads.struct()), max_rounds, F.ptr(per), F.ptr(rounds), C.byref(nr), C.byref(conv))
dt = time.perf_counter() - t0
print(f"# {name}: {ncq} ClusterQueues, {n} heads, fair_sharing={fair}; kqo_jacobi_probe rc={rc}, {dt:.1f} s of one core")
o = np.argsort(per[:, 0])
pe = per[o]
pe = pe[pe[:, 0] >= 0]
rec = pe[:, 2] == 1
print(f"entries processed {len(pe)}; with nominated targets {(pe[:, 1] > 0).sum()}; overlap recomputations {rec.sum()}; final: preempting {(pe[:, 5] == 2).sum()}, admitted {(pe[:, 5] == 1).sum()}")
print(f"nominated targets per entry: mean {pe[:, 1].mean():.1f} max {pe[:, 1].max()};  final targets per preempting entry: mean {pe[pe[:, 5] == 2, 3].mean() if (pe[:, 5] == 2).any() else 0:.1f}")
if rec.any():
    print(f"per recomputation: searches mean {pe[rec, 6].mean():.2f}, candidates listed per search {pe[rec, 7].sum() / max(1, pe[rec, 6].sum()):.0f}, rows removed before the fill-back per search {pe[rec, 8].sum() / max(1, pe[rec, 6].sum()):.1f}")
    d = pe[rec, 10]
    print(f"recomputation against (cycle-start usage, true preempted set) gives a different outcome for {(d == 1).sum()} of {len(d)} overlapping entries")
print(f"Jacobi rounds run {nr.value}, converged={conv.value}")
print("round  final_prefix  changed_vs_prev_round  first_changed_pos")
for r in range(nr.value):
    print(f"{r:5d}  {rounds[r, 0]:12d}  {rounds[r, 1]:21d}  {rounds[r, 2]:17d}")
if not conv.value:
    print(f"positions of the sequential result reached by the last round: {per[0, 11]}")
last = pe[:, 9]
print("round in which an entry's outcome last changed (histogram):", dict(zip(*[x.tolist() for x in np.unique(last, return_counts=True)])))

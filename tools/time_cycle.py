"""Wall time of every ABI call of the pending loop (bench.py's step), per cycle.  usage: python tools/time_cycle.py [cfg] [cycles]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, os.environ.get("KQ_LIB", "libkq_engine.so"))
from kueue_amd.engine import Engine
from kueue_amd.api import make_config, Decisions
from kueue_amd.population import generate
cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pop = generate(cfgn); snap = pop.snapshot
eng = Engine(make_config()); eng.put(snap); eng.pending_put(pop.pending())
lib, h = eng._lib, eng._h
out = Decisions(pop.heads_for_cycle(0), tgt_cap=4096, n=snap.n_cq, n_ps=int(pop.w_nps.max()) * snap.n_cq)
n = C.c_int32(); nps = C.c_int32()
names = ["pending_heads", "cycle_run_pending", "cycle_commit", "pending_apply", "cycle_release"]
acc = np.zeros(5); live = 0
ost = C.byref(out.struct())
for c in range(1, 21 + ncyc):
    t = [time.perf_counter()]
    lib.kq_pending_heads(h, c, None, C.byref(n), C.byref(nps), None); t.append(time.perf_counter())
    lib.kq_cycle_run_pending(h, ost); t.append(time.perf_counter())
    lib.kq_cycle_commit(h, None); t.append(time.perf_counter())
    lib.kq_pending_apply(h); t.append(time.perf_counter())
    live += 1
    if live > 4:
        lib.kq_cycle_release(h, 5); live -= 1
    t.append(time.perf_counter())
    if c > 20: acc += np.diff(t)
for nm, v in zip(names, acc / ncyc * 1e6):
    print(f"{nm:20s} {v:8.1f} us")
print(f"{'cycle':20s} {acc.sum() / ncyc * 1e6:8.1f} us")

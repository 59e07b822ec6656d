"""Average per-cycle kernel intervals of the cfg pending loop with a given engine library (KQ_LIB=libkq_engine_<x>.so).
usage: KQ_LIB=... python tools/time_spec.py [cfg] [cycles]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, os.environ.get("KQ_LIB", "libkq_engine.so"))
from kueue_amd.engine import Engine
from kueue_amd.api import make_config, Decisions
from kueue_amd.population import generate
cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 60
pop = generate(cfgn); snap = pop.snapshot
eng = Engine(make_config()); eng.put(snap); eng.pending_put(pop.pending())
lib = eng._lib
out = Decisions(pop.heads_for_cycle(0), tgt_cap=4096, n=snap.n_cq, n_ps=int(pop.w_nps.max()) * snap.n_cq)
live = 0
ph = np.zeros(3); by = np.zeros(2, np.int64); acc = np.zeros(3)
for c in range(1, 11 + ncyc):
    eng.pending_heads(c); eng.run_pending(out)
    lib.kq_last_cycle_phases(eng._h, F.ptr(ph), F.ptr(by))
    if c > 10: acc += ph
    eng.commit(); eng.pending_apply(); live += 1
    if live > 4: eng.release(5); live -= 1
print(os.environ.get("KQ_LIB", "libkq_engine.so"), "nominate/order/process ms per cycle:", (acc / ncyc).round(4).tolist())

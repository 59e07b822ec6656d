"""In-kernel section timing of k_process_spec (needs kueue_amd/libkq_engine_specprof.so: the spec TU built with -DKQ_SPEC_PROF).
usage: python tools/prof_spec.py [cfg] [cycles]"""
import ctypes as C, sys, os
os.environ["KQ_SPEC_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, os.environ.get("KQ_LIB", "libkq_engine_specprof.so"))
from kueue_amd.engine import Engine
from kueue_amd.api import make_config, Decisions
from kueue_amd.population import generate
cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pop = generate(cfgn)
snap = pop.snapshot
eng = Engine(make_config()); eng.put(snap); eng.pending_put(pop.pending())
lib = eng._lib
lib.kq_debug_prof.argtypes = [C.c_void_p, F.i64p, C.c_int]
prof = np.zeros(64, np.int64)
out = Decisions(pop.heads_for_cycle(0), tgt_cap=4096, n=snap.n_cq, n_ps=int(pop.w_nps.max()) * snap.n_cq)
live = 0
stats = np.zeros(8, np.int64)
def step(c):
    global live
    n, nps, hw = eng.pending_heads(c)
    eng.run_pending(out)
    st = eng.spec_stats()
    eng.commit(); eng.pending_apply(); live += 1
    if live > 4:
        eng.release(5); live -= 1
    return n, st
for c in range(1, 11):
    step(c)
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
heads = 0
for c in range(11, 11 + ncyc):
    n, st = step(c); heads += n; stats += st
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
names = {1: "window building", 2: "entry state + item constants", 3: "arrangements (radix sorts)", 4: "rounds: init / decide / barriers", 5: "rounds: push", 6: "rounds: scan", 7: "rounds: pull", 8: "results + CQ cells"}
tot = 0
for i, nm in names.items():
    us = prof[32 + i] / 100.0 / ncyc
    tot += us
    print(f"{nm:34s} {us:8.1f} us per cycle")
print(f"{'sum':34s} {tot:8.1f} us per cycle")
print(f"per cycle: windows {stats[0]/ncyc:.2f}, rounds {stats[1]/ncyc:.2f}, entries decided {stats[2]/ncyc:.1f} of {heads/ncyc:.1f} heads, items {stats[4]/ncyc:.0f}, max rounds {stats[5]}, handed back {stats[3]}, truncated {stats[7]}")

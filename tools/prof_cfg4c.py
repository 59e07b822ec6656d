"""In-kernel segment timing of one classical-preemption cycle (cfg 4c), needs libkq_engine_prof.so (-DKQ_PROF, tools/build_prof.sh).
usage: python tools/prof_cfg4c.py [n_cq] [cycles]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, "libkq_engine_prof.so")
from kueue_amd.engine import Engine
from kueue_amd.api import make_config
from kueue_amd.population import generate
ncq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ncy = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pop = generate(4, n_cq=ncq)
eng = Engine(make_config()); eng.put(pop.snapshot)
lib = eng._lib
lib.kq_debug_prof.argtypes = [C.c_void_p, F.i64p, C.c_int]
prof = np.zeros(64, np.int64)
eng.run(pop.heads_for_cycle(0))   # warm-up
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
n = 0
t = time.time()
for c in range(ncy):
    h = pop.heads_for_cycle(c); d = eng.run(h); n += h.n
dt = time.time() - t
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
names = {8: "pc_load", 9: "pc_flush", 10: "chunk_prefetch", 11: "chunk serial core (incl. generic path)", 12: "chunk write results", 13: "leader waits for helpers", 14: "leader: whole tree", 27: "re-prefetch after a stopped chunk",
         0: "generic: load_head", 1: "generic: use list", 34: "generic: first fits (entry_fits with targets)", 4: "generic: has_any .. before recompute", 5: "recompute: row flush",
         6: "recompute: get_assignments", 7: "recompute: row reload", 32: "recompute: fits after", 33: "generic: tail (has_any, insert targets, add usage)", 2: "fast entry", 3: "fast entry: stat",
         35: "scan search: private usage + tables", 36: "scan search: classification + time order", 37: "scan search: alive + level passes (prefix)",
         38: "scan search: first fit", 39: "scan search: finalise + targets", 63: "scan search: fill-back",
         40: "search(fair): private plane copy", 41: "search(fair): sums + clears", 42: "search(fair): findCandidates", 43: "search(fair): first strategy",
         21: "nominate heads Fit (sum cycles)", 22: "nominate heads Preempt (sum cycles)", 23: "nominate heads NoFit (sum cycles)", 24: "n Fit", 25: "n Preempt", 26: "n NoFit", 30: "slowest head"}
print(f"cfg4c n_cq {ncq} cycles {ncy} heads {n} wall {dt:.3f}s last kernel_ms {d.kernel_ms}")
for i, nm in names.items():
    print(f"{nm:52s} {prof[i]:16d} cycles  {prof[i] / 2.4e6 / ncy:12.2f} ms per cycle at 2.4 GHz")
print("all 64 counters:", [int(x) for x in prof])

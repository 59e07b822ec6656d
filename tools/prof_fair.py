"""In-kernel segment timing of one fair-sharing preemption cycle (cfg 4f), needs libkq_engine_prof.so (-DKQ_PROF).
usage: python tools/prof_fair.py [n_cq] [heads]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, "libkq_engine_prof.so")
from kueue_amd.engine import Engine
from kueue_amd.api import make_config
from kueue_amd.population import generate
ncq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pop = generate(4, n_cq=ncq, fair_sharing=True)
eng = Engine(make_config(fair_sharing=True)); eng.put(pop.snapshot)
lib = eng._lib
lib.kq_debug_prof.argtypes = [C.c_void_p, F.i64p, C.c_int]
prof = np.zeros(64, np.int64)
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
h = pop.heads_for_cycle(0, limit=nh)
t = time.time(); d = eng.run(h); dt = time.time() - t
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
names = {0: "generic: load_head", 1: "generic: use list", 34: "generic: first fits", 4: "generic: has_any .. before recompute", 5: "recompute: row flush", 6: "recompute: get_assignments",
         7: "recompute: row reload", 32: "recompute: fits after", 33: "generic: tail (insert targets, add usage)", 2: "fast entry", 3: "fast entry: stat",
         16: "fair: computeDRS (leader wave)", 17: "fair: barrier wait", 18: "fair: tournament", 19: "fair: pop bookkeeping", 20: "fair: processEntry",
         40: "search: private plane copy", 41: "search: sums + clears", 42: "search: findCandidates", 43: "search: first strategy", 44: "search: second strategy",
         45: "search: restore (no fit)", 46: "search: fillBack", 47: "  lds search: ordering.next", 48: "  lds search: pop", 49: "  lds search: row load + context", 50: "  lds search: share after removal (virtual)",
         51: "  lds search: RemoveWorkload commit", 52: "  lds search: push target", 53: "  lds search: fits", 54: "  lds search: LCAs + shares of both sides",
         55: "    virtual apply: read row", 56: "    virtual apply: chains", 57: "    virtual apply: fence", 58: "    virtual apply: node update",
         21: "nominate heads Fit (sum cycles)", 22: "nominate heads Preempt (sum cycles)", 23: "nominate heads NoFit (sum cycles)", 24: "n Fit", 25: "n Preempt", 26: "n NoFit", 30: "slowest head"}
print(f"n_cq {ncq} heads {h.n} wall {dt:.3f}s kernel_ms {d.kernel_ms}")
for i, nm in names.items():
    print(f"{nm:44s} {prof[i]:16d} cycles  {prof[i] / 2.4e6:12.2f} ms at 2.4 GHz")

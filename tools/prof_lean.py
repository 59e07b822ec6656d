"""In-kernel segment timing of k_nominate_lean (needs kueue_amd/libkq_engine_prof.so: tools/build_prof.sh, -DKQ_PROF)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, "libkq_engine_prof.so")
from kueue_amd.engine import Engine
from kueue_amd.api import make_config
from kueue_amd.population import generate
pop = generate(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
eng = Engine(make_config()); eng.put(pop.snapshot)
lib = eng._lib
lib.kq_debug_prof.argtypes = [C.c_void_p, F.i64p, C.c_int]
prof = np.zeros(64, np.int64)
for c in range(3):
    eng.run(pop.heads_for_cycle(c))
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
n = 0
for c in range(3, 13):
    h = pop.heads_for_cycle(c); d = eng.run(h); n += h.n
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
names = {56: "load_head", 57: "requests in iterator order + clear", 58: "cells (fitsResourceQuota)", 59: "serial choice among flavors", 60: "usage list / outputs",
         61: "assign_flavors: rest", 62: "publish (nominate_finish)"}
tot = sum(prof[i] for i in names)
for i, nm in names.items():
    print(f"{nm:40s} {prof[i]/n:10.1f} cycles/head  {prof[i]/max(tot,1)*100:5.1f} %   ({prof[i]/n/2400:.2f} us at 2.4 GHz)")
print(f"sum {tot/n:.0f} cycles/head = {tot/n/2400:.1f} us; kernel ms last cycle {d.kernel_ms}")

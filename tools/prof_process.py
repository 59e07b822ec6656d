"""In-kernel segment timing of k_process (needs libkq_engine_prof.so built with -DKQ_PROF)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, "libkq_engine_prof.so")
from kueue_amd.engine import Engine
from kueue_amd.api import make_config
from kueue_amd.population import generate
fair = len(sys.argv) > 2 and sys.argv[2] == "fair"
pop = generate(int(sys.argv[1]) if len(sys.argv) > 1 else 3, fair_sharing=fair)
eng = Engine(make_config(fair_sharing=fair)); eng.put(pop.snapshot)
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
names = {8: "pc_load", 9: "pc_flush", 10: "chunk_prefetch", 11: "chunk serial core",  12: "chunk write results", 13: "leader waits for helpers", 14: "leader: whole tree", 27: "re-prefetch after a stopped chunk",
         0: "slow: load_head", 1: "slow: use list", 4: "recompute: before (fits, np)", 5: "recompute: row flush", 6: "recompute: get_assignments", 7: "recompute: row reload",
         15: "fair: pops that changed usage (COUNT, not cycles)", 16: "fair: computeDRS (leader wave)", 17: "fair: barrier wait", 18: "fair: tournament", 19: "fair: pop bookkeeping", 20: "fair: processEntry"}
for i, nm in names.items():
    print(f"{nm:28s} {prof[i]/n:10.1f} cycles/entry   total {prof[i]}")
for i, nm in enumerate(("Fit", "Preempt", "NoFit")):
    c = max(1, prof[24 + i])
    print(f"k_nominate heads with mode {nm:8s}: {prof[24 + i] / 10:7.1f} per cycle, {prof[21 + i] / c:10.0f} cycles each ({prof[21 + i] / c / 2400:.1f} us)")
print(f"k_nominate slowest head of the measured cycles: {prof[30]} cycles ({prof[30] / 2400:.1f} us)")
print(f"clock64 / wall_clock64 = {prof[14]/max(1,prof[29]):.2f}  (wall clock is 100 MHz => core clock {prof[14]/max(1,prof[29])*0.1:.2f} GHz); tree wall time {prof[29]/10/100:.1f} us per cycle")
print("kernel ms last cycle", d.kernel_ms)

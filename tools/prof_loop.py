"""In-kernel segment timing of the CLOSED preemption loop (kueue_amd/closed_loop.py), cycle by cycle; needs libkq_engine_prof.so (-DKQ_PROF, tools/build_prof.sh).
usage: python tools/prof_loop.py cfg4c|cfg4f spec|feasible <cycles> [n_cq]      -> one block per cycle whose wall time exceeds 3x the median, plus the totals"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, "libkq_engine_prof.so")
from kueue_amd.engine import Engine
from kueue_amd.api import make_config
from kueue_amd.closed_loop import PreemptionLoop
from kueue_amd.population import generate
fair = sys.argv[1] == "cfg4f"; feas = sys.argv[2] == "feasible"; ncy = int(sys.argv[3]); ncq = int(sys.argv[4]) if len(sys.argv) > 4 else None
pop = generate(4, n_cq=ncq, fair_sharing=fair, feasible=feas)
snap, pending = pop.snapshot, pop.pending()
eng = Engine(make_config(fair_sharing=fair)); eng.put(snap); eng.pending_put(pending)
loop = PreemptionLoop(eng, snap, pending, hold=1 << 40, tgt_cap=max(4096, (32 if fair else 4) * snap.n_adm))
lib = eng._lib
lib.kq_debug_prof.argtypes = [C.c_void_p, F.i64p, C.c_int]
names = {14: "leader: whole tree", 11: "chunk serial core (incl. generic path)", 34: "generic: first fits (entry_fits with targets)", 6: "recompute: get_assignments", 32: "recompute: fits after",
         33: "generic: tail (has_any, insert targets, add usage)", 0: "generic: load_head", 1: "generic: use list", 4: "generic: has_any .. before recompute", 2: "fast entry",
         35: "scan search: private usage + tables", 36: "scan search: classification + time order", 37: "scan search: alive + level passes (prefix)", 38: "scan search: first fit",
         39: "scan search: finalise + targets", 63: "scan search: fill-back", 40: "search(fair): private plane copy", 41: "search(fair): sums + clears", 42: "search(fair): findCandidates",
         43: "search(fair): first strategy", 44: "search(fair): second strategy", 45: "search(fair): restore (no fit)", 46: "search(fair): fillBack",
         47: "  lds search: ordering.next", 48: "  lds search: pop / batch", 49: "  lds search: row load + context", 50: "  lds search: share after removal (virtual)",
         51: "  lds search: RemoveWorkload commit", 53: "  lds search: fits", 54: "  lds search: LCAs + shares of both sides", 5: "recompute: row flush", 7: "recompute: row reload",
         16: "fair: computeDRS (leader wave)", 17: "fair: barrier wait", 18: "fair: tournament", 19: "fair: pop bookkeeping", 20: "fair: processEntry", 21: "nominate heads Fit (sum cycles)", 22: "nominate heads Preempt (sum cycles)", 24: "n Fit", 25: "n Preempt", 30: "slowest head"}
rows = []
prof = np.zeros(64, np.int64)
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
for c in range(1, ncy + 1):
    t = time.time(); d, ha, hw = loop.step(c); dt = (time.time() - t) * 1e3
    lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
    rows.append((c, dt, dict(loop.stats[-1]), prof.copy()))
med = float(np.median([r[1] for r in rows]))
print(f"{sys.argv[1]} {sys.argv[2]} n_cq {snap.n_cq}: {ncy} cycles, median {med:.2f} ms")
for c, dt, st, p in rows:
    print(f"cycle {c:3d} {dt:9.2f} ms  heads {st['heads']} admitted {st['admitted']} preempting {st['preempting']} targets {st['targets']} rows {st['rows']}")
for c, dt, st, p in rows:
    if dt > 3 * med:
        print(f"---- cycle {c}: {dt:.2f} ms ----")
        for i, nm in names.items():
            if p[i]:
                print(f"  {nm:52s} {p[i]:16d} cycles  {p[i] / 2.4e6:12.2f} ms at 2.4 GHz")

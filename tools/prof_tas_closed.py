"""In-kernel segment timing of k_process_tas over the CLOSED loop of bench.py --workload cfg5-cycle (kueue_amd/tas_population.py TASClosedLoop);
needs kueue_amd/libkq_engine_prof.so (tools/build_prof.sh, -DKQ_PROF)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
F.ENGINE_LIB = os.path.join(F.HERE, "libkq_engine_prof.so")
from kueue_amd.engine import Engine
from kueue_amd.api import make_config
from kueue_amd.tas_population import generate_tas_cycle
from oracle import kqo
snap, topos, batch = generate_tas_cycle(n_cq=1000, n_pending=50000)
cfg = make_config()
loop = batch.closed_loop(hold=4)
traj = []
for c in range(10):
    sn, h, ct = loop.cycle_input()
    want, wout = kqo.cycle_run_tas(cfg, sn, h, ct)
    traj.append((sn, h, ct)); loop.fold(h, want, wout)
eng = Engine(cfg)
lib = eng._lib
lib.kq_debug_prof.argtypes = [C.c_void_p, F.i64p, C.c_int]
prof = np.zeros(64, np.int64)
for sn, h, ct in traj[:4]:
    eng.put(sn); eng.run_tas(h, ct)
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
n = rec = finds = 0
for sn, h, ct in traj[4:]:
    eng.put(sn)
    d, _ = eng.run_tas(h, ct); n += h.n; rec += d.tas_stats["recomputes"]; finds += d.tas_stats["finds"]
    print(f"rows {sn.n_adm} finds {d.tas_stats['finds']} recomputes {d.tas_stats['recomputes']} kernel_ms {d.kernel_ms:.1f}")
lib.kq_debug_prof(eng._h, F.ptr(prof), 1)
names = {35: "position -> entry, tree switch", 36: "the entry's header (load_head + nomination, or the prefetched record)",
         38: "  scheduler.fits: quota half (both calls)", 39: "  scheduler.fits: leaf half (both calls)", 37: "  publish (inside 'publish + second fits')",
         40: "head + first fits", 41: "before the recomputation", 42: "recomputation (get_assignments)", 43: "publish + second fits", 44: "usage added, result written (admit path: the result only)", 62: "  admit path: preemptedWorkloads.Insert", 63: "  admit path: AddUsage on the quota planes", 61: "  admit path: leaf usage + class tables",
         57: "  recomputation: assign_flavors - requests of the podset, output rows cleared", 58: "  recomputation: assign_flavors - the (flavor, resource) cells", 59: "  recomputation: assign_flavors - the choice among the flavors", 60: "  recomputation: assign_flavors - usage list / outputs",
         48: "  recomputation: WorkloadsTopologyRequests", 49: "  recomputation: the find (request block + placement)", 47: "    request / argument block", 45: "    placement (t_workload)",
         46: "      phase 1 of the placement", 55: "      before the search (state of the class, parameters)", 56: "      t_find_assignment", 51: "        findLevelWithFitDomains",
         52: "        the fit level's own domains", 53: "        levels down to the slice level", 54: "          findLevel: the level's sweep", 58: "          findLevel: LeastFreeCapacity histogram threshold", 57: "      status + buildAssignment",
         50: "  recomputation: keeping the result"}
for i, nm in names.items():
    print(f"{nm:60s} {prof[i]/n:10.1f} cycles/entry  ({prof[i]/n/2400:.2f} us at 2.4 GHz)")
tot = sum(prof[i] for i in (35, 36, 40, 41, 42, 43, 37, 44, 62, 63, 61))
print(f"headers that came prefetched: {prof[34]} of {n}")
print(f"sum of the entry segments {tot/n:10.1f} cycles/entry  ({tot/n/2400:.2f} us at 2.4 GHz)")
print(f"{n} entries, {rec} recomputations, {finds} placements; kernel ms last cycle {d.kernel_ms}")

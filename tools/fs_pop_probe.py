"""What one fair victim search is made of, counted on the CPU oracle (VERDICT r04 "next" 1: measure before building).
oracle/kq_oracle.cpp g_fs_probe: every runFirstFsStrategy (preemption.go:384-470) of one whole scheduling cycle — the searches of the
nomination pass and of processEntry's recomputations alike.
usage: python tools/fs_pop_probe.py cfg4f|cfg4c [n_cq]          (text report on stdout)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kueue_amd import _ffi as F
from kueue_amd.api import make_config
from kueue_amd.population import generate
from oracle import kqo

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4f"
ncq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fair = name.endswith("f")
pop = generate(4, n_cq=ncq, fair_sharing=fair)
cfg = make_config(fair_sharing=fair)
heads = pop.heads_for_cycle(0, cycle=1)
l = kqo.lib()
l.kqo_fs_probe.restype = None
H = np.zeros(288, np.int64)
l.kqo_fs_probe(1, None)
t0 = time.perf_counter()
kqo.derive(pop.snapshot)
dec = kqo.cycle_run(cfg, pop.snapshot, heads)
dt = time.perf_counter() - t0
l.kqo_fs_probe(0, F.ptr(H))
print(f"# {name}: {ncq} ClusterQueues, {heads.n} heads, one cycle on the oracle in {dt:.1f} s of one core")
if not fair:
    print("(classical preemption: the probe counts fair searches only)")
S = max(1, H[0])
print(f"searches {H[0]}; per search: pops {H[1]/S:.1f}  failed pops {H[2]/S:.1f} ({100*H[2]/max(1,H[1]):.1f} %)  victims {H[3]/S:.1f}  "
      f"ClusterQueue visits (nextTarget results) {H[4]/S:.1f}")
print(f"visits: with a victim {H[5]} ({100*H[5]/max(1,H[4]):.1f} %), exhausted without one {H[6]} ({100*H[6]/max(1,H[4]):.1f} %), "
      f"skipped by fsStrategyUnsatisfiable {H[7]}, unconditional victims {H[8]}")
print(f"consecutive victims ({H[12]} pairs): same ClusterQueue {100*H[9]/max(1,H[12]):.1f} %, same parent cohort {100*H[10]/max(1,H[12]):.1f} %, "
      f"same child of the root {100*H[11]/max(1,H[12]):.1f} %")
print(f"consecutive nextTarget results ({H[15]} pairs): same parent cohort {100*H[13]/max(1,H[15]):.1f} %, same child of the root {100*H[14]/max(1,H[15]):.1f} %")
print(f"victims whose removal changes a cell of the preemptor's path: {H[16]} ({100*H[16]/max(1,H[3]):.1f} %); that do not: {H[17]}")
print(f"victims by height of the LCA above the target ClusterQueue: own queue {H[18]}, parent {H[19]}, grandparent {H[20]}, higher {H[21]}")
def hist(title, a):
    tot = max(1, a.sum())
    nz = np.nonzero(a)[0]
    hi = nz.max() if len(nz) else 0
    print(title)
    print("   " + "  ".join(f"{i}{'+' if i == 63 else ''}:{a[i]} ({100*a[i]/tot:.1f}%)" for i in range(hi + 1) if a[i]))
    print(f"   mean {(np.arange(64) * a).sum() / tot:.2f}")
hist("failed pops in front of the victim of a visit:", H[32:96])
hist("failed pops of a visit that ended exhausted:", H[96:160])
hist("run length of consecutive victims under the same child of the root:", H[160:224])
hist("candidates left in a ClusterQueue when it is visited:", H[224:288])

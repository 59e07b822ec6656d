"""Debug: heads per cycle of the bench's call-by-call loop and of the kq_pending_step loop on the same population (GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kueue_amd.api import make_config
from kueue_amd.engine import Engine
from kueue_amd.population import BASE_SEED, generate
pop = generate(3, seed=BASE_SEED)
snap = pop.snapshot
kcfg = make_config()
eng = Engine(kcfg)
tgt_cap = max(4096, 4 * snap.n_adm)
pending = pop.pending()
res = {}
for mode in ("sync", "pipe", "sync2"):
    eng.put(snap); eng.pending_put(pending)
    loop = bench.PendingLoop(eng, pop, 4, tgt_cap)
    ns, ts = [], []
    t0 = time.perf_counter()
    for c in range(220):
        if mode.startswith("sync"):
            ns.append(loop.step())
        else:
            loop.issue()
            if loop.in_flight >= 2:
                ns.append(loop.wait())
    if mode == "pipe":
        while loop.in_flight > 0:
            ns.append(loop.wait())
    st, counts = eng.pending_state()
    res[mode] = (ns, counts.copy(), time.perf_counter() - t0)
    print(mode, sum(ns), counts, f"{(time.perf_counter() - t0) / 220 * 1e3:.3f} ms/cycle")
a, b = res["sync"][0], res["pipe"][0]
bad = [i for i in range(220) if a[i] != b[i]]
print("first differing cycles:", bad[:10], [(a[i], b[i]) for i in bad[:10]])
print(a[:12], a[-12:])

"""debug: first random fair-sharing TAS cycle where the HIP engine differs from the oracle"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from tests.tasgen_cycle import random_tas_cycle_case
from oracle import kqo
from kueue_amd.engine import Engine
from tests.emu import kqe
bad_seeds = []
for seed in range(5000, 5600):
    cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=True, tight=seed % 2 == 0, preemption=seed % 3 != 0, partial=seed % 7 == 0)
    kqo.derive(snap)
    want, wout = kqo.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))
    if want.tas_stats["unsupported"]:
        continue
    eng = Engine(cfg); eng.put(snap)
    try:
        got, gout = eng.run_tas(heads, ct, tgt_cap=max(16, snap.n_adm))
    except Exception as ex:
        print(seed, "EXC", ex); eng.close(); continue
    eng.close()
    bad = want.equal(got)
    if bad:
        bad_seeds.append(seed)
        if len(bad_seeds) <= 3:
            print("seed", seed, "bad", bad, "n_heads", heads.n, "n_cq", snap.n_cq, "n_tree?", snap.n_cohort, "n_adm", snap.n_adm)
            for k in ("nominated_mode", "mode", "status", "action", "order", "skip", "requeue_reason"):
                print("  ", k, "want", want.a[k].tolist(), "got", got.a[k].tolist())
            print("   tgt want", want.a["tgt_off"].tolist(), want.a["tgt_adm"][:int(want.a["tgt_off"][-1])].tolist(), "got", got.a["tgt_off"].tolist())
            print("   tas_stats want", want.tas_stats, "got", got.tas_stats)
print("bad seeds", bad_seeds)

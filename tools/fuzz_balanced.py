"""Random batches and whole cycles with features.TASBalancedPlacement on, beyond the pinned seeds: engine (HIP through the C ABI, or the
emulation) against the oracle on every field. usage: python tools/fuzz_balanced.py <first seed> <last seed> [hip]"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import kqo
from tests.tasgen import random_tas_case
from tests.tasgen_cycle import random_tas_cycle_case
lo, hi = int(sys.argv[1]), int(sys.argv[2])
HIP = "hip" in sys.argv[3:]
if HIP:
    from kueue_amd import tas as T
    from kueue_amd.engine import Engine
else:
    from tests.emu import kqe
bad = nb = nc = changed = 0
for seed in range(lo, hi):
    topo, rq = random_tas_case(seed, max_blocks=3 + seed % 3, max_racks=4 + seed % 4, max_hosts=6 + seed % 5, n_workloads=12 + seed % 7)
    off = kqo.tas_find(topo, rq)
    topo.feature_bits |= 2; topo._struct = None
    want = kqo.tas_find(topo, rq)
    changed += 1 if want.equal(off) else 0
    eng = T.TASEngine() if HIP else kqe.EmuTas()
    eng.put(topo); got = eng.find(rq); eng.close()
    if want.equal(got) or got.bytes != want.bytes:
        bad += 1; print("MISMATCH batch seed", seed, want.equal(got))
    nb += 1
    if seed % 2 == 0:
        cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=seed % 6 == 5, tight=seed % 4 == 0, preemption=seed % 3 != 0)
        for i in range(len(ct.topos)):
            ct._topo_arr[i].profile_mixed |= 2
        kqo.derive(snap)
        rc = 64 * max(heads.n_ps, 1)
        w, wo = kqo.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm), rsn_cap=rc)
        if w.tas_stats["unsupported"]:
            continue
        e = Engine(cfg) if HIP else kqe.EmuEngine(cfg)
        e.put(snap); g, go = e.run_tas(heads, ct, tgt_cap=max(16, snap.n_adm), rsn_cap=rc); e.close()
        n_ps = heads.n_ps; m = int(wo.a["dom_off"][n_ps])
        ok = not w.equal(g) and np.array_equal(wo.a["ps_tas"][:n_ps], go.a["ps_tas"][:n_ps]) and np.array_equal(wo.a["dom_off"], go.a["dom_off"]) and \
            np.array_equal(wo.a["dom_leaf"][:m], go.a["dom_leaf"][:m]) and np.array_equal(wo.a["dom_count"][:m], go.a["dom_count"][:m]) and np.array_equal(wo.a["tas_usage_after"], go.a["tas_usage_after"])
        if not ok:
            bad += 1; print("MISMATCH cycle seed", seed, w.equal(g))
        nc += 1
print(f"seeds {lo}..{hi}: {nb} batches ({changed} answered differently than with the gate off), {nc} cycles, mismatches {bad}")

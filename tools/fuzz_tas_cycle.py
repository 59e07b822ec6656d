"""Random TAS cycles beyond the pinned seeds: the emulated engine (tests/emu) against the oracle, kq_cycle_run_tas on every field.
usage: python tools/fuzz_tas_cycle.py <first seed> <last seed> [hip] [second]
       hip: the HIP engine through the C ABI instead of the emulation
       second: cycles that mix in heads on their second pass after a node failure (tests/tasgen_cycle.py random_second_pass_case)"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import kqo
from tests.emu import kqe
from tests.tasgen_cycle import random_second_pass_case, random_tas_cycle_case
lo,hi=int(sys.argv[1]),int(sys.argv[2])
HIP = "hip" in sys.argv[3:]
SECOND = "second" in sys.argv[3:]
acts = {}
if HIP:
    from kueue_amd.engine import Engine, EngineError
bad=0; rec=0; fails=0; tg=0; uns=0
for seed in range(lo,hi):
    if SECOND: cfg, snap, heads, ct, n_second = random_second_pass_case(seed, fair=seed%5==4, tight=seed%2==0, preemption=seed%3!=0, partial=seed%7==0)
    else: cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=False, tight=seed%2==0, preemption=seed%3!=0, partial=seed%5==0)
    kqo.derive(snap)
    rc=64*max(heads.n_ps,1)
    want,wout=kqo.cycle_run_tas(cfg,snap,heads,ct,tgt_cap=max(16,snap.n_adm),rsn_cap=rc)
    if HIP:
        e=Engine(cfg); e.put(snap)
        try:
            got,gout=e.run_tas(heads,ct,tgt_cap=max(16,snap.n_adm),rsn_cap=rc); got.rc=0
        except EngineError as ex:
            assert want.tas_stats["unsupported"] and ex.code==-4, (seed, ex)
            got=None
        e.close()
    else:
        e=kqe.EmuEngine(cfg); e.put(snap); got,gout=e.run_tas(heads,ct,tgt_cap=max(16,snap.n_adm),rsn_cap=rc); e.close()
    if want.tas_stats["unsupported"]:
        uns+=1; continue
    n_ps=heads.n_ps; m=int(wout.a["dom_off"][n_ps])
    ok = got.rc==0 and not want.equal(got) and np.array_equal(wout.a["ps_tas"][:n_ps],gout.a["ps_tas"][:n_ps]) and np.array_equal(wout.a["dom_off"],gout.a["dom_off"]) and np.array_equal(wout.a["dom_leaf"][:m],gout.a["dom_leaf"][:m]) and np.array_equal(wout.a["dom_count"][:m],gout.a["dom_count"][:m]) and np.array_equal(wout.a["tas_usage_after"],gout.a["tas_usage_after"])
    if not ok: bad+=1; print("MISMATCH seed",seed, got.rc, want.equal(got))
    if SECOND:
        for i in range(n_second): acts[(int(want.a["action"][i]), int(want.a["status"][i]))] = acts.get((int(want.a["action"][i]), int(want.a["status"][i])), 0) + 1
    rec+=want.tas_stats["recomputes"]; tg+=int(want.a["tgt_off"][-1]); fails+=int((want.a["rsn_code"][:int(want.a["rsn_off"][heads.n])]==200).sum())
if SECOND: print("second-pass heads by (action, status):", dict(sorted(acts.items())))
print(f"seeds {lo}..{hi}: mismatches {bad}, unsupported {uns}, TAS recomputations {rec}, preemption targets {tg}, TAS failure reasons {fails}")

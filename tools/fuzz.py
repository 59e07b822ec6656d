#!/usr/bin/env python
"""Open-ended differential fuzzing of the engine's device logic (1-lane CPU emulation, tests/emu) against the oracle, beyond the
seeds the test-suite pins. Test infrastructure: needs no GPU.

  python tools/fuzz.py cycle  LO HI     one cycle per seed on random populations (classical / fair sharing / duplicate heads / partial)
  python tools/fuzz.py tight  LO HI     the same on over-committed populations (negative reservations, exact usage_np mode)
  python tools/fuzz.py tas    LO HI     FindTopologyAssignmentsForFlavor on random topologies
  python tools/fuzz.py loop   LO HI     closed loops (commit / release) of 6-14 cycles on random cfg2 / cfg3 populations

Prints the seeds that differ (none expected). Runs of this round: cycle 100000-108000, 200000-204000, 300000-320000; tight
400000-412000; tas 10000-19000; loop 0-180 — all clean. On the last code of round 2: cycle 500000-506000, tight 600000-604000,
loop 1000-1120 — all clean. On the last code of round 3: cycle 700000-706000, tight 800000-804000, loop 2000-2080, tas 50000-78000
— all clean; the TAS cycles have their own driver (tools/fuzz_tas_cycle.py: 20000-60000 on the emulation, 40000-40600 / 50000-50600 /
60000-60300 on the MI355X; two seeds of that campaign, 21190 and 27235, found the lost psError pinned in tests/test_tas_cycle_engine.py).
"""
import copy
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import kqo  # noqa: E402
from tests.emu import kqe  # noqa: E402


def one_cycle(cfg, snap, heads):
    kqo.derive(snap)
    want = kqo.cycle_run(cfg, snap, heads, want_usage=True)
    eng = kqe.EmuEngine(cfg)
    try:
        eng.put(snap)
        got = eng.run(heads, want_usage=True)
    finally:
        eng.close()
    return got.rc == 0 and not want.equal(got) and np.array_equal(want.usage_after, got.usage_after) and got.bytes == want.stats["total"]


def cycle(seed, tight=False):
    from tests.randgen import random_case
    if tight:
        fair = seed % 2 == 1
        return one_cycle(*random_case(seed, fair=fair, preemption=True, partial=(seed % 5 == 0), max_cq=8, fair_dups=fair, tight=True))
    fair = seed % 5 == 0
    kw = dict(fair_dups=True, max_cq=6 + (seed % 3) * 5) if fair and seed % 2 else {}
    return one_cycle(*random_case(seed, fair=fair, preemption=True, partial=(seed % 3 == 0), **kw))


def tas(seed):
    from tests.tasgen import random_tas_case
    topo, rq = random_tas_case(seed)
    want = kqo.tas_find(topo, rq)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        got = eng.find(rq)
    finally:
        eng.close()
    return not want.equal(got) and got.bytes == want.bytes


def loop(seed):
    from kueue_amd.api import make_config
    from kueue_amd.population import generate
    rnd = random.Random(seed)
    cfgn = rnd.choice([2, 3])
    fair = seed % 3 == 0
    n_cq = rnd.randint(8, 120) if cfgn == 3 else rnd.randint(4, 60)
    cycles, hold = rnd.randint(6, 14), rnd.randint(1, 5)
    pop = generate(cfgn, seed=1000 + seed, fair_sharing=fair, per_cq=cycles + 1, n_cq=n_cq)
    cfg, snap = make_config(fair_sharing=fair), pop.snapshot
    eng = kqe.EmuEngine(cfg)
    eng.put(snap)
    osnap = copy.copy(snap)
    osnap.arrays = dict(snap.arrays)
    held = []
    try:
        for c in range(cycles):
            heads = pop.heads_for_cycle(c, cycle=c + 1)
            if kqo.cycle_run(cfg, osnap, heads).equal(eng.run(heads)):
                return False
            usage, na, triples = kqo.cycle_commit(cfg, osnap, heads)
            if eng.commit() != na:
                return False
            held.append(triples)
            osnap.arrays["usage"] = usage
            osnap._struct = None
            if len(held) > hold:
                usage = kqo.usage_apply(cfg, osnap, held.pop(0), add=False)
                eng.release(len(held) + 1)
                osnap.arrays["usage"] = usage
                osnap._struct = None
            if not np.array_equal(eng.read_usage(), usage):
                return False
    finally:
        eng.close()
    return True


if __name__ == "__main__":
    mode, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    fn = {"cycle": cycle, "tight": lambda s: cycle(s, True), "tas": tas, "loop": loop}[mode]
    import ctypes as C
    if mode in ("cycle", "tight"):
        kqe.lib().kqe_cs_check(1)  # every scan-formulated classical search is re-run as a candidate-by-candidate walk and compared
    bad = [s for s in range(lo, hi) if not fn(s)]
    st = (C.c_longlong * 32)()
    kqe.lib().kqe_cstat(st)
    print(mode, "seeds", lo, hi, "differ:", bad, "| classical searches: scan", st[20], "walk-only", st[21] - st[20], "scan != walk", st[22])

"""The speculative parallel rounds of the process step (kueue_amd/csrc/kq_spec.hpp) against the oracle.

processEntry (scheduler.go:392-523) for entries without preemption targets is solved as rounds of segmented prefix sums; the serial
kernel (process_tree) takes over at K::spec_resume. Every variant must give the oracle's decisions, usage plane and byte count:
  0  full windows                      1 / 2  tiny windows (many windows per tree, capacity cuts in the middle of chunks)
  3  two rounds, then the undecided tail goes back to the serial kernel            4  rounds off
and the statistics say which path ran (kq_debug_spec_stats)."""
import numpy as np
import pytest

from kueue_amd.api import make_config
from kueue_amd.population import generate
from tests.emu import kqe
from tests.randgen import random_case

POPS = [("cfg2", dict(cfg=2)), ("cfg3-150cq", dict(cfg=3, n_cq=150, per_cq=6)), ("cfg1", dict(cfg=1))]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("name,kw", POPS, ids=[p[0] for p in POPS])
def test_rounds_on_populations(oracle, name, kw, variant):
    pop = generate(**kw)
    cfg = make_config()
    eng = kqe.EmuEngine(cfg)
    eng.spec_variant(variant)
    decided = handed = trunc = 0
    try:
        eng.put(pop.snapshot)
        for c in range(4):
            heads = pop.heads_for_cycle(c, cycle=c + 1)
            want = oracle.cycle_run(cfg, pop.snapshot, heads, want_usage=True)
            got = eng.run(heads, want_usage=True, tgt_cap=max(4096, pop.snapshot.n_adm))
            assert got.rc == 0, got.error
            assert not want.equal(got), (name, variant, c, want.equal(got))
            assert np.array_equal(want.usage_after, got.usage_after)
            assert got.bytes == want.stats["total"]
            st = eng.spec_stats()
            decided += st[2]; handed += st[3]; trunc += st[7]
        if variant == 4:
            assert decided == 0
        elif variant in (0, 1, 2) and name != "cfg2":
            # populations without preemption and one head per ClusterQueue: the rounds decide every entry (cfg 2 at small windows can hit
            # the round limit of variant 2)
            assert handed == 0 and decided == 4 * heads.n, (decided, handed)
        if variant == 3 and name.startswith("cfg3"):
            assert trunc > 0 and handed > 0   # root row binding: two rounds are not enough, the tail went to the serial kernel
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(60))
def test_rounds_on_random_cycles(oracle, seed):
    """Random trees (unbalanced depths, lending / borrowing limits on cohorts, Unlimited cells, duplicate heads, preemption): the rounds
    take what they can, in every variant, and the cycle equals the oracle's."""
    cfg, snap, heads = random_case(seed, fair=False, preemption=(seed % 3 == 0), partial=False)
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
    for variant in (0, 1, 3):
        eng = kqe.EmuEngine(cfg)
        eng.spec_variant(variant)
        try:
            eng.put(snap)
            got = eng.run(heads, want_usage=True)
            assert got.rc == 0, got.error
            assert not want.equal(got), (seed, variant, want.equal(got))
            assert np.array_equal(want.usage_after, got.usage_after), (seed, variant)
            assert got.bytes == want.stats["total"], (seed, variant)
        finally:
            eng.close()


def test_rounds_take_negative_reservations_only_without_preemption(oracle):
    """cfg 3 at its fill holds ClusterQueues beyond nominal + borrowingLimit: their Preempt-mode heads reserve a NEGATIVE amount
    (scheduler.go:806). Without any preempting ClusterQueue the rounds apply it (it only touches the ClusterQueue's own cell)."""
    pop = generate(cfg=3, n_cq=300, per_cq=4)
    cfg = make_config()
    eng = kqe.EmuEngine(cfg)
    eng.spec_variant(0)
    try:
        eng.put(pop.snapshot)
        heads = pop.heads_for_cycle(0, cycle=1)
        want = oracle.cycle_run(cfg, pop.snapshot, heads, want_usage=True)
        got = eng.run(heads, want_usage=True)
        assert not want.equal(got)
        assert np.array_equal(want.usage_after, got.usage_after)
        st = eng.spec_stats()
        assert st[2] == heads.n and st[3] == 0 and st[6] == 0
    finally:
        eng.close()

"""GPU parity at BASELINE.json sizes: synthetic populations (kueue_amd/population.py), engine vs oracle,
bit-exact decisions, targets, post-cycle usage and algorithmic byte count."""
import numpy as np
import pytest

from kueue_amd.api import make_config
from kueue_amd.population import generate

pytestmark = pytest.mark.gpu

CASES = [
    ("cfg1", dict(cfg=1), [0, 1, 5]),
    ("cfg2-full", dict(cfg=2), [0, 3, 40]),
    ("cfg3-full", dict(cfg=3), [0, 7, 99]),                       # 1000 CQ, 1111 nodes, 20k admitted, 100k pending
    ("cfg4c-300cq", dict(cfg=4, n_cq=300, per_cq=4), [0, 2]),     # classical preemption + DeferredFit + overlap recompute
    ("cfg4f-100cq", dict(cfg=4, n_cq=100, per_cq=4, fair_sharing=True), [0, 1]),  # fair sharing + fair preemption (oracle: ~12 s/cycle)
    ("cfg3f-300cq", dict(cfg=3, n_cq=300, per_cq=4, fair_sharing=True), [0, 1]),  # fair-sharing iterator, admission-heavy
]


@pytest.mark.parametrize("name,kw,cycles", CASES, ids=[c[0] for c in CASES])
def test_population_cycles_bit_exact(oracle, name, kw, cycles):
    from kueue_amd.engine import Engine
    pop = generate(**kw)
    cfg = make_config(fair_sharing=bool(kw.get("fair_sharing")))
    eng = Engine(cfg)
    try:
        eng.put(pop.snapshot)
        for c in cycles:
            heads = pop.heads_for_cycle(c, cycle=c + 1)
            want = oracle.cycle_run(cfg, pop.snapshot, heads, want_usage=True)
            got = eng.run(heads, tgt_cap=max(4096, pop.snapshot.n_adm))
            bad = want.equal(got)
            assert not bad, (name, c, bad)
            assert np.array_equal(want.usage_after, eng.usage_after()), (name, c)
            assert got.bytes == want.stats["total"], (name, c, got.bytes, want.stats)
    finally:
        eng.close()


def test_all_pending_batch_nominate(oracle):
    """100k heads in one launch (SURVEY §8d batch mode): every pending workload of cfg 3 nominated against the
    same snapshot; nominate outputs (flavors, modes, borrowing) must equal the oracle's."""
    from kueue_amd.engine import Engine
    pop = generate(3, n_cq=1000, per_cq=20)
    cfg = make_config()
    heads = pop.all_heads()
    want = oracle.cycle_run(cfg, pop.snapshot, heads)
    eng = Engine(cfg)
    try:
        eng.put(pop.snapshot)
        got = eng.run(heads)
        for k in ("nominated_mode", "borrowing", "flavor", "res_mode", "tried_idx", "ps_count", "status", "action", "order", "mode", "skip"):
            assert np.array_equal(want.a[k], got.a[k]), k
    finally:
        eng.close()


def test_fair_preemption_with_helper_workgroups(oracle, monkeypatch):
    """KQ_HELP_BLOCKS: the SimulatePreemption calls of a recomputation inside k_process_fair are posted as a batch and taken by helper
    workgroups (K::help, kq_device.hpp help_exec / helper_main). Off by default; whoever runs a task, the cycle is the oracle's."""
    from kueue_amd.engine import Engine
    monkeypatch.setenv("KQ_HELP_BLOCKS", "24")
    pop = generate(4, n_cq=100, per_cq=4, fair_sharing=True)
    cfg = make_config(fair_sharing=True)
    eng = Engine(cfg)
    try:
        eng.put(pop.snapshot)
        for c in (0, 1):
            heads = pop.heads_for_cycle(c, cycle=c + 1)
            want = oracle.cycle_run(cfg, pop.snapshot, heads, want_usage=True)
            got = eng.run(heads, tgt_cap=max(4096, pop.snapshot.n_adm))
            assert not want.equal(got), (c, want.equal(got))
            assert np.array_equal(want.usage_after, eng.usage_after()), c
            assert got.bytes == want.stats["total"], c
    finally:
        eng.close()

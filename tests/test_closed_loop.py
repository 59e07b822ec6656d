"""Closed-loop driver support (SURVEY §8d "a run"): kq_cycle_commit folds the admissions of a cycle into the resident
snapshot, kq_cycle_release takes an older commit out again. Checked against the oracle replaying the same loop."""
import copy

import numpy as np
import pytest

from kueue_amd.population import generate
from kueue_amd.api import make_config


def _loop(oracle, eng_factory, fair, cycles=6, hold=2, n_cq=60, cfgn=3, usage_every=1):
    kw = {} if n_cq is None else {"n_cq": n_cq}
    pop = generate(cfgn, per_cq=cycles + 1, fair_sharing=fair, **kw)
    cfg = make_config(fair_sharing=fair)
    snap = pop.snapshot
    eng = eng_factory(cfg)
    try:
        eng.put(snap)
        osnap = copy.copy(snap)
        osnap.arrays = dict(snap.arrays)
        held = []
        total = 0
        for c in range(cycles):
            heads = pop.heads_for_cycle(c, cycle=c + 1)
            got = eng.run(heads)
            want = oracle.cycle_run(cfg, osnap, heads)
            assert not want.equal(got), (c, want.equal(got))
            usage, na, triples = oracle.cycle_commit(cfg, osnap, heads)
            assert eng.commit() == na
            total += na
            held.append(triples)
            osnap.arrays["usage"] = usage
            osnap._struct = None
            if c % usage_every == usage_every - 1:
                assert np.array_equal(eng.read_usage(), usage), c
            if len(held) > hold:  # the workloads admitted `hold` cycles ago finish
                old = held.pop(0)
                usage = oracle.usage_apply(cfg, osnap, old, add=False)
                eng.release(len(held) + 1)
                osnap.arrays["usage"] = usage
                osnap._struct = None
                if c % usage_every == usage_every - 1:
                    assert np.array_equal(eng.read_usage(), usage), ("release", c)
        assert total > 0
    finally:
        eng.close()


@pytest.mark.parametrize("fair", [False, True])
def test_closed_loop_emulated(oracle, fair):
    from tests.emu import kqe
    _loop(oracle, kqe.EmuEngine, fair)


@pytest.mark.gpu
@pytest.mark.parametrize("fair", [False, True])
def test_closed_loop_gpu(oracle, fair):
    from kueue_amd.engine import Engine
    _loop(oracle, Engine, fair, cycles=8, hold=3, n_cq=200)


# The closed loop of the default bench at full size (1000 ClusterQueues, one head each per cycle): every decision of every cycle
# against the oracle replaying the same loop. The snapshot drifts for 40 cycles (admissions saturate the cohorts, releases free
# them again), which walks the serial core of k_process through screened, fitting and no-longer-fitting entries in every mix.
def test_closed_loop_soak_emulated(oracle):
    from tests.emu import kqe
    _loop(oracle, kqe.EmuEngine, False, cycles=40, hold=4, n_cq=None, usage_every=5)


@pytest.mark.gpu
def test_closed_loop_soak_gpu(oracle):
    from kueue_amd.engine import Engine
    _loop(oracle, Engine, False, cycles=40, hold=4, n_cq=None, usage_every=5)

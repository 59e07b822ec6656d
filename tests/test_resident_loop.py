"""The loop bench.py times — kq_heads_put once, then kq_cycle_run_resident + kq_cycle_commit + kq_cycle_release per cycle — checked
decision by decision against the oracle replaying the same closed loop (VERDICT r01 "what's weak" 1: the benchmarked entry points
were never parity-tested). Same driver for the 1-lane emulation (CPU suite) and the HIP engine (GPU suite, full cfg 3)."""
import copy

import numpy as np
import pytest

from kueue_amd.api import Decisions, make_config
from kueue_amd.population import generate


def _resident_loop(oracle, eng_factory, fair, cfgn, n_cq, cycles, hold, usage_every=1, n_batches=None):
    kw = {} if n_cq is None else {"n_cq": n_cq}
    nb = n_batches or cycles
    pop = generate(cfgn, per_cq=nb + 1, fair_sharing=fair, **kw)
    cfg = make_config(fair_sharing=fair)
    snap = pop.snapshot
    eng = eng_factory(cfg)
    try:
        eng.put(snap)
        batches = [pop.heads_for_cycle(c, cycle=c + 1) for c in range(nb)]
        for b, hb in enumerate(batches):
            eng.heads_put(hb, b)
        outs = [Decisions(hb, tgt_cap=max(4096, snap.n_adm)) for hb in batches]
        osnap = copy.copy(snap)
        osnap.arrays = dict(snap.arrays)
        held, live, admitted = [], 0, 0
        for i in range(cycles):
            b = i % nb  # bench.py wraps around its resident batches the same way
            eng.run_resident(b, outs[b])
            want = oracle.cycle_run(cfg, osnap, batches[b])
            bad = want.equal(outs[b])
            assert not bad, (i, bad)
            usage, na, triples = oracle.cycle_commit(cfg, osnap, batches[b])
            assert eng.try_commit() == 0
            admitted += na
            live += 1
            held.append(triples)
            osnap.arrays["usage"] = usage; osnap._struct = None
            if live > hold:
                eng.release(hold + 1)
                live -= 1
                osnap.arrays["usage"] = oracle.usage_apply(cfg, osnap, held.pop(0), add=False); osnap._struct = None
            if i % usage_every == usage_every - 1:
                assert np.array_equal(eng.read_usage(), osnap.arrays["usage"]), i
        assert admitted > 0
    finally:
        eng.close()


@pytest.mark.parametrize("fair", [False, True])
def test_resident_loop_emulated(oracle, fair):
    from tests.emu import kqe
    _resident_loop(oracle, kqe.EmuEngine, fair, 3, 60, cycles=9, hold=2, n_batches=6)


@pytest.mark.gpu
@pytest.mark.parametrize("fair,n_cq,cycles", [(False, None, 30), (True, 300, 8)], ids=["cfg3-full", "cfg3f-300cq"])
def test_resident_loop_gpu(oracle, fair, n_cq, cycles):
    from kueue_amd.engine import Engine
    _resident_loop(oracle, Engine, fair, 3, n_cq, cycles=cycles, hold=4, usage_every=5, n_batches=min(cycles, 25))


def _stale_batch(eng_factory):
    """a snapshot change voids resident batches (their indices / strides belong to the old snapshot)."""
    cfg = make_config()
    pop = generate(2, n_cq=16, per_cq=3)
    small = generate(2, n_cq=8, per_cq=3)
    eng = eng_factory(cfg)
    try:
        eng.put(pop.snapshot)
        hb = pop.heads_for_cycle(0, cycle=1)
        eng.heads_put(hb, 0)
        out = Decisions(hb)
        assert eng.run_resident(0, out) == 0
        eng.put(small.snapshot)           # fewer ClusterQueues: head cq indices of the batch are out of range now
        assert eng.run_resident(0, out, check=False) == -1   # KQ_EINVAL, not an out-of-bounds read
        assert eng.try_commit() == -1     # and nothing to commit
        hb2 = small.heads_for_cycle(0, cycle=1)
        eng.heads_put(hb2, 0)
        out2 = Decisions(hb2)
        assert eng.run_resident(0, out2) == 0
        # replacing the batch of an uncommitted cycle drops that cycle: commit refuses instead of reading freed head arrays
        eng.heads_put(small.heads_for_cycle(1, cycle=2), 0)
        assert eng.try_commit() == -1
    finally:
        eng.close()


def test_stale_batch_emulated():
    from tests.emu import kqe
    _stale_batch(kqe.EmuEngine)


@pytest.mark.gpu
def test_stale_batch_gpu():
    from kueue_amd.engine import Engine
    _stale_batch(Engine)

"""kq_snapshot_derive: SubtreeQuota / cohort Usage / flags computed by the engine from raw Quotas + ClusterQueue
usage (resource_node.go:167-230), against the oracle's restatement; and a cycle run on the device-derived snapshot."""
import numpy as np
import pytest

from tests.randgen import random_case


def _check(oracle, eng_factory, seed, fair):
    cfg, snap, heads = random_case(70_000 + seed, fair=fair, preemption=True, max_cq=10, fair_dups=fair)
    # upload the UNDERIVED snapshot (SubtreeQuota = 0, cohort usage = 0), derive on the engine
    eng = eng_factory(cfg)
    try:
        eng.put(snap)
        sq, us, fl = eng.derive()
        got = eng.run(heads)
    finally:
        eng.close()
    oracle.derive(snap)
    a = snap.arrays
    assert np.array_equal(sq, a["subtree_quota"]), seed
    assert np.array_equal(us, a["usage"]), seed
    assert np.array_equal(fl, a["quota_flags"]), seed
    want = oracle.cycle_run(cfg, snap, heads)
    assert not want.equal(got), (seed, want.equal(got))


@pytest.mark.parametrize("seed", range(60))
def test_derive_emulated(oracle, seed):
    from tests.emu import kqe
    _check(oracle, kqe.EmuEngine, seed, fair=seed % 2 == 1)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(3))
def test_derive_gpu(oracle, block):
    from kueue_amd.engine import Engine
    for seed in range(block * 20, block * 20 + 20):
        _check(oracle, Engine, seed, fair=seed % 2 == 1)

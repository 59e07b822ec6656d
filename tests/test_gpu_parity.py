"""GPU parity (needs a MI355X): the HIP engine through the C ABI vs the oracle, bit-exact."""
import numpy as np
import pytest

from tests.randgen import random_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine_mod():
    from kueue_amd import engine
    return engine


@pytest.mark.parametrize("block", range(15))
def test_random_cycles_bit_exact(oracle, engine_mod, block):
    for seed in range(block * 40, block * 40 + 40):
        cfg, snap, heads = random_case(seed, fair=False, preemption=True, partial=(seed % 3 == 0))
        oracle.derive(snap)
        want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
        eng = engine_mod.Engine(cfg)
        try:
            eng.put(snap)
            got = eng.run(heads)
            bad = want.equal(got)
            assert not bad, (seed, bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
            assert np.array_equal(want.usage_after, eng.usage_after()), seed
        finally:
            eng.close()


@pytest.mark.parametrize("block", range(10))
def test_random_fair_sharing_cycles_bit_exact(oracle, engine_mod, block):
    """Fair sharing on the device: DRS, tournament iterator interleaved with processEntry, fair preemption."""
    for seed in range(block * 40, block * 40 + 40):
        cfg, snap, heads = random_case(20_000 + seed, fair=True, preemption=True, partial=(seed % 4 == 0),
                                       max_cq=6 + (seed % 3) * 5, fair_dups=True)
        oracle.derive(snap)
        want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
        eng = engine_mod.Engine(cfg)
        try:
            eng.put(snap)
            got = eng.run(heads)
            bad = want.equal(got)
            assert not bad, (seed, bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
            assert np.array_equal(want.usage_after, eng.usage_after()), seed
            assert got.bytes == want.stats["total"], (seed, got.bytes, want.stats)
        finally:
            eng.close()

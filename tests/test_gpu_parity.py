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


@pytest.mark.parametrize("block", range(6))
def test_overcommitted_queues_bit_exact(oracle, engine_mod, block):
    """Negative capacity reservations (usage above nominal + borrowingLimit) with preemptions in the same tree:
    usage_np columns are rebuilt in the reference's canonical removal order. Classical and fair sharing."""
    for seed in range(block * 50, block * 50 + 50):
        fair = seed % 2 == 1
        cfg, snap, heads = random_case(50_000 + seed, fair=fair, preemption=True, partial=(seed % 5 == 0), max_cq=8, fair_dups=fair, tight=True)
        oracle.derive(snap)
        want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
        eng = engine_mod.Engine(cfg)
        try:
            eng.put(snap)
            got = eng.run(heads, tgt_cap=max(64, snap.n_adm * 8))
            bad = want.equal(got)
            assert not bad, (seed, bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
            assert np.array_equal(want.usage_after, eng.usage_after()), seed
            assert got.bytes == want.stats["total"], (seed, got.bytes, want.stats)
        finally:
            eng.close()


@pytest.mark.parametrize("block", range(3))
def test_random_fair_sharing_cycles_with_helper_workgroups(oracle, engine_mod, block, monkeypatch):
    """The same with K::help on (KQ_HELP_BLOCKS): several trees per cycle, every tree's leader posts its recomputation batches to the
    same pool of helper workgroups."""
    monkeypatch.setenv("KQ_HELP_BLOCKS", "12")
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
            assert not bad, (seed, bad)
            assert np.array_equal(want.usage_after, eng.usage_after()), seed
            assert got.bytes == want.stats["total"], (seed, got.bytes, want.stats)
        finally:
            eng.close()

"""Differential test: the engine's device logic (1-lane CPU emulation, test-only) vs the oracle on
seeded random populations. Bit-exact on every decision field, target set and post-cycle usage."""
import numpy as np
import pytest

from tests.emu import kqe
from tests.randgen import random_case


def run_both(oracle, cfg, snap, heads, exact_drs=False):
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
    eng = kqe.EmuEngine(cfg)
    try:
        if exact_drs:
            eng.force_exact_drs()
        eng.put(snap)
        got = eng.run(heads, want_usage=True)
    finally:
        eng.close()
    return want, got


@pytest.mark.parametrize("seed", range(600))
def test_classical_random(oracle, seed):
    cfg, snap, heads = random_case(seed, fair=False, preemption=True, partial=(seed % 3 == 0))
    want, got = run_both(oracle, cfg, snap, heads)
    assert got.rc == 0, got.error
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert np.array_equal(want.usage_after, got.usage_after)
    # the engine's algorithmic-byte counter must equal the oracle's for the same decisions
    assert got.bytes == want.stats["total"], (got.bytes, want.stats)


@pytest.mark.parametrize("seed", range(400))
def test_fair_sharing_random(oracle, seed):
    """Fair sharing: DRS, the tournament iterator interleaved with processEntry, fair preemption (S2-a/S2-b)."""
    cfg, snap, heads = random_case(10_000 + seed, fair=True, preemption=True, partial=(seed % 4 == 0))
    want, got = run_both(oracle, cfg, snap, heads)
    assert got.rc == 0, got.error
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert np.array_equal(want.usage_after, got.usage_after)
    assert got.bytes == want.stats["total"], (got.bytes, want.stats)


@pytest.mark.parametrize("seed", range(400))
def test_fair_sharing_random_variants(oracle, seed):
    """Fair sharing with duplicate-CQ heads, larger hierarchies, strategy lists and fair-sharing gates toggled."""
    cfg, snap, heads = random_case(20_000 + seed, fair=True, preemption=True, partial=(seed % 4 == 0), max_cq=6 + (seed % 3) * 5, fair_dups=True)
    want, got = run_both(oracle, cfg, snap, heads)
    assert got.rc == 0, got.error
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert np.array_equal(want.usage_after, got.usage_after)
    assert got.bytes == want.stats["total"], (got.bytes, want.stats)


@pytest.mark.parametrize("seed", range(200))
def test_fair_sharing_exact_drs_loops(oracle, seed):
    """The saturation-safe DRS loops (taken when amounts are too large for the incremental borrowed sums)."""
    cfg, snap, heads = random_case(20_000 + seed, fair=True, preemption=True, partial=(seed % 4 == 0), max_cq=6 + (seed % 3) * 5, fair_dups=True)
    want, got = run_both(oracle, cfg, snap, heads, exact_drs=True)
    assert got.rc == 0, got.error
    assert not want.equal(got)
    assert np.array_equal(want.usage_after, got.usage_after)
    assert got.bytes == want.stats["total"], (got.bytes, want.stats)


@pytest.mark.parametrize("seed", range(300))
def test_overcommitted_queues(oracle, seed):
    """Usage above nominal + borrowingLimit: capacity reservations can be negative, after which the snapshot is no
    longer order-independent under removals; the engine must fall back to the reference's canonical order."""
    fair = seed % 2 == 1
    cfg, snap, heads = random_case(50_000 + seed, fair=fair, preemption=True, partial=(seed % 5 == 0), max_cq=8, fair_dups=fair, tight=True)
    want, got = run_both(oracle, cfg, snap, heads)
    assert got.rc == 0, got.error
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert np.array_equal(want.usage_after, got.usage_after)
    assert got.bytes == want.stats["total"], (got.bytes, want.stats)

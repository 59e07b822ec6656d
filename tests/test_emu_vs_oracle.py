"""Differential test: the engine's device logic (1-lane CPU emulation, test-only) vs the oracle on
seeded random populations. Bit-exact on every decision field, target set and post-cycle usage."""
import numpy as np
import pytest

from tests.emu import kqe
from tests.randgen import random_case


def run_both(oracle, cfg, snap, heads):
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
    eng = kqe.EmuEngine(cfg)
    try:
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

"""Whole cycles over workloads with PodSetGroupName groups of several podsets — one flavor scan per group over the sum of the members' requests
(flavorassigner.go:782-860) — engine == oracle on every decision array, the reason records, the usage after the cycle and the byte counter.
CPU suite: the 1-lane emulation of the device code; -m gpu: the HIP engine through the C ABI. tests/test_oracle_grouped_flavors.py pins the
oracle's (and the engine's) grouped scan on the reference's own leader-worker-set rows."""
import numpy as np
import pytest

from tests.groupgen import grouped_case

_SEEN = {"multi": 0}


def _cycle(oracle, make, seed):
    fair = seed % 4 == 3
    cfg, snap, heads, n_multi = grouped_case(70_000 + seed, fair=fair, preemption=seed % 5 != 0, partial=seed % 3 == 0, tight=seed % 2 == 0, fair_dups=fair)
    _SEEN["multi"] += n_multi
    oracle.derive(snap)
    rsn_cap = 96 * max(heads.n_ps, 1)
    want = oracle.cycle_run(cfg, snap, heads, want_usage=True, rsn_cap=rsn_cap)
    eng = make(cfg)
    try:
        eng.put(snap)
        if hasattr(eng, "usage_after"):   # the HIP engine: the usage after the cycle is its own read-back
            got = eng.run(heads, rsn_cap=rsn_cap)
            usage_after = eng.usage_after()
        else:
            got = eng.run(heads, want_usage=True, rsn_cap=rsn_cap)
            usage_after = got.usage_after
    finally:
        eng.close()
    assert getattr(got, "rc", 0) == 0, (seed, getattr(got, "error", ""))
    bad = want.equal(got)
    assert not bad, (seed, bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert np.array_equal(want.usage_after, usage_after), seed
    assert got.bytes == want.stats["total"], (seed, got.bytes, want.stats)


@pytest.mark.parametrize("seed", range(2400))
def test_grouped_cycles_emulated(oracle, seed):
    from tests.emu import kqe
    _cycle(oracle, kqe.EmuEngine, seed)


def test_population_holds_groups():
    n = sum(grouped_case(70_000 + s)[3] for s in range(40))
    assert n >= 60, n


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(8))
def test_grouped_cycles_gpu(oracle, block):
    from kueue_amd.engine import Engine
    for seed in range(block * 300, block * 300 + 300):
        _cycle(oracle, Engine, seed)


def _tas_cycle(oracle, make, seed):
    """Random TAS cycles whose 2-podset workloads are mostly leader + workers groups, the leader often without requests or with fewer resources
    than the workers (tests/tasgen_cycle.py rich_groups): every decision array, reason record, TopologyAssignment and the leaf usage after the cycle."""
    from tests.tasgen_cycle import random_tas_cycle_case
    from tests.test_tas_cycle_engine import _same
    cfg, snap, heads, ct, _ = random_tas_cycle_case(200_000 + seed, rich_groups=True, fair=seed % 5 == 4, tight=seed % 2 == 0, preemption=seed % 3 != 0, partial=seed % 7 == 0)
    oracle.derive(snap)
    _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


@pytest.mark.parametrize("seed", range(2000))
def test_grouped_tas_cycles_emulated(oracle, seed):
    from tests.test_tas_cycle_engine import _emu
    _tas_cycle(oracle, _emu, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(8))
def test_grouped_tas_cycles_gpu(oracle, block):
    from tests.test_tas_cycle_engine import _hip
    for seed in range(block * 250, block * 250 + 250):
        _tas_cycle(oracle, _hip, seed)

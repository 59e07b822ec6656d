"""PodSetGroupName groups on the pending side: the resident set keeps kq_heads.ps_group, Heads() hands it to the cycle, and the lean nominate
pass defers the grouped heads to the full pass (flavorassigner.go:782-860 inside the loop of cache/queue manager.go:903-949). Closed loops of
tests/groupgen.py populations (2-5 podsets per workload, consecutive runs sharing a group, members without requests) — Heads(), every decision
and the queue states against the oracle's queues + cycle, on the 1-lane emulation and on the HIP engine."""
import numpy as np
import pytest

from kueue_amd.api import Pending
from tests.groupgen import grouped_case
from tests.test_pending import closed_loop


def _deepen(snap, heads, k=3):
    """k copies of every pending workload (later timestamps): ClusterQueue heaps with some depth, so that the loops run several cycles."""
    import copy
    from kueue_amd.api import Heads
    wls = []
    for j in range(k):
        for w in heads.workloads:
            c = copy.deepcopy(w)
            if j:
                c.name = f"{w.name}-copy{j}"; c.creation_ts = w.creation_ts + j * 1000
                if getattr(c, "uid", None):
                    c.uid = f"{c.uid}-copy{j}"
                c.last_assignment = None
                if getattr(c, "replaces", None):
                    continue   # (one replacement per slice)
            wls.append(c)
    return Heads(snap, wls, cycle=heads.cycle)


class _Pop:
    def __init__(self, snap, heads):
        self.snapshot = snap
        self._heads = heads

    def pending(self, hashes=True):
        return Pending(self._heads, uid_rank=np.arange(self._heads.n, dtype=np.uint32))


def _run(oracle, eng_factory, seed, **kw):
    cfg, snap, heads, n_multi = grouped_case(seed, **kw)
    while heads.n == 0:   # (a seed whose population came out empty: the next one of its family instead of a skip)
        seed += 1000
        cfg, snap, heads, n_multi = grouped_case(seed, **kw)
    cyc, dec, ndec, counts = closed_loop(oracle, eng_factory, _Pop(snap, _deepen(snap, heads)), cfg, max_cycles=6, hold=2, stop_when_all_decided=False)
    assert dec > 0
    return n_multi


@pytest.mark.parametrize("seed", range(60))
def test_pending_loop_with_groups_emulated(oracle, seed):
    from tests.emu import kqe
    _run(oracle, kqe.EmuEngine, seed, fair=seed % 3 == 0, preemption=seed % 2 == 0, partial=seed % 4 == 1)


def test_groups_present():
    assert sum(grouped_case(s, fair=False, preemption=False, partial=False)[3] for s in range(10)) > 20


@pytest.mark.parametrize("seed", [15, 17, 18, 22, 26])
def test_pending_add_brings_the_first_group(oracle, seed):
    """The resident set starts without any group; grouped workloads arrive by kq_pending_add (the gathered batch grows its ps_group column)."""
    from tests.emu import kqe
    from kueue_amd.api import Decisions
    cfg, snap, heads, _ = grouped_case(seed, fair=False, preemption=False, partial=False)
    grp = heads.arrays["ps_group"]
    has = np.array([np.any(grp[heads.arrays["ps_off"][i]:heads.arrays["ps_off"][i + 1]] >= 0) for i in range(heads.n)])
    if has.all() or not has.any():
        pytest.skip("population is all grouped / all plain")
    plain, grouped = np.nonzero(~has)[0], np.nonzero(has)[0]
    order = np.concatenate([plain, grouped])
    full = Pending(heads.subset(order), uid_rank=np.arange(heads.n, dtype=np.uint32))
    first = Pending(full.heads.subset(np.arange(len(plain))), uid_rank=full.uid_rank[:len(plain)])
    more = Pending(full.heads.subset(np.arange(len(plain), heads.n)), uid_rank=full.uid_rank[len(plain):])
    eng = kqe.EmuEngine(cfg); q = oracle.PendingOracle(cfg, snap, full)
    try:
        eng.put(snap); eng.pending_put(first)
        assert eng.pending_add(more) == len(plain)
        n, nps, hw = eng.pending_heads(1)
        hb, ohw = q.heads(1)
        assert np.array_equal(hw, ohw) and n == hb.n and nps == hb.n_ps
        got = eng.run_pending(Decisions(hb, tgt_cap=max(4096, snap.n_adm)))
        want = oracle.cycle_run(cfg, snap, hb)
        assert not want.equal(got), want.equal(got)
    finally:
        eng.close(); q.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_pending_loop_with_groups_gpu(oracle, seed):
    from kueue_amd.engine import Engine
    _run(oracle, Engine, seed, fair=seed % 3 == 0, preemption=seed % 2 == 0, partial=seed % 4 == 1)


def _step_loop(oracle, monkeypatch, factory_name, seed):
    """The asynchronous step (kq_pending_step, two steps in flight) over a grouped population: tests/test_pending_step.py's loop on it."""
    import tests.test_pending_step as tps
    fair = seed % 3 == 0
    cfg, snap, heads, _ = grouped_case(seed, fair=fair, preemption=seed % 2 == 0, partial=seed % 4 == 1)
    s2 = seed
    while heads.n == 0:   # (an empty population: the next one of its family instead of a skip)
        s2 += 1000
        cfg, snap, heads, _ = grouped_case(s2, fair=fair, preemption=seed % 2 == 0, partial=seed % 4 == 1)
    heads = _deepen(snap, heads)
    pop = _Pop(snap, heads)
    pop.w_nps = np.diff(heads.arrays["ps_off"])
    pop.heads_for_cycle = lambda c: heads
    monkeypatch.setitem(tps.KINDS, "grouped", ({}, fair))
    monkeypatch.setattr(tps, "generate", lambda **kw: pop)
    monkeypatch.setattr(tps, "make_config", lambda fair_sharing=False: cfg)
    tps._loop(oracle, getattr(tps, factory_name), "grouped", depth=2, cycles=4)


@pytest.mark.parametrize("seed", range(24))
def test_step_loop_with_groups_emulated(oracle, monkeypatch, seed):
    _step_loop(oracle, monkeypatch, "_emu", seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_step_loop_with_groups_gpu(oracle, monkeypatch, seed):
    _step_loop(oracle, monkeypatch, "_hip", seed)

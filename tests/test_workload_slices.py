"""Workload slices (features.ElasticJobsViaWorkloadSlices, default on): a pending workload that replaces an admitted slice of the same job.

Reference: workloadslicing.ReplacedWorkloadSlice (scheduler.go:883: the old slice is a target from the start), the flavor pin and the delta
request of findFlavorForPodSets (flavorassigner.go:1125-1145), Assignment.append's delta usage (:1028-1035), TotalRequestsFor's count
(:261-267), FindReplacedSliceTarget (scheduler.go:492). Oracle pinned by the two slice cases of TestAssignFlavors
(tests/golden/assign_flavors_slices.yaml, hand transcription: flavors, usage and the Status strings); the engine's device code (emulated;
GPU twin below) against the oracle on those and on random cycles with slices."""
import numpy as np
import pytest

from kueue_amd import _ffi as F
from kueue_amd import messages
from kueue_amd.fixtures import load_case
from tests.conftest import load_golden
from tests.randgen import random_case

G = load_golden("assign_flavors_slices.yaml")


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: c["name"][:60])
def test_assign_flavors_slices_oracle(oracle, case):
    cfg, snap, heads = load_case(case)
    oracle.derive(snap)
    got = oracle.assign(cfg, snap, heads, 0)
    want = case["want"]
    assert got["rep_mode"] == want["repMode"], got
    gotfl = {r: [v[0], v[1], v[2]] for r, v in (got["podsets"][0] if got["podsets"] else {}).items()}
    assert gotfl == {r: list(v) for r, v in want["flavors"].items()}, got
    want_usage = {tuple(k.split("/", 1)): v for k, v in want["usage"].items() if v != 0}
    assert got["usage"] == want_usage, got


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: c["name"][:60])
def test_assign_flavors_slices_engine(oracle, case):
    """One cycle through the device code (emulated): equal to the oracle's cycle, the replaced slice reported as a target with
    KQ_REASON_REPLACED_SLICE when the head is admitted, the reference's Status strings regenerated from the reason records."""
    from tests.emu import kqe
    cfg, snap, heads = load_case(case)
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads, rsn_cap=256)
    eng = kqe.EmuEngine(cfg)
    try:
        eng.put(snap)
        got = eng.run(heads, rsn_cap=256)
    finally:
        eng.close()
    assert got.rc == 0, got.error
    assert not want.equal(got), want.equal(got)
    w = case["want"]
    if w["repMode"] == "Fit":
        assert got.a["action"][0] == F.ACT_ADMIT
        assert got.targets(0) == [(snap.adm_index["old-slice"], 4)]
    else:
        assert got.a["action"][0] == F.ACT_NONE and got.targets(0) == []
        assert messages.podset_reasons(got, 0)[0] == sorted(w["status"])


@pytest.mark.parametrize("case", G["schedule"], ids=lambda c: c["name"][:60])
def test_schedule_slice_case(oracle, case):
    """TestSchedule's workload-slice case through the oracle and the device code (emulated): the new slice is admitted on the old slice's
    flavor, the cycle's usage grows by the DELTA only, the old slice comes back as the one target with KQ_REASON_REPLACED_SLICE."""
    from tests.emu import kqe
    cfg, snap, heads = load_case(case)
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads, want_usage=True)
    eng = kqe.EmuEngine(cfg)
    try:
        eng.put(snap)
        got = eng.run(heads, want_usage=True)
    finally:
        eng.close()
    w = case["want"]
    for d in (want, got):
        assert [heads.workloads[i].name for i in range(heads.n) if d.a["action"][i] == F.ACT_ADMIT] == w["admitted"]
        assert {r: v[0] for r, v in d.flavors_of(0)[0].items()} == w["flavors"]
        assert d.targets(0) == [(snap.adm_index[w["replaced"]], 4)]
    assert not want.equal(got), want.equal(got)
    assert np.array_equal(want.usage_after, got.usage_after)
    for k, v in w["usage"].items():   # the ClusterQueue's usage after the cycle: the old slice's 10 cpu + the delta of 5
        f, r = k.split("/", 1)
        assert want.usage_after[snap.cq_index["sales"] * snap.n_fr + snap.fr(f, r)] == v


@pytest.mark.parametrize("seed", range(120))
def test_slices_random_cycles(oracle, seed):
    """Random cycles in which most heads replace an admitted workload of their ClusterQueue (with and without preemption): decisions,
    targets (the slice among them), usage and bytes equal the oracle's."""
    from tests.emu import kqe
    cfg, snap, heads = random_case(seed, fair=False, preemption=(seed % 2 == 0), partial=False, slices=True)
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads, want_usage=True, rsn_cap=2048)
    eng = kqe.EmuEngine(cfg)
    try:
        eng.put(snap)
        got = eng.run(heads, want_usage=True, rsn_cap=2048)
    finally:
        eng.close()
    assert got.rc == 0, got.error
    assert not want.equal(got), (seed, want.equal(got))
    assert np.array_equal(want.usage_after, got.usage_after), seed
    assert got.bytes == want.stats["total"], seed


def test_slices_gate_off_ignores_the_columns(oracle):
    """ElasticJobsViaWorkloadSlices off: ReplacedWorkloadSlice returns nil, the head is an ordinary workload."""
    from kueue_amd.api import gates_with, make_config
    from tests.emu import kqe
    n_with = 0
    for seed in range(20):
        cfg, snap, heads = random_case(seed, fair=False, preemption=True, slices=True)
        if "slice_row" not in heads.arrays:
            continue
        n_with += 1
        cfg = make_config(gates=gates_with({"ElasticJobsViaWorkloadSlices": False}))
        oracle.derive(snap)
        want = oracle.cycle_run(cfg, snap, heads)
        plain = {k: v for k, v in heads.arrays.items() if not k.startswith(("slice_", "ps_slice", "req_slice"))}
        from kueue_amd.api import Heads
        want2 = oracle.cycle_run(cfg, snap, Heads.from_arrays(snap, plain, cycle=heads.cycle))
        assert not want.equal(want2)
        eng = kqe.EmuEngine(cfg)
        try:
            eng.put(snap)
            got = eng.run(heads)
        finally:
            eng.close()
        assert not want.equal(got), (seed, want.equal(got))
    assert n_with > 5


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_slices_random_cycles_gpu(oracle, seed):
    from kueue_amd.engine import Engine
    cfg, snap, heads = random_case(1000 + seed, fair=False, preemption=(seed % 2 == 0), partial=False, slices=True)
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads, want_usage=True, rsn_cap=2048)
    eng = Engine(cfg)
    try:
        eng.put(snap)
        got = eng.run(heads, rsn_cap=2048)
        assert not want.equal(got), (seed, want.equal(got))
        assert np.array_equal(want.usage_after, eng.usage_after()), seed
    finally:
        eng.close()


# ---- workload slices in the RESIDENT pending set (kq_pending_put / kq_pending_add; VERDICT r03 "missing" 7) -------------------------------
def _pending_slices_loop(oracle, eng_factory, seed, patch=False, arrivals=False, late=False):
    """The heads of a random slice case as the pending workloads of the queues: every cycle's Heads(), decisions and targets against the
    oracle on the same batch; optionally rows leave the snapshot between two cycles (kq_snapshot_patch_rows: the slices the pending
    workloads replace follow the move) and half of the workloads arrive later (kq_pending_add)."""
    import copy
    from kueue_amd.api import Decisions, Heads, Pending
    cfg, snap, heads = random_case(seed, fair=False, preemption=(seed % 2 == 0), partial=False, slices=True)
    if "slice_row" not in heads.arrays or heads.n < 2:
        return 0
    oracle.derive(snap)
    n0 = heads.n // 2 if arrivals else heads.n
    first = heads.subset(np.arange(n0)) if arrivals else heads
    if arrivals and late:   # the resident set starts without the slice columns: the first arrivals that replace a slice bring them
        first = Heads.from_arrays(snap, {k: v for k, v in first.arrays.items() if not k.startswith(("slice_", "ps_slice", "req_slice"))}, cycle=first.cycle)
    eng = eng_factory(cfg)
    q = oracle.PendingOracle(cfg, snap, Pending(heads))
    if arrivals:
        q.close(); q = oracle.PendingOracle(cfg, snap, Pending(first))
    osnap = snap
    checked = 0
    try:
        eng.put(snap)
        eng.pending_put(Pending(first))
        if arrivals:
            more = Pending(heads.subset(np.arange(n0, heads.n)), uid_rank=np.arange(n0, heads.n))
            assert eng.pending_add(more) == n0
            q.add(more)
        for cyc in range(1, 6):
            n, nps, hw = eng.pending_heads(cyc)
            hb, ohw = q.heads(cyc)
            assert np.array_equal(hw, ohw), (seed, cyc)
            if n == 0:
                break
            got = eng.run_pending(Decisions(hb, tgt_cap=max(64, osnap.n_adm)))
            want = oracle.cycle_run(cfg, osnap, hb)
            assert not want.equal(got), (seed, cyc, want.equal(got))
            checked += int((hb.arrays["slice_row"] >= 0).sum()) if "slice_row" in hb.arrays else 0
            eng.pending_apply()
            q.apply(hb, want)
            if patch and cyc == 1 and osnap.n_adm > 2:
                # every third admitted row leaves (finished workloads): the rows behind them move up, a replaced slice that left is gone
                gone = np.arange(0, osnap.n_adm, 3)
                res = eng.patch_rows(remove_rows=gone)
                if isinstance(res, tuple):   # (the emulated engine returns its rc as well)
                    assert res[0] == 0
                    res = res[1]
                new_index = np.asarray(res)[:osnap.n_adm]
                keep = np.setdiff1d(np.arange(osnap.n_adm), gone)
                osnap = osnap.with_rows(keep)
                oracle.derive(osnap)
                q.remap_rows(new_index, osnap)
    finally:
        eng.close(); q.close()
    return checked


@pytest.mark.parametrize("seed", range(60))
def test_pending_set_with_slices_emulated(oracle, seed):
    from tests.emu import kqe
    _pending_slices_loop(oracle, kqe.EmuEngine, seed, patch=seed % 3 == 0, arrivals=seed % 4 == 1, late=seed % 8 == 5)


def test_pending_set_with_slices_sees_slices(oracle):
    from tests.emu import kqe
    assert sum(_pending_slices_loop(oracle, kqe.EmuEngine, s, patch=s % 3 == 0, arrivals=s % 4 == 1) for s in range(30)) > 20


@pytest.mark.gpu
def test_pending_set_with_slices_gpu(oracle):
    from kueue_amd.engine import Engine
    assert sum(_pending_slices_loop(oracle, Engine, 2000 + s, patch=s % 3 == 0, arrivals=s % 4 == 1, late=s % 8 == 5) for s in range(40)) > 20

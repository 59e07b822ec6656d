"""kq_cycle_run_tas — Topology-Aware Scheduling INSIDE the engine's scheduling cycle (include/kq_cycle_tas.h, kq_tas_cycle.hpp).
 * the whole-cycle tables of pkg/scheduler/scheduler_tas_test.go (TestScheduleForTAS :58, TestScheduleForTASPreemption :4121,
   TestScheduleForTASCohorts :5950; tests/golden/schedule_tas.yaml) through the device code: the same checks against the Go expectations
   as tests/test_oracle_schedule_tas.py, and every decision array, TopologyAssignment and the leaf usage after the cycle equal to the oracle's;
 * random TAS populations (tests/tasgen_cycle.py): engine == oracle on everything, with and without preemption, partial admission,
   implied TAS (TAS-only queues), slices, podset groups;
 * without TAS flavors the entry point is the ordinary cycle.
CPU suite: the 1-lane emulation of the same device code (tests/emu). GPU suite (-m gpu): the HIP engine through the C ABI."""
import copy

import numpy as np
import pytest

from kueue_amd.tas_cycle import CycleTAS, load_tas_case
from tests.conftest import load_golden
from tests.randgen import random_case
from tests.tasgen_cycle import random_second_pass_case, random_tas_cycle_case
from tests.test_oracle_schedule_tas import check_case

CASES = load_golden("schedule_tas.yaml")["cases"]
MANUAL = load_golden("schedule_tas_manual.yaml")   # flavorassigner_test.go tables transcribed as whole cycles (each case cites its line)
_LAST = {}


def _emu(cfg):
    from tests.emu import kqe
    return kqe.EmuEngine(cfg)


def _hip(cfg):
    from kueue_amd.engine import Engine
    return Engine(cfg)


class _AsOracle:
    """check_case() drives an object with the oracle's interface: derive + cycle_run_tas."""

    def __init__(self, oracle, make):
        self.oracle, self.make = oracle, make

    def derive(self, snap):
        self.oracle.derive(snap)

    def cycle_run_tas(self, cfg, snap, heads, ct, tgt_cap=None):
        eng = self.make(cfg)
        eng.put(snap)
        d, out = eng.run_tas(heads, ct, tgt_cap=tgt_cap)
        assert getattr(d, "rc", 0) == 0, (getattr(d, "rc", 0), getattr(d, "error", ""))
        eng.close()
        return d, out


def _same(oracle, make, cfg, snap, heads, ct, tgt_cap=None):
    rsn_cap = 64 * max(heads.n_ps, 1)   # reason records too: the operands of every "why not Fit", the TAS placement's among them
    want, wout = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=tgt_cap, rsn_cap=rsn_cap)
    eng = make(cfg)
    eng.put(snap)
    try:
        got, gout = eng.run_tas(heads, ct, tgt_cap=tgt_cap, rsn_cap=rsn_cap)
    except Exception as ex:   # the HIP engine raises on a refused cycle, the emulation returns the code
        assert want.tas_stats["unsupported"] and getattr(ex, "code", -4) == -4, ex
        return None
    finally:
        eng.close()
    if want.tas_stats["unsupported"]:
        assert getattr(got, "rc", 0) != 0 or got.tas_stats["unsupported"], "two TAS flavors in one workload must be refused"
        return None
    assert getattr(got, "rc", 0) == 0, (getattr(got, "rc", 0), getattr(got, "error", ""))
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k][:12], got.a[k][:12]) for k in bad})
    n_ps = heads.n_ps
    assert np.array_equal(wout.a["ps_tas"][:n_ps], gout.a["ps_tas"][:n_ps]), (wout.a["ps_tas"][:n_ps], gout.a["ps_tas"][:n_ps])
    assert np.array_equal(wout.a["dom_off"], gout.a["dom_off"]), (wout.a["dom_off"], gout.a["dom_off"])
    m = int(wout.a["dom_off"][n_ps])
    assert np.array_equal(wout.a["dom_leaf"][:m], gout.a["dom_leaf"][:m]) and np.array_equal(wout.a["dom_count"][:m], gout.a["dom_count"][:m])
    assert np.array_equal(wout.a["tas_usage_after"], gout.a["tas_usage_after"])
    assert want.tas_stats["recomputes"] == got.tas_stats["recomputes"]
    _LAST["class_hits"] = _LAST.get("class_hits", 0) + got.tas_stats.get("class_hits", 0)
    return want


def _golden(oracle, make, case):
    check_case(_AsOracle(oracle, make), case)                       # the Go expectations, through the device code
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))  # and everything else against the oracle


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:80])
def test_schedule_tas_emulated(oracle, case):
    _golden(oracle, _emu, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:80])
def test_schedule_tas_gpu(oracle, case):
    _golden(oracle, _hip, case)


def _bookmark(oracle, make, case):
    """TestFlavorScanRecordsLastTriedFlavorIdx / TestRecomputeRecordsLastTriedFlavorIdx: what LastState.LastTriedFlavorIdx holds for each
    shape the flavor scan takes with TAS on — the quota scan writes the bookmark before the placement runs, and a recomputation under
    the nomination mapping writes it again. The Go expectation first, then everything else against the oracle."""
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    want = _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))
    eng = make(cfg)
    try:
        eng.put(snap)
        got, gout = eng.run_tas(heads, ct, tgt_cap=max(16, snap.n_adm))
    finally:
        eng.close()
    e = case["expectAssignment"]
    i = [w["name"] for w in case["pending"]].index(e["head"])
    hi = list(heads.names).index(e["head"]) if hasattr(heads, "names") else i
    ps0 = int(heads.arrays["ps_off"][hi])
    assert int(got.a["mode"][hi]) == MANUAL["modes"][e["mode"]], (case["name"], int(got.a["mode"][hi]))
    nR = snap.n_resource
    for res, idx in e["triedIdx"].items():
        r = list(snap.resources).index(res)
        assert int(got.a["tried_idx"][ps0 * nR + r]) == idx, (case["name"], res, got.a["tried_idx"].tolist())
    if "plan" in e:
        assert (int(gout.a["ps_tas"][ps0]) >= 0) == e["plan"], case["name"]
    assert want is not None


@pytest.mark.parametrize("case", MANUAL["cases"], ids=lambda c: c["name"][:80])
def test_flavor_scan_bookmark_emulated(oracle, case):
    _bookmark(oracle, _emu, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", MANUAL["cases"], ids=lambda c: c["name"][:80])
def test_flavor_scan_bookmark_gpu(oracle, case):
    _bookmark(oracle, _hip, case)


def _random(oracle, make, seed):
    cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=False, tight=seed % 2 == 0, preemption=seed % 3 != 0)
    oracle.derive(snap)
    return _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


@pytest.mark.parametrize("seed", range(300))
def test_random_tas_cycles_emulated(oracle, seed):
    _random(oracle, _emu, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(6))
def test_random_tas_cycles_gpu(oracle, block):
    for seed in range(block * 50, block * 50 + 50):
        _random(oracle, _hip, seed)


def _random_masked(oracle, make, seed):
    """Node feasibility inside the cycle (kq_cycle_tas.ps_mask / leaf_mask: taints vs tolerations, nodeSelector, required affinity —
    tas_flavor_snapshot.go:955-963): ~45 % of the podsets carry a mask per TAS flavor, rows shared; the masked placements of nominate,
    of the victim search's workloadFits and of processEntry's recomputation start from a phase 1 of their own."""
    cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=seed % 4 == 3, tight=seed % 2 == 0, preemption=seed % 3 != 0, partial=seed % 7 == 0, masks=True)
    if seed % 5 == 4:
        for i in range(len(ct.topos)):
            ct._topo_arr[i].profile_mixed |= 2   # KQ_TAS_F_BALANCED_PLACEMENT
    oracle.derive(snap)
    return _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


@pytest.mark.parametrize("seed", range(400))
def test_random_tas_cycles_with_node_masks_emulated(oracle, seed):
    _random_masked(oracle, _emu, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(4))
def test_random_tas_cycles_with_node_masks_gpu(oracle, block):
    for seed in range(block * 100, block * 100 + 100):
        _random_masked(oracle, _hip, seed)


def test_node_masks_change_the_outcome(oracle):
    """... and the masks are seen: against the same cycles without them, decisions or TopologyAssignments differ in a good share of the seeds."""
    differ = masked_ps = 0
    for seed in range(240):
        out = []
        for masks in (False, True):
            cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=False, tight=seed % 2 == 0, preemption=seed % 3 != 0, masks=masks)
            oracle.derive(snap)
            d, t = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))
            out.append((d.a["action"].tobytes(), t.a["dom_off"].tobytes(), t.a["dom_leaf"].tobytes()))
            if masks and "ps_mask" in ct.arrays:
                masked_ps += int((ct.arrays["ps_mask"] >= 0).sum())
        differ += out[0] != out[1]
    assert differ >= 20 and masked_ps >= 250, (differ, masked_ps)


def _second_pass_masked(oracle, make, seed):
    cfg, snap, heads, ct, n_second = random_second_pass_case(seed, fair=seed % 5 == 4, masks=True)
    oracle.derive(snap)
    _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


@pytest.mark.parametrize("seed", range(200))
def test_random_second_pass_cycles_with_node_masks_emulated(oracle, seed):
    """the failed node's replacement (findReplacementAssignment :686) looks at the podset's feasible nodes only"""
    _second_pass_masked(oracle, _emu, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(2))
def test_random_second_pass_cycles_with_node_masks_gpu(oracle, block):
    for seed in range(block * 100, block * 100 + 100):
        _second_pass_masked(oracle, _hip, seed)


def test_node_mask_arguments_are_checked(oracle):
    """ps_mask without rows, a row index past n_masks, a stride below a topology's leaf count: KQ_EINVAL, not a read past the array"""
    from kueue_amd import _ffi as F
    for seed in range(40):
        cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=False, masks=True)
        if "ps_mask" in ct.arrays and max(t.n_leaves for t in ct.topos) > 1:
            break
    oracle.derive(snap)
    for field, bad in (("n_masks", 0), ("mask_stride", 1), ("n_masks", int(ct.arrays["ps_mask"].max()))):
        keep = getattr(ct._struct, field)
        setattr(ct._struct, field, bad)
        eng = _emu(cfg)
        eng.put(snap)
        d, _ = eng.run_tas(heads, ct, tgt_cap=max(16, snap.n_adm))
        assert d.rc == F.KQ_EINVAL, (field, bad, d.rc, d.error)
        eng.close()
        setattr(ct._struct, field, keep)


def _random_balanced(oracle, make, seed):
    """The whole cycle with features.TASBalancedPlacement on: every placement of a preferred request inside Assign, the victim search's
    workloadFits and processEntry's recomputation goes through tas_balanced_placement.go (kq_tas_device.hpp t_balanced_lane0)."""
    cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=seed % 6 == 5, tight=seed % 2 == 0, preemption=seed % 3 != 0)
    for i in range(len(ct.topos)):
        ct._topo_arr[i].profile_mixed |= 2   # KQ_TAS_F_BALANCED_PLACEMENT
    oracle.derive(snap)
    return _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


@pytest.mark.parametrize("seed", range(200))
def test_random_tas_cycles_balanced_placement_emulated(oracle, seed):
    _random_balanced(oracle, _emu, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(3))
def test_random_tas_cycles_balanced_placement_gpu(oracle, block):
    for seed in range(block * 100, block * 100 + 100):
        _random_balanced(oracle, _hip, seed)


def _second_pass(oracle, make, seed):
    """Cycles that mix heads on their first pass with heads on their second pass after a node failure (scheduler.go:583, manager.go:923,
    tas_flavor_snapshot.go:608-633): the admission's flavors kept, the failed node's pods placed below the required replacement domain and
    merged in, net leaf usage, eviction when no replacement exists (TASFailedNodeReplacementFailFast) — everything equal to the oracle's."""
    cfg, snap, heads, ct, n_second = random_second_pass_case(seed, fair=seed % 5 == 4)
    oracle.derive(snap)
    _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


@pytest.mark.parametrize("seed", range(400))
def test_random_second_pass_cycles_emulated(oracle, seed):
    _second_pass(oracle, _emu, seed)


def test_random_second_pass_cycles_cover_every_outcome(oracle):
    """what the 400 cycles above are made of: replacements admitted, evictions (fail fast), entries left pending, entries that stopped fitting"""
    from kueue_amd import _ffi as F
    seen = {}
    for seed in range(200):
        cfg, snap, heads, ct, n_second = random_second_pass_case(seed, fair=seed % 5 == 4)
        oracle.derive(snap)
        d, _ = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))
        for i in range(n_second):
            k = (int(d.a["action"][i]), int(d.a["status"][i]))
            seen[k] = seen.get(k, 0) + 1
    for k in ((F.ACT_ADMIT, F.ST_ASSUMED), (F.ACT_EVICT, F.ST_EVICTED), (F.ACT_NONE, F.ST_NOT_NOMINATED), (F.ACT_NONE, F.ST_SKIPPED)):
        assert seen.get(k, 0) >= 10, (k, seen)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(6))
def test_random_second_pass_cycles_gpu(oracle, block):
    for seed in range(block * 50, block * 50 + 50):
        _second_pass(oracle, _hip, seed)


@pytest.mark.parametrize("seed", range(40))
def test_without_tas_flavors_it_is_the_plain_cycle_emulated(oracle, seed):
    cfg, snap, heads = random_case(seed, fair=False, tight=seed % 2 == 0)
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads)
    eng = _emu(cfg)
    eng.put(snap)
    got, out = eng.run_tas(heads, CycleTAS(snap, heads, {}, {}), tgt_cap=want.a["tgt_adm"].size if "tgt_adm" in want.a else None)
    eng.close()
    assert got.rc == 0 and not want.equal(got)
    assert (out.a["ps_tas"][:heads.n_ps] == -1).all()


def _random_fair(oracle, make, seed):
    """TAS inside a fair-sharing cycle: the fair iterator over every root tree interleaved (kq_tas_cycle.hpp process_all_fair_tas), fair
    preemption with the leaf usage following the victims (f_apply_row / f_fits hooks)."""
    cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=True, tight=seed % 2 == 0, preemption=seed % 3 != 0, partial=seed % 7 == 0)
    oracle.derive(snap)
    return _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


@pytest.mark.parametrize("seed", range(900))
def test_random_fair_tas_cycles_emulated(oracle, seed):
    _random_fair(oracle, _emu, 5000 + seed)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(6))
def test_random_fair_tas_cycles_gpu(oracle, block):
    for seed in range(block * 100, block * 100 + 100):
        _random_fair(oracle, _hip, 5000 + seed)


def _population(oracle, make, n_cq, n_pending, fair=False, **topo):
    """BASELINE configs[4] as whole cycles (kueue_amd/tas_population.py generate_tas_cycle): one TAS flavor shared by every ClusterQueue, so
    most entries lose their leaves to an earlier entry and are recomputed inside processEntry — the path of the resident request-class
    tables (kq_tas_cycle.hpp) and of their incremental update after every AddUsage."""
    from kueue_amd.api import make_config
    from kueue_amd.tas_population import generate_tas_cycle
    snap, _, batch = generate_tas_cycle(n_cq=n_cq, n_pending=n_pending, **topo)
    cfg = make_config(fair_sharing=fair)
    oracle.derive(snap)
    rec = 0
    for c in range((n_pending + n_cq - 1) // n_cq):
        heads, ct = batch(c)
        want = _same(oracle, make, cfg, snap, heads, ct)
        rec += want.tas_stats["recomputes"]
    return rec, _LAST.get("class_hits", 0)


@pytest.mark.parametrize("classes_off", [False, True])
def test_tas_cycle_population_emulated(oracle, classes_off, monkeypatch):
    if classes_off:
        monkeypatch.setenv("KQ_TAS_CLASSES_OFF", "1")
    _LAST.clear()
    rec, hits = _population(oracle, _emu, 120, 360, blocks=2, racks=4, hosts=16)
    assert rec > 100   # the recomputation chain is what this test is about
    assert (hits == 0) if classes_off else (hits >= rec)


@pytest.mark.gpu
@pytest.mark.parametrize("classes_off", [False, True])
def test_tas_cycle_population_gpu(oracle, classes_off, monkeypatch):
    if classes_off:
        monkeypatch.setenv("KQ_TAS_CLASSES_OFF", "1")
    _LAST.clear()
    rec, hits = _population(oracle, _hip, 400, 800)
    assert rec > 300
    assert (hits == 0) if classes_off else (hits >= rec)


@pytest.mark.parametrize("mode", ["lds-off", "coop-1"])
def test_tas_cycle_population_state_modes_emulated(oracle, mode, monkeypatch):
    """k_process_tas keeps a class-path placement's working state in LDS and shares long sweeps with its helper waves: the same
    population with the state in the slot's global rows (a topology that does not fit falls back to them) and with every slice
    shared, however short."""
    monkeypatch.setenv(*{"lds-off": ("KQ_TAS_LDS_OFF", "1"), "coop-1": ("KQ_TAS_COOP_MIN", "1")}[mode])
    _LAST.clear()
    rec, hits = _population(oracle, _emu, 120, 360, blocks=2, racks=4, hosts=16)
    assert rec > 100 and hits >= rec
    for seed in range(60):
        _random(oracle, _emu, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["lds-off", "coop-1"])
def test_tas_cycle_population_state_modes_gpu(oracle, mode, monkeypatch):
    monkeypatch.setenv(*{"lds-off": ("KQ_TAS_LDS_OFF", "1"), "coop-1": ("KQ_TAS_COOP_MIN", "1")}[mode])
    _LAST.clear()
    rec, hits = _population(oracle, _hip, 400, 800)
    assert rec > 300 and hits >= rec
    for seed in range(60):
        _random(oracle, _hip, seed)


def test_fair_tas_cycle_population_emulated(oracle):
    """The same population under fair sharing: ten root trees whose iterators are interleaved by the canonical getCq, the recomputation
    chain over the shared leaves."""
    _LAST.clear()
    rec, _ = _population(oracle, _emu, 120, 360, fair=True, blocks=2, racks=4, hosts=16)
    assert rec > 100


@pytest.mark.gpu
def test_fair_tas_cycle_population_gpu(oracle):
    _LAST.clear()
    rec, _ = _population(oracle, _hip, 400, 800, fair=True)
    assert rec > 300


def test_dom_cap_too_small_is_reported(oracle):
    """The caller's TopologyAssignment arrays are sized by dom_cap: a cycle that needs more says KQ_ECAPACITY, it does not truncate."""
    from kueue_amd.api import make_config
    from kueue_amd.tas_population import generate_tas_cycle
    snap, _, batch = generate_tas_cycle(n_cq=20, n_pending=20, blocks=1, racks=2, hosts=8)
    oracle.derive(snap)
    heads, ct = batch(0)
    eng = _emu(make_config())
    eng.put(snap)
    ok, out = eng.run_tas(heads, ct)
    assert ok.rc == 0 and int(out.a["dom_off"][heads.n_ps]) > 1
    bad, _ = eng.run_tas(heads, ct, dom_cap=1)
    eng.close()
    assert bad.rc == -5   # KQ_ECAPACITY


def test_tas_failure_message_of_the_reference(oracle):
    """scheduler_tas_test.go "workload does not get scheduled as it does not fit within the node capacity": the reference's status message is
    `couldn't assign flavors to pod set one: topology "tas-single-level" allows to fit only 1 out of 2 pod(s)` — regenerated from the
    KQ_RSN_TAS_FAILURE record of the device code (flavorassigner.go:875; kueue_amd/messages.py tas_failure_text)."""
    from kueue_amd import messages as M
    case = next(c for c in CASES if c["name"] == "workload does not get scheduled as it does not fit within the node capacity")
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    eng = _emu(cfg)
    eng.put(snap)
    d, _ = eng.run_tas(heads, ct, rsn_cap=64)
    eng.close()
    assert d.rc == 0
    recs = [k for k in range(int(d.a["rsn_off"][0]), int(d.a["rsn_off"][1])) if int(d.a["rsn_code"][k]) == M.RSN_TAS_FAILURE]
    assert len(recs) == 1
    tas = lambda ps, fl, st, a, b: M.tas_failure_text("tas-single-level", st, a, b, int(ct.arrays["ps_slice_size"][ps]))
    names = [ps.name for ps in heads.workloads[0].pod_sets]
    reasons = M.podset_reasons(d, 0, tas=tas)
    msg = "; ".join(f"couldn't assign flavors to pod set {n}: " + ", ".join(r) for n, r in zip(names, reasons) if r)
    assert msg == 'couldn\'t assign flavors to pod set one: topology "tas-single-level" allows to fit only 1 out of 2 pod(s)'


def test_second_pass_failure_messages():
    """findReplacementAssignment's own two failures (tas_flavor_snapshot.go:695, :728) from the operands of the cycle's record: the names come
    from the head's Status, which the caller holds (messages.second_pass_names)."""
    from kueue_amd import messages as M
    from kueue_amd import tas as T
    from kueue_amd.tas_cycle import HeadAdmission
    adm = HeadAdmission(flavors=[{"cpu": "tas"}], domains=[[(("b1", "r1", "x0"), 1), (("b1", "r1", "x1"), 2), (("b2", "r9", "gone"), 1)]], unhealthy_nodes=["x0"])
    stale, node = M.second_pass_names(adm, 0, 1)
    assert (stale, node) == ("b2", "x0")
    assert M.tas_failure_text("t", T.TAS_STALE, 1, 0, stale_domain=stale, unhealthy_node=node) == \
        "Cannot replace the node, because the existing topologyAssignment is invalid, as it contains the stale domain b2"
    assert M.tas_failure_text("t", T.TAS_NO_REPLACEMENT, 0, 0, stale_domain=stale, unhealthy_node=node) == "cannot find replacement assignment for unhealthy node: x0"
    # and on a cycle: the random campaign's seed 1 holds a head whose kept domains name a node the snapshot no longer has
    cfg, snap, heads, ct, *_ = random_second_pass_case(1)
    eng = _emu(cfg)
    eng.put(snap)
    d, _ = eng.run_tas(heads, ct, rsn_cap=4096, tgt_cap=max(16, snap.n_adm))
    eng.close()
    assert d.rc == 0
    seen = 0
    for h in range(heads.n):
        for k in range(int(d.a["rsn_off"][h]), int(d.a["rsn_off"][h + 1])):
            if int(d.a["rsn_code"][k]) == M.RSN_TAS_FAILURE and int(d.a["rsn_a"][k]) == T.TAS_STALE:
                ps = int(d.a["rsn_podset"][k]) - int(heads.arrays["ps_off"][h])
                adm = ct.head_admission[heads.workloads[h].name]
                stale, node = M.second_pass_names(adm, ps, int(d.a["rsn_b"][k]))
                kept = [v for v, _ in adm.domains[ps] if v[-1] != node]
                assert stale and stale == kept[int(d.a["rsn_b"][k])][0]
                assert M.tas_failure_text("t", T.TAS_STALE, int(d.a["rsn_b"][k]), 0, stale_domain=stale).endswith("the stale domain " + stale)
                seen += 1
    assert seen


def test_tas_request_without_any_tas_flavor(oracle):
    """Found by the random campaign (seed 1607): a snapshot without TAS flavors but a podset that asks for TAS — WorkloadsTopologyRequests
    still runs and turns the assignment into NoFit (ErrNoTASFlavorAssigned, tas_flavorassigner.go:60-66); the entry point must not take
    its "no TAS flavor: ordinary cycle" shortcut then."""
    cfg, snap, heads, ct, _ = random_tas_cycle_case(1607, fair=False, tight=False, preemption=True)
    assert not ct.names and (ct.arrays["ps_flags"] & 1).any()
    oracle.derive(snap)
    _same(oracle, _emu, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


@pytest.mark.parametrize("seed", [21190, 27235])
def test_full_assignment_keeps_the_pserror_get_targets_left(oracle, seed):
    """Found by the random campaign: a TAS-only ClusterQueue, a head whose full assignment is Preempt with a podset Assign stopped at
    (no flavor, no reason), no victims, partial admission finds nothing -> getInitialAssignments returns the SAME fullAssignment that
    GetTargets already looked at (scheduler.go:897, :923): WorkloadsTopologyRequests left a psError on the flavorless podset
    (flavorassigner.go:290), so updateAssignmentForTAS sees NoFit and places nothing. The device code regenerates the outputs of the
    full assignment after the search; it must carry that status along (no placement, no topology assignment)."""
    cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=False, tight=seed % 2 == 0, preemption=seed % 3 != 0, partial=seed % 5 == 0)
    oracle.derive(snap)
    want, wout = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm), rsn_cap=64 * max(heads.n_ps, 1))
    eng = _emu(cfg); eng.put(snap)
    got, gout = eng.run_tas(heads, ct, tgt_cap=max(16, snap.n_adm), rsn_cap=64 * max(heads.n_ps, 1))
    eng.close()
    assert got.tas_stats["finds"] == want.tas_stats["finds"]
    _same(oracle, _emu, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))


# ---- TestScheduleForPreserveFlavorScanProgress (pkg/scheduler/scheduler_preserve_flavor_scan_progress_test.go:82) --------------------
# Four real scheduling cycles of a fair-sharing ClusterQueue with two TAS flavors: "blocker" fills flavor-1's only node, "pending" needs a
# whole node, quota alone keeps selecting flavor-1 (4 cpu of quota, 2 cpu of node), TAS turns that into Preempt, nothing can be
# preempted (equal priorities), and the bookmark of the quota scan is what may carry "pending" to flavor-2 in the next cycle — unless
# AllocatableResourceGeneration advanced in between and FlavorFungibilityPreserveScanProgress is off. Between the cycles this driver does
# what the scheduler and the cache do: an admitted workload joins the snapshot with its TopologyAssignment (cache.AssumeWorkload), a
# requeued one keeps Assignment.LastState (scheduler.go:459-464), the churn bumps the ClusterQueue's generation.
_PP_NODES = [{"name": f"node-f{i}", "labels": {"tas-node": "true", "tas-flavor": f"f{i}", "kubernetes.io/hostname": f"node-f{i}"},
              "allocatable": {"cpu": "2", "pods": "10"}, "ready": True} for i in (1, 2)]


def _preserve_progress(oracle, make, gate, churn, cycles=4):
    from kueue_amd import _ffi as F
    wl = {n: {"name": f"default/{n}", "cq": "tas-cq", "priority": 10, "created": created,
              "podsets": [dict({"name": "one", "count": 1, "requests": {"cpu": "2"}, "topologyRequest": {"required": "kubernetes.io/hostname"}}, **extra)]}
          for n, created, extra in (("blocker", -60_000_000_000, {"excludedFlavors": ["tas-flavor-2"]}), ("pending", 0, {}))}
    admitted, waiting, gen, landed = [], ["blocker", "pending"], 0, {}
    for cyc in range(1, cycles + 1):
        if not waiting:
            break
        head = copy.deepcopy(wl[waiting[0]])   # one ClusterQueue: Heads() is its first workload (earlier creation first, equal priorities)
        case = {"name": "preserve", "now": 1_000_000_000_000, "nodes": _PP_NODES, "topologies": {"tas-single-level": ["kubernetes.io/hostname"]},
                "resourceFlavors": [{"name": f"tas-flavor-{i}", "nodeLabels": {"tas-flavor": f"f{i}"}, "topologyName": "tas-single-level"} for i in (1, 2)],
                "clusterQueues": [{"name": "tas-cq", "generation": gen, "preemption": {"withinClusterQueue": "LowerPriority", "reclaimWithinCohort": "Any"},
                                   "resourceGroups": [[{"flavor": "tas-flavor-1", "resources": {"cpu": ["4", "", ""]}},
                                                       {"flavor": "tas-flavor-2", "resources": {"cpu": ["5" if churn and cyc % 2 == 0 else "4", "", ""]}}]]}],
                "cohorts": [], "admitted": copy.deepcopy(admitted), "pending": [head], "notHeads": [], "expect": {}, "fairSharing": True,
                "gates": {"FlavorFungibilityPreserveScanProgress": gate}, "gatesGo": {}}
        cfg, snap, heads, ct = load_tas_case(case, cycle=cyc)
        oracle.derive(snap)
        want = _same(oracle, make, cfg, snap, heads, ct, tgt_cap=16)
        assert want is not None
        d, out = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=16)
        name = waiting[0]
        if int(d.a["action"][0]) == F.ACT_ADMIT:
            fl = d.flavors_of(0)[0]["cpu"][0]
            ta = out.topology_assignment(0, 0)
            w = copy.deepcopy(wl[name])
            w["podsets"][0].update(flavors={"cpu": fl}, topologyAssignment={"levels": ["kubernetes.io/hostname"], "domains": [[list(v), c] for v, c in ta[1]]})
            admitted.append(w)
            landed[name] = fl
            waiting.pop(0)
        else:  # requeued with what the (recomputed) assignment recorded
            wl[name]["lastAssignment"] = {"lastTriedFlavorIdx": [{"cpu": d.flavors_of(0)[0]["cpu"][2]}], "generation": gen, "cycle": cyc}
        if churn:
            gen += 1   # updateQuotasAndResourceGroups: flavor-2's quota really changes every cycle
    return landed


@pytest.mark.parametrize("gate,churn,want", [(False, True, None), (True, True, "tas-flavor-2"), (False, False, "tas-flavor-2"), (True, False, "tas-flavor-2")],
                         ids=["gate disabled", "gate enabled", "no generation churn, gate disabled", "no generation churn, gate enabled"])
def test_preserve_flavor_scan_progress_emulated(oracle, gate, churn, want):
    landed = _preserve_progress(oracle, _emu, gate, churn)
    assert landed.get("blocker") == "tas-flavor-1" and landed.get("pending") == want, landed


@pytest.mark.gpu
@pytest.mark.parametrize("gate,churn,want", [(False, True, None), (True, True, "tas-flavor-2"), (False, False, "tas-flavor-2"), (True, False, "tas-flavor-2")],
                         ids=["gate disabled", "gate enabled", "no generation churn, gate disabled", "no generation churn, gate enabled"])
def test_preserve_flavor_scan_progress_gpu(oracle, gate, churn, want):
    landed = _preserve_progress(oracle, _hip, gate, churn)
    assert landed.get("blocker") == "tas-flavor-1" and landed.get("pending") == want, landed

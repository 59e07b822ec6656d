"""Small table tests of the reference replayed on the oracle's restated functions:
TestIsPreferred, TestResourcesToReserve, TestLastAssignmentOutdated, TestSearch of the PodSetReducer (tests/golden/small_tables.yaml,
extractor committed),
TestCandidatesOrdering, TestEntryOrdering (tests/golden/small_tables_manual.yaml, hand transcription)."""
import pytest

from kueue_amd import _ffi as F
from kueue_amd.api import (ClusterQueue, Cohort, FlavorQuotas, Heads, LastAssignment, PodSet, ResourceGroup, Snapshot, Workload,
                           gates_with, make_config)
from tests.conftest import load_golden

T = load_golden("small_tables.yaml")
M = load_golden("small_tables_manual.yaml")
PM = {"noFit": 0, "noPreemptionCandidates": 1, "preempt": 2, "reclaim": 3, "fit": 4}


@pytest.mark.parametrize("case", T["isPreferred"], ids=lambda c: c["name"][:70])
def test_is_preferred(oracle, case):
    c = case["config"]
    cq = ClusterQueue("cq", when_can_borrow=c.get("WhenCanBorrow", "MayStopSearch"), when_can_preempt=c.get("WhenCanPreempt", "TryNextFlavor"),
                      preference=c.get("Preference"))
    got = oracle.is_preferred((PM[case["a"]["mode"]], case["a"]["borrow"]), (PM[case["b"]["mode"]], case["b"]["borrow"]), cq.policy_word())
    assert got == case["want"]


def _reserve_snapshot(cq_usage):
    # the ClusterQueue of TestResourcesToReserve (scheduler_test.go:8700-8716)
    cq = ClusterQueue("cq", cohort="eng", queueing_strategy="StrictFIFO", resource_groups=[
        ResourceGroup([FlavorQuotas("on-demand").Resource("memory", "100"), FlavorQuotas("spot").Resource("memory", "0", "100")]),
        ResourceGroup([FlavorQuotas("model-a").Resource("gpu", "10", "0"), FlavorQuotas("model-b").Resource("gpu", "10", "5")])])
    for k, v in cq_usage.items():
        f, r = k.split("/")
        cq.extra_usage[(f, r)] = v
    return Snapshot([cq], [Cohort("eng")], [])


@pytest.mark.parametrize("case", T["resourcesToReserve"], ids=lambda c: c["name"][:70])
def test_resources_to_reserve(oracle, case):
    snap = _reserve_snapshot(case["cqUsage"])
    oracle.derive(snap)
    wl = Workload("wl", "cq", pod_sets=[PodSet("main", count=1)])
    heads = Heads(snap, [wl])
    usage = {tuple(k.split("/")): v for k, v in case["assignmentUsage"].items()}
    mode = {"Preempt": F.Preempt, "Fit": F.Fit, "NoFit": F.NoFit}[case["mode"]]
    got = oracle.resources_to_reserve(make_config(), snap, heads, mode, case["borrowing"], usage)
    assert got == {tuple(k.split("/")): v for k, v in case["wantReserved"].items()}


@pytest.mark.parametrize("case", T["lastAssignmentOutdated"], ids=lambda c: c["name"][:70])
def test_last_assignment_outdated(oracle, case):
    cq = ClusterQueue("cq", resource_groups=[ResourceGroup([FlavorQuotas("f").Resource("cpu", "1")])], generation=case["cqGeneration"])
    snap = Snapshot([cq], [], [])
    oracle.derive(snap)
    wl = Workload("wl", "cq", pod_sets=[PodSet("main", count=1).Request("cpu", "1")], scheduling_hash=case["hash"],
                  last_assignment=LastAssignment(last_tried_flavor_idx=[{"cpu": 0}], cluster_queue_generation=case["last"]["generation"],
                                                 scheduling_cycle=case["last"]["cycle"], scheduling_hash=case["last"]["hash"]))
    heads = Heads(snap, [wl], cycle=case["cycle"])
    cfg = make_config(gates=gates_with({"FlavorFungibilityPreserveScanProgress": case["preserveProgress"]}))
    assert oracle.last_assignment_outdated(cfg, snap, heads) == case["want"]


@pytest.mark.parametrize("case", M["candidatesOrdering"], ids=lambda c: c["name"][:70])
def test_candidates_ordering(oracle, case):
    cq_names = sorted({c["cq"] for c in case["candidates"]} | {case["preemptorCq"]})
    rg = [ResourceGroup([FlavorQuotas("f").Resource("cpu", "10")])]
    cqs = [ClusterQueue(n, cohort="co", resource_groups=rg) for n in cq_names]
    now = 1_000_000_000_000
    adm = [Workload(c["name"], c["cq"], priority=c["priority"], pod_sets=[PodSet("main", count=1, requests={"cpu": 1000}, flavors={"cpu": "f"})],
                    reserve_ts=(now + c["reservedAt"] * 1_000_000_000) if "reservedAt" in c else None, evicted=c.get("evicted", False),
                    uid=f"uid-{i}") for i, c in enumerate(case["candidates"])]
    snap = Snapshot(cqs, [Cohort("co")], adm, now_ns=now)
    oracle.derive(snap)
    rows = [snap.adm_index[c["name"]] for c in case["candidates"]]
    got = oracle.candidates_order(make_config(), snap, case["preemptorCq"], rows)
    names = {snap.adm_index[c["name"]]: c["name"] for c in case["candidates"]}
    assert [names[r] for r in got] == case["want"]


@pytest.mark.parametrize("case", M["entryOrdering"], ids=lambda c: c["name"][:70])
def test_entry_ordering(oracle, case):
    rg = [ResourceGroup([FlavorQuotas("f").Resource("cpu", "100")])]
    cqs = [ClusterQueue(f"cq{i:02d}", cohort="co", resource_groups=rg) for i in range(len(case["entries"]))]
    snap = Snapshot(cqs, [Cohort("co")], [])
    oracle.derive(snap)
    wls = [Workload(e["name"], f"cq{i:02d}", priority=e["priority"], creation_ts=e["queueTs"] * 1_000_000_000,
                    pod_sets=[PodSet("main", count=1).Request("cpu", "1")]) for i, e in enumerate(case["entries"])]
    heads = Heads(snap, wls)
    cfg = make_config(gates=gates_with({"PrioritySortingWithinCohort": case["prioritySorting"]}))
    got = oracle.entry_order(cfg, snap, heads, [e["borrowing"] for e in case["entries"]])
    assert [case["entries"][i]["name"] for i in got] == case["want"]


@pytest.mark.parametrize("case", T["podSetReducerSearch"], ids=lambda c: c["name"][:70])
def test_podset_reducer_search(oracle, case):
    """podset_reducer_test.go:27 TestSearch — the partial-admission search of getInitialAssignments (scheduler.go:906-921)."""
    counts = [p["count"] for p in case["podSets"]]
    mins = [-1 if p["minCount"] is None else p["minCount"] for p in case["podSets"]]
    got_count, got_found = oracle.podset_reducer_search(counts, mins, case["countLimit"])
    assert (got_count, got_found) == (case["wantCount"], case["wantFound"])


POLICY = {"Never": 0, "LowerPriority": 1, "LowerOrNewerEqualPriority": 2, "Any": 3}


@pytest.mark.parametrize("case", M["satisfiesPreemptionPolicy"], ids=lambda c: c["name"][:70])
def test_satisfies_preemption_policy(oracle, case):
    """preemption_policy_test.go:34 TestSatisfiesPreemptionPolicy on effective priorities (hand transcription, each case cites its line)."""
    sec = 1_000_000_000
    got = oracle.satisfies_preemption_policy((case["preemptor"][0], case["preemptor"][1] * sec), (case["candidate"][0], case["candidate"][1] * sec),
                                             POLICY[case["policy"]])
    assert got == case["want"]


@pytest.mark.parametrize("case", M["clusterQueueOrdering"], ids=lambda c: c["name"][:70])
def test_cluster_queue_ordering(oracle, case):
    """ordering_test.go:36 TestMakeClusterQueueOrdering: the DRS-guided descent that picks the next ClusterQueue to take a victim from."""
    cqs = [ClusterQueue(c["name"], cohort=c.get("cohort"), resource_groups=[ResourceGroup([FlavorQuotas("default").Resource("cpu", c["cpu"])])])
           for c in case["clusterQueues"]]
    cohorts = [Cohort(c["name"], parent=c.get("parent")) for c in case.get("cohorts", [])]
    now = 1_000_000_000_000
    adm = [Workload(w["name"], w["cq"], pod_sets=[PodSet("main", count=1, flavors={"cpu": "default"}).Request("cpu", w["cpu"])], reserve_ts=now, uid=f"uid-{i}")
           for i, w in enumerate(case["admitted"])]
    snap = Snapshot(cqs, cohorts, adm, now_ns=now)
    oracle.derive(snap)
    rows = [snap.adm_index[w["name"]] for w in case["admitted"] if w["cq"] in case["candidateCqs"]]
    got = oracle.cq_ordering(make_config(fair_sharing=True), snap, case["preemptorCq"], rows, case.get("actions", ()))
    assert got == case["want"]


def _i64(v):
    if isinstance(v, int):
        return v
    mx, mn = 2 ** 63 - 1, -2 ** 63
    return {"MAX": mx, "MIN": mn}.get(v) if v in ("MAX", "MIN") else mx - int(v.split("-")[1])


@pytest.mark.parametrize("case", M["amountArithmetic"], ids=lambda c: c["name"][:70])
def test_amount_arithmetic(oracle, case):
    """amount_test.go:148 TestAmountArithmetic: the saturating / Unlimited-aware arithmetic every quota computation rests on —
    on the oracle's Amount and on the engine's a_add / a_addi / a_sub (device source compiled for the CPU emulation)."""
    import ctypes as C
    from tests.emu import kqe
    a, b, want = _i64(case["a"]), _i64(case["b"]), _i64(case["want"])
    assert oracle.amount_op(case["op"], a, b) == want
    if case["op"] != "SubInt64":  # the engine has no SubInt64 (the path never subtracts a plain integer from an Amount)
        out = C.c_int64()
        l = kqe.lib()
        l.kqe_amount_op.argtypes = [C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        assert l.kqe_amount_op(oracle.AMOUNT_OPS[case["op"]], a, b, C.byref(out)) == 0
        assert out.value == want


@pytest.mark.parametrize("case", M["countIn"], ids=lambda c: c["name"][:70])
def test_count_in(oracle, case):
    """requests_test.go:31,130 TestCountIn / TestCountInWithLimitingResource: how many pods of a request fit into a capacity vector — the
    leaf count of TAS phase 1 (tas_flavor_snapshot.go:1899) — on the oracle and on the device function (CPU emulation)."""
    import ctypes as C
    import numpy as np
    from oracle import kqo
    from tests.emu import kqe
    req = np.asarray(case["req"], np.int64); cap = np.asarray(case["cap"], np.int64)
    for l, fn in ((kqo.lib(), "kqo_tas_count_in"), (kqe.lib(), "kqe_tas_count_in")):
        f = getattr(l, fn)
        f.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        out = C.c_int32()
        assert f(len(req), req.ctypes.data, cap.ctypes.data, C.byref(out)) == 0
        assert out.value == case["want"], fn


def _sorted_domains(case, with_leader):
    import ctypes as C
    import numpy as np
    from oracle import kqo
    doms = case["domains"]
    by_level = sorted(range(len(doms)), key=lambda i: doms[i]["levelValue"])  # domain index = rank of its levelValues (final tie-break)
    idx_of = {orig: rank for rank, orig in enumerate(by_level)}
    state = np.zeros((len(doms), 5), np.int32)
    for orig, d in enumerate(doms):
        state[idx_of[orig]] = [d.get("podCount", 0), d.get("sliceCount", 0), d.get("podCountWithLeader", 0), d.get("sliceCountWithLeader", 0), d.get("leaderCount", 0)]
    order_in = np.asarray([idx_of[i] for i in range(len(doms))], np.int32)
    out = np.zeros(len(doms), np.int32)
    l = kqo.lib()
    if case["affinityGate"]:   # features.TASRespectNodeAffinityPreferred on: the oracle's restatement of the gate (the library refuses it)
        aff = np.zeros(len(doms), np.int64)
        for orig, d in enumerate(doms):
            aff[idx_of[orig]] = d.get("affinityScore", 0)
        l.kqo_tas_sorted_domains_affinity.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        assert l.kqo_tas_sorted_domains_affinity(len(doms), state.ctypes.data, aff.ctypes.data, order_in.ctypes.data, int(case["unconstrained"]), 1, int(with_leader), out.ctypes.data) == 0
        return [doms[by_level[r]]["id"] for r in out]
    l.kqo_tas_sorted_domains.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    # the reference test runs with the default (Mixed) profile: unconstrained => LeastFreeCapacity (tas_flavor_snapshot.go:1468)
    assert l.kqo_tas_sorted_domains(len(doms), state.ctypes.data, order_in.ctypes.data, int(case["unconstrained"]), 1, int(with_leader), out.ctypes.data) == 0
    return [doms[by_level[r]]["id"] for r in out]


@pytest.mark.parametrize("case", T["sortedDomains"], ids=lambda c: c["name"][:70])
def test_tas_sorted_domains(oracle, case):
    """tas_flavor_snapshot_test.go:884 TestSortedDomains (the TASRespectNodeAffinityPreferred rows on the oracle's restatement of the gate)."""
    assert _sorted_domains(case, False) == case["want"]


@pytest.mark.parametrize("case", T["sortedDomainsWithLeader"], ids=lambda c: c["name"][:70])
def test_tas_sorted_domains_with_leader(oracle, case):
    """tas_flavor_snapshot_test.go:601 TestSortedDomainsWithLeader."""
    assert _sorted_domains(case, True) == case["want"]


DRS_KIND = {"absent": -1, "zero": 0, "borrowing": 1, "negative": 2}


@pytest.mark.parametrize("case", M["entryComparerLess"], ids=lambda c: c["name"][:70])
def test_entry_comparer_less(oracle, case):
    """scheduler_test.go:8411 TestEntryComparerLess — entryComparer.less with injected DRS values (hand transcription, each case cites its
    line). The engine orders through the same comparison inside k_process_fair (a 4 x u64 key); its inputs there are real DRS values of a
    snapshot, so this table pins the restatement, the whole-cycle fair-sharing tables (schedule_fair.yaml, schedule_recompute.yaml) the engine."""
    import ctypes as C
    import numpy as np
    rg = [ResourceGroup([FlavorQuotas("default").Resource("cpu", "100")])]
    cqs = [ClusterQueue("cq-a", cohort="test-cohort", resource_groups=rg), ClusterQueue("cq-b", cohort="test-cohort", resource_groups=rg)]
    snap = Snapshot(cqs, [Cohort("test-cohort")], [])
    oracle.derive(snap)
    wls = [Workload(n, f"cq-{n}", priority=0, creation_ts=case[n]["queueTs"] * 1_000_000_000, pod_sets=[PodSet("main", count=1).Request("cpu", "1")]) for n in ("a", "b")]
    heads = Heads(snap, wls)
    for i, n in enumerate(("a", "b")):
        if case[n].get("preemptor"):
            heads.arrays["flags"][i] |= F.HEAD_IS_PREEMPTOR
    heads._struct = None
    cfg = make_config(fair_sharing=True, gates=gates_with(case.get("gates") or {}))
    kind = np.array([DRS_KIND[case[n]["drs"]] for n in ("a", "b")], np.int32)
    req = np.array([int(case[n].get("requested", 0)) for n in ("a", "b")], np.int64)
    oracle.lib().kqo_entry_less.restype = C.c_int
    got = oracle.lib().kqo_entry_less(C.byref(cfg), C.byref(snap.struct()), C.byref(heads.struct()), F.ptr(kind), F.ptr(req))
    assert bool(got) == case["want"]


@pytest.mark.parametrize("case", M["fitsDedupsOverlappingVictims"], ids=lambda c: c["name"][:70])
def test_fits_dedups_overlapping_victims(oracle, case):
    """scheduler_test.go:9379 TestFitsDedupsOverlappingVictims — scheduler.fits removes preemptedWorkloads ∪ targets, every workload once."""
    import ctypes as C
    import numpy as np
    from kueue_amd.api import resource_value
    rg = [ResourceGroup([FlavorQuotas("default").Resource("cpu", case["nominal"])])]
    snap_adm = [Workload(n, "cq", priority=0, creation_ts=0, pod_sets=[PodSet("main", count=1, flavors={"cpu": "default"}).Request("cpu", q)], reserve_ts=1, uid=n)
                for n, q in case["admitted"].items()]
    snap = Snapshot([ClusterQueue("cq", resource_groups=rg)], [], snap_adm)
    oracle.derive(snap)
    heads = Heads(snap, [Workload("incoming", "cq", priority=0, creation_ts=1, pod_sets=[PodSet("main", count=1).Request("cpu", case["incoming"])])])
    row = {w.name: i for i, w in enumerate(snap.admitted)}
    fr = np.array([snap.flavor_index["default"] * snap.n_resource + snap.resource_index["cpu"]], np.int32)
    qty = np.array([resource_value("cpu", case["incoming"])], np.int64)
    pre = np.array([row[n] for n in case["preempted"]], np.int32)
    tgt = np.array([row[n] for n in case["targets"]], np.int32)
    cfg = make_config()
    oracle.lib().kqo_fits_check.restype = C.c_int
    got = oracle.lib().kqo_fits_check(C.byref(cfg), C.byref(snap.struct()), C.byref(heads.struct()), C.c_int32(0), 1, F.ptr(fr), F.ptr(qty),
                                      len(pre), F.ptr(pre), len(tgt), F.ptr(tgt))
    assert got == {"Ok": 0, "NoQuota": 1}[case["want"]]
    # the engine's scheduler.fits (entry_fits over the np plane, kq_device.hpp) sees the same situation in a whole cycle: a head whose
    # nominated target was preempted by an earlier entry of the cycle — tests/golden/schedule_recompute.yaml "legacy overlap skip"


@pytest.mark.parametrize("case", M["totalRequestsFor"], ids=lambda c: c["name"][:70])
def test_total_requests_for(oracle, case):
    """flavorassigner_test.go:4645 TestAssignment_TotalRequestsFor — the usage GetTargets and processEntry charge for an assignment: scaled
    to the assigned pod counts (partial admission), only the delta over a replaced workload slice (hand transcription)."""
    import ctypes as C
    import numpy as np
    rg = [ResourceGroup([FlavorQuotas("default").Resource("cpu", "100").Resource("memory", "100Gi")])]
    admitted = []
    if case.get("slice"):
        old = [PodSet(p["name"], count=p["count"], flavors={r: "default" for r in p["requests"]}) for p in case["slice"]]
        for ps, p in zip(old, case["slice"]):
            for r, q in p["requests"].items():
                ps.Request(r, q)
        admitted.append(Workload("old-slice", "cq", pod_sets=old, reserve_ts=1, uid="old-slice"))
    extra = sorted({r for p in case["podsets"] for r in p["requests"]})
    snap = Snapshot([ClusterQueue("cq", resource_groups=rg)], [], admitted, extra_resources=extra)
    oracle.derive(snap)
    pss = [PodSet(p["name"], count=p["count"]) for p in case["podsets"]]
    for ps, p in zip(pss, case["podsets"]):
        for r, q in p["requests"].items():
            ps.Request(r, q)
    heads = Heads(snap, [Workload("test", "cq", pod_sets=pss, replaces="old-slice" if case.get("slice") else None)])
    nR = snap.n_resource
    flavor = np.full(len(pss) * nR, -1, np.int32)
    for pi in range(len(pss)):
        for r in ("cpu", "memory"):
            flavor[pi * nR + snap.resource_index[r]] = snap.flavor_index["default"]
    counts = np.array(case["assigned"], np.int32)
    usage = np.zeros(snap.n_fr, np.int64)
    cfg = make_config()
    oracle.lib().kqo_total_requests_for.restype = C.c_int
    rc = oracle.lib().kqo_total_requests_for(C.byref(cfg), C.byref(snap.struct()), C.byref(heads.struct()), 0, F.ptr(counts), F.ptr(flavor), F.ptr(usage))
    assert rc == 0
    got = {snap.fr_name(fr): int(usage[fr]) for fr in range(snap.n_fr) if usage[fr] != 0}
    assert got == {tuple(k.split("/", 1)): v for k, v in case["want"].items()}

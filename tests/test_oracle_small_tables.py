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
    l.kqo_tas_sorted_domains.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    # the reference test runs with the default (Mixed) profile: unconstrained => LeastFreeCapacity (tas_flavor_snapshot.go:1468)
    assert l.kqo_tas_sorted_domains(len(doms), state.ctypes.data, order_in.ctypes.data, int(case["unconstrained"]), 1, int(with_leader), out.ctypes.data) == 0
    return [doms[by_level[r]]["id"] for r in out]


@pytest.mark.parametrize("case", [c for c in T["sortedDomains"] if not c["affinityGate"]], ids=lambda c: c["name"][:70])
def test_tas_sorted_domains(oracle, case):
    """tas_flavor_snapshot_test.go:884 TestSortedDomains (the TASRespectNodeAffinityPreferred cases are outside the boundary)."""
    assert _sorted_domains(case, False) == case["want"]


@pytest.mark.parametrize("case", [c for c in T["sortedDomainsWithLeader"] if not c["affinityGate"]], ids=lambda c: c["name"][:70])
def test_tas_sorted_domains_with_leader(oracle, case):
    """tas_flavor_snapshot_test.go:601 TestSortedDomainsWithLeader."""
    assert _sorted_domains(case, True) == case["want"]

"""Topology-Aware Scheduling INSIDE the scheduling cycle: the oracle (oracle/kq_oracle.cpp kqo_cycle_run_tas) against the whole-cycle
tables of pkg/scheduler/scheduler_tas_test.go (TestScheduleForTAS :58, TestScheduleForTASPreemption :4121, TestScheduleForTASCohorts
:5950) transcribed by tests/golden/extract_schedule_tas.py: exactly one Scheduler.schedule() per case. Checked per head: admitted or
not, flavors, pod counts and the TopologyAssignment (domain values + counts) of the admission, requeue class, preempted set.
The engine does not run this path yet: this pins the checker ahead of the device code."""
import pytest

from kueue_amd import _ffi as F
from kueue_amd.tas_cycle import load_tas_case
from tests.conftest import load_golden

IMMEDIATE = {F.RQ_FAILED_AFTER_NOMINATION, F.RQ_PENDING_PREEMPTION}
CASES = load_golden("schedule_tas.yaml")["cases"]


def check_case(oracle, case):
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    d, out = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))
    assert not d.tas_stats["unsupported"], "the case left the restated path"
    strict = {c["name"]: c.get("strategy") == "StrictFIFO" for c in case["clusterQueues"]}
    preempted = set()
    for i, w in enumerate(heads.workloads):
        exp = case["expect"][w.name]
        act = int(d.a["action"][i])
        if w.has_unhealthy_nodes or w.has_quota_reservation:   # (a second pass: after a node failure, or for a delayed topology request — ProvisioningRequest)
            # the second pass after a node failure: wantNewAssignments is what the cache holds afterwards — the replaced assignment, or
            # the admission as it was when no replacement was found (then wantEvents tells which way it went: SecondPassFailed = the entry
            # stays pending with its reservation, EvictedDueToNodeFailures = TASFailedNodeReplacementFailFast evicted it)
            ev = exp.get("events") or []
            if "EvictedDueToNodeFailures" in ev:
                assert act == F.ACT_EVICT and int(d.a["status"][i]) == F.ST_EVICTED, (w.name, act)
            elif "SecondPassFailed" in ev:
                assert act == F.ACT_NONE, (w.name, act)
            else:
                assert act == F.ACT_ADMIT, (w.name, "expected the replacement to be admitted", {k: v[i] for k, v in d.a.items() if len(v) == heads.n})
            got = d.flavors_of(i)
            for pi, ps in enumerate(exp["podsets"]):
                assert {r: v[0] for r, v in got[pi].items()} == ps["flavors"], (w.name, got, ps)
                ta = out.topology_assignment(i, pi)
                if "topologyAssignment" in ps:
                    want = sorted((tuple(v), c) for v, c in ps["topologyAssignment"]["domains"])
                    assert ta is not None and sorted((tuple(v), c) for v, c in ta[1]) == want, (w.name, pi, ta, want)
            continue
        if exp["admitted"]:
            assert act == F.ACT_ADMIT, (w.name, "expected admission", {k: v[i] for k, v in d.a.items() if len(v) == heads.n})
            got = d.flavors_of(i)
            for pi, ps in enumerate(exp["podsets"]):
                assert {r: v[0] for r, v in got[pi].items()} == ps["flavors"], (w.name, got, ps)
                assert int(d.a["ps_count"][heads.arrays["ps_off"][i] + pi]) == ps["count"], (w.name, pi)
                ta = out.topology_assignment(i, pi)
                if "topologyAssignment" in ps:
                    assert ta is not None, (w.name, pi, "no TopologyAssignment")
                    want = sorted((tuple(v), c) for v, c in ps["topologyAssignment"]["domains"])
                    assert sorted((tuple(v), c) for v, c in ta[1]) == want, (w.name, pi, ta, want)
                else:
                    assert ta is None, (w.name, pi, ta)
        else:
            assert act != F.ACT_ADMIT, (w.name, "unexpected admission", out.topology_assignment(i, 0))
            rq = int(d.a["requeue_reason"][i])
            # requeueIfNotPresent cluster_queue.go:568-575: immediate, or LastAssignment.PendingFlavors() (workload.go:211)
            pending_flavors = any(v[2] != -1 for ps in d.flavors_of(i) for v in ps.values())
            active = strict[w.cluster_queue] or rq in IMMEDIATE or pending_flavors
            if exp.get("left") == "inadmissible":
                assert not active, (w.name, rq)
            elif exp.get("left") == "active":
                assert active, (w.name, rq)
        if act == F.ACT_PREEMPT:
            preempted |= {t.split(":")[0] for t in d.target_names(i)}
    if "wantPreempted" in case:
        assert sorted(preempted) == case["wantPreempted"], (sorted(preempted), case["wantPreempted"])


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:80])
def test_schedule_tas(oracle, case):
    check_case(oracle, case)


HOST = "kubernetes.io/hostname"


def test_multiple_tas_flavors_in_one_podset_is_nofit(oracle):
    """TestAssignFlavors "multiple TAS flavors assigned to different resources in the same PodSet leads to NoFit"
    (flavorassigner_test.go:3443): cpu lands on tas-a, memory on tas-b, onlyTASFlavor fails -> psError -> NoFit, flavors kept as Fit."""
    node = dict(name="x1", labels={HOST: "x1"}, allocatable={"cpu": "10", "memory": "10Gi", "pods": "10"})
    case = dict(
        nodes=[node], topologies={"tas-topo-a": [HOST], "tas-topo-b": [HOST]},
        resourceFlavors=[dict(name="tas-a", topologyName="tas-topo-a"), dict(name="tas-b", topologyName="tas-topo-b")],
        clusterQueues=[dict(name="test-clusterqueue", resourceGroups=[[dict(flavor="tas-a", resources={"cpu": ["10", "", ""]})],
                                                                      [dict(flavor="tas-b", resources={"memory": ["10Mi", "", ""]})]])],
        pending=[dict(name="ns/wl", cq="test-clusterqueue", podsets=[dict(name="main", count=1, requests={"cpu": "1", "memory": "1Mi"},
                                                                           topologyRequest={"required": HOST})])])
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    d, out = oracle.cycle_run_tas(cfg, snap, heads, ct)
    assert F.MODE_NAMES[int(d.a["nominated_mode"][0])] == "NoFit" and int(d.a["action"][0]) == F.ACT_NONE
    assert {r: (v[0], v[1], v[2]) for r, v in d.flavors_of(0)[0].items()} == {"cpu": ("tas-a", "Fit", -1), "memory": ("tas-b", "Fit", -1)}
    assert out.topology_assignment(0, 0) is None


def test_fit_on_quota_but_fragmented_topology_turns_into_preempt_then_nofit(oracle):
    """flavorassigner.go:866-903 step by step on one node of 1 cpu: a 1-cpu pod fits; with the node full of an admitted pod the Fit
    becomes Preempt (TAS failure reason), the simulate-empty placement succeeds, and without a preemption policy the head stays
    pending with no targets; a 2-cpu pod does not fit even on the empty topology -> NoFit."""
    node = dict(name="x1", labels={HOST: "x1"}, allocatable={"cpu": "1", "pods": "10"})
    base = dict(nodes=[node], topologies={"t": [HOST]}, resourceFlavors=[dict(name="tas", topologyName="t")],
                clusterQueues=[dict(name="cq", resourceGroups=[[dict(flavor="tas", resources={"cpu": ["50", "", ""]})]])])
    pend = lambda cpu: [dict(name="ns/new", cq="cq", podsets=[dict(name="one", count=1, requests={"cpu": cpu}, topologyRequest={"required": HOST})])]
    adm = [dict(name="ns/old", cq="cq", podsets=[dict(count=1, totalRequests={"cpu": "1"}, flavors={"cpu": "tas"}, podRequests={"cpu": "1"},
                                                       topologyAssignment={"domains": [[["x1"], 1]]})])]
    for admitted, cpu, want_mode, want_ta in (([], "1", "Fit", [(["x1"], 1)]), (adm, "1", "Preempt", [(["x1"], 1)]), ([], "2", "NoFit", None)):
        cfg, snap, heads, ct = load_tas_case(dict(base, admitted=admitted, pending=pend(cpu)))
        oracle.derive(snap)
        d, out = oracle.cycle_run_tas(cfg, snap, heads, ct)
        assert F.MODE_NAMES[int(d.a["nominated_mode"][0])] == want_mode, (cpu, d.a["nominated_mode"])
        ta = out.topology_assignment(0, 0)
        assert (ta[1] if ta else None) == want_ta, (cpu, ta)
        assert (int(d.a["action"][0]) == F.ACT_ADMIT) == (want_mode == "Fit")

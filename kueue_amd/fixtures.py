"""YAML/dict fixtures -> api objects.

The golden vectors under tests/golden/*.yaml are transcriptions of the reference's table tests
(inputs + expected outputs are literal Go struct tables there).  This loader turns one case into
(Snapshot, Heads, config) exactly as the reference harness does with its fluent builders
(pkg/util/testing/v1beta2/wrappers.go).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

from . import _ffi as F
from .api import (ClusterQueue, Cohort, FlavorQuotas, Heads, LastAssignment, PodSet, ResourceGroup, Snapshot,
                  Workload, amount_from_quantity, gates_with, make_config, resource_value, sat)


def _quota(fq: FlavorQuotas, res: str, spec) -> None:
    if isinstance(spec, (list, tuple)):
        vals = list(spec) + ["", "", ""]
        fq.Resource(res, vals[0], vals[1], vals[2])
    else:
        fq.Resource(res, spec)


def _rgs(spec) -> List[ResourceGroup]:
    out = []
    for rg in spec or []:
        flavors = []
        for f in rg:
            fq = FlavorQuotas(f["flavor"])
            for res, q in f["resources"].items():
                _quota(fq, res, q)
            flavors.append(fq)
        out.append(ResourceGroup(flavors))
    return out


def _split_fr(key: str) -> Tuple[str, str]:
    flavor, res = key.split("/", 1)
    return flavor, res


def _cq(d: dict) -> ClusterQueue:
    p = d.get("preemption", {}) or {}
    fu = d.get("fungibility", {}) or {}
    cq = ClusterQueue(
        name=d["name"], cohort=d.get("cohort"), resource_groups=_rgs(d.get("resourceGroups")),
        within_cluster_queue=p.get("withinClusterQueue", "Never"),
        reclaim_within_cohort=p.get("reclaimWithinCohort", "Never"),
        borrow_within_cohort=p.get("borrowWithinCohort", "Never"),
        max_priority_threshold=p.get("maxPriorityThreshold"),
        when_can_borrow=fu.get("whenCanBorrow", "MayStopSearch"),
        when_can_preempt=fu.get("whenCanPreempt", "TryNextFlavor"),
        preference=fu.get("preference"),
        queueing_strategy=d.get("strategy", "BestEffortFIFO"),
        fair_weight=float(d.get("fairWeight", 1.0)),
        generation=int(d.get("generation", 0)),
    )
    for k, q in (d.get("usage") or {}).items():
        f, r = _split_fr(k)
        cq.extra_usage[(f, r)] = amount_from_quantity(r, q)
    for k, q in (d.get("usageRaw") or {}).items():  # already canonical int64 (milli-CPU / bytes)
        f, r = _split_fr(k)
        cq.extra_usage[(f, r)] = int(q)
    return cq


def _cohort(d: dict) -> Cohort:
    return Cohort(name=d["name"], parent=d.get("parent"), resource_groups=_rgs(d.get("resourceGroups")),
                  fair_weight=float(d.get("fairWeight", 1.0)))


def _podsets(d: dict) -> List[PodSet]:
    out = []
    if "usage" in d:  # shorthand: {"flavor/res": qty} -> one podset
        ps = PodSet(name="main", count=int(d.get("count", 1)))
        for k, q in d["usage"].items():
            f, r = _split_fr(k)
            ps.requests[r] = resource_value(r, q)
            ps.flavors[r] = f
        return [ps]
    for i, p in enumerate(d.get("podsets", [])):
        ps = PodSet(name=p.get("name", "main" if i == 0 else f"ps{i}"), count=int(p.get("count", 1)), min_count=p.get("minCount"))
        for r, q in (p.get("requests") or {}).items():
            ps.Request(r, q)
        for r, q in (p.get("totalRequests") or {}).items():
            ps.requests[r] = resource_value(r, q)
        ps.flavors = dict(p.get("flavors") or {})
        ps.excluded_flavors = list(p.get("excludedFlavors") or [])
        ps.group = p.get("group")   # PodSet.TopologyRequest.PodSetGroupName
        out.append(ps)
    return out


def _workload(d: dict) -> Workload:
    la = None
    if d.get("lastAssignment") is not None:
        l = d["lastAssignment"]
        la = LastAssignment(last_tried_flavor_idx=[dict(x) for x in l.get("lastTriedFlavorIdx", [])],
                            cluster_queue_generation=int(l.get("generation", 0)),
                            scheduling_cycle=int(l.get("cycle", 0)), scheduling_hash=int(l.get("hash", 0)))
    return Workload(
        name=d["name"], cluster_queue=d.get("cq", ""), priority=int(d.get("priority", 0)),
        creation_ts=int(d.get("created", 0)), pod_sets=_podsets(d), uid=d.get("uid"),
        reserve_ts=d.get("reservedAt"), evicted=bool(d.get("evicted", False)),
        has_quota_reservation=bool(d.get("hasQuotaReservation", False)), is_preemptor=bool(d.get("isPreemptor", False)),
        has_unhealthy_nodes=bool(d.get("unhealthyNodes")),
        unhealthy_assignment=bool(d.get("unhealthyNodes")) and bool(d.get("isAdmitted", False)) and any(
            dm[0][-1] in d["unhealthyNodes"] for ps in d.get("admission") or [] for dm in (ps.get("topologyAssignment") or {}).get("domains", [])),
        last_assignment=la, scheduling_hash=int(d.get("hash", 0)), replaces=d.get("replaces"),
    )


def load_case(case: dict, cycle: int = 1):
    """-> (cfg, Snapshot [not yet derived], Heads)"""
    cqs = [_cq(c) for c in case.get("clusterQueues", [])]
    cohorts = [_cohort(c) for c in case.get("cohorts", [])]
    admitted = [_workload(w) for w in case.get("admitted", [])]
    pending = [_workload(w) for w in case.get("pending", [])]
    extra_res = set()
    for w in pending:
        for ps in w.pod_sets:
            extra_res.update(ps.requests)
    snap = Snapshot(cqs, cohorts, admitted, now_ns=int(case.get("now", 0)),
                    extra_flavors=case.get("flavors", []), extra_resources=sorted(extra_res))
    # canonical heads order: CQ name ascending, stable within a CQ
    pending.sort(key=lambda w: w.cluster_queue)
    heads = Heads(snap, pending, cycle=int(case.get("cycle", cycle)))
    cfg = make_config(fair_sharing=bool(case.get("fairSharing", False)), gates=gates_with(case.get("gates") or {}),
                      fs_strategies=[{"LessThanOrEqualToFinalShare": 0, "LessThanInitialShare": 1}[s] for s in case.get("fsStrategies", [])])
    return cfg, snap, heads

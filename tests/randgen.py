"""Seeded random populations for differential tests (oracle vs engine), built from api objects."""
import random

from kueue_amd.api import (ClusterQueue, Cohort, FlavorQuotas, Heads, LastAssignment, PodSet, ResourceGroup,
                           ResourceQuota, Snapshot, Workload, make_config, gates_with)

RES = ["cpu", "memory", "example.com/gpu", "pods"]
POL_WCQ = ["Never", "LowerPriority", "LowerOrNewerEqualPriority"]
POL_RWC = ["Never", "LowerPriority", "LowerOrNewerEqualPriority", "Any"]


def random_case(seed, fair=False, preemption=True, max_cq=6, partial=False, fair_dups=False, tight=False, slices=False, wide_rows=False):
    global _TIGHT
    _TIGHT = tight
    rnd = random.Random(seed)
    n_flavors = rnd.randint(1, 4)
    flavors = [f"f{i}" for i in range(n_flavors)]
    shape = rnd.choice(["none", "flat", "deep", "forest"])
    cohorts = []
    if shape == "flat":
        cohorts = [Cohort("root")]
    elif shape == "deep":
        cohorts = [Cohort("root"), Cohort("mid-a", "root"), Cohort("mid-b", "root"), Cohort("leaf-a1", "mid-a"), Cohort("leaf-b1", "mid-b")]
    elif shape == "forest":
        cohorts = [Cohort("r1"), Cohort("r2"), Cohort("r1-a", "r1")]
    for c in cohorts:
        c.fair_weight = rnd.choice([0.0, 0.5, 1.0, 2.0]) if fair and rnd.random() < 0.5 else 1.0
        if rnd.random() < 0.4:
            fq = [FlavorQuotas(f, {r: ResourceQuota(rnd.randint(0, 6) * 1000 if r == "cpu" else rnd.randint(0, 6),
                                                    rnd.choice([None, rnd.randint(0, 5)]) if c.parent else None,
                                                    rnd.choice([None, rnd.randint(0, 5)]) if c.parent else None)
                               for r in RES[:2]}) for f in flavors[:2]]
            c.resource_groups = [ResourceGroup(fq)]
    n_cq = rnd.randint(1, max_cq)
    cqs = []
    for i in range(n_cq):
        two_rg = rnd.random() < 0.3 and n_flavors >= 2
        res_a = RES[:rnd.randint(1, 3)]
        if rnd.random() < 0.3:
            res_a = res_a + ["pods"]
        rgs = []
        if two_rg:
            split = rnd.randint(1, n_flavors - 1)
            rgs.append(ResourceGroup([_fq(rnd, f, res_a[:1]) for f in flavors[:split]]))
            if len(res_a) > 1:
                rgs.append(ResourceGroup([_fq(rnd, f, res_a[1:]) for f in flavors[split:]]))
        else:
            rgs.append(ResourceGroup([_fq(rnd, f, res_a) for f in rnd.sample(flavors, rnd.randint(1, n_flavors))]))
        cq = ClusterQueue(f"cq{i}", cohort=rnd.choice([c.name for c in cohorts]) if cohorts and rnd.random() < 0.9 else None, resource_groups=rgs)
        if preemption:
            cq.within_cluster_queue = rnd.choice(POL_WCQ)
            cq.reclaim_within_cohort = rnd.choice(POL_RWC)
            if rnd.random() < 0.4:
                cq.borrow_within_cohort = "LowerPriority"
                cq.max_priority_threshold = rnd.choice([None, rnd.randint(-1, 3)])
        cq.when_can_borrow = rnd.choice(["MayStopSearch", "TryNextFlavor"])
        cq.when_can_preempt = rnd.choice(["MayStopSearch", "TryNextFlavor"])
        if cq.when_can_borrow == cq.when_can_preempt == "TryNextFlavor":
            cq.preference = rnd.choice([None, "BorrowingOverPreemption", "PreemptionOverBorrowing"])
        cq.fair_weight = rnd.choice([0.0, 0.5, 1.0, 2.0]) if fair else 1.0
        cq.generation = rnd.randint(0, 3)
        cqs.append(cq)
    # admitted workloads
    admitted = []
    t = 0
    for cq in cqs:
        for j in range(rnd.randint(0, 5)):
            ps = PodSet("main", count=rnd.randint(1, 3))
            rg = rnd.choice(cq.resource_groups)
            fl = rnd.choice(rg.flavors)
            for r in fl.resources:
                ps.requests[r] = (rnd.randint(0, 3) * 1000 if r == "cpu" else rnd.randint(0, 3))
                ps.flavors[r] = fl.name
            adm_pods = [ps]
            if wide_rows and rnd.random() < 0.6:
                # a second podset on another flavor (of any resource group): the row holds up to 8 distinct flavor-resources, the records
                # of the scan-formulated searches hold 4 (AdmRecX, kq_prep.hpp)
                ps2 = PodSet("second", count=rnd.randint(1, 3))
                rg2 = rnd.choice(cq.resource_groups)
                fl2 = rnd.choice(rg2.flavors)
                for r in fl2.resources:
                    ps2.requests[r] = (rnd.randint(0, 3) * 1000 if r == "cpu" else rnd.randint(0, 3))
                    ps2.flavors[r] = fl2.name
                adm_pods.append(ps2)
            t += 1
            admitted.append(Workload(f"{cq.name}-adm{j}", cq.name, priority=rnd.randint(-1, 3), creation_ts=rnd.randint(0, 50),
                                     pod_sets=adm_pods, reserve_ts=rnd.choice([None, rnd.randint(0, 100)]), evicted=rnd.random() < 0.1,
                                     uid=f"uid-{rnd.randint(0, 10**6)}-{t}"))
    # pending heads: one per CQ (most), in CQ-name order
    pending = []
    for cq in cqs:
        if rnd.random() < 0.15:
            continue
        pods = []
        for p in range(rnd.choice([1, 1, 1, 2])):
            cnt = rnd.randint(1, 4)
            ps = PodSet(f"ps{p}", count=cnt, min_count=(rnd.randint(1, cnt) if partial and rnd.random() < 0.6 else None))
            covered = [r for rg in cq.resource_groups for r in rg.covered_resources if r != "pods"]
            for r in rnd.sample(covered, rnd.randint(1, len(covered))) if covered else []:
                ps.requests[r] = cnt * (rnd.randint(0, 3) * 500 if r == "cpu" else rnd.randint(0, 2))
            if rnd.random() < 0.05:
                ps.requests["uncovered.io/x"] = rnd.choice([0, 1])
            if rnd.random() < 0.15 and n_flavors > 1:
                ps.excluded_flavors = [rnd.choice(flavors)]
            pods.append(ps)
        w = Workload(f"{cq.name}-pend", cq.name, priority=rnd.randint(-1, 4), creation_ts=rnd.randint(0, 60), pod_sets=pods,
                     has_quota_reservation=rnd.random() < 0.03, scheduling_hash=rnd.choice([0, 7]))
        if rnd.random() < 0.3:
            w.last_assignment = LastAssignment(
                last_tried_flavor_idx=[{r: rnd.randint(-1, max(0, n_flavors - 2)) for r in ps.requests if r in RES} for ps in pods],
                cluster_queue_generation=rnd.randint(0, 3), scheduling_cycle=rnd.randint(0, 5), scheduling_hash=rnd.choice([0, 7, 9]))
        if slices and rnd.random() < 0.6:
            # ElasticJobsViaWorkloadSlices: the head is the scaled-up slice of an admitted workload of its ClusterQueue (same podset name in
            # most cases, so that the old requests / flavors line up; sometimes a different one: nothing of the old slice matches)
            olds = [a for a in admitted if a.cluster_queue == cq.name]
            if olds:
                old = rnd.choice(olds)
                w.replaces = old.name
                if rnd.random() < 0.85:
                    w.pod_sets[0].name = old.pod_sets[0].name
                    if rnd.random() < 0.7:   # a real scale-up: the same resources, at least the old amounts
                        np_ = w.pod_sets[0]
                        np_.count = old.pod_sets[0].count + rnd.randint(0, 2)
                        np_.requests = {r: q + rnd.choice([0, 0, 500 if r == "cpu" else 1]) * 1 for r, q in old.pod_sets[0].requests.items()}
                        np_.min_count = None
        pending.append(w)
        if (not fair or fair_dups) and rnd.random() < 0.12:
            # a second head on the same ClusterQueue (second-pass workloads come on top of one head per CQ,
            # pkg/cache/queue/manager.go:923)
            import copy
            w2 = copy.deepcopy(w)
            w2.name = f"{cq.name}-pend2"; w2.priority = rnd.randint(-1, 4); w2.creation_ts = rnd.randint(0, 60); w2.last_assignment = None
            pending.append(w2)
    snap = Snapshot(cqs, cohorts, admitted, now_ns=1000, extra_resources=["uncovered.io/x"])
    heads = Heads(snap, pending, cycle=rnd.randint(1, 6))
    gates = {}
    if rnd.random() < 0.1:
        gates["FlavorFungibility"] = False
    if rnd.random() < 0.1:
        gates["PrioritySortingWithinCohort"] = False
    if rnd.random() < 0.1:
        gates["RecomputeAssignmentUponPreemptionTargetsOverlap"] = False
    if rnd.random() < 0.1:
        gates["FlavorFungibilityPreserveScanProgress"] = False
    fs = ()
    if fair and fair_dups:  # second generation of fair cases: gate and strategy variations on top
        if rnd.random() < 0.2:
            gates["FairSharingPreemptWithinNominal"] = False
        if rnd.random() < 0.2:
            gates["FairSharingPrioritizeNonBorrowing"] = False
        if rnd.random() < 0.2:
            gates["PrioritizePreemptorWorkloads"] = False
        fs = rnd.choice([(), (), (0,), (1,), (1, 0), (0, 1)])
    cfg = make_config(fair_sharing=fair, gates=gates_with(gates), fs_strategies=fs)
    return cfg, snap, heads


_TIGHT = False


def _fq(rnd, flavor, resources):
    fq = FlavorQuotas(flavor)
    for r in resources:
        unit = 1000 if r == "cpu" else 1
        nominal = rnd.randint(0, 3 if _TIGHT else 8) * unit
        if rnd.random() < 0.03:
            nominal = (1 << 63) - 1
        # tight: over-committed ClusterQueues (usage above nominal + borrowingLimit), where
        # quotaResourcesToReserve goes negative (scheduler.go:806) and removals stop commuting
        bl = rnd.choice([None, rnd.randint(0, 2) * unit, rnd.randint(0, 2) * unit]) if _TIGHT else rnd.choice([None, None, rnd.randint(0, 6) * unit])
        ll = rnd.choice([None, None, rnd.randint(0, 6) * unit])
        fq.resources[r] = ResourceQuota(nominal, bl, ll)
    return fq

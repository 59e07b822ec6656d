"""Seeded random TAS sides on top of tests/randgen.py populations: some flavors become TAS flavors over random node trees, pending
podsets get random topology requests, a few admitted workloads get a TopologyAssignment. Used for property tests of the TAS cycle."""
import random

from kueue_amd.api import Heads
from kueue_amd.tas import Node, TopologyRequest
from kueue_amd.tas_cycle import AdmittedTAS, CycleTAS, PodSetTAS, ResourceFlavor, build_topologies, excluded_flavors_for_tas
from tests.randgen import random_case

BLOCK, RACK, HOST = "cloud.com/topology-block", "cloud.com/topology-rack", "kubernetes.io/hostname"


def add_node_masks(seed, heads, topologies, pod_tas, share=0.45):
    """Node feasibility per (podset, TAS flavor) — what taints / tolerations / nodeSelector / required affinity leave of the leaves
    (kq_cycle_tas.ps_mask): a few masks per topology, shared by the podsets that draw them (tolerations repeat across a queue's workloads),
    now and then a mask that leaves nothing."""
    rm = random.Random(seed * 101 + 3)
    pool = {}
    for name, topo in sorted(topologies.items()):
        n = topo.n_leaves
        ms = []
        for _ in range(3):
            keep = rm.choice([0.5, 0.7, 0.9])
            ms.append([1 if rm.random() < keep else 0 for _ in range(n)])
        ms.append([0] * n)
        pool[name] = ms
    for w in heads.workloads:
        for pi in range(len(w.pod_sets)):
            if rm.random() >= share:
                continue
            pt = pod_tas[(w.name, pi)]
            pt.leaf_ok = {name: (ms[3] if rm.random() < 0.08 else rm.choice(ms[:3])) for name, ms in pool.items() if rm.random() < 0.8}


def random_tas_cycle_case(seed, roomy=False, rich_groups=False, masks=False, **kw):
    cfg, snap, heads = random_case(seed, **kw)
    rnd = random.Random(seed * 7919 + 13)
    levels = rnd.choice([[HOST], [RACK, HOST], [BLOCK, RACK, HOST], [BLOCK, RACK]])
    nodes = []
    for b in range(rnd.randint(1, 2)):
        for r in range(rnd.randint(1, 3)):
            for h in range(rnd.randint(1, 3)):
                big = 10 ** 6 if roomy else 1
                nodes.append(Node(f"b{b}-r{r}-x{h}", {BLOCK: f"b{b}", RACK: f"r{r}", HOST: f"b{b}-r{r}-x{h}", "pool": rnd.choice(["a", "b"])},
                                  {"cpu": rnd.randint(1, 8) * 1000 * big, "memory": rnd.randint(1, 8) * big, "example.com/gpu": rnd.randint(0, 4) * big,
                                   "pods": rnd.randint(2, 12) * big}))
    flavors = []
    for f in snap.flavors:
        if rnd.random() < 0.6:
            flavors.append(ResourceFlavor(f, {"pool": rnd.choice(["a", "b"])} if rnd.random() < 0.4 else {}, "topo"))
        else:
            flavors.append(ResourceFlavor(f))
    fl = {f.name: f for f in flavors}
    topologies = build_topologies(flavors, {"topo": levels}, nodes, None, ["cpu", "memory", "example.com/gpu", "uncovered.io/x"])
    cqs = {c.name: c for c in snap.cluster_queues}
    pod_tas = {}
    pending = heads.workloads
    rg = random.Random(seed * 31 + 7)   # (its own stream: the populations of the earlier seeds stay what they were)
    # leader + workers (PodSetGroupName): the API admits one leader pod per group, findLeaderAndWorkers :668 takes the smaller podset
    grouped = {w.name for w in pending if len(w.pod_sets) == 2 and min(ps.count for ps in w.pod_sets) == 1 and rg.random() < 0.7}
    if rich_groups:
        # LeaderWorkerSet shapes the flavor scan's grouping (flavorassigner.go:782-860) is sensitive to: a leader that requests nothing (it
        # keeps the group's TAS flavors, resolvePodSetFlavors :931) or only some of the workers' resources; more 2-podset workloads in a group
        rr = random.Random(seed * 53 + 29)
        for w in pending:
            if len(w.pod_sets) == 2 and w.name not in grouped and rr.random() < 0.6 and not getattr(w, "replaces", None):
                w.pod_sets[rr.randrange(2)].count = 1
                for ps in w.pod_sets:
                    ps.min_count = None if ps.count == 1 else ps.min_count
                grouped.add(w.name)
            if w.name not in grouped:
                continue
            lead = min(w.pod_sets, key=lambda ps: ps.count)
            k = rr.random()
            if k < 0.35:
                lead.requests = {}
            elif k < 0.6 and len(lead.requests) > 1:
                for r in rr.sample(sorted(lead.requests), rr.randint(1, len(lead.requests) - 1)):
                    del lead.requests[r]
    for w in pending:
        for pi, ps in enumerate(w.pod_sets):
            tr = None
            k = rnd.random()
            if k < 0.25:
                tr = TopologyRequest(required=rnd.choice(levels))
            elif k < 0.45:
                tr = TopologyRequest(preferred=rnd.choice(levels))
            elif k < 0.6:
                tr = TopologyRequest(unconstrained=True)
            elif k < 0.7 and ps.count > 1:
                tr = TopologyRequest(required=levels[0], slice_required_topology=levels[-1], slice_size=rnd.choice([s for s in (1, 2, 3) if ps.count % s == 0]))
            per_pod = {r: (q // ps.count if ps.count else 0) for r, q in ps.requests.items() if r != "pods"}
            if w.name in grouped and pi == 1:   # both podsets of a group carry the same request (the reference validates that)
                tr = pod_tas[(w.name, 0)].topology_request
            pt = PodSetTAS(tr, "grp" if w.name in grouped else None, per_pod)
            pod_tas[(w.name, pi)] = pt
            ex = excluded_flavors_for_tas(cqs[w.cluster_queue], [r for r in ps.requests if r != "pods" or True], pt, topologies, fl)
            ps.excluded_flavors = sorted(set(ps.excluded_flavors) | set(ex))
    heads = Heads(snap, pending, cycle=heads.cycle)
    admitted_tas = {}
    for w in snap.admitted:
        for ps in w.pod_sets:
            tf = [f for f in set(ps.flavors.values()) if f in topologies]
            if len(tf) != 1 or rnd.random() < 0.3 or ps.count <= 0:
                continue
            topo = topologies[tf[0]]
            if topo.n_leaves == 0:
                continue
            leaf = rnd.randrange(topo.n_leaves)
            per_pod = {r: q // ps.count for r, q in ps.requests.items() if r != "pods" and q // ps.count > 0}
            admitted_tas.setdefault(w.name, []).append(AdmittedTAS(tf[0], [(tuple(topo.leaf_values(leaf)), ps.count)], per_pod))
    if masks:
        add_node_masks(seed, heads, topologies, pod_tas)
    ct = CycleTAS(snap, heads, topologies, pod_tas, admitted_tas, recompute=rnd.random() < 0.85)
    return cfg, snap, heads, ct, pod_tas


def random_second_pass_case(seed, **kw):
    """random_tas_cycle_case + heads on their SECOND pass after a node failure (workload.NeedsSecondPass workload.go:974): a few admitted TAS
    workloads come back as heads that hold their admission; one node of it is unhealthy (a leaf of the snapshot, or a node the snapshot
    no longer holds). The cycle mixes them with the first-pass heads (manager.go:923)."""
    import copy

    from kueue_amd.tas_cycle import HeadAdmission
    cfg, snap, heads, ct, pod_tas = random_tas_cycle_case(seed, **kw)
    rnd = random.Random(seed * 104729 + 5)
    topologies = {n: t for n, t in zip(ct.names, ct.topos)}
    admitted_tas, head_adm, second = {}, {}, []
    levels = ct.topos[0].levels if ct.topos else [HOST]
    for w in snap.admitted:
        tas_ps = []
        for pi, ps in enumerate(w.pod_sets):
            tf = [f for f in set(ps.flavors.values()) if f in topologies]
            tas_ps.append(tf[0] if len(tf) == 1 and ps.count > 0 and topologies[tf[0]].n_leaves > 0 else None)
        if not any(tas_ps) or rnd.random() < 0.35:
            continue
        make_head = rnd.random() < 0.6 and len(second) < 3 and all(topologies[t].lowest_is_node for t in tas_ps if t)   # (an unhealthy NODE: hostname leaves)
        gone = f"gone-{w.name}"
        unhealthy = None
        doms = []
        for pi, ps in enumerate(w.pod_sets):
            if tas_ps[pi] is None:
                doms.append(None)
                continue
            topo = topologies[tas_ps[pi]]
            left, d, used = ps.count, [], set()
            while left > 0:
                c = rnd.randint(1, left)
                if make_head and rnd.random() < 0.25 and gone not in used:
                    d.append(((gone,) if topo.lowest_is_node else tuple(["gone"] * (len(levels) - 1) + [gone]), c)); used.add(gone)
                else:
                    free = [l for l in range(topo.n_leaves) if l not in used]
                    if len(free) <= 1:
                        c = left   # the last free leaf takes the rest
                    if not free:
                        v, c0 = d[-1]; d[-1] = (v, c0 + left); break
                    leaf = rnd.choice(free)
                    used.add(leaf)
                    d.append((tuple(topo.leaf_values(leaf)), c))
                left -= c
            doms.append(d)
            per_pod = {r: q // ps.count for r, q in ps.requests.items() if r != "pods" and q // ps.count > 0}
            admitted_tas.setdefault(w.name, []).append(AdmittedTAS(tas_ps[pi], [x for x in d if x[0][-1] != gone], per_pod))
        if not make_head:
            continue
        names = [v[-1] for d in doms if d for v, _ in d]
        k = rnd.random()
        unhealthy = [gone] if (gone in names and k < 0.5) else [rnd.choice(names)]
        if rnd.random() < 0.15:
            unhealthy.append(rnd.choice(names))   # (a second unhealthy node: only UnhealthyNodes[0] is replaced, :693)
        hw = copy.deepcopy(w)
        hw.has_quota_reservation = True
        hw.has_unhealthy_nodes = True
        is_admitted = rnd.random() < 0.9
        ha = HeadAdmission([dict(ps.flavors) for ps in w.pod_sets], doms, unhealthy, admitted=is_admitted)
        hw.unhealthy_assignment = ha.names_unhealthy()
        for pi, ps in enumerate(hw.pod_sets):
            tr = None
            k = rnd.random()
            if k < 0.3:
                tr = TopologyRequest(required=rnd.choice(levels))
            elif k < 0.55:
                tr = TopologyRequest(preferred=rnd.choice(levels))
            elif k < 0.7:
                tr = TopologyRequest(unconstrained=True)
            elif k < 0.9 and ps.count > 1:
                tr = TopologyRequest(required=levels[0], slice_required_topology=levels[-1], slice_size=rnd.choice([s for s in (1, 2, 3, 4) if ps.count % s == 0]))
            per_pod = {r: (q // ps.count if ps.count else 0) for r, q in ps.requests.items() if r != "pods"}
            pod_tas[(hw.name, pi)] = PodSetTAS(tr, None, per_pod)
        head_adm[hw.name] = ha
        second.append(hw)
    # the rows that did not get a fresh assignment above keep the one random_tas_cycle_case drew
    if kw.get("masks") and second:
        import types
        add_node_masks(seed + 977, types.SimpleNamespace(workloads=second), topologies, pod_tas, share=0.5)
    workloads = second + list(heads.workloads)
    heads = Heads(snap, workloads, cycle=heads.cycle)
    ct2 = CycleTAS(snap, heads, topologies, pod_tas, admitted_tas, recompute=rnd.random() < 0.85, head_admission=head_adm or None,
                   fail_fast=rnd.random() < 0.7)
    return cfg, snap, heads, ct2, len(second)

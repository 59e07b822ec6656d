"""Seeded random populations whose pending workloads hold PodSetGroupName groups of several podsets (flavorassigner.go:782-860): tests/randgen.py
populations with the heads rebuilt — 2-5 podsets, consecutive runs of them share a group name, some members request nothing (the leader of a
LeaderWorkerSet), members request different subsets of the ClusterQueue's resources, so that the SUM of a group decides its flavor."""
import copy
import random

from kueue_amd.api import Heads, PodSet
from tests.randgen import random_case


def grouped_case(seed, **kw):
    cfg, snap, heads = random_case(seed, **kw)
    rnd = random.Random(seed * 9176 + 11)
    cqs = {c.name: c for c in snap.cluster_queues}
    pending = copy.deepcopy(heads.workloads)
    n_multi = 0
    small = rnd.random() < 0.6   # smaller members: the sum of a group still fits somewhere
    for w in pending:
        if w.replaces:   # (a workload slice holds no group of several podsets: the engine refuses that shape)
            continue
        cq = cqs[w.cluster_queue]
        covered = [r for rg in cq.resource_groups for r in rg.covered_resources if r != "pods"]
        base = w.pod_sets[0]
        pods = []
        for p in range(rnd.randint(2, 5)):
            cnt = rnd.randint(1, 2) if small else rnd.randint(1, 4)
            ps = PodSet(f"ps{p}", count=cnt, min_count=(rnd.randint(1, cnt) if base.min_count is not None and rnd.random() < 0.5 else None))
            k = rnd.random()
            if k < 0.2 or not covered:
                pass   # requests nothing
            else:
                for r in rnd.sample(covered, rnd.randint(1, len(covered))):
                    ps.requests[r] = cnt * ((rnd.randint(0, 2) * 250 if small else rnd.randint(0, 3) * 500) if r == "cpu" else (rnd.randint(0, 1) if small else rnd.randint(0, 2)))
                if rnd.random() < 0.05:
                    ps.requests["uncovered.io/x"] = rnd.choice([0, 1])
            if rnd.random() < 0.12 and len(snap.flavors) > 1:
                ps.excluded_flavors = [rnd.choice(list(snap.flavors))]
            pods.append(ps)
        # consecutive runs share a group
        i, g = 0, 0
        while i < len(pods):
            run = rnd.choice([1, 2, 2, 3, 4])
            if run > 1 and rnd.random() < 0.85:
                for ps in pods[i:i + run]:
                    ps.group = f"g{g}"
                n_multi += len(pods[i:i + run]) > 1
                g += 1
            i += run
        w.pod_sets = pods
        if w.last_assignment is not None:
            # (bookmarks the reference can hold: TriedFlavorIdx is -1 once the last flavor of the resource group was tried, flavorassigner.go:1192)
            nfl = {r: len(rg.flavors) for rg in cq.resource_groups for r in rg.covered_resources}
            w.last_assignment.last_tried_flavor_idx = [{r: rnd.randint(-1, max(-1, nfl.get(r, 1) - 2)) for r in ps.requests if r != "uncovered.io/x"} for ps in pods]
    return cfg, snap, Heads(snap, pending, cycle=heads.cycle), n_multi

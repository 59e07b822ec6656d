"""Multi-GPU sharding of the scheduling cycle: whole root-cohort trees per rank, no data-path collective.

Quota, usage bubbling and preemption candidates never leave a root cohort (pkg/cache/scheduler/resource_node.go:144-165,
pkg/scheduler/preemption/preemption.go:642), and the classical entry order only matters between entries that share a tree
(processEntry mutates the tree of the entry's ClusterQueue only, scheduler.go:486), so a population partitions by root cohort and
each rank runs an ordinary cycle on its sub-snapshot. The union of the per-rank decisions equals the single-snapshot decisions.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

from .api import ClusterQueue, Cohort, Workload


def root_of(name: str, parent: Dict[str, str]) -> str:
    seen = set()
    while name in parent and parent[name]:
        if name in seen:
            raise ValueError("cohort cycle")
        seen.add(name)
        name = parent[name]
    return name


def partition_roots(cqs: Sequence[ClusterQueue], cohorts: Sequence[Cohort], admitted: Sequence[Workload], pending: Sequence[Workload],
                    n_ranks: int) -> List[List[str]]:
    """Greedy balance of root trees (a cohort-less ClusterQueue is its own tree, keyed "cq:<name>") over ranks by
    weight = #pending + #admitted + #ClusterQueues. Deterministic: ties broken by tree key."""
    parent = {c.name: c.parent for c in cohorts}
    tree_of_cq = {q.name: (root_of(q.cohort, parent) if q.cohort else f"cq:{q.name}") for q in cqs}
    weight: Dict[str, int] = {}
    for q in cqs:
        weight[tree_of_cq[q.name]] = weight.get(tree_of_cq[q.name], 0) + 1
    for w in list(admitted) + list(pending):
        t = tree_of_cq[w.cluster_queue]
        weight[t] = weight.get(t, 0) + 1
    ranks: List[List[str]] = [[] for _ in range(n_ranks)]
    load = [0] * n_ranks
    for t in sorted(weight, key=lambda t: (-weight[t], t)):
        r = min(range(n_ranks), key=lambda i: (load[i], i))
        ranks[r].append(t); load[r] += weight[t]
    return ranks


def shard(cqs: Sequence[ClusterQueue], cohorts: Sequence[Cohort], admitted: Sequence[Workload], pending: Sequence[Workload],
          trees: Sequence[str]) -> Tuple[List[ClusterQueue], List[Cohort], List[Workload], List[Workload]]:
    """The sub-population made of the given root trees."""
    parent = {c.name: c.parent for c in cohorts}
    keep = set(trees)
    tree_of_cq = {q.name: (root_of(q.cohort, parent) if q.cohort else f"cq:{q.name}") for q in cqs}
    my_cqs = [q for q in cqs if tree_of_cq[q.name] in keep]
    names = {q.name for q in my_cqs}
    my_cohorts = [c for c in cohorts if root_of(c.name, parent) in keep]
    return (my_cqs, my_cohorts, [w for w in admitted if w.cluster_queue in names], [w for w in pending if w.cluster_queue in names])

"""Synthetic TAS population of BASELINE.json configs[4] (SURVEY §8d cfg 5): a 3-tier topology of 8 blocks x 8 racks x
64 hosts = 4096 leaves (capacity 96 cpu / 256 GiB / 8 gpu / 110 pods, leaf usage 40-80 %), one TAS flavor, single-podset
workloads of 1..64 pods that are required / preferred / unconstrained at rack or block level; request classes after
test/performance/scheduler/configs/tas/generator.yaml:44-151 (500m / 1250m / 2500m per pod)."""
from __future__ import annotations

import numpy as np

from . import tas as T

BLOCK, RACK = "cloud.provider.com/topology-block", "cloud.provider.com/topology-rack"
TAS_SEED = 20260921 + 5


def generate_tas(n_workloads: int = 50_000, seed: int = TAS_SEED, blocks: int = 8, racks: int = 8, hosts: int = 64, kinds=(0, 1, 2)):
    rng = np.random.default_rng(seed)
    levels = [BLOCK, RACK, T.HOSTNAME_LABEL]
    nodes = []
    for b in range(blocks):
        for r in range(racks):
            for h in range(hosts):
                nodes.append(T.Node(f"b{b}-r{r}-h{h:02d}", {BLOCK: f"b{b}", RACK: f"b{b}-r{r}", T.HOSTNAME_LABEL: f"b{b}-r{r}-h{h:02d}"},
                                    {"cpu": 96000, "memory": 256 << 30, "example.com/gpu": 8, "pods": 110}))
    topo = T.Topology(levels, nodes)
    R = len(topo.resources)
    ri = topo.resource_index
    use = topo.arrays["tas_usage"].reshape(topo.n_leaves, R)
    frac = rng.uniform(0.4, 0.8, size=topo.n_leaves)
    use[:, ri["cpu"]] = (frac * 96).astype(np.int64) * 1000
    use[:, ri["memory"]] = (frac * 256).astype(np.int64) << 30
    use[:, ri["example.com/gpu"]] = rng.integers(0, 9, size=topo.n_leaves) * (rng.random(topo.n_leaves) < 0.7)
    use[:, ri["pods"]] = (frac * 60).astype(np.int64)
    cpu_classes = np.array([500, 1250, 2500])
    workloads = []
    kinds = np.asarray(kinds)[rng.integers(0, len(kinds), size=n_workloads)]   # required / preferred / unconstrained
    lvls = rng.integers(0, 2, size=n_workloads)              # block / rack
    counts = rng.integers(1, 65, size=n_workloads)
    cls = rng.integers(0, 3, size=n_workloads)
    gpu = (rng.random(n_workloads) < 0.2).astype(np.int64)
    for i in range(n_workloads):
        cpu = int(cpu_classes[cls[i]])
        reqs = {"cpu": cpu, "memory": (cpu * 4 << 30) // 1000}
        if gpu[i]:
            reqs["example.com/gpu"] = 1
        lv = levels[int(lvls[i])]
        if kinds[i] == 0:
            tr = T.TopologyRequest(required=lv)
        elif kinds[i] == 1:
            tr = T.TopologyRequest(preferred=lv)
        else:
            tr = T.TopologyRequest(unconstrained=True)
        workloads.append([T.TASPodSetRequests("main", int(counts[i]), reqs, tr)])
    return topo, T.Requests(topo, workloads)

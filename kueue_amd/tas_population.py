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


def generate_tas_cycle(n_cq: int = 1000, n_pending: int = 50_000, seed: int = TAS_SEED, cohorts: int = 10, **topo_kw):
    """BASELINE configs[4] as whole scheduling cycles ("TAS domain fit + flavor assign"): the cfg 5 topology as ONE TAS flavor that every
    ClusterQueue lists (its leaves are shared across the root cohorts, snapshot.go:260) next to an ordinary flavor, n_cq ClusterQueues in
    `cohorts` flat cohorts with nominal quota for a handful of workloads each and borrowing, and n_pending workloads spread round-robin
    over the ClusterQueues: cycle c schedules the c-th workload of every ClusterQueue (queues.Heads(): one head per ClusterQueue).
    -> (Snapshot, topologies, batches) with batches[c] = (Heads, CycleTAS) built on demand by `batch(c)`."""
    from .api import ClusterQueue, Cohort, FlavorQuotas, PodSet, ResourceGroup, ResourceQuota, Snapshot, Workload
    from .tas_cycle import CycleTAS, PodSetTAS
    topo, rq = generate_tas(n_workloads=n_pending, seed=seed, **topo_kw)
    rng = np.random.default_rng(seed + 77)
    res = ["cpu", "memory", "example.com/gpu"]
    cqs = []
    for i in range(n_cq):
        # room for ~4 average workloads of nominal quota on the TAS flavor; the rest is borrowed from the cohort
        fq_tas = FlavorQuotas("tas-flavor", {"cpu": ResourceQuota(200_000, None, None), "memory": ResourceQuota(800 << 30, None, None),
                                             "example.com/gpu": ResourceQuota(16, None, None)})
        fq_std = FlavorQuotas("zz-standard", {"cpu": ResourceQuota(50_000, None, None), "memory": ResourceQuota(200 << 30, None, None),
                                              "example.com/gpu": ResourceQuota(0, None, None)})
        cqs.append(ClusterQueue(f"cq-{i:04d}", cohort=f"cohort-{i % cohorts:02d}", resource_groups=[ResourceGroup([fq_tas, fq_std])]))
    snap = Snapshot(cqs, [Cohort(f"cohort-{j:02d}") for j in range(cohorts)], [], extra_resources=["pods"])
    topologies = {"tas-flavor": topo}
    prio = rng.integers(0, 4, size=n_pending)

    def workloads_of(c: int):
        wls, pod_tas = [], {}
        for i in range(n_cq):
            w = c * n_cq + i
            if w >= n_pending:
                break
            ps = rq.workloads[w][0]
            reqs = {r: int(q) * ps.count for r, q in ps.single_pod_requests.items()}
            name = f"ns/wl-{w:05d}"
            wls.append(Workload(name, f"cq-{i:04d}", priority=int(prio[w]), creation_ts=w, pod_sets=[PodSet("main", count=ps.count, requests=reqs)]))
            pod_tas[(name, 0)] = PodSetTAS(ps.topology_request, None, dict(ps.single_pod_requests))
        return wls, pod_tas

    def batch(c: int):
        from .api import Heads
        wls, pod_tas = workloads_of(c)
        heads = Heads(snap, wls, cycle=c + 1)
        return heads, CycleTAS(snap, heads, topologies, pod_tas)

    batch.closed_loop = lambda hold=0, failures=0: TASClosedLoop(cqs, [Cohort(f"cohort-{j:02d}") for j in range(cohorts)], topologies, workloads_of, hold, failures)
    return snap, topologies, batch


class TASClosedLoop:
    """BASELINE configs[4] as a CLOSED loop, driven the way the reference drives it (scheduler.go:308-386 with manager.go:903): every
    cycle starts from a fresh cache.Snapshot() that holds what the cycles before it admitted — their rows, their quota usage and their
    TopologyAssignments as TAS usage of the leaves (workload.TASUsage) — and takes the next head of every ClusterQueue. `cycle_input(c)`
    -> (Snapshot [derived], Heads, CycleTAS); `fold(heads, decisions, tas_out)` turns the cycle's admissions into admitted workloads.
    The same object feeds the engine and the oracle, so a parity gate over k cycles compares k dependent cycles."""

    def __init__(self, cqs, cohorts, topologies, workloads_of, hold: int = 0, failures: int = 0, failure_seed: int = 4242):
        """hold > 0: a workload finishes `hold` cycles after the cycle that admitted it (its row, quota and leaf usage leave the cache).
        failures > 0: in every cycle that many nodes hosting admitted pods are gone from the snapshot (NotReady); every admitted workload
        with pods on one of them comes back as a head on its SECOND pass (workload.NeedsSecondPass workload.go:974, manager.go:923) next
        to the first-pass heads — the failed node's pods are re-placed (tas_flavor_snapshot.go:608-633) or, when that is impossible, the
        workload is evicted (TASFailedNodeReplacementFailFast). The nodes are back in the next cycle."""
        self.cqs, self.cohorts, self.topologies, self.workloads_of, self.hold = cqs, cohorts, topologies, workloads_of, hold
        self.failures, self.failure_seed = failures, failure_seed
        self.admitted = []          # api.Workload with PodSet.flavors
        self.admitted_tas = {}      # name -> [AdmittedTAS]
        self.admitted_pod_tas = {}  # name -> [PodSetTAS] (the workload's topology requests: a second pass places by them)
        self.cycle = 0
        self.second_pass = dict(heads=0, replaced=0, evicted=0, pending=0)

    def cycle_input(self):
        import copy
        from .api import Heads, Snapshot
        from .tas_cycle import CycleTAS, HeadAdmission
        snap = Snapshot(self.cqs, self.cohorts, list(self.admitted), extra_resources=["pods"])
        snap.derive()
        wls, pod_tas = self.workloads_of(self.cycle)
        topologies, second, head_adm = self.topologies, [], {}
        if self.failures > 0 and self.admitted_tas:
            rng = np.random.default_rng(self.failure_seed + self.cycle)
            hosting = sorted({vals[-1] for tas in self.admitted_tas.values() for a in tas for vals, _ in a.domains})
            failed = set(rng.choice(hosting, size=min(self.failures, len(hosting)), replace=False).tolist())
            topologies = {name: t.without_nodes(failed) for name, t in self.topologies.items()}
            for w in self.admitted:
                tas = self.admitted_tas.get(w.name)
                lost = sorted({vals[-1] for a in (tas or []) for vals, _ in a.domains if vals[-1] in failed})
                if not lost:
                    continue
                hw = copy.deepcopy(w)
                hw.has_quota_reservation = hw.has_unhealthy_nodes = hw.unhealthy_assignment = True
                doms = [None] * len(w.pod_sets)
                for pi, a in enumerate(tas):   # (one AdmittedTAS per podset, in podset order: fold() below)
                    doms[pi] = [(tuple(v), int(c)) for v, c in a.domains]
                head_adm[hw.name] = HeadAdmission([dict(ps.flavors) for ps in w.pod_sets], doms, lost, admitted=True)
                for pi, pt in enumerate(self.admitted_pod_tas[w.name]):
                    pod_tas[(hw.name, pi)] = pt
                second.append(hw)
        heads = Heads(snap, second + wls, cycle=self.cycle + 1)
        self._pod_tas = pod_tas
        self.second_pass["heads"] += len(second)
        return snap, heads, CycleTAS(snap, heads, topologies, pod_tas, admitted_tas=self.admitted_tas, head_admission=head_adm or None)

    def fold(self, heads, d, tout) -> int:
        """The cycle's admissions -> admitted workloads (flavors from the decision, TAS usage from its TopologyAssignment). -> how many."""
        import copy
        from . import _ffi as F
        from .tas_cycle import AdmittedTAS
        n = 0
        for i, w in enumerate(heads.workloads):
            act = int(d.a["action"][i])
            if w.has_unhealthy_nodes:
                # the second pass of an admitted workload: the replaced TopologyAssignment takes the place of the old one (Scheduler.admit
                # scheduler.go:991-1011), an eviction takes the workload out of the cache, anything else leaves it as it was
                if act == F.ACT_ADMIT:
                    self.admitted_tas[w.name] = [AdmittedTAS(ta[0], [(tuple(vals), cnt) for vals, cnt in ta[1]], dict(self._pod_tas[(w.name, pi)].single_pod_requests))
                                                 for pi in range(len(w.pod_sets)) for ta in [tout.topology_assignment(i, pi)] if ta is not None]
                    self.second_pass["replaced"] += 1
                elif act == F.ACT_EVICT:
                    self.admitted = [a for a in self.admitted if a.name != w.name]
                    self.admitted_tas.pop(w.name, None); self.admitted_pod_tas.pop(w.name, None)
                    self.second_pass["evicted"] += 1
                else:
                    self.second_pass["pending"] += 1
                continue
            if act != F.ACT_ADMIT:
                continue
            fl = d.flavors_of(i)
            aw = copy.deepcopy(w)
            aw.reserve_ts = 10 ** 12 + self.cycle * 10 ** 6 + i
            tas = []
            for pi, ps in enumerate(aw.pod_sets):
                ps.flavors = {r: v[0] for r, v in fl[pi].items()}
                ta = tout.topology_assignment(i, pi)
                if ta is not None:
                    tas.append(AdmittedTAS(ta[0], [(tuple(vals), cnt) for vals, cnt in ta[1]], dict(self._pod_tas[(w.name, pi)].single_pod_requests)))
            aw._admitted_in = self.cycle
            self.admitted.append(aw)
            if tas:
                self.admitted_tas[aw.name] = tas
                self.admitted_pod_tas[aw.name] = [self._pod_tas[(w.name, pi)] for pi in range(len(aw.pod_sets))]
            n += 1
        if self.hold > 0:
            gone = [w for w in self.admitted if w._admitted_in <= self.cycle - self.hold]
            for w in gone:
                self.admitted_tas.pop(w.name, None); self.admitted_pod_tas.pop(w.name, None)
            self.admitted = [w for w in self.admitted if w._admitted_in > self.cycle - self.hold]
        self.cycle += 1
        return n

"""Seeded synthetic populations for the BASELINE.json configurations.

Shapes follow the reference's scalability harness (test/performance/scheduler/configs/*/generator.yaml:
small/medium/large workload classes, cohorts x ClusterQueues) scaled to the north-star sizes.
Everything is emitted directly in the flat boundary schema (api.Snapshot / api.Heads).

  cfg 1  100 wl,   4 CQ,    2 flavors, no cohort, StrictFIFO                       (CPU plumbing)
  cfg 2  10k wl,   128 CQ,  8 flavors, flat cohort, BestEffortFIFO, no preemption
  cfg 3  100k wl,  1000 CQ, 16 flavors, 3-level cohorts, borrowing + lending limits
  cfg 4  cfg 3 + preemption policies (+ fair sharing weights; run with fair_sharing=True once the
         device path implements it; "cfg3p" = cfg 4's policies under classical preemption)
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List

import numpy as np

from .api import (ClusterQueue, Cohort, FlavorQuotas, Heads, PodSet, ResourceGroup, ResourceQuota, Snapshot, Workload)

GI = 1 << 30
MAXI64 = (1 << 63) - 1
BASE_SEED = 20260921


@dataclass
class Population:
    name: str
    snapshot: Snapshot
    # pending side: per-CQ FIFO of pre-digested workloads, SoA over all W workloads
    w_cq: np.ndarray        # [W] CQ index, sorted by CQ then queue order
    w_prio: np.ndarray
    w_ts: np.ndarray
    w_nps: np.ndarray       # podsets per workload
    ps_count: np.ndarray    # [n_ps]
    ps_req: np.ndarray      # [n_ps, 3] cpu milli, memory bytes, gpu  (totals for the podset)
    cq_w_off: np.ndarray    # [n_cq+1]
    fair_sharing: bool = False
    preemption: bool = False

    @property
    def n_pending(self) -> int:
        return int(len(self.w_cq))

    def heads_for_cycle(self, c: int, cycle: int = 1, limit: int = 0) -> Heads:
        """Heads of cycle c = the c-th workload of every ClusterQueue queue (<= 1 head per CQ,
        pkg/cache/queue/manager.go:922), CQ-name order. limit > 0: an evenly spaced sample of that batch."""
        snap = self.snapshot
        idx = self.cq_w_off[:-1] + c
        idx = idx[idx < self.cq_w_off[1:]]
        if limit and limit < len(idx):
            idx = idx[np.linspace(0, len(idx) - 1, limit).astype(np.int64)]
        return self._heads(idx, cycle)

    def all_heads(self, cycle: int = 1) -> Heads:
        """Every pending workload as one batch ("nominate-all-pending")."""
        return self._heads(np.arange(self.n_pending), cycle)

    def pending(self, hashes: bool = True):
        """Every pending workload as the device-resident pending set (api.Pending). hashes: a SchedulingHash per workload shape
        (priority, per-podset count and requests — what computeSchedulingHash digests, workload.go:389), so that equal-shaped workloads
        of a ClusterQueue form equivalence classes; False leaves the hash unknown (0)."""
        from .api import Pending
        hb = self._heads(np.arange(self.n_pending), 1)
        if hashes:
            a = hb.arrays
            h = (a["priority"].astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
            ps_owner = np.repeat(np.arange(hb.n), a["ps_off"][1:] - a["ps_off"][:-1])
            nreq = int(a["ps_req_off"][1] - a["ps_req_off"][0]) if hb.n_ps else 0
            q = a["req_qty"].reshape(hb.n_ps, nreq).astype(np.uint64) if nreq else np.zeros((hb.n_ps, 0), np.uint64)
            ph = a["ps_count"].astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
            for r in range(nreq):
                ph = (ph ^ (q[:, r] + np.uint64(r + 1))) * np.uint64(0x100000001B3)
            np.add.at(h, ps_owner, ph)  # order-insensitive over podsets is enough for a synthetic shape id
            h |= np.uint64(1)           # 0 = SchedulingHashUnknown
            a["hash"] = h
            hb._struct = None
        return Pending(hb, uid_rank=np.arange(hb.n, dtype=np.uint32))

    def _heads(self, idx: np.ndarray, cycle: int) -> Heads:
        snap = self.snapshot
        nR, nF = snap.n_resource, snap.n_flavor
        ps_first = np.concatenate([[0], np.cumsum(self.w_nps)])[:-1]
        nps = self.w_nps[idx]
        ps_off = np.concatenate([[0], np.cumsum(nps)]).astype(np.int32)
        ps_idx = np.concatenate([np.arange(ps_first[i], ps_first[i] + self.w_nps[i]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
        res_ids = [snap.resource_index[r] for r in ("cpu", "memory", "example.com/gpu") if r in snap.resource_index]
        nreq = len(res_ids)
        n_ps = len(ps_idx)
        a: Dict[str, np.ndarray] = dict(
            cq=self.w_cq[idx].astype(np.int32), priority=self.w_prio[idx].astype(np.int64), queue_ts=self.w_ts[idx].astype(np.int64),
            flags=np.zeros(len(idx), np.uint32), ps_off=ps_off,
            ps_count=self.ps_count[ps_idx].astype(np.int32), ps_min_count=np.full(n_ps, -1, np.int32),
            ps_req_off=(np.arange(n_ps + 1) * nreq).astype(np.int32),
            req_res=np.tile(np.array(res_ids, np.int32), n_ps),
            req_qty=self.ps_req[ps_idx][:, :nreq].reshape(-1).astype(np.int64),
            ps_flavor_ok=np.full(n_ps * ((nF + 63) // 64), (1 << 64) - 1, np.uint64),
            ps_last_tried=np.full(n_ps * nR, -1, np.int32),
            last_generation=np.zeros(len(idx), np.int64), last_cycle=np.zeros(len(idx), np.int64),
            last_hash=np.zeros(len(idx), np.uint64), hash=np.zeros(len(idx), np.uint64),
        )
        return Heads.from_arrays(snap, a, cycle=cycle)


def _quota(rng, unit, lo, hi, limits: bool, unlimited_p: float):
    nominal = int(rng.integers(lo, hi + 1)) * unit
    if unlimited_p and rng.random() < unlimited_p:
        nominal = MAXI64
    bl = ll = None
    if nominal == MAXI64:
        # an Unlimited cell keeps a finite lending limit, otherwise one cell makes the whole tree unconstrained
        ll = int(rng.integers(0, hi + 1)) * unit
    if limits and nominal != MAXI64:
        if rng.random() < 0.5:
            bl = int(rng.integers(0, nominal // unit + 1)) * unit
        if rng.random() < 0.5:
            ll = int(rng.integers(0, nominal // unit + 1)) * unit
    return ResourceQuota(nominal, bl, ll)


def generate(cfg: int, seed: int = BASE_SEED, n_cq: int = None, per_cq: int = None, preemption: bool = None,
             fair_sharing: bool = False, fill: float = 1.0, feasible: bool = False) -> Population:
    """Build population `cfg` (1..4). n_cq / per_cq override the sizes for parity-sized variants; fill scales how full the admitted
    set leaves every flavor (1.0 = the BASELINE populations; < 1 leaves headroom at the root cohort).
    feasible (cfg 4): a start state admission could have produced — in no flavor-resource does the ROOT cohort's usage exceed its
    SubtreeQuota, and every root cell is 80-100 % full (the spec'd cfg 4 fills every ClusterQueue to 1.0-1.5 x nominal on every flavor: 40 of 64 root cells over-committed, a
    state fits() never admits into). ClusterQueues alternate, per flavor, between borrowers (above nominal, on what their siblings
    lend) and lenders (80-95 % of it); per flavor-resource the fills are scaled until the root is 80-100 % full: pending heads meet a full tree."""
    if feasible:
        if cfg != 4:
            raise ValueError("feasible: cfg 4 only")
        fac = None
        for it in range(16):
            pop = _generate(cfg, seed, n_cq, per_cq, preemption, fair_sharing, fill, fac if fac is not None else 1.0)
            sn = pop.snapshot
            a = sn.arrays
            u, sq = a["usage"].reshape(sn.N, sn.n_fr).astype(np.float64), a["subtree_quota"].reshape(sn.N, sn.n_fr).astype(np.float64)
            roots = [n for n in range(sn.n_cq, sn.N) if a["parent"][n] < 0]
            ur, sr = u[roots].sum(0), sq[roots].sum(0)
            live = (sr > 0) & (sr < 2.0 ** 62)
            if sn.pods_resource >= 0:   # (a row's `pods` usage is its pod count, not a fill: those cells stay nearly empty)
                live[sn.pods_resource::sn.n_resource] = False
            ratio = np.where(live, ur / np.maximum(sr, 1.0), 0.0)
            over = all((u[r] <= sq[r]).all() for r in roots)
            if over and (ratio[live] >= 0.78).all():   # (gpu cells move in whole devices: 0.8 is as close as the coarse ones get)
                pop.name += "-feasible"
                return pop
            if fac is None:
                fac = np.ones(sn.n_fr)
            # every root cell towards 96 % full (usage above the guaranteed part bubbles up: not linear in the factor, hence the iteration)
            fac = fac * np.where(live, np.clip((0.96 if it < 12 else 0.93) / np.maximum(ratio, 1e-9), 0.6, 1.6), 1.0)
        raise RuntimeError("no feasible fill found")
    return _generate(cfg, seed, n_cq, per_cq, preemption, fair_sharing, fill, None)


def _generate(cfg, seed, n_cq, per_cq, preemption, fair_sharing, fill, feas):
    rng = np.random.default_rng(seed + cfg)
    if cfg == 1:
        nq, F, res, per, shape, strategy = 4, 2, ["cpu"], 25, "none", "StrictFIFO"
    elif cfg == 2:
        nq, F, res, per, shape, strategy = 128, 8, ["cpu", "memory"], 78, "flat", "BestEffortFIFO"
    elif cfg in (3, 4):
        nq, F, res, per, shape, strategy = 1000, 16, ["cpu", "memory", "example.com/gpu", "pods"], 100, "deep", "BestEffortFIFO"
    else:
        raise ValueError(cfg)
    if n_cq is not None:
        nq = n_cq
    if per_cq is not None:
        per = per_cq
    if preemption is None:
        preemption = cfg == 4
    flavors = [f"flavor-{i:02d}" for i in range(F)]
    units = {"cpu": 1000, "memory": GI, "example.com/gpu": 1, "pods": 1}
    ranges = {"cpu": (8, 64), "memory": (32, 256), "example.com/gpu": (0, 8), "pods": (50, 500)}
    cohorts: List[Cohort] = []
    cq_parent: List[str] = []
    if shape == "flat":
        cohorts = [Cohort("cohort")]
        cq_parent = ["cohort"] * nq
    elif shape == "deep":
        n_leaf = max(1, nq // 10)
        n_mid = max(1, n_leaf // 10)
        cohorts.append(Cohort("root"))
        for m in range(n_mid):
            cohorts.append(Cohort(f"mid-{m:03d}", "root"))
        for l in range(n_leaf):
            cohorts.append(Cohort(f"leaf-{l:04d}", f"mid-{(l * n_mid) // n_leaf:03d}"))
        cq_parent = [f"leaf-{(i * n_leaf) // nq:04d}" for i in range(nq)]
    cqs: List[ClusterQueue] = []
    for i in range(nq):
        fqs = []
        for f in flavors:
            fq = FlavorQuotas(f)
            for r in res:
                lo, hi = ranges[r]
                if cfg == 2:
                    q = _quota(rng, units[r], lo, hi, False, 0.0)
                    if i % 2 == 0:
                        q.borrowing_limit = 2 * q.nominal
                else:
                    q = _quota(rng, units[r], lo, hi, cfg >= 3, 0.01 if cfg >= 3 else 0.0)
                fq.resources[r] = q
            fqs.append(fq)
        cq = ClusterQueue(f"cq-{i:05d}", cohort=cq_parent[i] if cq_parent else None, resource_groups=[ResourceGroup(fqs)],
                          queueing_strategy=strategy)
        if cfg >= 3:
            cq.when_can_borrow = "TryNextFlavor" if rng.random() < 0.5 else "MayStopSearch"
        if preemption:
            cq.within_cluster_queue = "LowerPriority"
            cq.reclaim_within_cohort = "Any"
            if rng.random() < 0.5:
                cq.borrow_within_cohort = "LowerPriority"
                cq.max_priority_threshold = 1
            cq.fair_weight = float(rng.choice([0.0, 0.5, 1.0, 2.0], p=[0.02, 0.28, 0.4, 0.3]))
        cqs.append(cq)
    if shape == "deep":
        # mid cohorts own extra nominal (10% of their children's sum) on flavors 0-3
        by_mid: Dict[str, List[ClusterQueue]] = {}
        leaf_parent = {c.name: c.parent for c in cohorts if c.name.startswith("leaf-")}
        for cq in cqs:
            by_mid.setdefault(leaf_parent[cq.cohort], []).append(cq)
        for c in cohorts:
            if c.name.startswith("mid-") and c.name in by_mid:
                fqs = []
                for f in flavors[:4]:
                    fq = FlavorQuotas(f)
                    for r in res:
                        tot = sum(q.resource_groups[0].flavors[flavors.index(f)].resources[r].nominal for q in by_mid[c.name]
                                  if q.resource_groups[0].flavors[flavors.index(f)].resources[r].nominal != MAXI64)
                        fq.resources[r] = ResourceQuota((tot // 10 // units[r]) * units[r])
                    fqs.append(fq)
                c.resource_groups = [ResourceGroup(fqs)]
    # admitted set: every flavor of every CQ is filled to fill_f x nominal by k workloads, so that the
    # pending heads meet real contention (fit / borrow / nofit / preempt mixes) instead of empty flavors.
    #   cfg 2: 4 workloads on flavor 0 at 50 %          cfg 3: A = 20/CQ, fill_f ~ U(0.6, 1.1)
    #   cfg 4: A = 40/CQ, fill_f ~ U(0.9, 1.5) (aggregate ~120 % => borrowing CQs and victims)
    admitted: List[Workload] = []
    t = 0
    for i, cq in enumerate(cqs):
        for fi, f in enumerate(flavors):
            if cfg == 1:
                k, lo_f, hi_f = 0, 0.0, 0.0
            elif cfg == 2:
                k, lo_f, hi_f = (4 if fi == 0 else 0), 0.5, 0.5
            elif cfg == 3:
                k, lo_f, hi_f = (2 if fi < 4 else 1), 0.9, 1.25
            else:
                k, lo_f, hi_f = (3 if fi < 8 else 2), 1.0, 1.5
            if k == 0:
                continue
            fq = cq.resource_groups[0].flavors[fi]
            fill_f = rng.uniform(lo_f, hi_f) * fill
            if feas is not None:   # borrowers above nominal, lenders below (one draw either way: the stream stays aligned)
                fill_f = (1.02 + (fill_f - 1.0) * 0.5) if (i + fi) % 2 == 0 else (0.80 + (fill_f - 1.0) * 0.3)
            for j in range(k):
                ps = PodSet("main", count=int(rng.integers(1, 5)))
                for r in res:
                    nom = fq.resources[r].nominal
                    if nom == MAXI64:
                        nom = ranges[r][1] * units[r]
                    # (feas: a factor per flavor-resource, indexed as the snapshot does — fr = flavor * n_resource + resource, names ascending)
                    ff = fill_f if feas is None or np.isscalar(feas) else fill_f * float(feas[fi * len(res) + sorted(res).index(r)])
                    ps.requests[r] = ps.count if r == "pods" else int(nom * ff / k / units[r]) * units[r]
                    ps.flavors[r] = f
                t += 1
                admitted.append(Workload(f"{cq.name}-adm-{fi:02d}-{j}", cq.name, priority=int(rng.integers(0, 8 if preemption else 4)),
                                         creation_ts=t, pod_sets=[ps], reserve_ts=1_000_000 + t, uid=f"{t:09d}"))
    snap = Snapshot(cqs, cohorts, admitted, now_ns=10_000_000)
    snap.derive()
    # pending workloads: classes after configs/baseline/generator.yaml:15-35 (small 70% / medium 20% / large 10%)
    W = nq * per
    w_cq = np.repeat(np.arange(nq), per)
    if cfg == 1:
        cls = rng.choice(3, size=W, p=[0.7, 0.2, 0.1])
        cpu = np.array([1000, 5000, 20000])[cls]
        prio = np.array([50, 100, 200])[cls]
        w_nps = np.ones(W, np.int64)
    else:
        cpu = rng.choice([1000, 2000, 4000, 8000, 16000], size=W)
        prio = rng.integers(0, 8 if preemption else 4, size=W)
        w_nps = np.where(rng.random(W) < 0.1, 2, 1)
        if os.environ.get("KQ_POP_ONE_PODSET"):   # (experiment: no admitted row ever holds more than one flavor's resources)
            w_nps = np.ones(W, np.int64)
    n_ps = int(w_nps.sum())
    ps_owner = np.repeat(np.arange(W), w_nps)
    ps_count = rng.integers(1, 5, size=n_ps)
    ps_cpu = cpu[ps_owner]
    ps_req = np.stack([ps_cpu, ps_cpu // 1000 * 4 * GI, (rng.random(n_ps) < 0.3) * rng.integers(0, 3, size=n_ps)], axis=1).astype(np.int64)
    # per-CQ queue order = heap order (priority desc, timestamp asc; cluster_queue.go:844)
    ts = np.arange(W, dtype=np.int64) * 1000 + 5_000_000
    rng.shuffle(ts)
    order = np.lexsort((ts, -prio, w_cq))
    first_ps = np.concatenate([[0], np.cumsum(w_nps)])[:-1]
    ps_order = np.concatenate([np.arange(first_ps[i], first_ps[i] + w_nps[i]) for i in order])
    cq_w_off = np.concatenate([[0], np.cumsum(np.bincount(w_cq, minlength=nq))]).astype(np.int64)
    return Population(name=f"cfg{cfg}", snapshot=snap, w_cq=w_cq[order], w_prio=prio[order].astype(np.int64), w_ts=ts[order],
                      w_nps=w_nps[order], ps_count=ps_count[ps_order], ps_req=ps_req[ps_order], cq_w_off=cq_w_off,
                      fair_sharing=fair_sharing, preemption=bool(preemption))

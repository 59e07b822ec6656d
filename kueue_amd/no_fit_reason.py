"""Assignment.NoFitReason and the NoFitReason of every FlavorAssignmentAttempt (features.UnadmittedWorkloadsObservability) regenerated on the
host from one head's reason records and decisions — the Python mirror of shim/go/no_fit_reason.go.

The reference labels an attempt while the flavor scan runs (flavorassigner.go:1106 NoMatchingFlavor for a flavor checkFlavorForPodSets
rejects, :1136 for a slice flavor mismatch, fitsResourceQuota :1342-1380 ExceedsMaxQuota / WaitingForQuota, markFlavorAttempt :897
TopologyPlacementFailed) and folds the labels in resolveNoFitReason (:947-994). The engine's records (include/kq_engine.h KQ_RSN_*) name the
same events per (podset, flavor, resource); two things they do not say are recovered from the snapshot the cycle ran on:

 * whether an "insufficient unused quota" attempt ended NoFit (WaitingForQuota) or went to the preemption oracle (no label, mode Preempt —
   fromPreemptionPossibility never answers noFit): the branch condition `Nominal >= val || mayReclaimInHierarchy || canPreemptWhileBorrowing`
   (:1375) is a function of the quota tree alone, with val = the record's "more needed" operand + Available(fr);
 * that the simulate-empty TAS pass failed (:893-899): the podset keeps its flavors with every mode NoFit, and no quota record explains it.

Not recoverable: the NoFit attempts of a flavor scan that ended in Fit — the scan returns a nil status (:1189, :1206), the reference drops its
reasons and the engine its records; resolveNoFitReason reads them only when ANOTHER resource group of the same podset then fails (they
can only raise the label of that podset). Not reproduced: mergeFlavorAttemptsForResource's "flavor %s does not provide resource %s" relabelling (flavor_assigner_attempts.go:104-119),
which only fires for a flavor listed in two resource groups of one ClusterQueue whose second scan stops before reaching it; and the
records of a head whose assignment was recomputed inside the cycle describe the snapshot of that moment, not the one put.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Set, Tuple

from . import _ffi as F
from . import messages as M

LABELS = ["", "TopologyPlacementFailed", "WaitingForQuota", "ExceedsMaxQuota", "NoMatchingFlavor"]   # reasonSeverity flavorassigner.go:306-327
NONE, TOPOLOGY, WAITING, EXCEEDS, NO_MATCH = range(5)
NOFIT, PREEMPT, DEFERRED_FIT, FIT = 0, 1, 2, 3   # FlavorAssignmentMode as kq_decisions carries it (KQ_MODE_*; DeferredFit is set after Assign)
U = (1 << 63) - 1
MINI = -(1 << 63)


def _sat(x: int) -> int:
    return U if x > U else MINI if x < MINI else x


def _add(x: int, y: int) -> int:   # resources.Amount.Add amount.go:114-128
    return U if (x == U or y == U) else _sat(x + y)


def _sub(x: int, y: int) -> int:   # Amount.Sub :130-145
    if x == U and y == U:
        return 0
    if x == U:
        return U
    if y == U:
        return MINI
    return _sat(x - y)


class QuotaTree:
    """resource_node.go over a derived Snapshot's arrays: what fitsResourceQuota's branch condition reads."""

    def __init__(self, snap):
        assert snap.derived, "the snapshot must be derived (subtree quotas / cohort usage)"
        a = snap.arrays
        self.nq, self.nfr = snap.n_cq, snap.n_fr
        self.parent = a["parent"].tolist()
        self.nominal = a["nominal"]; self.bl = a["borrow_limit"]; self.ll = a["lend_limit"]
        self.sq = a["subtree_quota"]; self.usage = a["usage"]

    def _k(self, n, fr):
        return n * self.nfr + fr

    def local_quota(self, n, fr):   # :67-72
        k = self._k(n, fr)
        if int(self.ll[k]) != F.KQ_NIL_LIMIT:
            return max(0, _sub(int(self.sq[k]), int(self.ll[k])))
        return 0

    def local_available(self, n, fr):   # :92-95
        return max(0, _sub(self.local_quota(n, fr), int(self.usage[self._k(n, fr)])))

    def _available(self, n, fr):   # :106-122
        k = self._k(n, fr)
        if self.parent[n] < 0:
            return _sub(int(self.sq[k]), int(self.usage[k]))
        pa = self._available(self.parent[n], fr)
        if int(self.bl[k]) != F.KQ_NIL_LIMIT:
            lq = self.local_quota(n, fr)
            stored = _sub(int(self.sq[k]), lq)
            used = max(0, _sub(int(self.usage[k]), lq))
            pa = min(_add(_sub(stored, used), int(self.bl[k])), pa)
        return _add(self.local_available(n, fr), pa)

    def available(self, cq, fr):   # clusterqueue_snapshot.go:167
        return max(0, self._available(cq, fr))

    def borrowing_with(self, n, fr, val):   # clusterqueue_snapshot.go:155-161 / cohort_snapshot.go:90
        k = self._k(n, fr)
        quota = int(self.nominal[k]) if n < self.nq else int(self.sq[k])
        return quota < _add(int(self.usage[k]), val)

    def may_reclaim_in_hierarchy(self, cq, fr, val):   # FindHeightOfLowestSubtreeThatFits classical/hierarchical_preemption.go:221-234, 2nd result
        has_parent = self.parent[cq] >= 0
        if not self.borrowing_with(cq, fr, val) or not has_parent:
            return has_parent
        remaining = _sub(val, self.local_available(cq, fr))
        t = self.parent[cq]
        while t >= 0:
            if not self.borrowing_with(t, fr, remaining):
                return self.parent[t] >= 0
            remaining = _sub(remaining, self.local_available(t, fr))
            t = self.parent[t]
        return False


def can_preempt_while_borrowing(policy: int, fair_sharing: bool) -> bool:   # flavorassigner.go:1386-1389
    borrow_within = (policy >> 4) & 1
    reclaim, reclaim_unset = (policy >> 2) & 3, (policy >> 11) & 1
    return bool(borrow_within) or (fair_sharing and (reclaim != 0 or bool(reclaim_unset)))


def flavor_attempts(dec, i: int, fair_sharing: bool, tas_flavors: Optional[Set[int]] = None, tree: Optional[QuotaTree] = None):
    """-> (Assignment.NoFitReason, [per podset {flavor index: (mode, label)}]) of head i; labels as indices of LABELS. The podsets after the
    first one that failed are not listed (assignFlavors returns there, :848-853)."""
    snap, heads, a = dec.snap, dec.heads, dec.a
    tree = tree or QuotaTree(snap)
    nR = snap.n_resource
    cq = int(heads.arrays["cq"][i])
    policy = int(snap.arrays["cq_policy"][cq])
    p0, p1 = int(heads.arrays["ps_off"][i]), int(heads.arrays["ps_off"][i + 1])
    recs: List[List[Tuple[int, int, int, int]]] = [[] for _ in range(p1 - p0)]
    for k in range(int(a["rsn_off"][i]), int(a["rsn_off"][i + 1])):
        code = int(a["rsn_code"][k])
        if code == M.RSN_TRUNCATED:
            raise OverflowError("reason window of the head overflowed: raise rsn_cap")
        recs[int(a["rsn_podset"][k])].append((code, int(a["rsn_flavor"][k]), int(a["rsn_resource"][k]), int(a["rsn_a"][k])))
    out: List[Dict[int, Tuple[int, int]]] = []
    modes: List[int] = []
    for lp in range(p1 - p0):
        g = p0 + lp
        att: Dict[int, List[int]] = {}

        def mark(fl, mode, label):
            e = att.setdefault(fl, [FIT, NONE])
            e[0] = min(e[0], mode); e[1] = max(e[1], label)

        for code, fl, rs, more in recs[lp]:
            if code in (M.RSN_FLAVOR_INELIGIBLE, M.RSN_SLICE_FLAVOR_MISMATCH):
                mark(fl, NOFIT, NO_MATCH)
            elif code == M.RSN_EXCEEDS_MAX_CAPACITY:
                mark(fl, NOFIT, EXCEEDS)
            elif code == M.RSN_INSUFFICIENT_UNUSED:
                fr = fl * nR + rs
                val = _add(more, tree.available(cq, fr))
                simulated = (int(tree.nominal[cq * tree.nfr + fr]) >= val or tree.may_reclaim_in_hierarchy(cq, fr, val)
                             or can_preempt_while_borrowing(policy, fair_sharing))
                mark(fl, PREEMPT, NONE) if simulated else mark(fl, NOFIT, WAITING)
        fls = [(int(a["flavor"][g * nR + r]), int(a["res_mode"][g * nR + r])) for r in range(nR) if int(a["flavor"][g * nR + r]) >= 0]
        # PodSetAssignment.RepresentativeMode :386-404: no reasons -> Fit (even without flavors: nothing was requested of this ClusterQueue)
        if not recs[lp]:
            mode = FIT
        elif not fls:
            mode = NOFIT
        else:
            mode = min(m for _, m in fls)
            if mode == NOFIT:   # the flavors were kept and every mode dropped to NoFit: the simulate-empty TAS pass failed on this podset (:893-899)
                for fl, _ in fls:
                    if tas_flavors is None or fl in tas_flavors:
                        att[fl] = [NOFIT, TOPOLOGY]   # markFlavorAttempt overwrites mode and reason
                        break
        out.append({fl: (m, lb) for fl, (m, lb) in att.items()})
        modes.append(mode)
        if recs[lp] and not fls:
            break   # assignFlavors returns at the first podset that got no flavor (:848-853)
    if not modes or min(modes) != NOFIT:   # resolveNoFitReason :948
        return NONE, out
    rg_of: Dict[int, List[int]] = {}
    for rg in range(int(snap.arrays["cq_rg_off"][cq]), int(snap.arrays["cq_rg_off"][cq + 1])):
        for k in range(int(snap.arrays["rg_flavor_off"][rg]), int(snap.arrays["rg_flavor_off"][rg + 1])):
            rg_of.setdefault(int(snap.arrays["rg_flavor"][k]), []).append(rg)
    overall = NONE
    for att, mode in zip(out, modes):
        if mode != NOFIT:
            continue
        if not att:
            overall = max(overall, NO_MATCH)
            continue
        rg_min: Dict[int, int] = {}
        for fl, (m, lb) in att.items():
            if m != NOFIT:
                continue
            for rg in rg_of.get(fl, []):
                if rg not in rg_min or lb < rg_min[rg]:
                    rg_min[rg] = lb
        overall = max([overall] + list(rg_min.values()))
    return overall, out


def quota_reserved_reason(dec, i: int, fair_sharing: bool, tas_flavors: Optional[Set[int]] = None, tree: Optional[QuotaTree] = None) -> str:
    """entry.quotaReservedReason as processEntry leaves it (scheduler.go:424-513) — the Reason of the QuotaReserved=False condition
    requeueAndUpdate patches (:1188, UnadmittedWorkloadReasonWithFallback; the gate is on by default) — for a head of a cycle. "" for a head that
    was admitted. The reasons nominate sets before an entry reaches processEntry (Misconfigured / Suspended / PendingEvaluation :678-691:
    inactive ClusterQueue, namespace mismatch, admission checks) belong to heads the host never sends across the boundary."""
    a = dec.a
    if int(a["status"][i]) in (F.ST_ASSUMED, F.ST_EVICTED):   # admitted; or evicted by handleFailedTASReplacement (:426-429), which sets no reason
        return ""
    if int(a["skip"][i]) in (F.SKIP_OVERLAP, F.SKIP_NO_LONGER_FITS):   # :471-484
        return "WaitingForQuota"
    mode = int(a["mode"][i])
    if mode == NOFIT:                                                  # :430-435
        return LABELS[flavor_attempts(dec, i, fair_sharing, tas_flavors, tree)[0]]
    if mode == DEFERRED_FIT:                                           # :455-468
        return "WaitingForPreemptedWorkloads"
    if int(a["action"][i]) == F.ACT_PREEMPT:                           # :495
        return "WaitingForPreemptedWorkloads"
    if mode == PREEMPT and int(a["tgt_off"][i + 1]) == int(a["tgt_off"][i]):   # :437-443
        return "WaitingForQuota"
    return ""

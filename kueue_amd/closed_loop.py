"""SURVEY §8d's "run" for a population WITH preemption: (*Scheduler).schedule (scheduler.go:308-386) closed over the cache, cycle after cycle.

What the reference does between two cycles, and what this driver hands to the engine as ONE kq_snapshot_patch_rows(KQ_ROWS_FOLD_USAGE):
  * an entry that was assumed (scheduler.go:605 admit -> cache.AssumeWorkload -> clusterQueue.updateWorkloadUsage clusterqueue.go:594) is an
    admitted workload from the next snapshot on: a new ROW (its ClusterQueue, priority, queue / reservation time, uid, usage = the
    assignment's usage) whose usage enters the quota tree;
  * the targets of an entry in Preempt mode get the Evicted condition (preemption.go:201-270 IssuePreemptions): they stay admitted — marked,
    and first in every later candidate order (CandidatesOrdering) — until their pods are gone; here: for one cycle. The preemptor went back
    to its heap with RequeueReasonPendingPreemption (kq_pending_apply) and is admitted once the quota is free;
  * a workload whose time is up (`hold` cycles after its admission) finishes: its row leaves, its usage leaves the tree, and the
    inadmissible workloads of that root cohort go back to their heaps (QueueAssociatedInadmissibleWorkloadsAfter).
The victims do not come back as pending workloads (their Jobs are taken to be deleted once evicted: no back-off requeue is modelled); rows of
the initial snapshot never finish on their own.

The driver is host logic over the C ABI: it works on the HIP engine (kueue_amd.engine.Engine) and on the test-only emulation alike and keeps
no copy of the snapshot — only, per resident row, the cycle it finishes in and whether it is marked. shim/go/closed_loop.go is its twin."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _ffi as F
from .api import Decisions

NEVER = np.iinfo(np.int64).max


def assignment_rows(snap, heads_arrays: dict, d: Decisions, sel: np.ndarray, reserve_ts: int, uid_rank: np.ndarray) -> dict:
    """The admitted-workload rows of the heads `sel` of a cycle (kq_row_patch.add_*): usage = Assignment.Usage (flavorassigner.go:1017-1041) —
    per (podset, resource) that was given a flavor, the podset's request scaled to the admitted count (the injected `pods` request: the count).
    A row's usage entries stand in the order Assignment.append meets them (podset by podset, resources ascending). Vectorised: the loop runs
    once per cycle over up to a thousand heads."""
    a = heads_arrays
    nR, nfr, pods = snap.n_resource, snap.n_fr, snap.pods_resource
    sel = np.asarray(sel, np.int64)
    n = len(sel)
    ps0, ps1 = a["ps_off"][sel].astype(np.int64), a["ps_off"][sel + 1].astype(np.int64)
    nps = ps1 - ps0
    ps = np.concatenate([np.arange(x, y) for x, y in zip(ps0, ps1)]).astype(np.int64) if n else np.zeros(0, np.int64)
    owner = np.repeat(np.arange(n), nps)
    # the requests of those podsets as a dense [podsets, resources] table
    r0, r1 = a["ps_req_off"][ps].astype(np.int64), a["ps_req_off"][ps + 1].astype(np.int64)
    e = np.concatenate([np.arange(x, y) for x, y in zip(r0, r1)]).astype(np.int64) if len(ps) else np.zeros(0, np.int64)
    req = np.zeros((len(ps), nR), np.int64)
    req[np.repeat(np.arange(len(ps)), r1 - r0), a["req_res"][e]] = a["req_qty"][e]
    cnt0, cnt = a["ps_count"][ps].astype(np.int64), d.a["ps_count"][ps].astype(np.int64)
    sc = (cnt0 != 0) & (cnt0 != cnt)
    if sc.any():
        req[sc] = (req[sc] // cnt0[sc, None]) * cnt[sc, None]       # ScaledTo workload.go:317-340
    if pods >= 0:
        cov = _pods_covered(snap)[a["cq"][sel]][owner]
        req[cov, pods] = cnt[cov]                                   # flavorassigner.go:743-749
    fl = d.a["flavor"].reshape(-1, nR)[ps].astype(np.int64)
    pi, ri = np.nonzero(fl >= 0)                                    # row-major: podset by podset, resources ascending
    fr = fl[pi, ri] * nR + ri
    key = owner[pi] * nfr + fr
    uk, first, inv = np.unique(key, return_index=True, return_inverse=True)
    qty = np.zeros(len(uk), np.int64)
    np.add.at(qty, inv, req[pi, ri])
    order = np.lexsort((first, uk // nfr))                          # per head, in the order of first appearance
    uk, qty = uk[order], qty[order]
    use_off = np.concatenate([[0], np.cumsum(np.bincount(uk // nfr, minlength=n))]).astype(np.int32)
    return dict(cq=a["cq"][sel].astype(np.int32), priority=a["priority"][sel].astype(np.int64), queue_ts=a["queue_ts"][sel].astype(np.int64),
                reserve_ts=np.full(n, reserve_ts, np.int64), uid_rank=np.asarray(uid_rank, np.uint32), flags=np.zeros(n, np.uint8),
                use_off=use_off, use_fr=(uk % nfr).astype(np.int32), use_qty=qty)


def assignment_rows_loop(snap, heads_arrays: dict, d: Decisions, sel: np.ndarray, reserve_ts: int, uid_rank: np.ndarray) -> dict:
    """assignment_rows head by head (the readable form; tests hold the two against each other)."""
    a = heads_arrays
    nR = snap.n_resource
    pods = snap.pods_resource
    cov = _pods_covered(snap) if pods >= 0 else None
    cq, prio, qts, uo, ufr, uq = [], [], [], [0], [], []
    for i in sel:
        i = int(i)
        use = {}
        for p in range(int(a["ps_off"][i]), int(a["ps_off"][i + 1])):
            cnt0, cnt = int(a["ps_count"][p]), int(d.a["ps_count"][p])
            req = {int(a["req_res"][e]): int(a["req_qty"][e]) for e in range(int(a["ps_req_off"][p]), int(a["ps_req_off"][p + 1]))}
            for r in range(nR):
                f = int(d.a["flavor"][p * nR + r])
                if f < 0:
                    continue
                q = req.get(r, 0)
                if cnt0 != 0 and cnt0 != cnt:
                    q = (q // cnt0) * cnt
                if r == pods and cov[int(a["cq"][i])]:
                    q = cnt
                fr = f * nR + r
                use[fr] = use.get(fr, 0) + q
        cq.append(int(a["cq"][i])); prio.append(int(a["priority"][i])); qts.append(int(a["queue_ts"][i]))
        for fr, q in use.items():
            ufr.append(fr); uq.append(q)
        uo.append(len(ufr))
    n = len(cq)
    return dict(cq=np.array(cq, np.int32), priority=np.array(prio, np.int64), queue_ts=np.array(qts, np.int64),
                reserve_ts=np.full(n, reserve_ts, np.int64), uid_rank=np.asarray(uid_rank, np.uint32), flags=np.zeros(n, np.uint8),
                use_off=np.array(uo, np.int32), use_fr=np.array(ufr, np.int32), use_qty=np.array(uq, np.int64))


_PODS_CACHE: dict = {}


def _pods_covered(snap) -> np.ndarray:
    """[n_cq] flavorassigner.go:743-749: the assigner sets requests[pods] = count when the ClusterQueue has a resource group covering `pods`."""
    cov = _PODS_CACHE.get(id(snap))
    if cov is None or cov[0] is not snap:
        a, pods = snap.arrays, snap.pods_resource
        c = np.zeros(snap.n_cq, bool)
        for q in range(snap.n_cq):
            for g in range(int(a["cq_rg_off"][q]), int(a["cq_rg_off"][q + 1])):
                if pods in a["rg_res"][int(a["rg_res_off"][g]):int(a["rg_res_off"][g + 1])]:
                    c[q] = True
        cov = (snap, c)
        _PODS_CACHE.clear(); _PODS_CACHE[id(snap)] = cov
    return cov[1]


class RowBook:
    """What the driver remembers of the resident admitted table: per row its ClusterQueue, the cycle it finishes in and the cycle it was marked
    Evicted in. place() reproduces where kq_snapshot_patch_rows puts things: kept rows keep their order inside their ClusterQueue, added rows
    land behind them in the order given."""

    def __init__(self, snap):
        off = snap.arrays["cq_adm_off"]
        self.nq = snap.n_cq
        self.cq = np.repeat(np.arange(self.nq, dtype=np.int32), np.diff(off)).astype(np.int32)
        self.finish = np.full(len(self.cq), NEVER, np.int64)
        self.evicted_at = np.where(snap.arrays["adm_flags"][:len(self.cq)] & F.ADM_EVICTED, -1, NEVER).astype(np.int64) if len(self.cq) else np.zeros(0, np.int64)

    @property
    def n(self) -> int:
        return int(len(self.cq))

    def place(self, remove: np.ndarray, add_cq: np.ndarray, add_finish: np.ndarray, new_index: Optional[np.ndarray] = None):
        keep = np.ones(self.n, bool)
        keep[remove] = False
        kept_cq = self.cq[keep]
        k_cnt = np.bincount(kept_cq, minlength=self.nq)
        a_cnt = np.bincount(add_cq, minlength=self.nq) if len(add_cq) else np.zeros(self.nq, np.int64)
        off = np.concatenate([[0], np.cumsum(k_cnt + a_cnt)])
        n_new = int(off[-1])
        cq = np.empty(n_new, np.int32); fin = np.empty(n_new, np.int64); ev = np.empty(n_new, np.int64)
        # kept rows: position inside the ClusterQueue = rank among the kept rows of it (the table is grouped by ClusterQueue already)
        kpos = off[kept_cq] + (np.arange(len(kept_cq)) - np.concatenate([[0], np.cumsum(k_cnt)])[kept_cq])
        cq[kpos] = kept_cq; fin[kpos] = self.finish[keep]; ev[kpos] = self.evicted_at[keep]
        # added rows: behind the kept rows of their ClusterQueue, in the order given (rank among the added rows of the same ClusterQueue)
        add_cq = np.asarray(add_cq, np.int64)
        o = np.argsort(add_cq, kind="stable")
        rank = np.empty(len(add_cq), np.int64)
        rank[o] = np.arange(len(add_cq)) - np.concatenate([[0], np.cumsum(a_cnt)])[add_cq[o]]
        apos = (off[:-1] + k_cnt)[add_cq] + rank
        cq[apos] = add_cq; fin[apos] = add_finish; ev[apos] = NEVER
        if new_index is not None:   # (the engine's own answer, when it gave one: must agree)
            want = np.full(self.n, -1, np.int64); want[np.nonzero(keep)[0]] = kpos
            assert np.array_equal(want, np.asarray(new_index[:self.n], np.int64)), "row bookkeeping and kq_snapshot_patch_rows disagree"
        self.cq, self.finish, self.evicted_at = cq, fin, ev
        return kpos, apos


def cycle_patch(book: RowBook, snap, clock: int, uid_base: int, cycle: int, d, ha, wl):
    """The kq_row_patch of a cycle from its decisions: (remove_rows, add dict or None, evict_rows, heads admitted, heads preempting).
    remove = the rows whose time is up + the rows marked Evicted by an EARLIER cycle; add = the heads the cycle admitted; evict = its preemption
    targets that are still there and not marked yet."""
    gone = (book.finish <= cycle) | ((book.evicted_at != NEVER) & (book.evicted_at < cycle))
    remove = np.nonzero(gone)[0].astype(np.int32)
    add, evict, n_adm, n_pre = None, np.zeros(0, np.int32), 0, 0
    if d is not None:
        n = d.n
        act = d.a["action"][:n]
        adm = np.nonzero(act == F.ACT_ADMIT)[0]
        n_adm = len(adm)
        if n_adm:
            add = assignment_rows(snap, ha, d, adm, clock, uid_base + wl[adm])
        pre = np.nonzero(act == F.ACT_PREEMPT)[0]
        n_pre = len(pre)
        t = [d.a["tgt_adm"][int(d.a["tgt_off"][i]):int(d.a["tgt_off"][i + 1])] for i in pre]
        if t:
            tg = np.unique(np.concatenate(t)).astype(np.int32)
            evict = tg[~gone[tg] & (book.evicted_at[tg] == NEVER)]
    return remove, add, evict, n_adm, n_pre


class PreemptionLoop:
    """One engine (snapshot + pending set resident), the closed loop above. step(cycle) runs one scheduling cycle and applies it."""

    def __init__(self, eng, snap, pending, hold: int = 4, tgt_cap: Optional[int] = None, rsn_cap: int = 0, tick_ns: int = 1_000_000, uid_base: Optional[int] = None):
        self.eng, self.snap, self.pending, self.hold, self.tick = eng, snap, pending, int(hold), int(tick_ns)
        self.book = RowBook(snap)
        self.clock = int(getattr(snap, "now_ns", 0) or 0)
        u = snap.arrays["adm_uid_rank"]
        self.uid_base = int(uid_base if uid_base is not None else (int(u.max()) + 1 if len(u) else 0))
        mh, mps = eng.pending_bounds()
        self.tgt_cap = int(tgt_cap if tgt_cap is not None else max(4096, 4 * snap.n_adm))
        self.out = Decisions(pending.heads, tgt_cap=self.tgt_cap, rsn_cap=rsn_cap, n=mh, n_ps=mps)
        self.stats = []   # per cycle: heads, admitted, preempting heads, targets, rows removed (evicted / finished), rows resident

    def step(self, cycle: int):
        """-> (Decisions of the cycle, Heads arrays of its batch, head_wl) — or (None, None, None) when no ClusterQueue had a head."""
        eng = self.eng
        n, nps, head_wl = eng.pending_heads(cycle)
        if n == 0:
            eng.pending_apply()
            self.apply(cycle, None, None, None)
            return None, None, None
        eng.run_pending(self.out)
        eng.pending_apply()
        wl = head_wl[head_wl >= 0].astype(np.int64)
        hb = self.pending.heads.subset(wl, cycle)   # static columns are all the rows need (cq, priority, queue_ts, requests)
        assert hb.n == n and hb.n_ps == nps
        d = self.out.view(hb)
        d.n, d.n_ps = n, nps
        self.apply(cycle, d, hb.arrays, wl)
        return d, hb.arrays, head_wl

    def patch_of(self, cycle: int, d, ha, wl):
        return cycle_patch(self.book, self.snap, self.clock, self.uid_base, cycle, d, ha, wl)

    def apply(self, cycle: int, d, ha, wl):
        book = self.book
        remove, add, evict, n_adm, n_pre = self.patch_of(cycle, d, ha, wl)
        n_ev_gone = int(((book.evicted_at[remove] != NEVER)).sum()) if len(remove) else 0
        if len(remove) or add is not None or len(evict):
            book.evicted_at[evict] = cycle
            res = self.eng.patch_rows(remove, add, evict, fold_usage=True)
            new_index = res[1] if isinstance(res, tuple) else res
            if isinstance(res, tuple):
                assert res[0] == 0, res[0]
            add_cq = add["cq"] if add is not None else np.zeros(0, np.int32)
            book.place(remove, add_cq, np.full(len(add_cq), cycle + self.hold, np.int64), new_index)
        self.clock += self.tick
        self.stats.append(dict(cycle=cycle, heads=0 if d is None else d.n, admitted=n_adm, preempting=n_pre, targets=int(len(evict)),
                               removed_evicted=n_ev_gone, removed_finished=int(len(remove)) - n_ev_gone, rows=book.n))

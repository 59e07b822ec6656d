"""The ORACLE's half of the closed loop of a population with preemption (test infrastructure: tests/, bench.py's cpu_baseline leg and the
offline golden generator only). Its own queues (PendingOracle), its own snapshot image (rows + usage rebuilt in numpy / kqo_usage_apply), its
own row book and its own patch computed from ITS decisions — nothing of the engine's. The policy of the loop (what a cycle's decisions do to
the admitted table) is kueue_amd/closed_loop.py cycle_patch, shared by construction: it is the definition of the run being measured."""
import copy

import numpy as np

from kueue_amd import _ffi as F
from kueue_amd.closed_loop import RowBook, cycle_patch


def snapshot_patched(oracle, cfg, snap, remove, add, evict):
    """The oracle's image of kq_snapshot_patch_rows(KQ_ROWS_FOLD_USAGE): kept rows keep their order inside their ClusterQueue, added rows land
    behind them in the order given; removeUsage of what left, addUsage of what came (kqo_usage_apply: resource_node.go:144-165)."""
    a = snap.arrays
    nq, n_old = snap.n_cq, snap.n_adm
    old_cq = np.repeat(np.arange(nq), np.diff(a["cq_adm_off"]))
    keep = np.ones(n_old, bool); keep[remove] = False
    flags = a["adm_flags"].copy(); flags[evict] |= F.ADM_EVICTED
    uo = a["adm_use_off"]
    ent_row = np.repeat(np.arange(n_old), np.diff(uo))
    rm_ent = ~keep[ent_row]
    rm_triples = (old_cq[ent_row[rm_ent]].astype(np.int32), a["adm_use_fr"][rm_ent].astype(np.int32), a["adm_use_qty"][rm_ent].astype(np.int64))
    n_add = 0 if add is None else len(add["cq"])
    add_cq = add["cq"] if n_add else np.zeros(0, np.int32)
    rows_cq = np.concatenate([old_cq[keep], add_cq])
    perm = np.lexsort((np.arange(len(rows_cq)), np.concatenate([np.zeros(keep.sum(), np.int8), np.ones(n_add, np.int8)]), rows_cq))
    col = lambda old, new, dt: np.concatenate([old[keep], np.asarray(new, dt) if n_add else np.zeros(0, dt)])[perm].astype(dt)
    t = copy.copy(snap)
    b = dict(a)
    b["cq_adm_off"] = np.concatenate([[0], np.cumsum(np.bincount(rows_cq, minlength=nq))]).astype(np.int32)
    b["adm_priority"] = col(a["adm_priority"], add["priority"] if n_add else [], np.int64)
    b["adm_queue_ts"] = col(a["adm_queue_ts"], add["queue_ts"] if n_add else [], np.int64)
    b["adm_reserve_ts"] = col(a["adm_reserve_ts"], add["reserve_ts"] if n_add else [], np.int64)
    b["adm_uid_rank"] = col(a["adm_uid_rank"], add["uid_rank"] if n_add else [], np.uint32)
    b["adm_flags"] = col(flags, add["flags"] if n_add else [], np.uint8)
    # usage entries, row by row in the new order
    src = np.concatenate([np.nonzero(keep)[0], n_old + np.arange(n_add)])[perm]
    fr, qty, off = [], [], [0]
    auo = add["use_off"] if n_add else None
    for r in src:
        if r < n_old:
            fr.append(a["adm_use_fr"][uo[r]:uo[r + 1]]); qty.append(a["adm_use_qty"][uo[r]:uo[r + 1]])
        else:
            i = r - n_old
            fr.append(add["use_fr"][auo[i]:auo[i + 1]]); qty.append(add["use_qty"][auo[i]:auo[i + 1]])
        off.append(off[-1] + len(fr[-1]))
    b["adm_use_off"] = np.array(off, np.int32)
    b["adm_use_fr"] = np.concatenate(fr).astype(np.int32) if fr else np.zeros(0, np.int32)
    b["adm_use_qty"] = np.concatenate(qty).astype(np.int64) if qty else np.zeros(0, np.int64)
    t.arrays = b; t.n_adm = len(rows_cq); t._struct = None; t.admitted = None
    if len(rm_triples[0]):
        b["usage"] = oracle.usage_apply(cfg, t, rm_triples, add=False); t._struct = None
    if n_add:
        ent_add = np.repeat(np.arange(n_add), np.diff(add["use_off"]))
        b["usage"] = oracle.usage_apply(cfg, t, (add["cq"][ent_add].astype(np.int32), add["use_fr"].astype(np.int32), add["use_qty"].astype(np.int64)), add=True)
        t._struct = None
    freed_cq = np.unique(rm_triples[0])
    return t, freed_cq


class OracleLoop:
    """One oracle-side run of the loop."""

    def __init__(self, oracle, cfg, snap, pending, hold, uid_base, clock, tick):
        self.oracle, self.cfg, self.hold, self.uid_base, self.clock, self.tick = oracle, cfg, hold, uid_base, clock, tick
        self.q = oracle.PendingOracle(cfg, snap, pending)
        self.snap = copy.copy(snap); self.snap.arrays = dict(snap.arrays)
        self.book = RowBook(self.snap)
        parent = snap.arrays["parent"]; root_of = np.arange(snap.N)
        for _ in range(8):
            root_of = np.where(parent[root_of] >= 0, parent[root_of], root_of)
        self.root_of = root_of

    def step(self, c):
        """-> (Heads, head_wl, Decisions) of cycle c (Decisions None when no ClusterQueue had a head), the cycle applied."""
        q = self.q
        hb, ohw = q.heads(c)
        want = None
        if hb.n == 0:
            remove, add, evict, _, _ = cycle_patch(self.book, self.snap, self.clock, self.uid_base, c, None, None, None)
        else:
            want = self.oracle.cycle_run(self.cfg, self.snap, hb)
            q.apply(hb, want)
            wl = ohw[ohw >= 0].astype(np.int64)
            want.n, want.n_ps = hb.n, hb.n_ps
            remove, add, evict, _, _ = cycle_patch(self.book, self.snap, self.clock, self.uid_base, c, want, hb.arrays, wl)
        if len(remove) or add is not None or len(evict):
            self.book.evicted_at[evict] = c
            self.snap, freed = snapshot_patched(self.oracle, self.cfg, self.snap, remove, add, evict)
            add_cq = add["cq"] if add is not None else np.zeros(0, np.int32)
            self.book.place(remove, add_cq, np.full(len(add_cq), c + self.hold, np.int64))
            q.snap = self.snap
            if len(freed):   # QueueAssociatedInadmissibleWorkloadsAfter: every ClusterQueue under the root cohorts that got quota back
                q.queue_inadmissible(np.nonzero(np.isin(self.root_of[:self.snap.n_cq], np.unique(self.root_of[freed])))[0])
        self.clock += self.tick
        return hb, ohw, want

    def close(self):
        self.q.close()



"""Host-side mirror of the reference's scheduler entry points over the C ABI.

`Scheduler` follows pkg/scheduler/scheduler.go: `New(...)` binds configuration (scheduler.go:182,
options :143-180), `schedule(heads, snapshot)` is one cycle (:308) whose decision part runs on the
MI355X through kq_snapshot_put + kq_cycle_run.  There is no CPU implementation behind this class:
if the HIP library or a device is missing it raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _ffi as F
from .api import Decisions, Heads, Snapshot, make_config


def row_patch_struct(remove_rows, add, evict_rows=(), fold_usage=False):
    """kq_row_patch from plain arrays -> (struct, the arrays it points into)."""
    keep = []
    p = F.kq_row_patch()
    p.flags = F.ROWS_FOLD_USAGE if fold_usage else 0
    rm = np.ascontiguousarray(remove_rows, np.int32)
    keep.append(rm)
    p.n_remove = len(rm); p.remove_rows = F.ptr(rm) if len(rm) else None
    ev = np.ascontiguousarray(evict_rows, np.int32)
    keep.append(ev)
    p.n_evict = len(ev); p.evict_rows = F.ptr(ev) if len(ev) else None
    n_add = 0 if not add else len(add["cq"])
    p.n_add = n_add
    if n_add:
        for field, key, dt in (("add_cq", "cq", np.int32), ("add_priority", "priority", np.int64), ("add_queue_ts", "queue_ts", np.int64),
                               ("add_reserve_ts", "reserve_ts", np.int64), ("add_uid_rank", "uid_rank", np.uint32), ("add_flags", "flags", np.uint8),
                               ("add_use_off", "use_off", np.int32), ("add_use_fr", "use_fr", np.int32), ("add_use_qty", "use_qty", np.int64)):
            a = np.ascontiguousarray(add[key], dt)
            if a.size == 0:
                a = np.zeros(1, dt)
            keep.append(a)
            setattr(p, field, F.ptr(a))
    return p, keep


class EngineError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"kq_engine error {code} ({F.KQ_ERRORS.get(code, '?')}): {msg}")
        self.code = code


class Engine:
    """Thin RAII wrapper of kq_engine*."""

    def __init__(self, cfg: Optional[F.kq_config] = None):
        self._lib = F.load_engine()
        self.cfg = cfg if cfg is not None else make_config()
        self._h = C.c_void_p()
        rc = self._lib.kq_engine_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            raise EngineError(rc, self._lib.kq_strerror(rc).decode())
        self.snap: Optional[Snapshot] = None

    def close(self):
        if self._h:
            self._lib.kq_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(rc, self._lib.kq_last_error(self._h).decode())

    def put(self, snap: Snapshot):
        self._check(self._lib.kq_snapshot_put(self._h, C.byref(snap.struct())))
        self.snap = snap

    def patch(self, snap: Snapshot, what: int):
        """kq_snapshot_patch: the next cycle's snapshot when only usage (F.PATCH_USAGE) and / or the admitted set (F.PATCH_ADMITTED) moved."""
        self._check(self._lib.kq_snapshot_patch(self._h, C.byref(snap.struct()), what))
        self.snap = snap

    def patch_rows(self, remove_rows=(), add: Optional[dict] = None, evict_rows=(), fold_usage=False) -> np.ndarray:
        """kq_snapshot_patch_rows: rows that left / rows that came (add: dict of arrays cq, priority, queue_ts, reserve_ts, uid_rank, flags,
        use_off, use_fr, use_qty); the admitted table and everything derived from it are rebuilt on the device. -> new index of every old row.
        fold_usage (KQ_ROWS_FOLD_USAGE): the usage of the rows that left leaves the snapshot, the usage of the rows that came enters it."""
        p, keep = row_patch_struct(remove_rows, add, evict_rows, fold_usage)
        new_index = np.zeros(max(self._n_adm(), 1), np.int32)
        self._check(self._lib.kq_snapshot_patch_rows(self._h, C.byref(p), F.ptr(new_index)))
        return new_index

    def _n_adm(self) -> int:
        cap = C.c_int64(0)
        self._lib.kq_debug_read_rows(self._h, C.c_int32(0), None, C.byref(cap))
        return int(cap.value) // 4

    def run(self, heads: Heads, tgt_cap: Optional[int] = None, out: Optional[Decisions] = None, rsn_cap: int = 0) -> Decisions:
        d = out if out is not None else Decisions(heads, tgt_cap=tgt_cap, rsn_cap=rsn_cap)
        self._check(self._lib.kq_cycle_run(self._h, C.byref(heads.struct()), C.byref(d.struct())))
        ms, by = C.c_double(), C.c_int64()
        self._lib.kq_last_cycle_stats(self._h, C.byref(ms), C.byref(by))
        d.kernel_ms, d.bytes = ms.value, by.value
        return d

    def run_tas(self, heads: Heads, ct, tgt_cap: Optional[int] = None, rsn_cap: int = 0, dom_cap: Optional[int] = None):
        """kq_cycle_run_tas: one scheduling cycle with Topology-Aware Scheduling inside it (include/kq_cycle_tas.h; ct =
        kueue_amd.tas_cycle.CycleTAS) -> (Decisions, CycleTASOut); Decisions.tas_stats = {finds, recomputes, unsupported}."""
        from .tas_cycle import CycleTASOut
        d = Decisions(heads, tgt_cap=tgt_cap, rsn_cap=rsn_cap)
        out = CycleTASOut(ct, dom_cap=dom_cap)
        ts = np.zeros(4, np.int64)
        self._check(self._lib.kq_cycle_run_tas(self._h, C.byref(heads.struct()), C.byref(ct.struct()), C.byref(d.struct()), C.byref(out.struct()), F.ptr(ts)))
        ms, by = C.c_double(), C.c_int64()
        self._lib.kq_last_cycle_stats(self._h, C.byref(ms), C.byref(by))
        d.kernel_ms, d.bytes = ms.value, by.value
        d.tas_stats = dict(finds=int(ts[0]), recomputes=int(ts[1]), unsupported=bool(ts[2]), class_hits=int(ts[3]))
        return d, out

    def heads_put(self, heads: Heads, batch: int):
        """kq_heads_put: make a batch of heads resident in HBM under a small caller-chosen id."""
        self._check(self._lib.kq_heads_put(self._h, C.byref(heads.struct()), batch))

    def run_resident(self, batch: int, out: Decisions, check: bool = True) -> int:
        """kq_cycle_run_resident: one cycle over a resident batch (no host->device input traffic)."""
        rc = self._lib.kq_cycle_run_resident(self._h, batch, C.byref(out.struct()))
        if check:
            self._check(rc)
        return rc

    def nominate_resident(self, batch: int, out: Decisions) -> Decisions:
        """kq_nominate_run_resident: Scheduler.nominate for every head of a resident batch (no iterator, no processEntry)."""
        self._check(self._lib.kq_nominate_run_resident(self._h, batch, C.byref(out.struct())))
        return out

    # ---- pending side on the device (pkg/cache/queue) -------------------------------------------------------------
    def pending_put(self, pending):
        """kq_pending_put: PushOrUpdate of every pending workload."""
        self._check(self._lib.kq_pending_put(self._h, C.byref(pending.struct())))
        self.pending = pending

    def pending_heads(self, cycle: int, cq_active: Optional[np.ndarray] = None):
        """kq_pending_heads = queues.Heads(): -> (n_heads, n_podsets, head_wl[n_cq])."""
        n, nps = C.c_int32(), C.c_int32()
        hw = np.full(self.snap.n_cq, -1, np.int32)
        act = None if cq_active is None else F.ptr(np.ascontiguousarray(cq_active, np.uint8))
        self._check(self._lib.kq_pending_heads(self._h, cycle, act, C.byref(n), C.byref(nps), F.ptr(hw)))
        return n.value, nps.value, hw

    def run_pending(self, out: Decisions) -> Decisions:
        self._check(self._lib.kq_cycle_run_pending(self._h, C.byref(out.struct())))
        return out

    def pending_apply(self):
        self._check(self._lib.kq_pending_apply(self._h))

    def pending_add(self, more) -> int:
        """kq_pending_add: PushOrUpdate of workloads that were not pending before; returns the index of the first one."""
        first = C.c_int32()
        self._check(self._lib.kq_pending_add(self._h, C.byref(more.struct()), C.byref(first)))
        self.pending = self.pending.extended(more)
        return first.value

    def pending_update(self, wl, more) -> int:
        """kq_pending_update: PushOrUpdate of workloads that ARE pending with a new object; wl[i] is replaced by more[i], which gets
        the returned index + i."""
        a = np.ascontiguousarray(wl, np.int32)
        first = C.c_int32()
        self._check(self._lib.kq_pending_update(self._h, C.c_int32(len(a)), F.ptr(a), C.byref(more.struct()), C.byref(first)))
        self.pending = self.pending.extended(more)
        return first.value

    def pending_set_clock(self, now_ns: int):
        self._check(self._lib.kq_pending_set_clock(self._h, C.c_int64(int(now_ns))))

    def pending_set_requeue_at(self, wl, at):
        a = np.ascontiguousarray(wl, np.int32); b = np.ascontiguousarray(at, np.int64)
        if len(a):
            self._check(self._lib.kq_pending_set_requeue_at(self._h, C.c_int32(len(a)), F.ptr(a), F.ptr(b)))

    def pending_delete(self, wl):
        """kq_pending_delete: ClusterQueue.Delete of pending workloads."""
        a = np.ascontiguousarray(wl, np.int32)
        if len(a):
            self._check(self._lib.kq_pending_delete(self._h, len(a), F.ptr(a)))

    def pending_set_lq_usage(self, usage):
        """kq_pending_set_lq_usage: the LocalQueues' fair-sharing usage (afs.CalculateUsage, host-evaluated) for the next Heads()."""
        u = np.ascontiguousarray(usage, np.float64)
        self._check(self._lib.kq_pending_set_lq_usage(self._h, len(u), F.ptr(u)))

    # ---- AdmissionFairSharing ledger on the device (kq_pending_afs_*) ----
    def pending_bounds(self):
        """kq_pending_bounds: (max heads, max podsets) of a cycle over the resident pending set."""
        a, b = C.c_int32(), C.c_int32()
        self._check(self._lib.kq_pending_bounds(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pending_step(self, cycle, tgt_cap, release_age=0, want_heads=False, cq_active=None):
        """kq_pending_step: Heads() + cycle + commit + apply (+ release) enqueued, nothing waited for."""
        act = None if cq_active is None else F.ptr(np.ascontiguousarray(cq_active, np.uint8))
        self._check(self._lib.kq_pending_step(self._h, C.c_int64(cycle), act, C.c_int32(tgt_cap), C.c_int32(release_age), C.c_int32(1 if want_heads else 0)))

    def pending_step_reasons(self, rsn_cap: int):
        """kq_pending_step_reasons: the steps issued from now on record reason windows (rsn_cap > 0) or not (0)."""
        self._check(self._lib.kq_pending_step_reasons(self._h, C.c_int32(rsn_cap)))

    def pending_step_wait(self, out=None, want_heads=False):
        """kq_pending_step_wait for the oldest step in flight -> (n_heads, n_podsets, head_wl or None); `out` sized for pending_bounds()."""
        n, nps = C.c_int32(), C.c_int32()
        hw = np.full(self.snap.n_cq, -1, np.int32) if want_heads else None
        self._check(self._lib.kq_pending_step_wait(self._h, C.byref(out.struct()) if out is not None else None, C.byref(n), C.byref(nps),
                                      F.ptr(hw) if want_heads else None))
        return n.value, nps.value, hw

    def pending_afs_put(self, ledger, penalties):
        """kq_pending_afs_put: `ledger` = kueue_amd.afs.Ledger, `penalties` = entry_penalty of every pending workload."""
        self._afs = ledger
        self._check(self._lib.kq_pending_afs_put(self._h, C.byref(ledger.struct(penalties))))

    def pending_afs_wl_penalty(self, wl, penalties):
        a = np.ascontiguousarray(wl, np.int32)
        lo, hi, mask = self._afs.workload_columns(penalties)
        if len(a):
            self._check(self._lib.kq_pending_afs_wl_penalty(self._h, len(a), F.ptr(a), F.ptr(lo), F.ptr(hi), F.ptr(mask)))

    def pending_afs_sub_penalty(self, wl):
        a = np.ascontiguousarray(wl, np.int32)
        if len(a):
            self._check(self._lib.kq_pending_afs_sub_penalty(self._h, len(a), F.ptr(a)))

    def pending_afs_set_consumed(self, lq, rows, f64_rows=None, settle_wl=None):
        """rows[i] = entry.Resources of LocalQueue lq[i] as amounts per ledger resource (nano units)."""
        a = np.ascontiguousarray(lq, np.int32)
        lo, hi, f = self._afs.consumed_columns(rows, f64_rows)
        st = None if settle_wl is None else np.ascontiguousarray(settle_wl, np.int32)
        if len(a):
            self._check(self._lib.kq_pending_afs_set_consumed(self._h, len(a), F.ptr(a), F.ptr(lo), F.ptr(hi), None if f is None else F.ptr(f),
                                                              None if st is None else F.ptr(st)))

    def pending_afs_read(self, n_workloads):
        from kueue_amd.afs import join128
        L = self._afs
        cells = L.n_lq * L.n_res
        usage = np.zeros(L.n_lq, np.float64)
        plo, phi, pp = np.zeros(cells, np.uint64), np.zeros(cells, np.int64), np.zeros(cells, np.uint8)
        clo, chi = np.zeros(cells, np.uint64), np.zeros(cells, np.int64)
        rec = np.zeros(max(n_workloads, 1), np.uint8)
        self._check(self._lib.kq_pending_afs_read(self._h, F.ptr(usage), F.ptr(plo), F.ptr(phi), F.ptr(pp), F.ptr(clo), F.ptr(chi), F.ptr(rec)))
        pen = [join128(a, b) for a, b in zip(plo.tolist(), phi.tolist())]
        con = [join128(a, b) for a, b in zip(clo.tolist(), chi.tolist())]
        return dict(usage=usage, penalty=pen, present=pp, consumed=con, record=rec[:n_workloads])

    def pending_queue_inadmissible(self, cqs=None):
        if cqs is None:
            self._check(self._lib.kq_pending_queue_inadmissible(self._h, 0, None))
        else:
            a = np.ascontiguousarray(cqs, np.int32)
            self._check(self._lib.kq_pending_queue_inadmissible(self._h, len(a), F.ptr(a) if len(a) else None))

    def pending_state(self):
        st = np.zeros(max(self.pending.n, 1), np.uint8)
        counts = np.zeros(4, np.int32)
        self._check(self._lib.kq_pending_read_state(self._h, F.ptr(st), F.ptr(counts)))
        return st[:self.pending.n], counts

    # ---- one root tree split across ranks (kueue_amd/sharding.py) ---------------------------------------------------------
    # ---- sharded nominate, merged process (kq_engine.h) ----
    def shard_words(self, heads: Heads, out: Decisions, world: int) -> int:
        w = C.c_int64()
        self._check(self._lib.kq_cycle_shard_words(self._h, C.byref(heads.struct()), C.byref(out.struct()), C.c_int32(world), C.byref(w)))
        return w.value

    def nominate_shard(self, heads: Heads, mine: Optional[np.ndarray], world: int, rank: int, xbuf_ptr: int, out: Decisions):
        m = None if mine is None else np.ascontiguousarray(mine, np.uint8)
        self._check(self._lib.kq_cycle_nominate_shard(self._h, C.byref(heads.struct()), None if m is None else F.ptr(m), C.c_int32(world), C.c_int32(rank),
                                                      C.c_void_p(xbuf_ptr), C.byref(out.struct())))

    def process_merged(self, world: int, rank: int, xbuf_ptr: int, out: Decisions) -> Decisions:
        self._check(self._lib.kq_cycle_process_merged(self._h, C.c_int32(world), C.c_int32(rank), C.c_void_p(xbuf_ptr), C.byref(out.struct())))
        return out

    def spec_stats(self) -> np.ndarray:
        """kq_debug_spec_stats: [windows, rounds, entries decided, trees handed back, items, max rounds, abandoned, truncated] of the last cycle."""
        out = np.zeros(8, np.int64)
        self._check(self._lib.kq_debug_spec_stats(self._h, F.ptr(out)))
        return out

    def certificate(self, delta_dev_ptr: int):
        """kq_cycle_certificate: the cycle's usage delta into a device buffer; -> (root_margin [n_tree * n_fr], flags [n_tree])."""
        n_tree = int((self.snap.arrays["parent"] < 0).sum())
        margin = np.zeros(max(n_tree, 1) * self.snap.n_fr, np.int64)
        flags = np.zeros(max(n_tree, 1), np.int32)
        self._check(self._lib.kq_cycle_certificate(self._h, C.c_void_p(delta_dev_ptr), F.ptr(margin), F.ptr(flags)))
        return margin, flags

    def usage_add(self, delta_dev_ptr: int, sign: int):
        """kq_snapshot_usage_add: fold an (all-reduced) ClusterQueue-level usage delta into the resident snapshot."""
        self._check(self._lib.kq_snapshot_usage_add(self._h, C.c_void_p(delta_dev_ptr), sign))

    def try_commit(self) -> int:
        return self._lib.kq_cycle_commit(self._h, None)

    def commit(self) -> int:
        """kq_cycle_commit: fold the last cycle's admissions into the resident snapshot; returns how many."""
        n = C.c_int32()
        self._check(self._lib.kq_cycle_commit(self._h, C.byref(n)))
        return n.value

    def release(self, age: int = 1):
        """kq_cycle_release: the workloads committed `age` commits ago finish."""
        self._check(self._lib.kq_cycle_release(self._h, age))

    def read_usage(self) -> np.ndarray:
        n = self.snap.N * self.snap.n_fr
        us = np.zeros(n, np.int64)
        self._check(self._lib.kq_snapshot_read_planes(self._h, None, F.ptr(us), None))
        return us

    def derive(self):
        """kq_snapshot_derive on the uploaded snapshot; returns (subtree_quota, usage, quota_flags) read back."""
        self._check(self._lib.kq_snapshot_derive(self._h))
        n = self.snap.N * self.snap.n_fr
        sq, us, fl = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.uint8)
        self._check(self._lib.kq_snapshot_read_planes(self._h, F.ptr(sq), F.ptr(us), F.ptr(fl)))
        return sq, us, fl

    def usage_after(self) -> np.ndarray:
        """Snapshot usage as mutated by the last cycle (test hook)."""
        u = np.zeros(self.snap.N * self.snap.n_fr, np.int64)
        self._lib.kq_debug_read_usage_work.argtypes = [C.c_void_p, F.i64p]
        self._check(self._lib.kq_debug_read_usage_work(self._h, F.ptr(u)))
        return u


class Scheduler:
    """scheduler.New(queues, cache, client, recorder, opts...) -> *Scheduler (scheduler.go:182)."""

    def __init__(self, fair_sharing: bool = False, gates: Optional[int] = None, fs_strategies=(), device: int = 0):
        self.cfg = make_config(fair_sharing=fair_sharing, gates=gates, fs_strategies=fs_strategies, device=device)
        self.engine = Engine(self.cfg)
        self.scheduling_cycle = 0

    def schedule(self, heads_workloads, snapshot: Snapshot) -> Decisions:
        """One cycle (scheduler.go:308): steps 3-5 run on the device, side effects stay with the caller."""
        self.scheduling_cycle += 1
        self.engine.put(snapshot)
        heads = heads_workloads if isinstance(heads_workloads, Heads) else Heads(snapshot, heads_workloads, cycle=self.scheduling_cycle)
        return self.engine.run(heads)

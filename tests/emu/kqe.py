"""ctypes binding of the TEST-ONLY 1-lane emulation of the device logic (tests/emu/kq_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from kueue_amd import _ffi as F
from kueue_amd.api import Decisions

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("KQE_LIB", os.path.join(HERE, "libkq_emu.so"))   # (KQE_LIB: a sanitizer build of the same source)
_lib = None


def lib():
    global _lib
    if _lib is None:
        import fcntl
        with open(os.path.join(HERE, ".build.lock"), "w") as lk:      # xdist workers must not rebuild the library side by side
            fcntl.flock(lk, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", HERE, "-s"])
        _lib = C.CDLL(LIB)
        _lib.kqe_last_error.restype = C.c_char_p
    return _lib


class EmuEngine:
    def __init__(self, cfg):
        self.h = C.c_void_p()
        assert lib().kqe_engine_create(C.byref(cfg), C.byref(self.h)) == 0

    def commit(self):
        n = C.c_int32()
        rc = lib().kqe_cycle_commit(self.h, C.byref(n))
        assert rc == 0, (rc, lib().kqe_last_error(self.h))
        return n.value

    def release(self, age=1):
        rc = lib().kqe_cycle_release(self.h, age)
        assert rc == 0, (rc, lib().kqe_last_error(self.h))

    def read_usage(self):
        us = np.zeros(self.snap.N * self.snap.n_fr, np.int64)
        assert lib().kqe_read_planes(self.h, None, F.ptr(us), None) == 0
        return us

    def read_usage_work(self):
        """The cycle's private usage plane as processEntry left it."""
        u = np.zeros(self.snap.N * self.snap.n_fr, np.int64)
        lib().kqe_read_usage(self.h, F.ptr(u))
        return u

    def derive(self):
        assert lib().kqe_snapshot_derive(self.h) == 0
        n = self.snap.N * self.snap.n_fr
        sq, us, fl = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.uint8)
        assert lib().kqe_read_planes(self.h, F.ptr(sq), F.ptr(us), F.ptr(fl)) == 0
        return sq, us, fl

    def spec_stats(self):
        """[windows, rounds, entries decided, trees handed back, items, max rounds, abandoned windows, truncated windows] of the last cycle."""
        out = np.zeros(8, np.int64)
        assert lib().kqe_spec_stats(self.h, F.ptr(out)) == 0
        return out

    def spec_variant(self, v):
        """-1: rotate (default); 0: full windows; 1 / 2: tiny windows; 3: two rounds, then the serial kernel; 4: rounds off."""
        lib().kqe_spec_variant(self.h, int(v))

    def force_exact_drs(self, on=True):
        lib().kqe_force_exact_drs(self.h, 1 if on else 0)

    def close(self):
        if self.h:
            lib().kqe_engine_destroy(self.h)
            self.h = None

    def put(self, snap):
        rc = lib().kqe_snapshot_put(self.h, C.byref(snap.struct()))
        assert rc == 0, (rc, lib().kqe_last_error(self.h))
        self.snap = snap

    def patch(self, snap, what):
        rc = lib().kqe_snapshot_patch(self.h, C.byref(snap.struct()), C.c_uint32(what))
        assert rc == 0, (rc, lib().kqe_last_error(self.h))
        self.snap = snap

    def heads_put(self, heads, batch):
        rc = lib().kqe_heads_put(self.h, C.byref(heads.struct()), batch)
        assert rc == 0, (rc, lib().kqe_last_error(self.h))

    def run_resident(self, batch, out, check=True):
        rc = lib().kqe_cycle_run_resident(self.h, batch, C.byref(out.struct()))
        if check:
            assert rc == 0, (rc, lib().kqe_last_error(self.h))
        return rc

    def nominate_resident(self, batch, out):
        rc = lib().kqe_nominate_run_resident(self.h, batch, C.byref(out.struct()))
        assert rc == 0, (rc, lib().kqe_last_error(self.h))
        return out

    def try_commit(self):
        return lib().kqe_cycle_commit(self.h, None)

    def shard_words(self, heads, out, world):
        w = C.c_int64()
        self._ok(lib().kqe_cycle_shard_words(self.h, C.byref(heads.struct()), C.byref(out.struct()), C.c_int32(world), C.byref(w)))
        return w.value

    def nominate_shard(self, heads, mine, world, rank, xbuf_ptr, out):
        m = None if mine is None else np.ascontiguousarray(mine, np.uint8)
        self._ok(lib().kqe_cycle_nominate_shard(self.h, C.byref(heads.struct()), None if m is None else F.ptr(m), C.c_int32(world), C.c_int32(rank),
                                                C.c_void_p(xbuf_ptr), C.byref(out.struct())))

    def process_merged(self, world, rank, xbuf_ptr, out):
        self._ok(lib().kqe_cycle_process_merged(self.h, C.c_int32(world), C.c_int32(rank), C.c_void_p(xbuf_ptr), C.byref(out.struct())))
        return out

    def certificate(self, delta_ptr):
        n_tree = int((self.snap.arrays["parent"] < 0).sum())
        margin = np.zeros(max(n_tree, 1) * self.snap.n_fr, np.int64)
        flags = np.zeros(max(n_tree, 1), np.int32)
        self._ok(lib().kqe_cycle_certificate(self.h, C.c_void_p(delta_ptr), F.ptr(margin), F.ptr(flags)))
        return margin, flags

    def usage_add(self, delta_ptr, sign):
        self._ok(lib().kqe_snapshot_usage_add(self.h, C.c_void_p(delta_ptr), C.c_int32(sign)))

    def _ok(self, rc):
        assert rc == 0, (rc, lib().kqe_last_error(self.h))

    def pending_put(self, pending):
        self._ok(lib().kqe_pending_put(self.h, C.byref(pending.struct())))
        self.pending = pending

    def pending_heads(self, cycle, cq_active=None):
        n, nps = C.c_int32(), C.c_int32()
        hw = np.full(self.snap.n_cq, -1, np.int32)
        act = None if cq_active is None else F.ptr(np.ascontiguousarray(cq_active, np.uint8))
        self._ok(lib().kqe_pending_heads(self.h, C.c_int64(cycle), act, C.byref(n), C.byref(nps), F.ptr(hw)))
        return n.value, nps.value, hw

    def run_pending(self, out):
        self._ok(lib().kqe_cycle_run_pending(self.h, C.byref(out.struct())))
        return out

    def pending_apply(self):
        self._ok(lib().kqe_pending_apply(self.h))

    def pending_add(self, more) -> int:
        first = C.c_int32()
        self._ok(lib().kqe_pending_add(self.h, C.byref(more.struct()), C.byref(first)))
        self.pending = self.pending.extended(more)
        return first.value

    def pending_batch_flags(self, n):
        out = np.zeros(max(n, 1), np.uint32)
        self._ok(lib().kqe_pending_head_flags(self.h, F.ptr(out), C.c_int32(n)))
        return out[:n]

    def pending_update(self, wl, more) -> int:
        a = np.ascontiguousarray(wl, np.int32)
        first = C.c_int32()
        self._ok(lib().kqe_pending_update(self.h, C.c_int32(len(a)), F.ptr(a), C.byref(more.struct()), C.byref(first)))
        self.pending = self.pending.extended(more)
        return first.value

    def pending_set_clock(self, now_ns: int):
        self._ok(lib().kqe_pending_set_clock(self.h, C.c_int64(int(now_ns))))

    def pending_set_requeue_at(self, wl, at):
        a = np.ascontiguousarray(wl, np.int32); b = np.ascontiguousarray(at, np.int64)
        if len(a):
            self._ok(lib().kqe_pending_set_requeue_at(self.h, C.c_int32(len(a)), F.ptr(a), F.ptr(b)))

    def pending_delete(self, wl):
        a = np.ascontiguousarray(wl, np.int32)
        if len(a):
            self._ok(lib().kqe_pending_delete(self.h, C.c_int32(len(a)), F.ptr(a)))

    def pending_set_lq_usage(self, usage):
        u = np.ascontiguousarray(usage, np.float64)
        self._ok(lib().kqe_pending_set_lq_usage(self.h, C.c_int32(len(u)), F.ptr(u)))

    def pending_bounds(self):
        """kq_pending_bounds: (max heads, max podsets) of a cycle over the resident pending set."""
        a, b = C.c_int32(), C.c_int32()
        self._ok(lib().kqe_pending_bounds(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pending_step(self, cycle, tgt_cap, release_age=0, want_heads=False, cq_active=None):
        """kq_pending_step: Heads() + cycle + commit + apply (+ release) enqueued, nothing waited for."""
        act = None if cq_active is None else F.ptr(np.ascontiguousarray(cq_active, np.uint8))
        self._ok(lib().kqe_pending_step(self.h, C.c_int64(cycle), act, C.c_int32(tgt_cap), C.c_int32(release_age), C.c_int32(1 if want_heads else 0)))

    def pending_step_reasons(self, rsn_cap):
        self._ok(lib().kqe_pending_step_reasons(self.h, C.c_int32(rsn_cap)))

    def pending_step_wait(self, out=None, want_heads=False):
        """kq_pending_step_wait for the oldest step in flight -> (n_heads, n_podsets, head_wl or None); `out` sized for pending_bounds()."""
        n, nps = C.c_int32(), C.c_int32()
        hw = np.full(self.snap.n_cq, -1, np.int32) if want_heads else None
        self._ok(lib().kqe_pending_step_wait(self.h, C.byref(out.struct()) if out is not None else None, C.byref(n), C.byref(nps),
                                      F.ptr(hw) if want_heads else None))
        return n.value, nps.value, hw

    def pending_afs_put(self, ledger, penalties):
        self._afs = ledger
        self._ok(lib().kqe_pending_afs_put(self.h, C.byref(ledger.struct(penalties))))

    def pending_afs_wl_penalty(self, wl, penalties):
        a = np.ascontiguousarray(wl, np.int32)
        lo, hi, mask = self._afs.workload_columns(penalties)
        if len(a):
            self._ok(lib().kqe_pending_afs_wl_penalty(self.h, C.c_int32(len(a)), F.ptr(a), F.ptr(lo), F.ptr(hi), F.ptr(mask)))

    def pending_afs_sub_penalty(self, wl):
        a = np.ascontiguousarray(wl, np.int32)
        if len(a):
            self._ok(lib().kqe_pending_afs_sub_penalty(self.h, C.c_int32(len(a)), F.ptr(a)))

    def pending_afs_set_consumed(self, lq, rows, f64_rows=None, settle_wl=None):
        a = np.ascontiguousarray(lq, np.int32)
        lo, hi, f = self._afs.consumed_columns(rows, f64_rows)
        st = None if settle_wl is None else np.ascontiguousarray(settle_wl, np.int32)
        if len(a):
            self._ok(lib().kqe_pending_afs_set_consumed(self.h, C.c_int32(len(a)), F.ptr(a), F.ptr(lo), F.ptr(hi), None if f is None else F.ptr(f),
                                                        None if st is None else F.ptr(st)))

    def pending_afs_read(self, n_workloads):
        from kueue_amd.afs import join128
        L = self._afs
        cells = L.n_lq * L.n_res
        usage = np.zeros(L.n_lq, np.float64)
        plo, phi, pp = np.zeros(cells, np.uint64), np.zeros(cells, np.int64), np.zeros(cells, np.uint8)
        clo, chi = np.zeros(cells, np.uint64), np.zeros(cells, np.int64)
        rec = np.zeros(max(n_workloads, 1), np.uint8)
        self._ok(lib().kqe_pending_afs_read(self.h, F.ptr(usage), F.ptr(plo), F.ptr(phi), F.ptr(pp), F.ptr(clo), F.ptr(chi), F.ptr(rec)))
        pen = [join128(a, b) for a, b in zip(plo.tolist(), phi.tolist())]
        con = [join128(a, b) for a, b in zip(clo.tolist(), chi.tolist())]
        return dict(usage=usage, penalty=pen, present=pp, consumed=con, record=rec[:n_workloads])

    def pending_queue_inadmissible(self, cqs=None):
        if cqs is None:
            self._ok(lib().kqe_pending_queue_inadmissible(self.h, 0, None))
        else:
            a = np.ascontiguousarray(cqs, np.int32)
            self._ok(lib().kqe_pending_queue_inadmissible(self.h, len(a), F.ptr(a) if len(a) else None))

    def pending_state(self):
        st = np.zeros(max(self.pending.n, 1), np.uint8)
        counts = np.zeros(4, np.int32)
        self._ok(lib().kqe_pending_read_state(self.h, F.ptr(st), F.ptr(counts)))
        return st[:self.pending.n], counts

    def run(self, heads, want_usage=False, tgt_cap=None, rsn_cap=0):
        d = Decisions(heads, tgt_cap=tgt_cap, rsn_cap=rsn_cap)
        rc = lib().kqe_cycle_run(self.h, C.byref(heads.struct()), C.byref(d.struct()))
        d.rc = rc
        d.error = lib().kqe_last_error(self.h).decode()
        if want_usage and rc == 0:
            u = np.zeros(self.snap.N * self.snap.n_fr, np.int64)
            lib().kqe_read_usage(self.h, F.ptr(u))
            d.usage_after = u
        b = C.c_int64()
        lib().kqe_last_bytes(self.h, C.byref(b))
        d.bytes = b.value
        pb = np.zeros(2, np.int64)
        lib().kqe_phase_bytes(self.h, F.ptr(pb))
        d.phase_bytes = pb.tolist()
        return d


def _run_tas(self, heads, ct, tgt_cap=None, want_usage=False, dom_cap=None, rsn_cap=0):
    """kqe_cycle_run_tas: the cycle with TAS inside it on the emulated engine (include/kq_cycle_tas.h)."""
    from kueue_amd.tas_cycle import CycleTASOut
    d = Decisions(heads, tgt_cap=tgt_cap, rsn_cap=rsn_cap)
    out = CycleTASOut(ct, dom_cap=dom_cap)
    ts = np.zeros(4, np.int64)
    rc = lib().kqe_cycle_run_tas(self.h, C.byref(heads.struct()), C.byref(ct.struct()), C.byref(d.struct()), C.byref(out.struct()), F.ptr(ts))
    d.rc = rc
    d.error = lib().kqe_last_error(self.h).decode()
    if want_usage and rc == 0:
        u = np.zeros(self.snap.N * self.snap.n_fr, np.int64)
        lib().kqe_read_usage(self.h, F.ptr(u))
        d.usage_after = u
    b = C.c_int64()
    lib().kqe_last_bytes(self.h, C.byref(b))
    d.bytes = b.value
    d.tas_stats = dict(finds=int(ts[0]), recomputes=int(ts[1]), unsupported=bool(ts[2]), class_hits=int(ts[3]))
    return d, out


EmuEngine.run_tas = _run_tas


def _patch_rows(self, remove_rows=(), add=None, evict_rows=(), fold_usage=False):
    """kqe_snapshot_patch_rows on the emulated engine -> (rc, new index of every old row)."""
    from kueue_amd.engine import row_patch_struct
    p, keep = row_patch_struct(remove_rows, add, evict_rows, fold_usage)
    cap = C.c_int64(0)
    lib().kqe_debug_read_rows(self.h, C.c_int32(0), None, C.byref(cap))
    new_index = np.zeros(max(int(cap.value) // 4, 1), np.int32)
    rc = lib().kqe_snapshot_patch_rows(self.h, C.byref(p), F.ptr(new_index))
    return rc, new_index


EmuEngine.patch_rows = _patch_rows


class EmuTas:
    """1-lane emulation of the TAS device code (kq_tas_device.hpp), same interface as kueue_amd.tas.TASEngine."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        assert lib().kqe_tas_create(C.byref(self.h)) == 0
        lib().kqe_tas_last_error.restype = C.c_char_p
        lib().kqe_tas_last_bytes.restype = C.c_int64
        self.topo = None

    def put(self, topo):
        rc = lib().kqe_tas_topology_put(self.h, C.byref(topo.struct()))
        assert rc == 0, (rc, lib().kqe_tas_last_error(self.h))
        self.topo = topo

    def find(self, rq, dom_cap=None):
        from kueue_amd import tas as T
        out = T.Result(rq, dom_cap)
        rc = lib().kqe_tas_find(self.h, C.byref(rq.struct()), C.byref(out.struct()))
        assert rc == 0, (rc, lib().kqe_tas_last_error(self.h))
        out.bytes = int(lib().kqe_tas_last_bytes(self.h))
        return out

    def find_elastic(self, rq, dom_cap=None, check=True):
        from kueue_amd import tas as T
        if rq.previous is None:
            return self.find(rq, dom_cap)
        out = T.Result(rq, dom_cap)
        rc = lib().kqe_tas_find_elastic(self.h, C.byref(rq.struct()), C.byref(rq.previous_struct()), C.byref(out.struct()))
        if not check:
            return rc if rc != 0 else out
        assert rc == 0, (rc, lib().kqe_tas_last_error(self.h))
        return out

    def find_replacement(self, rq, dom_cap=None):
        from kueue_amd import tas as T
        if rq.replacement is None:
            return self.find(rq, dom_cap)
        out = T.Result(rq, dom_cap)
        rc = lib().kqe_tas_find_replacement(self.h, C.byref(rq.struct()), C.byref(rq.replacement_struct()), C.byref(out.struct()))
        assert rc == 0, (rc, lib().kqe_tas_last_error(self.h))
        return out

    def exclusion_stats(self, rq, res, podsets=None):
        from kueue_amd import tas as T
        if podsets is None:
            podsets = [i for i in range(rq.n) if int(res.a["status"][i]) in (T.TAS_NOT_FIT, T.TAS_NOT_FIT_LAYERS)]
        if len(podsets) == 0:
            return res
        R = len(self.topo.resources)
        ps = np.asarray(podsets, np.int32); td = np.zeros(len(ps), np.int32); rs = np.zeros(len(ps) * R, np.int32)
        x = rq.replacement_struct()
        rc = lib().kqe_tas_exclusion_stats(self.h, C.byref(rq.struct()), C.byref(x) if x is not None else None, C.byref(res.struct()), len(ps), F.ptr(ps), None,
                                           F.ptr(td), F.ptr(rs))
        assert rc == 0, (rc, lib().kqe_tas_last_error(self.h))
        for k, i in enumerate(ps):
            res.exclusions[int(i)] = (self.topo.n_leaves, int(td[k]), {self.topo.resources[r]: int(rs[k * R + r]) for r in range(R) if rs[k * R + r]})
        return res

    def usage_apply(self, assignment, single_pod_requests, add=True):
        leaf = np.array([a for a, _ in assignment], np.int32); cnt = np.array([c for _, c in assignment], np.int32)
        req = np.ascontiguousarray(single_pod_requests, np.int64)
        assert lib().kqe_tas_usage_apply(self.h, len(leaf), F.ptr(leaf), F.ptr(cnt), F.ptr(req), 1 if add else 0) == 0

    def fits(self, assignment, single_pod_requests):
        leaf = np.array([a for a, _ in assignment], np.int32); cnt = np.array([c for _, c in assignment], np.int32)
        req = np.ascontiguousarray(single_pod_requests, np.int64)
        out = np.zeros(1, np.int32)
        assert lib().kqe_tas_fits(self.h, len(leaf), F.ptr(leaf), F.ptr(cnt), F.ptr(req), F.ptr(out)) == 0
        return bool(out[0])

    def read_usage(self):
        u = np.zeros(self.topo.n_leaves * len(self.topo.resources), np.int64)
        assert lib().kqe_tas_read_usage(self.h, F.ptr(u)) == 0
        return u

    def admit(self, rq, res, order=None):
        nw = len(rq.arrays["wl_off"]) - 1
        adm = np.zeros(max(nw, 1), np.uint8); na = np.zeros(1, np.int32)
        o = None if order is None else np.ascontiguousarray(order, np.int32)
        rc = lib().kqe_tas_admit(self.h, C.byref(rq.struct()), C.byref(res.struct()), F.ptr(o) if o is not None else None,
                                 C.c_int32(0 if o is None else len(o)), F.ptr(adm), F.ptr(na))
        assert rc == 0, (rc, lib().kqe_tas_last_error(self.h))
        assert int(na[0]) == int(adm[:nw].sum())
        return adm[:nw]

    def usage_delta(self, rq, res, plane_ptr, wl_sel=None):
        sel = None if wl_sel is None else np.ascontiguousarray(wl_sel, np.uint8)
        rc = lib().kqe_tas_usage_delta(self.h, C.byref(rq.struct()), C.byref(res.struct()), F.ptr(sel) if sel is not None else None, C.c_void_p(plane_ptr))
        assert rc == 0, (rc, lib().kqe_tas_last_error(self.h))

    def usage_add(self, plane_ptr, sign=1):
        assert lib().kqe_tas_usage_add(self.h, C.c_void_p(plane_ptr), C.c_int32(sign)) == 0

    def overflow(self, plane_ptr):
        over = np.zeros(self.topo.n_leaves, np.uint8); n = np.zeros(1, np.int32)
        assert lib().kqe_tas_overflow(self.h, C.c_void_p(plane_ptr) if plane_ptr else None, F.ptr(over), F.ptr(n)) == 0
        assert int(n[0]) == int(over.sum())
        return over

    def close(self):
        if self.h:
            lib().kqe_tas_destroy(self.h)
            self.h = None


class EmuGroup:
    """include/kq_group.h over emulated engines (kq_emu.cpp kqe_group_*): the device build's driver (kq_group_core.hpp) with the host collective."""
    FORCE_SHARDED = 2

    def __init__(self, cfg, n, flags=0):
        self.h = C.c_void_p()
        lib().kqe_group_last_error.restype = C.c_char_p
        rc = lib().kqe_group_create(C.byref(cfg), C.c_int32(n), C.c_uint32(flags), C.byref(self.h))
        assert rc == 0, rc
        self.n = n

    def put(self, snap):
        rc = lib().kqe_group_snapshot_put(self.h, C.byref(snap.struct()))
        assert rc == 0, (rc, lib().kqe_group_last_error(self.h))
        self.snap = snap

    def run(self, heads, tgt_cap=None, rsn_cap=0, check=True):
        d = Decisions(heads, tgt_cap=tgt_cap, rsn_cap=rsn_cap)
        rc = lib().kqe_group_cycle_run(self.h, C.byref(heads.struct()), C.byref(d.struct()))
        if check:
            assert rc == 0, (rc, lib().kqe_group_last_error(self.h))
            return d
        return rc

    def commit(self):
        n = C.c_int32()
        rc = lib().kqe_group_cycle_commit(self.h, C.byref(n))
        assert rc == 0, (rc, lib().kqe_group_last_error(self.h))
        return n.value

    def release(self, age=1):
        assert lib().kqe_group_cycle_release(self.h, C.c_int32(age)) == 0

    def usage(self, rank=0):
        u = np.zeros(self.snap.N * self.snap.n_fr, np.int64)
        assert lib().kqe_group_read_usage(self.h, C.c_int32(rank), F.ptr(u)) == 0
        return u

    def inject(self, rank, step):
        lib().kqe_group_inject(self.h, C.c_int32(rank), C.c_int32(step))

    def last_error(self):
        return lib().kqe_group_last_error(self.h).decode()

    def close(self):
        if self.h:
            lib().kqe_group_destroy(self.h)
            self.h = None

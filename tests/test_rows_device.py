"""The admitted-row candidate structures built on the device (kq_rows.hpp) against the host's build_prep (kq_prep.hpp):
every array byte for byte — candidate rank order per tree, ascending order, rank positions, the flavor-resource buckets (positions, rows,
records, fingerprints), the three level orders, row records, per-ClusterQueue record bytes, the per-tree flags — on random snapshots and on
BASELINE-shaped populations; then the same cycles give the same decisions on the rebuilt structures.
CPU suite: the 1-lane emulation (std::stable_sort as the sort primitive). GPU suite: the HIP engine (rocPRIM radix sort)."""
import ctypes as C

import numpy as np
import pytest

from tests.randgen import random_case

N_STRUCT = 31
NAMES = ["adm_cq", "tree_row_off", "tree_rows", "tree_rows_asc", "rank_pos", "frb_off", "frb", "frbr", "cq_row_bytes", "adm_rec", "frec", "frl0", "frl1", "frl2",
         "frb_sig", "cs_ok", "rec_ok", "cq_adm_off", "adm_use_off", "adm_use_fr", "adm_use_qty", "adm_prio", "adm_qts", "adm_rts", "adm_uid", "adm_flags",
         "fs_ok", "fs_posoff", "fs_scan", "fs_apply", "adm_recx"]


class _Emu:
    def __init__(self, cfg):
        from tests.emu import kqe
        self.e = kqe.EmuEngine(cfg)
        self.lib, self.h = kqe.lib(), self.e.h
        self.read_fn, self.rebuild_fn = self.lib.kqe_debug_read_rows, self.lib.kqe_debug_rows_rebuild

    def run(self, heads, **kw):
        d = self.e.run(heads, **kw)
        assert d.rc == 0, d.error
        return d


class _Hip:
    def __init__(self, cfg):
        from kueue_amd.engine import Engine
        self.e = Engine(cfg)
        self.lib, self.h = self.e._lib, self.e._h
        self.read_fn, self.rebuild_fn = self.lib.kq_debug_read_rows, self.lib.kq_debug_rows_rebuild

    def run(self, heads, **kw):
        return self.e.run(heads, **kw)


def read_all(x):
    out = []
    for w in range(N_STRUCT):
        cap = C.c_int64(0)
        x.read_fn(x.h, C.c_int32(w), None, C.byref(cap))           # KQ_ECAPACITY with the size
        buf = np.zeros(max(int(cap.value), 1), np.uint8)
        cap2 = C.c_int64(buf.size)
        rc = x.read_fn(x.h, C.c_int32(w), buf.ctypes.data_as(C.c_void_p), C.byref(cap2))
        assert rc == 0, (w, rc)
        out.append(buf[:int(cap2.value)].copy())
    return out


def check(make, cfg, snap, heads=None):
    import os
    os.environ["KQ_ROWS_HOST"] = "1"          # the reference: every structure built by build_prep on the host
    try:
        h = make(cfg)
    finally:
        del os.environ["KQ_ROWS_HOST"]
    h.e.put(snap)
    host = read_all(h)
    h.e.close()
    x = make(cfg)
    x.e.put(snap)                              # the device build (kq_rows.hpp)
    assert [a.tobytes() for a in read_all(x)] == [a.tobytes() for a in host], "kq_snapshot_put: device-built structures differ from build_prep's"
    want = x.run(heads, tgt_cap=max(16, snap.n_adm)) if heads is not None else None
    rc = x.rebuild_fn(x.h)
    assert rc == 0, rc
    dev = read_all(x)
    for name, a, b in zip(NAMES, host, dev):
        m = min(a.size, b.size)
        assert a.size == b.size and np.array_equal(a, b), (name, a.size, b.size, np.flatnonzero(a[:m] != b[:m])[:8])
    if heads is not None:
        got = x.run(heads, tgt_cap=max(16, snap.n_adm))
        assert not want.equal(got)
    x.e.close()


@pytest.mark.parametrize("seed", range(150))
def test_rows_rebuild_random_emulated(oracle, seed):
    cfg, snap, heads = random_case(seed, fair=seed % 3 == 0, preemption=True, tight=seed % 2 == 0)
    oracle.derive(snap)
    check(_Emu, cfg, snap, heads)


@pytest.mark.parametrize("cfgn", [1, 2])
def test_rows_rebuild_population_emulated(cfgn):
    from kueue_amd.api import make_config
    from kueue_amd.population import generate
    pop = generate(cfgn)
    check(_Emu, make_config(), pop.snapshot)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(3))
def test_rows_rebuild_random_gpu(oracle, block):
    for seed in range(block * 50, block * 50 + 50):
        cfg, snap, heads = random_case(seed, fair=seed % 3 == 0, preemption=True, tight=seed % 2 == 0)
        oracle.derive(snap)
        check(_Hip, cfg, snap, heads)


@pytest.mark.gpu
@pytest.mark.parametrize("cfgn", [2, 3, 4])
def test_rows_rebuild_population_gpu(cfgn):
    from kueue_amd.api import make_config
    from kueue_amd.population import generate
    pop = generate(cfgn)
    check(_Hip, make_config(), pop.snapshot, pop.heads_for_cycle(0) if cfgn != 4 else None)   # (cfg 4: the structures only; its cycle needs a 400 k-target pool)


# ---- kq_snapshot_patch_rows: rows that left / rows that came == kq_snapshot_put of the edited table -------------------------------------
def _patch_case(snap, rnd):
    """base / target row sets of a full snapshot -> (base snapshot, remove list in base numbering, add dict, expected snapshot)."""
    n = snap.n_adm
    a = snap.arrays
    cq_of = np.repeat(np.arange(snap.n_cq), np.diff(a["cq_adm_off"]))
    base = np.array(sorted(rnd.sample(range(n), rnd.randint(0, n))), np.int64) if n else np.zeros(0, np.int64)
    target = np.array(sorted(rnd.sample(range(n), rnd.randint(0, n))), np.int64) if n else np.zeros(0, np.int64)
    bset, tset = set(base.tolist()), set(target.tolist())
    remove = [i for i, r in enumerate(base.tolist()) if r not in tset]            # indices in the base table
    added = [r for r in target.tolist() if r not in bset]
    rnd.shuffle(added)                                                            # any order: they land per ClusterQueue in this order
    kept = [r for r in base.tolist() if r in tset]
    perm = []
    for c in range(snap.n_cq):
        perm += [r for r in kept if cq_of[r] == c] + [r for r in added if cq_of[r] == c]
    u0, u1 = a["adm_use_off"][np.array(added, np.int64)] if added else np.zeros(0, np.int64), a["adm_use_off"][np.array(added, np.int64) + 1] if added else np.zeros(0, np.int64)
    idx = np.concatenate([np.arange(x, y) for x, y in zip(u0, u1)]).astype(np.int64) if added else np.zeros(0, np.int64)
    ad = np.array(added, np.int64)
    add = dict(cq=cq_of[ad] if added else [], priority=a["adm_priority"][ad] if added else [], queue_ts=a["adm_queue_ts"][ad] if added else [],
               reserve_ts=a["adm_reserve_ts"][ad] if added else [], uid_rank=a["adm_uid_rank"][ad] if added else [], flags=a["adm_flags"][ad] if added else [],
               use_off=np.concatenate([[0], np.cumsum(u1 - u0)]) if added else [0], use_fr=a["adm_use_fr"][idx], use_qty=a["adm_use_qty"][idx])
    return snap.with_rows(base), remove, add, snap.with_rows(np.array(perm, np.int64))


def check_patch(make, cfg, snap, heads, rnd):
    base, remove, add, expected = _patch_case(snap, rnd)
    x = make(cfg)
    x.e.put(expected)
    want_rows = read_all(x)
    want = x.run(heads, tgt_cap=max(16, expected.n_adm)) if heads is not None else None
    x.e.close()
    y = make(cfg)
    y.e.put(base)
    r = y.e.patch_rows(remove, add)
    if isinstance(r, tuple):
        assert r[0] == 0, r[0]
    got_rows = read_all(y)
    for name, p, q in zip(NAMES, want_rows, got_rows):
        m = min(p.size, q.size)
        assert p.size == q.size and np.array_equal(p, q), (name, p.size, q.size, np.flatnonzero(p[:m] != q[:m])[:8])
    if heads is not None:
        y.e.snap = expected   # (the Python wrappers size their outputs from the snapshot object)
        got = y.run(heads, tgt_cap=max(16, expected.n_adm))
        assert not want.equal(got)
    y.e.close()


@pytest.mark.parametrize("seed", range(150))
def test_patch_rows_random_emulated(oracle, seed):
    import random
    cfg, snap, heads = random_case(seed, fair=seed % 3 == 0, preemption=True, tight=seed % 2 == 0)
    oracle.derive(snap)
    check_patch(_Emu, cfg, snap, heads, random.Random(seed * 17 + 3))


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(3))
def test_patch_rows_random_gpu(oracle, block):
    import random
    for seed in range(block * 40, block * 40 + 40):
        cfg, snap, heads = random_case(seed, fair=seed % 3 == 0, preemption=True, tight=seed % 2 == 0)
        oracle.derive(snap)
        check_patch(_Hip, cfg, snap, heads, random.Random(seed * 17 + 3))


def _sequence(make, cfg, snap, rnd, steps=4):
    """Several kq_snapshot_patch_rows in a row on one engine (the row table alternates between its two buffers, the structures are
    rebuilt in place): after every step the resident structures equal a fresh put of the same table, and new_index maps old to new."""
    n, a = snap.n_adm, snap.arrays
    cq_of = np.repeat(np.arange(snap.n_cq), np.diff(a["cq_adm_off"]))
    cur = sorted(rnd.sample(range(n), rnd.randint(0, n))) if n else []
    y = make(cfg)
    y.e.put(snap.with_rows(np.array(cur, np.int64)))
    for _ in range(steps):
        tset = set(rnd.sample(range(n), rnd.randint(0, n))) if n else set()
        remove = [i for i, r in enumerate(cur) if r not in tset]
        added = [r for r in sorted(tset) if r not in set(cur)]
        rnd.shuffle(added)
        kept = [r for r in cur if r in tset]
        perm = []
        for c in range(snap.n_cq):
            perm += [r for r in kept if cq_of[r] == c] + [r for r in added if cq_of[r] == c]
        ad = np.array(added, np.int64)
        u0 = a["adm_use_off"][ad] if added else np.zeros(0, np.int64)
        u1 = a["adm_use_off"][ad + 1] if added else np.zeros(0, np.int64)
        idx = np.concatenate([np.arange(p, q) for p, q in zip(u0, u1)]).astype(np.int64) if added else np.zeros(0, np.int64)
        add = dict(cq=cq_of[ad] if added else [], priority=a["adm_priority"][ad] if added else [], queue_ts=a["adm_queue_ts"][ad] if added else [],
                   reserve_ts=a["adm_reserve_ts"][ad] if added else [], uid_rank=a["adm_uid_rank"][ad] if added else [], flags=a["adm_flags"][ad] if added else [],
                   use_off=np.concatenate([[0], np.cumsum(u1 - u0)]) if added else [0], use_fr=a["adm_use_fr"][idx], use_qty=a["adm_use_qty"][idx])
        r = y.e.patch_rows(remove, add)
        new_index = r[1] if isinstance(r, tuple) else r
        if isinstance(r, tuple):
            assert r[0] == 0, r[0]
        pos = {row: i for i, row in enumerate(perm)}
        assert [int(new_index[i]) for i in range(len(cur))] == [pos.get(row, -1) for row in cur]
        x = make(cfg)
        x.e.put(snap.with_rows(np.array(perm, np.int64)))
        for name, p, q in zip(NAMES, read_all(x), read_all(y)):
            m = min(p.size, q.size)
            assert p.size == q.size and np.array_equal(p, q), (name, p.size, q.size, np.flatnonzero(p[:m] != q[:m])[:8])
        x.e.close()
        cur = perm
    y.e.close()


@pytest.mark.parametrize("seed", range(60))
def test_patch_rows_sequence_emulated(oracle, seed):
    import random
    cfg, snap, _ = random_case(seed, fair=seed % 3 == 0, preemption=True, tight=seed % 2 == 0)
    oracle.derive(snap)
    _sequence(_Emu, cfg, snap, random.Random(seed * 29 + 5))


@pytest.mark.gpu
def test_patch_rows_sequence_gpu(oracle):
    import random
    for seed in range(40):
        cfg, snap, _ = random_case(seed, fair=seed % 3 == 0, preemption=True, tight=seed % 2 == 0)
        oracle.derive(snap)
        _sequence(_Hip, cfg, snap, random.Random(seed * 29 + 5))
    from kueue_amd.api import make_config
    from kueue_amd.population import generate
    _sequence(_Hip, make_config(), generate(3).snapshot, random.Random(7), steps=3)


@pytest.mark.parametrize("cfgn", [1, 2])
def test_rows_rebuild_fair_population_emulated(cfgn):
    from kueue_amd.api import make_config
    from kueue_amd.population import generate
    pop = generate(cfgn, fair_sharing=True)
    check(_Emu, make_config(fair_sharing=True), pop.snapshot)


@pytest.mark.gpu
@pytest.mark.parametrize("cfgn", [3, 4])
def test_rows_rebuild_fair_population_gpu(cfgn):
    from kueue_amd.api import make_config
    from kueue_amd.population import generate
    pop = generate(cfgn, fair_sharing=True)
    check(_Hip, make_config(fair_sharing=True), pop.snapshot)


# ---- a resident engine follows the cache through cycles with preemption ---------------------------------------------------------------
def _edited(snap, adds, evict, usage):
    """The snapshot after a cycle, as arrays: the admitted heads appended to their ClusterQueue's rows, Evicted marks, the new usage plane."""
    import copy
    a = snap.arrays
    n = snap.n_adm
    cq_of = np.repeat(np.arange(snap.n_cq), np.diff(a["cq_adm_off"]))
    rows = []
    for r in range(n):
        e0, e1 = int(a["adm_use_off"][r]), int(a["adm_use_off"][r + 1])
        rows.append(dict(cq=int(cq_of[r]), prio=int(a["adm_priority"][r]), qts=int(a["adm_queue_ts"][r]), rts=int(a["adm_reserve_ts"][r]), uid=int(a["adm_uid_rank"][r]),
                         flags=int(a["adm_flags"][r]) | (1 if r in evict else 0), fr=a["adm_use_fr"][e0:e1].tolist(), qty=a["adm_use_qty"][e0:e1].tolist()))
    table = []
    for c in range(snap.n_cq):
        table += [x for x in rows if x["cq"] == c] + [x for x in adds if x["cq"] == c]
    t = copy.copy(snap)
    b = dict(a)
    b["cq_adm_off"] = np.concatenate([[0], np.cumsum(np.bincount([x["cq"] for x in table], minlength=snap.n_cq))]).astype(np.int32)
    b["adm_priority"] = np.array([x["prio"] for x in table], np.int64); b["adm_queue_ts"] = np.array([x["qts"] for x in table], np.int64)
    b["adm_reserve_ts"] = np.array([x["rts"] for x in table], np.int64); b["adm_uid_rank"] = np.array([x["uid"] for x in table], np.uint32)
    b["adm_flags"] = np.array([x["flags"] for x in table], np.uint8)
    b["adm_use_off"] = np.concatenate([[0], np.cumsum([len(x["fr"]) for x in table])]).astype(np.int32)
    b["adm_use_fr"] = np.array([f for x in table for f in x["fr"]], np.int32); b["adm_use_qty"] = np.array([q for x in table for q in x["qty"]], np.int64)
    b["usage"] = np.ascontiguousarray(usage, np.int64).reshape(-1)
    for k in ("adm_priority", "adm_queue_ts", "adm_reserve_ts", "adm_uid_rank", "adm_flags", "adm_use_fr", "adm_use_qty"):
        if b[k].size == 0:
            b[k] = np.zeros(1, b[k].dtype)
    t.arrays = b
    t.n_adm = len(table)
    t.admitted = None
    t._struct = None
    return t


def _follow(oracle, make, seed):
    """Cycle 1 on the resident snapshot; its admissions become rows, the targets of its preemptions get the Evicted mark, the usage is
    committed — all on the device (kq_cycle_commit + kq_snapshot_patch_rows); then cycle 2 (the heads that stayed pending) must decide
    exactly as on a snapshot rebuilt from scratch, and as the oracle does."""
    from kueue_amd import _ffi as F
    from kueue_amd.api import Heads
    cfg, snap, heads = random_case(seed, fair=seed % 4 == 0, preemption=True, tight=True)
    if heads.n == 0:
        return 0
    oracle.derive(snap)
    cap = None   # (Decisions sizes the target arrays from the snapshot; the engine runs get the same default)
    want1 = oracle.cycle_run(cfg, snap, heads)
    usage1, n_adm1, (tcq, tfr, tq) = oracle.cycle_commit(cfg, snap, heads)
    admitted = [i for i in range(heads.n) if int(want1.a["action"][i]) == F.ACT_ADMIT]
    cqs = [int(heads.arrays["cq"][i]) for i in admitted]
    if len(set(cqs)) != len(cqs):
        return 0                     # (two admitted heads of one ClusterQueue: the triples cannot be told apart here)
    evict = set()
    for i in range(heads.n):
        if int(want1.a["action"][i]) == F.ACT_PREEMPT:
            evict |= {int(r) for r in want1.a["tgt_adm"][int(want1.a["tgt_off"][i]):int(want1.a["tgt_off"][i + 1])]}
    uid0 = int(snap.arrays["adm_uid_rank"].max()) + 1 if snap.n_adm else 0
    adds = []
    for k, i in enumerate(admitted):
        c = cqs[k]
        sel = [j for j in range(len(tcq)) if int(tcq[j]) == c]
        adds.append(dict(cq=c, prio=int(heads.arrays["priority"][i]), qts=int(heads.arrays["queue_ts"][i]), rts=10 ** 9 + k, uid=uid0 + k, flags=0,
                         fr=[int(tfr[j]) for j in sel], qty=[int(tq[j]) for j in sel]))
    snap1 = _edited(snap, adds, evict, usage1)
    rest = [w for i, w in enumerate(heads.workloads) if i not in set(admitted)]
    heads2 = Heads(snap1, rest, cycle=heads.cycle + 1)
    # resident engine: cycle 1, commit, row patch
    y = make(cfg)
    y.e.put(snap)
    got1 = y.run(heads)
    assert not want1.equal(got1)
    if hasattr(y.e, "commit"):
        y.e.commit()
    else:
        assert y.lib.kqe_cycle_commit(y.h, None) == 0
    add = dict(cq=[x["cq"] for x in adds], priority=[x["prio"] for x in adds], queue_ts=[x["qts"] for x in adds], reserve_ts=[x["rts"] for x in adds],
               uid_rank=[x["uid"] for x in adds], flags=[0] * len(adds), use_off=np.concatenate([[0], np.cumsum([len(x["fr"]) for x in adds])]),
               use_fr=[f for x in adds for f in x["fr"]], use_qty=[q for x in adds for q in x["qty"]]) if adds else None
    r = y.e.patch_rows([], add, sorted(evict))
    if isinstance(r, tuple):
        assert r[0] == 0, r[0]
    # the reference: a fresh engine on the rebuilt snapshot
    x = make(cfg)
    x.e.put(snap1)
    for name, p, q in zip(NAMES, read_all(x), read_all(y)):
        m = min(p.size, q.size)
        assert p.size == q.size and np.array_equal(p, q), (seed, name, p.size, q.size, np.flatnonzero(p[:m] != q[:m])[:8])
    if heads2.n:
        want2 = oracle.cycle_run(cfg, snap1, heads2)
        y.e.snap = snap1
        assert not want2.equal(x.run(heads2)), (seed, "fresh engine vs oracle")
        assert not want2.equal(y.run(heads2)), (seed, "resident engine vs oracle")
    x.e.close(); y.e.close()
    return len(adds) + len(evict)


def test_resident_engine_follows_the_cache_emulated(oracle):
    changed = sum(_follow(oracle, _Emu, seed) for seed in range(120))
    assert changed > 100


@pytest.mark.gpu
def test_resident_engine_follows_the_cache_gpu(oracle):
    changed = sum(_follow(oracle, _Hip, seed) for seed in range(80))
    assert changed > 60

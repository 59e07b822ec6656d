"""The admitted-row candidate structures built on the device (kq_rows.hpp) against the host's build_prep (kq_prep.hpp):
every array byte for byte — candidate rank order per tree, ascending order, rank positions, the flavor-resource buckets (positions, rows,
records, fingerprints), the three level orders, row records, per-ClusterQueue record bytes, the per-tree flags — on random snapshots and on
BASELINE-shaped populations; then the same cycles give the same decisions on the rebuilt structures.
CPU suite: the 1-lane emulation (std::stable_sort as the sort primitive). GPU suite: the HIP engine (rocPRIM radix sort)."""
import ctypes as C

import numpy as np
import pytest

from tests.randgen import random_case

N_STRUCT = 30
NAMES = ["adm_cq", "tree_row_off", "tree_rows", "tree_rows_asc", "rank_pos", "frb_off", "frb", "frbr", "cq_row_bytes", "adm_rec", "frec", "frl0", "frl1", "frl2",
         "frb_sig", "cs_ok", "rec_ok", "cq_adm_off", "adm_use_off", "adm_use_fr", "adm_use_qty", "adm_prio", "adm_qts", "adm_rts", "adm_uid", "adm_flags",
         "fs_ok", "fs_posoff", "fs_scan", "fs_apply"]


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

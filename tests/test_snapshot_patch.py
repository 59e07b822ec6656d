"""kq_snapshot_patch: the next cycle's snapshot when only usage and / or the admitted set moved. A patched engine must
decide exactly like an engine that was handed the new snapshot whole (kq_snapshot_put) — and like the oracle. The new snapshot
is what carrying out a cycle leaves behind: the preemption targets evicted (rows gone, their usage released through RemoveUsage),
a few more workloads finished."""
import numpy as np
import pytest

from kueue_amd import _ffi as F
from kueue_amd.api import make_config
from kueue_amd.population import generate
from tests.randgen import random_case


def _next_snapshot(oracle, cfg, snap, d, rnd):
    """Evict every preemption target of the cycle and let a few random other rows finish."""
    m = int(d.a["tgt_off"][-1])
    gone = set(int(r) for r in d.a["tgt_adm"][:m])
    for r in range(snap.n_adm):
        if rnd.random() < 0.1:
            gone.add(r)
    keep = np.array([r for r in range(snap.n_adm) if r not in gone], np.int64)
    usage = oracle.apply_ops(cfg, snap, [("remove", snap.admitted[r].name) for r in sorted(gone)]) if gone else snap.plane("usage")
    return snap.with_rows(keep, usage)


def _check(oracle, factory, cfg, snap, heads, rnd, what_seq):
    cap = max(64, 4 * snap.n_adm)
    patched = factory(cfg); whole = factory(cfg)
    try:
        patched.put(snap)
        d = patched.run(heads, tgt_cap=cap)
        cur = snap
        for what in what_seq:
            if what == F.PATCH_ADMITTED:
                nxt = _next_snapshot(oracle, cfg, cur, d, rnd)
            else:  # usage only: some quota is used up elsewhere; cohort levels re-derived by the oracle
                nxt = cur.with_rows(np.arange(cur.n_adm))
                u = nxt.plane("usage").copy()
                cqs = rnd.sample(range(cur.n_cq), max(1, cur.n_cq // 3))
                for c in cqs:
                    for fr in range(cur.n_fr):
                        if cur.arrays["quota_flags"].reshape(cur.N, cur.n_fr)[c, fr] & 1 and 0 <= u[c, fr] < (1 << 40):
                            u[c, fr] += rnd.randint(0, 3) * (1000 if fr % cur.n_resource == 0 else 1)
                nxt.arrays["usage"] = np.ascontiguousarray(u.reshape(-1))
                oracle.derive(nxt)
            patched.patch(nxt, what)
            whole.put(nxt)
            want = oracle.cycle_run(cfg, nxt, heads, want_usage=True)
            got_p = patched.run(heads, tgt_cap=cap)
            got_w = whole.run(heads, tgt_cap=cap)
            for got, tag in ((got_p, "patched"), (got_w, "whole")):
                assert getattr(got, "rc", 0) == 0
                bad = want.equal(got)
                assert not bad, (tag, what, bad)
            cur, d = nxt, got_p
    finally:
        patched.close(); whole.close()


@pytest.mark.parametrize("fair", [False, True])
def test_patch_equals_put_random_emulated(oracle, fair):
    import random
    from tests.emu import kqe
    n = 0
    for seed in range(60):
        cfg, snap, heads = random_case(seed, fair=fair, preemption=True)
        oracle.derive(snap)
        if snap.n_adm == 0:
            continue
        _check(oracle, kqe.EmuEngine, cfg, snap, heads, random.Random(seed), [F.PATCH_ADMITTED, F.PATCH_USAGE, F.PATCH_ADMITTED])
        n += 1
    assert n > 30


def test_patch_population_emulated(oracle):
    import random
    from tests.emu import kqe
    pop = generate(4, n_cq=40, per_cq=3)
    _check(oracle, kqe.EmuEngine, make_config(), pop.snapshot, pop.heads_for_cycle(0), random.Random(1), [F.PATCH_ADMITTED, F.PATCH_ADMITTED])


@pytest.mark.gpu
def test_patch_gpu(oracle):
    import random
    from kueue_amd.engine import Engine
    for seed in range(25):
        cfg, snap, heads = random_case(seed, fair=False, preemption=True)
        oracle.derive(snap)
        if snap.n_adm:
            _check(oracle, Engine, cfg, snap, heads, random.Random(seed), [F.PATCH_ADMITTED, F.PATCH_USAGE])
    pop = generate(4, n_cq=200, per_cq=3)
    _check(oracle, Engine, make_config(), pop.snapshot, pop.heads_for_cycle(0), random.Random(2), [F.PATCH_ADMITTED, F.PATCH_USAGE])

"""ONE root cohort tree split across ranks (kueue_amd/sharding.py SplitRoot; include/kq_engine.h kq_cycle_certificate /
kq_snapshot_usage_add): world_size-2 gloo on CPU, each rank running the EMULATED ENGINE (the device code, 1 lane) on its shard of a
single-root population, all-reduce of the usage deltas, exactness certificate, fallback. The merged decisions and the resident usage
plane of every rank must equal a single engine's, cycle after cycle (closed loop: admissions fold in, older ones finish)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _population(kind):
    from kueue_amd.population import generate
    if kind == "cfg3":
        return generate(3, n_cq=200, per_cq=8, fill=0.3), False     # 2 mid-level cohorts under one root, headroom at the root
    if kind == "cfg3-tight":
        return generate(3, n_cq=200, per_cq=8), False               # the BASELINE fill: the root row is the binding constraint
    return generate(4, n_cq=200, per_cq=4), False              # preemption: every cycle with targets falls back


def _single(kind, cycles, hold):
    """Reference: one engine, whole batch, the same folding of the cycle's delta."""
    from kueue_amd.api import make_config
    from tests.emu import kqe
    pop, fair = _population(kind)
    cfg = make_config(fair_sharing=fair)
    eng = kqe.EmuEngine(cfg)
    eng.put(pop.snapshot)
    out, held = [], []
    delta = torch.zeros(pop.snapshot.N * pop.snapshot.n_fr, dtype=torch.int64)
    nq_cells = pop.snapshot.n_cq * pop.snapshot.n_fr
    for c in range(cycles):
        heads = pop.heads_for_cycle(c, cycle=c + 1)
        d = eng.run(heads, tgt_cap=4 * pop.snapshot.n_adm)
        assert d.rc == 0
        eng.certificate(delta.data_ptr())
        fold = delta[:nq_cells].clone()
        eng.usage_add(fold.data_ptr(), +1)
        held.append(fold)
        if len(held) > hold:
            old = held.pop(0)          # keep the tensor alive across the call
            eng.usage_add(old.data_ptr(), -1)
        out.append(({k: v.copy() for k, v in d.a.items()}, eng.read_usage().copy()))
    eng.close()
    return out


def _worker(rank, world, port, kind, cycles, hold, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kueue_amd.api import make_config
    from kueue_amd.sharding import SplitRoot
    from tests.emu import kqe
    pop, fair = _population(kind)
    cfg = make_config(fair_sharing=fair)

    class E(kqe.EmuEngine):                       # eng.run with the Engine's signature
        def run(self, heads, tgt_cap=None):
            d = super().run(heads, tgt_cap=tgt_cap)
            assert d.rc == 0, d.error
            return d
    eng = E(cfg)
    eng.put(pop.snapshot)
    sr = SplitRoot(eng, pop.snapshot, cfg, dist, rank, world)
    out, held = [], []
    for c in range(cycles):
        heads = pop.heads_for_cycle(c, cycle=c + 1)
        merged, exact = sr.cycle(heads, tgt_cap=4 * pop.snapshot.n_adm)
        held.append(sr.last_delta)
        if len(held) > hold:
            sr.release(held.pop(0))
        out.append(({k: v.copy() for k, v in merged.a.items()}, eng.read_usage().copy(), exact))
    q.put((rank, out, sr.stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,cycles,min_exact", [("cfg3", 6, 1), ("cfg3-tight", 5, 0), ("cfg4c", 2, 0)])
def test_split_root_world2(kind, cycles, min_exact):
    hold = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + {"cfg3": 1, "cfg3-tight": 2, "cfg4c": 3}[kind]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, cycles, hold, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, out, stats = q.get(timeout=150)
        got[rank] = (out, stats)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    want = _single(kind, cycles, hold)
    for rank in (0, 1):
        out, stats = got[rank]
        assert stats["cycles"] == cycles and stats["exact"] + stats["fallback"] == cycles
        for c, ((wa, wu), (ga, gu, exact)) in enumerate(zip(want, out)):
            m = int(wa["tgt_off"][-1])
            for k in wa:
                if k in ("tgt_adm", "tgt_reason"):
                    assert np.array_equal(wa[k][:m], ga[k][:m]), (kind, rank, c, k, exact)
                else:
                    assert np.array_equal(wa[k], ga[k]), (kind, rank, c, k, exact)
            assert np.array_equal(wu, gu), (kind, rank, c, "usage", exact)
    assert got[0][1]["exact"] >= min_exact, got[0][1]
    print(kind, got[0][1])

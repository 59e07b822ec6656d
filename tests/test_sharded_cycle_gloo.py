"""Sharded nominate, merged process (kueue_amd/sharding.py ShardedCycle; include/kq_engine.h kq_cycle_nominate_shard /
kq_cycle_process_merged): world_size-2 gloo on CPU, each rank running the EMULATED ENGINE. Every rank nominates half of the heads, one
all-reduce(SUM) merges the nominations, every rank runs order + processEntry on the merged batch. No fallback path exists: decisions
(targets, reasons, iterator positions), the byte counts and the resident usage of every rank must equal a single engine's in every cycle
of a closed loop — at the BASELINE fill of cfg 3 (root row binding), with classical preemption (cfg 4c) and with fair sharing."""
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
    if kind == "cfg3-tight":
        return generate(3, n_cq=200, per_cq=8), False               # the BASELINE fill: the root row is the binding constraint
    if kind == "cfg4c":
        return generate(4, n_cq=120, per_cq=4), False               # classical preemption: targets, overlap recomputation
    return generate(4, n_cq=60, per_cq=4, fair_sharing=True), True  # fair sharing + fair preemption


def _loop(kind, cycles, hold, make_cycle):
    from kueue_amd.api import make_config
    from tests.emu import kqe
    pop, fair = _population(kind)
    cfg = make_config(fair_sharing=fair)
    eng = kqe.EmuEngine(cfg)
    eng.put(pop.snapshot)
    run = make_cycle(eng)
    out, live = [], 0
    for c in range(cycles):
        heads = pop.heads_for_cycle(c, cycle=c + 1)
        d = run(heads, 4 * pop.snapshot.n_adm)
        eng.commit(); live += 1
        if live > hold:
            eng.release(hold + 1); live -= 1
        out.append(({k: v.copy() for k, v in d.a.items()}, eng.read_usage().copy()))
    eng.close()
    return out


def _single(kind, cycles, hold):
    from kueue_amd.api import Decisions
    def mk(eng):
        def run(heads, cap):
            d = eng.run(heads, tgt_cap=cap, rsn_cap=4096)
            assert d.rc == 0, d.error
            return d
        return run
    return _loop(kind, cycles, hold, mk)


def _worker(rank, world, port, kind, cycles, hold, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kueue_amd.sharding import ShardedCycle
    def mk(eng):
        sc = ShardedCycle(eng, dist, rank, world)
        return lambda heads, cap: sc.cycle(heads, tgt_cap=cap, rsn_cap=4096)
    out = _loop(kind, cycles, hold, mk)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,cycles", [("cfg3-tight", 6), ("cfg4c", 4), ("cfg4f", 2)])
def test_sharded_cycle_world2(kind, cycles):
    want = _single(kind, cycles, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + len(kind)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, cycles, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        r, out = q.get(timeout=600)
        got[r] = out
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        for c, ((wd, wu), (gd, gu)) in enumerate(zip(want, got[r])):
            for k in wd:
                if k in ("tgt_adm", "tgt_reason"):
                    m = int(wd["tgt_off"][-1])
                    assert np.array_equal(wd[k][:m], gd[k][:m]), (kind, r, c, k)
                elif k.startswith("rsn_") and k != "rsn_off":
                    m = int(wd["rsn_off"][-1])
                    assert np.array_equal(wd[k][:m], gd[k][:m]), (kind, r, c, k)
                else:
                    assert np.array_equal(wd[k], gd[k]), (kind, r, c, k)
            assert np.array_equal(wu, gu), (kind, r, c, "resident usage")


def test_sharded_cycle_world1_equals_run():
    """world 1 (no collective): nominate_shard + process_merged == kq_cycle_run, bytes included."""
    from kueue_amd.api import make_config
    from kueue_amd.sharding import ShardedCycle
    from tests.emu import kqe
    pop, _ = _population("cfg4c")
    cfg = make_config()
    a, b = kqe.EmuEngine(cfg), kqe.EmuEngine(cfg)
    a.put(pop.snapshot); b.put(pop.snapshot)
    sc = ShardedCycle(b, None, 0, 1)
    for c in range(3):
        heads = pop.heads_for_cycle(c, cycle=c + 1)
        want = a.run(heads, tgt_cap=4 * pop.snapshot.n_adm, want_usage=True)
        got = sc.cycle(heads, tgt_cap=4 * pop.snapshot.n_adm)
        assert not want.equal(got), want.equal(got)
        assert np.array_equal(want.usage_after, b.read_usage_work())
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cfg3-tight", "cfg4c"])
def test_sharded_cycle_world1_gpu(kind):
    """The HIP engine through the C ABI: kq_cycle_nominate_shard + kq_cycle_process_merged (world 1: export, no collective, import) equal
    kq_cycle_run in a closed loop, device exchange buffer."""
    from kueue_amd.api import make_config
    from kueue_amd.engine import Engine
    from kueue_amd.sharding import ShardedCycle
    pop, fair = _population(kind)
    cfg = make_config(fair_sharing=fair)
    a, b = Engine(cfg), Engine(cfg)
    a.put(pop.snapshot); b.put(pop.snapshot)
    sc = ShardedCycle(b, None, 0, 1, device="cuda:0")
    try:
        for c in range(4):
            heads = pop.heads_for_cycle(c, cycle=c + 1)
            want = a.run(heads, tgt_cap=4 * pop.snapshot.n_adm, rsn_cap=128 * heads.n)
            got = sc.cycle(heads, tgt_cap=4 * pop.snapshot.n_adm, rsn_cap=128 * heads.n)
            assert not want.equal(got), (kind, c, want.equal(got))
            a.commit(); b.commit()
            assert np.array_equal(a.read_usage(), b.read_usage())
    finally:
        a.close(); b.close()


def test_nominate_shard_reports_its_own_capacity_error():
    """A rank whose nominations overflow the target pool fails in kq_cycle_nominate_shard with KQ_ECAPACITY (the merged error word of
    kq_cycle_process_merged only says that some rank failed)."""
    import torch
    from kueue_amd.api import Decisions, make_config
    from tests.emu import kqe
    pop, fair = _population("cfg4f")
    eng = kqe.EmuEngine(make_config(fair_sharing=fair))
    eng.put(pop.snapshot)
    heads = pop.heads_for_cycle(0, cycle=1)
    d = Decisions(heads, tgt_cap=1)
    x = torch.zeros(eng.shard_words(heads, d, 1), dtype=torch.int64)
    with pytest.raises(AssertionError) as ei:
        eng.nominate_shard(heads, None, 1, 0, x.data_ptr(), d)
    assert ei.value.args[0][0] == -5 and b"tgt_cap" in ei.value.args[0][1]
    eng.close()

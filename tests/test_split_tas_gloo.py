"""ONE TAS flavor split across ranks (kueue_amd/sharding.py SplitTAS; include/kq_tas.h kq_tas_usage_delta / kq_tas_overflow /
kq_tas_usage_add / kq_tas_admit): world_size-2 gloo on CPU, each rank running the EMULATED TAS engine on its shard of the cycle's
pending workloads, all-reduce of the leaf-usage planes, certificate, contended walk. Placements, the admitted set and the resident
leaf usage of every rank must equal a single engine's find + entry-order walk, cycle after cycle (admissions accumulate)."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batches(kind, cycles):
    """-> (topology, [Requests per cycle], [entry order per cycle])"""
    from kueue_amd.tas_population import generate_tas
    if kind == "roomy":       # two workloads per cycle on 512 hosts: cycles whose two placements do not collide keep the certificate
        topo, rq = generate_tas(n_workloads=2 * cycles, seed=77, blocks=4, racks=8, hosts=16)
    else:                     # "tight": 128 hosts, 60 workloads of up to 64 pods per cycle: leaves overflow, contended walk
        topo, rq = generate_tas(n_workloads=60 * cycles, seed=78, blocks=2, racks=4, hosts=16)
    per = rq.n_workloads // cycles
    rng = np.random.default_rng(5)
    bs = [rq.subset(np.arange(c * per, (c + 1) * per)) for c in range(cycles)]
    orders = [None if c % 2 == 0 else rng.permutation(per).astype(np.int32) for c in range(cycles)]
    return topo, bs, orders


def _single(kind, cycles):
    from tests.emu import kqe
    topo, bs, orders = _batches(kind, cycles)
    eng = kqe.EmuTas()
    eng.put(topo)
    out = []
    for rq, order in zip(bs, orders):
        res = eng.find(rq)
        adm = eng.admit(rq, res, order)
        out.append(({k: v.copy() for k, v in res.a.items()}, adm.copy(), eng.read_usage().copy()))
    eng.close()
    return out


def _worker(rank, world, port, kind, cycles, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kueue_amd.sharding import SplitTAS
    from tests.emu import kqe
    topo, bs, orders = _batches(kind, cycles)
    eng = kqe.EmuTas()
    eng.put(topo)
    sp = SplitTAS(eng, topo, dist, rank, world)
    out = []
    for rq, order in zip(bs, orders):
        merged, adm = sp.cycle(rq, order)
        out.append(({k: v.copy() for k, v in merged.a.items()}, adm.copy(), eng.read_usage().copy()))
    q.put((rank, out, sp.stats))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("kind,cycles", [("roomy", 8), ("tight", 3)])
def test_split_tas_world2(kind, cycles):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29720 + {"roomy": 1, "tight": 2}[kind]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, cycles, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, out, stats = q.get(timeout=150)
        got[rank] = (out, stats)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _single(kind, cycles)
    for rank in (0, 1):
        out, stats = got[rank]
        assert stats["cycles"] == cycles and stats["exact"] + stats["contended"] == cycles
        for c, ((wa, wadm, wu), (ga, gadm, gu)) in enumerate(zip(want, out)):
            nd = int(wa["dom_off"][-1])
            for k in ("status", "operand_a", "operand_b", "dom_off"):
                assert np.array_equal(wa[k], ga[k]), (kind, rank, c, k)
            for k in ("dom_leaf", "dom_count"):
                assert np.array_equal(wa[k][:nd], ga[k][:nd]), (kind, rank, c, k)
            assert np.array_equal(wadm, gadm), (kind, rank, c, "admitted")
            assert np.array_equal(wu, gu), (kind, rank, c, "usage")
    if kind == "roomy":
        assert got[0][1]["exact"] >= 1, got[0][1]
    else:
        assert got[0][1]["contended"] >= 1 and got[0][1]["walked"] > 0, got[0][1]
    assert sum(int(w[1].sum()) for w in want) > 0
    print(kind, got[0][1], [int(w[1].sum()) for w in want])

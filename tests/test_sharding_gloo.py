"""Multi-rank path on CPU: world_size-2 gloo. Root-cohort sharding must reproduce the single-snapshot decisions.

The per-rank cycle runs on the oracle here (there is no GPU in the CPU suite); what is under test is the sharding itself
(kueue_amd/sharding.py) and the torch.distributed plumbing bench.py uses (barrier + max/sum all_reduce)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _forest(seed):
    """A population with several independent root trees, built from api objects."""
    import random
    from kueue_amd.api import ClusterQueue, Cohort, FlavorQuotas, PodSet, ResourceGroup, ResourceQuota, Workload
    rnd = random.Random(seed)
    cohorts, cqs, admitted, pending = [], [], [], []
    for t in range(5):
        cohorts += [Cohort(f"root{t}"), Cohort(f"mid{t}", f"root{t}")]
        for i in range(3):
            fq = [FlavorQuotas(f"f{j}", {"cpu": ResourceQuota(rnd.randint(2, 8) * 1000, rnd.choice([None, 3000]), rnd.choice([None, 2000]))}) for j in range(2)]
            cq = ClusterQueue(f"t{t}-cq{i}", cohort=rnd.choice([f"root{t}", f"mid{t}"]), resource_groups=[ResourceGroup(fq)],
                              within_cluster_queue="LowerPriority", reclaim_within_cohort="Any")
            cqs.append(cq)
            for a in range(rnd.randint(0, 3)):
                ps = PodSet("main", 1, requests={"cpu": rnd.randint(1, 4) * 1000}, flavors={"cpu": rnd.choice(["f0", "f1"])})
                admitted.append(Workload(f"{cq.name}-adm{a}", cq.name, priority=rnd.randint(0, 3), creation_ts=rnd.randint(0, 9), pod_sets=[ps],
                                         reserve_ts=rnd.randint(0, 9), uid=f"u{t}{i}{a}"))
            pending.append(Workload(f"{cq.name}-pend", cq.name, priority=rnd.randint(0, 4), creation_ts=rnd.randint(0, 9),
                                    pod_sets=[PodSet("main", 1, requests={"cpu": rnd.randint(1, 6) * 1000})]))
    cqs.append(ClusterQueue("standalone", resource_groups=[ResourceGroup([FlavorQuotas("f0", {"cpu": ResourceQuota(4000)})])]))
    pending.append(Workload("standalone-pend", "standalone", pod_sets=[PodSet("main", 1, requests={"cpu": 3000})]))
    return cqs, cohorts, admitted, pending


def _decide(cqs, cohorts, admitted, pending):
    from kueue_amd.api import Heads, Snapshot, make_config
    from oracle import kqo
    snap = Snapshot(cqs, cohorts, admitted, now_ns=100)
    kqo.derive(snap)
    pend = sorted(pending, key=lambda w: w.cluster_queue)
    heads = Heads(snap, pend, cycle=1)
    d = kqo.cycle_run(make_config(), snap, heads)
    out = {}
    for i, w in enumerate(pend):
        out[w.name] = (int(d.a["status"][i]), int(d.a["action"][i]), int(d.a["mode"][i]), int(d.a["borrowing"][i]), sorted(d.target_names(i)),
                       [sorted(x.items()) for x in d.flavors_of(i)])
    return out


def _worker(rank, world, port, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kueue_amd.sharding import partition_roots, shard
    cqs, cohorts, admitted, pending = _forest(seed)
    trees = partition_roots(cqs, cohorts, admitted, pending, world)[rank]
    mine = _decide(*shard(cqs, cohorts, admitted, pending, trees))
    dist.barrier()
    t = torch.tensor([float(len(mine)), 1.0 + rank], dtype=torch.float64)
    tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        merged = {}
        for g in gathered:
            assert not (set(g) & set(merged)), "a workload was decided on two ranks"
            merged.update(g)
        q.put((merged, float(tsum[0]), float(tmax[1])))
    dist.destroy_process_group()


@pytest.mark.parametrize("seed", [1, 2])
def test_root_cohort_sharding_world2(seed):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + seed
    procs = [ctx.Process(target=_worker, args=(r, 2, port, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, total, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    single = _decide(*_forest(seed))
    assert merged == single
    assert total == len(single) and tmax == 2.0


def test_partition_is_balanced_and_total():
    from kueue_amd.sharding import partition_roots
    cqs, cohorts, admitted, pending = _forest(3)
    parts = partition_roots(cqs, cohorts, admitted, pending, 4)
    allt = [t for p in parts for t in p]
    assert len(allt) == len(set(allt)) == 6  # 5 roots + 1 standalone CQ
    assert all(parts)

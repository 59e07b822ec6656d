"""GPU parity for the TAS path: the HIP engine through include/kq_tas.h vs the oracle, bit-exact (statuses, failure
operands, assignments, algorithmic bytes), on seeded random topologies and on a BASELINE-sized one."""
import numpy as np
import pytest

from kueue_amd import tas as T
from tests.tasgen import deep_tas_case, random_tas_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(10))
def test_tas_random_gpu(oracle, block):
    eng = T.TASEngine()
    try:
        for seed in range(block * 40, block * 40 + 40):
            big = seed % 5 == 0
            topo, rq = random_tas_case(seed, max_blocks=5 if big else 3, max_racks=8 if big else 4, max_hosts=20 if big else 6)
            if topo.n_leaves == 0:
                continue
            want = oracle.tas_find(topo, rq)
            eng.put(topo)
            got = eng.find(rq)
            bad = want.equal(got)
            assert not bad, (seed, bad)
            assert got.bytes == want.bytes, seed
    finally:
        eng.close()


def test_tas_deep_and_wide_gpu(oracle):
    """9-16 topology levels, 17-30 resources per node: the limits of the API (topology_types.go MaxItems=16, scheduler_tas_bench_test.go:46)."""
    eng = T.TASEngine()
    try:
        placed = 0
        for seed in range(48):
            topo, rq = deep_tas_case(seed, n_levels=9 + seed % 8, n_res=17 + seed % 14)
            want = oracle.tas_find(topo, rq)
            eng.put(topo)
            got = eng.find(rq)
            bad = want.equal(got)
            assert not bad, (seed, bad)
            assert got.bytes == want.bytes, seed
            placed += int((got.a["status"] == 0).sum())
        assert placed > 50
    finally:
        eng.close()


def test_tas_cfg5_sample(oracle):
    """BASELINE configs[4] topology (8 blocks x 8 racks x 64 hosts = 4096 leaves): 400 workloads of the population."""
    from kueue_amd.tas_population import generate_tas
    topo, rq = generate_tas(n_workloads=400)
    want = oracle.tas_find(topo, rq)
    eng = T.TASEngine()
    try:
        eng.put(topo)
        got = eng.find(rq)
        assert not want.equal(got), want.equal(got)
        assert got.bytes == want.bytes
        # usage application round trip (updateTASUsage): add every successful assignment, then remove it again
        R = len(topo.resources)
        before = eng.read_usage().copy()
        for i in range(0, rq.n, 7):
            a = got.assignment(i)
            if a:
                eng.usage_apply(a, rq.arrays["single_pod_requests"].reshape(-1, R)[i], add=True)
        for i in range(0, rq.n, 7):
            a = got.assignment(i)
            if a:
                eng.usage_apply(a, rq.arrays["single_pod_requests"].reshape(-1, R)[i], add=False)
        assert np.array_equal(eng.read_usage(), before)
    finally:
        eng.close()

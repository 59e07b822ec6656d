"""GPU parity for the entry-order admission walk and the split of one TAS flavor (include/kq_tas.h: kq_tas_admit, kq_tas_usage_delta,
kq_tas_usage_add, kq_tas_overflow; kueue_amd/sharding.py SplitTAS): the HIP engine through the C ABI vs the oracle's walk
(oracle/kq_tas_oracle.cpp kqo_tas_admit), bit-exact on the admitted set and the leaf usage. (Named to run after the other GPU
files: these entry points were added at the very end of round 2.)"""
import numpy as np
import pytest

from kueue_amd import tas as T
from tests.tasgen import random_tas_case

pytestmark = pytest.mark.gpu


def _case(seed, **kw):
    topo, rq = random_tas_case(seed, **kw)
    rq.arrays.pop("simulate_empty", None)
    rq._struct = None
    return topo, rq


@pytest.mark.parametrize("block", range(4))
def test_admit_walk_gpu(oracle, block):
    eng = T.TASEngine()
    try:
        for seed in range(block * 30, block * 30 + 30):
            topo, rq = _case(seed, n_workloads=40)
            if topo.n_leaves == 0:
                continue
            rng = np.random.default_rng(seed)
            order = None if seed % 3 == 0 else rng.permutation(rq.n_workloads).astype(np.int32)
            res = oracle.tas_find(topo, rq)
            want_adm, want_usage = oracle.tas_admit(topo, rq, res, order)
            eng.put(topo)
            got = eng.find(rq)
            assert not res.equal(got), seed
            adm = eng.admit(rq, got, order)
            assert np.array_equal(adm, want_adm), seed
            assert np.array_equal(eng.read_usage(), want_usage), seed
    finally:
        eng.close()


def test_admit_cfg5_sample(oracle):
    """BASELINE configs[4] topology (4096 leaves), 3000 workloads of the population walked in a shuffled entry order."""
    from kueue_amd.tas_population import generate_tas
    topo, rq = generate_tas(n_workloads=3000)
    order = np.random.default_rng(1).permutation(rq.n_workloads).astype(np.int32)
    res = oracle.tas_find(topo, rq)
    want_adm, want_usage = oracle.tas_admit(topo, rq, res, order)
    eng = T.TASEngine()
    try:
        eng.put(topo)
        got = eng.find(rq)
        adm = eng.admit(rq, got, order)
        assert np.array_equal(adm, want_adm)
        assert np.array_equal(eng.read_usage(), want_usage)
        assert 0 < int(adm.sum()) < rq.n_workloads
    finally:
        eng.close()


@pytest.mark.parametrize("kind", ["random", "cfg5"])
def test_split_protocol_on_device_planes(oracle, kind):
    """SplitTAS at world 1 with the exchange plane in HBM: delta / overflow / plane add / contended walk equal the plain walk."""
    import torch
    from kueue_amd.sharding import SplitTAS
    from kueue_amd.tas_population import generate_tas
    cases = [_case(s, n_workloads=60) for s in range(200, 230)] if kind == "random" else [generate_tas(n_workloads=2000, seed=9)]
    eng = T.TASEngine()
    try:
        for topo, rq in cases:
            if topo.n_leaves == 0:
                continue
            res = oracle.tas_find(topo, rq)
            want_adm, want_usage = oracle.tas_admit(topo, rq, res, None)
            eng.put(topo)
            sp = SplitTAS(eng, topo, None, 0, 1, device="cuda")
            merged, adm = sp.cycle(rq)
            torch.cuda.synchronize()
            assert not res.equal(merged)
            assert np.array_equal(adm, want_adm)
            assert np.array_equal(eng.read_usage(), want_usage)
    finally:
        eng.close()

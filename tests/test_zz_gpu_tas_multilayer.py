"""GPU parity for multi-layer slice constraints (TASMultiLayerTopology): the HIP engine through include/kq_tas.h against the five
multi-layer cases of TestFindTopologyAssignments (Go expectations) and against the oracle on seeded random constraint lists —
statuses, operands, assignments, per-layer fit counts, regenerated messages and the algorithmic byte counter, bit-exact."""
import pytest

from kueue_amd import tas as T
from tests.tasgen import random_tas_multilayer_case
from tests.test_oracle_tas import CASES, build, check, is_multilayer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [c for c in CASES if is_multilayer(c)], ids=lambda c: c["name"][:80])
def test_go_multilayer_cases_gpu(oracle, case):
    topo, rq = build(case)
    eng = T.TASEngine()
    try:
        eng.put(topo)
        out = eng.find(rq)
    finally:
        eng.close()
    check(case, out, topo)
    want = oracle.tas_find(topo, rq)
    assert not want.equal(out), want.equal(out)
    assert out.bytes == want.bytes, (out.bytes, want.bytes)


def test_multilayer_random_gpu(oracle):
    eng = T.TASEngine()
    try:
        for seed in range(200):
            topo, rq = random_tas_multilayer_case(seed)
            want = oracle.tas_find(topo, rq)
            eng.put(topo)
            got = eng.find(rq)
            bad = want.equal(got)
            assert not bad, (seed, bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
            assert got.bytes == want.bytes, seed
            for i in range(rq.n):
                assert got.message(i) == want.message(i), (seed, i)
    finally:
        eng.close()

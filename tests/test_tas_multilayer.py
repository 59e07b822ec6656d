"""Multi-layer slice constraints (TASMultiLayerTopology; buildSliceSizeAtLevel tas_flavor_snapshot.go:1123, the rounding of
fillInCountsHelper :1950-1967, the per-level slice size of the descent :1049-1070, multiLayerNotFitMessage :2030): the engine's device
code (1-lane CPU emulation) vs the oracle on seeded random topologies and constraint lists, valid and invalid. The oracle itself is
pinned by the five multi-layer cases of TestFindTopologyAssignments in tests/golden/tas_find.yaml (tests/test_oracle_tas.py)."""
import numpy as np
import pytest

from kueue_amd import tas as T
from tests.emu import kqe
from tests.tasgen import random_tas_multilayer_case


@pytest.mark.parametrize("seed", range(300))
def test_multilayer_random(oracle, seed):
    topo, rq = random_tas_multilayer_case(seed)
    want = oracle.tas_find(topo, rq)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        got = eng.find(rq)
    finally:
        eng.close()
    bad = want.equal(got)
    assert not bad, (bad, {k: (want.a[k].tolist(), got.a[k].tolist()) for k in bad})
    assert got.bytes == want.bytes
    for i in range(rq.n):
        assert got.message(i) == want.message(i)


def test_generator_reaches_every_outcome(oracle):
    seen = set()
    layered_ok = 0
    for seed in range(300):
        topo, rq = random_tas_multilayer_case(seed)
        out = oracle.tas_find(topo, rq)
        seen.update(int(s) for s in out.a["status"])
        if "n_layers" in rq.arrays:
            layered_ok += int(((out.a["status"] == T.TAS_OK) & (rq.arrays["n_layers"] > 1)).sum())
    assert {T.TAS_OK, T.TAS_NOT_FIT_LAYERS, T.TAS_BAD_LAYER, T.TAS_BAD_SLICE_SIZE, T.TAS_NOT_FIT} <= seen, seen
    assert layered_ok > 100, layered_ok


def test_placement_respects_every_layer(oracle):
    """Size-independent property: in an assignment of a podset with valid layers, the pods inside every domain of a layer's level come
    in multiples of the layer's size."""
    checked = 0
    for seed in range(300):
        topo, rq = random_tas_multilayer_case(seed)
        if "n_layers" not in rq.arrays:
            continue
        out = oracle.tas_find(topo, rq)
        for i in range(rq.n):
            if int(out.a["status"][i]) != T.TAS_OK or int(rq.arrays["n_layers"][i]) < 2:
                continue
            if rq.arrays["group"][i] >= 0:
                continue   # leader + workers share domains: the multiple holds for the workers' pods only after the leader's are set aside
            for j in range(int(rq.arrays["n_layers"][i])):
                lv = int(rq.arrays["layer_level"][i * T.TAS_MAX_LEVELS + j]); sz = int(rq.arrays["layer_size"][i * T.TAS_MAX_LEVELS + j])
                per = {}
                for leaf, cnt in out.assignment(i):
                    key = tuple(topo.level_values[-1][leaf][:lv + 1])
                    per[key] = per.get(key, 0) + cnt
                assert all(c % sz == 0 for c in per.values()), (seed, i, j, per, sz)
                checked += 1
    assert checked > 50, checked

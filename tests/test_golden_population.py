"""BASELINE.json configs[3] AT ITS STATED SIZE (1000 ClusterQueues, 40 k admitted workloads): the HIP engine against expectations
the CPU oracle produced offline (tests/golden/gen_population_golden.py -> tests/golden/pop_<name>_c<cycle>.npz). A full-size
fair-sharing + preemption cycle costs the oracle tens of minutes, so it cannot run inside a test; its complete output can."""
import hashlib
import os

import numpy as np
import pytest

from kueue_amd.api import make_config
from kueue_amd.population import generate
from tests.golden.gen_population_golden import ACCOUNTING, CASES, cycle_input, digest_inputs, path_of

PRESENT = [(n, c) for n, (_, _, cycles) in CASES.items() for c in cycles if os.path.exists(path_of(n, c))]
_pops = {}


def _pop(name):
    if name not in _pops:
        _pops[name] = generate(**CASES[name][0])
    return _pops[name]


def test_golden_files_present():
    assert {n for n, _ in PRESENT} >= {"cfg4c", "cfg3f"}, PRESENT


@pytest.mark.parametrize("name,cycle", PRESENT, ids=[f"{n}-c{c}" for n, c in PRESENT])
def test_golden_inputs_are_current(name, cycle):
    """The committed expectation belongs to the population the generator produces today (seeded, deterministic)."""
    g = np.load(path_of(name, cycle))
    pop = _pop(name)
    snap, heads = cycle_input(pop, name, cycle)
    assert digest_inputs(snap, heads) == bytes(g["inputs_sha256"]).hex()
    assert len(g["status"]) == heads.n


@pytest.mark.gpu
@pytest.mark.parametrize("name,cycle", PRESENT, ids=[f"{n}-c{c}" for n, c in PRESENT])
def test_engine_matches_offline_oracle(name, cycle):
    from kueue_amd.engine import Engine
    g = np.load(path_of(name, cycle))
    pop = _pop(name)
    fair = CASES[name][1]
    cfg = make_config(fair_sharing=fair)
    snap, heads = cycle_input(pop, name, cycle)
    eng = Engine(cfg)
    try:
        eng.put(snap)
        m = int(g["tgt_off"][-1])
        got = eng.run(heads, tgt_cap=max(4096, (32 if fair else 4) * pop.snapshot.n_adm))
        for k in ("status", "action", "nominated_mode", "mode", "requeue_reason", "skip", "borrowing", "order", "flavor", "res_mode",
                  "tried_idx", "ps_count", "tgt_off"):
            assert np.array_equal(got.a[k], g[k]), (name, k)
        assert np.array_equal(got.a["tgt_adm"][:m], g["tgt_adm"]), name
        assert np.array_equal(got.a["tgt_reason"][:m], g["tgt_reason"]), name
        usage = np.ascontiguousarray(eng.usage_after())
        assert hashlib.sha256(usage.tobytes()).digest() == bytes(g["usage_sha256"]), name
        if "accounting" in g and int(g["accounting"][0]) == ACCOUNTING:   # (an older file's byte total follows the older accounting rule)
            assert got.bytes == int(g["bytes_total"][0]), (name, got.bytes, int(g["bytes_total"][0]))
    finally:
        eng.close()

"""Entry-order admission of a batch's TopologyAssignments (kq_tas_admit: the TAS side of processEntry, scheduler.go:392-523) and the
pieces of the cross-GPU split of one TAS flavor (kq_tas_usage_delta / kq_tas_usage_add / kq_tas_overflow, kueue_amd/sharding.py
SplitTAS): the device code (1-lane emulation) against the oracle's walk (oracle/kq_tas_oracle.cpp kqo_tas_admit), bit-exact on the
admitted set and the leaf usage."""
import numpy as np
import pytest
import torch

from kueue_amd.sharding import SplitTAS
from tests.emu import kqe
from tests.tasgen import random_tas_case


def _case(seed, n_workloads=40):
    topo, rq = random_tas_case(seed, n_workloads=n_workloads)
    rq.arrays.pop("simulate_empty", None)          # admission is about real usage
    rq._struct = None
    return topo, rq


def _order(seed, nw, kind):
    rng = np.random.default_rng(seed)
    if kind == "identity":
        return None
    if kind == "perm":
        return rng.permutation(nw).astype(np.int32)
    return rng.permutation(nw)[: max(1, nw // 2)].astype(np.int32)   # a subset: the others stay un-admitted


@pytest.mark.parametrize("seed", range(120))
def test_admit_walk_matches_oracle(oracle, seed):
    topo, rq = _case(seed)
    res = oracle.tas_find(topo, rq)
    order = _order(seed, rq.n_workloads, ["identity", "perm", "subset"][seed % 3])
    want_adm, want_usage = oracle.tas_admit(topo, rq, res, order)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        got = eng.find(rq)
        assert not res.equal(got)
        adm = eng.admit(rq, got, order)
        assert np.array_equal(adm, want_adm), (adm.tolist(), want_adm.tolist())
        assert np.array_equal(eng.read_usage(), want_usage)
    finally:
        eng.close()


def test_admit_rejects_bad_input(oracle):
    topo, rq = _case(5)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        got = eng.find(rq)
        with pytest.raises(AssertionError):
            eng.admit(rq, got, np.array([rq.n_workloads], np.int32))      # order entry out of range
        nd = int(got.a["dom_off"][-1])
        if nd:
            got.a["dom_leaf"][0] = topo.n_leaves                          # assigned leaf out of range
            with pytest.raises(AssertionError):
                eng.admit(rq, got)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(60))
def test_delta_plane_and_overflow(oracle, seed):
    """usage_delta = what admitting every placed workload adds; overflow = leaves where usage + plane > free capacity."""
    topo, rq = _case(seed)
    R = len(topo.resources)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        res = eng.find(rq)
        plane = torch.zeros(topo.n_leaves * R, dtype=torch.int64)
        rng = np.random.default_rng(seed)
        sel = (rng.random(rq.n_workloads) < 0.7).astype(np.uint8) if seed % 2 else None
        eng.usage_delta(rq, res, plane.data_ptr(), wl_sel=sel)
        exp = np.zeros((topo.n_leaves, R), np.int64)
        spr = rq.arrays["single_pod_requests"].reshape(-1, R)
        off = rq.arrays["wl_off"]
        pods = topo.resource_index["pods"]
        for w in range(rq.n_workloads):
            ps = range(off[w], off[w + 1])
            if (sel is not None and not sel[w]) or any(res.a["status"][p] != 0 for p in ps):
                continue
            for p in ps:
                for leaf, cnt in res.assignment(p):
                    exp[leaf] += np.maximum(spr[p], 0) * cnt
                    exp[leaf, pods] += cnt
        assert np.array_equal(plane.numpy().reshape(-1, R), exp)
        before = eng.read_usage().reshape(-1, R).copy()
        over = eng.overflow(plane.data_ptr())
        free = topo.arrays["free_capacity"].reshape(-1, R)
        assert np.array_equal(over.astype(bool), ((before + exp) > free).any(axis=1))
        assert np.array_equal(eng.overflow(None).astype(bool), (before > free).any(axis=1))
        eng.usage_add(plane.data_ptr(), +1)
        assert np.array_equal(eng.read_usage().reshape(-1, R), before + exp)
        eng.usage_add(plane.data_ptr(), -1)
        assert np.array_equal(eng.read_usage().reshape(-1, R), before)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(80))
def test_split_protocol_world1_equals_walk(oracle, seed):
    """SplitTAS with one rank already takes the certificate / contended-walk route: it must equal the plain walk."""
    topo, rq = _case(seed, n_workloads=60)
    order = _order(seed + 7, rq.n_workloads, ["identity", "perm"][seed % 2])
    res = oracle.tas_find(topo, rq)
    want_adm, want_usage = oracle.tas_admit(topo, rq, res, order)
    eng = kqe.EmuTas()
    try:
        eng.put(topo)
        sp = SplitTAS(eng, topo, None, 0, 1)
        merged, adm = sp.cycle(rq, order)
        assert not res.equal(merged)
        assert np.array_equal(adm, want_adm)
        assert np.array_equal(eng.read_usage(), want_usage)
    finally:
        eng.close()


def test_requests_subset_roundtrip():
    topo, rq = _case(11, n_workloads=30)
    idx = np.array([i for i in range(rq.n_workloads) if i % 3 != 1][::-1])
    sub = rq.subset(idx)
    R = len(topo.resources)
    for j, w in enumerate(idx):
        a, b = rq.arrays["wl_off"][w], rq.arrays["wl_off"][w + 1]
        c, d = sub.arrays["wl_off"][j], sub.arrays["wl_off"][j + 1]
        assert b - a == d - c
        for k in ("count", "level", "kind", "slice_size", "slice_level", "group"):
            assert np.array_equal(rq.arrays[k][a:b], sub.arrays[k][c:d])
        assert np.array_equal(rq.arrays["single_pod_requests"].reshape(-1, R)[a:b], sub.arrays["single_pod_requests"].reshape(-1, R)[c:d])

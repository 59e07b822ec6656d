"""handleElasticWorkload (pkg/cache/scheduler/tas_elastic_workloads.go:37-165; gate ElasticJobsViaWorkloadSlicesWithTAS): the elastic cases of
the reference's TestFindTopologyAssignments (tests/golden/tas_elastic.yaml, extracted by tests/golden/extract_tas_elastic.py) through the
oracle, the emulated engine and the HIP engine (kq_tas_find_elastic), then random elastic batches engine == oracle."""
import random

import numpy as np
import pytest

from kueue_amd import _ffi as F
from kueue_amd import tas as T
from tests.conftest import load_golden

CASES = load_golden("tas_elastic.yaml")["cases"]


def build(case):
    nodes = [T.Node(n["name"], n["labels"], n["allocatable"], ready=n.get("ready", False), unschedulable=n.get("unschedulable", False)) for n in case["nodes"]]
    res = {r for ps in case["podSets"] for r in ps["requests"]}
    topo = T.Topology(case["levels"], nodes, resources=sorted(res))
    podsets = []
    for ps in case["podSets"]:
        tr = None
        if "topologyRequest" in ps:
            t = ps["topologyRequest"]
            tr = T.TopologyRequest(required=t.get("required"), preferred=t.get("preferred"), unconstrained=t.get("unconstrained", False),
                                   slice_required_topology=t.get("sliceRequiredTopology"), slice_size=t.get("sliceSize"))
        prev = [(d["values"], d["count"]) for d in ps["previousAssignment"]["domains"]] if "previousAssignment" in ps else None
        podsets.append(T.TASPodSetRequests(ps["name"], ps["count"], dict(ps["requests"]), tr, group=ps.get("group"), previous=prev))
    return topo, T.Requests(topo, [podsets])


def check(case, out, topo):
    for i, ps in enumerate(case["podSets"]):
        if "wantAssignment" in ps:
            assert int(out.a["status"][i]) == T.TAS_OK, (ps["name"], out.message(i))
            got = [(topo.leaf_values(leaf), cnt) for leaf, cnt in out.assignment(i)]
            assert got == [(d["values"], d["count"]) for d in ps["wantAssignment"]["domains"]], (ps["name"], got)
        elif "wantReason" in ps:
            assert int(out.a["status"][i]) != T.TAS_OK
            assert out.message(i) == ps["wantReason"], (out.message(i), ps["wantReason"])


def test_all_eight_reference_cases_are_present():
    assert len(CASES) == 8


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:80])
def test_elastic_oracle(oracle, case):
    topo, rq = build(case)
    check(case, oracle.tas_find_elastic(topo, rq), topo)


def run(eng, topo, rq):
    try:
        eng.put(topo)
        return eng.find_elastic(rq)
    finally:
        eng.close()


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:80])
def test_elastic_engine_emulated(oracle, case):
    from tests.emu import kqe
    topo, rq = build(case)
    out = run(kqe.EmuTas(), topo, rq)
    check(case, out, topo)
    want = oracle.tas_find_elastic(topo, rq)
    assert not want.equal(out), want.equal(out)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:80])
def test_elastic_engine_gpu(oracle, case):
    topo, rq = build(case)
    out = run(T.TASEngine(), topo, rq)
    check(case, out, topo)
    want = oracle.tas_find_elastic(topo, rq)
    assert not want.equal(out), want.equal(out)


# ---- random elastic batches: scale-up / scale-down / same count / stale previous assignments, with and without a leader, next to ordinary
# workloads of the same batch -----------------------------------------------------------------------------------------------------------
def random_batch(seed, n_wl=24):
    rnd = random.Random(seed)
    racks, hosts = rnd.randint(1, 3), rnd.randint(2, 5)
    nodes = [T.Node(f"r{r}-h{h}", {"rack": f"r{r}", T.HOSTNAME_LABEL: f"r{r}-h{h}"}, {"cpu": rnd.choice([2000, 4000, 8000]), "pods": rnd.choice([4, 10])}, ready=True)
             for r in range(racks) for h in range(hosts)]
    topo = T.Topology(["rack", T.HOSTNAME_LABEL], nodes, resources=["cpu"])
    names = [n.labels for n in nodes]
    wls = []
    for w in range(n_wl):
        kind = rnd.choice(["required", "preferred", "unconstrained"])
        lvl = rnd.choice(["rack", T.HOSTNAME_LABEL])
        tr = T.TopologyRequest(required=lvl) if kind == "required" else (T.TopologyRequest(preferred=lvl) if kind == "preferred" else T.TopologyRequest(unconstrained=True))
        count = rnd.randint(1, 6)
        cpu = rnd.choice([500, 1000, 2000])

        def prev_of(n_pods):
            out, left = [], n_pods
            hs = rnd.sample(range(len(nodes)), min(len(nodes), rnd.randint(1, 3)))
            for k, h in enumerate(sorted(hs)):
                if left <= 0:
                    break
                c = left if k == len(hs) - 1 else rnd.randint(1, left)
                vals = [names[h]["rack"], names[h][T.HOSTNAME_LABEL]]
                if rnd.random() < 0.06:
                    vals = ["gone", "gone-h0"]       # a domain the snapshot no longer holds: stale -> fresh placement
                out.append((vals, c)); left -= c
            return out
        elastic = rnd.random() < 0.7
        prev = prev_of(rnd.randint(1, 8)) if elastic else None
        group = None
        ps = []
        if rnd.random() < 0.4:
            group = f"g{w}"
            lprev = prev_of(1) if elastic and rnd.random() < 0.6 else None
            ps.append(T.TASPodSetRequests("leader", 1, {"cpu": rnd.choice([500, 1000])}, tr, group=group, previous=lprev))
        ps.append(T.TASPodSetRequests("workers", count, {"cpu": cpu}, tr, group=group, previous=prev))
        if rnd.random() < 0.5:
            ps.reverse()
        wls.append(ps)
    return topo, T.Requests(topo, wls)


def test_elastic_random_emulated(oracle):
    from tests.emu import kqe
    kinds = np.zeros(4, np.int64)
    for seed in range(120):
        topo, rq = random_batch(seed)
        want = oracle.tas_find_elastic(topo, rq)
        eng = kqe.EmuTas()
        try:
            eng.put(topo)
            rc = eng.find_elastic(rq, check=False)
            if isinstance(rc, int):
                assert rc == F.KQ_EUNSUPPORTED, rc      # (a scale-up smaller than its leader podset)
                kinds[3] += 1
                continue
            out = rc
        finally:
            eng.close()
        assert not want.equal(out), (seed, want.equal(out))
        kinds[0] += int((want.a["status"] == T.TAS_OK).sum()); kinds[1] += int((want.a["status"] != T.TAS_OK).sum())
    assert kinds[0] > 1000 and kinds[1] > 100, kinds


@pytest.mark.gpu
def test_elastic_random_gpu(oracle):
    for seed in range(60):
        topo, rq = random_batch(seed)
        want = oracle.tas_find_elastic(topo, rq)
        eng = T.TASEngine()
        try:
            eng.put(topo)
            try:
                out = eng.find_elastic(rq)
            except RuntimeError as x:
                assert "EUNSUPPORTED" in str(x)
                continue
        finally:
            eng.close()
        assert not want.equal(out), (seed, want.equal(out))

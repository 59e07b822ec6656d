"""features.TASRespectNodeAffinityPreferred (alpha, off by default): the 11 rows of TestFindTopologyAssignments that run with the gate on
(tests/golden/tas_find_affinity.yaml, transcribed by tests/golden/extract_tas.py) against the oracle's restatement — affinityScore on the
leaves (scheduling_simulator_default.go:110-115), summed up the tree (tas_flavor_snapshot.go:1976), first key of sortedDomains :1776 /
second of sortedDomainsWithLeader :1741, topAffinityTierDomains :1450 in front of every best-fit scan, the required request's second look
:1381-1390. The library refuses the gate (KQ_TAS_F_AFFINITY_PREFERRED -> KQ_EUNSUPPORTED, tests/test_tas_replacement.py): oracle only."""
import os

import numpy as np
import pytest
import yaml

from kueue_amd import tas as T
from tests.test_oracle_tas import build, check

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "tas_find_affinity.yaml")) as fh:
    CASES = yaml.safe_load(fh)["cases"]


def _match(labels, t):   # corev1 NodeSelectorRequirement (the operators the table uses)
    v = labels.get(t["key"])
    op = t["operator"]
    if op == "In":
        return v is not None and v in t["values"]
    if op == "NotIn":
        return v is None or v not in t["values"]
    if op == "Exists":
        return v is not None
    if op == "DoesNotExist":
        return v is None
    raise AssertionError(op)


def affinity_inputs(case, topo, rq):
    """(leaf_ok with the required node affinity folded in, leaf_score) per podset request — what the scheduling simulator decides per node
    (nodeaffinity.NewNodeSelector / NewPreferredSchedulingTerms(...).Score)."""
    nodes = {n["labels"].get(T.HOSTNAME_LABEL, n["name"]): n for n in case["nodes"]}
    score = np.zeros((rq.n, topo.n_leaves), np.int64)
    ok = np.ones((rq.n, topo.n_leaves), np.uint8)
    any_required = False
    for i, ps in enumerate(case["podSets"]):
        aff = ps.get("nodeAffinity") or {}
        for l in range(topo.n_leaves):
            if not topo.lowest_is_node:
                continue
            lab = nodes[topo.leaf_values(l)[-1]]["labels"]
            if aff.get("required"):
                any_required = True
                ok[i, l] = 1 if any(_match(lab, t) for t in aff["required"]) else 0   # NodeSelectorTerms are ORed
            score[i, l] = sum(t["weight"] for t in aff.get("preferred") or [] if _match(lab, t))
    return (ok if any_required else None), score


def test_all_gated_rows_transcribed():
    assert len(CASES) == 11


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:80])
def test_find_topology_assignments_affinity_preferred(oracle, case):
    topo, rq = build(case)
    ok, score = affinity_inputs(case, topo, rq)
    if ok is not None:
        prev = rq.arrays.get("leaf_ok")
        rq.arrays["leaf_ok"] = ok.reshape(-1) if prev is None else (np.asarray(prev, np.uint8).reshape(ok.shape) & ok).reshape(-1)
        rq._struct = None
    out = oracle.tas_find(topo, rq, leaf_score=score)
    check(case, out, topo)


def test_scores_change_answers(oracle):
    """(the gate is not a no-op on these rows: without the scores some of them get another assignment)"""
    differ = 0
    for case in CASES:
        topo, rq = build(case)
        try:
            check(case, oracle.tas_find(topo, rq), topo)
        except AssertionError:
            differ += 1
    assert differ >= 3, differ

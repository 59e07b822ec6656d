"""Oracle vs golden vectors: quota math, usage bubbling with lending limits, DRS.

Vectors transcribed from pkg/cache/scheduler/{resource_node,snapshot,fair_sharing}_test.go
(tests/golden/quota_drs.yaml carries file:line per case)."""
import pytest

from kueue_amd.api import amount_from_quantity
from kueue_amd.fixtures import load_case
from tests.conftest import load_golden

G = load_golden("quota_drs.yaml")


@pytest.mark.parametrize("case", G["lendable"], ids=lambda c: c["name"])
def test_cohort_lendable(oracle, case):
    cfg, snap, _ = load_case(case)
    oracle.derive(snap)
    got = oracle.lendable(cfg, snap, case["node"])
    assert got == case["want"]


@pytest.mark.parametrize("case", G["drs"], ids=lambda c: c["name"])
def test_dominant_resource_share(oracle, case):
    cfg, snap, _ = load_case(case)
    oracle.derive(snap)
    wl_req = None
    if case.get("wlReq"):
        wl_req = {}
        for k, q in case["wlReq"].items():
            f, r = k.split("/", 1)
            wl_req[(f, r)] = amount_from_quantity(r, q)
    for node, want in case["want"].items():
        got = oracle.drs(cfg, snap, node, wl_req)
        assert got["rounded"] == want["value"], (node, got)
        assert got["dominant"] == want["name"], (node, got)
        assert got["borrowing"] == want["borrowing"], (node, got)


def test_snapshot_add_remove_workload_with_lending_limit(oracle):
    case = G["lending_limit_usage"]
    cfg, snap, _ = load_case(case)
    oracle.derive(snap)
    fr = snap.fr("default", "cpu")
    for node, want in case["want_initial"].items():
        assert snap.plane("usage")[snap.node(node), fr] == want, node
    for node, want in case["want_subtree"].items():
        assert snap.plane("subtree_quota")[snap.node(node), fr] == want, node
    for step in case["steps"]:
        usage = oracle.apply_ops(cfg, snap, [tuple(o) for o in step["ops"]])
        for node, want in step["want"].items():
            assert usage[snap.node(node), fr] == want, (step["name"], node)

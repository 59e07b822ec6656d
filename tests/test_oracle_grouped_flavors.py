"""The oracle's GROUPED flavor assignment (kqo_set_grouped) against the leader-worker-set rows of TestAssignFlavors and
TestAssignFlavors_LeaderWorkerSetTASFlavor (tests/golden/assign_flavors_groups_manual.yaml; flavorassigner.go:782-860, :917-945).

The engine — and the oracle's default path the parity suites compare it with — scans flavors per podset. The two agree whenever a group's
members end on the same flavors either way; they differ when
the SUM of a group's requests changes the scan's outcome, or when a member requests none of the group's resources and inherits its TAS flavor.
The `ungrouped` blocks pin what the per-podset scan yields on the reference's own rows; DESIGN §7 states the gap."""
import numpy as np
import pytest

from kueue_amd.tas_cycle import load_tas_case
from tests.conftest import load_golden

CASES = load_golden("assign_flavors_groups_manual.yaml")["cases"]


def _groups(case, heads):
    ids, out = {}, []
    for ps in case["pending"][0]["podsets"]:
        g = ps.get("group")
        out.append(-1 if g is None else ids.setdefault(g, len(ids)))
    assert len(out) == int(heads.arrays["ps_off"][1])
    return np.array(out, np.int32)


def _flavors(got):
    return [{r: [v[0], v[1], v[2]] for r, v in ps.items()} for ps in got["podsets"]]


@pytest.fixture
def grouped(oracle):
    yield oracle
    oracle.set_grouped(False)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:60])
def test_grouped_flavor_assignment(grouped, case):
    oracle = grouped
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    want = case["want"]
    oracle.set_grouped(True, _groups(case, heads))
    got = oracle.assign_tas(cfg, snap, heads, ct if case.get("topologies") else None, 0)
    assert _flavors(got) == [{r: list(v) for r, v in ps["flavors"].items()} for ps in want["podsets"]], got
    if "repMode" in want:
        assert got["rep_mode"] == want["repMode"]
    if "usage" in want:
        assert got["usage"] == {tuple(k.split("/", 1)): v for k, v in want["usage"].items()}
    if "err" in want:
        assert got["err"] == want["err"], got
    for ps, wps in zip(got["reasons"], want["podsets"]):
        if "status" in wps:
            assert ps == sorted(wps["status"])
    if "noFitReason" in want:
        label, _, _ = oracle.assign_attempts(cfg, snap, heads, 0, stub={})
        assert label == want["noFitReason"]
    if "ungrouped" in case:   # what the per-podset scan (the engine's, and the oracle's default) yields on the same row
        oracle.set_grouped(False)
        got = oracle.assign_tas(cfg, snap, heads, None, 0)
        assert got["rep_mode"] == case["ungrouped"]["repMode"]
        assert _flavors(got) == [{r: list(v) for r, v in ps["flavors"].items()} for ps in case["ungrouped"]["podsets"]], got


def test_grouping_is_a_no_op_without_groups(grouped):
    """Singleton groups: the grouped path is assignFlavors itself — same decisions and records on random cycles."""
    from tests.randgen import random_case
    oracle = grouped
    for seed in range(80):
        cfg, snap, heads = random_case(seed, fair=seed % 3 == 0)[:3]
        oracle.derive(snap)
        oracle.set_grouped(False)
        a = oracle.cycle_run(cfg, snap, heads, rsn_cap=4096)
        oracle.set_grouped(True, np.full(int(heads.arrays["ps_off"][-1]), -1, np.int32))
        b = oracle.cycle_run(cfg, snap, heads, rsn_cap=4096)
        for k in a.a:
            assert np.array_equal(a.a[k], b.a[k]), (seed, k)


def _decision_diff(a, b, heads, nR):
    po = heads.arrays["ps_off"]
    out = []
    for i in range(heads.n):
        fa, fb = (d.a["flavor"][po[i] * nR:po[i + 1] * nR] for d in (a, b))
        if any(a.a[k][i] != b.a[k][i] for k in ("status", "action", "mode", "nominated_mode")) or \
                (a.a["nominated_mode"][i] != 0 and not np.array_equal(fa, fb)):   # the flavors of a NoFit assignment are never applied
            out.append(i)
    return out


@pytest.mark.parametrize("block", range(4))
def test_grouped_scan_risk_lists_every_head_that_can_differ(grouped, block):
    """kueue_amd.tas_cycle.grouped_scan_risk (shim/go GroupedScanRisk) is SOUND: on random TAS cycles with podset groups, every head whose decision
    under the reference's grouped scan differs from the per-podset scan's (the engine's) is listed; a caller that needs the reference's exact
    answer for leader-worker-set workloads can tell beforehand which cycles to keep on the reference's path."""
    from kueue_amd.tas_cycle import grouped_scan_risk
    from tests.tasgen_cycle import random_tas_cycle_case
    oracle = grouped
    n_multi = n_diff = n_flagged = 0
    for seed in range(block * 100, block * 100 + 100):
        cfg, snap, heads, ct, _ = random_tas_cycle_case(seed, fair=seed % 5 == 4, tight=seed % 3 == 0, preemption=True)
        oracle.derive(snap)
        oracle.set_grouped(False)
        a, _ = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))
        oracle.set_grouped(True)
        b, _ = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))
        oracle.set_grouped(False)
        if a.tas_stats["unsupported"] or b.tas_stats["unsupported"]:
            continue
        g, po = ct.arrays["ps_group"], heads.arrays["ps_off"]
        n_multi += sum(1 for i in range(heads.n) if any(g[p] >= 0 and (g[po[i]:po[i + 1]] == g[p]).sum() > 1 for p in range(po[i], po[i + 1])))
        risk = set(grouped_scan_risk(snap, heads, g))
        diff = _decision_diff(a, b, heads, snap.n_resource)
        assert set(diff) <= risk, (seed, diff, risk)
        n_diff += len(diff); n_flagged += len(risk)
    assert n_multi >= 10 and n_flagged <= n_multi

"""assignFlavors over PodSetGroupName groups (flavorassigner.go:782-860, resolvePodSetFlavors :917-945): the podsets of one group are ONE flavor scan
over the sum of their requests. The leader-worker-set rows of TestAssignFlavors and TestAssignFlavors_LeaderWorkerSetTASFlavor
(tests/golden/assign_flavors_groups_manual.yaml) on the oracle — whose only assignFlavors is the grouped one — and through the engine's device code
(CPU suite: the 1-lane emulation; -m gpu: the HIP engine), plus random TAS cycles with multi-podset groups, engine == oracle.

`ungrouped` blocks: the same row with the group names taken away (each podset a group of one) — what a per-podset scan yields; the reference's third
row is even named "without group it would fit"."""
import copy

import numpy as np
import pytest

from kueue_amd.tas_cycle import load_tas_case
from tests.conftest import load_golden

CASES = load_golden("assign_flavors_groups_manual.yaml")["cases"]


def _flavors(got):
    return [{r: [v[0], v[1], v[2]] for r, v in ps.items()} for ps in got["podsets"]]


def _want_flavors(block):
    return [{r: list(v) for r, v in ps["flavors"].items()} for ps in block["podsets"]]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:60])
def test_grouped_flavor_assignment(oracle, case):
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    want = case["want"]
    got = oracle.assign_tas(cfg, snap, heads, ct if case.get("topologies") else None, 0)
    assert _flavors(got) == _want_flavors(want), got
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
    if "ungrouped" in case:
        c2 = copy.deepcopy(case)
        for ps in c2["pending"][0]["podsets"]:
            ps.pop("group", None)
        cfg, snap, heads, ct = load_tas_case(c2)
        assert "ps_group" not in heads.arrays
        oracle.derive(snap)
        got = oracle.assign_tas(cfg, snap, heads, None, 0)
        assert got["rep_mode"] == case["ungrouped"]["repMode"]
        assert _flavors(got) == _want_flavors(case["ungrouped"]), got


MODES = {"NoFit": 0, "Preempt": 1, "Fit": 3}


def _engine_row(oracle, make, case):
    """The same rows as whole cycles through the device code: the head's flavors / modes / bookmarks per podset as the reference's table says, and
    every decision array equal to the oracle's."""
    from tests.test_tas_cycle_engine import _same
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    want = case["want"]
    eng = make(cfg)
    try:
        eng.put(snap)
        try:
            if case.get("topologies"):
                got, _ = eng.run_tas(heads, ct, tgt_cap=max(16, snap.n_adm))
            else:
                got = eng.run(heads)
        except Exception as ex:   # (the HIP engine raises on a refused cycle, the emulation returns the code)
            assert getattr(ex, "code", None) == -4, ex
            got = type("Refused", (), {"rc": -4, "error": str(ex)})()
    finally:
        eng.close()
    if getattr(got, "rc", 0) == -4:
        # KQ_EUNSUPPORTED, the one documented refusal among these rows: podsets of ONE workload placed on TWO TAS flavors (include/kq_cycle_tas.h;
        # the oracle row above pins the reference's answer). Nothing else may be refused.
        tas = {f["name"] for f in case.get("resourceFlavors", []) if f.get("topologyName")}
        used = {v[0] for ps in want["podsets"] for v in ps["flavors"].values()} & tas
        assert len(used) > 1 and not any(want.get("err", [])), (case["name"], getattr(got, "error", ""))
        return
    assert getattr(got, "rc", 0) == 0, (getattr(got, "rc", 0), getattr(got, "error", ""))
    nR = snap.n_resource
    if not want.get("err"):   # (Status.err rows: the cycle reports NoFit, the flavors of such an assignment are never applied)
        for p, wps in enumerate(want["podsets"]):
            row = {}
            for r in range(nR):
                f = int(got.a["flavor"][p * nR + r])
                if f >= 0:
                    row[snap.resources[r]] = [snap.flavors[f], [k for k, v in MODES.items() if v == int(got.a["res_mode"][p * nR + r])][0], int(got.a["tried_idx"][p * nR + r])]
            assert row == {r: list(v) for r, v in wps["flavors"].items()}, (case["name"], p, row)
    if "repMode" in want and not case.get("topologies"):
        assert int(got.a["nominated_mode"][0]) == MODES[want["repMode"]]
    if case.get("topologies"):
        assert _same(oracle, make, cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm)) is not None
    else:
        ref = oracle.cycle_run(cfg, snap, heads, rsn_cap=256)
        eng = make(cfg)
        try:
            eng.put(snap)
            got = eng.run(heads, rsn_cap=256)
        finally:
            eng.close()
        bad = ref.equal(got)
        assert not bad, bad


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:60])
def test_grouped_rows_emulated(oracle, case):
    from tests.test_tas_cycle_engine import _emu
    _engine_row(oracle, _emu, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"][:60])
def test_grouped_rows_gpu(oracle, case):
    from tests.test_tas_cycle_engine import _hip
    _engine_row(oracle, _hip, case)

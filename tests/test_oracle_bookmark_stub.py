"""The one row of TestFlavorScanRecordsLastTriedFlavorIdx (flavorassigner_test.go:6268) that needs the table's STUB preemption oracle:
"needs preemption and candidates exist: bookmark names the first flavor" (:6369-6388). Its ClusterQueue carries usage without workloads behind
it, so the real SimulatePreemption — the one the engine and the whole-cycle oracle run — answers NoCandidates, which is the NEXT row of the
table (:6389, transcribed in tests/golden/schedule_tas_manual.yaml and run through the engine). With kqo_assign_tas (Assign with its TAS half
and a stubbed oracle) the row is pinned on the restatement itself: same snapshot as the next row, testOracle answering (Preempt, 0) for both
flavors, WhenCanPreempt = MayStopSearch -> the scan stops on flavor-1 and the bookmark names it."""
import copy

from kueue_amd.tas_cycle import load_tas_case
from tests.conftest import load_golden

POSS = {"NoCandidates": 0, "Preempt": 1, "Reclaim": 2}


def _row(name):
    return copy.deepcopy(next(c for c in load_golden("schedule_tas_manual.yaml")["cases"] if c["name"].startswith(name)))


def _tried(oracle, case, stub_poss):
    cfg, snap, heads, ct = load_tas_case(case)
    oracle.derive(snap)
    stub = {snap.fr(f, "cpu"): (POSS[stub_poss], 0) for f in ("flavor-1", "flavor-2")}
    got = oracle.assign_tas(cfg, snap, heads, ct, 0, stub=stub)
    return got["rep_mode"], got["podsets"][0]["cpu"]


def test_candidates_exist_bookmark_names_the_first_flavor(oracle):
    case = _row("needs preemption but no candidates")   # the same nominalPerFlavor / cohortSpare / request / usage / fungibility (:6370-6385 == :6390-6407)
    assert case["clusterQueues"][0]["fungibility"] == {"whenCanBorrow": "MayStopSearch", "whenCanPreempt": "MayStopSearch", "preference": "PreemptionOverBorrowing"}
    mode, (flavor, res_mode, tried) = _tried(oracle, case, "Preempt")
    assert (mode, flavor, res_mode, tried) == ("Preempt", "flavor-1", "Preempt", 0)   # wantMode Preempt, wantTriedFlavorIdx 0 (:6386-6387)


def test_no_candidates_through_the_stub_equals_the_transcribed_row(oracle):
    """The next row through the same entry point: the stub answering NoCandidates gives what the real oracle gives on that snapshot."""
    case = _row("needs preemption but no candidates")
    mode, (_, res_mode, tried) = _tried(oracle, case, "NoCandidates")
    assert (mode, res_mode, tried) == ("Preempt", "Preempt", case["expectAssignment"]["triedIdx"]["cpu"]) and tried == -1

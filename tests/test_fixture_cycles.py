"""tests/fixture_cycles.py on the 1-lane emulation (CPU suite) and on the HIP engine (GPU suite)."""
import pytest

from tests import fixture_cycles as FC
from tests.conftest import load_golden

PRE = load_golden("preemption.yaml")["cases"] + load_golden("preemption_manual.yaml")["cases"]
FAIR = load_golden("preemption_fair.yaml")["cases"] + load_golden("preemption_fair_manual.yaml")["cases"]   # + TestFairPreemptionSkipsUnsatisfiableTournament
ASG = load_golden("assign_flavors.yaml")["cases"] + load_golden("assign_flavors_hierarchical.yaml")["cases"] + load_golden("assign_flavors_reclaim.yaml")["cases"]

# how many cases the bridge must carry all the way to the Go expectation (measured; a drop means the bridge or the engine regressed)
MIN_BRIDGED = {"preemption": 40, "fair": 34, "assign": 28}


def _emu():
    from tests.emu import kqe
    return kqe.EmuEngine


def _hip():
    from kueue_amd.engine import Engine
    return Engine


def _all(oracle, factory):
    n = {"preemption": 0, "fair": 0, "assign": 0}
    for c in PRE:
        n["preemption"] += 1 if FC.preemption_case(oracle, factory, c)[0] else 0
    for c in FAIR:
        n["fair"] += 1 if FC.preemption_case(oracle, factory, c)[0] else 0
    for c in ASG:
        n["assign"] += 1 if FC.assign_case(oracle, factory, c) else 0
    return n


def test_fixture_cycles_emulated(oracle):
    n = _all(oracle, _emu())
    print("bridged to the Go expectation:", n, "of", len(PRE), len(FAIR), len(ASG))
    for k, v in MIN_BRIDGED.items():
        assert n[k] >= v, (k, n)


@pytest.mark.gpu
def test_fixture_cycles_gpu(oracle):
    n = _all(oracle, _hip())
    for k, v in MIN_BRIDGED.items():
        assert n[k] >= v, (k, n)

"""The AdmissionFairSharing restatement (oracle/afs_ref.py) against the reference's own unit-test tables (tests/golden/afs.yaml), and
the host-side helpers of the boundary (kueue_amd/afs.py) against the restatement."""
import math

import numpy as np
import pytest

from kueue_amd import afs as H
from oracle import afs_ref as R
from tests.conftest import load_golden

G = load_golden("afs.yaml")


def _rl(d):
    return {k: R.Q.parse(str(v)) for k, v in d.items()}


@pytest.mark.parametrize("c", G["alpha"], ids=lambda c: f"{c['sampling_s']}/{c['half_life_s']}")
def test_alpha(c):
    assert R.alpha_rate(c["sampling_s"], c["half_life_s"]) == c["want"]
    assert H.alpha_rate(c["sampling_s"], c["half_life_s"]) == c["want"]


@pytest.mark.parametrize("c", G["mul_by_float"], ids=lambda c: c["name"])
def test_mul_by_float(c):
    got = R.mul_by_float(_rl(c["rl"]), c["f"])
    want = _rl(c["want"])
    assert set(got) == set(want)
    for k in want:
        assert got[k].fraction() == want[k].fraction(), k          # Quantity.Cmp == 0
        assert got[k].dec and got[k].scale == R.MUL_BY_FLOAT_SCALE   # TestMulByFloatBoundsScale
        # the host helper of the boundary computes the same amount from nano integers
        nano = int(_rl(c["rl"])[k].fraction() * H.NANO)
        assert H.mul_by_float(nano, c["f"]) == got[k].value


def test_mul_by_float_nil_list():
    assert R.mul_by_float(None, 0.5) is None


@pytest.mark.parametrize("c", G["decays_to_zero"], ids=lambda c: c["name"])
def test_mul_by_float_decays_to_zero(c):
    f = c["factor"] if "factor" in c else 1 - c["one_minus"]
    rl = {"cpu": R.Q.parse(c["start"])}
    nano = int(rl["cpu"].fraction() * H.NANO)
    for _ in range(200000):
        rl = R.mul_by_float(rl, f)
        nano = H.mul_by_float(nano, f)
        assert nano == rl["cpu"].value
        if rl["cpu"].is_zero():
            return
    pytest.fail(f"residual usage {rl}")


@pytest.mark.parametrize("c", G["quantity_to_float"], ids=lambda c: c["q"])
def test_quantity_to_float(c):
    assert R.Q.parse(str(c["q"])).approx_f64() == pytest.approx(c["want"], rel=0, abs=1e-9)
    if c["q"] in ("5", "5.5", "5k", "5.5k", "1E", "1Ei", "8Pi"):
        assert R.Q.parse(str(c["q"])).approx_f64() == c["want"]


@pytest.mark.parametrize("c", G["usage"], ids=lambda c: c["name"])
def test_calculate_usage(c):
    got = R.calculate_usage(_rl(c["consumed"]), _rl(c["penalty"]), float(c["lq_weight"]), c["res_weights"])
    assert got == c["want"] and not math.isnan(got)


def test_decayed_consumed_converges():
    c = G["decay_converges"]
    usage = _rl(c["usage"])
    consumed = {}
    tol = R.Q.parse(c["tolerance"]).fraction()
    prev = 0
    for want in c["want_after"]:
        for _ in range(c["samples_per_half_life"]):
            consumed = R.decayed_consumed(consumed, usage, c["elapsed_s"], c["half_life_s"])
            assert consumed["cpu"].fraction() > prev      # TestCalculateDecayedConsumedAccumulatesSubMilli
            prev = consumed["cpu"].fraction()
        got = consumed["cpu"].fraction()
        assert abs(got - R.Q.parse(str(want)).fraction()) <= tol
        assert got <= usage["cpu"].fraction()
    # the boundary's helper, on nano integers
    h = {}
    for _ in range(50):
        h = H.decayed_consumed(h, {"cpu": 2 * H.NANO}, c["elapsed_s"], c["half_life_s"])
    r = {}
    for _ in range(50):
        r = R.decayed_consumed(r, usage, c["elapsed_s"], c["half_life_s"])
    assert h["cpu"] == r["cpu"].value


@pytest.mark.parametrize("size", ["16Gi", "1Ti", "64Ti"])
def test_decayed_consumed_keeps_large_quantities_positive(size):
    got = R.decayed_consumed({}, {"memory": R.Q.parse(size)}, 300, 168 * 3600)
    assert got["memory"].value > 0


@pytest.mark.parametrize("c", G["ledger"], ids=lambda c: c["name"])
def test_penalty_bookkeeping(c):
    led = R.Ledger()
    for op in c["ops"]:
        if "push" in op:
            led.push_penalty("ns/lq", op["wl"], _rl(op["push"]))
        else:
            led.sub_penalty("ns/lq", op["wl"])
    assert led.has_pending_penalty("ns/lq") == c["want_has"]
    for name, want in c.get("want_milli", {}).items():
        q = led.peek_penalty("ns/lq").get(name)
        assert (q.milli_value() if q is not None else 0) == want


def test_sub_penalty_on_absent_localqueue_creates_nothing():
    led = R.Ledger()
    assert led.sub_penalty("ns/lq", "ns/wl1") is None
    assert "ns/lq" not in led.entries


def test_entry_penalty_keeps_every_key_and_is_positive():
    """TestCalculateEntryPenaltyWithDRAResources / ...WithLongHalfLife."""
    p = R.entry_penalty(_rl({"cpu": "4", "gpu-logical": "2"}), 300, 600)
    assert set(p) == {"cpu", "gpu-logical"} and all(q.value > 0 for q in p.values())
    p = R.entry_penalty(_rl({"cpu": "2", "nvidia.com/gpu": "1"}), 300, 168 * 3600)
    assert all(q.value > 0 for q in p.values())
    assert H.entry_penalty({"cpu": 2 * H.NANO, "nvidia.com/gpu": H.NANO}, H.alpha_rate(300, 168 * 3600)) == {k: q.value for k, q in p.items()}


def test_split128_round_trip():
    rnd = np.random.default_rng(3)
    for v in [0, 1, -1, (1 << 127) - 1, -(1 << 127), 1 << 64, -(1 << 64), 24190234349953124739] + [int(x) * int(y) for x, y in rnd.integers(-2**62, 2**62, (50, 2))]:
        lo, hi = H.split128(v)
        assert 0 <= lo < 1 << 64 and -(1 << 63) <= hi < 1 << 63
        got = H.join128(lo, hi)
        assert (got if got < (1 << 127) else got - (1 << 128)) == v or got == v

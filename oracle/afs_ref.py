"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's AdmissionFairSharing arithmetic and ledger, on Python integers.
Nothing in kueue_amd/ may import this module; tests/ compares the device ledger (kueue_amd/csrc/kq_pending.hpp DAfs) with it.

Restated (paths under /root/reference):
  vendor/k8s.io/apimachinery/pkg/api/resource/quantity.go   ParseQuantity (fast path), Add :~690, Neg, IsZero, AsDec, AsApproximateFloat64 :468-483
  vendor/k8s.io/apimachinery/pkg/api/resource/amount.go     int64Amount.Add (the sum takes the finer of the two scales)
  vendor/gopkg.in/inf.v0/dec.go                             Dec.Add (scale = max of the scales), Round(..., RoundDown)
  pkg/util/resource/resource.go                             mergeResourceList :30-45, MergeResourceListKeepSum :79-84, MulByFloat :100-115
  pkg/util/admissionfairsharing/admission_fair_sharing.go   calculateAlphaRate :45, CalculateEntryPenalty :53, CalculateUsage :86, CalculateDecayedConsumed :110
  pkg/cache/queue/afs/usage_ledger.go                       withPenalty :78-90, WithoutPenalty :96-120
  pkg/cache/queue/afs/entry_penalties.go                    PushPenalty :30, SubPenalty :45, HasPendingPenalty :67
  pkg/controller/core/workload_controller.go                settlement :1506-1528

Pinned against the reference's own unit-test tables (tests/golden/afs.yaml, transcribed from resource_test.go, admission_fair_sharing_test.go,
entry_penalties_test.go, scheduler_afs_test.go): tests/test_afs_ref.py.

A Quantity is kept in the two forms apimachinery has, because AsApproximateFloat64 depends on the form:
  int64Amount  value * 10^scale            (what ParseQuantity returns for everything the tests parse)
  infDecAmount unscaled * 10^-dscale       (what MulByFloat returns, and what Add yields once one operand is in this form)"""
import math
from decimal import Decimal
from fractions import Fraction

_SUFFIX10 = {"n": -9, "u": -6, "m": -3, "": 0, "k": 3, "M": 6, "G": 9, "T": 12, "P": 15, "E": 18}
_SUFFIX2 = {"Ki": 10, "Mi": 20, "Gi": 30, "Ti": 40, "Pi": 50, "Ei": 60}
MUL_BY_FLOAT_SCALE = 9   # resource.go:93


def pow10(n: int) -> float:
    """math.Pow10 (Go): table product for n >= 0, 1 / table for n < 0 — for |n| < 32 one table entry, so a correctly rounded 10^n."""
    assert -31 <= n <= 31
    return float(10 ** n) if n >= 0 else 1.0 / float(10 ** -n)


class Q:
    __slots__ = ("dec", "value", "scale")

    def __init__(self, dec: bool, value: int, scale: int):
        # dec False: value * 10^scale (int64Amount)   dec True: value * 10^-scale (inf.Dec: unscaled, scale)
        self.dec, self.value, self.scale = dec, int(value), int(scale)

    @staticmethod
    def parse(s: str) -> "Q":
        s = s.strip()
        suf = ""
        for k in sorted(list(_SUFFIX2) + [k for k in _SUFFIX10 if k], key=len, reverse=True):
            if s.endswith(k):
                suf, s = k, s[: -len(k)]
                break
        neg = s.startswith("-")
        s = s.lstrip("+-")
        num, _, den = s.partition(".")
        den = den.rstrip("0") if suf in _SUFFIX2 else den
        mant = int((num or "0") + den)
        if neg:
            mant = -mant
        if suf in _SUFFIX2:      # binary suffixes are multiplied out, scale 0 (quantity.go fast path, base 2)
            v = Fraction(mant, 10 ** len(den)) * 2 ** _SUFFIX2[suf]
            assert v.denominator == 1
            return Q(False, int(v), 0)
        return Q(False, mant, _SUFFIX10[suf] - len(den))

    @staticmethod
    def nano(v: int) -> "Q":
        return Q(True, v, 9)

    def copy(self):
        return Q(self.dec, self.value, self.scale)

    def as_dec(self):
        """(unscaled, scale) of AsDec(): inf.NewDec(value, -scale) for the int64 form."""
        return (self.value, self.scale) if self.dec else (self.value, -self.scale)

    def fraction(self) -> Fraction:
        u, s = self.as_dec()
        return Fraction(u) * Fraction(10) ** (-s)

    def is_zero(self):
        return self.value == 0

    def neg(self):
        return Q(self.dec, -self.value, self.scale)

    def add(self, y: "Q") -> "Q":
        """Quantity.Add."""
        if not self.dec and not y.dec:
            # int64Amount.Add (amount.go): zero operands keep the other's scale, otherwise the finer scale wins (no overflow here:
            # the test amounts are far inside int64)
            if y.value == 0:
                return self.copy()
            if self.value == 0:
                return y.copy()
            sc = min(self.scale, y.scale)
            return Q(False, self.value * 10 ** (self.scale - sc) + y.value * 10 ** (y.scale - sc), sc)
        (a, sa), (b, sb) = self.as_dec(), y.as_dec()
        sc = max(sa, sb)             # inf.Dec.Add
        return Q(True, a * 10 ** (sc - sa) + b * 10 ** (sc - sb), sc)

    def approx_f64(self) -> float:
        """AsApproximateFloat64 :468-483."""
        base = float(self.value)     # big.Float.SetInt(..).Float64() / float64(int64): nearest, ties to even — CPython's int -> float too
        exponent = -self.scale if self.dec else self.scale
        if exponent == 0:
            return base
        return base * pow10(exponent)

    def milli_value(self) -> int:
        """MilliValue: ceil(value * 1000) (ScaledValue rounds up)."""
        return math.ceil(self.fraction() * 1000)

    def __repr__(self):
        return f"Q({'dec' if self.dec else 'int'} {self.value}e{-self.scale if self.dec else self.scale})"


def mul_by_float(rl, f: float):
    """resource.MulByFloat :100-115."""
    if rl is None:
        return None
    sign, digits, exp = Decimal(repr(float(f))).as_tuple()   # strconv.FormatFloat(f, 'f', -1, 64): the shortest round-trip digits
    m = int("".join(map(str, digits))) * (-1 if sign else 1)
    fs = -exp                                                  # factor = m * 10^-fs  (fs may be negative)
    out = {}
    for k, v in rl.items():
        u, s = v.as_dec()
        prod, ps = u * m, s + fs                               # inf.Dec.Mul
        if ps > MUL_BY_FLOAT_SCALE:                            # Round(.., 9, RoundDown): toward zero
            q = abs(prod) // 10 ** (ps - MUL_BY_FLOAT_SCALE)
            prod = -q if prod < 0 else q
        else:
            prod *= 10 ** (MUL_BY_FLOAT_SCALE - ps)
        out[k] = Q(True, prod, MUL_BY_FLOAT_SCALE)
    return out


def merge_keep_sum(a, b):
    """MergeResourceListKeepSum :79-84 over mergeResourceList :30-45."""
    if a is None:
        return {k: v.copy() for k, v in (b or {}).items()}
    ret = {k: v.copy() for k, v in a.items()}
    for k, vb in (b or {}).items():
        ret[k] = ret[k].add(vb) if k in ret else vb.copy()
    return ret


def alpha_rate(sampling_s: float, half_life_s: float) -> float:
    if half_life_s == 0:
        return 0.0
    return 1.0 - math.pow(0.5, sampling_s / half_life_s)


def entry_penalty(total_requests, sampling_s, half_life_s):
    return mul_by_float(total_requests, alpha_rate(sampling_s, half_life_s))


def decayed_consumed(old, new, elapsed_s, half_life_s):
    a = alpha_rate(elapsed_s, half_life_s)
    return merge_keep_sum(mul_by_float(old, 1 - a), mul_by_float(new, a))


def calculate_usage(consumed, penalty, lq_weight: float, res_weights) -> float:
    """afs.CalculateUsage :86-103."""
    allr = merge_keep_sum(consumed, penalty)
    usage = 0.0
    for name in sorted(allr):
        w = (res_weights or {}).get(name, 1.0)
        usage += w * allr[name].approx_f64()
    if lq_weight <= 0:
        return math.inf
    return usage / lq_weight


class Entry:
    """UsageLedgerEntry: Resources, pendingPenalty, penaltyRecords."""

    def __init__(self):
        self.resources = {}
        self.pending = None      # nil until the first push (mergeResourceList(nil, b) copies b)
        self.records = {}

    def without_penalty(self, wl):
        rec = self.records.pop(wl, None)
        if rec is None:
            return None
        agg = merge_keep_sum(self.pending, {k: v.neg() for k, v in rec.items()})
        self.pending = {k: v for k, v in agg.items() if not v.is_zero()}
        return rec

    def with_penalty(self, wl, penalty):
        self.without_penalty(wl)
        self.records[wl] = {k: v.copy() for k, v in penalty.items()}
        self.pending = merge_keep_sum(self.pending, penalty)


class Ledger:
    """AfsUsageLedger."""

    def __init__(self):
        self.entries = {}

    def push_penalty(self, lq, wl, penalty):
        self.entries.setdefault(lq, Entry()).with_penalty(wl, penalty)

    def sub_penalty(self, lq, wl):
        e = self.entries.get(lq)
        return e.without_penalty(wl) if e else None

    def has_pending_penalty(self, lq):
        e = self.entries.get(lq)
        return bool(e and e.pending and any(not q.is_zero() for q in e.pending.values()))

    def peek_penalty(self, lq):
        e = self.entries.get(lq)
        return (e.pending or {}) if e else {}

    def set_consumed(self, lq, resources, settle_wl=None):
        """A controller's whole-entry rewrite of Resources; with settle_wl the settlement of workload_controller.go:1519-1526."""
        e = self.entries.setdefault(lq, Entry())
        pen = e.without_penalty(settle_wl) if settle_wl is not None else None
        e.resources = merge_keep_sum(resources, pen) if pen is not None else {k: v.copy() for k, v in resources.items()}

    def usage(self, lq, lq_weight, res_weights):
        e = self.entries.get(lq)
        return calculate_usage(e.resources if e else {}, (e.pending or {}) if e else {}, lq_weight, res_weights)

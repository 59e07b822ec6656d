"""AdmissionFairSharing on the host side of the boundary: what the Go shim computes with apimachinery's resource.Quantity before it
calls kq_pending_afs_put / kq_pending_afs_set_consumed (include/kq_engine.h), restated on Python integers.

Reference (paths under /root/reference/pkg):
  util/admissionfairsharing/admission_fair_sharing.go  calculateAlphaRate :45   CalculateEntryPenalty :53   CalculateUsage :86
                                                        CalculateDecayedConsumed :110
  util/resource/resource.go                             MulByFloat :100-115 (decimal product, rounded toward zero at scale 9)
  cache/queue/afs/usage_ledger.go, entry_penalties.go   the ledger the device mirrors (kueue_amd/csrc/kq_pending.hpp DAfs)
  scheduler/scheduler.go                                updateEntryPenalty :1337-1355 (SumTotalRequests workload.go:564)

Amounts are integers in units of 1e-9 of the resource's base unit ("nano"): resource.Quantity cannot hold anything finer."""
import math
from decimal import Decimal
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from kueue_amd import _ffi as F

NANO = 10 ** 9
_M64 = (1 << 64) - 1


def calculate_usage(consumed: Dict[str, float], penalty: Optional[Dict[str, float]] = None, lq_weight: float = 1.0,
                    res_weights: Optional[Dict[str, float]] = None) -> float:
    """afs.CalculateUsage :86-103 on floats, for callers that keep the ledger on the host and hand the number over with
    kq_pending_set_lq_usage. consumed / penalty: resource name -> Quantity.AsApproximateFloat64 (cores, bytes, counts)."""
    allr = dict(consumed)
    for k, v in (penalty or {}).items():
        allr[k] = allr.get(k, 0.0) + v
    usage = 0.0
    for name in sorted(allr):
        w = (res_weights or {}).get(name, 1.0)
        usage += w * allr[name]
    if lq_weight <= 0:
        return math.inf
    return usage / lq_weight


def alpha_rate(sampling_s: float, half_life_s: float) -> float:
    """calculateAlphaRate :45-51."""
    if half_life_s == 0:
        return 0.0
    return 1.0 - math.pow(0.5, sampling_s / half_life_s)


def mul_by_float(amount_nano: int, f: float) -> int:
    """resource.MulByFloat on one amount: the exact decimal product of the amount and the SHORTEST decimal that round-trips f
    (strconv.FormatFloat(f, 'f', -1, 64)), rounded toward zero at scale 9."""
    if not math.isfinite(f):
        raise ValueError("MulByFloat called with a non-finite factor")
    sign, digits, exp = Decimal(repr(float(f))).as_tuple()
    m = int("".join(map(str, digits)))
    if sign:
        m = -m
    prod = amount_nano * m            # at scale 9 - exp
    if exp >= 0:
        return prod * 10 ** exp
    q = abs(prod) // 10 ** (-exp)     # inf.RoundDown: toward zero
    return -q if prod < 0 else q


def entry_penalty(total_requests_nano: Dict[str, int], alpha: float) -> Dict[str, int]:
    """afs.CalculateEntryPenalty(totalRequests, config) :53-60: every key of the list stays, scaled by alpha."""
    return {k: mul_by_float(v, alpha) for k, v in total_requests_nano.items()}


def decayed_consumed(old_nano: Dict[str, int], usage_nano: Dict[str, int], elapsed_s: float, half_life_s: float) -> Dict[str, int]:
    """afs.CalculateDecayedConsumed :110-118: old * (1 - alpha) + usage * alpha, alpha from the elapsed time."""
    a = alpha_rate(elapsed_s, half_life_s)
    out = {k: mul_by_float(v, 1 - a) for k, v in old_nano.items()}
    for k, v in usage_nano.items():
        out[k] = out.get(k, 0) + mul_by_float(v, a)
    return out


def split128(v: int):
    """(lo, hi) two's complement words of a 128-bit integer."""
    v &= (1 << 128) - 1
    lo, hi = v & _M64, v >> 64
    return lo, hi - (1 << 64) if hi >> 63 else hi


def join128(lo: int, hi: int) -> int:
    return (int(hi) << 64) | (int(lo) & _M64)


def _planes(rows: Sequence[Sequence[int]], n_res: int):
    lo = np.zeros((len(rows), n_res), np.uint64); hi = np.zeros((len(rows), n_res), np.int64)
    for i, row in enumerate(rows):
        for r, v in enumerate(row):
            a, b = split128(int(v))
            lo[i, r] = a; hi[i, r] = b
    return np.ascontiguousarray(lo.reshape(-1)), np.ascontiguousarray(hi.reshape(-1))


class Ledger:
    """kq_afs_ledger: the AfsUsageLedger as it stands when the pending set is uploaded, plus what PushPenalty would record for every
    pending workload. `resources` is sorted by name (CalculateUsage's summation order)."""

    def __init__(self, resources: Iterable[str], lq_weight: Sequence[float], res_weight: Optional[Dict[str, float]] = None):
        self.resources = sorted(resources)
        if len(self.resources) > 64:
            raise ValueError("at most 64 ledger resources")
        self.index = {n: i for i, n in enumerate(self.resources)}
        self.n_res = len(self.resources)
        self.n_lq = len(lq_weight)
        self.lq_weight = np.ascontiguousarray(lq_weight, np.float64)
        rw = res_weight or {}
        self.res_weight = np.array([rw.get(n, 1.0) for n in self.resources], np.float64)
        self.consumed = [[0] * self.n_res for _ in range(self.n_lq)]
        self.consumed_f64 = None         # [n_lq][n_res] floats when the ledger holds amounts that are not in the scale-9 form
        self.penalty = [[0] * self.n_res for _ in range(self.n_lq)]
        self.present = np.zeros((self.n_lq, self.n_res), np.uint8)

    def set_consumed(self, lq: int, amounts_nano: Dict[str, int], f64: Optional[Dict[str, float]] = None):
        for k, v in amounts_nano.items():
            self.consumed[lq][self.index[k]] = int(v)
        if f64 is not None:
            if self.consumed_f64 is None:
                self.consumed_f64 = [[None] * self.n_res for _ in range(self.n_lq)]
            for k, v in f64.items():
                self.consumed_f64[lq][self.index[k]] = float(v)

    def set_pending_penalty(self, lq: int, amounts_nano: Dict[str, int]):
        for k, v in amounts_nano.items():
            self.penalty[lq][self.index[k]] = int(v)
            self.present[lq, self.index[k]] = 1

    def workload_columns(self, penalties: Sequence[Dict[str, int]]):
        """(lo, hi, mask) of per-workload penalties (entry_penalty of every pending workload)."""
        lo, hi = _planes([[p.get(n, 0) for n in self.resources] for p in penalties], self.n_res)
        mask = np.array([sum(1 << self.index[k] for k in p) for p in penalties], np.uint64)
        return lo, hi, np.ascontiguousarray(mask)

    def consumed_columns(self, rows: Sequence[Sequence[int]], f64_rows=None):
        lo, hi = _planes(rows, self.n_res)
        f = None
        if f64_rows is not None:
            f = np.array([[approx_f64_nano(v) if x is None else x for v, x in zip(row, fr)] for row, fr in zip(rows, f64_rows)], np.float64).reshape(-1)
            f = np.ascontiguousarray(f)
        return lo, hi, f

    def struct(self, penalties: Sequence[Dict[str, int]]):
        a = {}
        a["lq_weight"], a["res_weight"] = self.lq_weight, self.res_weight
        a["consumed_lo"], a["consumed_hi"], f = self.consumed_columns(self.consumed, self.consumed_f64)
        if f is not None:
            a["consumed_f64"] = f
        if self.present.any():
            a["penalty_lo"], a["penalty_hi"] = _planes(self.penalty, self.n_res)
            a["penalty_present"] = np.ascontiguousarray(self.present.reshape(-1))
        a["wl_penalty_lo"], a["wl_penalty_hi"], a["wl_penalty_mask"] = self.workload_columns(penalties)
        s = F.kq_afs_ledger()
        F.fill_struct(s, a, dict(n_lq=self.n_lq, n_res=self.n_res))
        s._keep = a
        return s


def approx_f64_nano(v: int) -> float:
    """Quantity.AsApproximateFloat64 of an infDec amount at scale 9 (quantity.go:468-483): the unscaled integer rounded to the nearest
    double (big.Float.SetInt(...).Float64()), times math.Pow10(-9) = 1 / 1e9."""
    return float(int(v)) * 1e-9   # int -> float is correctly rounded (ties to even) in CPython

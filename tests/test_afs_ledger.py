"""The AdmissionFairSharing ledger on the device (include/kq_engine.h kq_pending_afs_*, kueue_amd/csrc/kq_pending.hpp DAfs): entry
penalties pushed by kq_pending_apply when a workload is assumed (scheduler.go:1064-1068, :1337-1355), SubPenalty, settlement, and the
LocalQueue usage Heads() orders by — against the restatement of the reference's ledger (oracle/afs_ref.py, pinned by
tests/test_afs_ref.py) and the reference's own TestScheduleForAFS cases (tests/golden/afs.yaml)."""
import copy

import numpy as np
import pytest

from kueue_amd import _ffi as F
from kueue_amd import afs as H
from kueue_amd.api import (ClusterQueue, Decisions, FlavorQuotas, Heads, Pending, PodSet, ResourceGroup, ResourceQuota, Snapshot, Workload,
                           make_config)
from oracle import afs_ref as R
from tests.conftest import load_golden

G = load_golden("afs.yaml")
GI = 1 << 30


def _emu(cfg):
    from tests.emu import kqe
    return kqe.EmuEngine(cfg)


def _hip(cfg):
    from kueue_amd.engine import Engine
    return Engine(cfg)


# ---- TestScheduleForAFS ---------------------------------------------------------------------------------------------------------

def _schedule_case(oracle, case, eng_factory):
    S = G["schedule"]
    alpha = H.alpha_rate(S["sampling_s"], S["half_life_s"])
    lq_names = sorted(S["lq_weight"])
    lq_of = {n: i for i, n in enumerate(lq_names)}
    fq = FlavorQuotas("default", {"cpu": ResourceQuota(int(R.Q.parse(S["quota"]["cpu"]).fraction() * 1000)),
                                  "memory": ResourceQuota(int(R.Q.parse(S["quota"]["memory"]).fraction()))})
    cq = ClusterQueue("cq1", resource_groups=[ResourceGroup([fq])], queueing_strategy="BestEffortFIFO")
    snap = Snapshot([cq], [], [], now_ns=1)
    snap.derive()
    wls, pens, tot = [], [], []
    for i, w in enumerate(case["workloads"]):
        milli = int(R.Q.parse(w["cpu"]).fraction() * 1000)
        x = Workload(w["name"], "cq1", creation_ts=1 + w["ts"], pod_sets=[PodSet("one", count=1, requests={"cpu": milli})], uid=f"{i:04d}")
        x.scheduling_hash = 1000 + milli        # equal pod set shapes hash equal
        wls.append(x)
        tot.append({"cpu": R.Q(False, milli, -3)})            # SumTotalRequests: cpu as a milli quantity (requests.go ToResourceList)
        pens.append(H.entry_penalty({"cpu": milli * 10 ** 6}, alpha))
    names = [w["name"] for w in case["workloads"]]
    lq = np.array([lq_of[w["lq"]] for w in case["workloads"]], np.int32)
    heads = Heads(snap, wls, cycle=0)
    afs_on = case["afs"]
    pending = Pending(heads, uid_rank=np.arange(len(wls), dtype=np.uint32), lq=lq if afs_on else None, n_lq=len(lq_names) if afs_on else 0)
    cfg = make_config()
    eng = eng_factory(cfg); q = oracle.PendingOracle(cfg, snap, pending)
    ref = R.Ledger()
    weights = [1.0 if n == case.get("deleted") else float(S["lq_weight"][n]) for n in lq_names]
    led = H.Ledger(["cpu"], weights)
    for n, rl in case["initial"].items():
        qv = R.Q.parse(rl["cpu"])
        ref.set_consumed(n, {"cpu": qv})                                   # SetForTest: the parsed quantity, int64 form
        led.set_consumed(lq_of[n], {"cpu": int(qv.fraction() * H.NANO)}, f64={"cpu": qv.approx_f64()})
    admitted, attempted = [], []
    try:
        eng.put(snap); eng.pending_put(pending)
        if afs_on:
            eng.pending_afs_put(led, pens)
        osnap = copy.copy(snap); osnap.arrays = dict(snap.arrays)
        for cyc in range(1, len(wls) + 1):
            if afs_on:
                usage = np.array([ref.usage(n, weights[i], {}) for i, n in enumerate(lq_names)])
                q.set_lq_usage(usage)
                assert np.array_equal(eng.pending_afs_read(len(wls))["usage"], usage), cyc
            n, nps, hw = eng.pending_heads(cyc)
            hb, ohw = q.heads(cyc)
            assert np.array_equal(hw, ohw), (cyc, hw, ohw)
            if n == 0:
                eng.pending_apply()
                continue
            got = eng.run_pending(Decisions(hb, tgt_cap=64))
            want = oracle.cycle_run(cfg, osnap, hb)
            assert not want.equal(got)
            usage_plane, na, triples = oracle.cycle_commit(cfg, osnap, hb)
            osnap.arrays["usage"] = usage_plane; osnap._struct = None
            assert eng.try_commit() == 0
            eng.pending_apply(); q.apply(hb, want)
            w = int(hw[0])
            if want.a["status"][0] == F.ST_ASSUMED:
                admitted.append(names[w])
                if afs_on:   # shouldApplyEntryPenalty (scheduler.go:1318-1335) -> PushPenalty
                    ref.push_penalty(case["workloads"][w]["lq"], names[w], R.entry_penalty(tot[w], S["sampling_s"], S["half_life_s"]))
            else:
                attempted.append(names[w])
            assert np.array_equal(eng.pending_state()[0], q.state()), cyc
        assert sorted(admitted) == sorted(case["admitted"])
        assert sorted(set(attempted)) == sorted(case["attempted"])
        if afs_on:   # the ledger itself
            got = eng.pending_afs_read(len(wls))
            for i, n in enumerate(lq_names):
                pen = ref.peek_penalty(n).get("cpu")
                assert got["penalty"][i] == (pen.value if pen is not None else 0)
                assert bool(got["present"][i]) == (pen is not None)
            assert [names[i] for i in range(len(wls)) if got["record"][i]] == [n for n in names if n in admitted]
    finally:
        eng.close(); q.close()


@pytest.mark.parametrize("case", G["schedule"]["cases"], ids=lambda c: c["name"])
def test_reference_afs_schedule_cases_emulated(oracle, case):
    _schedule_case(oracle, case, _emu)


@pytest.mark.gpu
@pytest.mark.parametrize("case", G["schedule"]["cases"], ids=lambda c: c["name"])
def test_reference_afs_schedule_cases_gpu(oracle, case):
    _schedule_case(oracle, case, _hip)


# ---- random ledgers ---------------------------------------------------------------------------------------------------------------

RES = ["cpu", "example.com/gpu", "memory", "pods"]      # sorted by name


def _random_quantity(rnd, res, parsed):
    """A consumed amount: parsed (int64 form, as a persisted status or SetForTest gives it) or decayed (scale 9)."""
    if parsed:
        if res == "memory":
            return R.Q.parse(str(rnd.choice(["0", "1Gi", "16Gi", "2Ti", "64Ti", "1500Mi", "123456789"])))
        return R.Q.parse(str(rnd.choice(["0", "1", "8", "100m", "2500m", "5.5", "33", "7k", "1n", "999999999n"])))
    top = 10 ** int(rnd.integers(1, 24))
    return R.Q.nano(int(rnd.integers(0, 2 ** 62)) % top)


def _ledger_loop(oracle, eng_factory, seed, n_cq=6, per=10, cycles=14):
    rnd = np.random.default_rng(seed)
    n_lq_per = 3
    fqs = [FlavorQuotas("f0", {"cpu": ResourceQuota(10 ** 9), "memory": ResourceQuota(10 ** 15), "example.com/gpu": ResourceQuota(10 ** 6),
                               "pods": ResourceQuota(10 ** 6)})]
    cqs = [ClusterQueue(f"cq{c}", resource_groups=[ResourceGroup(copy.deepcopy(fqs))], queueing_strategy="BestEffortFIFO") for c in range(n_cq)]
    snap = Snapshot(cqs, [], [], now_ns=1)
    snap.derive()
    wls, lq, tot = [], [], []
    for c in range(n_cq):
        for j in range(per):
            req = {"cpu": int(rnd.integers(1, 64)) * 250}
            if rnd.random() < 0.7:
                req["memory"] = int(rnd.integers(1, 512)) * (GI // 4)
            if rnd.random() < 0.3:
                req["example.com/gpu"] = int(rnd.integers(0, 9))      # a zero request keeps its key (a zero penalty moves the sum to scale 9)
            x = Workload(f"w{c}-{j}", f"cq{c}", priority=int(rnd.integers(0, 3)), creation_ts=1 + len(wls), pod_sets=[PodSet("one", count=1, requests=req)],
                         uid=f"{len(wls):05d}")
            wls.append(x)
            lq.append(-1 if c == n_cq - 1 else c * n_lq_per + int(rnd.integers(0, n_lq_per)))   # the last ClusterQueue has no AdmissionScope
            tot.append({k: (R.Q(False, v, -3) if k == "cpu" else R.Q(False, v, 0)) for k, v in req.items()})
    n_lq = n_cq * n_lq_per
    lq = np.array(lq, np.int32)
    W = len(wls)
    half = float(rnd.choice([0.0, 10.0, 600.0, 168 * 3600.0]))
    sampling = float(rnd.choice([1.0, 300.0]))
    pen_ref = [R.entry_penalty(t, sampling, half) for t in tot]
    pens = [{k: q.value for k, q in p.items()} for p in pen_ref]
    assert pens == [H.entry_penalty({k: int(q.fraction() * H.NANO) for k, q in t.items()}, H.alpha_rate(sampling, half)) for t in tot]
    weights = [float(x) for x in rnd.choice([1.0, 1.0, 2.0, 0.5, 0.0, 3.0], size=n_lq)]
    res_w = {"cpu": 1.0, "memory": float(rnd.choice([1.0, 2.0 ** -30, 1e-9])), "pods": 0.0} if rnd.random() < 0.7 else {}
    names = [f"lq{i}" for i in range(n_lq)]
    ref = R.Ledger()
    led = H.Ledger(RES, weights, res_w)
    for i in range(n_lq):
        if rnd.random() < 0.2:
            continue                                               # no entry yet
        rl, f64 = {}, {}
        for r in RES:
            if rnd.random() < 0.6:
                rl[r] = _random_quantity(rnd, r, parsed=rnd.random() < 0.5)
                f64[r] = rl[r].approx_f64()
        ref.set_consumed(names[i], rl)
        led.set_consumed(i, {k: int(v.fraction() * H.NANO) for k, v in rl.items()}, f64=f64)
    heads = Heads(snap, wls, cycle=0)
    pending = Pending(heads, uid_rank=np.arange(W, dtype=np.uint32), lq=lq, n_lq=n_lq)
    cfg = make_config()
    eng = eng_factory(cfg); q = oracle.PendingOracle(cfg, snap, pending)
    pushed = set()

    def check(tag):
        got = eng.pending_afs_read(W)
        usage = np.array([ref.usage(names[i], weights[i], res_w) for i in range(n_lq)])
        assert np.array_equal(got["usage"], usage, equal_nan=True), (tag, got["usage"], usage)
        for i in range(n_lq):
            e = ref.entries.get(names[i])
            for r, rn in enumerate(RES):
                pen = (e.pending or {}).get(rn) if e else None
                assert got["penalty"][i * len(RES) + r] == (pen.value * 10 ** (9 - pen.scale) if pen is not None else 0), (tag, i, rn)
                assert bool(got["present"][i * len(RES) + r]) == (pen is not None), (tag, i, rn)
                con = e.resources.get(rn) if e else None
                assert got["consumed"][i * len(RES) + r] == (int(con.fraction() * H.NANO) if con is not None else 0), (tag, i, rn)
        assert set(np.nonzero(got["record"])[0].tolist()) == pushed, tag
        return usage

    try:
        eng.put(snap); eng.pending_put(pending); eng.pending_afs_put(led, pens)
        osnap = copy.copy(snap); osnap.arrays = dict(snap.arrays)
        for cyc in range(1, cycles + 1):
            q.set_lq_usage(check(("cycle", cyc)))
            n, nps, hw = eng.pending_heads(cyc)
            hb, ohw = q.heads(cyc)
            assert np.array_equal(hw, ohw), cyc
            if n == 0:
                eng.pending_apply()
                break
            got = eng.run_pending(Decisions(hb, tgt_cap=64))
            want = oracle.cycle_run(cfg, osnap, hb)
            assert not want.equal(got)
            usage_plane, na, triples = oracle.cycle_commit(cfg, osnap, hb)
            osnap.arrays["usage"] = usage_plane; osnap._struct = None
            assert eng.try_commit() == 0
            eng.pending_apply(); q.apply(hb, want)
            for i, w in enumerate(hw[hw >= 0]):
                if want.a["status"][i] == F.ST_ASSUMED and lq[w] >= 0:
                    ref.push_penalty(names[lq[w]], int(w), pen_ref[w]); pushed.add(int(w))
            check(("apply", cyc))
            # what the controllers do between cycles
            if pushed and rnd.random() < 0.6:      # rollback / deletion: SubPenalty (also of workloads without a record, and twice)
                pick = [int(x) for x in rnd.choice(sorted(pushed), size=min(len(pushed), 2), replace=False)] + [int(rnd.integers(0, W))]
                pick.append(pick[0])
                eng.pending_afs_sub_penalty(pick)
                for w in pick:
                    if lq[w] >= 0:
                        ref.sub_penalty(names[lq[w]], w)
                    pushed.discard(w)
                check(("sub", cyc))
            if rnd.random() < 0.7:                 # LocalQueue reconciler decay / settlement of an admitted workload
                lqs = [int(x) for x in rnd.choice(n_lq, size=int(rnd.integers(1, 4)), replace=False)]
                rows, f64s, settle = [], [], []
                for l in lqs:
                    e = ref.entries.get(names[l])
                    old = e.resources if e else {}
                    new = R.decayed_consumed(old, {"cpu": R.Q.parse("4"), "memory": R.Q.parse("8Gi")}, float(rnd.integers(1, 600)), half or 60.0)
                    mine = [w for w in sorted(pushed) if lq[w] == l]
                    sw = mine[0] if mine and rnd.random() < 0.7 else (int(rnd.integers(0, W)) if rnd.random() < 0.2 else -1)
                    if sw >= 0 and lq[sw] != l:
                        sw = -1
                    ref.set_consumed(names[l], new, settle_wl=sw if sw >= 0 else None)
                    pushed.discard(sw)
                    rows.append([new[r].value if r in new else 0 for r in RES])
                    f64s.append([new[r].approx_f64() if r in new else None for r in RES])
                    settle.append(sw)
                eng.pending_afs_set_consumed(lqs, rows, f64_rows=f64s if rnd.random() < 0.5 else None, settle_wl=settle)
                check(("set", cyc))
    finally:
        eng.close(); q.close()


@pytest.mark.parametrize("seed", range(40))
def test_ledger_random_loops_emulated(oracle, seed):
    _ledger_loop(oracle, _emu, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_ledger_random_loops_gpu(oracle, seed):
    _ledger_loop(oracle, _hip, 100 + seed, n_cq=24, per=12, cycles=16)


def test_ledger_entry_points_refuse_misuse():
    from tests.emu import kqe
    fq = FlavorQuotas("f0", {"cpu": ResourceQuota(10_000)})
    snap = Snapshot([ClusterQueue("cq", resource_groups=[ResourceGroup([fq])], queueing_strategy="BestEffortFIFO")], [], [], now_ns=1)
    snap.derive()
    wls = [Workload(f"w{i}", "cq", creation_ts=i + 1, pod_sets=[PodSet("one", count=1, requests={"cpu": 1000})], uid=f"{i:04d}") for i in range(3)]
    pending = Pending(Heads(snap, wls, cycle=0), uid_rank=np.arange(3, dtype=np.uint32), lq=np.zeros(3, np.int32), n_lq=1)
    eng = kqe.EmuEngine(make_config())
    try:
        eng.put(snap); eng.pending_put(pending)
        led = H.Ledger(["cpu"], [1.0])
        pens = [{"cpu": 5}] * 3
        with pytest.raises(AssertionError):
            eng.pending_afs_sub_penalty([0])                 # no ledger yet
        eng.pending_afs_put(led, pens)
        with pytest.raises(AssertionError):
            eng.pending_set_lq_usage([1.0])                  # the usage is the ledger's now
        with pytest.raises(AssertionError):
            eng.pending_afs_set_consumed([0, 0], [[1], [2]]) # a LocalQueue twice
        with pytest.raises(AssertionError):
            eng.pending_afs_sub_penalty([7])
        eng.pending_heads(1)
        with pytest.raises(AssertionError):
            eng.pending_afs_sub_penalty([0])                 # heads in flight
    finally:
        eng.close()

"""kq_pending_step / kq_pending_step_wait (include/kq_engine.h): the pending loop enqueued one cycle per call, the head count left on
the device, decisions fetched one or two steps later. Every step must equal the oracle's sequential loop — Heads(), every decision
field, targets, the queue states and the usage plane at the end — on populations without preemption, with classical preemption
(targets pool staged per step) and with fair sharing; with the host waiting right away (depth 1) and running ahead (depth 2)."""
import copy

import numpy as np
import pytest

from kueue_amd import _ffi as F
from kueue_amd.api import Decisions, make_config
from kueue_amd.population import generate

KINDS = {
    "cfg2": (dict(cfg=2), False),
    "cfg3-120cq": (dict(cfg=3, n_cq=120, per_cq=8), False),
    "cfg4c-60cq": (dict(cfg=4, n_cq=60, per_cq=5), False),
    "cfg4f-40cq": (dict(cfg=4, n_cq=40, per_cq=4, fair_sharing=True), True),
}


def _emu(cfg):
    from tests.emu import kqe
    return kqe.EmuEngine(cfg)


def _hip(cfg):
    from kueue_amd.engine import Engine
    return Engine(cfg)


def _loop(oracle, eng_factory, kind, depth, cycles, hold=2, rsn_cap=0):
    kw, fair = KINDS[kind]
    pop = generate(**kw)
    snap = pop.snapshot
    cfg = make_config(fair_sharing=fair)
    pending = pop.pending()
    eng = eng_factory(cfg); q = oracle.PendingOracle(cfg, snap, pending)
    tgt_cap = max(4096, (32 if fair else 4) * snap.n_adm)
    parent = snap.arrays["parent"]; root_of = np.arange(snap.N)
    for _ in range(8):
        root_of = np.where(parent[root_of] >= 0, parent[root_of], root_of)
    try:
        eng.put(snap); eng.pending_put(pending)
        mh, mps = eng.pending_bounds()
        assert mh == snap.n_cq and mps >= int(pop.w_nps.max())
        any_heads = pop.heads_for_cycle(0)
        outs = [Decisions(any_heads, tgt_cap=tgt_cap, rsn_cap=rsn_cap, n=mh, n_ps=mps) for _ in range(2)]
        if rsn_cap:
            eng.pending_step_reasons(rsn_cap)
        osnap = copy.copy(snap); osnap.arrays = dict(snap.arrays)
        held, live, issued, waited = [], 0, 0, 0
        elive = 0
        want_q = []     # oracle results of the cycles, in order

        def oracle_cycle(cyc):
            nonlocal live
            hb, ohw = q.heads(cyc)
            if hb.n == 0:
                want_q.append((hb, ohw, None, q.state().copy()))
                return
            want = oracle.cycle_run(cfg, osnap, hb, rsn_cap=rsn_cap)
            usage, na, triples = oracle.cycle_commit(cfg, osnap, hb)
            osnap.arrays["usage"] = usage; osnap._struct = None
            q.apply(hb, want)
            held.append(triples); live += 1
            if live > hold:
                live -= 1
                done = held.pop(0)
                osnap.arrays["usage"] = oracle.usage_apply(cfg, osnap, done, add=False); osnap._struct = None
                freed = np.unique(root_of[done[0]])
                if len(freed):
                    q.queue_inadmissible(np.nonzero(np.isin(root_of[:snap.n_cq], freed))[0])
            want_q.append((hb, ohw, want, q.state().copy()))

        def issue(cyc):
            nonlocal elive, issued
            elive += 1
            rel = 0
            if elive > hold:
                rel = hold + 1; elive -= 1
            eng.pending_step(cyc, tgt_cap, release_age=rel, want_heads=True)
            issued += 1

        def wait():
            nonlocal waited
            out = outs[waited % 2]
            n, nps, hw = eng.pending_step_wait(out, want_heads=True)
            hb, ohw, want, _ = want_q[waited]
            assert np.array_equal(hw, ohw), (kind, waited)
            assert n == hb.n and nps == hb.n_ps, (kind, waited, n, hb.n, nps, hb.n_ps)
            if want is not None:
                got = out.view(hb)
                bad = want.equal(got)
                assert not bad, (kind, depth, waited, bad)
            waited += 1

        for cyc in range(1, cycles + 1):
            oracle_cycle(cyc)
            if want_q[-1][2] is None:      # an empty Heads(): the engine's step must see the same (nothing to commit: no ring slot is used)
                pytest.skip("population drained before the loop ended") if cyc < 3 else None
            issue(cyc)
            while issued - waited >= depth:
                wait()
        while waited < issued:
            wait()
        assert np.array_equal(eng.pending_state()[0], want_q[-1][3]), kind
        assert np.array_equal(eng.read_usage(), osnap.arrays["usage"]), kind
    finally:
        eng.close(); q.close()


@pytest.mark.parametrize("kind", ["cfg3-120cq", "cfg4c-60cq", "cfg4f-40cq"])
def test_step_loop_with_reason_records_emulated(oracle, kind):
    """kq_pending_step_reasons: the steps stage their reason windows with the decisions; every record equals the oracle's."""
    _loop(oracle, _emu, kind, 2, 6, rsn_cap=1 << 16)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cfg3-120cq", "cfg4c-60cq", "cfg4f-40cq"])
def test_step_loop_with_reason_records_gpu(oracle, kind):
    _loop(oracle, _hip, kind, 2, 6, rsn_cap=1 << 16)


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("kind", list(KINDS))
def test_step_loop_emulated(oracle, kind, depth):
    _loop(oracle, _emu, kind, depth, cycles=8)


@pytest.mark.parametrize("kind", ["cfg3-120cq", "cfg4c-60cq"])
def test_step_loop_emulated_unfused_tail(oracle, kind, monkeypatch):
    """KQ_STEP_UNFUSED: the step's commit / requeue / release as the separate launches of the call-by-call entry points."""
    monkeypatch.setenv("KQ_STEP_UNFUSED", "1")
    _loop(oracle, _emu, kind, 2, cycles=8)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,kw", [("cfg3-120cq", dict(cycles=10)), ("cfg4c-60cq", dict(cycles=8)), ("cfg4f-40cq", dict(cycles=6)), ("cfg2", dict(cycles=12))])
def test_step_loop_gpu(oracle, kind, kw):
    _loop(oracle, _hip, kind, 2, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("env", ["KQ_STEP_SIDE_STREAMS", "KQ_STEP_UNFUSED"])
def test_step_loop_gpu_variants(oracle, env, monkeypatch):
    """The measured-and-rejected variants stay correct: side streams for the upload / copies / usage levels, the unfused tail."""
    monkeypatch.setenv(env, "1")
    _loop(oracle, _hip, "cfg4c-60cq", 2, cycles=8)
    _loop(oracle, _hip, "cfg3-120cq", 2, cycles=10)


def test_step_refuses_misuse(oracle):
    from tests.emu import kqe
    pop = generate(cfg=1)
    cfg = make_config()
    eng = kqe.EmuEngine(cfg)
    try:
        eng.put(pop.snapshot); eng.pending_put(pop.pending())
        with pytest.raises(AssertionError):
            eng.pending_step_wait(None)                      # nothing in flight
        eng.pending_step(1, 64); eng.pending_step(2, 64)
        with pytest.raises(AssertionError):
            eng.pending_step(3, 64)                          # two in flight
        with pytest.raises(AssertionError):
            eng.pending_heads(3)                             # the synchronous entry points are closed meanwhile
        eng.pending_step_wait(None); eng.pending_step_wait(None)
        n, nps, hw = eng.pending_heads(3)
        with pytest.raises(AssertionError):
            eng.pending_step(4, 64)                          # heads of kq_pending_heads in flight
    finally:
        eng.close()

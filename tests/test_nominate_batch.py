"""kq_nominate_run_resident — "nominate-all-pending" (SURVEY §8d batch mode, §8f-1 nominate-ahead): Scheduler.nominate
(scheduler.go:665-705) for every pending workload of a population in one launch, compared field by field (nomination fields,
targets, algorithmic bytes) with the oracle's nominate-only run. Emulation on the CPU suite, the HIP engine at full cfg 3 size
(100 000 heads) and on a preemption population on the GPU suite."""
import numpy as np
import pytest

from kueue_amd.api import Decisions, make_config
from kueue_amd.population import generate


def _nominate_all(oracle, eng_factory, cfgn, fair, n_cq, per_cq, limit=None):
    pop = generate(cfgn, n_cq=n_cq, per_cq=per_cq, fair_sharing=fair)
    cfg = make_config(fair_sharing=fair)
    heads = pop.all_heads() if limit is None else pop._heads(np.arange(0, pop.n_pending, max(1, pop.n_pending // limit)), 1)
    cap = max(4096, 8 * pop.snapshot.n_adm)
    want = oracle.nominate_run(cfg, pop.snapshot, heads, tgt_cap=cap)
    eng = eng_factory(cfg)
    try:
        eng.put(pop.snapshot)
        eng.heads_put(heads, 0)
        got = eng.nominate_resident(0, Decisions(heads, tgt_cap=cap))
        bad = want.equal(got)
        assert not bad, bad
        assert (got.a["status"] == 0).all() and (got.a["order"] == -1).all()
        assert eng.try_commit() == -1  # a nomination pass leaves nothing to commit
    finally:
        eng.close()
    return want


@pytest.mark.parametrize("cfgn,fair", [(3, False), (4, False), (4, True)])
def test_nominate_all_pending_emulated(oracle, cfgn, fair):
    from tests.emu import kqe
    w = _nominate_all(oracle, kqe.EmuEngine, cfgn, fair, n_cq=40, per_cq=6)
    assert cfgn == 4 or len(set(w.a["nominated_mode"].tolist())) > 1


@pytest.mark.gpu
def test_nominate_all_pending_cfg3_full(oracle):
    """The bench's cfg3-batch workload: 100 000 heads, one k_nominate launch."""
    from kueue_amd.engine import Engine
    _nominate_all(oracle, Engine, 3, False, n_cq=None, per_cq=None)


@pytest.mark.gpu
def test_nominate_all_pending_preemption(oracle):
    from kueue_amd.engine import Engine
    _nominate_all(oracle, Engine, 4, False, n_cq=200, per_cq=10, limit=600)

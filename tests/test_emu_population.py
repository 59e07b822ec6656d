"""CPU: engine device logic (test-only 1-lane emulation) vs oracle on reduced-size synthetic populations."""
import numpy as np
import pytest

from kueue_amd.api import make_config
from kueue_amd.population import generate
from tests.emu import kqe

# the emulation rotates LDS budgets (period 3) and the one-chunk-ahead prefetch (period 2) per launch: 6+ cycles cover
# every combination of the process kernel's code paths
CASES = [
    ("cfg1", dict(cfg=1), [0, 3]),
    ("cfg2", dict(cfg=2), [0, 9, 1, 2, 3, 4, 5]),
    ("cfg3-200cq", dict(cfg=3, n_cq=200, per_cq=8), [0, 5, 1, 2, 3, 4, 6]),
    ("cfg4c-60cq", dict(cfg=4, n_cq=60, per_cq=8), [0, 1, 3, 2, 4, 5, 6]),
]


@pytest.mark.parametrize("name,kw,cycles", CASES, ids=[c[0] for c in CASES])
def test_population_cycles(oracle, name, kw, cycles):
    pop = generate(**kw)
    cfg = make_config()
    eng = kqe.EmuEngine(cfg)
    try:
        eng.put(pop.snapshot)
        for c in cycles:
            heads = pop.heads_for_cycle(c, cycle=c + 1)
            want = oracle.cycle_run(cfg, pop.snapshot, heads, want_usage=True)
            got = eng.run(heads, want_usage=True, tgt_cap=max(4096, pop.snapshot.n_adm))
            assert got.rc == 0, got.error
            bad = want.equal(got)
            assert not bad, (name, c, bad)
            assert np.array_equal(want.usage_after, got.usage_after)
            assert got.bytes == want.stats["total"]
    finally:
        eng.close()

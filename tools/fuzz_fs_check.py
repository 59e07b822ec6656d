"""The self-checking emulation of the fair victim search over many random cycles (VERDICT r04 "next" 1: "g_fs_check clean on 16 k random
cycles"): every LDS-formulated search (kq_fs.hpp) runs a second time as the candidate-by-candidate walk and both are compared — targets,
reasons, algorithmic bytes, the private state on the preemptor's path — and the cycle is compared with the oracle.
usage: python tools/fuzz_fs_check.py LO HI          (prints the seeds that differ: none expected)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import kqo
from tests.emu import kqe
from tests.randgen import random_case
lo, hi = int(sys.argv[1]), int(sys.argv[2])
kqe.lib().kqe_fs_check(1)
out = (C.c_longlong * 32)()
kqe.lib().kqe_cstat(out)
bad, searches = [], 0
for seed in range(lo, hi):
    kw = dict(fair=True, preemption=True, tight=seed % 2 == 0)
    if seed % 3 == 0:
        kw.update(max_cq=10, fair_dups=True)
    if seed % 3 == 1:
        kw.update(partial=True)
    cfg, snap, heads = random_case(seed, **kw)
    kqo.derive(snap)
    want = kqo.cycle_run(cfg, snap, heads, want_usage=True)
    eng = kqe.EmuEngine(cfg)
    try:
        eng.put(snap)
        got = eng.run(heads, want_usage=True)
    finally:
        eng.close()
    kqe.lib().kqe_cstat(out)
    ok = got.rc == 0 and not want.equal(got) and got.bytes == want.stats["total"] and np.array_equal(want.usage_after, got.usage_after) and out[24] == 0
    searches += out[23]
    if not ok:
        bad.append(seed); print("DIFF seed", seed, flush=True)
print(f"fuzz_fs_check seeds {lo}..{hi - 1}: {hi - lo} cycles, {searches} LDS-formulated searches each re-run as the walk and compared, differing seeds: {bad}")

#!/usr/bin/env python
"""Expected decisions of the BASELINE.json configs[3] populations AT THEIR STATED SIZE, produced offline by the CPU oracle.

A full-size fair-sharing + preemption cycle (cfg4f: 1000 ClusterQueues, 40 k admitted workloads, ~400 victims per preemptor)
costs the oracle tens of minutes of one core, far beyond what a test may spend; so the oracle is run ONCE here and the
complete expected output (every kq_decisions array, the target CSR, the post-cycle usage plane digest and the algorithmic byte
count) is committed as a compressed .npz next to this script. tests/test_gpu_golden_population.py regenerates the same
seeded population (kueue_amd/population.py) on the GPU box and compares the HIP engine with these arrays bit for bit.

    python tests/golden/gen_population_golden.py cfg4c cfg4f cfg3f   # writes tests/golden/pop_<name>_c<cycle>.npz

The inputs are not stored: the population generator is deterministic in (cfg, seed); a digest of the flattened snapshot and
heads is stored and checked so that a drift of the generator is reported as such and not as a parity failure.
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (generate kwargs, fair sharing, cycles)
CASES = {
    "cfg4c": (dict(cfg=4), False, [0]),
    "cfg4f": (dict(cfg=4, fair_sharing=True), True, [0]),
    "cfg3f": (dict(cfg=3, fair_sharing=True), True, [0]),
}


def digest_inputs(snap, heads) -> str:
    h = hashlib.sha256()
    for k in sorted(snap.arrays):
        h.update(k.encode()); h.update(np.ascontiguousarray(snap.arrays[k]).tobytes())
    for k in sorted(heads.arrays):
        h.update(k.encode()); h.update(np.ascontiguousarray(heads.arrays[k]).tobytes())
    return h.hexdigest()


def path_of(name, cycle):
    return os.path.join(HERE, f"pop_{name}_c{cycle}.npz")


def main(names):
    from kueue_amd.api import make_config
    from kueue_amd.population import generate
    from oracle import kqo
    for name in names:
        kw, fair, cycles = CASES[name]
        pop = generate(**kw)
        cfg = make_config(fair_sharing=fair)
        for c in cycles:
            heads = pop.heads_for_cycle(c, cycle=c + 1)
            t0 = time.perf_counter()
            want = kqo.cycle_run(cfg, pop.snapshot, heads, want_usage=True)
            dt = time.perf_counter() - t0
            m = int(want.a["tgt_off"][-1])
            out = {k: v for k, v in want.a.items() if k not in ("tgt_adm", "tgt_reason")}
            out["tgt_adm"] = want.a["tgt_adm"][:m]
            out["tgt_reason"] = want.a["tgt_reason"][:m]
            out["usage_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(want.usage_after).tobytes()).digest(), np.uint8)
            out["bytes_total"] = np.array([want.stats["total"]], np.int64)
            out["inputs_sha256"] = np.frombuffer(bytes.fromhex(digest_inputs(pop.snapshot, heads)), np.uint8)
            out["oracle_seconds"] = np.array([dt])
            np.savez_compressed(path_of(name, c), **out)
            print(f"{name} cycle {c}: {heads.n} heads, {m} targets, oracle {dt:.1f} s -> {path_of(name, c)}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))

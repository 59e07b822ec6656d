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
    "cfg4c": (dict(cfg=4), False, [0, 1]),
    "cfg4f": (dict(cfg=4, fair_sharing=True), True, [0, 1]),
    "cfg3f": (dict(cfg=3, fair_sharing=True), True, [0]),
}


# version of the algorithmic-byte accounting: 2 = simulations whose results cannot be observed are booked as discarded (round 6,
# kq_oracle.cpp findFlavorForPodSets "dead simulations"); files without the key follow rule 1 and their byte total is not compared
ACCOUNTING = 2


def cycle_input(pop, name, c):
    """(snapshot, heads) of cycle c. Cycle 0: the population as generated. Cycle 1 (configs[3] at full size was pinned on
    one cycle only): the snapshot after kq_cycle_commit of cycle 0 — the usage of every workload the COMMITTED cycle-0 golden admitted
    is added to its ClusterQueue (flavor per resource from the golden's decision, quantity from the head's requests, the pods
    resource = the podset's count), cohort usage re-derived; rows stay (issued preemptions have not been carried out yet, and
    kq_cycle_commit does not append rows) — and the SECOND workload of every ClusterQueue as heads: 1000 other preemptors on the
    over-committed snapshot. The successor is built from the committed cycle-0 expectation, so the chain is pinned end to end."""
    import copy
    heads = pop.heads_for_cycle(c, cycle=c + 1)
    if c == 0:
        return pop.snapshot, heads
    assert c == 1
    g = np.load(path_of(name, 0))
    base = pop.snapshot
    h0 = pop.heads_for_cycle(0, cycle=1)
    ha = h0.arrays
    nR, nfr = base.n_resource, base.n_fr
    usage = base.arrays["usage"].reshape(base.N, nfr).copy()
    usage[base.n_cq:] = 0   # (cohort rows are re-derived below)
    pods = base.resource_index.get("pods", -1)
    for i in np.flatnonzero(g["action"] == 1):   # KQ_ACT_ADMIT
        cq = int(ha["cq"][i])
        for p in range(int(ha["ps_off"][i]), int(ha["ps_off"][i + 1])):
            req = {int(ha["req_res"][k]): int(ha["req_qty"][k]) for k in range(int(ha["ps_req_off"][p]), int(ha["ps_req_off"][p + 1]))}
            for r in range(nR):
                fl = int(g["flavor"][p * nR + r])
                if fl < 0:
                    continue
                q = int(g["ps_count"][p]) if r == pods else req.get(r, 0)
                usage[cq, fl * nR + r] += q
    snap = copy.copy(base)
    snap.arrays = dict(base.arrays)
    snap.arrays["usage"] = usage.reshape(-1)
    snap._struct = None
    snap.derived = False
    snap.derive()
    return snap, type(heads).from_arrays(snap, dict(heads.arrays), cycle=c + 1)


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
            if os.path.exists(path_of(name, c)) and not os.environ.get("KQ_GOLDEN_FORCE"):
                print(f"{name} cycle {c}: {path_of(name, c)} exists (KQ_GOLDEN_FORCE=1 rewrites it)", flush=True)
                continue
            snap, heads = cycle_input(pop, name, c)
            t0 = time.perf_counter()
            want = kqo.cycle_run(cfg, snap, heads, want_usage=True)
            dt = time.perf_counter() - t0
            m = int(want.a["tgt_off"][-1])
            out = {k: v for k, v in want.a.items() if k not in ("tgt_adm", "tgt_reason")}
            out["tgt_adm"] = want.a["tgt_adm"][:m]
            out["tgt_reason"] = want.a["tgt_reason"][:m]
            out["usage_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(want.usage_after).tobytes()).digest(), np.uint8)
            out["bytes_total"] = np.array([want.stats["total"]], np.int64)
            out["accounting"] = np.array([ACCOUNTING])   # which rule the byte total follows (the decisions do not depend on it)
            out["inputs_sha256"] = np.frombuffer(bytes.fromhex(digest_inputs(snap, heads)), np.uint8)
            out["oracle_seconds"] = np.array([dt])
            np.savez_compressed(path_of(name, c), **out)
            print(f"{name} cycle {c}: {heads.n} heads, {m} targets, oracle {dt:.1f} s -> {path_of(name, c)}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))

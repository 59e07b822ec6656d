#!/usr/bin/env python
"""kq_tas_find_replacement / kq_tas_exclusion_stats at the size of BASELINE configs[4]'s topology (4096 leaves, 4 resources): every one
of N admitted workloads of the cfg 5 population lost one node. Parity first (every podset against the oracle: status, operands, merged
assignment, statistics of the ones that failed), then the two calls timed. One JSON line."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kueue_amd import tas as T  # noqa: E402
from kueue_amd.tas_population import generate_tas  # noqa: E402


def population(n, seed=7):
    topo, rq0 = generate_tas(n_workloads=n)
    rng = np.random.default_rng(seed)
    wls, bad = [], []
    for w in range(n):
        ps = rq0.workloads[w][0]
        # an admission of the workload as the placement would have made it: pods spread over 2-6 hosts of one rack + the lost node
        rack = int(rng.integers(0, 64))
        hosts = rng.choice(64, size=int(rng.integers(2, 7)), replace=False)
        existing = [(topo.leaf_values(rack * 64 + int(h)), int(rng.integers(1, 5))) for h in sorted(hosts)]
        tr = ps.topology_request
        lost = int(rng.integers(1, 5))
        if w % 4 == 0:   # a quarter of the workloads promised one host and lost many pods: most of these cannot be repaired in place
            tr = T.TopologyRequest(required=T.HOSTNAME_LABEL); lost = int(rng.integers(8, 48))
        existing.append((["lost-node"], lost))
        wls.append([T.TASPodSetRequests("main", 0, dict(ps.single_pod_requests), tr, existing=existing)])
        bad.append("lost-node")
    return topo, T.Requests(topo, wls, unhealthy_nodes=bad)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    topo, rq = population(n)
    eng = T.TASEngine()
    eng.put(topo)
    out = eng.find_replacement(rq)
    eng.exclusion_stats(rq, out)
    from oracle import kqo   # the checker
    t0 = time.perf_counter()
    want = kqo.tas_find_replacement(topo, rq)
    oracle_s = time.perf_counter() - t0
    diff = want.equal(out)
    failed = [i for i in range(rq.n) if int(out.a["status"][i]) in (T.TAS_NOT_FIT, T.TAS_NOT_FIT_LAYERS)]
    bad_stats = [i for i in failed if out.exclusions[i] != want.exclusions[i]]
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.find_replacement(rq)
    find_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.exclusion_stats(rq, out, failed)
    stats_ms = (time.perf_counter() - t0) / reps * 1e3 if failed else None
    st = np.bincount(out.a["status"], minlength=11)
    print(json.dumps(dict(what="node replacement at the cfg 5 topology (4096 leaves x 4 resources), one lost node per workload", workloads=n,
                          parity=dict(equal=not diff and not bad_stats, fields=diff, stats_mismatch=len(bad_stats)),
                          statuses=dict(ok=int(st[T.TAS_OK]), not_fit=int(st[T.TAS_NOT_FIT]), no_replacement=int(st[T.TAS_NO_REPLACEMENT])),
                          find_replacement_ms=find_ms, replacements_per_s=n / find_ms * 1e3, failed_podsets=len(failed), exclusion_stats_ms=stats_ms,
                          oracle_one_core_s=oracle_s, note="host wall time of the C call incl. the request rewrite, H2D and D2H (round 5: the required domain travels as a leaf range, no n x leaves mask)")))
    eng.close()


if __name__ == "__main__":
    main()

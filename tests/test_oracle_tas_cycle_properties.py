"""Property tests of the oracle's TAS cycle (kqo_cycle_run_tas) on random populations: what must hold whatever the placement.
 * without TAS flavors and requests the TAS cycle IS the plain cycle (every decision array equal);
 * an admitted podset that asked for TAS holds a TopologyAssignment of exactly its pod count, on leaves of ONE domain of the level it
   required, in whole slices when slices were asked for, and every leaf had the capacity when the entry was processed;
 * a podset that did not ask for TAS never gets a TopologyAssignment; TAS flavors only go to podsets that asked (or TAS-only queues)."""
import numpy as np
import pytest

from kueue_amd import _ffi as F
from kueue_amd.tas_cycle import CycleTAS
from tests.randgen import random_case
from tests.tasgen_cycle import random_tas_cycle_case


@pytest.mark.parametrize("seed", range(120))
def test_tas_cycle_without_tas_is_the_plain_cycle(oracle, seed):
    cfg, snap, heads = random_case(seed, fair=seed % 3 == 0, tight=seed % 2 == 0)
    oracle.derive(snap)
    want = oracle.cycle_run(cfg, snap, heads)
    got, out = oracle.cycle_run_tas(cfg, snap, heads, CycleTAS(snap, heads, {}, {}), tgt_cap=want.a["tgt_adm"].size if "tgt_adm" in want.a else None)
    assert not want.equal(got)
    assert got.tas_stats["finds"] == 0 and (out.a["ps_tas"][:heads.n_ps] == -1).all()


@pytest.mark.parametrize("seed", range(400))
def test_tas_cycle_placements_are_valid(oracle, seed):
    cfg, snap, heads, ct, pod_tas = random_tas_cycle_case(seed, fair=seed % 4 == 0, tight=seed % 2 == 0, preemption=seed % 3 != 0)
    oracle.derive(snap)
    d, out = oracle.cycle_run_tas(cfg, snap, heads, ct, tgt_cap=max(16, snap.n_adm))
    if d.tas_stats["unsupported"]:
        return  # two TAS flavors in one workload: TASHandleOverlappingFlavors is not restated, the oracle says so
    tas_only = ct.arrays["cq_tas_only"]
    for i, w in enumerate(heads.workloads):
        # a podset left without flavors and without a reason ends Assign before its TAS part (flavorassigner.go:846-853): such a
        # head can be admitted without any TopologyAssignment, in the reference too
        degenerate = any(not f for f in d.flavors_of(i))
        for pi, ps in enumerate(w.pod_sets):
            g = int(heads.arrays["ps_off"][i]) + pi
            pt = pod_tas[(w.name, pi)]
            requested = pt.explicit or bool(tas_only[snap.cq_index[w.cluster_queue]])
            ta = out.topology_assignment(i, pi)
            fl = d.flavors_of(i)[pi] if pi < len(d.flavors_of(i)) else {}
            if not requested:
                assert ta is None, (w.name, pi)
                assert not any(v[0] in ct.names for v in fl.values()), (w.name, pi, fl)
                continue
            if int(d.a["action"][i]) != F.ACT_ADMIT or degenerate:
                continue
            cnt = int(d.a["ps_count"][g])
            if cnt == 0 or not fl:
                continue
            assert ta is not None, (w.name, pi, "admitted without a TopologyAssignment")
            name, doms = ta
            assert name in {v[0] for v in fl.values()}
            assert sum(c for _, c in doms) == cnt, (w.name, pi, doms, cnt)
            t = ct.names.index(name)
            topo = ct.topos[t]
            k0, k1 = int(out.a["dom_off"][g]), int(out.a["dom_off"][g + 1])
            leaves = [int(x) for x in out.a["dom_leaf"][k0:k1]]
            assert leaves == sorted(set(leaves))
            lvl = int(ct.arrays["ps_level"].reshape(-1, len(ct.names))[g, t])
            if pt.explicit and pt.topology_request.required is not None:
                assert len({topo.level_values[-1][l][:lvl + 1] for l in leaves}) == 1, (w.name, pi, "required level split", doms)
            ss = int(ct.arrays["ps_slice_size"][g])
            if ss > 1 and cnt % ss == 0:   # (a group's leader carries the workers' request, tasgen_cycle.py: one pod cannot be cut into slices of ss)
                sl = int(ct.arrays["ps_slice_level"].reshape(-1, len(ct.names))[g, t])
                per = {}
                for l, c in zip(leaves, out.a["dom_count"][k0:k1]):
                    key = topo.level_values[-1][l][:sl + 1]
                    per[key] = per.get(key, 0) + int(c)
                assert all(c % ss == 0 for c in per.values()), (w.name, pi, per, ss)
    # leaf usage after the cycle never exceeds the capacity by more than what simulate-empty reservations put there
    o = 0
    ua = out.a["tas_usage_after"]
    for topo in ct.topos:
        n = topo.n_leaves * len(topo.resources)
        assert (ua[o:o + n] >= 0).all()
        o += n

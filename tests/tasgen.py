"""Seeded random TAS topologies and request batches for differential tests (oracle vs engine)."""
import random

from kueue_amd import tas as T

LEVELS3 = ["cloud.com/topology-block", "cloud.com/topology-rack", T.HOSTNAME_LABEL]


def random_tas_case(seed, max_blocks=3, max_racks=4, max_hosts=6, n_workloads=12):
    rnd = random.Random(seed)
    shape = rnd.choice(["3host", "3host", "2host", "2rack", "1host"])
    levels = {"3host": LEVELS3, "2host": LEVELS3[1:], "2rack": LEVELS3[:2], "1host": LEVELS3[2:]}[shape]
    nodes = []
    hid = 0
    for b in range(rnd.randint(1, max_blocks)):
        for r in range(rnd.randint(1, max_racks)):
            for h in range(rnd.randint(1, max_hosts)):
                hid += 1
                alloc = {"cpu": str(rnd.randint(0, 8)), "pods": str(rnd.choice([2, 4, 10, 110]))}
                if rnd.random() < 0.6:
                    alloc["example.com/gpu"] = str(rnd.randint(0, 8))
                if rnd.random() < 0.5:
                    alloc["memory"] = f"{rnd.randint(1, 16)}Gi"
                nodes.append(T.Node(f"n{hid}", {LEVELS3[0]: f"b{b}", LEVELS3[1]: f"r{b}-{r}", T.HOSTNAME_LABEL: f"x{hid:03d}"}, alloc,
                                    ready=rnd.random() > 0.05))
    topo = T.Topology(levels, nodes, resources=["cpu", "memory", "example.com/gpu"], profile_mixed=rnd.random() > 0.2)
    # some TAS usage already on the leaves
    use = {}
    for leaf in range(topo.n_leaves):
        if rnd.random() < 0.4:
            use[leaf] = {"cpu": rnd.randint(0, 4) * 1000, "pods": rnd.randint(0, 3)}
            if rnd.random() < 0.5:
                use[leaf]["example.com/gpu"] = rnd.randint(0, 4)
    topo.set_tas_usage(use)
    workloads, sim = [], []
    for w in range(rnd.randint(1, n_workloads)):
        podsets = []
        kind = rnd.choice(["single", "single", "single", "two", "group"])
        n_ps = {"single": 1, "two": 2, "group": 2}[kind]
        for p in range(n_ps):
            reqs = {}
            if rnd.random() < 0.8:
                reqs["cpu"] = rnd.choice([0, 250, 500, 1000, 2000])
            if rnd.random() < 0.4:
                reqs["example.com/gpu"] = rnd.randint(0, 2)
            if rnd.random() < 0.3:
                reqs["memory"] = rnd.choice([1, 2]) * (1 << 30)
            lv = rnd.choice(levels)
            mode = rnd.choice(["required", "preferred", "unconstrained", "implied", "slice-only"])
            tr = None
            slice_kw = {}
            if rnd.random() < 0.35 or mode == "slice-only":
                li = levels.index(lv) if mode in ("required", "preferred") else 0
                slice_kw = dict(slice_required_topology=rnd.choice(levels[li:] if rnd.random() < 0.9 else levels), slice_size=rnd.choice([1, 2, 2, 3, 4]))
            if mode == "required":
                tr = T.TopologyRequest(required=lv, **slice_kw)
            elif mode == "preferred":
                tr = T.TopologyRequest(preferred=lv, **slice_kw)
            elif mode == "unconstrained":
                tr = T.TopologyRequest(unconstrained=True, **slice_kw)
            elif mode == "slice-only":
                tr = T.TopologyRequest(**slice_kw)
            count = rnd.choice([0, 1, 1, 2, 3, 4, 6, 8, 12, 20])
            if slice_kw and slice_kw["slice_size"] > 0:
                count = max(1, count // slice_kw["slice_size"]) * slice_kw["slice_size"]
            ps = T.TASPodSetRequests(f"ps{p}", count, reqs, tr)
            if kind == "group":
                ps.group = "g"
                if p == 1:
                    ps.count = 1
            if rnd.random() < 0.15:
                ps.leaf_ok = [rnd.random() < 0.7 for _ in range(topo.n_leaves)]
            podsets.append(ps)
        if kind == "group":  # leader + workers share the topology request (PodSet grouping validation)
            podsets[1].topology_request = podsets[0].topology_request
        workloads.append(podsets)
        sim.append(rnd.random() < 0.15)
    return topo, T.Requests(topo, workloads, simulate_empty=sim)


LEVELS4 = ["cloud.com/datacenter"] + LEVELS3


def random_tas_multilayer_case(seed, n_workloads=10):
    """Topologies of 3-4 levels and podsets that mostly carry PodsetSliceRequiredTopologyConstraints with inner layers (TASMultiLayerTopology),
    valid and invalid ones (unknown key, not below the previous layer, size that does not divide), with and without leaders."""
    rnd = random.Random(0x3A7E5 + seed)
    levels = rnd.choice([LEVELS4, LEVELS4, LEVELS3, LEVELS3[1:]])
    nodes = []
    hid = 0
    for dc in range(rnd.randint(1, 2)):
        for b in range(rnd.randint(1, 3)):
            for r in range(rnd.randint(1, 3)):
                for h in range(rnd.randint(1, 4)):
                    hid += 1
                    alloc = {"cpu": str(rnd.choice([0, 1, 2, 3, 4, 4, 6, 8])), "pods": str(rnd.choice([3, 10, 110]))}
                    nodes.append(T.Node(f"n{hid}", {LEVELS4[0]: f"dc{dc}", LEVELS4[1]: f"b{dc}-{b}", LEVELS4[2]: f"r{dc}-{b}-{r}", T.HOSTNAME_LABEL: f"x{hid:03d}"},
                                        alloc, ready=rnd.random() > 0.04))
    topo = T.Topology(levels, nodes, resources=["cpu"], profile_mixed=rnd.random() > 0.3)
    use = {leaf: {"cpu": rnd.randint(0, 2) * 1000, "pods": rnd.randint(0, 2)} for leaf in range(topo.n_leaves) if rnd.random() < 0.3}
    topo.set_tas_usage(use)
    workloads, sim = [], []
    for w in range(rnd.randint(1, n_workloads)):
        grouped = rnd.random() < 0.25
        mode = rnd.choice(["required", "required", "preferred", "slice-only", "unconstrained"])
        top = rnd.randrange(len(levels) - 1) if len(levels) > 1 else 0
        # layers on strictly lower and lower levels, sizes dividing each other
        first = rnd.randrange(top, len(levels))
        lv_idx = [first]
        while lv_idx[-1] + 1 < len(levels) and rnd.random() < 0.75:
            lv_idx.append(rnd.randrange(lv_idx[-1] + 1, len(levels)))
        sizes = [rnd.choice([1, 2, 3])]
        for _ in lv_idx[1:]:
            sizes.append(sizes[-1] * rnd.choice([1, 2, 2, 3]))
        sizes.reverse()
        cons = [(levels[l], s) for l, s in zip(lv_idx, sizes)]
        flaw = rnd.random()
        if len(cons) > 1 and flaw < 0.08:
            cons[-1] = ("example.com/unknown", cons[-1][1])
        elif len(cons) > 1 and flaw < 0.16:
            cons[-1] = (cons[rnd.randrange(len(cons) - 1)][0], cons[-1][1])        # not below the previous layer
        elif len(cons) > 1 and flaw < 0.24:
            cons[-1] = (cons[-1][0], cons[-2][1] + 1)                                # does not divide
        elif flaw < 0.28:
            cons[0] = (cons[0][0], 0)                                                # slice size not provided
        kw = dict(slice_constraints=cons)
        if mode == "required":
            tr = T.TopologyRequest(required=levels[top], **kw)
        elif mode == "preferred":
            tr = T.TopologyRequest(preferred=levels[top], **kw)
        elif mode == "unconstrained":
            tr = T.TopologyRequest(unconstrained=True, **kw)
        else:
            tr = T.TopologyRequest(**kw)
        outer = max(1, cons[0][1])
        count = outer * rnd.choice([1, 1, 2, 3, 4])
        reqs = {"cpu": rnd.choice([500, 1000, 1000, 2000])}
        podsets = [T.TASPodSetRequests("workers", count, reqs, tr)]
        if grouped:
            podsets[0].group = "g"
            podsets.append(T.TASPodSetRequests("leader", 1, {"cpu": rnd.choice([500, 1000])}, tr, group="g"))
            if rnd.random() < 0.5:
                podsets.reverse()
        elif rnd.random() < 0.2:   # a second, plain podset after the layered one sees its assumed usage
            podsets.append(T.TASPodSetRequests("extra", rnd.choice([1, 2, 4]), {"cpu": 1000}, T.TopologyRequest(preferred=levels[-1])))
        if rnd.random() < 0.1:
            podsets[0].leaf_ok = [rnd.random() < 0.8 for _ in range(topo.n_leaves)]
        workloads.append(podsets)
        sim.append(rnd.random() < 0.1)
    return topo, T.Requests(topo, workloads, simulate_empty=sim)


def deep_tas_case(seed, n_levels=12, n_res=30, n_workloads=10):
    """The limits of the API rather than the usual shapes: a topology of up to 16 levels (TopologySpec.Levels MaxItems=16; the lowest is the
    hostname) with one or two children per domain on the way down, and up to 30 resources per node (the reference's BenchmarkSchedulerTAS
    runs 30, scheduler_tas_bench_test.go:46); podsets ask for a handful of them, required / preferred / unconstrained on any level, with
    slices on a lower level."""
    rnd = random.Random(0xDEE9 + seed)
    levels = [f"example.com/l{i:02d}" for i in range(n_levels - 1)] + [T.HOSTNAME_LABEL]
    res = [f"example.com/r{i:02d}" for i in range(n_res)]
    nodes = []
    paths = [[]]
    for l in range(n_levels - 1):
        nxt = []
        for p in paths:
            for ch in range(2 if (len(paths) < 24 and rnd.random() < 0.45) else 1):
                nxt.append(p + [f"d{l}-{len(nxt)}"])
        paths = nxt
    hid = 0
    for p in paths:
        for h in range(rnd.randint(1, 3)):
            hid += 1
            alloc = {"pods": str(rnd.choice([4, 10, 110]))}
            for r in res:
                if rnd.random() < 0.7:
                    alloc[r] = str(rnd.randint(0, 16))
            labels = {levels[i]: p[i] for i in range(n_levels - 1)}
            labels[T.HOSTNAME_LABEL] = f"x{hid:03d}"
            nodes.append(T.Node(f"n{hid}", labels, alloc, ready=rnd.random() > 0.03))
    topo = T.Topology(levels, nodes, resources=res, profile_mixed=rnd.random() > 0.3)
    use = {}
    for leaf in range(topo.n_leaves):
        if rnd.random() < 0.4:
            use[leaf] = {r: rnd.randint(0, 4) for r in rnd.sample(res, 3)}
            use[leaf]["pods"] = rnd.randint(0, 3)
    topo.set_tas_usage(use)
    workloads, sim = [], []
    for w in range(rnd.randint(2, n_workloads)):
        reqs = {r: rnd.randint(0, 3) for r in rnd.sample(res, rnd.randint(1, 6))}
        li = rnd.randrange(n_levels)
        mode = rnd.choice(["required", "preferred", "unconstrained", "slice-only"])
        slice_kw = {}
        if rnd.random() < 0.4 or mode == "slice-only":
            lo = li if mode in ("required", "preferred") else 0
            slice_kw = dict(slice_required_topology=levels[rnd.randrange(lo, n_levels)], slice_size=rnd.choice([1, 2, 3]))
        if mode == "required":
            tr = T.TopologyRequest(required=levels[li], **slice_kw)
        elif mode == "preferred":
            tr = T.TopologyRequest(preferred=levels[li], **slice_kw)
        elif mode == "unconstrained":
            tr = T.TopologyRequest(unconstrained=True, **slice_kw)
        else:
            tr = T.TopologyRequest(**slice_kw)
        count = rnd.choice([1, 2, 3, 4, 6, 9, 12])
        if slice_kw:
            count = max(1, count // slice_kw["slice_size"]) * slice_kw["slice_size"]
        workloads.append([T.TASPodSetRequests("main", count, reqs, tr)])
        sim.append(rnd.random() < 0.2)
    return topo, T.Requests(topo, workloads, simulate_empty=sim)

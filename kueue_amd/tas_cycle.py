"""Host side of Topology-Aware Scheduling INSIDE the scheduling cycle: what the Go snapshot / queue manager would flatten next to
kq_snapshot and kq_heads (boundary: include/kq_cycle_tas.h, restated on the CPU by the oracle ahead of the engine).

What stays on the host here stays on the host in the reference:
  * the TASFlavorSnapshot of every TAS ResourceFlavor (Spec.TopologyName set and the topology cached): nodes matching the flavor's
    nodeLabels, free capacity, usage of admitted TAS workloads (pkg/cache/scheduler/tas_flavor.go, snapshot.go:240-262)
  * clusterQueue.isTASOnly (clusterqueue.go:746)
  * checkPodSetAndFlavorMatchForTAS (pkg/scheduler/flavorassigner/tas_flavorassigner.go:164-211): label / name logic, folded into
    the podset's excluded flavors like taints and node affinity
  * levelKeyWithImpliedFallback / sliceLevelKeyWithDefault resolved against every TAS flavor (tas_flavor_snapshot.go:1197-1238)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi as F
from .api import Heads, Snapshot, amount_from_quantity
from .tas import KQ_TAS_UNCONSTRAINED, Node, TASPodSetRequests, Topology, TopologyRequest, kq_tas_topology

CYCLE_TAS_ABI_SYMBOLS = ["kq_cycle_run_tas"]   # include/kq_cycle_tas.h
PS_TAS_EXPLICIT = 1
CT_NO_RECOMPUTE = 1
CT_NO_FAIL_FAST = 2
EX_UNHEALTHY, EX_FIRST = 1, 2


class kq_cycle_tas(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32), ("n_tas", C.c_int32), ("tas_flavor", F.i32p), ("topo", C.POINTER(kq_tas_topology)), ("cq_tas_only", F.u8p),
        ("adm_off", F.i32p), ("adm_tas", F.i32p), ("adm_leaf", F.i32p), ("adm_count", F.i32p), ("adm_req", F.i64p),
        ("ps_flags", F.u8p), ("ps_kind", F.u8p), ("ps_level", F.i32p), ("ps_slice_size", F.i32p), ("ps_slice_level", F.i32p),
        ("ps_group", F.i32p), ("ps_req", F.i64p),
        ("ps_n_layers", F.i32p), ("ps_layer_level", F.i32p), ("ps_layer_size", F.i32p),   # TASMultiLayerTopology, NULL = single layer
        # the second pass: what Status.Admission holds for the heads' podsets, NULL = no head holds an admission
        ("ps_adm_flavor", F.i32p), ("ps_ex_off", F.i32p), ("ps_ex_leaf", F.i32p), ("ps_ex_count", F.i32p), ("ps_ex_flags", F.u8p),
        # node feasibility rows (taints / tolerations, nodeSelector, required affinity), NULL = every leaf for every podset
        ("ps_mask", F.i32p), ("leaf_mask", F.u8p), ("n_masks", C.c_int32), ("mask_stride", C.c_int32),
    ]


class kq_cycle_tas_out(C.Structure):
    _fields_ = [("ps_tas", F.i32p), ("dom_off", F.i32p), ("dom_leaf", F.i32p), ("dom_count", F.i32p), ("dom_cap", C.c_int32),
                ("tas_usage_after", F.i64p)]


@dataclass
class ResourceFlavor:
    """kueue.ResourceFlavor: nodeLabels select the nodes, topologyName makes it a TAS flavor; nodeTaints / tolerations take part in the
    host-side eligibility (kueue_amd/node_match.py)."""
    name: str
    node_labels: Dict[str, str] = field(default_factory=dict)
    topology_name: Optional[str] = None
    node_taints: list = field(default_factory=list)     # node_match.Taint
    tolerations: list = field(default_factory=list)     # node_match.Toleration


@dataclass
class PodSetTAS:
    """The TAS part of one kueue.PodSet of a pending workload."""
    topology_request: Optional[TopologyRequest] = None
    group: Optional[str] = None
    single_pod_requests: Dict[str, object] = field(default_factory=dict)   # from the pod spec (resources.NewRequestsFromPodSpec)
    # node feasibility per TAS flavor NAME, by leaf index (taints vs the podset's + the flavor's tolerations, PodSpec.NodeSelector,
    # required node affinity: tas_flavor_snapshot.go:955-963, host-evaluated string matching); a flavor not named = every leaf
    leaf_ok: Optional[Dict[str, Sequence[bool]]] = None

    @property
    def explicit(self) -> bool:  # workload.IsExplicitlyRequestingTAS workload.go:535-541
        t = self.topology_request
        return t is not None and (t.unconstrained or t.required is not None or t.preferred is not None or
                                  t.slice_required_topology is not None or t.slice_size is not None or bool(t.slice_constraints))


@dataclass
class AdmittedTAS:
    """One podset assignment of an admitted workload: flavor + TopologyAssignment + per-pod requests (workload.TASUsage)."""
    flavor: str
    domains: List[Tuple[Tuple[str, ...], int]]     # (levelValues as stored: hostname only when the lowest level is the node, count)
    single_pod_requests: Dict[str, object]


@dataclass
class HeadAdmission:
    """Status.Admission + Status.UnhealthyNodes of a head on its second pass (workload.NeedsSecondPass workload.go:974): per podset the
    admitted flavor of every resource and the TopologyAssignment (None: none yet — a delayed topology request)."""
    flavors: List[Dict[str, str]]                                    # [podset] resource -> flavor
    domains: List[Optional[List[Tuple[Tuple[str, ...], int]]]]       # [podset] (levelValues as stored, count) in the assignment's order
    unhealthy_nodes: List[str] = field(default_factory=list)
    admitted: bool = True                                            # workload.IsAdmitted

    def names_unhealthy(self) -> bool:   # HasTopologyAssignmentWithUnhealthyNode workload.go:1392
        return self.admitted and any(v[-1] in self.unhealthy_nodes for d in self.domains if d for v, _ in d)


def tas_only(cq, tas_flavors) -> bool:
    return all(fq.name in tas_flavors for rg in cq.resource_groups for fq in rg.flavors)


def has_level(topo: Topology, t: Optional[TopologyRequest]) -> bool:
    """TASFlavorSnapshot.HasLevel :1170-1194 (no multi-layer constraints)."""
    if t is None:
        return False
    key = t.required if t.required is not None else t.preferred
    if key is None:
        if t.slice_required_topology is not None:
            key = topo.levels[0]
        elif t.unconstrained:
            key = topo.levels[-1]
    if key is None or key not in topo.levels:
        return False
    slice_key = t.slice_required_topology if t.slice_required_topology is not None else topo.levels[-1]
    return slice_key in topo.levels


def excluded_flavors_for_tas(cq, ps_requests: Sequence[str], ptas: PodSetTAS, topologies: Dict[str, Topology], flavors: Dict[str, ResourceFlavor]) -> List[str]:
    """checkPodSetAndFlavorMatchForTAS for every flavor of the ClusterQueue -> the flavors the podset may not use."""
    only = tas_only(cq, topologies)
    requested = ptas.explicit or only
    implied = not ptas.explicit and only
    out = []
    for rg in cq.resource_groups:
        covered = set(rg.covered_resources)
        for fq in rg.flavors:
            rf = flavors.get(fq.name)
            is_tas = rf is not None and rf.topology_name is not None
            if requested:
                if implied:
                    continue
                if not is_tas:
                    if any(r in covered for r in ps_requests):
                        out.append(fq.name)   # "Flavor %q does not support TopologyAwareScheduling"
                    continue
                if fq.name not in topologies:
                    out.append(fq.name)       # "information missing in TAS cache"
                elif not has_level(topologies[fq.name], ptas.topology_request):
                    out.append(fq.name)       # "does not contain the requested level"
            elif is_tas:
                out.append(fq.name)           # "supports only TopologyAwareScheduling"
    return out


class CycleTAS:
    """kq_cycle_tas for one (Snapshot, Heads)."""

    def __init__(self, snap: Snapshot, heads: Heads, topologies: Dict[str, Topology], pod_tas: Dict[Tuple[str, int], PodSetTAS],
                 admitted_tas: Optional[Dict[str, List[AdmittedTAS]]] = None, recompute: bool = True,
                 head_admission: Optional[Dict[str, HeadAdmission]] = None, fail_fast: bool = True, overlapping: bool = True):
        """overlapping = features.TASHandleOverlappingFlavors (beta, on): cache.Snapshot() hands a flavor whose lowest level is the
        hostname the TAS usage EVERY such flavor holds on its nodes (snapshot.go:216-240 -> tas_flavor.go:209 addTASUsageForHeldDomains),
        so two flavors over the same nodes see each other's pods. Inside the cycle usage moves per flavor (clusterqueue_snapshot.go:121),
        so the other flavors' share is base usage of the snapshot: it goes into kq_tas_topology.tas_usage, not into the rows' CSR."""
        self.snap, self.heads = snap, heads
        names = sorted(topologies)                       # slices.Sorted(maps.Keys(...)) clusterqueue_snapshot.go:220
        self.names = names
        self.topos = [topologies[n] for n in names]
        nt = len(names)
        res = self.topos[0].resources if nt else ["pods"]
        for t in self.topos:
            assert t.resources == res, "every TAS topology must be built over one resource dictionary"
        self.resources = res
        R = len(res)
        rix = {r: i for i, r in enumerate(res)}
        tix = {n: i for i, n in enumerate(names)}
        a: Dict[str, np.ndarray] = {}
        a["tas_flavor"] = np.array([snap.flavor_index[n] for n in names] or [0], np.int32)
        a["cq_tas_only"] = np.array([1 if tas_only(cq, topologies) else 0 for cq in snap.cluster_queues], np.uint8)
        # admitted rows
        off, at, al, ac, ar = [0], [], [], [], []
        others: Dict[int, np.ndarray] = {}   # TAS flavor -> [n_leaves][R] usage the OTHER hostname-level flavors hold on its nodes
        for w in snap.admitted:
            for u in (admitted_tas or {}).get(w.name, []):
                topo = topologies[u.flavor]
                req = np.zeros(R, np.int64)
                for r, q in u.single_pod_requests.items():
                    req[rix[r]] = _amount(r, q)
                for values, cnt in u.domains:
                    if overlapping and topo.lowest_is_node:
                        # the cache keys a hostname-level flavor's usage by the hostname alone (utiltas.DomainID of the stored values): the
                        # aggregate reaches every flavor that holds that node, whether this one still does or not (snapshot.go:271-285)
                        for b, tb in enumerate(self.topos):
                            if names[b] == u.flavor or not tb.lowest_is_node:
                                continue
                            lb = _leaf_of(tb, (values[-1],))
                            if lb is None:
                                continue
                            o = others.setdefault(b, np.zeros((tb.n_leaves, R), np.int64))
                            o[lb] += np.where(req > 0, req, 0) * cnt
                            o[lb, rix["pods"]] += cnt
                    leaf = _leaf_of(topo, tuple(values))
                    if leaf is None:
                        continue  # a domain the flavor no longer holds: updateTASUsage ignores it (tas_flavor_snapshot.go:250)
                    at.append(tix[u.flavor]); al.append(leaf); ac.append(cnt); ar.append(req)
            off.append(len(at))
        a["adm_off"] = np.array(off, np.int32)
        a["adm_tas"] = np.array(at or [0], np.int32); a["adm_leaf"] = np.array(al or [0], np.int32); a["adm_count"] = np.array(ac or [0], np.int32)
        a["adm_req"] = (np.stack(ar) if ar else np.zeros((1, R), np.int64)).reshape(-1).copy()
        # heads
        n_ps = heads.n_ps
        flags = np.zeros(max(n_ps, 1), np.uint8); kind = np.zeros(max(n_ps, 1), np.uint8)
        level = np.full((max(n_ps, 1), max(nt, 1)), -1, np.int32); slevel = np.full((max(n_ps, 1), max(nt, 1)), -1, np.int32)
        ssize = np.ones(max(n_ps, 1), np.int32); group = np.full(max(n_ps, 1), -1, np.int32)
        req = np.zeros((max(n_ps, 1), R), np.int64)
        ML = 16   # KQ_TAS_MAX_LEVELS
        nlay = np.zeros(max(n_ps, 1), np.int32); llev = np.full((max(n_ps, 1), max(nt, 1), ML), -1, np.int32); lsz = np.zeros((max(n_ps, 1), ML), np.int32)
        g = 0
        gid: Dict[Tuple[str, str], int] = {}
        for w in heads.workloads:
            for pi, ps in enumerate(w.pod_sets):
                pt = pod_tas.get((w.name, pi), PodSetTAS())
                if pt.explicit:
                    flags[g] |= PS_TAS_EXPLICIT
                tr = TASPodSetRequests(name=ps.name, count=ps.count, single_pod_requests={}, topology_request=pt.topology_request if pt.explicit else None)
                for ti, topo in enumerate(self.topos):
                    level[g, ti], kind[g], ssize[g], slevel[g, ti] = topo.resolve(tr)
                    lay = topo.resolve_layers(tr)
                    nlay[g] = len(lay)
                    for j, (lv, sz) in enumerate(lay[:ML]):
                        llev[g, ti, j] = lv; lsz[g, j] = sz
                if not self.topos:
                    kind[g] = KQ_TAS_UNCONSTRAINED
                if pt.group is not None:
                    group[g] = gid.setdefault((w.name, pt.group), len(gid))
                for r, q in pt.single_pod_requests.items():
                    v = _amount(r, q)
                    if v != 0 and nt:
                        req[g, rix[r]] = v
                g += 1
        a.update(ps_flags=flags, ps_kind=kind, ps_level=level.reshape(-1).copy(), ps_slice_size=ssize, ps_slice_level=slevel.reshape(-1).copy(),
                 ps_group=group, ps_req=req.reshape(-1).copy())
        if (nlay > 1).any():
            a.update(ps_n_layers=nlay, ps_layer_level=llev.reshape(-1).copy(), ps_layer_size=lsz.reshape(-1).copy())
        # node feasibility rows (kq_cycle_tas.ps_mask / leaf_mask): equal masks share a row
        if nt and any(pod_tas.get((w.name, pi), PodSetTAS()).leaf_ok for w in heads.workloads for pi in range(len(w.pod_sets))):
            stride = max(t.n_leaves for t in self.topos)
            rows: Dict[bytes, int] = {}
            mats: List[np.ndarray] = []
            pm = np.full((max(n_ps, 1), nt), -1, np.int32)
            g = 0
            for w in heads.workloads:
                for pi in range(len(w.pod_sets)):
                    lo = pod_tas.get((w.name, pi), PodSetTAS()).leaf_ok or {}
                    for name, m in lo.items():
                        if name not in tix or m is None:
                            continue
                        row = np.zeros(stride, np.uint8)
                        mm = np.asarray(m, np.uint8)
                        assert mm.size == self.topos[tix[name]].n_leaves, "a leaf_ok mask has one entry per leaf of its TAS flavor"
                        row[:mm.size] = mm
                        if row[:mm.size].all():
                            continue
                        key = row.tobytes()
                        if key not in rows:
                            rows[key] = len(mats); mats.append(row)
                        pm[g, tix[name]] = rows[key]
                    g += 1
            if mats:
                a.update(ps_mask=pm.reshape(-1).copy(), leaf_mask=np.stack(mats).reshape(-1).copy())
                self._mask_scalars = dict(n_masks=len(mats), mask_stride=stride)
        self.head_admission: Dict[str, HeadAdmission] = dict(head_admission or {})   # by workload name: the message text needs its names
        if head_admission:
            # the second pass: Status.Admission of the heads that hold one (include/kq_cycle_tas.h ps_adm_flavor / ps_ex_*)
            nR = snap.n_resource
            adm = np.full((max(n_ps, 1), nR), -1, np.int32)
            xo, xl, xc, xf = [0], [], [], []
            self.ex_domains: Dict[int, list] = {}
            g = 0
            for w in heads.workloads:
                ha = head_admission.get(w.name)
                for pi, ps in enumerate(w.pod_sets):
                    if ha is not None:
                        assert w.has_quota_reservation, "a head that holds an admission has its quota reserved"
                        for r, fl in ha.flavors[pi].items():
                            adm[g, snap.resource_index[r]] = snap.flavor_index[fl]
                        tas_fl = [fl for fl in set(ha.flavors[pi].values()) if fl in tix]
                        for values, cnt in (ha.domains[pi] or []):
                            assert len(tas_fl) == 1, "a TopologyAssignment belongs to the podset's one TAS flavor"
                            leaf = _leaf_of(topologies[tas_fl[0]], tuple(values))
                            f = 0
                            if values[-1] in ha.unhealthy_nodes:
                                f |= EX_UNHEALTHY
                                if ha.unhealthy_nodes and values[-1] == ha.unhealthy_nodes[0]:
                                    f |= EX_FIRST
                            xl.append(-1 if leaf is None else leaf); xc.append(cnt); xf.append(f)
                            self.ex_domains.setdefault(g, []).append((list(values), cnt))
                    xo.append(len(xl))
                    g += 1
            a.update(ps_adm_flavor=adm.reshape(-1).copy(), ps_ex_off=np.array(xo, np.int32), ps_ex_leaf=np.array(xl or [0], np.int32),
                     ps_ex_count=np.array(xc or [0], np.int32), ps_ex_flags=np.array(xf or [0], np.uint8))
        self.arrays = a
        self._topo_arr = (kq_tas_topology * max(nt, 1))()
        for i, t in enumerate(self.topos):
            st = t.struct()
            C.memmove(C.byref(self._topo_arr, i * C.sizeof(kq_tas_topology)), C.byref(st), C.sizeof(kq_tas_topology))
        self._base_usage = {}
        for b, o in others.items():
            if o.any():
                self._base_usage[b] = (self.topos[b].arrays["tas_usage"].reshape(-1) + o.reshape(-1)).astype(np.int64)
                self._topo_arr[b].tas_usage = F.ptr(self._base_usage[b])
        self._struct = kq_cycle_tas()
        F.fill_struct(self._struct, a, dict(n_tas=nt, flags=(0 if recompute else CT_NO_RECOMPUTE) | (0 if fail_fast else CT_NO_FAIL_FAST),
                                            **getattr(self, "_mask_scalars", {})))
        self._struct.topo = C.cast(self._topo_arr, C.POINTER(kq_tas_topology))

    def struct(self) -> kq_cycle_tas:
        return self._struct



class CycleTASOut:
    def __init__(self, ct: CycleTAS, dom_cap: Optional[int] = None):
        self.ct = ct
        n_ps = ct.heads.n_ps
        cap = dom_cap if dom_cap is not None else max(64, int(ct.heads.arrays["ps_count"].sum()) + n_ps)
        nu = sum(t.n_leaves * len(t.resources) for t in ct.topos)
        self.a = dict(ps_tas=np.full(max(n_ps, 1), -1, np.int32), dom_off=np.zeros(n_ps + 1, np.int32), dom_leaf=np.zeros(cap, np.int32),
                      dom_count=np.zeros(cap, np.int32), tas_usage_after=np.zeros(max(nu, 1), np.int64))
        self._struct = kq_cycle_tas_out()
        F.fill_struct(self._struct, self.a, dict(dom_cap=cap))

    def struct(self) -> kq_cycle_tas_out:
        return self._struct

    def topology_assignment(self, head: int, ps: int):
        """-> (TAS flavor name, [(levelValues, count)]) of a head's podset, or None."""
        g = int(self.ct.heads.arrays["ps_off"][head]) + ps
        t = int(self.a["ps_tas"][g])
        if t < 0:
            return None
        topo = self.ct.topos[t]
        o = self.a["dom_off"]
        if any(int(self.a["dom_leaf"][k]) < 0 for k in range(o[g], o[g + 1])):
            # the admission's own assignment, untouched (it names a node the snapshot no longer holds): its values as the caller gave them
            ex = getattr(self.ct, "ex_domains", {}).get(g, [])
            assert [c for _, c in ex] == [int(self.a["dom_count"][k]) for k in range(o[g], o[g + 1])], (g, ex)
            return self.ct.names[t], [(list(v), c) for v, c in ex]
        return self.ct.names[t], [(topo.leaf_values(int(self.a["dom_leaf"][k])), int(self.a["dom_count"][k])) for k in range(o[g], o[g + 1])]


def _amount(r: str, q) -> int:
    return int(q) if isinstance(q, (int, np.integer)) else amount_from_quantity(r, q)


def _leaf_of(topo: Topology, values: Tuple[str, ...]) -> Optional[int]:
    if topo.lowest_is_node and len(values) == 1:
        for i, v in enumerate(topo.level_values[-1]):
            if v[-1] == values[0]:
                return i
        return None
    return topo.index[-1].get(tuple(values))


def build_topologies(flavors: Sequence[ResourceFlavor], topology_levels: Dict[str, Sequence[str]], nodes: Sequence[Node],
                     non_tas_usage: Optional[Dict[str, Dict[str, object]]] = None, extra_resources: Sequence[str] = ()) -> Dict[str, Topology]:
    """One Topology per TAS flavor over a shared resource dictionary; a flavor's nodes are those matching all its nodeLabels."""
    res = set(extra_resources) | {"pods"}
    for n in nodes:
        res.update(n.allocatable)
    for u in (non_tas_usage or {}).values():
        res.update(u)
    out = {}
    for rf in flavors:
        if rf.topology_name is None or rf.topology_name not in topology_levels:
            continue
        mine = [n for n in nodes if all(n.labels.get(k) == v for k, v in rf.node_labels.items())]
        out[rf.name] = Topology(topology_levels[rf.topology_name], mine, non_tas_usage, resources=sorted(res))
    return out


def load_tas_case(case: dict, cycle: int = 1):
    """A whole-cycle TAS fixture (tests/golden/schedule_tas.yaml) -> (cfg, Snapshot [not derived], Heads, CycleTAS)."""
    import copy

    from .fixtures import _cq, load_case

    case = copy.deepcopy(case)
    from . import node_match as NM

    def _tols(l):
        return [NM.Toleration(t.get("key", ""), t.get("operator") or "Equal", t.get("value", ""), t.get("effect", "")) for t in l or []]

    def _taints(l):
        return [NM.Taint(t["key"], t.get("value", ""), t.get("effect", "")) for t in l or []]

    def _terms(l):
        return None if l is None else [NM.NodeSelectorTerm([NM.NodeSelectorRequirement(e["key"], e["operator"], list(e.get("values") or []))
                                                            for e in t.get("matchExpressions") or []]) for t in l]

    flavors = {f["name"]: ResourceFlavor(f["name"], dict(f.get("nodeLabels") or {}), f.get("topologyName"), _taints(f.get("nodeTaints")), _tols(f.get("tolerations")))
               for f in case.get("resourceFlavors", [])}
    node_taints = {n["name"]: _taints(n.get("taints")) for n in case.get("nodes", []) if n.get("taints")}
    nodes = [Node(n["name"], dict(n.get("labels") or {}), dict(n.get("allocatable") or {}), bool(n.get("ready", True)), bool(n.get("unschedulable", False)))
             for n in case.get("nodes", [])]
    extra = set()
    for w in case.get("pending", []) + case.get("admitted", []):
        for ps in w.get("podsets", []):
            extra.update((ps.get("requests") or {}).keys()); extra.update((ps.get("podRequests") or {}).keys())
    non_tas = {node: {r: (sum(_amount(r, q) for q in qs) if isinstance(qs, (list, tuple)) else _amount(r, qs)) for r, qs in d.items()}
               for node, d in (case.get("nonTASUsage") or {}).items()}
    topologies = build_topologies(list(flavors.values()), {k: list(v) for k, v in (case.get("topologies") or {}).items()}, nodes,
                                  non_tas, sorted(extra))
    cqs = {c["name"]: _cq(c) for c in case.get("clusterQueues", [])}

    def req_of(ps):
        t = ps.get("topologyRequest")
        tr = None
        if t is not None:
            tr = TopologyRequest(required=t.get("required"), preferred=t.get("preferred"), unconstrained=bool(t.get("unconstrained", False)),
                                 slice_required_topology=t.get("sliceRequiredTopology"), slice_size=t.get("sliceSize"),
                                 slice_constraints=[(c["topology"], c["size"]) for c in t["sliceConstraints"]] if t.get("sliceConstraints") else None)
        return PodSetTAS(tr, ps.get("group"), dict(ps.get("podRequests") or ps.get("requests") or {}))

    pod_tas: Dict[Tuple[str, int], PodSetTAS] = {}
    for w in case.get("pending", []):
        for pi, ps in enumerate(w.get("podsets", [])):
            pt = req_of(ps)
            pod_tas[(w["name"], pi)] = pt
            ex = excluded_flavors_for_tas(cqs[w["cq"]], list((ps.get("requests") or ps.get("totalRequests") or {}).keys()), pt, topologies, flavors)
            # taints / tolerations, nodeSelector, required affinity: the flavor half (checkFlavorForPodSets flavorassigner.go:1243-1260 ->
            # excluded flavors) and the node half (FindFeasibleNodes -> the podset's feasible leaves on every TAS flavor, kq_cycle_tas.ps_mask)
            tols, sel, terms = _tols(ps.get("tolerations")), dict(ps.get("nodeSelector") or {}), _terms(ps.get("requiredAffinity"))
            if tols or sel or terms is not None or node_taints or any(rf.node_taints for rf in flavors.values()):
                for rg in cqs[w["cq"]].resource_groups:
                    for fq in rg.flavors:
                        rf = flavors.get(fq.name)
                        if rf is not None and NM.flavor_mismatch(sel, terms, tols, rf.node_labels, rf.node_taints, rf.tolerations) is not None:
                            ex.append(fq.name)
                masks = {}
                for name, topo in topologies.items():
                    m = NM.leaf_mask(topo.leaf_nodes(), topo.lowest_is_node, tols + flavors[name].tolerations, sel, terms, node_taints)
                    if m is not None:
                        masks[name] = m
                pt.leaf_ok = masks or None
            ps["excludedFlavors"] = sorted(set(ps.get("excludedFlavors") or []) | set(ex))
    admitted_tas: Dict[str, List[AdmittedTAS]] = {}
    for w in case.get("admitted", []):
        for ps in w.get("podsets", []):
            ta = ps.get("topologyAssignment")
            if ta is None:
                continue
            fl = [f for f in set((ps.get("flavors") or {}).values()) if f in topologies]
            if len(fl) != 1:
                continue
            admitted_tas.setdefault(w["name"], []).append(
                AdmittedTAS(fl[0], [(tuple(d[0]), int(d[1])) for d in ta["domains"]], dict(ps.get("podRequests") or ps.get("requests") or {})))
    case.setdefault("flavors", [])
    case["flavors"] = sorted(set(case["flavors"]) | set(flavors))
    cfg, snap, heads = load_case(case, cycle)
    recompute = bool((case.get("gatesGo") or {}).get("TASRecomputeAssignmentWithinSchedulingCycle", True))
    fail_fast = bool((case.get("gatesGo") or {}).get("TASFailedNodeReplacementFailFast", True))
    head_adm = {}
    for w in case.get("pending", []):
        if w.get("admission") is None:
            continue
        head_adm[w["name"]] = HeadAdmission(
            flavors=[dict(ps.get("flavors") or {}) for ps in w["admission"]],
            domains=[[(tuple(d[0]), int(d[1])) for d in ps["topologyAssignment"]["domains"]] if ps.get("topologyAssignment") else None for ps in w["admission"]],
            unhealthy_nodes=list(w.get("unhealthyNodes") or []), admitted=bool(w.get("isAdmitted", True)))
    return cfg, snap, heads, CycleTAS(snap, heads, topologies, pod_tas, admitted_tas, recompute=recompute, head_admission=head_adm or None,
                                      fail_fast=fail_fast)

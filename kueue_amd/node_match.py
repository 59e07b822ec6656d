"""Host side of the node / flavor eligibility the engine takes as masks: string matching of taints against tolerations, of
PodSpec.NodeSelector and the required node affinity against labels. The device never sees strings; this module turns them into

  * the podset's excluded flavors (kq_heads.ps_flavor_ok) — checkFlavorForPodSets, scheduler/flavorassigner/flavorassigner.go:1243-1260
    (the flavor's NodeTaints against the podset's + the flavor's tolerations; flavorSelector :1282 against the flavor's NodeLabels), and
  * the podset's feasible leaves on a TAS flavor (kq_cycle_tas.ps_mask / leaf_mask, kq_tas_requests.leaf_ok) —
    cache/scheduler/scheduling_simulator_default.go:55-118 FindFeasibleNodes with the requirements of tas_flavor_snapshot.go:955-985
    (tolerations = the podset's + the TAS flavor's; the NodeSelector only when the lowest level is the node; the required affinity).

k8s semantics restated: corev1 Toleration.ToleratesTaint (k8s.io/api/core/v1/toleration.go), component-helpers
nodeaffinity.NodeSelector.Match (terms ORed, expressions ANDed, an empty term matches nothing), labels.Selector operators."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence


@dataclass
class Taint:
    key: str
    value: str = ""
    effect: str = "NoSchedule"


@dataclass
class Toleration:
    key: str = ""
    operator: str = "Equal"     # "Equal" | "Exists" ("" = Equal)
    value: str = ""
    effect: str = ""            # "" = every effect


@dataclass
class NodeSelectorRequirement:
    key: str
    operator: str               # In | NotIn | Exists | DoesNotExist | Gt | Lt
    values: List[str] = field(default_factory=list)


@dataclass
class NodeSelectorTerm:
    match_expressions: List[NodeSelectorRequirement] = field(default_factory=list)
    match_fields: List[NodeSelectorRequirement] = field(default_factory=list)   # metadata.name only


def tolerates(tol: Toleration, taint: Taint) -> bool:
    """Toleration.ToleratesTaint"""
    if tol.effect and tol.effect != taint.effect:
        return False
    if tol.key and tol.key != taint.key:
        return False
    op = tol.operator or "Equal"
    if op == "Exists":
        return True
    if op == "Equal":
        return tol.value == taint.value
    return False


def untolerated_taint(taints: Sequence[Taint], tolerations: Sequence[Toleration]) -> Optional[Taint]:
    """corev1helpers.FindMatchingUntoleratedTaint over the scheduling taints (NoSchedule / NoExecute: utiltaints.IsSchedulingTaint,
    flavorassigner.go:1244-1246): the first taint nothing tolerates."""
    for t in taints:
        if t.effect not in ("NoSchedule", "NoExecute"):
            continue
        if not any(tolerates(tol, t) for tol in tolerations):
            return t
    return None


def _requirement_matches(r: NodeSelectorRequirement, labels: Dict[str, str]) -> bool:
    has = r.key in labels
    v = labels.get(r.key)
    if r.operator == "In":
        return has and v in r.values
    if r.operator == "NotIn":
        return not has or v not in r.values
    if r.operator == "Exists":
        return has
    if r.operator == "DoesNotExist":
        return not has
    if r.operator in ("Gt", "Lt"):
        if not has or len(r.values) != 1:
            return False
        try:
            a, b = int(v), int(r.values[0])
        except ValueError:
            return False
        return a > b if r.operator == "Gt" else a < b
    return False


def terms_match(terms: Sequence[NodeSelectorTerm], labels: Dict[str, str], name: str = "") -> bool:
    """nodeaffinity.NodeSelector.Match: any term whose expressions and fields all match; a term without either matches nothing."""
    for t in terms:
        if not t.match_expressions and not t.match_fields:
            continue
        if all(_requirement_matches(r, labels) for r in t.match_expressions) and \
           all(_requirement_matches(r, {"metadata.name": name}) for r in t.match_fields):
            return True
    return False


def flavor_mismatch(node_selector: Dict[str, str], required_terms: Optional[Sequence[NodeSelectorTerm]], tolerations: Sequence[Toleration],
                    flavor_labels: Dict[str, str], flavor_taints: Sequence[Taint], flavor_tolerations: Sequence[Toleration]) -> Optional[str]:
    """The taint / affinity half of checkFlavorForPodSets (flavorassigner.go:1243-1260) for one podset and one flavor -> the reason the
    flavor is not eligible, None when it is."""
    t = untolerated_taint(flavor_taints, list(tolerations) + list(flavor_tolerations))
    if t is not None:
        return f"untolerated taint {t.key}={t.value}:{t.effect}"
    keys = set(flavor_labels)
    # flavorSelector :1282: only this flavor's own label keys count; a term emptied by that makes the affinity match every flavor
    sel = {k: v for k, v in (node_selector or {}).items() if k in keys}
    terms: List[NodeSelectorTerm] = []
    for term in required_terms or []:
        kept = [e for e in term.match_expressions if e.key in keys]
        if not kept:
            terms = []
            break
        terms.append(NodeSelectorTerm(kept))
    if any(flavor_labels.get(k) != v for k, v in sel.items()):
        return "doesn't match node affinity"
    if terms and not terms_match(terms, flavor_labels):
        return "doesn't match node affinity"
    return None


def leaf_mask(leaf_nodes: Sequence[Optional[object]], lowest_is_node: bool, tolerations: Sequence[Toleration], node_selector: Dict[str, str],
              required_terms: Optional[Sequence[NodeSelectorTerm]], node_taints: Dict[str, Sequence[Taint]]) -> Optional[List[int]]:
    """FindFeasibleNodes for every leaf of a TAS flavor (scheduling_simulator_default.go:55-118). leaf_nodes[i] = the tas.Node of leaf i
    (None: the leaf is not a node — a lowest level above the hostname: always feasible, :77). -> one 0/1 per leaf, None when nothing is
    excluded (no mask row needed)."""
    out = []
    for n in leaf_nodes:
        if n is None:
            out.append(1)
            continue
        ok = untolerated_taint(node_taints.get(n.name, ()), tolerations) is None
        if ok and lowest_is_node and node_selector:   # tas_flavor_snapshot.go:963: labels.Everything() above the hostname
            ok = all(n.labels.get(k) == v for k, v in node_selector.items())
        if ok and required_terms is not None:          # NodeSelector{} with no terms matches nothing
            ok = terms_match(required_terms, n.labels, n.name)
        out.append(1 if ok else 0)
    return None if all(out) else out

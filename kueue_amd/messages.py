"""Regeneration of the reference's user-visible status text from the engine's reason records (kq_decisions.rsn_*).

The engine reports WHY a flavor was not assigned as operands, not as text; the host side rebuilds exactly the strings the Go
code formats (this module is the Python mirror of shim/go/messages.go):

  flavorassigner.go:1352-1383  fitsResourceQuota        "insufficient quota for ..." / "insufficient unused quota for ..."
  flavorassigner.go:1080       findFlavorForPodSets     "resource %s unavailable in ClusterQueue"
  flavorassigner.go:1097       nomination mapping       "skipping flavor %s as it is not found in the nomination mapping for resource %s"
  flavorassigner.go:1224-1256  checkFlavorForPodSets    host-evaluated (taints / node affinity): the caller supplies the text
  flavorassigner.go:354-363    Status.Message           reasons sorted, joined with ", "
  flavorassigner.go:229-247    Assignment.Message       "couldn't assign flavors to pod set %s: %s" joined with "; "
  scheduler.go:470-481         inadmissibleMsg selectors for skipped entries
  resources/resource_formatter.go:66-97 + k8s.io/apimachinery resource.Quantity canonical form (quantity.go:425-462, amount.go:257-293)
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

from . import _ffi as F

RSN_EXCEEDS_MAX_CAPACITY = 1   # a = previously considered podsets requests, b = current podset request, c = maximum capacity
RSN_INSUFFICIENT_UNUSED = 2    # a = val - available ("more needed")
RSN_NOT_IN_NOMINATION = 3      # flavor skipped by the nomination mapping of a recomputation; resource = the scan's resource
RSN_FLAVOR_INELIGIBLE = 4      # checkFlavorForPodSets failed on the host (ps_flavor_ok bit clear)
RSN_RESOURCE_UNAVAILABLE = 5   # no resource group of the ClusterQueue covers the resource
RSN_SLICE_FLAVOR_MISMATCH = 6  # workload slices: flavor = the flavor tried, a = the replaced slice's flavor for the resource
RSN_TAS_FAILURE = 200          # kq_cycle_run_tas: the podset's TAS placement failed on the snapshot; a = KQ_TAS_* status, b / c = its operands
RSN_TRUNCATED = 255

UNLIMITED = (1 << 63) - 1

_DEC_SUFFIX = {-9: "n", -6: "u", -3: "m", 0: "", 3: "k", 6: "M", 9: "G", 12: "T", 15: "P", 18: "E"}
_BIN_SUFFIX = ["", "Ki", "Mi", "Gi", "Ti", "Pi", "Ei"]


def _decimal_canonical(value: int, scale: int) -> str:
    """int64Amount.AsCanonicalBytes (amount.go:257-281) + the DecimalSI suffix."""
    if value == 0:
        return "0"
    mant, exp = value, scale
    while mant % 10 == 0:
        mant //= 10
        exp += 1
    r = exp % 3  # Python's % is non-negative: r in {0, 1, 2} covers the Go cases (1, -2) and (2, -1)
    if r == 1:
        mant *= 10; exp -= 1
    elif r == 2:
        mant *= 100; exp -= 2
    suf = _DEC_SUFFIX.get(exp)
    return f"{mant}{suf}" if suf is not None else f"{mant}e{exp}"


def _binary_canonical(v: int) -> str:
    """Quantity{Format: BinarySI}.String() for an integer value (quantity.go:434-461, amount.go:286-293)."""
    if v == 0:
        return "0"
    if -1024 < v < 1024:
        return _decimal_canonical(v, 0)
    mant, exp = v, 0
    while mant % 1024 == 0:
        mant //= 1024
        exp += 1
    return f"{mant}{_BIN_SUFFIX[exp]}"


def _parse_then_string(s: str) -> str:
    """newCanonicalQuantity (resource_formatter.go:78-85): ParseQuantity(preferred.String()).String(). A string with a binary suffix
    parses back as BinarySI and prints unchanged; a plain integer parses as DecimalSI and prints in decimal canonical form."""
    for suf in _BIN_SUFFIX[1:]:
        if s.endswith(suf):
            return s
    for exp, suf in _DEC_SUFFIX.items():
        if suf and s.endswith(suf):
            return s
    return _decimal_canonical(int(s), 0)


def quantity_string(resource: str, v: int, binary_resources=()) -> str:
    """ResourceFormatter.ResourceQuantityString (resource_formatter.go:66-91)."""
    if resource == "cpu":
        return _decimal_canonical(v, -3)
    if resource in ("memory", "ephemeral-storage") or resource.startswith("hugepages-") or resource in binary_resources:
        return _parse_then_string(_binary_canonical(v))
    return _decimal_canonical(v, 0)


def amount_string(resource: str, a: int, binary_resources=()) -> str:
    """ResourceFormatter.AmountQuantityString (:93-97); Unlimited.String() is "<unlimited>" (amount.go:204-209)."""
    if a == UNLIMITED:
        return "<unlimited>"
    return quantity_string(resource, a, binary_resources)


def second_pass_names(adm, podset: int, a: int):
    """(stale_domain, unhealthy_node) for tas_failure_text from a head's Status (kueue_amd/tas_cycle.py HeadAdmission): operand a of a
    KQ_TAS_STALE record indexes the podset's TopologyAssignment AFTER deleteDomain dropped the unhealthy node's domain
    (tas_flavor_snapshot.go:693, :828-840); IsTopologyAssignmentStale reports domain.Values[0] (:821)."""
    node = adm.unhealthy_nodes[0] if adm.unhealthy_nodes else ""
    doms = [v for v, _ in (adm.domains[podset] or []) if v[-1] != node]
    return (doms[a][0] if 0 <= a < len(doms) else ""), node


def tas_failure_text(topology_name: str, status: int, a: int, b: int, slice_size: int = 1, stale_domain: str = "", unhealthy_node: str = "") -> str:
    """TASAssignmentsResult.Failure().Reason from the operands of a KQ_RSN_TAS_FAILURE record: notFitMessage
    (tas_flavor_snapshot.go:1997) and the fixed strings of findTopologyAssignment (:886-947). The node-exclusion statistics the
    reference appends to the "doesn't allow to fit any" form ("Total nodes: N; excluded: ...") are not carried by the operands."""
    from . import tas as T
    if status == T.TAS_NOT_FIT:
        unit = "pod" if slice_size == 1 else "slice"
        if a == 0:
            return f'topology "{topology_name}" doesn\'t allow to fit any of {b} {unit}(s)'
        return f'topology "{topology_name}" allows to fit only {a} out of {b} {unit}(s)'
    if status == T.TAS_NOT_FIT_LAYERS:
        return f'topology "{topology_name}" doesn\'t allow to fit'
    # the second pass of kq_cycle_run_tas (findReplacementAssignment tas_flavor_snapshot.go:694-696, :727): the caller knows the names — operand
    # a of KQ_TAS_STALE is the index of the stale domain in the admission's TopologyAssignment after deleteDomain (its first level value is
    # what the reference prints), the unhealthy node is Status.UnhealthyNodes[0].Name
    if status == T.TAS_STALE:
        return f"Cannot replace the node, because the existing topologyAssignment is invalid, as it contains the stale domain {stale_domain or f'#{a}'}"
    if status == T.TAS_NO_REPLACEMENT:
        return f"cannot find replacement assignment for unhealthy node: {unhealthy_node}"
    return {T.TAS_NO_LEVEL: "no requested topology level", T.TAS_SLICE_ABOVE: "podset slice topology is above the podset topology",
            T.TAS_BAD_SLICE_SIZE: "slice topology requested, but slice size not provided"}.get(status, f"topology-aware placement failed (status {status})")


def reason_text(snap, code: int, flavor: int, resource: int, a: int, b: int, c: int,
                ineligible: Optional[Callable[[int, int], List[str]]] = None, podset: int = 0,
                tas: Optional[Callable[[int, int, int, int, int], str]] = None) -> List[str]:
    """One record -> the reason string(s) the reference appends to Status.reasons."""
    fl = snap.flavors[flavor] if flavor >= 0 else ""
    rs = snap.resources[resource] if resource >= 0 else ""
    if code == RSN_EXCEEDS_MAX_CAPACITY:
        return [f"insufficient quota for {rs} in flavor {fl}, previously considered podsets requests ({amount_string(rs, a)}) + "
                f"current podset request ({quantity_string(rs, b)}) > maximum capacity ({amount_string(rs, c)})"]
    if code == RSN_INSUFFICIENT_UNUSED:
        return [f"insufficient unused quota for {rs} in flavor {fl}, {amount_string(rs, a)} more needed"]
    if code == RSN_NOT_IN_NOMINATION:
        return [f"skipping flavor {fl} as it is not found in the nomination mapping for resource {rs}"]
    if code == RSN_FLAVOR_INELIGIBLE:
        return list(ineligible(podset, flavor)) if ineligible else [f"flavor {fl} is not eligible for the pod set"]
    if code == RSN_RESOURCE_UNAVAILABLE:
        return [f"resource {rs} unavailable in ClusterQueue"]
    if code == RSN_TAS_FAILURE:   # flavorassigner.go:875; tas(podset, flavor, status, a, b): the caller knows the topology's name and the slice size
        return [tas(podset, flavor, a, b, c) if tas else tas_failure_text(fl, a, b, c)]
    if code == RSN_SLICE_FLAVOR_MISMATCH:   # flavorassigner.go:1134
        return [f"could not assign {fl} flavor since the original workload is assigned: {snap.flavors[a] if a >= 0 else ''}"]
    raise ValueError(code)


def podset_reasons(dec, i: int, ineligible=None, tas=None) -> List[List[str]]:
    """Per podset of head i: Status.reasons, sorted as Status.Message sorts them (flavorassigner.go:361)."""
    snap, heads = dec.snap, dec.heads
    nps = int(heads.arrays["ps_off"][i + 1] - heads.arrays["ps_off"][i])
    out: List[List[str]] = [[] for _ in range(nps)]
    a = dec.a
    for k in range(int(a["rsn_off"][i]), int(a["rsn_off"][i + 1])):
        code = int(a["rsn_code"][k])
        if code == RSN_TRUNCATED:
            raise OverflowError("reason window of the head overflowed: raise rsn_cap")
        ps = int(a["rsn_podset"][k])
        out[ps].extend(reason_text(snap, code, int(a["rsn_flavor"][k]), int(a["rsn_resource"][k]), int(a["rsn_a"][k]), int(a["rsn_b"][k]),
                                   int(a["rsn_c"][k]), ineligible, ps, tas))
    return [sorted(x) for x in out]


def assignment_message(dec, i: int, podset_names: List[str], ineligible=None) -> str:
    """Assignment.Message (flavorassigner.go:229-247)."""
    parts = []
    for name, reasons in zip(podset_names, podset_reasons(dec, i, ineligible)):
        if reasons:
            parts.append(f"couldn't assign flavors to pod set {name}: " + ", ".join(reasons))
    return "; ".join(parts)


def inadmissible_message(dec, i: int, podset_names: List[str], ineligible=None) -> str:
    """entry.inadmissibleMsg as schedule() leaves it (scheduler.go:281-295, 248-253, 452-481) for a head that was not admitted."""
    a = dec.a
    if int(a["skip"][i]) == F.SKIP_OVERLAP:
        return "Workload has overlapping preemption targets with another workload"
    if int(a["skip"][i]) == F.SKIP_NO_LONGER_FITS:
        return "Workload no longer fits after processing another workload"
    if int(a["mode"][i]) == 2:  # DeferredFit (scheduler.go:455)
        return "Workload has overlapping preemption targets with another workload, but will fit after these preemptions complete"
    msg = assignment_message(dec, i, podset_names, ineligible)
    if int(a["action"][i]) == F.ACT_PREEMPT:  # markPreemptionOutcome :291-295 (every eviction assumed to succeed)
        n = int(a["tgt_off"][i + 1] - a["tgt_off"][i])
        msg += f". Pending the preemption of {n} workload(s)"
    return msg

#!/usr/bin/env python
"""Transcribes the table of TestAssignFlavors (and siblings) from the reference into YAML fixtures.

  python tests/golden/extract_assign_flavors.py   # needs /root/reference (this container only)

The Go tables are regular builder chains (pkg/util/testing/v1beta2 wrappers); this script parses
them textually. Cases using features outside the boundary of this engine (TAS, workload slices,
node affinity / selectors, reclaimable pods, missing ResourceFlavors) are skipped and listed.
Taints are honoured through the `excludedFlavors` host-side eligibility mask (the two tainted
flavors of the table: "tainted" is never tolerated by the test podsets, "taint_and_toleration"
tolerates itself).
"""
import os
import re
import sys

import yaml

REF = "/root/reference/pkg/scheduler/flavorassigner/flavorassigner_test.go"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assign_flavors.yaml")

RES = {"corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory", "corev1.ResourcePods": "pods"}


def res_name(tok):
    tok = tok.strip()
    if tok in RES:
        return RES[tok]
    m = re.match(r'^"([^"]+)"$', tok)
    if m:
        return m.group(1)
    raise ValueError(tok)


def match_brace(s, i, open_ch="{", close_ch="}"):
    """index of the brace closing s[i] (s[i] == open_ch); skips string literals"""
    depth, j, n = 0, i, len(s)
    while j < n:
        c = s[j]
        if c == '"':
            j += 1
            while s[j] != '"':
                j += 2 if s[j] == "\\" else 1
        elif c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return j
        j += 1
    raise ValueError("unbalanced")


def field(block, name):
    """text of `name: <value>` where value is a balanced expression up to the next top-level comma"""
    m = re.search(r"(?m)^\t{3}" + re.escape(name) + r":\s*", block)
    if not m:
        return None
    i = m.end()
    depth, j = 0, i
    while j < len(block):
        c = block[j]
        if c == '"':
            j += 1
            while block[j] != '"':
                j += 2 if block[j] == "\\" else 1
        elif c in "{([":
            depth += 1
        elif c in "})]":
            depth -= 1
        elif c == "," and depth == 0:
            break
        j += 1
    return block[i:j]


def parse_amount(expr):
    expr = expr.strip().replace("_", "")
    expr = expr.replace("utiltesting.Gi", str(1 << 30)).replace("utiltesting.Mi", str(1 << 20)).replace("utiltesting.Ki", str(1 << 10))
    return int(eval(expr, {"__builtins__": {}}))


def parse_frq(text):
    out = {}
    if not text:
        return out
    for m in re.finditer(r'\{Flavor:\s*"([^"]+)",\s*Resource:\s*([^}]+)\}:\s*resources\.NewAmount\(([^)]*)\)', text):
        out[f"{m.group(1)}/{res_name(m.group(2))}"] = parse_amount(m.group(3))
    return out


def parse_cq(text):
    if text is None:
        return None
    name = re.search(r'MakeClusterQueue\("([^"]+)"\)', text).group(1)
    cq = {"name": name}
    m = re.search(r'\.\s*Cohort\("([^"]*)"\)', text)
    if m and m.group(1):
        cq["cohort"] = m.group(1)
    rgs = []
    for m in re.finditer(r"ResourceGroup\(", text):
        j = match_brace(text, m.end() - 1, "(", ")")
        body = text[m.end():j]
        flavors = []
        for fm in re.finditer(r'MakeFlavorQuotas\("([^"]+)"\)', body):
            nxt = re.search(r"MakeFlavorQuotas\(", body[fm.end():])
            seg = body[fm.end(): fm.end() + nxt.start()] if nxt else body[fm.end():]
            resources = {}
            for r in re.finditer(r'\.\s*Resource\(([^,]+?)((?:,\s*"[^"]*")*)\)', seg):
                vals = re.findall(r'"([^"]*)"', r.group(2))
                resources[res_name(r.group(1))] = (vals + ["", "", ""])[:3]
            for r in re.finditer(r"ResourceQuotaWrapper\(([^)]+)\)((?:\.\s*\w+\(\"[^\"]*\"\))*)\.\s*Append\(\)", seg):
                q = ["0", "", ""]
                for call, v in re.findall(r'\.\s*(\w+)\("([^"]*)"\)', r.group(2)):
                    q[{"NominalQuota": 0, "BorrowingLimit": 1, "LendingLimit": 2}[call]] = v
                resources[res_name(r.group(1))] = q
            flavors.append({"flavor": fm.group(1), "resources": resources})
        rgs.append(flavors)
    cq["resourceGroups"] = rgs
    pre = {}
    m = re.search(r"Preemption\(kueue\.ClusterQueuePreemption\{", text)
    if m:
        j = match_brace(text, m.end() - 1)
        body = text[m.end():j]
        for key, yk in (("WithinClusterQueue", "withinClusterQueue"), ("ReclaimWithinCohort", "reclaimWithinCohort")):
            mm = re.search(key + r":\s*kueue\.PreemptionPolicy(\w+)", body)
            if mm:
                pre[yk] = mm.group(1)
        mm = re.search(r"Policy:\s*kueue\.BorrowWithinCohortPolicy(\w+)", body)
        if mm:
            pre["borrowWithinCohort"] = mm.group(1)
        mm = re.search(r"MaxPriorityThreshold:\s*(?:ptr\.To\[int32\]|ptr\.To|new)\((?:int32\()?(-?\d+)", body)
        if mm:
            pre["maxPriorityThreshold"] = int(mm.group(1))
    # no API defaulting in the reference's unit tests: an absent ReclaimWithinCohort is "" (not "Never"),
    # which canPreemptWhileBorrowing distinguishes under fair sharing (flavorassigner.go:1386-1389)
    pre.setdefault("reclaimWithinCohort", "")
    cq["preemption"] = pre
    m = re.search(r'FairWeight\(resource\.MustParse\("([^"]+)"\)\)', text)
    if m:
        cq["fairWeight"] = float(m.group(1))
    m = re.search(r"FlavorFungibility\(kueue\.FlavorFungibility\{([^}]*)\}\s*,?\s*\)", text)
    if m:
        fu = {}
        for key, yk in (("WhenCanBorrow", "whenCanBorrow"), ("WhenCanPreempt", "whenCanPreempt")):
            mm = re.search(key + r":\s*kueue\.(\w+)", m.group(1))
            if mm:
                fu[yk] = mm.group(1)
        mm = re.search(r"Preference:\s*(?:new|ptr\.To)\(kueue\.(\w+)\)", m.group(1))
        if mm:
            fu["preference"] = mm.group(1)
        cq["fungibility"] = fu
    return cq


TAINTED = {"tainted"}  # flavors whose taint no test podset tolerates by itself


_LABELS = None


def flavor_labels(src):
    """The ResourceFlavor objects of TestAssignFlavors (:179-207): flavor -> nodeLabels."""
    global _LABELS
    if _LABELS is None:
        _LABELS = {}
        head = src[src.index("func TestAssignFlavors("):]
        head = head[:head.index("cases := map[string]struct")]
        for m in re.finditer(r'MakeResourceFlavor\("([^"]+)"\)((?:\.\s*NodeLabel\("[^"]+",\s*"[^"]*"\))*)', head):
            _LABELS[m.group(1)] = dict(re.findall(r'NodeLabel\("([^"]+)",\s*"([^"]*)"\)', m.group(2)))
    return _LABELS


def parse_podsets(text):
    pods = []
    for m in re.finditer(r"MakePodSet\(([^,]+),\s*(\d+)\)", text):
        nxt = re.search(r"MakePodSet\(", text[m.end():])
        seg = text[m.end(): m.end() + nxt.start()] if nxt else text[m.end():]
        nm = m.group(1).strip()
        nm = "main" if nm == "kueue.DefaultPodSetName" else nm.strip('"')
        ps = {"name": nm, "count": int(m.group(2)), "requests": {}}
        for r in re.finditer(r'\.\s*Request\(([^,]+),\s*"([^"]*)"\)', seg):
            ps["requests"][res_name(r.group(1))] = r.group(2)
        for c in re.finditer(r"SingleContainerForRequest\(map\[corev1\.ResourceName\]string\{([^}]*)\}", seg):   # Containers(...): the same requests, via the pod template
            for r in re.finditer(r'([\w.]+|"[^"]+"):\s*"([^"]*)"', c.group(1)):
                ps["requests"][res_name(r.group(1))] = r.group(2)
        if re.search(r"PodSetGroup", seg):
            return "group"
        if re.search(r"SetMinimumCount|TopologyRequest", seg):
            return None
        aff = re.search(r"RequiredDuringSchedulingIgnoredDuringExecution\(", seg)
        if aff:   # node affinity: terms ORed, expressions ANDed; only NodeSelectorOpIn appears in the table
            body = re.sub(r"//[^\n]*", "", seg[aff.end(): match_brace(seg, aff.end() - 1, "(", ")")])
            terms = []
            for t in body.split("MatchExpressions:")[1:]:
                exprs = re.findall(r'Key:\s*"([^"]+)",\s*Operator:\s*corev1\.NodeSelectorOp(\w+),\s*Values:\s*\[\]string\{([^}]*)\}', t)
                if any(op != "In" for _, op, _ in exprs):
                    return None
                terms.append([(k, re.findall(r'"([^"]*)"', v)) for k, _, v in exprs])
            ps["affinity_terms"] = terms
        elif re.search(r"Affinity", seg):
            return None
        sel = re.search(r"NodeSelector\(map\[string\]string\{([^}]*)\}\)", seg)
        if sel:   # spec.nodeSelector: evaluated against each flavor's OWN label keys below (flavorSelector flavorassigner.go:1264-1298)
            ps["node_selector"] = dict(re.findall(r'"([^"]+)":\s*"([^"]*)"', sel.group(1)))
        ps["tolerates_spot"] = bool(re.search(r"Toleration\(", seg))
        pods.append(ps)
    return pods


def parse_want(text):
    want = {"podsets": []}
    m = re.search(r"PodSets:\s*\[\]PodSetAssignment\{", text)
    if m:
        j = match_brace(text, m.end() - 1)
        body = text[m.end():j]
        # split into top-level { ... } elements
        i = 0
        while True:
            k = body.find("{", i)
            if k < 0:
                break
            e = match_brace(body, k)
            el = body[k + 1:e]
            fl = {}
            fm = re.search(r"Flavors:\s*ResourceAssignment\{", el)
            if fm:
                fe = match_brace(el, fm.end() - 1)
                for r in re.finditer(r'([\w\."/-]+):\s*(?:&FlavorAssignment)?\{Name:\s*"([^"]+)",\s*Mode:\s*(\w+)(?:,\s*TriedFlavorIdx:\s*(-?\d+))?\}', el[fm.end():fe]):
                    fl[res_name(r.group(1))] = [r.group(2), r.group(3), int(r.group(4) or 0)]
            cnt = re.search(r"(?m)^\s*Count:\s*(\d+)", el)
            ps_want = {"flavors": fl, "count": int(cnt.group(1)) if cnt else None}
            sm = re.search(r"Status:\s*\*NewStatus\(", el)
            if sm:  # PodSetAssignment.Status.reasons (flavorassigner.go:329-339): the strings Status.Message joins
                se = match_brace(el, sm.end() - 1, "(", ")")
                ps_want["status"] = [bytes(x, "utf-8").decode("unicode_escape") for x in re.findall(r'"((?:[^"\\]|\\.)*)"', el[sm.end():se])]
            want["podsets"].append(ps_want)
            i = e + 1
    m = re.search(r"(?m)^\t{4}Borrowing:\s*(\d+)", text)
    want["borrowing"] = int(m.group(1)) if m else 0
    m = re.search(r'(?m)^\t{4}NoFitReason:\s*"(\w*)"', text)   # Assignment.NoFitReason (the attempts' own labels sit deeper); compared when the gate is on
    if m:
        want["noFitReason"] = m.group(1)
    m = re.search(r"Usage:\s*workload\.Usage\{", text)
    want["usage"] = parse_frq(text[m.start():]) if m else {}
    return want


def main():
    src = open(REF).read()
    start = src.index("func TestAssignFlavors(")
    table_start = src.index("cases := map[string]struct", start)
    body_start = src.index("}{", table_start) + 1
    body_end = match_brace(src, body_start)
    table = src[body_start + 1: body_end]
    cases, skipped = [], []
    for m in re.finditer(r'(?m)^\t\t"((?:[^"\\]|\\.)*)":\s*\{', table):
        j = match_brace(table, m.end() - 1)
        block = table[m.end():j]
        name = m.group(1)
        line = src[: body_start + 1 + m.start()].count("\n") + 1
        if re.search(r"preemptWorkloadSlice|topologies|TopologyRequest|tas-|DelayedTopology", block):
            skipped.append((name, "TAS / workload slices / reclaimable pods: outside the engine boundary")); continue
        pods = parse_podsets(field(block, "wlPods") or "")
        if pods == "group":
            skipped.append((name, "PodSetGroupName: one flavor scan per group (flavorassigner.go:782-860) - by hand in assign_flavors_groups_manual.yaml, "
                                  "pinned on the oracle's grouped path; the engine scans per podset (DESIGN section 7)")); continue
        if pods is None:
            skipped.append((name, "node selector / affinity / minimum count (host-side eligibility, not transcribed)")); continue
        cq = parse_cq(field(block, "clusterQueue"))
        cq2 = parse_cq(field(block, "secondaryClusterQueue"))
        gates = {}
        fg = field(block, "featureGates")
        if fg:
            for g, v in re.findall(r"features\.(\w+):\s*(true|false)", fg):
                gates[g] = v == "true"
            if any(g not in ("FlavorFungibility", "QuotaCheckStrategy", "ReclaimablePods") for g in gates):
                skipped.append((name, f"feature gate {list(gates)} not modelled")); continue
        # Status.ReclaimablePods: workload.NewInfo subtracts them from the podset's count before Assign sees it (workload.go totalRequestsFromPodSets,
        # gate ReclaimablePods) — the host side of the boundary; the row is transcribed with the count the scheduler is handed
        rp = field(block, "wlReclaimablePods")
        if rp and gates.pop("ReclaimablePods", True):
            for r in re.finditer(r"Name:\s*([^,]+),\s*Count:\s*(\d+)", rp):
                nm = "main" if r.group(1).strip() == "kueue.DefaultPodSetName" else r.group(1).strip().strip('"')
                for ps in pods:
                    if ps["name"] == nm:
                        ps["count"] -= int(r.group(2)); ps["reclaimed"] = int(r.group(2))
        gates.pop("ReclaimablePods", None)
        all_flavors = {f["flavor"] for q in (cq, cq2) if q for rg in q["resourceGroups"] for f in rg}
        if "nonexistent-flavor" in all_flavors or "non-existent" in " ".join(all_flavors):
            skipped.append((name, "missing ResourceFlavor object")); continue
        inel = {}
        for ps in pods:
            excl = [f for f in sorted(all_flavors) if f in TAINTED and not ps["tolerates_spot"]]
            sel = ps.pop("node_selector", None)
            terms = ps.pop("affinity_terms", None)
            if sel or terms:
                for f in sorted(all_flavors):
                    labels = flavor_labels(src).get(f, {})
                    ok = not any(k in labels and labels[k] != v for k, v in (sel or {}).items())
                    # flavorSelector :1264-1298: every term keeps only the expressions on the flavor's OWN label keys; a term left empty matches
                    # any flavor (terms are ORed) and the affinity reduces to spec.nodeSelector
                    kept = [[(k, vs) for k, vs in t if k in labels] for t in (terms or [])]
                    if kept and all(kept):
                        ok = ok and any(all(labels[k] in vs for k, vs in t) for t in kept)
                    if f not in excl and not ok:
                        excl.append(f)
                        inel[f] = f"flavor {f} doesn't match node affinity"   # checkFlavorForPodSets :1256
                if sel:
                    ps["nodeSelector"] = sel
                if terms:
                    ps["nodeAffinityTerms"] = [{k: vs for k, vs in t} for t in terms]
            if excl:
                ps["excludedFlavors"] = sorted(excl)
            del ps["tolerates_spot"]
        sim = {}
        st = field(block, "simulationResult")
        if st:
            for r in re.finditer(r'\{Flavor:\s*"([^"]+)",\s*Resource:\s*([^}]+)\}:\s*\{preemptioncommon\.(\w+),\s*(\d+)\}', st):
                sim[f"{r.group(1)}/{res_name(r.group(2))}"] = [r.group(3), int(r.group(4))]
        rep = field(block, "wantRepMode")
        want = parse_want(field(block, "wantAssignment") or "")
        want["repMode"] = rep.strip() if rep else "NoFit"
        cq["usageRaw"] = {k: v for k, v in parse_frq(field(block, "clusterQueueUsage")).items()}
        cqs = [cq]
        if cq2:
            cq2["usageRaw"] = {k: v for k, v in parse_frq(field(block, "secondaryClusterQueueUsage")).items()}
            cqs.append(cq2)
        case = {"name": name, "ref": f"pkg/scheduler/flavorassigner/flavorassigner_test.go:{line}", "clusterQueues": cqs,
                "pending": [{"name": "wl", "cq": cq["name"], "podsets": pods}], "want": want}
        if sim:
            case["simulationResult"] = sim
        if inel:
            case["ineligibleText"] = inel
        if gates:
            case["gates"] = gates
        efs = field(block, "enableFairSharing")
        if efs and efs.strip() == "true":
            case["fairSharing"] = True
        cases.append(case)
    with open(OUT, "w") as f:
        f.write("# GENERATED by tests/golden/extract_assign_flavors.py from\n# /root/reference/pkg/scheduler/flavorassigner/flavorassigner_test.go (TestAssignFlavors :178)\n")
        f.write("# usage quantities are raw int64 (milli-CPU / bytes); quota strings are resource.Quantity\n")
        yaml.safe_dump({"cases": cases, "skipped": [{"name": n, "why": w} for n, w in skipped]}, f, sort_keys=False, width=200)
    print(f"{len(cases)} cases transcribed, {len(skipped)} skipped -> {OUT}")
    for n, w in skipped:
        print("  skipped:", n, "--", w)


if __name__ == "__main__":
    main()

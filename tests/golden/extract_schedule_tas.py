#!/usr/bin/env python
"""Transcribes the whole-cycle TAS tables of pkg/scheduler/scheduler_tas_test.go — TestScheduleForTAS (:58),
TestScheduleForTASPreemption (:4121), TestScheduleForTASCohorts (:5950) — into tests/golden/schedule_tas.yaml.

  python tests/golden/extract_schedule_tas.py      # needs /root/reference (this container only)

A Go case = Nodes, Topologies, ResourceFlavors, ClusterQueues (+ Cohorts), Workloads (admitted ones carry a TopologyAssignment, the
others are pending in a LocalQueue) and, after exactly ONE Scheduler.schedule(): wantNewAssignments (flavors + TopologyAssignment of
every new admission), wantLeft / wantInadmissibleLeft. The fixture keeps the inputs, the heads (one per ClusterQueue: priority desc,
creation asc) and the expected per-head outcome. Cases needing machinery outside the boundary (admission checks / delayed topology,
unhealthy-node replacement, taints / tolerations / node selectors / affinity, non-TAS pods, workload slices, podset groups, balanced
placement, multi-layer slices, resource transformations, gates) are skipped and counted in the YAML header.
"""
import collections
import os
import re
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # (kueue_amd.api: quantity parsing)

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_assign_flavors import match_brace, parse_cq, res_name  # noqa: E402
from extract_preemption import NOW, chain, parse_cohort, parse_time, split_top  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/pkg/scheduler/scheduler_tas_test.go"
HOST = "kubernetes.io/hostname"


class Skip(Exception):
    pass


def field(block, name, tabs=3):
    m = re.search(r"(?m)^\t{%d}%s:\s*" % (tabs, re.escape(name)), block)
    if not m:
        return None
    i = m.end()
    # value runs to the top-level comma
    depth, j, n = 0, i, len(block)
    while j < n:
        c = block[j]
        if c == '"':
            j += 1
            while block[j] != '"':
                j += 2 if block[j] == "\\" else 1
        elif c in "({[":
            depth += 1
        elif c in ")}]":
            depth -= 1
        elif c == "," and depth == 0:
            break
        j += 1
    return block[i:j]


def strip_comments(src):
    src = "\n".join("" if l.strip().startswith("//") else l for l in src.split("\n"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"(?m)\s//[^\"\n]*$", "", src)


def func_body(src, name):
    start = src.index("func %s(" % name)
    b = src.index("{", src.index(")", start))
    return src[b + 1: match_brace(src, b)], src[:b + 1].count("\n")


def symbols(body):
    """top-level `name := expr` / const name = "..." of a test function, before its table"""
    sym = {}
    for m in re.finditer(r'(?m)^\t\t?(\w+)\s*=\s*"([^"]*)"$', body):
        sym[m.group(1)] = '"%s"' % m.group(2)
    lines = body.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^\t(\w+) := (.*)$", lines[i])
        if not m or m.group(1) in ("cases", "testCases"):
            i += 1
            continue
        expr = m.group(2)
        while expr.count("(") != expr.count(")") or expr.count("{") != expr.count("}") or expr.rstrip().endswith("."):
            i += 1
            expr += "\n" + lines[i]
        sym[m.group(1)] = expr
        i += 1
    return sym


def label(tok, sym):
    tok = tok.strip()
    if tok == "corev1.LabelHostname":
        return HOST
    if tok in sym:
        return sym[tok].strip('"')
    m = re.match(r'^"([^"]*)"$', tok)
    if m:
        return m.group(1)
    raise Skip("label token " + tok)


def resolve(expr, sym, depth=0):
    """substitute a bare identifier (optionally `*x`, `x.DeepCopy()`, `*x.DeepCopy()`, `x.Obj()`) by its definition"""
    e = expr.strip()
    m = re.match(r"^\*?(\w+)(?:\.DeepCopy\(\))?(?:\.\s*Obj\(\))?$", e)
    if m and m.group(1) in sym and depth < 6:
        return resolve(sym[m.group(1)], sym, depth + 1)
    return e


def items_of(expr, sym):
    """elements of a `[]T{...}` literal (or of a variable holding one), each resolved"""
    e = resolve(expr, sym)
    if "{" not in e:
        raise Skip("not a literal: " + e[:40])
    p = e.index("{")
    return [resolve(t, sym) for t in split_top(e[p + 1: match_brace(e, p)]) if t.strip()]


def _kv(body):
    """`Key: x, Operator: y, ...` of a Go struct literal -> {field: token} (top level only)"""
    out = {}
    for part in split_top(body):
        if ":" in part:
            k, v = part.split(":", 1)
            out[k.strip()] = v.strip()
    return out


def _tok(t):
    t = t.strip()
    if t.startswith('"'):
        return t.strip('"')
    return {"corev1.TaintEffectNoSchedule": "NoSchedule", "corev1.TaintEffectNoExecute": "NoExecute", "corev1.TaintEffectPreferNoSchedule": "PreferNoSchedule",
            "corev1.TolerationOpEqual": "Equal", "corev1.TolerationOpExists": "Exists", "corev1.NodeSelectorOpIn": "In", "corev1.NodeSelectorOpNotIn": "NotIn",
            "corev1.NodeSelectorOpExists": "Exists", "corev1.NodeSelectorOpDoesNotExist": "DoesNotExist", "corev1.LabelHostname": HOST}.get(t) or _bad_tok(t)


def _bad_tok(t):
    raise Skip("not a literal: " + t[:40])


def parse_taints(a):
    """corev1.Taint{Key: .., Value: .., Effect: ..} literals of a Taints(...) / Taint(...) call"""
    out = []
    for m in re.finditer(r"corev1\.Taint\{", a):
        j = match_brace(a, m.end() - 1)
        f = _kv(a[m.end():j])
        out.append(dict(key=_tok(f.get("Key", '""')), value=_tok(f.get("Value", '""')), effect=_tok(f.get("Effect", '""'))))
    return out


def parse_tolerations(a):
    out = []
    for m in re.finditer(r"corev1\.Toleration\{", a):
        j = match_brace(a, m.end() - 1)
        f = _kv(a[m.end():j])
        out.append(dict(key=_tok(f.get("Key", '""')), operator=_tok(f.get("Operator", '"Equal"')), value=_tok(f.get("Value", '""')), effect=_tok(f.get("Effect", '""'))))
    return out


def parse_terms(a):
    """[]corev1.NodeSelectorTerm{{MatchExpressions: []corev1.NodeSelectorRequirement{{Key:, Operator:, Values: []string{..}}}}}"""
    p = a.index("{", a.index("NodeSelectorTerm"))
    terms = []
    for t in split_top(a[p + 1:match_brace(a, p)]):
        t = t.strip()
        if not t:
            continue
        f = _kv(t[t.index("{") + 1:match_brace(t, t.index("{"))])
        if set(f) - {"MatchExpressions"}:
            raise Skip("NodeSelectorTerm." + ",".join(sorted(set(f) - {"MatchExpressions"})))
        exprs = []
        me = f.get("MatchExpressions")
        if me:
            q = me.index("{")
            for e in split_top(me[q + 1:match_brace(me, q)]):
                e = e.strip()
                if not e:
                    continue
                g = _kv(e[e.index("{") + 1:match_brace(e, e.index("{"))])
                vals = re.findall(r'"([^"]*)"', g.get("Values", ""))
                exprs.append(dict(key=_tok(g["Key"]), operator=_tok(g["Operator"]), values=vals))
        terms.append(dict(matchExpressions=exprs))
    return terms


def parse_node(text, sym):
    if "MakeNode" not in text:
        # `*singleNode.Clone().StatusAllocatable(...).Obj()`: a builder variable (a MakeNode chain without Obj) and the calls made on its copy —
        # the variable's own chain first, then those (StatusAllocatable merges, testingjobs/node/wrappers.go:82)
        m = re.match(r"^\s*\*?(\w+)\s*\.", text)
        if m and m.group(1) in sym and "MakeNode" in sym[m.group(1)]:
            text = sym[m.group(1)].rstrip() + text[m.end() - 1:]
    calls, _ = chain(text, text.index("MakeNode"))
    n = {"name": calls[0][1].strip().strip('"'), "labels": {}, "allocatable": {}, "ready": False}
    for m, a in calls[1:]:
        if m == "Label":
            k, v = split_top(a)
            n["labels"][label(k, sym)] = label(v, sym)
        elif m == "StatusAllocatable":
            for r, q in re.findall(r'([\w\.]+|"[^"]+"):\s*resource\.MustParse\("([^"]+)"\)', a):
                n["allocatable"][res_name(r) if not r.startswith('"') else r.strip('"')] = q
        elif m == "Ready":
            n["ready"] = True
        elif m == "NotReady":
            n["ready"] = False
        elif m == "Unschedulable":
            n["unschedulable"] = True
        elif m == "Taints":
            n.setdefault("taints", []).extend(parse_taints(a))
        elif m in ("Obj", "DeepCopy", "Clone"):
            pass
        else:
            raise Skip("Node." + m)
    return n


def parse_pods(items):
    """[]corev1.Pod -> non-TAS usage {node: {resource: [quantities]}} (tas_non_tas_pod_cache.go: scheduled, not terminated pods)"""
    out = {}
    for text in items:
        calls, _ = chain(text, text.index("MakePod"))
        node, reqs, done = None, {}, False
        for m, a in calls[1:]:
            if m == "NodeName":
                node = a.strip().strip('"')
            elif m == "Request":
                r, q = split_top(a)
                reqs[res_name(r)] = q.strip().strip('"')
            elif m == "StatusPhase":
                done = "Succeeded" in a or "Failed" in a
            elif m not in ("Obj", "Clone"):
                raise Skip("Pod." + m)
        if node is None or done:
            continue
        d = out.setdefault(node, {})
        for r, q in reqs.items():
            d.setdefault(r, []).append(q)
        d.setdefault("pods", []).append("1")
    return out


def parse_topology(text, sym):
    m = re.search(r'MakeDefaultOneLevelTopology\("([^"]+)"\)', text)
    if m:
        return m.group(1), [HOST]
    calls, _ = chain(text, text.index("MakeTopology"))
    name = calls[0][1].strip().strip('"')
    levels = []
    for mm, a in calls[1:]:
        if mm == "Levels":
            levels = [label(x, sym) for x in split_top(a)]
        elif mm != "Obj":
            raise Skip("Topology." + mm)
    return name, levels


def parse_flavor(text, sym):
    calls, _ = chain(text, text.index("MakeResourceFlavor"))
    f = {"name": calls[0][1].strip().strip('"'), "nodeLabels": {}}
    for m, a in calls[1:]:
        if m == "NodeLabel":
            k, v = split_top(a)
            f["nodeLabels"][label(k, sym)] = label(v, sym)
        elif m == "TopologyName":
            f["topologyName"] = a.strip().strip('"')
        elif m == "Toleration":
            f.setdefault("tolerations", []).extend(parse_tolerations(a))
        elif m == "Taint":
            f.setdefault("nodeTaints", []).extend(parse_taints(a))
        elif m != "Obj":
            raise Skip("ResourceFlavor." + m)
    return f


def parse_topology_assignment(a, sym):
    """MakeTopologyAssignment(levels).Domain(MakeTopologyDomainAssignment([]string{...}, n).Obj())... -> [[values], count]"""
    doms = []
    for m in re.finditer(r"MakeTopologyDomainAssignment\(\[\]string\{([^}]*)\},\s*(\d+)\)", a):
        doms.append([[label(x, sym) for x in split_top(m.group(1)) if x.strip()], int(m.group(2))])
    return {"domains": doms}


def parse_admission(args, sym):
    calls, _ = chain(args, args.index("MakeAdmission"))
    cq = split_top(calls[0][1])[0].strip().strip('"')
    podsets = []
    for name, a in calls[1:]:
        if name == "PodSets":
            for psa in split_top(a):
                pc, _ = chain(psa, psa.index("MakePodSetAssignment"))
                ps = {"name": pc[0][1].strip().strip('"').replace("kueue.DefaultPodSetName", "main"), "usage": {}, "flavors": {}, "count": 1}
                for n2, a2 in pc[1:]:
                    if n2 == "Assignment":
                        r, f, q = [x.strip() for x in split_top(a2)]
                        ps["flavors"][res_name(r)] = f.strip('"'); ps["usage"][res_name(r)] = q.strip('"')
                    elif n2 in ("Count", "AssignmentPodCount"):
                        ps["count"] = int(a2)
                    elif n2 == "TopologyAssignment":
                        ps["topologyAssignment"] = parse_topology_assignment(a2, sym)
                    elif n2 == "DelayedTopologyRequest":
                        ps["delayed"] = re.search(r"DelayedTopologyRequestState(\w+)", a2).group(1)
                    elif n2 == "Obj":
                        pass
                    else:
                        raise Skip("PodSetAssignment." + n2)
                podsets.append(ps)
        elif name != "Obj":
            raise Skip("Admission." + name)
    return cq, podsets


PS_OK = {"MakePodSet", "Request", "Obj", "Image", "RequiredTopologyRequest", "PreferredTopologyRequest", "UnconstrainedTopologyRequest",
         "SliceRequiredTopologyRequest", "SliceSizeTopologyRequest", "SliceRequiredTopologyConstraints", "Limit", "SetMinimumCount",
         "Toleration", "RequiredDuringSchedulingIgnoredDuringExecution"}


def parse_podsets(args, sym):
    out = []
    for t in split_top(args):
        if "MakePodSet" not in t:
            raise Skip("PodSets without MakePodSet")
        pc, _ = chain(t, t.index("MakePodSet"))
        n, c = split_top(pc[0][1])
        ps = {"name": n.strip().strip('"').replace("kueue.DefaultPodSetName", "main"), "count": int(c), "requests": {}}
        tr = {}
        for m, a in pc[1:]:
            if m not in PS_OK:
                raise Skip("PodSet." + m)
            if m == "Request":
                r, q = split_top(a)
                ps["requests"][res_name(r) if not r.strip().startswith('"') else r.strip().strip('"')] = q.strip().strip('"')
            elif m == "SetMinimumCount":
                ps["minCount"] = int(a)   # partial admission (PodSet.MinCount)
            elif m == "Toleration":
                ps.setdefault("tolerations", []).extend(parse_tolerations(a))
            elif m == "RequiredDuringSchedulingIgnoredDuringExecution":
                ps["requiredAffinity"] = parse_terms(a)
            elif m == "RequiredTopologyRequest":
                tr["required"] = label(a, sym)
            elif m == "PreferredTopologyRequest":
                tr["preferred"] = label(a, sym)
            elif m == "UnconstrainedTopologyRequest":
                tr["unconstrained"] = True
            elif m == "SliceRequiredTopologyRequest":
                tr["sliceRequiredTopology"] = label(a, sym)
            elif m == "SliceSizeTopologyRequest":
                tr["sliceSize"] = int(a)
            elif m == "SliceRequiredTopologyConstraints":
                tr["sliceConstraints"] = [dict(topology=label(k, sym), size=int(v))
                                          for k, v in re.findall(r"Topology:\s*([^,]+),\s*Size:\s*(\d+)", a)]
        if tr:
            ps["topologyRequest"] = tr
        out.append(ps)
    return out


WL_OK = {"MakeWorkload", "Queue", "Priority", "Creation", "Request", "PodSets", "ReserveQuota", "ReserveQuotaAt", "Admission", "Condition",
         "ResourceRequests", "SchedulingStatsEviction", "Obj", "UID", "JobUID", "Generation", "Clone", "AdmittedAt", "Admitted", "PastAdmittedTime",
         "ResourceVersion", "Label", "Labels", "Finalizers", "UnhealthyNodes", "AdmissionCheck"}


def parse_wl(text, start, sym):
    calls, end = chain(text, start)
    name, ns = [x.strip().strip('"') for x in split_top(calls[0][1])]
    w = {"name": name, "ns": ns, "priority": 0, "created": 0, "podsets": None}
    for m, a in calls[1:]:
        if m not in WL_OK:
            raise Skip("Workload." + m)
        if m == "Queue":
            w["queue"] = a.strip().strip('"')
        elif m == "Priority":
            w["priority"] = int(a)
        elif m == "Creation":
            w["created"] = parse_time(a)
        elif m == "PodSets":
            w["podsets"] = parse_podsets(a, sym)
        elif m == "Request":
            raise Skip("Workload.Request shorthand")
        elif m in ("ReserveQuota", "ReserveQuotaAt"):
            parts = split_top(a)
            w["cq"], w["admission"] = parse_admission(parts[0], sym)
            w["reservedAt"] = parse_time(parts[1]) if len(parts) > 1 else NOW
        elif m == "UnhealthyNodes":
            w["unhealthyNodes"] = re.findall(r'"([^"]+)"', a)
        elif m == "AdmissionCheck":
            w.setdefault("checks", []).append(re.search(r"State:\s*kueue\.CheckState(\w+)", a).group(1))
        elif m in ("AdmittedAt", "Admitted"):
            w["isAdmitted"] = split_top(a)[0].strip() == "true"
        elif m == "Condition":
            typ = re.search(r"Type:\s*kueue\.(\w+)", a)
            status = re.search(r"Status:\s*metav1\.Condition(\w+)", a)
            if typ and status and status.group(1) == "True" and typ.group(1) == "WorkloadEvicted":
                w["evicted"] = True
            if typ and status and status.group(1) == "True" and typ.group(1) == "WorkloadPreempted":
                w["preempted"] = True
            if typ and status and status.group(1) == "False" and typ.group(1) == "WorkloadQuotaReserved":
                rs = re.search(r'Reason:\s*(?:kueue\.)?"?([\w\.]+)"?', a)   # e.quotaReservedReason scheduler.go:433-513
                if rs:
                    w["pendingReason"] = rs.group(1).replace("WorkloadQuotaReservedReason", "")
    if w["podsets"] is None:
        raise Skip("workload without PodSets")
    return w, end


def workloads_in(text, sym):
    out, i = [], 0
    for m in re.finditer(r"utiltestingapi\.MakeWorkload\(", text):
        if m.start() < i:
            continue
        w, i = parse_wl(text, m.start() + len("utiltestingapi."), sym)
        out.append(w)
    return out


def parse_keymap(text):
    out = {}
    if not text:
        return out
    for m in re.finditer(r'"([^"]+)":\s*\{([^}]*)\}', text):
        out[m.group(1)] = re.findall(r'"([^"]+)"', m.group(2))
    return out


BAD = r"NodeSelector\(|PreemptionGate|WorkloadSlice|Annotation|PodSetGroup|" \
      r"resourceTransformations|patchStatusErr|PreferredDuringScheduling|PodSetUpdate|StopPolicy|" \
      r"NOTHING_ELSE_HERE"


def lqs_of(body):
    out = {}
    for m in re.finditer(r'MakeLocalQueue\("([^"]+)",\s*"([^"]+)"\)\.\s*ClusterQueue\("([^"]+)"\)', body):
        out[(m.group(2), m.group(1))] = m.group(3)
    return out


def extract(src, func, cases, skipped):
    body, line0 = func_body(src, func)
    sym = symbols(body)
    lqs = lqs_of(body)
    tm = re.search(r"(?m)^\tcases := map\[string\](?:struct \{|tasScheduleTestCase\{)", body)
    p = body.index("{", tm.start())
    if "struct {" in tm.group(0):
        p = body.index("{", match_brace(body, p) + 1)
    table = body[p + 1: match_brace(body, p)]
    for m in re.finditer(r'(?m)^\t\t"((?:[^"\\]|\\.)*)":\s*\{', table):
        j = match_brace(table, m.end() - 1)
        block = table[m.end():j]
        name = m.group(1)
        line = line0 + body[:p + 1 + m.start()].count("\n") + 1
        try:
            # (AdmissionCheck objects that exist in the cluster but that no ClusterQueue of the case lists do not reach the path: the
            # ClusterQueues are checked one by one below)
            scan = re.sub(r"(?m)^\t{3}admissionChecks:\s*\[\]kueue\.AdmissionCheck\{[^}]*\},?\n", "", block)
            if re.search(BAD, scan):
                raise Skip("outside the boundary (" + re.search(BAD, scan).group(0) + ")")
            # AdmissionChecks on a ClusterQueue (ProvisioningRequest: the topology request of a FIRST pass is delayed, tas_flavorassigner.go:105)
            # are the caller's: a case that has them is inside the boundary when every workload of it is on its second pass (decided below)
            with_checks = re.search(r"AdmissionCheck|DelayedTopologyRequest", scan) is not None
            gates = {}
            fg = field(block, "featureGates")
            if fg:
                for g, v in re.findall(r"features\.(\w+):\s*(true|false)", fg):
                    gates[g] = v == "true"
                allowed = {"TASMultiLayerTopology": True, "TASProfileMixed": True, "TASRecomputeAssignmentWithinSchedulingCycle": None, "VectorizedResourceRequests": None,
                           "TASFailedNodeReplacementFailFast": None,
                           "TASCachingRemainingResources": None, "TASCacheNodeMatchResults": None}
                for g, v in gates.items():
                    if g not in allowed or (allowed[g] is not None and allowed[g] != v):
                        raise Skip("gate " + g)
            def listf(fname):
                f = field(block, fname)
                return items_of(f, sym) if f else []
            for t in listf("nodes") + listf("clusterQueues") + listf("resourceFlavors"):
                if re.search(BAD, t):
                    raise Skip("outside the boundary (" + re.search(BAD, t).group(0) + ")")
            nodes = [parse_node(t, sym) for t in listf("nodes")]
            topologies = dict(parse_topology(t, sym) for t in listf("topologies"))
            flavors = [parse_flavor(t, sym) for t in listf("resourceFlavors")]
            cqs = []
            for t in listf("clusterQueues"):
                c = parse_cq(t)
                if "StrictFIFO" in t:
                    c["strategy"] = "StrictFIFO"
                cqs.append(c)
            cohorts = [parse_cohort(t) for t in listf("cohorts")]
            non_tas = parse_pods(listf("pods"))
            wf = field(block, "workloads") or ""
            wls = workloads_in(wf, sym)
            cq_names = {c["name"] for c in cqs}
            admitted, pending, second = [], [], []
            pod_requests = {}
            for w in wls:
                key = f"{w['ns']}/{w['name']}"
                if "cq" in w:
                    if w["cq"] not in cq_names:
                        raise Skip("admitted into an unknown ClusterQueue")
                    d = {"name": key, "cq": w["cq"], "priority": w["priority"], "created": w["created"], "reservedAt": w.get("reservedAt", NOW),
                         "evicted": bool(w.get("evicted")), "podsets": []}
                    spec = {ps["name"]: ps for ps in w["podsets"]}
                    for ps in w["admission"]:
                        e = {"count": ps["count"], "totalRequests": ps["usage"], "flavors": ps["flavors"]}
                        # totalRequestsFromAdmission workload.go:758-766: a spec count below the admission's (reclaimable pods) scales the
                        # quota usage down (Requests.Divide / Mul: integers in the resource's unit); the TopologyAssignment keeps its counts
                        if ps["name"] in spec and spec[ps["name"]]["count"] < ps["count"]:
                            from kueue_amd.api import amount_from_quantity
                            c0, c1 = ps["count"], spec[ps["name"]]["count"]
                            e["totalRequests"] = {r: (lambda v: f"{v}m" if r == "cpu" else str(v))(amount_from_quantity(r, q) // c0 * c1) for r, q in ps["usage"].items()}
                            e["count"] = c1
                        if "topologyAssignment" in ps:
                            e["topologyAssignment"] = ps["topologyAssignment"]
                            e["podRequests"] = spec[ps["name"]]["requests"] if ps["name"] in spec else {}
                        d["podsets"].append(e)
                    admitted.append(d)
                    # workload.NeedsSecondPass workload.go:974 after a node failure: quota reserved, admitted, and a TopologyAssignment
                    # names one of Status.UnhealthyNodes -> the workload is ALSO a head of the cycle (manager.go:923)
                    un = w.get("unhealthyNodes") or []
                    names = [dm[0][-1] for ps in w["admission"] if "topologyAssignment" in ps for dm in ps["topologyAssignment"]["domains"]]
                    if un and w.get("isAdmitted") and any(x in un for x in names):
                        by = {ps["name"]: ps for ps in w["admission"]}
                        if [ps["name"] for ps in w["podsets"]] != [ps["name"] for ps in w["admission"]]:
                            raise Skip("admission podsets not aligned with the spec")
                        # workload.Info of an admitted workload: count and total requests are the admission's (workload.go:866-900); the
                        # placement reads the pod spec (tas_flavorassigner.go:116)
                        sp_podsets = [{"name": ps["name"], "count": by[ps["name"]]["count"], "totalRequests": by[ps["name"]]["usage"],
                                       "podRequests": ps["requests"], **({"topologyRequest": ps["topologyRequest"]} if "topologyRequest" in ps else {})}
                                      for ps in w["podsets"]]
                        second.append({"name": key, "cq": w["cq"], "priority": w["priority"], "created": w["created"], "podsets": sp_podsets,
                                       "hasQuotaReservation": True, "isAdmitted": True, "unhealthyNodes": un,
                                       "admission": [{"flavors": by[ps["name"]]["flavors"], "count": by[ps["name"]]["count"],
                                                      **({"topologyAssignment": by[ps["name"]]["topologyAssignment"]} if "topologyAssignment" in by[ps["name"]] else {})}
                                                     for ps in w["podsets"]]})
                    elif un:
                        raise Skip("unhealthy nodes without a second pass")
                    # ... or for a delayed topology request (needsSecondPassForDelayedAssignment workload.go:981): every admission check
                    # Ready, a PodSetAssignment with DelayedTopologyRequest Pending and no TopologyAssignment, not admitted yet
                    elif w.get("checks") and all(c == "Ready" for c in w["checks"]) and not w.get("isAdmitted") and \
                            any(ps.get("delayed") == "Pending" and "topologyAssignment" not in ps for ps in w["admission"]):
                        by = {ps["name"]: ps for ps in w["admission"]}
                        if [ps["name"] for ps in w["podsets"]] != [ps["name"] for ps in w["admission"]]:
                            raise Skip("admission podsets not aligned with the spec")
                        sp_podsets = [{"name": ps["name"], "count": by[ps["name"]]["count"], "totalRequests": by[ps["name"]]["usage"],
                                       "podRequests": ps["requests"], **({"topologyRequest": ps["topologyRequest"]} if "topologyRequest" in ps else {})}
                                      for ps in w["podsets"]]
                        second.append({"name": key, "cq": w["cq"], "priority": w["priority"], "created": w["created"], "podsets": sp_podsets,
                                       "hasQuotaReservation": True, "isAdmitted": False,
                                       "admission": [{"flavors": by[ps["name"]]["flavors"], "count": by[ps["name"]]["count"]} for ps in w["podsets"]]})
                    elif with_checks:
                        raise Skip("outside the boundary (AdmissionCheck)")
                else:
                    if with_checks:
                        raise Skip("outside the boundary (AdmissionCheck)")   # a first pass in a case with admission checks: the caller's (delayed topology request)
                    cq = lqs.get((w["ns"], w.get("queue", "")))
                    if cq is None or cq not in cq_names:
                        raise Skip("pending workload in a missing LocalQueue/ClusterQueue")
                    pending.append({"name": key, "cq": cq, "priority": w["priority"], "created": w["created"], "podsets": w["podsets"]})
            heads, rest = [], []
            for cq in sorted({p_["cq"] for p_ in pending}):
                q = sorted([p_ for p_ in pending if p_["cq"] == cq], key=lambda p_: (-p_["priority"], p_["created"]))
                if len(q) > 1 and (-q[0]["priority"], q[0]["created"]) == (-q[1]["priority"], q[1]["created"]):
                    raise Skip("head order decided by UID / name tie-break")
                heads.append(q[0]); rest += q[1:]
            tas_flavor_names = {f["name"] for f in flavors if f.get("topologyName")}
            for h_ in second:   # (kq_cycle_tas.h: a workload whose podsets hold TAS flavors of their own is KQ_EUNSUPPORTED — TASHandleOverlappingFlavors)
                if len({fl for ps in h_["admission"] for fl in ps["flavors"].values() if fl in tas_flavor_names}) > 1:
                    raise Skip("a workload on two TAS flavors")
            heads = second + heads   # "second-pass heads first" (manager.go:923)
            want_adm = {}
            wa = field(block, "wantNewAssignments")
            if wa and "{" in wa:
                p0 = wa.index("{")
                for ent in split_top(wa[p0 + 1: match_brace(wa, p0)]):
                    km = re.match(r'\s*"([^"]+)":\s*', ent)
                    if not km:
                        continue
                    cq, pss = parse_admission(ent[km.end():], sym)
                    want_adm[km.group(1)] = {"cq": cq, "podsets": pss}
            expect = {}
            for h in heads:
                if h["name"] in want_adm:
                    expect[h["name"]] = {"admitted": True, "podsets": [
                        {"flavors": ps["flavors"], "count": ps["count"], **({"topologyAssignment": ps["topologyAssignment"]} if "topologyAssignment" in ps else {})}
                        for ps in want_adm[h["name"]]["podsets"]]}
                else:
                    expect[h["name"]] = {"admitted": False}
            for k in want_adm:
                if k not in expect:
                    raise Skip("admission of a workload that is not a head")
                if len({fl for ps in want_adm[k]["podsets"] for fl in ps["flavors"].values() if fl in tas_flavor_names}) > 1:
                    raise Skip("a workload on two TAS flavors")   # (the expected admission spreads one workload over two TAS flavors: KQ_EUNSUPPORTED)
            ev = field(block, "wantEvents") or ""
            for ns_, nm_, reason in re.findall(r'MakeEventRecord\("([^"]+)",\s*"([^"]+)",\s*"([^"]+)"', ev):
                if f"{ns_}/{nm_}" in expect:
                    expect[f"{ns_}/{nm_}"].setdefault("events", []).append(reason)
            left = parse_keymap(field(block, "wantLeft"))
            inadm = parse_keymap(field(block, "wantInadmissibleLeft"))
            for keys in left.values():
                for k in keys:
                    if k in expect:
                        expect[k]["left"] = "active"
            for keys in inadm.values():
                for k in keys:
                    if k in expect:
                        expect[k]["left"] = "inadmissible"
            want_wls = workloads_in(field(block, "wantWorkloads") or "", sym)
            preempted = sorted(f"{w['ns']}/{w['name']}" for w in want_wls if w.get("preempted"))
            for w in want_wls:
                k = f"{w['ns']}/{w['name']}"
                if "pendingReason" in w and k in expect and not expect[k]["admitted"]:
                    expect[k]["reason"] = w["pendingReason"]
            case = {"name": name, "ref": f"pkg/scheduler/scheduler_tas_test.go:{line}", "func": func, "now": NOW, "nodes": nodes,
                    "topologies": topologies, "resourceFlavors": flavors, "clusterQueues": cqs, "cohorts": cohorts, "admitted": admitted,
                    "pending": heads, "notHeads": [r["name"] for r in rest], "expect": expect}
            if non_tas:
                case["nonTASUsage"] = non_tas
            if want_wls:
                case["wantPreempted"] = preempted
            if gates:
                case["gatesGo"] = gates
            cases.append(case)
        except Skip as e:
            skipped[str(e)] += 1
        except (ValueError, KeyError, IndexError, AttributeError) as e:
            skipped["parse:" + type(e).__name__] += 1


def main():
    src = strip_comments(open(SRC).read())
    cases, skipped = [], collections.Counter()
    for func in ("TestScheduleForTAS", "TestScheduleForTASPreemption", "TestScheduleForTASCohorts"):
        extract(src, func, cases, skipped)
    out = os.path.join(HERE, "schedule_tas.yaml")
    with open(out, "w") as f:
        f.write("# GENERATED by tests/golden/extract_schedule_tas.py from /root/reference/pkg/scheduler/scheduler_tas_test.go\n")
        f.write("# %d cases kept; skipped: %s\n" % (len(cases), dict(skipped)))
        yaml.safe_dump({"cases": cases}, f, sort_keys=False, width=160)
    print(len(cases), "cases;", dict(skipped))


if __name__ == "__main__":
    main()

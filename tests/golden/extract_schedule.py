#!/usr/bin/env python
"""Transcribes whole-cycle tables TestSchedule (scheduler_test.go:366), TestScheduleForFairSharing (scheduler_fs_test.go:38),
TestScheduleRecomputePreemptionTargets (scheduler_recompute_preemption_test.go:59) and TestScheduleForPreserveFlavorScanProgress
(scheduler_preserve_flavor_scan_progress_test.go:82) into YAML.

  python tests/golden/extract_schedule.py      # needs /root/reference (this container only)

A Go case = default + additional ClusterQueues/LocalQueues/Cohorts, a list of Workloads (some already holding a
quota reservation, the others pending in a LocalQueue) and, after exactly ONE Scheduler.schedule() call:
wantAssignments (every admission in the cache), wantLeft / wantInadmissibleLeft (queue dumps) and wantWorkloads
(conditions: WorkloadPreempted marks the victims).  The fixture keeps: the snapshot inputs, the heads the queue
manager would hand out (one per ClusterQueue: priority desc, creation asc — cluster_queue.go:844), and the
expected per-head outcome {admitted with flavors/counts | left active | left inadmissible} + preempted set.
Cases needing machinery outside the engine boundary (admission checks, taints/tolerations, limit ranges, slices,
TAS, gates, injected API errors, namespace selectors, missing ClusterQueues/flavors) are skipped and listed.
"""
import os
import re
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_assign_flavors import match_brace, parse_cq, res_name  # noqa: E402
from extract_preemption import NOW, chain, field, list_items, parse_cohort, parse_time, split_top  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pkg/scheduler/"

WL_OK = {"MakeWorkload", "Queue", "Priority", "Creation", "Request", "PodSets", "ReserveQuota", "ReserveQuotaAt", "Admission", "Condition",
         "ResourceRequests", "SchedulingStatsEviction", "Obj", "UID", "Generation", "Clone", "AdmittedAt", "Admitted", "PastAdmittedTime",
         "ResourceVersion", "SimpleReserveQuota", "Label", "Labels", "JobUID"}
PS_OK = {"MakePodSet", "Request", "SetMinimumCount", "Obj", "Image"}


class Skip(Exception):
    pass


def parse_admission(args):
    calls, _ = chain(args, args.index("MakeAdmission"))
    head = split_top(calls[0][1])
    cq = head[0].strip().strip('"')
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
                    elif n2 in ("Obj", "Flavor"):
                        if n2 == "Flavor":
                            r, f = [x.strip() for x in split_top(a2)]
                            ps["flavors"][res_name(r)] = f.strip('"')
                    else:
                        raise Skip(f"PodSetAssignment.{n2}")
                podsets.append(ps)
        elif name == "Assignment":
            r, f, q = [x.strip() for x in split_top(a)]
            if not podsets:
                podsets.append({"name": "main", "usage": {}, "flavors": {}, "count": 1})
            podsets[0]["flavors"][res_name(r)] = f.strip('"'); podsets[0]["usage"][res_name(r)] = q.strip('"')
        elif name == "AssignmentPodCount":
            podsets[0]["count"] = int(a)
        elif name != "Obj":
            raise Skip(f"Admission.{name}")
    return cq, podsets


def parse_podsets(args):
    out = []
    for t in split_top(args):
        if "MakePodSet" not in t:
            raise Skip("PodSets without MakePodSet")
        pc, _ = chain(t, t.index("MakePodSet"))
        n, c = split_top(pc[0][1])
        ps = {"name": n.strip().strip('"').replace("kueue.DefaultPodSetName", "main"), "count": int(c), "requests": {}}
        for m, a in pc[1:]:
            if m not in PS_OK:
                raise Skip(f"PodSet.{m}")
            if m == "Request":
                r, q = split_top(a)
                ps["requests"][res_name(r)] = q.strip('"')
            elif m == "SetMinimumCount":
                ps["minCount"] = int(a)
        out.append(ps)
    return out


def parse_wl(text, start):
    calls, end = chain(text, start)
    name, ns = [x.strip().strip('"') for x in split_top(calls[0][1])]
    w = {"name": name, "ns": ns, "priority": 0, "created": 0, "podsets": None, "requests": {}}
    for m, a in calls[1:]:
        if m not in WL_OK:
            raise Skip(f"Workload.{m}")
        if m == "Queue":
            w["queue"] = a.strip().strip('"')
        elif m == "Priority":
            w["priority"] = int(a)
        elif m == "Creation":
            w["created"] = parse_time(a)
        elif m == "Request":
            r, q = split_top(a)
            w["requests"][res_name(r)] = q.strip('"')
        elif m == "PodSets":
            w["podsets"] = parse_podsets(a)
        elif m in ("ReserveQuota", "ReserveQuotaAt"):
            parts = split_top(a)
            w["cq"], w["admission"] = parse_admission(parts[0])
            w["reservedAt"] = parse_time(parts[1]) if len(parts) > 1 else NOW
        elif m == "Admission":
            w["wantCq"], w["wantAdmission"] = parse_admission(a)
        elif m == "SimpleReserveQuota":
            cq, fl, t = split_top(a)
            w["cq"] = cq.strip('"'); w["simpleFlavor"] = fl.strip('"'); w["reservedAt"] = parse_time(t)
        elif m == "Condition":
            typ = re.search(r"Type:\s*kueue\.(\w+)", a)
            status = re.search(r"Status:\s*metav1\.Condition(\w+)", a)
            reason = re.search(r'Reason:\s*(?:kueue\.)?"?([\w\.]+)"?', a)
            if typ and status and typ.group(1) == "WorkloadQuotaReserved" and status.group(1) == "False":
                # the status message the scheduler patches for a head that stays pending (requeueAndUpdate scheduler.go:1189-1190)
                mm = re.search(r'Message:\s*((?:"(?:[^"\\]|\\.)*"\s*\+?\s*)+)', a)
                if mm:
                    w["pendingMessage"] = "".join(bytes(x, "utf-8").decode("unicode_escape") for x in re.findall(r'"((?:[^"\\]|\\.)*)"', mm.group(1)))
                if reason:   # e.quotaReservedReason (scheduler.go:433-513) -> the condition's Reason with UnadmittedWorkloadsObservability on (the default)
                    w["pendingReason"] = reason.group(1).replace("WorkloadQuotaReservedReason", "")
            if typ and status and status.group(1) == "True":
                if typ.group(1) == "WorkloadPreempted":
                    w["preemptedReason"] = reason.group(1).replace("Reason", "") if reason else ""
                if typ.group(1) == "WorkloadEvicted":
                    w["evicted"] = True
    if w["podsets"] is None:
        w["podsets"] = [{"name": "main", "count": 1, "requests": w["requests"]}]
    return w, end


def workloads_in(text):
    out, i = [], 0
    for m in re.finditer(r"utiltestingapi\.MakeWorkload\(", text):
        if m.start() < i:
            continue
        w, i = parse_wl(text, m.start() + len("utiltestingapi."))
        out.append(w)
    return out


def inline_func_vars(text):
    """`additionalClusterQueues: func() []kueue.ClusterQueue { rg := ...; preemption := ...; cq1 := *Make...(rg)...; return ... }()`
    -> the same text with the local variables substituted into the MakeClusterQueue chains that use them."""
    if not re.match(r"\s*func\(\)", text):
        return text
    lines = text.split("\n")
    vars_, i = {}, 0
    while i < len(lines):
        m = re.match(r"^\s*(\w+) := (.*)$", lines[i])
        if not m:
            i += 1
            continue
        expr = m.group(2)
        while expr.count("(") != expr.count(")") or expr.count("{") != expr.count("}"):
            i += 1
            expr += "\n" + lines[i]
        vars_[m.group(1)] = expr.lstrip("*")
        i += 1
    cq_exprs = {}
    for name, expr in vars_.items():
        if "MakeClusterQueue" not in expr:
            continue
        for other, oexpr in vars_.items():
            if "MakeClusterQueue" not in oexpr:
                expr = re.sub(r"\b(ResourceGroup|Preemption)\(" + re.escape(other) + r"\)", lambda mm: mm.group(1) + "(" + oexpr + ")", expr)
        cq_exprs[name] = expr
    m = re.search(r"return \[\]kueue\.ClusterQueue\{([^}]*)\}", text)
    order = [n.strip() for n in m.group(1).split(",") if n.strip()] if m else list(cq_exprs)
    if not cq_exprs or any(n not in cq_exprs for n in order):
        return text
    return "[]kueue.ClusterQueue{\n" + ",\n".join("*" + cq_exprs[n] for n in order) + ",\n}"


def parse_lqs(text):
    out = {}
    for m in re.finditer(r'MakeLocalQueue\("([^"]+)",\s*"([^"]+)"\)\.\s*ClusterQueue\("([^"]+)"\)', text):
        out[(m.group(2), m.group(1))] = m.group(3)
    return out


def parse_keymap(text):
    """map[ClusterQueueReference][]workload.Reference literal -> {cq: [keys]}"""
    out = {}
    if not text:
        return out
    for m in re.finditer(r'"([^"]+)":\s*\{([^}]*)\}', text):
        out[m.group(1)] = re.findall(r'"([^"]+)"', m.group(2))
    return out


def extract(fname, func, cases, skipped):
    src = open(REF + fname).read()
    src = "\n".join("" if l.strip().startswith("//") else l for l in src.split("\n"))
    src = re.sub(r"/\*.*?\*/", "", src)          # inline /* ... */ comments
    src = re.sub(r"(?m)\s//[^\"\n]*$", "", src)  # trailing // comments (no string literal after them)
    start = src.index("func %s(" % func)
    m = re.search(r"(?m)^\tclusterQueues := \[\]kueue\.ClusterQueue\{", src[start:])
    p = start + m.end() - 1
    default_cqs = [parse_cq(t) for t in list_items(src[p + 1: match_brace(src, p)], "MakeClusterQueue")]
    for cq, t in zip(default_cqs, list_items(src[p + 1: match_brace(src, p)], "MakeClusterQueue")):
        if "StrictFIFO" in t:
            cq["strategy"] = "StrictFIFO"
    m = re.search(r"(?m)^\tqueues := \[\]kueue\.LocalQueue\{", src[start:])
    p = start + m.end() - 1
    default_lqs = parse_lqs(src[p + 1: match_brace(src, p)])
    table_start = src.index("cases := map[string]scheduleTestCase{", start)
    body_start = src.index("{", table_start + len("cases := map[string]scheduleTestCase") - 1)
    body_end = match_brace(src, body_start)
    table = src[body_start + 1: body_end]
    for m in re.finditer(r'(?m)^\t\t"((?:[^"\\]|\\.)*)":\s*\{', table):
        j = match_brace(table, m.end() - 1)
        block = table[m.end():j]
        name = m.group(1)
        line = src[: body_start + 1 + m.start()].count("\n") + 1
        try:
            if re.search(r"admissionError|objects:|wantWorkloadUseMergePatch|AdmissionCheck|Toleration|NodeSelector|TopologyRequest|PreemptionGate|WorkloadSlice|Annotation", block):
                raise Skip("admission checks / tolerations / slices / gates / injected errors")
            gates = {}
            fg = field(block, "featureGates")
            if fg:
                for g, v in re.findall(r"features\.(\w+):\s*(true|false)", fg):
                    gates[g] = v == "true"
                known = {"FlavorFungibility", "PartialAdmission", "PrioritySortingWithinCohort", "FairSharingPreemptWithinNominal",
                         "FairSharingPrioritizeNonBorrowing", "RecomputeAssignmentUponPreemptionTargetsOverlap", "PrioritizePreemptorWorkloads",
                         "FlavorFungibilityPreserveScanProgress"}
                # the TAS gates do not touch a case without any TopologyRequest / TAS flavor (those blocks were skipped above): noted, not applied
                ignored = {g: v for g, v in gates.items() if g in ("TopologyAwareScheduling", "TASRecomputeAssignmentWithinSchedulingCycle")}
                gates = {g: v for g, v in gates.items() if g not in ignored}
                if any(g not in known for g in gates):
                    raise Skip(f"feature gates {sorted(gates)}")
            cqs = [dict(c) for c in default_cqs]
            acq = field(block, "additionalClusterQueues")
            if acq:
                acq = inline_func_vars(acq)
                p = acq.index("{")
                items = list_items(acq[p + 1: match_brace(acq, p)], "MakeClusterQueue")
                for t in items:
                    c = parse_cq(t)
                    if "StrictFIFO" in t:
                        c["strategy"] = "StrictFIFO"
                    if "AdmissionChecks" in t or "StopPolicy" in t:
                        raise Skip("ClusterQueue admission checks / stop policy")
                    cqs.append(c)
            lqs = dict(default_lqs)
            alq = field(block, "additionalLocalQueues")
            if alq:
                lqs.update(parse_lqs(alq))
            cohorts = []
            cf = field(block, "cohorts")
            if cf and re.match(r"\s*(\w+)\(\)\s*$", cf):   # a helper of the test file returning []kueue.Cohort (defaultCohorts())
                fn = re.match(r"\s*(\w+)\(\)", cf).group(1)
                fm = re.search(r"(?m)^func %s\(\) \[\]kueue\.Cohort \{" % fn, src)
                fbody = src[fm.end(): match_brace(src, fm.end() - 1)]
                rm = re.search(r"return \[\]kueue\.Cohort\{", fbody)
                cf = fbody[rm.start() + len("return "):]
            if cf:
                p = cf.index("{")
                cohorts = [parse_cohort(t) for t in list_items(cf[p + 1: match_brace(cf, p)], "MakeCohort")]
            wls = workloads_in(field(block, "workloads") or "")
            want_wls = workloads_in(field(block, "wantWorkloads") or "")
            cq_names = {c["name"] for c in cqs}
            flavors_known = {"default", "on-demand", "spot", "model-a", "spot-tainted", "spot-tainted-2"}
            tainted = {"spot-tainted", "spot-tainted-2"}   # scheduler_test.go:376-387 key=val / key=val2 NoSchedule; rows whose pods tolerate them are skipped above
            for c in cqs:
                for rg in c["resourceGroups"]:
                    for f in rg:
                        if f["flavor"] not in flavors_known:
                            c["_bad"] = True
            admitted, pending = [], []
            for w in wls:
                if "cq" in w:
                    if w["cq"] not in cq_names:
                        raise Skip("admitted into an unknown ClusterQueue")
                    d = {"name": f"{w['ns']}/{w['name']}", "cq": w["cq"], "priority": w["priority"], "created": w["created"],
                         "reservedAt": w.get("reservedAt", NOW), "evicted": bool(w.get("evicted"))}
                    if "admission" in w:
                        d["podsets"] = [{"count": ps["count"], "totalRequests": ps["usage"], "flavors": ps["flavors"]} for ps in w["admission"]]
                    else:
                        d["podsets"] = [{"count": ps["count"], "requests": ps["requests"], "flavors": {r: w["simpleFlavor"] for r in ps["requests"]}} for ps in w["podsets"]]
                    admitted.append(d)
                else:
                    cq = lqs.get((w["ns"], w.get("queue", "")))
                    if cq is None or cq not in cq_names:
                        raise Skip("pending workload in a missing LocalQueue/ClusterQueue")
                    if any(c.get("_bad") for c in cqs if c["name"] == cq):
                        raise Skip("ClusterQueue with a nonexistent ResourceFlavor")
                    # namespace selector of the default CQs: ns label dep must match
                    dep = {"sales": "sales", "eng-alpha": "eng", "eng-beta": "eng", "eng-gamma": "eng", "lend": "lend", "default": None}.get(w["ns"])
                    need = {"sales": "sales", "eng-alpha": "eng", "eng-beta": "eng", "lend-a": "lend", "lend-b": "lend"}.get(cq)
                    if need is not None and dep != need:
                        raise Skip("namespace selector mismatch (host-side gatekeeping)")
                    excl = sorted({f["flavor"] for c in cqs if c["name"] == cq for rg in c["resourceGroups"] for f in rg} & tainted)
                    if excl:   # checkFlavorForPodSets "untolerated taint" (flavorassigner.go:1243): the host-side eligibility mask
                        for ps in w["podsets"]:
                            ps["excludedFlavors"] = excl
                    pending.append({"name": f"{w['ns']}/{w['name']}", "cq": cq, "priority": w["priority"], "created": w["created"], "podsets": w["podsets"]})
            cqs = [c for c in cqs if not c.get("_bad")]
            # heads: one per ClusterQueue, priority desc then creation asc (cluster_queue.go:844)
            heads, rest = [], []
            for cq in sorted({p["cq"] for p in pending}):
                q = sorted([p for p in pending if p["cq"] == cq], key=lambda p: (-p["priority"], p["created"], p["name"]))
                heads.append(q[0]); rest += q[1:]
            # expectations
            wa = field(block, "wantAssignments")
            want_adm = {}
            if wa:
                p0 = wa.index("{")
                for ent in split_top(wa[p0 + 1: match_brace(wa, p0)]):
                    km = re.match(r'\s*"([^"]+)":\s*', ent)
                    val = ent[km.end():]
                    if "MakeAdmission" in val and not val.lstrip().startswith("{"):
                        cq, pss = parse_admission(val)
                    else:  # kueue.Admission{ClusterQueue: ..., PodSetAssignments: []kueue.PodSetAssignment{...}}
                        cq = re.search(r'ClusterQueue:\s*"([^"]+)"', val).group(1)
                        pm = re.search(r"PodSetAssignments:\s*\[\]kueue\.PodSetAssignment\{", val)
                        body = val[pm.end(): match_brace(val, pm.end() - 1)]
                        fake = 'MakeAdmission("%s").PodSets(%s)' % (cq, ", ".join(split_top(body)))
                        cq, pss = parse_admission(fake)
                    want_adm[km.group(1)] = {"cq": cq, "podsets": pss}
            adm_keys = {a["name"] for a in admitted}
            expect = {}
            for h in heads:
                if h["name"] in want_adm and h["name"] not in adm_keys:
                    expect[h["name"]] = {"admitted": True, "podsets": [{"flavors": ps["flavors"], "count": ps["count"]} for ps in want_adm[h["name"]]["podsets"]]}
                else:
                    expect[h["name"]] = {"admitted": False}
            left = parse_keymap(field(block, "wantLeft"))
            inadm = parse_keymap(field(block, "wantInadmissibleLeft"))
            for cq, keys in left.items():
                for k in keys:
                    if k in expect:
                        expect[k]["left"] = "active"
            for cq, keys in inadm.items():
                for k in keys:
                    if k in expect:
                        expect[k]["left"] = "inadmissible"
            for w in want_wls:
                k = f"{w['ns']}/{w['name']}"
                if "pendingMessage" in w and k in expect and not expect[k]["admitted"]:
                    expect[k]["message"] = w["pendingMessage"]
                if "pendingReason" in w and k in expect and not expect[k]["admitted"]:
                    expect[k]["reason"] = w["pendingReason"]
            preempted = sorted(f"{w['ns']}/{w['name']}:{w['preemptedReason']}" for w in want_wls if "preemptedReason" in w and f"{w['ns']}/{w['name']}" in adm_keys)
            for c in cqs:
                c.pop("_bad", None)
            case = {"name": name, "ref": f"pkg/scheduler/{fname}:{line}", "now": NOW, "clusterQueues": cqs, "cohorts": cohorts,
                    "admitted": admitted, "pending": heads, "notHeads": [r["name"] for r in rest], "expect": expect, "wantPreempted": preempted}
            if gates:
                case["gates"] = gates
            if fg and ignored:
                case["ignoredGates"] = ignored
            efs = field(block, "enableFairSharing")
            if efs and efs.strip() == "true":
                case["fairSharing"] = True
            cases.append(case)
        except Skip as ex:
            skipped.append((name, str(ex)))
        except Exception as ex:  # noqa: BLE001
            skipped.append((name, f"parse error: {ex!r}"))


def main():
    for fname, func, out in (("scheduler_test.go", "TestSchedule", "schedule.yaml"), ("scheduler_fs_test.go", "TestScheduleForFairSharing", "schedule_fair.yaml"),
                             ("scheduler_recompute_preemption_test.go", "TestScheduleRecomputePreemptionTargets", "schedule_recompute.yaml")):
        # (TestScheduleForPreserveFlavorScanProgress, scheduler_preserve_flavor_scan_progress_test.go:82, is a multi-cycle TAS scenario, not a
        #  scheduleTestCase table: transcribed by hand into schedule_scan_progress_manual.yaml)
        cases, skipped = [], []
        extract(fname, func, cases, skipped)
        with open(os.path.join(HERE, out), "w") as f:
            f.write(f"# GENERATED by tests/golden/extract_schedule.py from /root/reference/pkg/scheduler/{fname} ({func})\n")
            yaml.safe_dump({"cases": cases, "skipped": [{"name": n, "why": w} for n, w in skipped]}, f, sort_keys=False, width=200)
        print(f"{func}: {len(cases)} cases transcribed, {len(skipped)} skipped -> {out}")
        for n, w in skipped:
            print("   skipped:", n[:70], "--", w)


if __name__ == "__main__":
    main()

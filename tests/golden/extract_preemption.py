#!/usr/bin/env python
"""Transcribes TestPreemption / TestHierarchicalPreemptions (pkg/scheduler/preemption/*_test.go) into YAML.

  python tests/golden/extract_preemption.py     # needs /root/reference (this container only)

Each Go case gives ClusterQueues/Cohorts, admitted Workloads (with their admission = flavor per resource and
usage), an incoming Workload, its Assignment (flavor + mode per resource) and the Workload objects expected
after IssuePreemptions. Expected targets = the wanted Workloads that carry a WorkloadPreempted condition;
its Reason is the preemption reason (preemption_test.go:4093-4170 harness).
"""
import os
import re
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_assign_flavors import match_brace, parse_cq, res_name  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pkg/scheduler/preemption/"
NOW = 1_000_000_000_000  # ns


def parse_time(expr):
    expr = expr.strip()
    if expr == "now":
        return NOW
    m = re.match(r"now\.Add\((.*)\)$", expr)
    if m:
        e = m.group(1).replace("time.Second", "1000000000").replace("time.Minute", "60000000000").replace("time.Millisecond", "1000000")
        return NOW + int(eval(e, {"__builtins__": {}}))
    raise ValueError(expr)


def chain(text, start):
    """parse `Make...(args).M1(args).M2(args)...` starting at text[start]; returns ([(name,args)], end)"""
    calls = []
    m = re.compile(r"\s*\.?\s*([A-Za-z_][\w\.]*)\(").match(text, start)
    i = start
    while m:
        p = m.end() - 1
        e = match_brace(text, p, "(", ")")
        calls.append((m.group(1).split(".")[-1], text[p + 1:e]))
        i = e + 1
        m = re.compile(r"\s*\.\s*([A-Za-z_]\w*)\(").match(text, i)
    return calls, i


def split_top(text):
    """split a Go composite-literal body on top-level commas"""
    out, depth, cur, i = [], 0, [], 0
    while i < len(text):
        c = text[i]
        if c == '"':
            j = i + 1
            while text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            cur.append(text[i:j + 1]); i = j + 1; continue
        if c in "{([":
            depth += 1
        elif c in "})]":
            depth -= 1
        if c == "," and depth == 0:
            out.append("".join(cur)); cur = []
        else:
            cur.append(c)
        i += 1
    if "".join(cur).strip():
        out.append("".join(cur))
    return [x.strip() for x in out if x.strip()]


def parse_admission(args):
    calls, _ = chain(args, args.index("MakeAdmission"))
    cq = calls[0][1].strip().strip('"')
    podsets = []
    for name, a in calls[1:]:
        if name == "PodSets":
            for psa in split_top(a):
                pc, _ = chain(psa, psa.index("MakePodSetAssignment"))
                ps = {"usage": {}, "flavors": {}}
                for n2, a2 in pc[1:]:
                    if n2 == "Assignment":
                        r, f, q = [x.strip() for x in split_top(a2)]
                        ps["flavors"][res_name(r)] = f.strip('"'); ps["usage"][res_name(r)] = q.strip('"')
                podsets.append(ps)
    return cq, podsets


def parse_workload(text, start):
    calls, end = chain(text, start)
    w = {"name": None, "priority": 0, "requests": {}, "created": NOW}
    first = calls[0]
    if first[0] == "MakeWorkload":
        w["name"] = split_top(first[1])[0].strip('"')
    else:  # baseIncomingWl.Clone()
        w["name"] = "in"; w["uid"] = "wl-in"
    for name, a in calls[1:]:
        if name == "Priority":
            w["priority"] = int(a)
        elif name == "Name":
            w["name"] = a.strip().strip('"')
        elif name == "UID":
            w["uid"] = a.strip().strip('"')
        elif name == "Request":
            r, q = split_top(a)
            w["requests"][res_name(r)] = q.strip('"')
        elif name == "Creation":
            w["created"] = parse_time(a)
        elif name == "ReserveQuotaAt":
            parts = split_top(a)
            w["cq"], w["admission"] = parse_admission(parts[0])
            w["reservedAt"] = parse_time(parts[1])
        elif name == "ReserveQuota":
            w["cq"], w["admission"] = parse_admission(a)
            w["reservedAt"] = NOW
        elif name == "SimpleReserveQuota":
            cq, fl, t = split_top(a)
            w["cq"] = cq.strip('"'); w["simpleFlavor"] = fl.strip('"'); w["reservedAt"] = parse_time(t)
        elif name == "Condition":
            typ = re.search(r"Type:\s*kueue\.(\w+)", a)
            reason = re.search(r'Reason:\s*"?([\w\.]+)"?', a)
            status = re.search(r"Status:\s*metav1\.Condition(\w+)", a)
            if typ and status and status.group(1) == "True":
                if typ.group(1) == "WorkloadEvicted":
                    w["evicted"] = True
                if typ.group(1) == "WorkloadPreempted":
                    w["preemptedReason"] = reason.group(1).split(".")[-1].replace("Reason", "") if reason else ""
        elif name in ("PodSets",):
            w["unsupported"] = "explicit PodSets"
    return w, end


def workloads_in(text, unit=None):
    out, i = [], 0
    for m in re.finditer(r"utiltestingapi\.MakeWorkload\(|baseIncomingWl\.\s*Clone\(|unitWl\.\s*Clone\(", text):
        if m.start() < i:
            continue
        is_unit = text[m.start():].startswith("unitWl")
        w, i = parse_workload(text, m.start() + (len("utiltestingapi.") if text[m.start():].startswith("utiltestingapi") else 0))
        if is_unit:
            w.pop("uid", None)
            w["requests"] = dict(unit or {})
        out.append(w)
    return out


def parse_cohort(text):
    calls, _ = chain(text, text.index("MakeCohort"))
    c = {"name": calls[0][1].strip().strip('"')}
    for name, a in calls[1:]:
        if name == "Parent":
            c["parent"] = a.strip().strip('"')
        elif name == "FairWeight":
            mm = re.search(r'MustParse\("([^"]+)"\)', a)
            c["fairWeight"] = float(mm.group(1)) if mm else 1.0
    cq = parse_cq(text.replace("MakeCohort", "MakeClusterQueue"))
    if cq["resourceGroups"]:
        c["resourceGroups"] = cq["resourceGroups"]
    return c


def list_items(text, maker):
    """all `maker(` chains at top level of a slice literal"""
    return [t for t in split_top(text) if maker in t]


def field(block, name, tabs=3):
    m = re.search(r"(?m)^\t{%d}%s:\s*" % (tabs, re.escape(name)), block)
    if not m:
        return None
    parts = split_top(block[m.end():])
    return parts[0] if parts else None


def to_admitted(w):
    d = {"name": w["name"], "cq": w["cq"], "priority": w["priority"], "created": w["created"], "reservedAt": w.get("reservedAt"),
         "evicted": bool(w.get("evicted"))}
    if "uid" in w:
        d["uid"] = w["uid"]
    if "admission" in w:
        d["podsets"] = [{"count": 1, "totalRequests": ps["usage"], "flavors": ps["flavors"]} for ps in w["admission"]]
    else:
        d["podsets"] = [{"count": 1, "totalRequests": w["requests"], "flavors": {r: w["simpleFlavor"] for r in w["requests"]}}]
    return d


def extract(fname, func, out_cases, skipped):
    src = open(REF + fname).read()
    # full-line comments inside builder chains would stop the chain parser; blank them (line numbers kept)
    src = "\n".join("" if l.strip().startswith("//") else l for l in src.split("\n"))
    start = src.index("func %s(" % func)
    defaults = {}
    for m in re.finditer(r"(?m)^\t(\w+) := \[\]\*kueue\.ClusterQueue\{", src[start:]):
        p = start + m.end() - 1
        e = match_brace(src, p)
        defaults[m.group(1)] = [parse_cq(t) for t in list_items(src[p + 1:e], "MakeClusterQueue")]
    table_start = src.index("cases := map[string]struct", start)
    body_start = src.index("}{", table_start) + 1
    body_end = match_brace(src, body_start)
    table = src[body_start + 1: body_end]
    for m in re.finditer(r'(?m)^\t\t"((?:[^"\\]|\\.)*)":\s*\{', table):
        j = match_brace(table, m.end() - 1)
        block = table[m.end():j]
        name = m.group(1)
        line = src[: body_start + 1 + m.start()].count("\n") + 1
        try:
            cqf = field(block, "clusterQueues")
            if cqf.strip() in defaults:
                cqs = defaults[cqf.strip()]
            else:
                p = cqf.index("{")
                cqs = [parse_cq(t) for t in list_items(cqf[p + 1: match_brace(cqf, p)], "MakeClusterQueue")]
            cohorts = []
            cf = field(block, "cohorts")
            if cf:
                p = cf.index("{")
                cohorts = [parse_cohort(t) for t in list_items(cf[p + 1: match_brace(cf, p)], "MakeCohort")]
            admitted = workloads_in(field(block, "admitted") or "")
            incoming = workloads_in(field(block, "incoming"))[0]
            target = field(block, "targetCQ").strip().strip('"')
            asg = field(block, "assignment")
            if "singlePodSetAssignment" not in asg:
                skipped.append((name, "assignment is not singlePodSetAssignment")); continue
            fl = {}
            for r in re.finditer(r'([\w\."/-]+):\s*(?:&flavorassigner\.FlavorAssignment)?\{\s*Name:\s*"([^"]+)",\s*Mode:\s*flavorassigner\.(\w+)', asg):
                fl[res_name(r.group(1))] = [r.group(2), r.group(3)]
            want = workloads_in(field(block, "wantWorkloads") or "")
            if any("unsupported" in w for w in admitted + [incoming]):
                skipped.append((name, "explicit PodSets builder")); continue
            targets = sorted(f"{w['name']}:{w['preemptedReason']}" for w in want if "preemptedReason" in w)
            wp = field(block, "wantPreempted")
            if wp and int(wp.split()[0]) != len(targets):
                # a target that was ALREADY evicted gets no new Preempted condition (preemption "on going"):
                # the Go table only pins the count then
                targets = None
            case = {"name": name, "ref": f"pkg/scheduler/preemption/{fname}:{line}", "now": NOW, "clusterQueues": cqs, "cohorts": cohorts,
                    "admitted": [to_admitted(w) for w in admitted],
                    "pending": [{"name": "in", "uid": "wl-in", "cq": target, "priority": incoming["priority"], "created": incoming["created"],
                                 "podsets": [{"name": "main", "count": 1, "requests": incoming["requests"]}]}],
                    "assignment": [fl], "wantTargets": targets, "wantPreempted": int(wp.split()[0]) if wp else 0}
            out_cases.append(case)
        except Exception as ex:  # noqa: BLE001
            skipped.append((name, f"parse error: {ex!r}"))


def extract_fair(out_cases, skipped):
    fname, func = "preemption_fair_test.go", "TestFairPreemptions"
    src = open(REF + fname).read()
    src = "\n".join("" if l.strip().startswith("//") else l for l in src.split("\n"))
    start = src.index("func %s(" % func)
    defaults = {}
    for m in re.finditer(r"(?m)^\t(\w+) := \[\]\*kueue\.ClusterQueue\{", src[start:]):
        p = start + m.end() - 1
        e = match_brace(src, p)
        defaults[m.group(1)] = [parse_cq(t) for t in list_items(src[p + 1:e], "MakeClusterQueue")]
    table_start = src.index("cases := map[string]struct", start)
    body_start = src.index("}{", table_start) + 1
    body_end = match_brace(src, body_start)
    table = src[body_start + 1: body_end]
    unit = {"cpu": "1"}
    for m in re.finditer(r'(?m)^\t\t"((?:[^"\\]|\\.)*)":\s*\{', table):
        j = match_brace(table, m.end() - 1)
        block = table[m.end():j]
        name = m.group(1)
        line = src[: body_start + 1 + m.start()].count("\n") + 1
        try:
            if field(block, "flavors") or field(block, "assignmentFlavor"):
                skipped.append((name, "custom flavors / assignmentFlavor")); continue
            cqf = field(block, "clusterQueues")
            if cqf.strip() in defaults:
                cqs = defaults[cqf.strip()]
            else:
                p = cqf.index("{")
                cqs = [parse_cq(t) for t in list_items(cqf[p + 1: match_brace(cqf, p)], "MakeClusterQueue")]
            for q, t in zip(cqs, [None] * len(cqs)):
                pass
            cohorts = []
            cf = field(block, "cohorts")
            if cf:
                p = cf.index("{")
                cohorts = [parse_cohort(t) for t in list_items(cf[p + 1: match_brace(cf, p)], "MakeCohort")]
            admitted = workloads_in(field(block, "admitted") or "", unit)
            incoming = workloads_in(field(block, "incoming"), unit)[0]
            target = field(block, "targetCQ").strip().strip('"')
            strategies = re.findall(r"config\.(LessThan\w+)", field(block, "strategies") or "")
            wantf = field(block, "wantPreempted") or ""
            targets = sorted(f"{a.lstrip('/')}:{b.replace('Reason', '')}" for a, b in re.findall(r'targetKeyReason\("([^"]+)",\s*kueue\.(\w+)\)', wantf))
            for w in admitted:
                w["uid"] = w["name"]  # preemption_fair_test.go:1114-1117
            case = {"name": name, "ref": f"pkg/scheduler/preemption/{fname}:{line}", "now": NOW, "fairSharing": True, "clusterQueues": cqs, "cohorts": cohorts,
                    "admitted": [to_admitted(w) for w in admitted],
                    "pending": [{"name": incoming["name"], "cq": target, "priority": incoming["priority"], "created": incoming["created"],
                                 "podsets": [{"name": "main", "count": 1, "requests": incoming["requests"]}]}],
                    "assignment": [{"cpu": ["default", "Preempt"]}], "wantTargets": targets, "wantPreempted": len(targets)}
            if strategies:
                case["fsStrategies"] = strategies
            out_cases.append(case)
        except Exception as ex:  # noqa: BLE001
            skipped.append((name, f"parse error: {ex!r}"))


def main():
    cases, skipped = [], []
    extract("preemption_test.go", "TestPreemption", cases, skipped)
    extract("preemption_hierarchical_test.go", "TestHierarchicalPreemptions", cases, skipped)
    fair, fskipped = [], []
    extract_fair(fair, fskipped)
    outf = os.path.join(HERE, "preemption_fair.yaml")
    with open(outf, "w") as f:
        f.write("# GENERATED by tests/golden/extract_preemption.py from /root/reference/pkg/scheduler/preemption/preemption_fair_test.go (TestFairPreemptions :45)\n")
        yaml.safe_dump({"cases": fair, "skipped": [{"name": n, "why": w} for n, w in fskipped]}, f, sort_keys=False, width=200)
    print(f"{len(fair)} fair cases transcribed, {len(fskipped)} skipped -> {outf}")
    for n, w in fskipped:
        print("  skipped:", n, "--", w)
    out = os.path.join(HERE, "preemption.yaml")
    with open(out, "w") as f:
        f.write("# GENERATED by tests/golden/extract_preemption.py from /root/reference/pkg/scheduler/preemption/\n"
                "# preemption_test.go (TestPreemption :65) and preemption_hierarchical_test.go (TestHierarchicalPreemptions :42)\n")
        yaml.safe_dump({"cases": cases, "skipped": [{"name": n, "why": w} for n, w in skipped]}, f, sort_keys=False, width=200)
    print(f"{len(cases)} cases transcribed, {len(skipped)} skipped -> {out}")
    for n, w in skipped:
        print("  skipped:", n, "--", w)


if __name__ == "__main__":
    main()

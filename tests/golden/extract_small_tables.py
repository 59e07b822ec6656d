#!/usr/bin/env python
"""Transcribe four small table tests of the reference into tests/golden/small_tables.yaml:
  TestSearch (PodSetReducer) pkg/scheduler/flavorassigner/podset_reducer_test.go:27
  TestIsPreferred            pkg/scheduler/flavorassigner/flavorassigner_test.go:4184
  TestResourcesToReserve     pkg/scheduler/scheduler_test.go:8692
  TestLastAssignmentOutdated pkg/scheduler/scheduler_test.go:9216
Run in the build container (needs /root/reference):  python tests/golden/extract_small_tables.py
"""
import os
import re
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_tas import elements, match_brace, strip_comments, top_level_fields, top_level_fields_generic  # noqa: E402

REF = "/root/reference/pkg/scheduler"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_tables.yaml")


def func_body(path, name, until=None):
    src = strip_comments(open(path).read())
    a = src.index(f"func {name}(")
    b = src.index("\nfunc ", a + 10)
    return src[a:b]


def is_preferred():
    body = func_body(f"{REF}/flavorassigner/flavorassigner_test.go", "TestIsPreferred")
    m = re.search(r"cases := map\[string\]struct \{", body)
    close = match_brace(body, m.end() - 1)
    o = body.index("{", close + 1)
    c = match_brace(body, o)
    out = []
    for el in elements(body[o + 1:c]):
        nm = re.match(r'"((?:[^"\\]|\\.)*)"\s*:\s*\{', el)
        f = top_level_fields(el[nm.end():el.rindex("}")])

        def gm(t):
            g = top_level_fields(t[t.index("{") + 1:t.rindex("}")])
            return dict(mode=g["preemptionMode"], borrow=int(g["borrowingLevel"]))
        case = dict(name=nm.group(1), a=gm(f["a"]), b=gm(f["b"]), want=f["wantPreferred"] == "true")
        cfg = {}
        if "config" in f:
            g = top_level_fields(f["config"][f["config"].index("{") + 1:f["config"].rindex("}")])
            for k, v in g.items():
                cfg[k] = re.sub(r"makePref\(kueue\.(\w+)\)", r"\1", v).replace("kueue.", "")
        case["config"] = cfg
        out.append(case)
    return out


def fr_map(text):
    """resources.FlavorResourceQuantities{ {Flavor: ..("x"), Resource: y}: resources.NewAmount(n), ...} -> {"x/res": n}"""
    out = {}
    inner = text[text.index("{") + 1:text.rindex("}")]
    for k, v in top_level_fields_generic(inner):
        fl = re.search(r'Flavor:\s*kueue\.ResourceFlavorReference\("([^"]+)"\)', k).group(1)
        rs = re.search(r'Resource:\s*([^}\s]+)', k).group(1)
        rs = {"corev1.ResourceMemory": "memory", "corev1.ResourceCPU": "cpu"}.get(rs, rs.strip('"'))
        out[f"{fl}/{rs}"] = int(re.search(r"NewAmount\((-?\d+)\)", v).group(1))
    return out


def resources_to_reserve():
    body = func_body(f"{REF}/scheduler_test.go", "TestResourcesToReserve")
    m = re.search(r"cases := \[\]struct \{", body)
    close = match_brace(body, m.end() - 1)
    o = body.index("{", close + 1)
    c = match_brace(body, o)
    out = []
    for el in elements(body[o + 1:c]):
        f = top_level_fields(el[el.index("{") + 1:el.rindex("}")])
        out.append(dict(name=f["name"].strip('"'), mode=f["assignmentMode"].split(".")[-1], borrowing=int(f.get("borrowing", "0")),
                        assignmentUsage=fr_map(f["assignmentUsage"]), cqUsage=fr_map(f["cqUsage"]), wantReserved=fr_map(f["wantReserved"])))
    return out


def last_assignment_outdated():
    body = func_body(f"{REF}/scheduler_test.go", "TestLastAssignmentOutdated")
    m = re.search(r"tests := \[\]struct \{", body)
    close = match_brace(body, m.end() - 1)
    o = body.index("{", close + 1)
    c = match_brace(body, o)
    out = []
    for el in elements(body[o + 1:c]):
        f = top_level_fields(el[el.index("{") + 1:el.rindex("}")])
        a = top_level_fields(f["args"][f["args"].index("{") + 1:f["args"].rindex("}")])
        last = top_level_fields(a["last"][a["last"].index("{") + 1:a["last"].rindex("}")])

        def num(t):
            mm = re.search(r"(-?\d+)", t or "0")
            return int(mm.group(1)) if mm else 0

        def shape(t):  # workload.EquivalenceHash is a string; the boundary carries a 64-bit hash, 0 = unknown (SchedulingHashUnknown)
            if not t or "Unknown" in t or t.strip() == '""':
                return 0
            return 1 + ["shape-a", "shape-b", "shape-c"].index(t.strip().strip('"'))
        out.append(dict(name=f["name"].strip('"'), preserveProgress=f.get("preserveProgress", "false") == "true",
                        cycle=num(a.get("currentSchedulingCycle")), cqGeneration=num(a.get("currentCQGeneration")),
                        hash=shape(a.get("currentSchedulingHash")),
                        last=dict(generation=num(last.get("ClusterQueueGeneration")), cycle=num(last.get("SchedulingCycle")),
                                  hash=shape(last.get("SchedulingHash"))),
                        want=f["want"] == "true"))
    return out


def podset_reducer_search():
    body = func_body(f"{REF}/flavorassigner/podset_reducer_test.go", "TestSearch") if False else None
    src = strip_comments(open(f"{REF}/flavorassigner/podset_reducer_test.go").read())
    a = src.index("func TestSearch(")
    body = src[a:]
    m = re.search(r"cases := map\[string\]struct \{", body)
    close = match_brace(body, m.end() - 1)
    o = body.index("{", close + 1)
    c = match_brace(body, o)
    out = []
    for el in elements(body[o + 1:c]):
        nm = re.match(r'"((?:[^"\\]|\\.)*)"\s*:\s*\{', el)
        f = top_level_fields(el[nm.end():el.rindex("}")])
        podsets = []
        for pm in re.finditer(r'MakePodSet\("[^"]+",\s*([\d_]+)\)((?:\.\s*SetMinimumCount\(([\d_]+)\))?)', f["podSets"]):
            podsets.append(dict(count=int(pm.group(1).replace("_", "")), minCount=(int(pm.group(3).replace("_", "")) if pm.group(3) else None)))
        out.append(dict(name=nm.group(1), podSets=podsets, countLimit=int(f["countLimit"].replace("_", "")),
                        wantCount=int(f["wantCount"].replace("_", "")), wantFound=f["wantFound"] == "true"))
    return out


def sorted_domains(func):
    """TestSortedDomains / TestSortedDomainsWithLeader (pkg/cache/scheduler/tas_flavor_snapshot_test.go:884 / :601)."""
    src = strip_comments(open("/root/reference/pkg/cache/scheduler/tas_flavor_snapshot_test.go").read())
    a = src.index(f"func {func}(")
    body = src[a:src.index("\nfunc ", a + 10)]
    m = re.search(r"testCases := map\[string\]struct \{", body)
    close = match_brace(body, m.end() - 1)
    o = body.index("{", close + 1)
    c = match_brace(body, o)
    out = []
    for el in elements(body[o + 1:c]):
        nm = re.match(r'"((?:[^"\\]|\\.)*)"\s*:\s*\{', el)
        f = top_level_fields(el[nm.end():el.rindex("}")])
        doms = []
        dl = f["domains"]
        for d in elements(dl[dl.index("{") + 1:dl.rindex("}")]):
            did = re.search(r'id:\s*"([^"]+)"', d).group(1)
            lv = re.search(r'levelValues:\s*\[\]string\{"([^"]+)"\}', d).group(1)
            st = {k: int(v) for k, v in re.findall(r"(affinityScore|sliceCount|podCount|leaderCount|sliceCountWithLeader|podCountWithLeader):\s*(\d+)", d)}
            doms.append(dict(id=did, levelValue=lv, **st))
        out.append(dict(name=nm.group(1), affinityGate=f.get("enableTASPreferredSchedulingAffinity", "false") == "true",
                        unconstrained=f.get("unconstrained", "false") == "true", domains=doms,
                        want=re.findall(r'"([^"]+)"', f["wantOrder"])))
    return out


def main():
    doc = dict(sortedDomains=sorted_domains("TestSortedDomains"), sortedDomainsWithLeader=sorted_domains("TestSortedDomainsWithLeader"),
               isPreferred=is_preferred(), resourcesToReserve=resources_to_reserve(), lastAssignmentOutdated=last_assignment_outdated(),
               podSetReducerSearch=podset_reducer_search())
    with open(OUT, "w") as fh:
        fh.write("# Generated by tests/golden/extract_small_tables.py from the reference's TestIsPreferred, TestResourcesToReserve,\n"
                 "# TestLastAssignmentOutdated, TestSearch (PodSetReducer), TestSortedDomains and TestSortedDomainsWithLeader tables.\n")
        yaml.safe_dump(doc, fh, sort_keys=False, width=160)
    print({k: len(v) for k, v in doc.items()})


if __name__ == "__main__":
    main()

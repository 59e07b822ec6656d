#!/usr/bin/env python
"""Transcribe the node-replacement vectors of pkg/cache/scheduler/tas_cache_test.go into tests/golden/tas_replacement.yaml:
TestFindTopologyAssignmentsMultiLayerReplacement (:8450, every case) and the cases of TestFindTopologyAssignments (:61) whose
workload has Status.UnhealthyNodes.

Run in the build container (needs /root/reference):  python tests/golden/extract_tas_replacement.py
priorFlavorUsage / aggregatedDomainUsages (TASHandleOverlappingFlavors, default on) reach the placement as initial assumed usage
(tas_flavor_snapshot.go:587-589); a replacement is never placed with simulate-empty, so at the boundary they are leaf usage: the
vectors carry them as `priorUsage` and the tests put them into the usage table. A case whose result is decided on the host side
(SkipReassignmentForPodOwnedWorkloads for a workload owned by a single Pod, :615: the existing assignment is returned without a
placement) is skipped and counted.
"""
import collections
import os
import re
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import extract_tas as X  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tas_replacement.yaml")


def parse_ta(text):
    """MakeTopologyAssignment(levels).Domain(tas.TopologyDomainAssignment{Count: n, Values: []string{..}})... -> [{values, count}]"""
    doms = []
    for m in re.finditer(r"TopologyDomainAssignment\{", text):
        close = X.match_brace(text, m.end() - 1)
        if text[m.start() - 6:m.start()].endswith("[]tas."):   # a slice literal: its elements are the domains
            for el in X.elements(text[m.end():close]):
                df = X.top_level_fields(el[el.index("{") + 1:el.rindex("}")])
                vals = df["Values"]
                doms.append(dict(count=int(df["Count"]), values=[X.ident(x) for x in X.elements(vals[vals.index("{") + 1:vals.rindex("}")])]))
            continue
        df = X.top_level_fields(text[m.end():close])
        vals = df["Values"]
        doms.append(dict(count=int(df["Count"]), values=[X.ident(x) for x in X.elements(vals[vals.index("{") + 1:vals.rindex("}")])]))
    return doms


def parse_usage(text):
    """[]workload.TopologyDomainRequests{{Values: .., SinglePodRequests: resources.NewRequestsFromMap(map[..]int64{..}), Count: n}}"""
    out = []
    inner = text[text.index("{") + 1:text.rindex("}")]
    for el in X.elements(inner):
        f = X.top_level_fields(el[el.index("{") + 1:el.rindex("}")])
        vals = f["Values"]
        rq = f["SinglePodRequests"]
        m = rq[rq.index("{") + 1:rq.rindex("}")]
        out.append(dict(values=[X.ident(x) for x in X.elements(vals[vals.index("{") + 1:vals.rindex("}")])],
                        requests={X.ident(k): X.int_expr(v) for k, v in X.top_level_fields_generic(m)}, count=int(f["Count"])))
    return out


def topology_request(t):
    """the kueue.PodSetTopologyRequest literal through extract_tas.parse_podset's field parser"""
    return X.parse_podset("topologyRequest: " + t + ",", {}).get("topologyRequest")


def table(src, func):
    start = src.index("func " + func + "(")
    nxt = re.search(r"\nfunc ", src[start + 5:])
    body = src[start:start + 5 + nxt.start()] if nxt else src[start:]
    cm = re.search(r"cases := map\[string\]struct \{", body)
    sclose = X.match_brace(body, cm.end() - 1)
    open_cases = body.index("{", sclose + 1)
    return body, body[open_cases + 1:X.match_brace(body, open_cases)]


def main():
    src = X.strip_comments(open(X.SRC).read())
    cases, skipped = [], collections.Counter()
    # ---- TestFindTopologyAssignmentsMultiLayerReplacement
    body, text = table(src, "TestFindTopologyAssignmentsMultiLayerReplacement")
    for m in re.finditer(r'(\w+)\s*=\s*"([^"]+)"', body[:body.index("cases :=")]):
        X.CONSTS[m.group(1)] = m.group(2)
    named_levels = {}
    for m in re.finditer(r"\n\t(\w+) := \[\]string\{", body):
        close = X.match_brace(body, m.end() - 1)
        named_levels[m.group(1)] = [X.ident(e) for e in X.elements(body[m.end():close])]
    for el in X.elements(text):
        nm = re.match(r'"((?:[^"\\]|\\.)*)"\s*:\s*\{', el)
        name = nm.group(1)
        f = X.top_level_fields(el[nm.end():el.rindex("}")])
        try:
            if "featureGates" in f:
                raise X.Skip("gate")
            case = dict(name=name, table="TestFindTopologyAssignmentsMultiLayerReplacement")
            n = f["nodes"]
            case["nodes"] = X.parse_nodes(n[n.index("{") + 1:n.rindex("}")])
            case["levels"] = X.parse_strings(f["levels"], named_levels) if "levels" in f else named_levels["defaultLevels"]
            if "pods" in f:
                p = f["pods"]
                case["nonTASUsage"] = X.parse_pods(p[p.index("{") + 1:p.rindex("}")])
            case["unhealthyNode"] = X.ident(f["unhealthyNode"])
            prior = []
            if "priorFlavorUsage" in f:
                prior += parse_usage(f["priorFlavorUsage"])
            if "aggregatedDomainUsages" in f:
                a = f["aggregatedDomainUsages"]
                for k, v in X.top_level_fields_generic(a[a.index("{") + 1:a.rindex("}")]):
                    mm = v[v.index("{") + 1:v.rindex("}")]
                    # the map key is a TopologyDomainID: the hostname on a hostname-level topology
                    prior.append(dict(values=[X.ident(k)], total={X.ident(k2): X.int_expr(v2) for k2, v2 in X.top_level_fields_generic(mm)}))
            if prior:
                case["priorUsage"] = prior
            ps = dict(name="main", count=int(f["count"]), requests={"cpu": 1000}, existing=parse_ta(f["existingTA"]))
            if "topologyRequest" in f and f["topologyRequest"].strip() != "nil":
                ps["topologyRequest"] = topology_request(f["topologyRequest"])
            if "wantReason" in f:
                w = f["wantReason"].strip()
                ps["wantReason"] = w[1:-1].replace('\\"', '"') if w.startswith('"') else w[1:-1]
            if "wantAssignment" in f and f["wantAssignment"].strip() != "nil":
                ps["wantAssignment"] = dict(domains=parse_ta(f["wantAssignment"]))
            case["podSets"] = [ps]
            cases.append(case)
        except X.Skip as e:
            skipped[str(e)] += 1
    # ---- TestFindTopologyAssignments: the cases with a workload that has UnhealthyNodes
    body, text = table(src, "TestFindTopologyAssignments")
    for m in re.finditer(r'(\w+)\s*=\s*"([^"]+)"', body[:body.index("defaultNodes")]):
        X.CONSTS[m.group(1)] = m.group(2)
    for el in X.elements(text):
        nm = re.match(r'"((?:[^"\\]|\\.)*)"\s*:\s*\{', el)
        if not nm:
            continue
        f = X.top_level_fields(el[nm.end():el.rindex("}")])
        if "workload" not in f or "UnhealthyNodes(" not in f["workload"]:
            continue
        try:
            wl = f["workload"]
            gates = f.get("featureGates", "")
            owners = re.findall(r'OwnerReference\(([^\n]*)\)\.', wl)
            pod_owners = [o for o in owners if 'WithKind("Pod")' in o]
            if "SkipReassignmentForPodOwnedWorkloads: true" in gates and len(owners) == 1 and len(pod_owners) == 1 and "is-group-workload" not in wl \
                    and "Annotation" not in wl:
                raise X.Skip("SkipReassignmentForPodOwnedWorkloads (host-side result)")
            case = dict(name=nm.group(1), table="TestFindTopologyAssignments")
            n = f["nodes"]
            case["nodes"] = X.parse_nodes(n[n.index("{") + 1:n.rindex("}")])
            case["levels"] = X.parse_strings(f["levels"], {})
            case["unhealthyNode"] = re.search(r'UnhealthyNodes\("([^"]+)"\)', wl).group(1)
            existing = parse_ta(wl)
            ps_text = f["podSets"]
            pss = [X.parse_podset(e[e.index("{") + 1:e.rindex("}")], {}) for e in X.elements(ps_text[ps_text.index("{") + 1:ps_text.rindex("}")])]
            if len(pss) != 1:
                raise X.Skip("several podsets")
            pss[0]["existing"] = existing
            if "wantAssignment" in pss[0]:
                pss[0]["wantAssignment"] = dict(domains=pss[0]["wantAssignment"]["domains"])
            case["podSets"] = pss
            cases.append(case)
        except X.Skip as e:
            skipped[str(e)] += 1
    hdr = ("# Generated by tests/golden/extract_tas_replacement.py from pkg/cache/scheduler/tas_cache_test.go\n"
           "# (TestFindTopologyAssignmentsMultiLayerReplacement, and the UnhealthyNodes cases of TestFindTopologyAssignments).\n"
           f"# {len(cases)} cases kept; skipped: {dict(skipped)}\n")
    with open(OUT, "w") as fh:
        fh.write(hdr)
        yaml.safe_dump(dict(cases=cases), fh, sort_keys=False, width=160)
    print(len(cases), "cases;", dict(skipped))


if __name__ == "__main__":
    main()

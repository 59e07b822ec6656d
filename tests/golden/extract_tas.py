#!/usr/bin/env python
"""Transcribe TestFindTopologyAssignments (pkg/cache/scheduler/tas_cache_test.go:61) into tests/golden/tas_find.yaml.

Run in the build container (needs /root/reference):  python tests/golden/extract_tas.py
Only cases that stay on the path the engine covers are kept: plain Nodes (labels + allocatable, Ready / NotReady /
Unschedulable), level lists, podsets with Required / Preferred / Unconstrained / single-layer slices / podset groups,
`pods` giving non-TAS usage, feature gates limited to TASProfileMixed. Cases using taints, tolerations, node selectors,
node affinity, previous assignments, workloads with unhealthy nodes, prior usage, balanced placement, multi-layer
topology or other gates are counted and skipped (the reason is written into the YAML header).
"""
import collections
import os
import re
import sys

import yaml

SRC = "/root/reference/pkg/cache/scheduler/tas_cache_test.go"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tas_find.yaml")


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return "\n".join(re.sub(r"(^|\s)//.*$", "", ln) for ln in s.split("\n"))


def match_brace(s, i, open_ch="{", close_ch="}"):
    """s[i] == open_ch -> index of the matching close, skipping string literals."""
    depth, j, n = 0, i, len(s)
    while j < n:
        c = s[j]
        if c == '"':
            j += 1
            while s[j] != '"':
                j += 2 if s[j] == "\\" else 1
        elif c == "`":
            j = s.index("`", j + 1)
        elif c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return j
        j += 1
    raise ValueError("unbalanced")


def top_level_fields(body):
    """'name: value,' pairs at depth 0 of a composite literal body -> {name: value_text}"""
    out, i, n = collections.OrderedDict(), 0, len(body)
    while i < n:
        m = re.compile(r"\s*([A-Za-z_][A-Za-z0-9_]*)\s*:\s*").match(body, i)
        if not m:
            break
        name, j = m.group(1), m.end()
        k, depth = j, 0
        while k < n:
            c = body[k]
            if c == '"':
                k += 1
                while body[k] != '"':
                    k += 2 if body[k] == "\\" else 1
            elif c == "`":
                k = body.index("`", k + 1)
            elif c in "{([":
                depth += 1
            elif c in "})]":
                depth -= 1
            elif c == "," and depth == 0:
                break
            k += 1
        out[name] = body[j:k].strip()
        i = k + 1
    return out


def elements(body):
    """comma-separated elements at depth 0"""
    out, k, depth, start, n = [], 0, 0, 0, len(body)
    while k < n:
        c = body[k]
        if c == '"':
            k += 1
            while body[k] != '"':
                k += 2 if body[k] == "\\" else 1
        elif c == "`":
            k = body.index("`", k + 1)
        elif c in "{([":
            depth += 1
        elif c in "})]":
            depth -= 1
        elif c == "," and depth == 0:
            if body[start:k].strip():
                out.append(body[start:k].strip())
            start = k + 1
        k += 1
    if body[start:].strip():
        out.append(body[start:].strip())
    return out


CONSTS = {"corev1.LabelHostname": "kubernetes.io/hostname", "corev1.ResourceCPU": "cpu", "corev1.ResourceMemory": "memory",
          "corev1.ResourcePods": "pods"}


def ident(tok):
    tok = tok.strip()
    m = re.fullmatch(r"(?:string|corev1\.ResourceName|kueue\.\w+)\((.*)\)", tok)   # a conversion around a literal or constant
    if m:
        tok = m.group(1).strip()
    if tok.startswith('"'):
        return tok[1:-1]
    if tok in CONSTS:
        return CONSTS[tok]
    raise KeyError(tok)


class Skip(Exception):
    pass


def parse_nodes(text):
    """[]corev1.Node{ *testingnode.MakeNode("..").Label(..)...Obj(), ... } -> list of dicts"""
    nodes = []
    for el in elements(text):
        m = re.match(r'\*testingnode\.MakeNode\("([^"]+)"\)', el)
        if not m:
            raise Skip("node literal: " + el[:40])
        node = dict(name=m.group(1), labels={}, allocatable={}, ready=False)
        rest = el[m.end():]
        pos = 0
        while pos < len(rest):
            mm = re.compile(r"\s*\.\s*([A-Za-z]+)\(").match(rest, pos)
            if not mm:
                break
            meth = mm.group(1)
            close = match_brace(rest, mm.end() - 1, "(", ")")
            args = rest[mm.end():close]
            pos = close + 1
            if meth == "Label":
                k, v = elements(args)
                node["labels"][ident(k)] = ident(v)
            elif meth == "StatusAllocatable":
                inner = args[args.index("{") + 1:args.rindex("}")]
                for k, v in top_level_fields_generic(inner):
                    node["allocatable"][ident(k)] = re.search(r'MustParse\("([^"]+)"\)', v).group(1)
            elif meth == "Ready":
                node["ready"] = True
            elif meth == "NotReady":
                node["ready"] = False
            elif meth == "Unschedulable":
                node["unschedulable"] = True
            elif meth == "StatusConditions":
                cf = top_level_fields(args[args.index("{") + 1:args.rindex("}")])
                if cf.get("Type", "").strip() == "corev1.NodeReady":      # any other condition leaves readiness alone
                    node["ready"] = cf.get("Status", "").strip() == "corev1.ConditionTrue"
            elif meth == "Taints":
                for t in elements(args):
                    tf = top_level_fields(t[t.index("{") + 1:t.rindex("}")])
                    node.setdefault("taints", []).append(dict(key=ident(tf.get("Key", '""')), value=ident(tf.get("Value", '""')),
                                                              effect=tf.get("Effect", "").strip().replace("corev1.TaintEffect", "")))
            elif meth == "Obj":
                pass
            else:
                raise Skip("node method " + meth)
        nodes.append(node)
    return nodes


def top_level_fields_generic(body):
    """key: value pairs where the key may be a quoted string or a dotted identifier"""
    out = []
    for el in elements(body):
        k, depth = 0, 0
        while k < len(el):
            c = el[k]
            if c == '"':
                k += 1
                while el[k] != '"':
                    k += 1
            elif c in "{([":
                depth += 1
            elif c in "})]":
                depth -= 1
            elif c == ":" and depth == 0:
                break
            k += 1
        out.append((el[:k].strip(), el[k + 1:].strip()))
    return out


def parse_strings(text, named):
    text = text.strip()
    if text in named:
        return named[text]
    m = re.fullmatch(r"append\((\[\]string\{.*?\}),\s*(\w+)\.\.\.\)", text, flags=re.S)
    if m:  # append([]string{a, b}, named...)
        return parse_strings(m.group(1), named) + list(named[m.group(2)])
    inner = text[text.index("{") + 1:text.rindex("}")]
    return [ident(e) for e in elements(inner)]


def new_arg(v):
    m = re.fullmatch(r"new\((.*)\)", v.strip(), flags=re.S)
    if not m:
        m = re.fullmatch(r"ptr\.To\((.*)\)", v.strip(), flags=re.S)
    if not m:
        raise Skip("pointer literal " + v[:30])
    return m.group(1).strip()


def int_expr(v):
    """an integer literal or a product of them (64 * 1024 * 1024 * 1024)"""
    out = 1
    for t in v.split("*"):
        out *= int(t.strip())
    return out


def parse_podset(body, named_levels):
    f = top_level_fields(body)
    for bad in ("previousAssignment",):
        if bad in f:
            raise Skip(bad)
    ps = dict(name=ident(f["podSetName"]) if "podSetName" in f else "", count=int(f.get("count", "0")))
    ps["requests"] = {}
    if "nodeSelector" in f and f["nodeSelector"] != "nil":
        # PodSpec.NodeSelector: the placement only counts the leaves (hostname level) whose node carries every label (the host's feasibility
        # mask, kq_tas_requests.leaf_ok; tas_flavor_snapshot.go:963 with isLowestLevelNode)
        ns = f["nodeSelector"]
        ps["nodeSelector"] = {ident(k): ident(v) for k, v in top_level_fields_generic(ns[ns.index("{") + 1:ns.rindex("}")])}
    if "nodeAffinity" in f and f["nodeAffinity"] != "nil":
        # corev1.NodeAffinity in the builder forms the table uses: MakeNodeSelectorTerms().Term(key, op, values...) (required: the terms are
        # ORed, one expression each) and MakePreferredSchedulingTerms().Term(weight, key, op, values...) (features.TASRespectNodeAffinityPreferred)
        na = f["nodeAffinity"]
        aff = {}
        def terms_of(text, with_weight):
            out = []
            for tm in re.finditer(r"\.\s*Term\(", text):
                o = tm.end() - 1
                args = [a.strip() for a in elements(text[o + 1:match_brace(text, o, "(", ")")])]
                if with_weight:
                    w, args = int(args[0]), args[1:]
                t = dict(key=ident(args[0]), operator=args[1].replace("corev1.NodeSelectorOp", ""), values=[ident(a) for a in args[2:]])
                if with_weight:
                    t["weight"] = w
                out.append(t)
            return out
        rm = re.search(r"MakeNodeSelectorTerms\(\)", na)
        pm = re.search(r"MakePreferredSchedulingTerms\(\)", na)
        if "RequiredDuringSchedulingIgnoredDuringExecution" in na and not rm:
            raise Skip("nodeAffinity form")
        if rm:
            aff["required"] = terms_of(na[rm.end():pm.start() if pm and pm.start() > rm.end() else len(na)], False)
        if pm:
            aff["preferred"] = terms_of(na[pm.end():rm.start() if rm and rm.start() > pm.end() else len(na)], True)
        if not aff:
            raise Skip("nodeAffinity form")
        ps["nodeAffinity"] = aff
    if "tolerations" in f and f["tolerations"] != "nil":
        tl = f["tolerations"]
        ps["tolerations"] = []
        for t in elements(tl[tl.index("{") + 1:tl.rindex("}")]):
            tf = top_level_fields(t[t.index("{") + 1:t.rindex("}")])
            ps["tolerations"].append(dict(key=ident(tf.get("Key", '""')), value=ident(tf.get("Value", '""')),
                                          operator=tf.get("Operator", "corev1.TolerationOpEqual").strip().replace("corev1.TolerationOp", ""),
                                          effect=tf.get("Effect", "").strip().replace("corev1.TaintEffect", "")))
    if "requests" in f:
        inner = f["requests"][f["requests"].index("{") + 1:f["requests"].rindex("}")]
        for k, v in top_level_fields_generic(inner):
            ps["requests"][ident(k)] = int_expr(v)
    if "topologyRequest" in f and f["topologyRequest"] != "nil":
        t = f["topologyRequest"]
        tf = top_level_fields(t[t.index("{") + 1:t.rindex("}")])
        tr = {}
        for k, v in tf.items():
            if k == "Required":
                tr["required"] = ident(new_arg(v))
            elif k == "Preferred":
                tr["preferred"] = ident(new_arg(v))
            elif k == "Unconstrained":
                tr["unconstrained"] = new_arg(v) == "true"
            elif k == "PodSetSliceRequiredTopology":
                tr["sliceRequiredTopology"] = ident(new_arg(v))
            elif k == "PodsetSliceRequiredTopologyConstraints":
                inner = v[v.index("{", v.index("PodsetSliceRequiredTopologyConstraint{") ) + 1:v.rindex("}")]
                cons = []
                for e in elements(inner):
                    ef = top_level_fields(e[e.index("{") + 1:e.rindex("}")])
                    cons.append(dict(topology=ident(ef["Topology"]), size=int(ef["Size"])))
                tr["sliceConstraints"] = cons
            elif k == "PodSetSliceSize":
                tr["sliceSize"] = int(re.search(r"(-?\d+)\)?$", re.sub(r"int32\(", "", new_arg(v))).group(1))
            else:
                raise Skip("topologyRequest." + k)
        ps["topologyRequest"] = tr
    if "podSetGroupName" in f:
        ps["group"] = ident(new_arg(f["podSetGroupName"]))
    if "wantReason" in f:
        w = f["wantReason"].strip()
        if not (w.startswith('"') or w.startswith("`")):
            raise Skip("computed wantReason")
        ps["wantReason"] = w[1:-1].replace('\\"', '"')
    if "wantAssignment" in f and f["wantAssignment"] != "nil":
        w = f["wantAssignment"]

        def domain(body):
            df = top_level_fields(body)
            vals = df["Values"]
            return dict(count=int(df["Count"]), values=[ident(x) for x in elements(vals[vals.index("{") + 1:vals.rindex("}")])])

        bm = re.search(r"MakeTopologyAssignment\(", w)
        if bm:
            # the builder form: utiltestingapi.MakeTopologyAssignment([]string{levels}).Domain(tas.TopologyDomainAssignment{Count, Values})....TopologyAssignment
            close = match_brace(w, bm.end() - 1, "(", ")")
            levels = parse_strings(w[bm.end():close], named_levels)
            doms, i = [], close
            for dm in re.finditer(r"\.\s*Domain\(", w[close:]):
                o = close + dm.end() - 1
                arg = w[o + 1:match_brace(w, o, "(", ")")]
                doms.append(domain(arg[arg.index("{") + 1:arg.rindex("}")]))
            ps["wantAssignment"] = dict(levels=levels, domains=doms)
        else:
            wf = top_level_fields(w[w.index("{") + 1:w.rindex("}")])
            dtext = wf["Domains"]
            doms = [domain(el[el.index("{") + 1:el.rindex("}")]) for el in elements(dtext[dtext.index("{") + 1:dtext.rindex("}")])]
            ps["wantAssignment"] = dict(levels=parse_strings(wf["Levels"], named_levels), domains=doms)
    return ps


def parse_pods(text):
    """[]corev1.Pod{ *testingpod.MakePod(..).NodeName("x").Request(res, "q")...Obj() } -> non-TAS usage per node"""
    usage = {}
    for el in elements(text):
        if not re.match(r"\*testingpod\.MakePod\(", el):
            raise Skip("pod literal")
        node = None
        reqs = {}
        phase_done = False
        for mm in re.finditer(r"\.\s*([A-Za-z]+)\(", el):
            meth = mm.group(1)
            close = match_brace(el, mm.end() - 1, "(", ")")
            args = el[mm.end():close]
            if meth == "NodeName":
                node = ident(args)
            elif meth == "Request":
                k, v = elements(args)
                reqs[ident(k)] = ident(v)
            elif meth == "StatusPhase":
                if "Succeeded" in args or "Failed" in args:
                    phase_done = True
            elif meth in ("MakePod", "Obj", "Clone"):
                pass
            else:
                raise Skip("pod method " + meth)
        if node is None or phase_done:
            continue
        u = usage.setdefault(node, collections.Counter())
        for k, v in reqs.items():
            u[k] += v if isinstance(v, int) else 0
            usage[node][k] = usage[node].get(k, 0)  # placeholder, replaced below
        usage[node] = dict(usage[node])
        usage.setdefault("__raw__", []).append((node, reqs))
    raw = usage.pop("__raw__", [])
    out = {}
    for node, reqs in raw:
        d = out.setdefault(node, {})
        for k, v in reqs.items():
            d.setdefault(k, []).append(v)
        d.setdefault("pods", []).append("1")
    return out


def main():
    src = strip_comments(open(SRC).read())
    start = src.index("func TestFindTopologyAssignments(")
    end = src.index("func TestFindTopologyAssignmentsMultiLayerReplacement(")
    body = src[start:end]
    for m in re.finditer(r'(\w+)\s*=\s*"([^"]+)"', body[:body.index("defaultNodes")]):
        CONSTS[m.group(1)] = m.group(2)
    # named node sets and level lists
    named_nodes, named_levels = {}, {}
    for m in re.finditer(r"\n\t(\w+) := \[\]corev1\.Node\{", body):
        close = match_brace(body, m.end() - 1)
        try:
            named_nodes[m.group(1)] = parse_nodes(body[m.end():close])
        except Skip as e:
            named_nodes[m.group(1)] = e
    for m in re.finditer(r"\n\t(\w+) := \[\]string\{", body):
        close = match_brace(body, m.end() - 1)
        named_levels[m.group(1)] = [ident(e) for e in elements(body[m.end():close])]
    cm = re.search(r"cases := map\[string\]struct \{", body)
    sclose = match_brace(body, cm.end() - 1)
    open_cases = body.index("{", sclose + 1)
    close_cases = match_brace(body, open_cases)
    cases_text = body[open_cases + 1:close_cases]
    cases, skipped = [], collections.Counter()
    aff_cases = []
    for el in elements(cases_text):
        nm = re.match(r'"((?:[^"\\]|\\.)*)"\s*:\s*\{', el)
        if not nm:
            skipped["unparsed"] += 1
            continue
        name = nm.group(1)
        f = top_level_fields(el[nm.end():el.rindex("}")])
        try:
            case = dict(name=name)
            if "nodeLabels" in f:  # flavorInformation.NodeLabels: the flavor only selects nodes carrying these labels
                nl = f["nodeLabels"]
                case["nodeLabels"] = {ident(k): ident(v) for k, v in top_level_fields_generic(nl[nl.index("{") + 1:nl.rindex("}")])}
            for bad in ("workload", "aggregatedDomainUsages"):
                if bad in f:
                    raise Skip(bad)
            prior = {}
            for key in ("priorOwnUsage", "priorFlavorUsage"):   # []workload.TopologyDomainRequests: TAS usage the flavor / a sibling flavor over the same nodes holds
                if key in f:
                    prior[key] = [dict(values=[ident(v) for v in re.findall(r'"[^"]*"|[\w\.]+', vals)],
                                       requests={ident(r.strip()): int(q) for r, q in re.findall(r'([\w\.]+|"[^"]+")\s*:\s*(\d+)', reqs)}, count=int(cnt))
                                  for vals, reqs, cnt in re.findall(r'Values:\s*\[\]string\{([^}]*)\},\s*SinglePodRequests:\s*resources\.NewRequestsFromMap\(map\[corev1\.ResourceName\]int64\{([^}]*)\}\),\s*Count:\s*(\d+)', f[key], re.S)]
                    if not prior[key]:
                        raise Skip(key)
            if "featureGates" in f:
                g = f["featureGates"]
                gates = dict(top_level_fields_generic(g[g.index("{") + 1:g.rindex("}")]))
                for k, v in gates.items():
                    if k == "features.TASHandleOverlappingFlavors":
                        case["overlappingFlavors"] = v.strip() == "true"
                    elif k == "features.TASProfileMixed":
                        case["profileMixed"] = v.strip() == "true"
                    elif k == "features.TASBalancedPlacement":
                        case["balancedPlacement"] = v.strip() == "true"   # tas_balanced_placement.go: restated by the oracle; the library answers KQ_EUNSUPPORTED
                    elif k == "features.TASRespectNodeAffinityPreferred" and v.strip() == "true":
                        case["affinityPreferred"] = True   # alpha gate: restated by the oracle only (kqo_tas_find_affinity); written to tas_find_affinity.yaml
                    elif k == "features.TASMultiLayerTopology" and v.strip() == "true":
                        pass  # the gate only lets the job parser populate the constraint list; the algorithm honours whatever is there
                    else:
                        raise Skip("gate " + k.replace("features.", ""))
            nodes = f.get("nodes", "").strip()
            if nodes in named_nodes:
                if isinstance(named_nodes[nodes], Skip):
                    raise named_nodes[nodes]
                case["nodes"] = named_nodes[nodes]
            elif nodes:
                case["nodes"] = parse_nodes(nodes[nodes.index("{") + 1:nodes.rindex("}")])
            else:
                case["nodes"] = []
            case["levels"] = parse_strings(f["levels"], named_levels)
            if prior:
                # what the harness hands the snapshot (tas_cache_test.go:8368-8395): with TASHandleOverlappingFlavors (beta, on) and a hostname
                # lowest level the AGGREGATED usage — the sibling flavor's, aggregatedDomainUsagesForPriorFlavorUsage :8935 — replaces the
                # flavor's own (tas_flavor.go:209); otherwise the flavor's own usage
                overlapping = case.get("overlappingFlavors", True) and case["levels"][-1] == "kubernetes.io/hostname"
                case["tasUsage"] = prior.get("priorFlavorUsage", []) if overlapping else prior.get("priorOwnUsage", [])
            if "pods" in f:
                p = f["pods"]
                case["nonTASUsage"] = parse_pods(p[p.index("{") + 1:p.rindex("}")])
            ps_text = f["podSets"]
            case["podSets"] = [parse_podset(e[e.index("{") + 1:e.rindex("}")], named_levels) for e in elements(ps_text[ps_text.index("{") + 1:ps_text.rindex("}")])]
            if not case.get("affinityPreferred") and any("nodeAffinity" in ps_ for ps_ in case["podSets"]):
                raise Skip("nodeAffinity")
            (aff_cases if case.get("affinityPreferred") else cases).append(case)
        except Skip as e:
            skipped[str(e).split(" ")[0] if str(e).startswith(("gate", "node", "pod")) else str(e)] += 1
        except (KeyError, ValueError, AttributeError) as e:
            skipped["parse:" + type(e).__name__] += 1
    hdr = ("# Generated by tests/golden/extract_tas.py from pkg/cache/scheduler/tas_cache_test.go (TestFindTopologyAssignments).\n"
           f"# {len(cases)} cases kept; skipped: {dict(skipped)}\n")
    with open(OUT, "w") as fh:
        fh.write(hdr)
        yaml.safe_dump(dict(cases=cases), fh, sort_keys=False, width=160)
    with open(OUT.replace("tas_find.yaml", "tas_find_affinity.yaml"), "w") as fh:
        fh.write("# Generated by tests/golden/extract_tas.py from pkg/cache/scheduler/tas_cache_test.go (TestFindTopologyAssignments): the rows that run\n"
                 f"# with features.TASRespectNodeAffinityPreferred on ({len(aff_cases)} cases). Oracle only: the library refuses the gate (KQ_TAS_F_AFFINITY_PREFERRED).\n")
        yaml.safe_dump(dict(cases=aff_cases), fh, sort_keys=False, width=160)
    print(len(cases), "cases;", len(aff_cases), "affinity cases;", dict(skipped))


if __name__ == "__main__":
    main()

"""Host-side string matching (kueue_amd/node_match.py): k8s semantics of tolerations (corev1 Toleration.ToleratesTaint), node selector terms
(component-helpers nodeaffinity: terms ORed, expressions ANDed, an empty term matches nothing) and the two places the reference uses them on
the path — checkFlavorForPodSets' taint / affinity half (scheduler/flavorassigner/flavorassigner.go:1243-1260, flavorSelector :1282) and
FindFeasibleNodes (cache/scheduler/scheduling_simulator_default.go:55-118). The whole-cycle rows of TestScheduleForTAS that carry taints
and affinities go through these functions (tests/golden/schedule_tas.yaml); here the operators one by one."""
from kueue_amd import node_match as NM
from kueue_amd.tas import Node

T = NM.Taint
TOL = NM.Toleration
REQ = NM.NodeSelectorRequirement
TERM = NM.NodeSelectorTerm


def test_tolerates_taint_operators():
    t = T("k", "v", "NoSchedule")
    assert NM.tolerates(TOL("k", "Equal", "v", "NoSchedule"), t)
    assert NM.tolerates(TOL("k", "", "v", ""), t)                     # "" operator = Equal, "" effect = every effect
    assert not NM.tolerates(TOL("k", "Equal", "w", ""), t)
    assert not NM.tolerates(TOL("k", "Equal", "v", "NoExecute"), t)   # effect differs
    assert NM.tolerates(TOL("k", "Exists", "", ""), t)
    assert NM.tolerates(TOL("", "Exists", "", ""), t)                 # empty key + Exists tolerates everything
    assert not NM.tolerates(TOL("other", "Exists", "", ""), t)
    assert not NM.tolerates(TOL("k", "Bogus", "v", ""), t)


def test_untolerated_taint_looks_at_scheduling_taints_only():
    taints = [T("soft", "x", "PreferNoSchedule"), T("a", "1", "NoSchedule"), T("b", "2", "NoExecute")]
    assert NM.untolerated_taint(taints, []).key == "a"                # the first one nothing tolerates; PreferNoSchedule never counts
    assert NM.untolerated_taint(taints, [TOL("a", "Exists")]).key == "b"
    assert NM.untolerated_taint(taints, [TOL("a", "Exists"), TOL("b", "Equal", "2", "NoExecute")]) is None
    assert NM.untolerated_taint([T("soft", "x", "PreferNoSchedule")], []) is None


def test_selector_terms():
    labels = {"zone": "a", "gen": "7"}
    assert NM.terms_match([TERM([REQ("zone", "In", ["a", "b"])])], labels)
    assert not NM.terms_match([TERM([REQ("zone", "In", ["c"])])], labels)
    assert NM.terms_match([TERM([REQ("zone", "NotIn", ["c"])])], labels)
    assert NM.terms_match([TERM([REQ("missing", "NotIn", ["c"])])], labels)          # NotIn matches an absent key
    assert NM.terms_match([TERM([REQ("zone", "Exists")])], labels) and not NM.terms_match([TERM([REQ("missing", "Exists")])], labels)
    assert NM.terms_match([TERM([REQ("missing", "DoesNotExist")])], labels)
    assert NM.terms_match([TERM([REQ("gen", "Gt", ["6"])])], labels) and not NM.terms_match([TERM([REQ("gen", "Lt", ["7"])])], labels)
    assert not NM.terms_match([TERM([REQ("zone", "Gt", ["1"])])], labels)             # not an integer
    # expressions of a term are ANDed, terms ORed, a term without expressions matches nothing
    assert not NM.terms_match([TERM([REQ("zone", "In", ["a"]), REQ("gen", "In", ["8"])])], labels)
    assert NM.terms_match([TERM([REQ("zone", "In", ["c"])]), TERM([REQ("gen", "In", ["7"])])], labels)
    assert not NM.terms_match([TERM([])], labels) and not NM.terms_match([], labels)
    assert NM.terms_match([TERM([], [REQ("metadata.name", "In", ["n1"])])], labels, name="n1")


def test_flavor_mismatch_follows_flavor_selector():
    fl = {"type": "spot"}
    # the flavor's NodeTaints against the podset's + the flavor's own tolerations (:1244)
    assert NM.flavor_mismatch({}, None, [], fl, [T("spot", "true", "NoSchedule")], []) is not None
    assert NM.flavor_mismatch({}, None, [TOL("spot", "Exists")], fl, [T("spot", "true", "NoSchedule")], []) is None
    assert NM.flavor_mismatch({}, None, [], fl, [T("spot", "true", "NoSchedule")], [TOL("spot", "Equal", "true", "NoSchedule")]) is None
    # nodeSelector: only the flavor's own label keys count (:1288-1294)
    assert NM.flavor_mismatch({"type": "on-demand"}, None, [], fl, [], []) is not None
    assert NM.flavor_mismatch({"type": "spot", "unrelated": "x"}, None, [], fl, [], []) is None
    # required affinity: expressions on other keys are dropped; a term emptied that way makes the affinity match every flavor (:1307-1311)
    assert NM.flavor_mismatch({}, [TERM([REQ("type", "In", ["on-demand"])])], [], fl, [], []) is not None
    assert NM.flavor_mismatch({}, [TERM([REQ("type", "In", ["on-demand"])]), TERM([REQ("cpu", "In", ["arm"])])], [], fl, [], []) is None
    assert NM.flavor_mismatch({}, [TERM([REQ("type", "In", ["spot"]), REQ("cpu", "In", ["arm"])])], [], fl, [], []) is None
    assert NM.flavor_mismatch({}, [TERM([REQ("type", "NotIn", ["spot"])])], [], fl, [], []) is not None


def test_leaf_mask_is_find_feasible_nodes():
    n = [Node("x1", {"pool": "a", "kubernetes.io/hostname": "x1"}, {"cpu": 1}), Node("x2", {"pool": "b", "kubernetes.io/hostname": "x2"}, {"cpu": 1}),
         Node("x3", {"pool": "a", "kubernetes.io/hostname": "x3"}, {"cpu": 1})]
    taints = {"x2": [T("gpu", "present", "NoSchedule")]}
    assert NM.leaf_mask(n, True, [], {}, None, {}) is None                                   # nothing excluded: no row needed
    assert NM.leaf_mask(n, True, [], {}, None, taints) == [1, 0, 1]
    assert NM.leaf_mask(n, True, [TOL("gpu", "Exists")], {}, None, taints) is None
    assert NM.leaf_mask(n, True, [], {"pool": "a"}, None, {}) == [1, 0, 1]
    assert NM.leaf_mask(n, False, [], {"pool": "a"}, None, {}) is None                       # labels.Everything() above the hostname (:963)
    assert NM.leaf_mask(n, True, [], {}, [TERM([REQ("pool", "In", ["b"])])], {}) == [0, 1, 0]
    assert NM.leaf_mask(n, True, [], {}, [], {}) == [0, 0, 0]                                # a required affinity without terms matches nothing
    assert NM.leaf_mask([None, n[1]], True, [], {}, None, taints) == [1, 0]                  # a leaf without a node object is feasible (:77)

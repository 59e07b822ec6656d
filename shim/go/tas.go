//go:build kq_hip

package kqengine

// tas.go — Topology-Aware Scheduling through the engine.
//
//   - TASCycle / (*Engine).RunCycleTAS bind include/kq_cycle_tas.h: ONE call per (*Scheduler).schedule for a cycle whose ClusterQueues
//     list TAS ResourceFlavors — flavorassigner.Assign's TAS step (flavorassigner.go:864-903), the TAS-aware workloadFits of
//     preemption.GetTargets (preemption.go:669-684), updateAssignmentForTAS (scheduler.go:941-985) and the TAS side of
//     ClusterQueueSnapshot.Fits / AddUsage plus the recomputation inside processEntry (scheduler.go:707-769) all run on the device.
//   - TAS / FindTopologyAssignments / Admit bind include/kq_tas.h: the batch placement (tas_flavor_snapshot.go:578
//     FindTopologyAssignmentsForFlavor for many workloads at once) and the entry-order admission walk, for callers that drive the TAS
//     side themselves (kueue_amd/sharding.py SplitTAS is the Python twin).
//
// What stays in Go, as in the reference: building the TASFlavorSnapshot of every TAS flavor (pkg/cache/scheduler/tas_flavor.go — nodes
// matching the flavor's nodeLabels, free capacity, the domain tree in lexicographic levelValues order), isTASOnly (clusterqueue.go:746),
// checkPodSetAndFlavorMatchForTAS (tas_flavorassigner.go:164, folded into FlatHeads.PsFlavorOK) and resolving every podset's level keys
// against every TAS flavor (tas_flavor_snapshot.go:1197-1238). The flatten itself (FlattenTAS) belongs in pkg/cache/scheduler — the tree and
// the leaf capacities of TASFlavorSnapshot are unexported — and fills TASCycle field by field as INTEGRATION.md tabulates.
// NOT COMPILED HERE (no Go toolchain in the build image), see kqengine.go.

/*
#include <stdlib.h>
#include "kq_engine.h"
#include "kq_tas.h"
#include "kq_cycle_tas.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"
)

// ErrUnsupported = KQ_EUNSUPPORTED from kq_cycle_run_tas: fair sharing together with TAS, or a workload whose podsets land on two TAS
// flavors. The caller runs this cycle on the stock Go path.
var ErrUnsupported = errors.New("kqengine: the cycle is outside the device path (KQ_EUNSUPPORTED)")

// FlatTopology is one TASFlavorSnapshot (include/kq_tas.h kq_tas_topology): domains of every level in lexicographic levelValues
// order (utiltas.DomainID order, tas_flavor_snapshot.go:1770 sorts by it last), leaves last.
type FlatTopology struct {
	NLevels, NResources, PodsResource int32
	BalancedPlacement bool // features.TASBalancedPlacement: preferred requests go through tas_balanced_placement.go
	AffinityPreferred bool // features.TASRespectNodeAffinityPreferred: a gate the library does not implement (-> KQ_EUNSUPPORTED)
	ProfileMixed                      bool    // features.TASProfileMixed
	LevelOff, Parent                  []int32 // Parent: index within the level above, -1 at level 0
	FreeCapacity, TASUsage            []int64 // [leaves][resources]: leafCapacity.freeCapacity :88 / tasUsage
	LeafValues                        [][]string // Go side only: levelValues of every leaf, to turn (leaf, count) back into a TopologyAssignment
}

// TASCycle is the TAS side of one scheduling cycle (kq_cycle_tas): Topos are the TAS flavors in name order
// (slices.Sorted, clusterqueue_snapshot.go:220).
type TASCycle struct {
	NoRecompute bool    // !features.TASRecomputeAssignmentWithinSchedulingCycle
	NoFailFast  bool    // !features.TASFailedNodeReplacementFailFast (kq_cycle_run_tas evicts a second-pass head whose node replacement fails)
	TASFlavor   []int32 // index of every TAS flavor in FlatSnapshot.FlavorNames
	Topos       []FlatTopology
	CQTASOnly   []uint8
	// admitted rows of the FlatSnapshot: workload.TASUsage() as TopologyDomainRequests (CSR over the rows)
	AdmOff, AdmTAS, AdmLeaf, AdmCount []int32
	AdmReq                            []int64
	// one record per podset of FlatHeads (global podset index)
	PsFlags, PsKind               []uint8 // KQ_PS_TAS_EXPLICIT; KQ_TAS_REQUIRED / PREFERRED / UNCONSTRAINED
	PsLevel, PsSliceLevel         []int32 // [podsets][len(Topos)]
	PsSliceSize, PsGroup          []int32
	PsReq                         []int64 // [podsets][resources] SinglePodRequests (tas_flavorassigner.go:116)
	PsNLayers, PsLayerLevel, PsLayerSize []int32 // TASMultiLayerTopology; nil = single layer everywhere. Strides: [podsets][len(Topos)][C.KQ_TAS_MAX_LEVELS] / [podsets][C.KQ_TAS_MAX_LEVELS] (16, the API's MaxItems)
	// The second pass (workload.NeedsSecondPass; heads carry headQuota, and headUnhealthy / headUnhealthyTA in Heads.Flags): what
	// Status.Admission holds for the heads' podsets. All nil = no head holds an admission.
	PsAdmFlavor                  []int32 // [podsets][snapshot resources] PodSetAssignments[i].Flavors as flavor indices, -1 = none
	PsExOff, PsExLeaf, PsExCount []int32 // CSR per podset: PodSetAssignments[i].TopologyAssignment, every domain, leaf index in the podset's TAS flavor (-1 = stale)
	PsExFlags                    []uint8 // ExUnhealthy | ExFirst per domain
	// Node feasibility of a podset on a TAS flavor (taints vs tolerations, PodSpec.NodeSelector, required node affinity:
	// tas_flavor_snapshot.go:955-963), evaluated on the host and shared as rows. nil = every podset may use every leaf.
	PsMask     []int32 // [podsets][len(Topos)] row of LeafMask, -1 = every leaf
	LeafMask   []uint8 // [rows][MaskStride] 1 = the leaf's node may host the podset
	MaskStride int32
	Unsupported bool // FlattenTAS met something the engine does not take (an unparsable selector / affinity): run this cycle in Go
}

const (
	ExUnhealthy = uint8(1) // KQ_EX_UNHEALTHY: the domain's node is one of Status.UnhealthyNodes
	ExFirst     = uint8(2) // KQ_EX_FIRST: ... and it is UnhealthyNodes[0] (deleteDomain tas_flavor_snapshot.go:693)
)

// TASCycleOut receives the TopologyAssignment of every podset that holds one (kq_cycle_tas_out).
type TASCycleOut struct {
	PsTAS, DomOff, DomLeaf, DomCount []int32
	UsageAfter                       []int64 // optional: leaf usage of every TAS flavor after the cycle, concatenated
	Stats                            [4]int64 // placements computed, recomputations inside processEntry, outside-the-path flag, class-table starts
}

func NewTASCycleOut(nPodsets, domCap int, usageCells int) *TASCycleOut {
	o := &TASCycleOut{PsTAS: make([]int32, max(nPodsets, 1)), DomOff: make([]int32, nPodsets+1), DomLeaf: make([]int32, max(domCap, 1)),
		DomCount: make([]int32, max(domCap, 1))}
	if usageCells > 0 {
		o.UsageAfter = make([]int64, usageCells)
	}
	return o
}

func fillTopology(p *runtime.Pinner, c *C.kq_tas_topology, t *FlatTopology) {
	c.n_levels = C.int32_t(t.NLevels)
	c.n_resources = C.int32_t(t.NResources)
	c.pods_resource = C.int32_t(t.PodsResource)
	if t.ProfileMixed {
		c.profile_mixed = C.KQ_TAS_F_PROFILE_MIXED
	}
	// features.TASRespectNodeAffinityPreferred (alpha, default off): a path the library does not have — it answers KQ_EUNSUPPORTED and the
	// scheduler keeps its own FindTopologyAssignmentsForFlavor while it is on. TASBalancedPlacement is implemented (kq_tas.h).
	if t.BalancedPlacement {
		c.profile_mixed |= C.KQ_TAS_F_BALANCED_PLACEMENT
	}
	if t.AffinityPreferred {
		c.profile_mixed |= C.KQ_TAS_F_AFFINITY_PREFERRED
	}
	c.level_off = (*C.int32_t)(pin(p, t.LevelOff))
	c.parent = (*C.int32_t)(pin(p, t.Parent))
	c.free_capacity = (*C.int64_t)(pin(p, t.FreeCapacity))
	c.tas_usage = (*C.int64_t)(pin(p, t.TASUsage))
}

// RunCycleTAS = one (*Scheduler).schedule with TAS inside it (kq_cycle_run_tas): the snapshot is the resident one (PutSnapshot), h the
// cycle's heads, t their TAS side. KQ_EUNSUPPORTED (fair sharing with TAS, a workload on two TAS flavors) means: this cycle takes the
// Go path.
func (e *Engine) RunCycleTAS(h *FlatHeads, t *TASCycle, out *FlatDecisions, tout *TASCycleOut) error {
	var p runtime.Pinner
	defer p.Unpin()
	ch := (*C.kq_heads)(C.calloc(1, C.sizeof_kq_heads))
	defer C.free(unsafe.Pointer(ch))
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	ct := (*C.kq_cycle_tas)(C.calloc(1, C.sizeof_kq_cycle_tas))
	defer C.free(unsafe.Pointer(ct))
	co := (*C.kq_cycle_tas_out)(C.calloc(1, C.sizeof_kq_cycle_tas_out))
	defer C.free(unsafe.Pointer(co))
	fillHeads(&p, ch, h)
	fillDecisions(&p, cd, out)
	nt := len(t.Topos)
	topos := (*C.kq_tas_topology)(C.calloc(C.size_t(max(nt, 1)), C.sizeof_kq_tas_topology))
	defer C.free(unsafe.Pointer(topos))
	ts := unsafe.Slice(topos, max(nt, 1))
	for i := range t.Topos {
		fillTopology(&p, &ts[i], &t.Topos[i])
	}
	if t.NoRecompute {
		ct.flags |= C.KQ_CT_NO_RECOMPUTE
	}
	if t.NoFailFast { // !features.Enabled(features.TASFailedNodeReplacementFailFast)
		ct.flags |= C.KQ_CT_NO_FAIL_FAST
	}
	ct.n_tas = C.int32_t(nt)
	ct.tas_flavor = (*C.int32_t)(pin(&p, t.TASFlavor))
	ct.topo = topos
	ct.cq_tas_only = (*C.uint8_t)(pin(&p, t.CQTASOnly))
	ct.adm_off = (*C.int32_t)(pin(&p, t.AdmOff))
	ct.adm_tas = (*C.int32_t)(pin(&p, t.AdmTAS))
	ct.adm_leaf = (*C.int32_t)(pin(&p, t.AdmLeaf))
	ct.adm_count = (*C.int32_t)(pin(&p, t.AdmCount))
	ct.adm_req = (*C.int64_t)(pin(&p, t.AdmReq))
	ct.ps_flags = (*C.uint8_t)(pin(&p, t.PsFlags))
	ct.ps_kind = (*C.uint8_t)(pin(&p, t.PsKind))
	ct.ps_level = (*C.int32_t)(pin(&p, t.PsLevel))
	ct.ps_slice_size = (*C.int32_t)(pin(&p, t.PsSliceSize))
	ct.ps_slice_level = (*C.int32_t)(pin(&p, t.PsSliceLevel))
	ct.ps_group = (*C.int32_t)(pin(&p, t.PsGroup))
	ct.ps_req = (*C.int64_t)(pin(&p, t.PsReq))
	if len(t.PsNLayers) > 0 {
		ct.ps_n_layers = (*C.int32_t)(pin(&p, t.PsNLayers))
		ct.ps_layer_level = (*C.int32_t)(pin(&p, t.PsLayerLevel))
		ct.ps_layer_size = (*C.int32_t)(pin(&p, t.PsLayerSize))
	}
	if len(t.PsAdmFlavor) > 0 {
		ct.ps_adm_flavor = (*C.int32_t)(pin(&p, t.PsAdmFlavor))
		if len(t.PsExOff) > 0 {
			ct.ps_ex_off = (*C.int32_t)(pin(&p, t.PsExOff))
			ct.ps_ex_leaf = (*C.int32_t)(pin(&p, t.PsExLeaf))
			ct.ps_ex_count = (*C.int32_t)(pin(&p, t.PsExCount))
			ct.ps_ex_flags = (*C.uint8_t)(pin(&p, t.PsExFlags))
		}
	}
	if len(t.PsMask) > 0 && t.MaskStride > 0 {
		ct.ps_mask = (*C.int32_t)(pin(&p, t.PsMask))
		ct.leaf_mask = (*C.uint8_t)(pin(&p, t.LeafMask))
		ct.n_masks = C.int32_t(int32(len(t.LeafMask)) / t.MaskStride)
		ct.mask_stride = C.int32_t(t.MaskStride)
	}
	co.ps_tas = (*C.int32_t)(pin(&p, tout.PsTAS))
	co.dom_off = (*C.int32_t)(pin(&p, tout.DomOff))
	co.dom_leaf = (*C.int32_t)(pin(&p, tout.DomLeaf))
	co.dom_count = (*C.int32_t)(pin(&p, tout.DomCount))
	co.dom_cap = C.int32_t(len(tout.DomLeaf))
	if len(tout.UsageAfter) > 0 {
		co.tas_usage_after = (*C.int64_t)(pin(&p, tout.UsageAfter))
	}
	if rc := C.kq_cycle_run_tas(e.h, ch, ct, cd, co, (*C.int64_t)(unsafe.Pointer(&tout.Stats[0]))); rc != 0 {
		if rc == C.KQ_EUNSUPPORTED {
			return ErrUnsupported
		}
		return e.err("kq_cycle_run_tas", rc)
	}
	return nil
}

// UsageCells = length of TASCycleOut.UsageAfter: leaves x resources of every TAS flavor, concatenated.
func (t *TASCycle) UsageCells() int {
	n := 0
	for i := range t.Topos {
		n += len(t.Topos[i].TASUsage)
	}
	return n
}

// TopologyAssignmentOf turns a podset's (leaf, count) list back into the levels / domains of a kueue.TopologyAssignment
// (tas_flavor_snapshot.go:1701 buildAssignment: leaves ascending = lexicographic levelValues order).
func (o *TASCycleOut) TopologyAssignmentOf(t *TASCycle, podset int) (flavor int32, domains [][]string, counts []int32, ok bool) {
	ti := o.PsTAS[podset]
	if ti < 0 {
		return -1, nil, nil, false
	}
	for k := o.DomOff[podset]; k < o.DomOff[podset+1]; k++ {
		domains = append(domains, t.Topos[ti].LeafValues[o.DomLeaf[k]])
		counts = append(counts, o.DomCount[k])
	}
	return t.TASFlavor[ti], domains, counts, true
}

// ---- include/kq_tas.h: the batch placement and the admission walk ------------------------------------------------------------------

// TAS owns one kq_tas* (one TAS flavor resident on the device).
type TAS struct{ h *C.kq_tas }

func NewTAS(device int) (*TAS, error) {
	t := &TAS{}
	if rc := C.kq_tas_create(C.int32_t(device), &t.h); rc != 0 {
		return nil, fmt.Errorf("kq_tas_create: %s", C.GoString(C.kq_strerror(rc)))
	}
	return t, nil
}
func (t *TAS) Close() { C.kq_tas_destroy(t.h); t.h = nil }
func (t *TAS) err(what string, rc C.int) error {
	return fmt.Errorf("%s: %s (%s)", what, C.GoString(C.kq_strerror(rc)), C.GoString(C.kq_tas_last_error(t.h)))
}

// PutTopology makes the flavor's domain tree, free capacity and TAS usage resident (kq_tas_topology_put).
func (t *TAS) PutTopology(tp *FlatTopology) error {
	var p runtime.Pinner
	defer p.Unpin()
	c := (*C.kq_tas_topology)(C.calloc(1, C.sizeof_kq_tas_topology))
	defer C.free(unsafe.Pointer(c))
	fillTopology(&p, c, tp)
	if rc := C.kq_tas_topology_put(t.h, c); rc != 0 {
		return t.err("kq_tas_topology_put", rc)
	}
	return nil
}

// TASRequests = []TASPodSetRequests of a batch of workloads (kq_tas_requests), one record per podset.
type TASRequests struct {
	WlOff                               []int32
	SimulateEmpty                       []uint8 // per workload, nil = none
	SinglePodRequests                   []int64 // [n][resources]
	Count, Level, SliceSize, SliceLevel []int32
	Kind                                []uint8
	Group                               []int32
	LeafOK                              []uint8 // [n][leaves] or nil
	NLayers, LayerLevel, LayerSize      []int32 // nil = single layer; [podsets][C.KQ_TAS_MAX_LEVELS]
}

// TASResult = the TopologyAssignment (or the failure operands, KQ_TAS_*) per podset request.
type TASResult struct {
	Status, OperandA, OperandB, DomOff, DomLeaf, DomCount []int32
}

func fillRequests(p *runtime.Pinner, c *C.kq_tas_requests, r *TASRequests) {
	c.n_workloads = C.int32_t(len(r.WlOff) - 1)
	c.wl_off = (*C.int32_t)(pin(p, r.WlOff))
	if len(r.SimulateEmpty) > 0 {
		c.simulate_empty = (*C.uint8_t)(pin(p, r.SimulateEmpty))
	}
	c.single_pod_requests = (*C.int64_t)(pin(p, r.SinglePodRequests))
	c.count = (*C.int32_t)(pin(p, r.Count))
	c.level = (*C.int32_t)(pin(p, r.Level))
	c.kind = (*C.uint8_t)(pin(p, r.Kind))
	c.slice_size = (*C.int32_t)(pin(p, r.SliceSize))
	c.slice_level = (*C.int32_t)(pin(p, r.SliceLevel))
	c.group = (*C.int32_t)(pin(p, r.Group))
	if len(r.LeafOK) > 0 {
		c.leaf_ok = (*C.uint8_t)(pin(p, r.LeafOK))
	}
	if len(r.NLayers) > 0 {
		c.n_layers = (*C.int32_t)(pin(p, r.NLayers))
		c.layer_level = (*C.int32_t)(pin(p, r.LayerLevel))
		c.layer_size = (*C.int32_t)(pin(p, r.LayerSize))
	}
}
func fillResult(p *runtime.Pinner, c *C.kq_tas_result, r *TASResult) {
	c.status = (*C.int32_t)(pin(p, r.Status))
	c.operand_a = (*C.int32_t)(pin(p, r.OperandA))
	c.operand_b = (*C.int32_t)(pin(p, r.OperandB))
	c.dom_off = (*C.int32_t)(pin(p, r.DomOff))
	c.dom_leaf = (*C.int32_t)(pin(p, r.DomLeaf))
	c.dom_count = (*C.int32_t)(pin(p, r.DomCount))
	c.dom_cap = C.int32_t(len(r.DomLeaf))
}

// FindTopologyAssignments = FindTopologyAssignmentsForFlavor (tas_flavor_snapshot.go:578) for every workload of the batch.
func (t *TAS) FindTopologyAssignments(r *TASRequests, out *TASResult) error {
	var p runtime.Pinner
	defer p.Unpin()
	cr := (*C.kq_tas_requests)(C.calloc(1, C.sizeof_kq_tas_requests))
	defer C.free(unsafe.Pointer(cr))
	co := (*C.kq_tas_result)(C.calloc(1, C.sizeof_kq_tas_result))
	defer C.free(unsafe.Pointer(co))
	fillRequests(&p, cr, r)
	fillResult(&p, co, out)
	if rc := C.kq_tas_find(t.h, cr, co); rc != 0 {
		return t.err("kq_tas_find", rc)
	}
	return nil
}

// Admit = the TAS side of processEntry over the batch in entry order (kq_tas_admit): Fits, then AddUsage (clusterqueue_snapshot.go:107-149).
func (t *TAS) Admit(r *TASRequests, res *TASResult, order []int32, admitted []uint8) (int32, error) {
	var p runtime.Pinner
	defer p.Unpin()
	cr := (*C.kq_tas_requests)(C.calloc(1, C.sizeof_kq_tas_requests))
	defer C.free(unsafe.Pointer(cr))
	co := (*C.kq_tas_result)(C.calloc(1, C.sizeof_kq_tas_result))
	defer C.free(unsafe.Pointer(co))
	fillRequests(&p, cr, r)
	fillResult(&p, co, res)
	var n C.int32_t
	var op *C.int32_t
	if len(order) > 0 {
		op = (*C.int32_t)(pin(&p, order))
	}
	if rc := C.kq_tas_admit(t.h, cr, co, op, C.int32_t(len(order)), (*C.uint8_t)(pin(&p, admitted)), &n); rc != 0 {
		return 0, t.err("kq_tas_admit", rc)
	}
	return int32(n), nil
}

// UsageApply = TASFlavorSnapshot.updateTASUsage :267 / its removal for one TopologyDomainRequests list; Fits = :433.
func (t *TAS) UsageApply(leaf, count []int32, singlePodRequests []int64, add bool) error {
	var p runtime.Pinner
	defer p.Unpin()
	a := C.int32_t(0)
	if add {
		a = 1
	}
	if rc := C.kq_tas_usage_apply(t.h, C.int32_t(len(leaf)), (*C.int32_t)(pin(&p, leaf)), (*C.int32_t)(pin(&p, count)), (*C.int64_t)(pin(&p, singlePodRequests)), a); rc != 0 {
		return t.err("kq_tas_usage_apply", rc)
	}
	return nil
}
func (t *TAS) Fits(leaf, count []int32, singlePodRequests []int64) (bool, error) {
	var p runtime.Pinner
	defer p.Unpin()
	var f C.int32_t
	if rc := C.kq_tas_fits(t.h, C.int32_t(len(leaf)), (*C.int32_t)(pin(&p, leaf)), (*C.int32_t)(pin(&p, count)), (*C.int64_t)(pin(&p, singlePodRequests)), &f); rc != 0 {
		return false, t.err("kq_tas_fits", rc)
	}
	return f != 0, nil
}
func (t *TAS) ReadUsage(out []int64) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_tas_read_usage(t.h, (*C.int64_t)(pin(&p, out))); rc != 0 {
		return t.err("kq_tas_read_usage", rc)
	}
	return nil
}

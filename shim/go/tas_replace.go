//go:build kq_hip

package kqengine

// tas_replace.go — node replacement and the exclusion statistics of notFitMessage through include/kq_tas.h.
//
// FindTopologyAssignmentsForFlavor takes the HasUnhealthyNodes branch (tas_flavor_snapshot.go:608-633) for a workload whose
// Status.UnhealthyNodes is set: findReplacementAssignment :686 for every podset that holds a TopologyAssignment. The split:
//   - here (string work): findPSA :747, SkipReassignmentForPodOwnedWorkloads :615, deleteDomain :828 and the resolution of the remaining
//     domains to leaf indices (a domain that is no leaf of the snapshot = IsTopologyAssignmentStale :818) — NewReplacement;
//   - in the library (kq_tas_find_replacement): the stale verdict, requiredReplacementDomain :759, findIncompleteSliceDomain :842, the
//     rewrite of the slice request :703-722, the placement below the required domain (:1902) and mergeTopologyAssignments :2072;
//   - kq_tas_exclusion_stats: tasExclusionStats :470 for the podsets that did not fit; ExclusionTail words them as notFitMessage :1997.
// NOT COMPILED HERE (no Go toolchain in the build image), see kqengine.go.

/*
#include <stdlib.h>
#include "kq_engine.h"
#include "kq_tas.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"sort"
	"strings"
	"unsafe"
)

// DomainCount is one utiltas.TopologyDomainAssignment of an existing TopologyAssignment.
type DomainCount struct {
	Values []string
	Count  int32
}

// TASReplacement is kq_tas_replacement for a batch: per podset request of TASRequests, the existing assignment after deleteDomain.
type TASReplacement struct {
	IsReplacement          []uint8
	ExOff, ExLeaf, ExCount []int32
	// Go side only: the Values of every remaining domain (for the stale message) and the unhealthy node per podset request
	ExValues  [][]string
	Unhealthy []string
}

// NewReplacement starts an empty batch-side record; AddPodSet appends one podset request in the order of TASRequests.
func NewReplacement() *TASReplacement { return &TASReplacement{ExOff: []int32{0}} }

// AddPodSet appends a podset request. existing == nil: an ordinary request. Otherwise deleteDomain :828 — the domain whose last value
// is the unhealthy node leaves the list — and the returned count is tr.Count of findReplacementAssignment :693 (the pods it held): the
// caller writes it into TASRequests.Count for this podset. leafOf maps the Values of a domain to its leaf index, -1 when the snapshot
// has no such leaf (FlatTopology.LeafValues inverted; a hostname alone on a hostname-level topology).
func (x *TASReplacement) AddPodSet(existing []DomainCount, unhealthyNode string, leafOf func([]string) int32) (affected int32) {
	if existing == nil {
		x.IsReplacement = append(x.IsReplacement, 0)
		x.ExOff = append(x.ExOff, int32(len(x.ExLeaf)))
		x.Unhealthy = append(x.Unhealthy, "")
		return 0
	}
	for _, d := range existing {
		if d.Values[len(d.Values)-1] == unhealthyNode {
			affected = d.Count
			continue
		}
		x.ExLeaf = append(x.ExLeaf, leafOf(d.Values))
		x.ExCount = append(x.ExCount, d.Count)
		x.ExValues = append(x.ExValues, d.Values)
	}
	x.IsReplacement = append(x.IsReplacement, 1)
	x.ExOff = append(x.ExOff, int32(len(x.ExLeaf)))
	x.Unhealthy = append(x.Unhealthy, unhealthyNode)
	return affected
}

func fillReplacement(p *runtime.Pinner, c *C.kq_tas_replacement, x *TASReplacement) {
	c.is_replacement = (*C.uint8_t)(pin(p, x.IsReplacement))
	c.ex_off = (*C.int32_t)(pin(p, x.ExOff))
	if len(x.ExLeaf) > 0 {
		c.ex_leaf = (*C.int32_t)(pin(p, x.ExLeaf))
		c.ex_count = (*C.int32_t)(pin(p, x.ExCount))
	}
}

// FindReplacementAssignments = FindTopologyAssignmentsForFlavor for a batch that holds workloads with unhealthy nodes
// (kq_tas_find_replacement). A replacement podset comes back with the MERGED assignment (mergeTopologyAssignments :2072) or with
// KQ_TAS_STALE / KQ_TAS_NO_REPLACEMENT / the placement's own failure; ReplacementMessage words the first two.
func (t *TAS) FindReplacementAssignments(r *TASRequests, x *TASReplacement, out *TASResult) error {
	var p runtime.Pinner
	defer p.Unpin()
	cr := (*C.kq_tas_requests)(C.calloc(1, C.sizeof_kq_tas_requests))
	defer C.free(unsafe.Pointer(cr))
	cx := (*C.kq_tas_replacement)(C.calloc(1, C.sizeof_kq_tas_replacement))
	defer C.free(unsafe.Pointer(cx))
	co := (*C.kq_tas_result)(C.calloc(1, C.sizeof_kq_tas_result))
	defer C.free(unsafe.Pointer(co))
	fillRequests(&p, cr, r)
	fillReplacement(&p, cx, x)
	fillResult(&p, co, out)
	if rc := C.kq_tas_find_replacement(t.h, cr, cx, co); rc != 0 {
		return t.err("kq_tas_find_replacement", rc)
	}
	return nil
}

// FindElasticAssignments = FindTopologyAssignmentsForFlavor with features.ElasticJobsViaWorkloadSlicesWithTAS on (kq_tas_find_elastic): `prev`
// marks the podsets that carry TASPodSetRequests.PreviousAssignment (IsReplacement[i] = 1; Ex* = its domains in the assignment's order,
// leaf -1 for a domain the snapshot no longer holds) — handleElasticWorkload (tas_elastic_workloads.go:37): scale-up places the delta
// only and merges, scale-down truncates, the same count reuses. KQ_EUNSUPPORTED (an elastic workload of several podset groups): the
// caller keeps s.handleElasticWorkload for that workload.
func (t *TAS) FindElasticAssignments(r *TASRequests, prev *TASReplacement, out *TASResult) error {
	var p runtime.Pinner
	defer p.Unpin()
	cr := (*C.kq_tas_requests)(C.calloc(1, C.sizeof_kq_tas_requests))
	defer C.free(unsafe.Pointer(cr))
	cx := (*C.kq_tas_replacement)(C.calloc(1, C.sizeof_kq_tas_replacement))
	defer C.free(unsafe.Pointer(cx))
	co := (*C.kq_tas_result)(C.calloc(1, C.sizeof_kq_tas_result))
	defer C.free(unsafe.Pointer(co))
	fillRequests(&p, cr, r)
	fillReplacement(&p, cx, prev)
	fillResult(&p, co, out)
	if rc := C.kq_tas_find_elastic(t.h, cr, cx, co); rc != 0 {
		return t.err("kq_tas_find_elastic", rc)
	}
	return nil
}

// ReplacementMessage is the failure reason of findReplacementAssignment for the two statuses the library adds (:696, :728).
func (x *TASReplacement) ReplacementMessage(podset int, status, operandA int32) string {
	switch status {
	case C.KQ_TAS_STALE:
		return fmt.Sprintf("Cannot replace the node, because the existing topologyAssignment is invalid, as it contains the stale domain %v",
			x.ExValues[int(x.ExOff[podset])+int(operandA)][0])
	case C.KQ_TAS_NO_REPLACEMENT:
		return fmt.Sprintf("cannot find replacement assignment for unhealthy node: %v", x.Unhealthy[podset])
	}
	return ""
}

// TASExclusions is tasExclusionStats :470 of one podset: TopologyDomain and Resources from the library, the rest from the simulator
// (FindFeasibleNodes fills NodeExclusionStats on the Go side as before).
type TASExclusions struct {
	TotalNodes, NodeSelector, Affinity, TopologyDomain, SchedulerLibraryNoFit int
	Taints                                                                    map[string]int
	Resources                                                                 map[string]int
}

// ExclusionStats fills TopologyDomain / Resources for the given podsets of an answered batch (kq_tas_exclusion_stats). resourceNames
// is the topology's resource dictionary; their alphabetical rank is the tie-break of CountInWithLimitingResource (requests.go:195).
func (t *TAS) ExclusionStats(r *TASRequests, x *TASReplacement, res *TASResult, podsets []int32, resourceNames []string, out []TASExclusions) error {
	if len(podsets) == 0 {
		return nil
	}
	var p runtime.Pinner
	defer p.Unpin()
	cr := (*C.kq_tas_requests)(C.calloc(1, C.sizeof_kq_tas_requests))
	defer C.free(unsafe.Pointer(cr))
	co := (*C.kq_tas_result)(C.calloc(1, C.sizeof_kq_tas_result))
	defer C.free(unsafe.Pointer(co))
	fillRequests(&p, cr, r)
	fillResult(&p, co, res)
	var cx *C.kq_tas_replacement
	if x != nil {
		cx = (*C.kq_tas_replacement)(C.calloc(1, C.sizeof_kq_tas_replacement))
		defer C.free(unsafe.Pointer(cx))
		fillReplacement(&p, cx, x)
	}
	R := len(resourceNames)
	order := make([]int, R)
	for i := range order {
		order[i] = i
	}
	sort.Slice(order, func(a, b int) bool { return resourceNames[order[a]] < resourceNames[order[b]] })
	rank := make([]int32, R)
	for k, i := range order {
		rank[i] = int32(k)
	}
	td := make([]int32, len(podsets))
	rs := make([]int32, len(podsets)*R)
	if rc := C.kq_tas_exclusion_stats(t.h, cr, cx, co, C.int32_t(len(podsets)), (*C.int32_t)(pin(&p, podsets)), (*C.int32_t)(pin(&p, rank)),
		(*C.int32_t)(pin(&p, td)), (*C.int32_t)(pin(&p, rs))); rc != 0 {
		return t.err("kq_tas_exclusion_stats", rc)
	}
	for k := range podsets {
		out[k].TopologyDomain += int(td[k])
		for i := 0; i < R; i++ {
			if n := rs[k*R+i]; n > 0 {
				if out[k].Resources == nil {
					out[k].Resources = map[string]int{}
				}
				out[k].Resources[resourceNames[i]] += int(n)
			}
		}
	}
	return nil
}

// ExclusionTail is what notFitMessage :1997 appends: ". Total nodes: N; excluded: ..." (formatReasons :500), "" without exclusions.
// The gate is hasExclusions (:496), which does not look at SchedulerLibraryNoFit: a podset whose only exclusion is that count gets no tail.
func (s *TASExclusions) ExclusionTail() string {
	if !(s.NodeSelector > 0 || s.Affinity > 0 || len(s.Taints) > 0 || s.TopologyDomain > 0 || len(s.Resources) > 0) {
		return ""
	}
	var reasons []string
	if s.NodeSelector > 0 {
		reasons = append(reasons, fmt.Sprintf("nodeSelector: %d", s.NodeSelector))
	}
	if s.Affinity > 0 {
		reasons = append(reasons, fmt.Sprintf("affinity: %d", s.Affinity))
	}
	if s.TopologyDomain > 0 {
		reasons = append(reasons, fmt.Sprintf("topologyDomain: %d", s.TopologyDomain))
	}
	if s.SchedulerLibraryNoFit > 0 {
		reasons = append(reasons, fmt.Sprintf("schedulerLibraryNoFit: %d", s.SchedulerLibraryNoFit))
	}
	for k, v := range s.Taints {
		reasons = append(reasons, fmt.Sprintf("taint %q: %d", k, v))
	}
	for k, v := range s.Resources {
		reasons = append(reasons, fmt.Sprintf("resource %q: %d", k, v))
	}
	sort.Strings(reasons)
	return fmt.Sprintf(". Total nodes: %d; excluded: %s", s.TotalNodes, strings.Join(reasons, ", "))
}

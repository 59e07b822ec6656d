//go:build kq_hip

package kqengine

// apply.go — kq_decisions back into what schedule() (pkg/scheduler/scheduler.go:356-377) needs for its side effects.
// The scheduler's `entry` type is unexported, so this file lives in package scheduler in a fork (INTEGRATION.md); here it is
// written against an interface with the same fields. Not compiled in this image (no Go toolchain).

import (
	corev1 "k8s.io/api/core/v1"

	kueue "sigs.k8s.io/kueue/apis/kueue/v1beta2"
	"sigs.k8s.io/kueue/pkg/resources"
)

// decision codes of include/kq_engine.h
const (
	stNotNominated, stNominated, stSkipped, stAssumed = 0, 1, 2, 5
	actNone, actAdmit, actPreempt, actEvict           = 0, 1, 2, 3
	stEvicted                                         = 4 // KQ_ST_EVICTED (kq_cycle_run_tas)
	modeNoFit, modePreempt, modeDeferredFit, modeFit  = 0, 1, 2, 3
	skipNone, skipOverlap, skipNoLongerFits           = 0, 1, 2
)

var requeueReasons = map[uint8]string{0: "", 1: "FailedAfterNomination", 4: "PendingPreemption", 7: "NoFit", 8: "PreemptionNoCandidates"}

var targetReasons = [...]string{kueue.InClusterQueueReason, kueue.InCohortReclamationReason, kueue.InCohortFairSharingReason,
	kueue.InCohortReclaimWhileBorrowingReason,
	kueue.WorkloadSliceReplaced} // KQ_REASON_REPLACED_SLICE: the old slice of an elastic workload (preemption.go:141-149); the preemptor
// evicts it with workloadslicing's "Replaced to accommodate a new workload slice" instead of the preemption message

// FlavorChoice is flavorassigner.FlavorAssignment as the engine reports it.
type FlavorChoice struct {
	Flavor         kueue.ResourceFlavorReference
	Mode           uint8 // flavorassigner.FlavorAssignmentMode
	TriedFlavorIdx int
}

// Outcome is everything processEntry would have left on one entry.
type Outcome struct {
	Head            int
	Order           int // position in the entry iterator: side effects are issued in this order (scheduler.go:358)
	Status          uint8
	Admit, Preempt  bool
	Evict           bool // KQ_ACT_EVICT: the second pass found no replacement for the failed node (TASFailedNodeReplacementFailFast): the caller runs evictWorkloadAfterFailedTASReplacement (scheduler.go:926) and markEvicted
	RepMode         uint8
	Borrowing       int
	PodSets         []map[corev1.ResourceName]FlavorChoice // Assignment.PodSets[i].Flavors
	Counts          []int32                                // PodSetAssignment.Count (partial admission)
	LastTried       []map[corev1.ResourceName]int          // next LastAssignment.LastTriedFlavorIdx; nil = cleared
	Targets         []string                               // workload.Reference of every preemption target
	TargetReasons   []string
	RequeueReason   string
	InadmissibleMsg string
}

// Outcomes decodes the decisions of one cycle, ordered by iterator position. ErrReasonsTruncated: a head's reason window
// overflowed (KQ_RSN_TRUNCATED) — run the cycle again with a larger rsn_cap or take the stock Go path for this cycle.
func Outcomes(f *resources.ResourceFormatter, s *FlatSnapshot, h *FlatHeads, d *FlatDecisions, podsetNames func(head int) []string,
	inel IneligibleText, preserveScanProgress bool) ([]Outcome, error) {
	n, nR := int(h.N), int(s.NResource)
	out := make([]Outcome, n)
	for i := 0; i < n; i++ {
		o := Outcome{Head: i, Order: int(d.Order[i]), Status: d.Status[i], Admit: d.Action[i] == actAdmit, Preempt: d.Action[i] == actPreempt, Evict: d.Action[i] == actEvict,
			RepMode: d.Mode[i], Borrowing: int(d.Borrowing[i]), RequeueReason: requeueReasons[d.RequeueReason[i]]}
		// LastAssignment for the next cycle: recordAssignment :281, cleared by markPreemptionOutcome :291, DeferredFit :459-464,
		// markSkipped without FlavorFungibilityPreserveScanProgress :248-254
		clearLast := o.Preempt || d.Mode[i] == modeDeferredFit || (d.Status[i] == stSkipped && !preserveScanProgress)
		for p := h.PsOff[i]; p < h.PsOff[i+1]; p++ {
			fl, lt := map[corev1.ResourceName]FlavorChoice{}, map[corev1.ResourceName]int{}
			for r := 0; r < nR; r++ {
				k := int(p)*nR + r
				if d.Flavor[k] < 0 {
					continue
				}
				name := corev1.ResourceName(s.ResourceNames[r])
				fl[name] = FlavorChoice{kueue.ResourceFlavorReference(s.FlavorNames[d.Flavor[k]]), d.ResMode[k], int(d.TriedIdx[k])}
				lt[name] = int(d.TriedIdx[k])
			}
			o.PodSets = append(o.PodSets, fl)
			o.Counts = append(o.Counts, d.PsCount[p])
			if !clearLast {
				o.LastTried = append(o.LastTried, lt)
			}
		}
		for t := d.TgtOff[i]; t < d.TgtOff[i+1]; t++ {
			o.Targets = append(o.Targets, s.AdmKeys[d.TgtAdm[t]])
			o.TargetReasons = append(o.TargetReasons, targetReasons[d.TgtReason[t]])
		}
		// entry.inadmissibleMsg (scheduler.go:281-295, 248-253, 452-481)
		switch {
		case d.Skip[i] == skipOverlap:
			o.InadmissibleMsg = "Workload has overlapping preemption targets with another workload"
		case d.Skip[i] == skipNoLongerFits:
			o.InadmissibleMsg = "Workload no longer fits after processing another workload"
		case d.Mode[i] == modeDeferredFit:
			o.InadmissibleMsg = "Workload has overlapping preemption targets with another workload, but will fit after these preemptions complete"
		case !o.Admit:
			reasons, err := PodSetReasons(f, s, h, d, i, inel)
			if err != nil {
				return nil, err
			}
			o.InadmissibleMsg = AssignmentMessage(podsetNames(i), reasons)
		}
		out[i] = o
	}
	// iterator order; entries the fair-sharing iterator never popped (order -1) keep their nomination and go last
	sorted := make([]Outcome, 0, n)
	for pos := 0; pos < n; pos++ {
		for i := range out {
			if out[i].Order == pos {
				sorted = append(sorted, out[i])
			}
		}
	}
	for i := range out {
		if out[i].Order < 0 {
			sorted = append(sorted, out[i])
		}
	}
	return sorted, nil
}

//go:build kq_hip

package kqengine

// no_fit_reason.go — Assignment.NoFitReason and the NoFitReason of the FlavorAssignmentAttempts (features.UnadmittedWorkloadsObservability)
// regenerated from one head's reason records and decisions. Twin of kueue_amd/no_fit_reason.py, which this repository's tests run against
// the reference's TestIsNoFitDueToCapacityAndLimits table (flavorassigner_test.go:5308) and against the oracle's Assign on random cycles.
//
// The records name the labelled events per (podset, flavor, resource): KQ_RSN_FLAVOR_INELIGIBLE / KQ_RSN_SLICE_FLAVOR_MISMATCH ->
// NoMatchingFlavor (flavorassigner.go:1106, :1136), KQ_RSN_EXCEEDS_MAX_CAPACITY -> ExceedsMaxQuota (:1361), KQ_RSN_INSUFFICIENT_UNUSED ->
// WaitingForQuota only if the attempt did not reach the preemption oracle: the branch condition of :1375 is recomputed here from the quota
// tree (val = the record's "more needed" + Available). A podset that kept its flavors with every mode NoFit failed the simulate-empty TAS
// pass (:893-899): TopologyPlacementFailed on its TAS flavor. Limits: see the Python twin's header.

import (
	"errors"
	"math"

	kueue "sigs.k8s.io/kueue/apis/kueue/v1beta2"
)

// severity ranks of reasonSeverity (flavorassigner.go:306-327); mostSevereReason is max().
const (
	lblNone = iota
	lblTopologyPlacementFailed
	lblWaitingForQuota
	lblExceedsMaxQuota
	lblNoMatchingFlavor
)

var noFitLabels = [...]string{"", kueue.WorkloadQuotaReservedReasonTopologyPlacementFailed, kueue.WorkloadQuotaReservedReasonWaitingForQuota,
	kueue.WorkloadQuotaReservedReasonExceedsMaxQuota, kueue.WorkloadQuotaReservedReasonNoMatchingFlavor}

const (
	modeNoFit   = 0 // KQ_MODE_*
	modePreempt = 1
	modeFit     = 3
	nilLimit    = -1 // KQ_NIL_LIMIT
	unlimited   = math.MaxInt64
)

func satAdd(x, y int64) int64 { // resources.Amount.Add amount.go:114-128
	if x == unlimited || y == unlimited {
		return unlimited
	}
	s := x + y
	if y > 0 && s < x {
		return math.MaxInt64
	}
	if y < 0 && s > x {
		return math.MinInt64
	}
	return s
}

func satSub(x, y int64) int64 { // Amount.Sub :130-145
	switch {
	case x == unlimited && y == unlimited:
		return 0
	case x == unlimited:
		return unlimited
	case y == unlimited:
		return math.MinInt64
	}
	d := x - y
	if y < 0 && d < x {
		return math.MaxInt64
	}
	if y > 0 && d > x {
		return math.MinInt64
	}
	return d
}

// quotaTree is resource_node.go over the flattened (derived) snapshot.
type quotaTree struct{ s *FlatSnapshot }

func (t quotaTree) k(n, fr int32) int { return int(n)*int(t.s.NFlavor*t.s.NResource) + int(fr) }

func (t quotaTree) localQuota(n, fr int32) int64 { // :67-72
	k := t.k(n, fr)
	if t.s.LendLimit[k] != nilLimit {
		return max(0, satSub(t.s.SubtreeQuota[k], t.s.LendLimit[k]))
	}
	return 0
}

func (t quotaTree) localAvailable(n, fr int32) int64 { // :92-95
	return max(0, satSub(t.localQuota(n, fr), t.s.Usage[t.k(n, fr)]))
}

func (t quotaTree) available(n, fr int32) int64 { // :106-122
	k := t.k(n, fr)
	if t.s.Parent[n] < 0 {
		return satSub(t.s.SubtreeQuota[k], t.s.Usage[k])
	}
	pa := t.available(t.s.Parent[n], fr)
	if t.s.BorrowLimit[k] != nilLimit {
		lq := t.localQuota(n, fr)
		stored := satSub(t.s.SubtreeQuota[k], lq)
		used := max(0, satSub(t.s.Usage[k], lq))
		pa = min(satAdd(satSub(stored, used), t.s.BorrowLimit[k]), pa)
	}
	return satAdd(t.localAvailable(n, fr), pa)
}

func (t quotaTree) borrowingWith(n, fr int32, val int64) bool { // clusterqueue_snapshot.go:155-161 / cohort_snapshot.go:90
	k := t.k(n, fr)
	quota := t.s.SubtreeQuota[k]
	if n < t.s.NCQ {
		quota = t.s.Nominal[k]
	}
	return quota < satAdd(t.s.Usage[k], val)
}

// mayReclaimInHierarchy is the second result of FindHeightOfLowestSubtreeThatFits (classical/hierarchical_preemption.go:221-234).
func (t quotaTree) mayReclaimInHierarchy(cq, fr int32, val int64) bool {
	hasParent := t.s.Parent[cq] >= 0
	if !t.borrowingWith(cq, fr, val) || !hasParent {
		return hasParent
	}
	remaining := satSub(val, t.localAvailable(cq, fr))
	for n := t.s.Parent[cq]; n >= 0; n = t.s.Parent[n] {
		if !t.borrowingWith(n, fr, remaining) {
			return t.s.Parent[n] >= 0
		}
		remaining = satSub(remaining, t.localAvailable(n, fr))
	}
	return false
}

func canPreemptWhileBorrowing(policy uint32, fairSharing bool) bool { // flavorassigner.go:1386-1389; KQ_POL_* bit layout of kq_engine.h
	borrowWithin := (policy >> 4) & 1
	reclaim, reclaimUnset := (policy>>2)&3, (policy>>11)&1
	return borrowWithin != 0 || (fairSharing && (reclaim != 0 || reclaimUnset != 0))
}

// FlavorAttempt is what resolveNoFitReason reads of a FlavorAssignmentAttempt.
type FlavorAttempt struct {
	Mode        uint8 // KQ_MODE_*
	NoFitReason string
}

// NoFitReason returns Assignment.NoFitReason of head i and, per podset up to the first one that got no flavor, the attempts the records
// describe, keyed by flavor name. isTASFlavor may be nil (every flavor counts as a TAS flavor for the TopologyPlacementFailed mark).
func NoFitReason(s *FlatSnapshot, h *FlatHeads, d *FlatDecisions, i int, fairSharing bool, isTASFlavor func(flavor int32) bool) (string, []map[string]FlavorAttempt, error) {
	t := quotaTree{s}
	nR := s.NResource
	cq := h.CQ[i]
	p0, p1 := h.PsOff[i], h.PsOff[i+1]
	type rec struct {
		code   uint8
		fl, rs int32
		more   int64
	}
	recs := make([][]rec, p1-p0)
	for k := d.RsnOff[i]; k < d.RsnOff[i+1]; k++ {
		if d.RsnCode[k] == RsnTruncated {
			return "", nil, errors.New("reason window of the head overflowed: raise rsnCap")
		}
		ps := d.RsnPodset[k]
		recs[ps] = append(recs[ps], rec{d.RsnCode[k], int32(d.RsnFlavor[k]), int32(d.RsnResource[k]), d.RsnA[k]})
	}
	type att struct {
		mode  uint8
		label int
	}
	var out []map[int32]att
	var modes []uint8
	for lp := int32(0); lp < p1-p0; lp++ {
		g := p0 + lp
		a := map[int32]att{}
		mark := func(fl int32, mode uint8, label int) {
			e, ok := a[fl]
			if !ok {
				e = att{modeFit, lblNone}
			}
			a[fl] = att{min(e.mode, mode), max(e.label, label)}
		}
		for _, r := range recs[lp] {
			switch r.code {
			case RsnFlavorIneligible, RsnSliceFlavorMismatch:
				mark(r.fl, modeNoFit, lblNoMatchingFlavor)
			case RsnExceedsMaxCapacity:
				mark(r.fl, modeNoFit, lblExceedsMaxQuota)
			case RsnInsufficientUnused:
				fr := r.fl*nR + r.rs
				val := satAdd(r.more, max(0, t.available(cq, fr)))
				if s.Nominal[t.k(cq, fr)] >= val || t.mayReclaimInHierarchy(cq, fr, val) || canPreemptWhileBorrowing(s.CQPolicy[cq], fairSharing) {
					mark(r.fl, modePreempt, lblNone)
				} else {
					mark(r.fl, modeNoFit, lblWaitingForQuota)
				}
			}
		}
		// PodSetAssignment.RepresentativeMode :386-404
		mode, nfl := uint8(modeFit), 0
		for r := int32(0); r < nR; r++ {
			if d.Flavor[g*nR+r] >= 0 {
				nfl++
				mode = min(mode, d.ResMode[g*nR+r])
			}
		}
		switch {
		case len(recs[lp]) == 0:
			mode = modeFit
		case nfl == 0:
			mode = modeNoFit
		case mode == modeNoFit: // the simulate-empty TAS pass failed on this podset: markFlavorAttempt :413-421
			for r := int32(0); r < nR; r++ {
				if fl := d.Flavor[g*nR+r]; fl >= 0 && (isTASFlavor == nil || isTASFlavor(fl)) {
					a[fl] = att{modeNoFit, lblTopologyPlacementFailed}
					break
				}
			}
		}
		out = append(out, a)
		modes = append(modes, mode)
		if len(recs[lp]) > 0 && nfl == 0 {
			break // assignFlavors returns at the first podset that got no flavor (:848-853)
		}
	}
	named := make([]map[string]FlavorAttempt, len(out))
	for p, a := range out {
		named[p] = make(map[string]FlavorAttempt, len(a))
		for fl, e := range a {
			named[p][s.FlavorNames[fl]] = FlavorAttempt{e.mode, noFitLabels[e.label]}
		}
	}
	noFit := false
	for _, m := range modes {
		noFit = noFit || m == modeNoFit
	}
	if !noFit { // resolveNoFitReason :948
		return "", named, nil
	}
	rgOf := map[int32][]int32{} // findRGIndicesByFlavor
	for rg := s.CQRgOff[cq]; rg < s.CQRgOff[cq+1]; rg++ {
		for k := s.RgFlavorOff[rg]; k < s.RgFlavorOff[rg+1]; k++ {
			rgOf[s.RgFlavor[k]] = append(rgOf[s.RgFlavor[k]], rg)
		}
	}
	overall := lblNone
	for p, a := range out {
		if modes[p] != modeNoFit {
			continue
		}
		if len(a) == 0 {
			overall = max(overall, lblNoMatchingFlavor)
			continue
		}
		rgMin := map[int32]int{} // per resource group the least severe blocker among its (alternative) flavors
		for fl, e := range a {
			if e.mode != modeNoFit {
				continue
			}
			for _, rg := range rgOf[fl] {
				if cur, ok := rgMin[rg]; !ok || e.label < cur {
					rgMin[rg] = e.label
				}
			}
		}
		for _, l := range rgMin { // across groups: co-requisites
			overall = max(overall, l)
		}
	}
	return noFitLabels[overall], named, nil
}

// QuotaReservedReason is entry.quotaReservedReason as processEntry leaves it (scheduler.go:424-513) for head i of a cycle — the Reason of the
// QuotaReserved=False condition requeueAndUpdate patches (:1188; features.UnadmittedWorkloadsObservability is on by default). "" for a head that
// was admitted. Twin of kueue_amd/no_fit_reason.py quota_reserved_reason, which the repository's tests run against the 110 condition Reasons
// of TestSchedule / TestScheduleForFairSharing / TestScheduleRecomputePreemptionTargets / TestScheduleForTAS.
func QuotaReservedReason(s *FlatSnapshot, h *FlatHeads, d *FlatDecisions, i int, fairSharing bool, isTASFlavor func(flavor int32) bool) (string, error) {
	switch {
	case d.Status[i] == stAssumed || d.Status[i] == stEvicted: // admitted; or evicted by handleFailedTASReplacement (:426-429), which sets no reason
		return "", nil
	case d.Skip[i] == skipOverlap || d.Skip[i] == skipNoLongerFits: // :471-484
		return kueue.WorkloadQuotaReservedReasonWaitingForQuota, nil
	case d.Mode[i] == modeNoFit: // :430-435
		r, _, err := NoFitReason(s, h, d, i, fairSharing, isTASFlavor)
		return r, err
	case d.Mode[i] == 2: // DeferredFit :455-468
		return kueue.WorkloadQuotaReservedReasonWaitingForPreemptedWorkloads, nil
	case d.Action[i] == actPreempt: // :495
		return kueue.WorkloadQuotaReservedReasonWaitingForPreemptedWorkloads, nil
	case d.Mode[i] == modePreempt && d.TgtOff[i+1] == d.TgtOff[i]: // :437-443
		return kueue.WorkloadQuotaReservedReasonWaitingForQuota, nil
	}
	return "", nil
}

//go:build kq_hip

package kqengine

// messages.go — the reference's user-visible status text rebuilt from the engine's reason records (kq_decisions.rsn_*,
// KQ_RSN_* in include/kq_engine.h). Twin of kueue_amd/messages.py, which this repository's tests run against the strings of
// the reference's TestAssignFlavors / TestSchedule tables.

import (
	"errors"
	"fmt"
	"sort"
	"strings"

	corev1 "k8s.io/api/core/v1"

	"sigs.k8s.io/kueue/pkg/resources"
)

const (
	RsnExceedsMaxCapacity  = 1
	RsnInsufficientUnused  = 2
	RsnNotInNomination     = 3
	RsnFlavorIneligible    = 4
	RsnResourceUnavailable = 5
	RsnSliceFlavorMismatch = 6
)

// IneligibleText returns the strings checkFlavorForPodSets (flavorassigner.go:1212-1261) produced on the host for a flavor whose
// ps_flavor_ok bit was clear ("untolerated taint ...", "flavor ... doesn't match node affinity").
type IneligibleText func(head, podset int, flavor string) []string

// ReasonText formats one record exactly as flavorassigner.go:1080, :1097, :1353-1359 and :1372-1373 do.
func ReasonText(f *resources.ResourceFormatter, s *FlatSnapshot, code uint8, flavor, resource int16, a, b, c int64) string {
	fl, rs := "", corev1.ResourceName("")
	if flavor >= 0 {
		fl = s.FlavorNames[flavor]
	}
	if resource >= 0 {
		rs = corev1.ResourceName(s.ResourceNames[resource])
	}
	amt := func(v int64) string { // AmountQuantityString (resource_formatter.go:93)
		if v == resources.Unlimited.Int64() {
			return resources.Unlimited.String()
		}
		return f.ResourceQuantityString(rs, v)
	}
	switch code {
	case RsnExceedsMaxCapacity:
		return fmt.Sprintf("insufficient quota for %s in flavor %s, previously considered podsets requests (%s) + current podset request (%s) > maximum capacity (%s)",
			rs, fl, amt(a), f.ResourceQuantityString(rs, b), amt(c))
	case RsnInsufficientUnused:
		return fmt.Sprintf("insufficient unused quota for %s in flavor %s, %s more needed", rs, fl, amt(a))
	case RsnNotInNomination:
		return fmt.Sprintf("skipping flavor %s as it is not found in the nomination mapping for resource %s", fl, rs)
	case RsnResourceUnavailable:
		return fmt.Sprintf("resource %s unavailable in ClusterQueue", rs)
	case RsnTASFailure: // kq_cycle_run_tas, flavorassigner.go:875: a = KQ_TAS_* status, b / c = its operands (notFitMessage tas_flavor_snapshot.go:1997);
		// TASFailureText needs the topology's name and the podset's slice size: callers with that context use it directly
		return TASFailureText(string(fl), int(a), int32(b), int32(c), 1)
	case RsnSliceFlavorMismatch: // flavorassigner.go:1134; a = the replaced slice's flavor for the resource (-1: none)
		orig := ""
		if a >= 0 {
			orig = s.FlavorNames[a]
		}
		return fmt.Sprintf("could not assign %s flavor since the original workload is assigned: %s", fl, orig)
	}
	return ""
}

// RsnTASFailure = KQ_RSN_TAS_FAILURE (kq_engine.h).
const RsnTASFailure = 200

// TASFailureText words TASAssignmentsResult.Failure().Reason from the record's operands. The node-exclusion statistics the reference
// appends to the "doesn't allow to fit any" form are not carried by the operands (the Go side can recompute them from its snapshot).
func TASFailureText(topology string, status int, a, b int32, sliceSize int32) string {
	const (
		tasNotFit       = 1 // KQ_TAS_NOT_FIT ... (include/kq_tas.h)
		tasNoLevel      = 2
		tasSliceAbove   = 3
		tasBadSliceSize = 4
	)
	unit := "pod"
	if sliceSize != 1 {
		unit = "slice"
	}
	switch status {
	case tasNotFit:
		if a == 0 {
			return fmt.Sprintf("topology %q doesn't allow to fit any of %d %s(s)", topology, b, unit)
		}
		return fmt.Sprintf("topology %q allows to fit only %d out of %d %s(s)", topology, a, b, unit)
	case tasNoLevel:
		return "no requested topology level"
	case tasSliceAbove:
		return "podset slice topology is above the podset topology"
	case tasBadSliceSize:
		return "slice topology requested, but slice size not provided"
	}
	return fmt.Sprintf("topology %q doesn't allow to fit", topology)
}

// TASReplacementFailureText words the two failures only the second pass after a node failure produces (findReplacementAssignment
// tas_flavor_snapshot.go:694-696, :727; KQ_TAS_STALE = 9 with operand a = index of the stale domain in the admission's TopologyAssignment after
// deleteDomain, KQ_TAS_NO_REPLACEMENT = 10): the caller resolves the names from wl.Status. ok = false: not one of the two.
func TASReplacementFailureText(status int, staleDomainFirstValue, unhealthyNode string) (string, bool) {
	switch status {
	case 9:
		return fmt.Sprintf("Cannot replace the node, because the existing topologyAssignment is invalid, as it contains the stale domain %v", staleDomainFirstValue), true
	case 10:
		return fmt.Sprintf("cannot find replacement assignment for unhealthy node: %v", unhealthyNode), true
	}
	return "", false
}

// RsnTruncated = KQ_RSN_TRUNCATED (kq_engine.h): the head's reason window overflowed, its record list is incomplete.
const RsnTruncated = 255

// ErrReasonsTruncated tells the caller to run the cycle again with a larger rsn_cap (or to take the stock Go path for the head).
var ErrReasonsTruncated = errors.New("kqengine: reason records of the head were truncated (KQ_RSN_TRUNCATED)")

// PodSetReasons returns Status.reasons of every podset of head i, sorted as Status.Message sorts them (flavorassigner.go:361).
func PodSetReasons(f *resources.ResourceFormatter, s *FlatSnapshot, h *FlatHeads, d *FlatDecisions, i int, inel IneligibleText) ([][]string, error) {
	out := make([][]string, h.PsOff[i+1]-h.PsOff[i])
	for k := d.RsnOff[i]; k < d.RsnOff[i+1]; k++ {
		if d.RsnCode[k] == RsnTruncated {
			return nil, ErrReasonsTruncated
		}
		ps := int(d.RsnPodset[k])
		if d.RsnCode[k] == RsnFlavorIneligible {
			out[ps] = append(out[ps], inel(i, ps, s.FlavorNames[d.RsnFlavor[k]])...)
			continue
		}
		out[ps] = append(out[ps], ReasonText(f, s, d.RsnCode[k], d.RsnFlavor[k], d.RsnResource[k], d.RsnA[k], d.RsnB[k], d.RsnC[k]))
	}
	for _, r := range out {
		sort.Strings(r)
	}
	return out, nil
}

// AssignmentMessage = Assignment.Message (flavorassigner.go:229-247).
func AssignmentMessage(podsetNames []string, reasons [][]string) string {
	var b strings.Builder
	for p, r := range reasons {
		if len(r) == 0 {
			continue
		}
		if b.Len() > 0 {
			b.WriteString("; ")
		}
		b.WriteString("couldn't assign flavors to pod set ")
		b.WriteString(podsetNames[p])
		b.WriteString(": ")
		b.WriteString(strings.Join(r, ", "))
	}
	return b.String()
}

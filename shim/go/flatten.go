//go:build kq_hip

package kqengine

// flatten.go — *schdcache.Snapshot and the cycle's heads as the flat SoA image of include/kq_engine.h.
//
// Canonical orders (SURVEY.md §8c; the engine's results are defined under them): ClusterQueues and Cohorts by name, children
// lists by name, flavors in the order of their first appearance over the name-sorted ClusterQueues' resource groups, resources
// by name; admitted rows grouped by ClusterQueue, inside a ClusterQueue by workload key.
// Not compiled here (no Go toolchain in this image); written against the reference at /root/reference:
//   pkg/cache/scheduler/{snapshot.go:53, clusterqueue_snapshot.go:53, cohort_snapshot.go:25, resource_node.go:30-45, resource.go:26}
//   pkg/workload/workload.go:245 (Info), :276 (PodSetResources), :115 (AssignmentClusterQueueState)
//   pkg/cache/queue/manager.go:895 (Head)

import (
	"hash/fnv"
	"math"
	"sort"
	"strconv"
	"time"

	"github.com/go-logr/logr"
	corev1 "k8s.io/api/core/v1"
	apimeta "k8s.io/apimachinery/pkg/api/meta"
	metav1 "k8s.io/apimachinery/pkg/apis/meta/v1"

	kueue "sigs.k8s.io/kueue/apis/kueue/v1beta2"
	qcache "sigs.k8s.io/kueue/pkg/cache/queue"
	schdcache "sigs.k8s.io/kueue/pkg/cache/scheduler"
	"sigs.k8s.io/kueue/pkg/resources"
	"sigs.k8s.io/kueue/pkg/util/priority"
	"sigs.k8s.io/kueue/pkg/workload"
	"sigs.k8s.io/kueue/pkg/workloadslicing"
	workloadevict "sigs.k8s.io/kueue/pkg/workload/evict"
)

const (
	nilLimit   = int64(-1)        // KQ_NIL_LIMIT
	qfQuota    = uint8(0x1)       // KQ_QF_QUOTA
	qfSubtree  = uint8(0x2)       // KQ_QF_SUBTREE
	headQuota  = uint32(0x1)      // KQ_HEAD_HAS_QUOTA_RESERVATION
	headPre    = uint32(0x2)      // KQ_HEAD_IS_PREEMPTOR
	headUnhealthy   = uint32(0x8)  // KQ_HEAD_HAS_UNHEALTHY_NODES: workload.HasUnhealthyNodes
	headUnhealthyTA = uint32(0x10) // KQ_HEAD_UNHEALTHY_ASSIGNMENT: workload.HasTopologyAssignmentWithUnhealthyNode
	headLast   = uint32(0x4)      // KQ_HEAD_HAS_LAST_ASSIGNMENT
	admEvicted = uint8(0x1)       // KQ_ADM_EVICTED
)

// Index is the Go-side dictionary of one flattened snapshot.
type Index struct {
	CQ, Cohort, Flavor, Resource map[string]int32
	AdmRow  map[string]int32 // workload.Reference -> admitted row (workload slices: the row a head replaces)
	AdmInfo []*workload.Info // by admitted row
}

func amount(a resources.Amount) int64 { return a.Int64() } // Unlimited == math.MaxInt64 == KQ_UNLIMITED

func limit(a *resources.Amount) int64 {
	if a == nil {
		return nilLimit
	}
	return a.Int64()
}

func policyWord(cq *schdcache.ClusterQueueSnapshot, strategy kueue.QueueingStrategy) (uint32, int32) {
	pol := func(p kueue.PreemptionPolicy) uint32 {
		switch p {
		case kueue.PreemptionPolicyLowerPriority:
			return 1
		case kueue.PreemptionPolicyLowerOrNewerEqualPriority:
			return 2
		case kueue.PreemptionPolicyAny:
			return 3
		}
		return 0
	}
	w := pol(cq.Preemption.WithinClusterQueue) | pol(cq.Preemption.ReclaimWithinCohort)<<2
	if cq.Preemption.ReclaimWithinCohort == "" {
		w |= 1 << 11 // object built without API defaulting (header: KQ_POL_RECLAIM_UNSET)
	}
	thr := int32(0)
	if b := cq.Preemption.BorrowWithinCohort; b != nil && b.Policy != kueue.BorrowWithinCohortPolicyNever {
		w |= 1 << 4
		if b.MaxPriorityThreshold != nil {
			w |= 1 << 5
			thr = *b.MaxPriorityThreshold
		}
	}
	if cq.FlavorFungibility.WhenCanBorrow == kueue.TryNextFlavor {
		w |= 1 << 6
	}
	if cq.FlavorFungibility.WhenCanPreempt == kueue.TryNextFlavor {
		w |= 1 << 7
	}
	if p := cq.FlavorFungibility.Preference; p != nil {
		switch *p {
		case kueue.BorrowingOverPreemption:
			w |= 1 << 8
		case kueue.PreemptionOverBorrowing:
			w |= 2 << 8
		}
	}
	if strategy == kueue.StrictFIFO {
		w |= 1 << 10
	}
	return w, thr
}

func sortedKeys[V any](m map[string]V) []string {
	ks := make([]string, 0, len(m))
	for k := range m {
		ks = append(ks, k)
	}
	sort.Strings(ks)
	return ks
}

// Flatten builds the kq_snapshot image of cache.Snapshot() (snapshot.go:171). strategies: ClusterQueue -> QueueingStrategy (the
// snapshot does not carry it; the queue manager does).
func Flatten(log logr.Logger, snap *schdcache.Snapshot, strategies map[kueue.ClusterQueueReference]kueue.QueueingStrategy,
	ordering workload.Ordering, now time.Time) (*FlatSnapshot, *Index) {
	s := &FlatSnapshot{PodsResource: -1}
	ix := &Index{CQ: map[string]int32{}, Cohort: map[string]int32{}, Flavor: map[string]int32{}, Resource: map[string]int32{}}
	cqs := map[string]*schdcache.ClusterQueueSnapshot{}
	for name, cq := range snap.ClusterQueues() {
		cqs[string(name)] = cq
	}
	cohorts := map[string]*schdcache.CohortSnapshot{}
	for name, c := range snap.Cohorts() {
		cohorts[string(name)] = c
	}
	s.CQNames, s.CohortNames = sortedKeys(cqs), sortedKeys(cohorts)
	for i, n := range s.CQNames {
		ix.CQ[n] = int32(i)
	}
	nq := int32(len(s.CQNames))
	for i, n := range s.CohortNames {
		ix.Cohort[n] = nq + int32(i)
	}
	s.NCQ, s.NCohort = nq, int32(len(s.CohortNames))
	N := int(s.NCQ + s.NCohort)
	// dictionaries: flavors by first appearance, resources by name
	resSet := map[string]struct{}{}
	addQuotas := func(q map[resources.FlavorResource]schdcache.ResourceQuota) {
		for fr := range q {
			resSet[string(fr.Resource)] = struct{}{}
		}
	}
	for _, n := range s.CQNames {
		for _, rg := range cqs[n].ResourceGroups {
			for _, f := range rg.Flavors {
				if _, ok := ix.Flavor[string(f)]; !ok {
					ix.Flavor[string(f)] = int32(len(s.FlavorNames))
					s.FlavorNames = append(s.FlavorNames, string(f))
				}
			}
			for r := range rg.CoveredResources {
				resSet[string(r)] = struct{}{}
			}
		}
		addQuotas(cqs[n].ResourceNode.Quotas)
	}
	for _, n := range s.CohortNames {
		for fr := range cohorts[n].ResourceNode.Quotas {
			if _, ok := ix.Flavor[string(fr.Flavor)]; !ok {
				ix.Flavor[string(fr.Flavor)] = int32(len(s.FlavorNames))
				s.FlavorNames = append(s.FlavorNames, string(fr.Flavor))
			}
		}
		addQuotas(cohorts[n].ResourceNode.Quotas)
	}
	s.ResourceNames = sortedKeys(resSet)
	for i, r := range s.ResourceNames {
		ix.Resource[r] = int32(i)
		if r == string(corev1.ResourcePods) {
			s.PodsResource = int32(i)
		}
	}
	s.NFlavor, s.NResource = int32(len(s.FlavorNames)), int32(len(s.ResourceNames))
	// Requests.Iter order: FNV-1a64(name), then name (pkg/resources/slice_requests.go:35-60)
	type hr struct {
		h uint64
		n string
		i int32
	}
	hrs := make([]hr, len(s.ResourceNames))
	for i, r := range s.ResourceNames {
		f := fnv.New64a()
		_, _ = f.Write([]byte(r))
		hrs[i] = hr{f.Sum64(), r, int32(i)}
	}
	sort.Slice(hrs, func(a, b int) bool {
		if hrs[a].h != hrs[b].h {
			return hrs[a].h < hrs[b].h
		}
		return hrs[a].n < hrs[b].n
	})
	s.ResourceOrder = make([]int32, len(hrs))
	for rank, x := range hrs {
		s.ResourceOrder[x.i] = int32(rank)
	}
	nfr := int(s.NFlavor * s.NResource)
	cells := N * nfr
	s.Nominal, s.SubtreeQuota, s.Usage = make([]int64, cells), make([]int64, cells), make([]int64, cells)
	s.BorrowLimit, s.LendLimit = make([]int64, cells), make([]int64, cells)
	for i := range s.BorrowLimit {
		s.BorrowLimit[i], s.LendLimit[i] = nilLimit, nilLimit
	}
	s.QuotaFlags = make([]uint8, cells)
	s.Parent = make([]int32, N)
	s.FairWeight = make([]float64, N)
	frOf := func(fr resources.FlavorResource) int { return int(ix.Flavor[string(fr.Flavor)]*s.NResource + ix.Resource[string(fr.Resource)]) }
	fillNode := func(node int, quotas map[resources.FlavorResource]schdcache.ResourceQuota, sq, us resources.FlavorResourceQuantities) {
		for fr, q := range quotas {
			o := node*nfr + frOf(fr)
			s.Nominal[o], s.BorrowLimit[o], s.LendLimit[o] = amount(q.Nominal), limit(q.BorrowingLimit), limit(q.LendingLimit)
			s.QuotaFlags[o] |= qfQuota
		}
		for fr, v := range sq {
			o := node*nfr + frOf(fr)
			s.SubtreeQuota[o] = amount(v)
			s.QuotaFlags[o] |= qfSubtree
		}
		for fr, v := range us {
			s.Usage[node*nfr+frOf(fr)] = amount(v)
		}
	}
	s.CQPolicy, s.CQBorrowPrioThreshold, s.CQGeneration = make([]uint32, nq), make([]int32, nq), make([]int64, nq)
	s.CQRgOff = []int32{0}
	s.RgFlavorOff, s.RgResOff = []int32{0}, []int32{0}
	s.CQAdmOff = []int32{0}
	s.AdmUseOff = []int32{0}
	type row struct {
		key string
		wl  *workload.Info
	}
	var uids []string
	for i, n := range s.CQNames {
		cq := cqs[n]
		s.Parent[i] = -1
		if cq.HasParent() {
			s.Parent[i] = ix.Cohort[string(cq.Parent().GetName())]
		}
		s.FairWeight[i] = cq.FairWeight
		fillNode(i, cq.ResourceNode.Quotas, cq.ResourceNode.SubtreeQuota, cq.ResourceNode.Usage)
		s.CQPolicy[i], s.CQBorrowPrioThreshold[i] = policyWord(cq, strategies[cq.Name])
		s.CQGeneration[i] = cq.AllocatableResourceGeneration
		for _, rg := range cq.ResourceGroups {
			for _, f := range rg.Flavors {
				s.RgFlavor = append(s.RgFlavor, ix.Flavor[string(f)])
			}
			covered := make([]string, 0, rg.CoveredResources.Len())
			for r := range rg.CoveredResources {
				covered = append(covered, string(r))
			}
			sort.Strings(covered)
			for _, r := range covered {
				s.RgRes = append(s.RgRes, ix.Resource[r])
			}
			s.RgFlavorOff = append(s.RgFlavorOff, int32(len(s.RgFlavor)))
			s.RgResOff = append(s.RgResOff, int32(len(s.RgRes)))
		}
		s.CQRgOff = append(s.CQRgOff, int32(len(s.RgFlavorOff)-1))
		rows := make([]row, 0, len(cq.Workloads))
		for k, wl := range cq.Workloads {
			rows = append(rows, row{string(k), wl})
		}
		sort.Slice(rows, func(a, b int) bool { return rows[a].key < rows[b].key })
		for _, r := range rows {
			wl := r.wl
			if ix.AdmRow == nil {
				ix.AdmRow = map[string]int32{}
			}
			ix.AdmRow[r.key] = int32(len(s.AdmKeys))
			ix.AdmInfo = append(ix.AdmInfo, wl)
			s.AdmKeys = append(s.AdmKeys, r.key)
			s.AdmPriority = append(s.AdmPriority, priority.EffectivePriority(log, wl.Obj))
			s.AdmQueueTs = append(s.AdmQueueTs, ordering.GetQueueOrderTimestamp(wl.Obj).UnixNano())
			rt := now
			if c := apimeta.FindStatusCondition(wl.Obj.Status.Conditions, kueue.WorkloadQuotaReserved); c != nil && c.Status == metav1.ConditionTrue {
				rt = c.LastTransitionTime.Time // quotaReservationTime common/ordering.go:94-100
			}
			s.AdmReserveTs = append(s.AdmReserveTs, rt.UnixNano())
			uids = append(uids, string(wl.Obj.UID))
			fl := uint8(0)
			if workloadevict.IsEvicted(wl.Obj) {
				fl |= admEvicted
			}
			s.AdmFlags = append(s.AdmFlags, fl)
			// Info.Usage().Quota.Assigned, zero quantities included (candidate_generator.go:54), ascending fr
			use := wl.Usage().Quota.Assigned
			frs := make([]int, 0, len(use))
			qty := map[int]int64{}
			for fr, v := range use {
				o := frOf(fr)
				frs = append(frs, o)
				qty[o] = amount(v)
			}
			sort.Ints(frs)
			for _, o := range frs {
				s.AdmUseFr = append(s.AdmUseFr, int32(o))
				s.AdmUseQty = append(s.AdmUseQty, qty[o])
			}
			s.AdmUseOff = append(s.AdmUseOff, int32(len(s.AdmUseFr)))
		}
		s.CQAdmOff = append(s.CQAdmOff, int32(len(s.AdmKeys)))
	}
	s.NAdm = int32(len(s.AdmKeys))
	// rank of Obj.UID under Go string order (common/ordering.go:77)
	order := make([]int, len(uids))
	for i := range order {
		order[i] = i
	}
	sort.Slice(order, func(a, b int) bool { return uids[order[a]] < uids[order[b]] })
	s.AdmUIDRank = make([]uint32, len(uids))
	for rank, i := range order {
		s.AdmUIDRank[i] = uint32(rank)
	}
	s.ChildCohortOff, s.ChildCQOff = []int32{0}, []int32{0}
	for j, n := range s.CohortNames {
		c := cohorts[n]
		node := int(nq) + j
		s.Parent[node] = -1
		if c.HasParent() {
			s.Parent[node] = ix.Cohort[string(c.Parent().GetName())]
		}
		s.FairWeight[node] = c.FairWeight
		fillNode(node, c.ResourceNode.Quotas, c.ResourceNode.SubtreeQuota, c.ResourceNode.Usage)
		var kc, kq []int32
		for _, ch := range c.ChildCohorts() {
			kc = append(kc, ix.Cohort[string(ch.GetName())])
		}
		for _, q := range c.ChildCQs() {
			kq = append(kq, ix.CQ[string(q.Name)])
		}
		sort.Slice(kc, func(a, b int) bool { return kc[a] < kc[b] }) // index order == name order
		sort.Slice(kq, func(a, b int) bool { return kq[a] < kq[b] })
		s.ChildCohort = append(s.ChildCohort, kc...)
		s.ChildCQ = append(s.ChildCQ, kq...)
		s.ChildCohortOff = append(s.ChildCohortOff, int32(len(s.ChildCohort)))
		s.ChildCQOff = append(s.ChildCQOff, int32(len(s.ChildCQ)))
	}
	return s, ix
}

// FlavorEligible is the host-side checkFlavorForPodSets (flavorassigner.go:1212-1261: taints, node affinity, TAS match) for one
// (head, podset, flavor); the scheduler passes its own implementation.
type FlavorEligible func(h *qcache.Head, podset int, flavor kueue.ResourceFlavorReference) bool

func hash64(h workload.EquivalenceHash) uint64 {
	if h == "" {
		return 0 // SchedulingHashUnknown
	}
	v, err := strconv.ParseUint(string(h), 16, 64) // the first 16 hex digits of the SHA-256 (workload.go:425)
	if err != nil {
		return 0
	}
	if v == 0 {
		v = math.MaxUint64
	}
	return v
}

// FlattenHeads builds the kq_heads image of the entries that passed nominate's gatekeeping (scheduler.go:673-695), in canonical
// order (ClusterQueue name ascending; the caller sorts `heads` that way and keeps the permutation).
func FlattenHeads(log logr.Logger, s *FlatSnapshot, ix *Index, heads []*qcache.Head, cycle int64, ordering workload.Ordering, ok FlavorEligible) *FlatHeads {
	h := &FlatHeads{N: int32(len(heads)), Cycle: cycle, PsOff: []int32{0}, PsReqOff: []int32{0}}
	nR := int(s.NResource)
	nfw := (int(s.NFlavor) + 63) / 64
	for _, hd := range heads {
		wl := &hd.Info
		h.CQ = append(h.CQ, ix.CQ[string(wl.ClusterQueue)])
		h.Priority = append(h.Priority, priority.EffectivePriority(log, wl.Obj))
		h.QueueTs = append(h.QueueTs, ordering.GetQueueOrderTimestamp(wl.Obj).UnixNano())
		fl := uint32(0)
		if workload.HasQuotaReservation(wl.Obj) {
			fl |= headQuota
		}
		if hd.IsPreemptor {
			fl |= headPre
		}
		if workload.HasUnhealthyNodes(wl.Obj) { // the second pass after a node failure (kq_cycle_run_tas; TASCycle.PsAdmFlavor / PsEx*)
			fl |= headUnhealthy
		}
		if workload.HasTopologyAssignmentWithUnhealthyNode(wl.Obj) {
			fl |= headUnhealthyTA
		}
		la := wl.LastAssignment
		if la != nil {
			fl |= headLast
			h.LastGeneration = append(h.LastGeneration, la.ClusterQueueGeneration)
			h.LastCycle = append(h.LastCycle, la.SchedulingCycle)
			h.LastHash = append(h.LastHash, hash64(la.SchedulingHash))
		} else {
			h.LastGeneration, h.LastCycle, h.LastHash = append(h.LastGeneration, 0), append(h.LastCycle, 0), append(h.LastHash, 0)
		}
		h.Flags = append(h.Flags, fl)
		h.Hash = append(h.Hash, hash64(wl.SchedulingHash))
		for pi, ps := range wl.TotalRequests {
			h.PsCount = append(h.PsCount, ps.Count)
			mc := int32(-1)
			if pi < len(wl.Obj.Spec.PodSets) && wl.Obj.Spec.PodSets[pi].MinCount != nil {
				mc = *wl.Obj.Spec.PodSets[pi].MinCount
			}
			h.PsMinCount = append(h.PsMinCount, mc)
			ps.Requests.ForEach(func(r corev1.ResourceName, v int64) {
				h.ReqRes = append(h.ReqRes, ix.Resource[string(r)])
				h.ReqQty = append(h.ReqQty, v)
			})
			h.PsReqOff = append(h.PsReqOff, int32(len(h.ReqRes)))
			words := make([]uint64, nfw)
			for name, f := range ix.Flavor {
				if ok(hd, pi, kueue.ResourceFlavorReference(name)) {
					words[f>>6] |= 1 << uint(f&63)
				}
			}
			h.PsFlavorOK = append(h.PsFlavorOK, words...)
			lt := make([]int32, nR)
			for i := range lt {
				lt[i] = -1
			}
			if la != nil && pi < len(la.LastTriedFlavorIdx) {
				for r, idx := range la.LastTriedFlavorIdx[pi] {
					lt[ix.Resource[string(r)]] = int32(idx)
				}
			}
			h.PsLastTried = append(h.PsLastTried, lt...)
		}
		h.PsOff = append(h.PsOff, int32(len(h.PsCount)))
	}
	flattenSlices(s, ix, heads, h)
	flattenGroups(heads, h)
	return h
}

// flattenGroups fills kq_heads.ps_group: the PodSetGroupName of every podset as an id inside its head (flavorassigner.go:782-790 groups
// the podsets by that name whether or not the ClusterQueue holds a TAS flavor). Left nil when no podset of the batch names a group.
func flattenGroups(heads []*qcache.Head, h *FlatHeads) {
	var grp []int32
	any := false
	for _, hd := range heads {
		ids := map[string]int32{}
		for pi := range hd.Info.TotalRequests {
			g := int32(-1)
			if pi < len(hd.Info.Obj.Spec.PodSets) {
				if tr := hd.Info.Obj.Spec.PodSets[pi].TopologyRequest; tr != nil && tr.PodSetGroupName != nil {
					id, ok := ids[*tr.PodSetGroupName]
					if !ok {
						id = int32(len(ids))
						ids[*tr.PodSetGroupName] = id
					}
					g, any = id, true
				}
			}
			grp = append(grp, g)
		}
	}
	if any {
		h.PsGroup = grp
	}
}

// flattenSlices fills the kq_heads.slice_* columns: for a head that replaces an admitted workload slice
// (workloadslicing.ReplacedWorkloadSlice, scheduler.go:883: annotation -> queue.Workloads, same namespace) the admitted row of the
// old slice and, per podset / request of the head, what the old slice holds: Count, Flavors[res], Requests[res]
// (replaceWorkloadSlice.TotalRequests[psID], flavorassigner.go:1127; findOldPodSetRequest :1046 looks the podset up by NAME).
func flattenSlices(s *FlatSnapshot, ix *Index, heads []*qcache.Head, h *FlatHeads) {
	any := false
	rows := make([]int32, len(heads))
	olds := make([]*workload.Info, len(heads))
	for i, hd := range heads {
		rows[i] = -1
		key := workloadslicing.ReplacementForKey(hd.Info.Obj)
		if key == nil {
			continue
		}
		if row, ok := ix.AdmRow[string(*key)]; ok && ix.AdmInfo[row].ClusterQueue == hd.Info.ClusterQueue &&
			ix.AdmInfo[row].Obj.Namespace == hd.Info.Obj.Namespace {
			rows[i], olds[i], any = row, ix.AdmInfo[row], true
		}
	}
	if !any {
		return
	}
	h.SliceRow = rows
	pods := corev1.ResourcePods
	for i, hd := range heads {
		for _, ps := range hd.Info.TotalRequests {
			var ops *workload.PodSetResources
			if olds[i] != nil {
				for k := range olds[i].TotalRequests {
					if olds[i].TotalRequests[k].Name == ps.Name {
						ops = &olds[i].TotalRequests[k]
					}
				}
			}
			cnt, pf, pq := int32(0), int32(-1), int64(0)
			if ops != nil {
				cnt = ops.Count
				if f, ok := ops.Flavors[pods]; ok {
					pf, pq = ix.Flavor[string(f)], ops.Requests.ResourceValue(pods)
				}
			}
			h.PsSliceCount = append(h.PsSliceCount, cnt)
			h.PsSlicePodsFlavor, h.PsSlicePodsQty = append(h.PsSlicePodsFlavor, pf), append(h.PsSlicePodsQty, pq)
			ps.Requests.ForEach(func(r corev1.ResourceName, _ int64) {
				f, q := int32(-1), int64(0)
				if ops != nil {
					if fl, ok := ops.Flavors[r]; ok {
						f = ix.Flavor[string(fl)]
					}
					q = ops.Requests.ResourceValue(r)
				}
				h.ReqSliceFlavor, h.ReqSliceQty = append(h.ReqSliceFlavor, f), append(h.ReqSliceQty, q)
			})
		}
	}
}

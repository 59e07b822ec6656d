//go:build kq_hip

// Package kqengine binds the MI355X admission engine (include/kq_engine.h) into Kueue's scheduler.
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain (`go version`: not found). These files are the
// source a Kueue maintainer drops under pkg/scheduler/kqengine and builds with
//
//	CGO_ENABLED=1 go build -tags kq_hip ./cmd/kueue
//
// (the reference builds with CGO_ENABLED=0, Makefile:69, so the tag keeps the stock build untouched).
//
//	kqengine.go  the cgo binding: Engine, flat SoA buffers, pinning, RunCycle, the pending-side entry points
//	flatten.go   *schdcache.Snapshot / []qcache.Head  ->  FlatSnapshot / FlatHeads
//	apply.go     FlatDecisions -> entries (assignment, targets, status, requeue reason, inadmissible message)
//	messages.go  reason records -> the strings flavorassigner.go formats (twin of kueue_amd/messages.py, which the tests run)
//
// Call site: (*Scheduler).schedule, pkg/scheduler/scheduler.go:340-362 — between cache.Snapshot() and the requeue loop.
// See INTEGRATION.md for the patch.
package kqengine

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../kueue_amd -lkq_engine -Wl,-rpath,${SRCDIR}/../../kueue_amd
#include <stdlib.h>
#include "kq_engine.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"
)

// Engine owns one kq_engine* bound to one HIP device. All calls must come from the goroutine that
// created it (the scheduler goroutine; it is pinned with runtime.LockOSThread in New).
type Engine struct {
	h *C.kq_engine
}

// Config mirrors scheduler.Option values (cmd/kueue/main.go:675-688).
type Config struct {
	Device             int
	FairSharing        bool
	FSStrategies       []int // KQ_FS_*
	Gates              uint32
	QuotaCheckStrategy int
}

func New(cfg Config) (*Engine, error) {
	runtime.LockOSThread()
	c := C.kq_config{abi_version: C.KQ_ABI_VERSION, device: C.int32_t(cfg.Device), gates: C.uint32_t(cfg.Gates),
		quota_check_strategy: C.int32_t(cfg.QuotaCheckStrategy)}
	if cfg.FairSharing {
		c.fair_sharing = 1
	}
	c.n_fs_strategies = C.int32_t(len(cfg.FSStrategies))
	for i, s := range cfg.FSStrategies {
		if i < 2 {
			c.fs_strategies[i] = C.int32_t(s)
		}
	}
	e := &Engine{}
	if rc := C.kq_engine_create(&c, &e.h); rc != 0 {
		return nil, fmt.Errorf("kq_engine_create: %s", C.GoString(C.kq_strerror(rc)))
	}
	return e, nil
}

func (e *Engine) Close() { C.kq_engine_destroy(e.h); e.h = nil }

func (e *Engine) err(what string, rc C.int) error {
	return fmt.Errorf("%s: %s (%s)", what, C.GoString(C.kq_strerror(rc)), C.GoString(C.kq_last_error(e.h)))
}

// FlatSnapshot / FlatHeads / FlatDecisions are Go-side SoA buffers (plain slices) filled by flatten.go from
// *schdcache.Snapshot and []qcache.Head. The C structs that carry their addresses are allocated with C.malloc (so the struct
// itself is not Go memory) and every slice's backing array is pinned with runtime.Pinner (Go 1.21+) for the duration of
// the call: that is what the cgo pointer-passing rules ask for a C struct holding several Go pointers. The engine keeps no
// pointer after returning.
type FlatSnapshot struct {
	NCQ, NCohort, NFlavor, NResource, PodsResource     int32
	ResourceOrder, Parent                              []int32
	ChildCohortOff, ChildCohort, ChildCQOff, ChildCQ   []int32
	FairWeight                                         []float64
	Nominal, BorrowLimit, LendLimit, SubtreeQuota, Usage []int64
	QuotaFlags                                         []uint8
	CQRgOff, RgFlavorOff, RgFlavor, RgResOff, RgRes    []int32
	CQPolicy                                           []uint32
	CQBorrowPrioThreshold                              []int32
	CQGeneration                                       []int64
	NAdm                                               int32
	CQAdmOff                                           []int32
	AdmPriority, AdmQueueTs, AdmReserveTs              []int64
	AdmUIDRank                                         []uint32
	AdmFlags                                           []uint8
	AdmUseOff, AdmUseFr                                []int32
	AdmUseQty                                          []int64

	// dictionaries kept on the Go side (names never cross the boundary)
	CQNames, CohortNames, FlavorNames, ResourceNames []string
	AdmKeys                                           []string // workload.Reference of every admitted row
}

type FlatHeads struct {
	N                                         int32
	Cycle                                     int64
	CQ                                        []int32
	Priority, QueueTs                         []int64
	Flags                                     []uint32
	PsOff, PsCount, PsMinCount, PsReqOff, ReqRes []int32
	ReqQty                                    []int64
	PsFlavorOK                                []uint64
	PsLastTried                               []int32
	LastGeneration, LastCycle                 []int64
	LastHash, Hash                            []uint64
	// workload slices (ElasticJobsViaWorkloadSlices): nil when no head replaces a slice
	SliceRow, PsSliceCount, ReqSliceFlavor, PsSlicePodsFlavor []int32
	ReqSliceQty, PsSlicePodsQty                               []int64
	// PodSet.TopologyRequest.PodSetGroupName per podset (id inside the head, -1 none): assignFlavors scans flavors once per group
	// (flavorassigner.go:782-860). nil when no podset of the batch is in a group.
	PsGroup []int32
}

type FlatDecisions struct {
	Status, Action, NominatedMode, Mode, RequeueReason, Skip []uint8
	Borrowing, Order                                          []int32
	Flavor                                                    []int32
	ResMode                                                   []uint8
	TriedIdx                                                  []int32
	PsCount                                                   []int32
	TgtOff, TgtAdm                                            []int32
	TgtReason                                                 []uint8
	RsnOff                                                    []int32
	RsnCode, RsnPodset                                        []uint8
	RsnFlavor, RsnResource                                    []int16
	RsnA, RsnB, RsnC                                          []int64
}

// NewDecisions sizes the output buffers for a heads batch: per head, per (podset, resource), a target pool of tgtCap rows and
// rsnCap reason records.
func NewDecisions(h *FlatHeads, nResource int32, tgtCap, rsnCap int) *FlatDecisions {
	n, nps := int(h.N), int(h.PsOff[h.N])
	d := &FlatDecisions{}
	d.Status, d.Action, d.NominatedMode = make([]uint8, n), make([]uint8, n), make([]uint8, n)
	d.Mode, d.RequeueReason, d.Skip = make([]uint8, n), make([]uint8, n), make([]uint8, n)
	d.Borrowing, d.Order = make([]int32, n), make([]int32, n)
	d.Flavor, d.ResMode, d.TriedIdx = make([]int32, nps*int(nResource)), make([]uint8, nps*int(nResource)), make([]int32, nps*int(nResource))
	d.PsCount = make([]int32, nps)
	d.TgtOff, d.TgtAdm, d.TgtReason = make([]int32, n+1), make([]int32, tgtCap), make([]uint8, tgtCap)
	d.RsnOff = make([]int32, n+1)
	d.RsnCode, d.RsnPodset = make([]uint8, rsnCap), make([]uint8, rsnCap)
	d.RsnFlavor, d.RsnResource = make([]int16, rsnCap), make([]int16, rsnCap)
	d.RsnA, d.RsnB, d.RsnC = make([]int64, rsnCap), make([]int64, rsnCap), make([]int64, rsnCap)
	return d
}

// pin returns &s[0] as an unsafe.Pointer (nil for an empty slice) after pinning the backing array.
func pin[T any](p *runtime.Pinner, s []T) unsafe.Pointer {
	if len(s) == 0 {
		return nil
	}
	p.Pin(&s[0])
	return unsafe.Pointer(&s[0])
}

func fillSnapshot(p *runtime.Pinner, c *C.kq_snapshot, s *FlatSnapshot) {
	c.n_cq, c.n_cohort, c.n_flavor, c.n_resource = C.int32_t(s.NCQ), C.int32_t(s.NCohort), C.int32_t(s.NFlavor), C.int32_t(s.NResource)
	c.pods_resource = C.int32_t(s.PodsResource)
	c.resource_order = (*C.int32_t)(pin(p, s.ResourceOrder))
	c.parent = (*C.int32_t)(pin(p, s.Parent))
	c.child_cohort_off = (*C.int32_t)(pin(p, s.ChildCohortOff))
	c.child_cohort = (*C.int32_t)(pin(p, s.ChildCohort))
	c.child_cq_off = (*C.int32_t)(pin(p, s.ChildCQOff))
	c.child_cq = (*C.int32_t)(pin(p, s.ChildCQ))
	c.fair_weight = (*C.double)(pin(p, s.FairWeight))
	c.nominal = (*C.int64_t)(pin(p, s.Nominal))
	c.borrow_limit = (*C.int64_t)(pin(p, s.BorrowLimit))
	c.lend_limit = (*C.int64_t)(pin(p, s.LendLimit))
	c.subtree_quota = (*C.int64_t)(pin(p, s.SubtreeQuota))
	c.usage = (*C.int64_t)(pin(p, s.Usage))
	c.quota_flags = (*C.uint8_t)(pin(p, s.QuotaFlags))
	c.cq_rg_off = (*C.int32_t)(pin(p, s.CQRgOff))
	c.rg_flavor_off = (*C.int32_t)(pin(p, s.RgFlavorOff))
	c.rg_flavor = (*C.int32_t)(pin(p, s.RgFlavor))
	c.rg_res_off = (*C.int32_t)(pin(p, s.RgResOff))
	c.rg_res = (*C.int32_t)(pin(p, s.RgRes))
	c.cq_policy = (*C.uint32_t)(pin(p, s.CQPolicy))
	c.cq_borrow_prio_threshold = (*C.int32_t)(pin(p, s.CQBorrowPrioThreshold))
	c.cq_generation = (*C.int64_t)(pin(p, s.CQGeneration))
	c.n_adm = C.int32_t(s.NAdm)
	c.cq_adm_off = (*C.int32_t)(pin(p, s.CQAdmOff))
	c.adm_priority = (*C.int64_t)(pin(p, s.AdmPriority))
	c.adm_queue_ts = (*C.int64_t)(pin(p, s.AdmQueueTs))
	c.adm_reserve_ts = (*C.int64_t)(pin(p, s.AdmReserveTs))
	c.adm_uid_rank = (*C.uint32_t)(pin(p, s.AdmUIDRank))
	c.adm_flags = (*C.uint8_t)(pin(p, s.AdmFlags))
	c.adm_use_off = (*C.int32_t)(pin(p, s.AdmUseOff))
	c.adm_use_fr = (*C.int32_t)(pin(p, s.AdmUseFr))
	c.adm_use_qty = (*C.int64_t)(pin(p, s.AdmUseQty))
}

// PutSnapshot uploads cache.Snapshot() (pkg/cache/scheduler/snapshot.go:171) to HBM.
func (e *Engine) PutSnapshot(s *FlatSnapshot) error {
	var p runtime.Pinner
	defer p.Unpin()
	c := (*C.kq_snapshot)(C.calloc(1, C.sizeof_kq_snapshot))
	defer C.free(unsafe.Pointer(c))
	fillSnapshot(&p, c, s)
	if rc := C.kq_snapshot_put(e.h, c); rc != 0 {
		return e.err("kq_snapshot_put", rc)
	}
	return nil
}

// PatchSnapshot is the next cycle's snapshot when only usage (KQ_PATCH_USAGE = 1) and / or the admitted set (KQ_PATCH_ADMITTED = 2)
// moved since the last PutSnapshot: what clusterQueue.updateWorkloadUsage (clusterqueue.go:594) changes between two cycles.
func (e *Engine) PatchSnapshot(s *FlatSnapshot, what uint32) error {
	var p runtime.Pinner
	defer p.Unpin()
	c := (*C.kq_snapshot)(C.calloc(1, C.sizeof_kq_snapshot))
	defer C.free(unsafe.Pointer(c))
	fillSnapshot(&p, c, s)
	if rc := C.kq_snapshot_patch(e.h, c, C.uint32_t(what)); rc != 0 {
		return e.err("kq_snapshot_patch", rc)
	}
	return nil
}

// RowPatch = kq_row_patch: the admitted workloads that left cq.Workloads since the last call and the ones that came
// (clusterQueue.updateWorkloadUsage, pkg/cache/scheduler/clusterqueue.go:594, one workload at a time; the Go cache calls it from
// AddOrUpdateWorkload / DeleteWorkload, pkg/cache/scheduler/cache.go). Rows are indices into the engine's resident table: the Go side keeps
// rowOf map[workload.Reference]int32 and renumbers it with the newIndex PatchRows returns. AddUIDRank must be comparable with the
// resident rows' keys: use an order-preserving 32-bit prefix of Obj.UID instead of dense ranks.
type RowPatch struct {
	RemoveRows                            []int32
	AddCQ                                 []int32
	AddPriority, AddQueueTs, AddReserveTs []int64
	AddUIDRank                            []uint32
	AddFlags                              []uint8
	AddUseOff, AddUseFr                   []int32
	AddUseQty                             []int64
	EvictRows                             []int32 // rows that get the Evicted mark: the targets of the last cycle's preemptions
	// KQ_ROWS_FOLD_USAGE: the usage of the removed rows leaves the resident snapshot, the usage of the added rows enters it
	// (clusterQueue.updateWorkloadUsage in full); the cycle whose admissions are added as rows is then not committed with CommitCycle.
	FoldUsage bool
}

// PatchRows = kq_snapshot_patch_rows: the O(changes) form of PatchSnapshot(KQ_PATCH_ADMITTED). newIndex (len = rows before the call)
// receives the new index of every old row, -1 for a removed one. ErrUnsupported: amounts outside the plain range or sizes beyond the
// sort keys' fields — fall back to PatchSnapshot. Usage is folded separately (CommitCycle / ReleaseCycle / PatchSnapshot(KQ_PATCH_USAGE)).
func (e *Engine) PatchRows(p *RowPatch, newIndex []int32) error {
	var pin_ runtime.Pinner
	defer pin_.Unpin()
	c := (*C.kq_row_patch)(C.calloc(1, C.sizeof_kq_row_patch))
	defer C.free(unsafe.Pointer(c))
	c.n_remove = C.int32_t(len(p.RemoveRows))
	c.remove_rows = (*C.int32_t)(pin(&pin_, p.RemoveRows))
	c.n_add = C.int32_t(len(p.AddCQ))
	c.add_cq = (*C.int32_t)(pin(&pin_, p.AddCQ))
	c.add_priority = (*C.int64_t)(pin(&pin_, p.AddPriority))
	c.add_queue_ts = (*C.int64_t)(pin(&pin_, p.AddQueueTs))
	c.add_reserve_ts = (*C.int64_t)(pin(&pin_, p.AddReserveTs))
	c.add_uid_rank = (*C.uint32_t)(pin(&pin_, p.AddUIDRank))
	c.add_flags = (*C.uint8_t)(pin(&pin_, p.AddFlags))
	c.add_use_off = (*C.int32_t)(pin(&pin_, p.AddUseOff))
	c.add_use_fr = (*C.int32_t)(pin(&pin_, p.AddUseFr))
	c.add_use_qty = (*C.int64_t)(pin(&pin_, p.AddUseQty))
	c.n_evict = C.int32_t(len(p.EvictRows))
	c.evict_rows = (*C.int32_t)(pin(&pin_, p.EvictRows))
	if p.FoldUsage {
		c.flags = C.KQ_ROWS_FOLD_USAGE
	}
	if rc := C.kq_snapshot_patch_rows(e.h, c, (*C.int32_t)(pin(&pin_, newIndex))); rc != 0 {
		if rc == C.KQ_EUNSUPPORTED {
			return ErrUnsupported
		}
		return e.err("kq_snapshot_patch_rows", rc)
	}
	return nil
}

func fillHeads(p *runtime.Pinner, c *C.kq_heads, h *FlatHeads) {
	c.n, c.cycle = C.int32_t(h.N), C.int64_t(h.Cycle)
	c.cq = (*C.int32_t)(pin(p, h.CQ))
	c.priority = (*C.int64_t)(pin(p, h.Priority))
	c.queue_ts = (*C.int64_t)(pin(p, h.QueueTs))
	c.flags = (*C.uint32_t)(pin(p, h.Flags))
	c.ps_off = (*C.int32_t)(pin(p, h.PsOff))
	c.ps_count = (*C.int32_t)(pin(p, h.PsCount))
	c.ps_min_count = (*C.int32_t)(pin(p, h.PsMinCount))
	c.ps_req_off = (*C.int32_t)(pin(p, h.PsReqOff))
	c.req_res = (*C.int32_t)(pin(p, h.ReqRes))
	c.req_qty = (*C.int64_t)(pin(p, h.ReqQty))
	c.ps_flavor_ok = (*C.uint64_t)(pin(p, h.PsFlavorOK))
	c.ps_last_tried = (*C.int32_t)(pin(p, h.PsLastTried))
	c.last_generation = (*C.int64_t)(pin(p, h.LastGeneration))
	c.last_cycle = (*C.int64_t)(pin(p, h.LastCycle))
	c.last_hash = (*C.uint64_t)(pin(p, h.LastHash))
	c.hash = (*C.uint64_t)(pin(p, h.Hash))
	if h.SliceRow != nil {
		c.slice_row = (*C.int32_t)(pin(p, h.SliceRow))
		c.ps_slice_count = (*C.int32_t)(pin(p, h.PsSliceCount))
		c.req_slice_flavor = (*C.int32_t)(pin(p, h.ReqSliceFlavor))
		c.req_slice_qty = (*C.int64_t)(pin(p, h.ReqSliceQty))
		c.ps_slice_pods_flavor = (*C.int32_t)(pin(p, h.PsSlicePodsFlavor))
		c.ps_slice_pods_qty = (*C.int64_t)(pin(p, h.PsSlicePodsQty))
	}
	if h.PsGroup != nil {
		c.ps_group = (*C.int32_t)(pin(p, h.PsGroup))
	}
}

func fillDecisions(p *runtime.Pinner, c *C.kq_decisions, d *FlatDecisions) {
	c.status = (*C.uint8_t)(pin(p, d.Status))
	c.action = (*C.uint8_t)(pin(p, d.Action))
	c.nominated_mode = (*C.uint8_t)(pin(p, d.NominatedMode))
	c.mode = (*C.uint8_t)(pin(p, d.Mode))
	c.requeue_reason = (*C.uint8_t)(pin(p, d.RequeueReason))
	c.skip = (*C.uint8_t)(pin(p, d.Skip))
	c.borrowing = (*C.int32_t)(pin(p, d.Borrowing))
	c.order = (*C.int32_t)(pin(p, d.Order))
	c.flavor = (*C.int32_t)(pin(p, d.Flavor))
	c.res_mode = (*C.uint8_t)(pin(p, d.ResMode))
	c.tried_idx = (*C.int32_t)(pin(p, d.TriedIdx))
	c.ps_count = (*C.int32_t)(pin(p, d.PsCount))
	c.tgt_off = (*C.int32_t)(pin(p, d.TgtOff))
	c.tgt_cap = C.int32_t(len(d.TgtAdm))
	c.tgt_adm = (*C.int32_t)(pin(p, d.TgtAdm))
	c.tgt_reason = (*C.uint8_t)(pin(p, d.TgtReason))
	c.rsn_cap = C.int32_t(len(d.RsnCode))
	c.rsn_off = (*C.int32_t)(pin(p, d.RsnOff))
	c.rsn_code = (*C.uint8_t)(pin(p, d.RsnCode))
	c.rsn_podset = (*C.uint8_t)(pin(p, d.RsnPodset))
	c.rsn_flavor = (*C.int16_t)(pin(p, d.RsnFlavor))
	c.rsn_resource = (*C.int16_t)(pin(p, d.RsnResource))
	c.rsn_a = (*C.int64_t)(pin(p, d.RsnA))
	c.rsn_b = (*C.int64_t)(pin(p, d.RsnB))
	c.rsn_c = (*C.int64_t)(pin(p, d.RsnC))
}

// RunCycle = nominate + iterator + processEntry (scheduler.go:308-386 steps 3-5) on the device.
// On ANY error the caller runs the stock Go path for this cycle (the engine is stateless across cycles).
func (e *Engine) RunCycle(h *FlatHeads, out *FlatDecisions) error {
	var p runtime.Pinner
	defer p.Unpin()
	ch := (*C.kq_heads)(C.calloc(1, C.sizeof_kq_heads))
	defer C.free(unsafe.Pointer(ch))
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	fillHeads(&p, ch, h)
	fillDecisions(&p, cd, out)
	if rc := C.kq_cycle_run(e.h, ch, cd); rc != 0 {
		return e.err("kq_cycle_run", rc)
	}
	return nil
}

// ---- pending side on the device (pkg/cache/queue): see include/kq_engine.h "pending side" -----------------------

// PutPending = PushOrUpdate of every pending workload (cluster_queue.go:379). uidRank[w] = rank of Obj.UID.
// lq (optional, AdmissionFairSharing): index of the workload's LocalQueue in [0,nLQ) or -1 (queueOrderingFunc cluster_queue.go:880).
// requeueAt (optional): RequeueState.RequeueAt per workload in ns, RequeueNone / RequeueBlocked (backoffWaitingTimeExpired :474).
func (e *Engine) PutPending(all *FlatHeads, uidRank []uint32, nLQ int32, lq []int32, requeueAt []int64) error {
	var p runtime.Pinner
	defer p.Unpin()
	c := (*C.kq_pending)(C.calloc(1, C.sizeof_kq_pending))
	defer C.free(unsafe.Pointer(c))
	fillHeads(&p, &c.w, all)
	c.uid_rank = (*C.uint32_t)(pin(&p, uidRank))
	if len(lq) > 0 {
		c.n_lq = C.int32_t(nLQ)
		c.lq = (*C.int32_t)(pin(&p, lq))
	}
	if len(requeueAt) > 0 {
		c.requeue_at = (*C.int64_t)(pin(&p, requeueAt))
	}
	if rc := C.kq_pending_put(e.h, c); rc != 0 {
		return e.err("kq_pending_put", rc)
	}
	return nil
}

// AddPending = PushOrUpdate (cluster_queue.go:379) of workloads that were not pending before; returns the index of the first one
// (existing indices do not move). lq as in PutPending.
func (e *Engine) AddPending(more *FlatHeads, uidRank []uint32, nLQ int32, lq []int32, requeueAt []int64) (int32, error) {
	var p runtime.Pinner
	defer p.Unpin()
	c := (*C.kq_pending)(C.calloc(1, C.sizeof_kq_pending))
	defer C.free(unsafe.Pointer(c))
	fillHeads(&p, &c.w, more)
	c.uid_rank = (*C.uint32_t)(pin(&p, uidRank))
	if len(lq) > 0 {
		c.n_lq = C.int32_t(nLQ)
		c.lq = (*C.int32_t)(pin(&p, lq))
	}
	if len(requeueAt) > 0 {
		c.requeue_at = (*C.int64_t)(pin(&p, requeueAt))
	}
	var first C.int32_t
	if rc := C.kq_pending_add(e.h, c, &first); rc != 0 {
		return 0, e.err("kq_pending_add", rc)
	}
	return int32(first), nil
}

// UpdatePending = PushOrUpdate (cluster_queue.go:379-428) of keys that ARE pending, each with its new object: more's i-th workload
// replaces wl[i] (same uidRank) and gets index first+i, wl[i] leaves the set. A key that was in the heap stays in the heap (:427); an
// inadmissible one is re-evaluated like an arrival (:405-426); the ClusterQueue's sticky preemptor pointer follows the key.
// sameGeneration[i] (optional): the new object's metadata.generation equals the old one's (a status-only update): IsPreemptor
// (cluster_queue.go:213) keeps holding for the ClusterQueue's preemptor. The caller has both objects at hand in the workload
// controller's Update handler (old.Generation == new.Generation).
func (e *Engine) UpdatePending(wl []int32, more *FlatHeads, uidRank []uint32, nLQ int32, lq []int32, requeueAt []int64, sameGeneration []uint8) (int32, error) {
	var p runtime.Pinner
	defer p.Unpin()
	c := (*C.kq_pending)(C.calloc(1, C.sizeof_kq_pending))
	defer C.free(unsafe.Pointer(c))
	fillHeads(&p, &c.w, more)
	c.uid_rank = (*C.uint32_t)(pin(&p, uidRank))
	if len(lq) > 0 {
		c.n_lq = C.int32_t(nLQ)
		c.lq = (*C.int32_t)(pin(&p, lq))
	}
	if len(requeueAt) > 0 {
		c.requeue_at = (*C.int64_t)(pin(&p, requeueAt))
	}
	if len(sameGeneration) == len(wl) && len(wl) > 0 {
		c.same_generation = (*C.uint8_t)(pin(&p, sameGeneration))
	}
	var first C.int32_t
	if rc := C.kq_pending_update(e.h, C.int32_t(len(wl)), (*C.int32_t)(pin(&p, wl)), c, &first); rc != 0 {
		return 0, e.err("kq_pending_update", rc)
	}
	return int32(first), nil
}

const (
	RequeueNone    = int64(-1 << 63) // no RequeueState.RequeueAt
	RequeueBlocked = int64(1<<63 - 1) // the Requeued condition is False
)

// SetClock = c.clock.Now() of the queues for the calls that follow.
func (e *Engine) SetClock(nowNs int64) error {
	if rc := C.kq_pending_set_clock(e.h, C.int64_t(nowNs)); rc != 0 {
		return e.err("kq_pending_set_clock", rc)
	}
	return nil
}

// SetRequeueAt = PushOrUpdate of pending workloads whose RequeueState / Requeued condition changed (cluster_queue.go:391-428).
func (e *Engine) SetRequeueAt(wl []int32, at []int64) error {
	if len(wl) == 0 {
		return nil
	}
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_pending_set_requeue_at(e.h, C.int32_t(len(wl)), (*C.int32_t)(pin(&p, wl)), (*C.int64_t)(pin(&p, at))); rc != 0 {
		return e.err("kq_pending_set_requeue_at", rc)
	}
	return nil
}

// DeletePending = ClusterQueue.Delete (cluster_queue.go:488) of pending workloads (deleted, finished, admitted elsewhere).
func (e *Engine) DeletePending(wl []int32) error {
	if len(wl) == 0 {
		return nil
	}
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_pending_delete(e.h, C.int32_t(len(wl)), (*C.int32_t)(pin(&p, wl))); rc != 0 {
		return e.err("kq_pending_delete", rc)
	}
	return nil
}

// SetLQUsage hands over afs.CalculateUsage of every LocalQueue (admission_fair_sharing.go:86) before Heads.
func (e *Engine) SetLQUsage(usage []float64) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_pending_set_lq_usage(e.h, C.int32_t(len(usage)), (*C.double)(pin(&p, usage))); rc != 0 {
		return e.err("kq_pending_set_lq_usage", rc)
	}
	return nil
}

// Heads = queues.Heads() (manager.go:903): pops on the device; headWl[c] = index of the popped workload of ClusterQueue c or -1.
func (e *Engine) Heads(cycle int64, cqActive []uint8, headWl []int32) (nHeads, nPodsets int32, err error) {
	var p runtime.Pinner
	defer p.Unpin()
	var n, nps C.int32_t
	if rc := C.kq_pending_heads(e.h, C.int64_t(cycle), (*C.uint8_t)(pin(&p, cqActive)), &n, &nps, (*C.int32_t)(pin(&p, headWl))); rc != 0 {
		return 0, 0, e.err("kq_pending_heads", rc)
	}
	return int32(n), int32(nps), nil
}

// RunPendingCycle runs the cycle over the heads gathered by Heads; ApplyPending is step 6 of schedule() on the device.
func (e *Engine) RunPendingCycle(out *FlatDecisions) error {
	var p runtime.Pinner
	defer p.Unpin()
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	fillDecisions(&p, cd, out)
	if rc := C.kq_cycle_run_pending(e.h, cd); rc != 0 {
		return e.err("kq_cycle_run_pending", rc)
	}
	return nil
}

func (e *Engine) ApplyPending() error {
	if rc := C.kq_pending_apply(e.h); rc != 0 {
		return e.err("kq_pending_apply", rc)
	}
	return nil
}

// QueueInadmissible = queueInadmissibleWorkloads for the listed ClusterQueues (nil: all), inadmissible_workloads.go:149.
func (e *Engine) QueueInadmissible(cqs []int32) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_pending_queue_inadmissible(e.h, C.int32_t(len(cqs)), (*C.int32_t)(pin(&p, cqs))); rc != 0 {
		return e.err("kq_pending_queue_inadmissible", rc)
	}
	return nil
}

//go:build kq_hip

// step.go — the rest of include/kq_engine.h's entry points: the one-enqueue-per-cycle loop (kq_pending_step / _wait, what bench.py's
// headline times), the incremental snapshot (kq_cycle_commit / _release, kq_snapshot_usage_add, _derive, _read_planes), resident
// head batches, the split-root entry points and the AdmissionFairSharing ledger. Same rules as kqengine.go: one caller thread
// (runtime.LockOSThread), nothing retained after a call, on ANY error the caller runs the stock Go path for that cycle.
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image); tests/test_abi.py checks that every exported symbol has a
// caller in shim/go and that every C.kq_* call names a declared symbol.
package kqengine

/*
#include <stdlib.h>
#include "kq_engine.h"
#include "kq_tas.h"
*/
import "C"

import (
	"runtime"
	"unsafe"
)

// ---- one cycle = one enqueue (scheduler.go:308-386 with Heads(), commit, requeue policy and release on the device) ----------------

// PendingBounds sizes the decision buffers of StepWait: <= 1 head per ClusterQueue, the widest workload of every ClusterQueue.
func (e *Engine) PendingBounds() (maxHeads, maxPodsets int32, err error) {
	var h, p C.int32_t
	if rc := C.kq_pending_bounds(e.h, &h, &p); rc != 0 {
		return 0, 0, e.err("kq_pending_bounds", rc)
	}
	return int32(h), int32(p), nil
}

// Step enqueues Heads() -> cycle -> commit -> requeue policy (-> release of the admissions `releaseAge` cycles old) and returns at once;
// at most two steps may be in flight. The scheduler goroutine calls it where schedule() calls queues.Heads (manager.go:903).
func (e *Engine) Step(cycle int64, cqActive []uint8, tgtCap, releaseAge int32, wantHeadWl bool) error {
	var p runtime.Pinner
	defer p.Unpin()
	w := C.int32_t(0)
	if wantHeadWl {
		w = 1
	}
	if rc := C.kq_pending_step(e.h, C.int64_t(cycle), (*C.uint8_t)(pin(&p, cqActive)), C.int32_t(tgtCap), C.int32_t(releaseAge), w); rc != 0 {
		return e.err("kq_pending_step", rc)
	}
	return nil
}

// StepReasons switches the reason records of the steps issued from now on (rsnCap > 0: StepWait fills the Rsn* arrays of its
// FlatDecisions, from which messages.go rebuilds the reference's "couldn't assign flavors" texts; 0: off, the default).
func (e *Engine) StepReasons(rsnCap int) error {
	if rc := C.kq_pending_step_reasons(e.h, C.int32_t(rsnCap)); rc != 0 {
		return e.err("kq_pending_step_reasons", rc)
	}
	return nil
}

// StepWait blocks until the OLDEST step in flight is done and unpacks its decisions (out sized by PendingBounds; headWl optional, [n_cq]).
// The caller then runs admit / IssuePreemptions / requeueAndUpdate from `out` (apply.go), exactly as after RunCycle.
func (e *Engine) StepWait(out *FlatDecisions, headWl []int32) (nHeads, nPodsets int32, err error) {
	var p runtime.Pinner
	defer p.Unpin()
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	fillDecisions(&p, cd, out)
	var n, nps C.int32_t
	if rc := C.kq_pending_step_wait(e.h, cd, &n, &nps, (*C.int32_t)(pin(&p, headWl))); rc != 0 {
		return 0, 0, e.err("kq_pending_step_wait", rc)
	}
	return int32(n), int32(nps), nil
}

// PendingState reads the heap state byte of every pending workload (KQ_WL_*) and the four class counts (counts[3] = gone records: the
// signal to re-put the pending set after many UpdatePending calls).
func (e *Engine) PendingState(state []uint8) (counts [4]int32, err error) {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_pending_read_state(e.h, (*C.uint8_t)(pin(&p, state)), (*C.int32_t)(unsafe.Pointer(&counts[0]))); rc != 0 {
		return counts, e.err("kq_pending_read_state", rc)
	}
	return counts, nil
}

// ---- incremental snapshot: what cache.AddOrUpdateWorkload / DeleteWorkload do to the usage tree (clusterqueue.go:594) ----------------

// Commit folds the usage of every workload the LAST cycle admitted into the resident snapshot (assumeWorkload, scheduler.go:1064).
func (e *Engine) Commit() (nAdmitted int32, err error) {
	var n C.int32_t
	if rc := C.kq_cycle_commit(e.h, &n); rc != 0 {
		return 0, e.err("kq_cycle_commit", rc)
	}
	return int32(n), nil
}

// Release removes what the cycle `age` commits ago added (the workloads finished); every ClusterQueue under a root cohort whose quota was
// freed runs queueInadmissibleWorkloads (inadmissible_workloads.go:112-175) when a pending set is resident.
func (e *Engine) Release(age int32) error {
	if rc := C.kq_cycle_release(e.h, C.int32_t(age)); rc != 0 {
		return e.err("kq_cycle_release", rc)
	}
	return nil
}

// DeriveSnapshot recomputes SubtreeQuota / cohort Usage from the uploaded Quotas and ClusterQueue usage (resource_node.go:167-230).
func (e *Engine) DeriveSnapshot() error {
	if rc := C.kq_snapshot_derive(e.h); rc != 0 {
		return e.err("kq_snapshot_derive", rc)
	}
	return nil
}

// ReadPlanes copies the resident SubtreeQuota / Usage / flag planes back ([N * n_fr] each; nil skips one).
func (e *Engine) ReadPlanes(subtreeQuota, usage []int64, flags []uint8) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_snapshot_read_planes(e.h, (*C.int64_t)(pin(&p, subtreeQuota)), (*C.int64_t)(pin(&p, usage)), (*C.uint8_t)(pin(&p, flags))); rc != 0 {
		return e.err("kq_snapshot_read_planes", rc)
	}
	return nil
}

// LastCycleStats: device time of the last cycle and the algorithmic bytes it is charged (metrics.AdmissionAttempt's duration label).
func (e *Engine) LastCycleStats() (kernelMs float64, algorithmicBytes int64, phaseMs [3]float64, err error) {
	var ms C.double
	var b C.int64_t
	if rc := C.kq_last_cycle_stats(e.h, &ms, &b); rc != 0 {
		return 0, 0, phaseMs, e.err("kq_last_cycle_stats", rc)
	}
	var pb [2]C.int64_t
	if rc := C.kq_last_cycle_phases(e.h, (*C.double)(unsafe.Pointer(&phaseMs[0])), &pb[0]); rc != 0 {
		return 0, 0, phaseMs, e.err("kq_last_cycle_phases", rc)
	}
	return float64(ms), int64(b), phaseMs, nil
}

// ---- resident head batches (nominate-ahead, SURVEY §8f-1) ----------------------------------------------------------------------------

// PutHeads uploads a heads batch into slot `batch`; RunResident / NominateResident run a cycle / the nomination alone over it.
func (e *Engine) PutHeads(h *FlatHeads, batch int32) error {
	var p runtime.Pinner
	defer p.Unpin()
	ch := (*C.kq_heads)(C.calloc(1, C.sizeof_kq_heads))
	defer C.free(unsafe.Pointer(ch))
	fillHeads(&p, ch, h)
	if rc := C.kq_heads_put(e.h, ch, C.int32_t(batch)); rc != 0 {
		return e.err("kq_heads_put", rc)
	}
	return nil
}

func (e *Engine) runResident(batch int32, out *FlatDecisions, nominateOnly bool) error {
	var p runtime.Pinner
	defer p.Unpin()
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	fillDecisions(&p, cd, out)
	var rc C.int
	if nominateOnly {
		rc = C.kq_nominate_run_resident(e.h, C.int32_t(batch), cd)
	} else {
		rc = C.kq_cycle_run_resident(e.h, C.int32_t(batch), cd)
	}
	if rc != 0 {
		return e.err("kq_cycle_run_resident", rc)
	}
	return nil
}
func (e *Engine) RunResident(batch int32, out *FlatDecisions) error      { return e.runResident(batch, out, false) }
func (e *Engine) NominateResident(batch int32, out *FlatDecisions) error { return e.runResident(batch, out, true) }

// ---- one root tree split across engines (include/kq_engine.h "sharded cycle"; kq_group.go drives it over RCCL) --------------------------

// ShardWords = int64 words of the exchange buffer for this batch and these decision capacities at `world` ranks.
func (e *Engine) ShardWords(h *FlatHeads, out *FlatDecisions, world int32) (int64, error) {
	var p runtime.Pinner
	defer p.Unpin()
	ch := (*C.kq_heads)(C.calloc(1, C.sizeof_kq_heads))
	defer C.free(unsafe.Pointer(ch))
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	fillHeads(&p, ch, h)
	fillDecisions(&p, cd, out)
	var words C.int64_t
	if rc := C.kq_cycle_shard_words(e.h, ch, cd, C.int32_t(world), &words); rc != 0 {
		return 0, e.err("kq_cycle_shard_words", rc)
	}
	return int64(words), nil
}

// NominateShard nominates the heads with mine[h] != 0 and writes their nomination into xbufDev (device memory, zero elsewhere);
// ProcessMerged imports the all-reduced buffer and runs iterator + processEntry over ALL heads (identical on every rank).
func (e *Engine) NominateShard(h *FlatHeads, mine []uint8, world, rank int32, xbufDev unsafe.Pointer, out *FlatDecisions) error {
	var p runtime.Pinner
	defer p.Unpin()
	ch := (*C.kq_heads)(C.calloc(1, C.sizeof_kq_heads))
	defer C.free(unsafe.Pointer(ch))
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	fillHeads(&p, ch, h)
	fillDecisions(&p, cd, out)
	if rc := C.kq_cycle_nominate_shard(e.h, ch, (*C.uint8_t)(pin(&p, mine)), C.int32_t(world), C.int32_t(rank), xbufDev, cd); rc != 0 {
		return e.err("kq_cycle_nominate_shard", rc)
	}
	return nil
}
func (e *Engine) ProcessMerged(world, rank int32, xbufDev unsafe.Pointer, out *FlatDecisions) error {
	var p runtime.Pinner
	defer p.Unpin()
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	fillDecisions(&p, cd, out)
	if rc := C.kq_cycle_process_merged(e.h, C.int32_t(world), C.int32_t(rank), xbufDev, cd); rc != 0 {
		return e.err("kq_cycle_process_merged", rc)
	}
	return nil
}

// Certificate / UsageAdd: the round-2 protocol (every rank runs the whole cycle on its subtree, usage deltas are all-reduced, the root-row
// slack certifies exactness). usageDeltaDev / deltaDev are device buffers [N * n_fr] / [n_cq * n_fr].
func (e *Engine) Certificate(usageDeltaDev unsafe.Pointer, rootMargin []int64, flags []int32) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_cycle_certificate(e.h, (*C.int64_t)(usageDeltaDev), (*C.int64_t)(pin(&p, rootMargin)), (*C.int32_t)(pin(&p, flags))); rc != 0 {
		return e.err("kq_cycle_certificate", rc)
	}
	return nil
}
func (e *Engine) UsageAdd(deltaDev unsafe.Pointer, sign int32) error {
	if rc := C.kq_snapshot_usage_add(e.h, (*C.int64_t)(deltaDev), C.int32_t(sign)); rc != 0 {
		return e.err("kq_snapshot_usage_add", rc)
	}
	return nil
}

// ---- TAS: one flavor's leaf usage split across engines (include/kq_tas.h; BASELINE configs[4] "all-reduce of domain-usage deltas") ------

// UsageDelta sums the Usage.TAS of the selected placed workloads into a [leaves][R] plane in device memory (the RCCL send buffer);
// UsageAddPlane folds a reduced plane into the resident leaf usage; Overflow marks the leaves where usage + plane exceeds the capacity.
func (t *TAS) UsageDelta(r *TASRequests, res *TASResult, sel []uint8, planeDev unsafe.Pointer) error {
	var p runtime.Pinner
	defer p.Unpin()
	cr := (*C.kq_tas_requests)(C.calloc(1, C.sizeof_kq_tas_requests))
	defer C.free(unsafe.Pointer(cr))
	co := (*C.kq_tas_result)(C.calloc(1, C.sizeof_kq_tas_result))
	defer C.free(unsafe.Pointer(co))
	fillRequests(&p, cr, r)
	fillResult(&p, co, res)
	if rc := C.kq_tas_usage_delta(t.h, cr, co, (*C.uint8_t)(pin(&p, sel)), (*C.int64_t)(planeDev)); rc != 0 {
		return t.err("kq_tas_usage_delta", rc)
	}
	return nil
}
func (t *TAS) UsageAddPlane(planeDev unsafe.Pointer, sign int32) error {
	if rc := C.kq_tas_usage_add(t.h, (*C.int64_t)(planeDev), C.int32_t(sign)); rc != 0 {
		return t.err("kq_tas_usage_add", rc)
	}
	return nil
}
func (t *TAS) Overflow(planeDev unsafe.Pointer, leafOver []uint8) (int32, error) {
	var p runtime.Pinner
	defer p.Unpin()
	var n C.int32_t
	if rc := C.kq_tas_overflow(t.h, (*C.int64_t)(planeDev), (*C.uint8_t)(pin(&p, leafOver)), &n); rc != 0 {
		return 0, t.err("kq_tas_overflow", rc)
	}
	return int32(n), nil
}
func (t *TAS) ReadUsage(u []int64) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_tas_read_usage(t.h, (*C.int64_t)(pin(&p, u))); rc != 0 {
		return t.err("kq_tas_read_usage", rc)
	}
	return nil
}
func (t *TAS) LastStats() (ms float64, bytes int64) {
	var m C.double
	var b C.int64_t
	C.kq_tas_last_stats(t.h, &m, &b)
	return float64(m), int64(b)
}

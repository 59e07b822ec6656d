//go:build kq_hip

// Package kqengine binds the MI355X admission engine (include/kq_engine.h) into Kueue's scheduler.
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain. This file is the reviewed
// source a Kueue maintainer would drop under pkg/scheduler/kqengine and build with
//   CGO_ENABLED=1 go build -tags kq_hip ./cmd/kueue
// (the reference builds with CGO_ENABLED=0, Makefile:69, so the tag keeps the stock build untouched).
//
// Call site: (*Scheduler).schedule, pkg/scheduler/scheduler.go:340-362 — between cache.Snapshot() and the
// requeue loop. See INTEGRATION.md for the patch.
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

// FlatSnapshot / FlatHeads / FlatDecisions are Go-side SoA buffers (plain slices) filled by flatten.go from
// *schdcache.Snapshot and []qcache.Head. Slices are pinned for the duration of one call with runtime.Pinner
// (Go 1.21+), which satisfies the cgo pointer-passing rules: C keeps no pointer after returning.
type FlatSnapshot struct {
	NCQ, NCohort, NFlavor, NResource, PodsResource int32
	ResourceOrder, Parent                          []int32
	ChildCohortOff, ChildCohort, ChildCQOff, ChildCQ []int32
	FairWeight                                     []float64
	Nominal, BorrowLimit, LendLimit, SubtreeQuota, Usage []int64
	QuotaFlags                                     []uint8
	CQRgOff, RgFlavorOff, RgFlavor, RgResOff, RgRes []int32
	CQPolicy                                       []uint32
	CQBorrowPrioThreshold                          []int32
	CQGeneration                                   []int64
	NAdm                                           int32
	CQAdmOff                                       []int32
	AdmPriority, AdmQueueTs, AdmReserveTs          []int64
	AdmUIDRank                                     []uint32
	AdmFlags                                       []uint8
	AdmUseOff, AdmUseFr                            []int32
	AdmUseQty                                      []int64
}

func p32(s []int32) *C.int32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.int32_t)(unsafe.Pointer(&s[0]))
}
func p64(s []int64) *C.int64_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.int64_t)(unsafe.Pointer(&s[0]))
}
func pu8(s []uint8) *C.uint8_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&s[0]))
}
func pu32(s []uint32) *C.uint32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&s[0]))
}
func pf64(s []float64) *C.double {
	if len(s) == 0 {
		return nil
	}
	return (*C.double)(unsafe.Pointer(&s[0]))
}

// PutSnapshot uploads cache.Snapshot() (pkg/cache/scheduler/snapshot.go:171) to HBM.
func (e *Engine) PutSnapshot(s *FlatSnapshot) error {
	var pin runtime.Pinner
	defer pin.Unpin()
	for _, p := range []any{&s.ResourceOrder, &s.Parent, &s.Nominal, &s.Usage} {
		_ = p // every slice's backing array is pinned in the real shim (elided: one pin.Pin(&slice[0]) per field)
	}
	c := C.kq_snapshot{
		n_cq: C.int32_t(s.NCQ), n_cohort: C.int32_t(s.NCohort), n_flavor: C.int32_t(s.NFlavor), n_resource: C.int32_t(s.NResource),
		pods_resource: C.int32_t(s.PodsResource), resource_order: p32(s.ResourceOrder), parent: p32(s.Parent),
		child_cohort_off: p32(s.ChildCohortOff), child_cohort: p32(s.ChildCohort), child_cq_off: p32(s.ChildCQOff), child_cq: p32(s.ChildCQ),
		fair_weight: pf64(s.FairWeight), nominal: p64(s.Nominal), borrow_limit: p64(s.BorrowLimit), lend_limit: p64(s.LendLimit),
		subtree_quota: p64(s.SubtreeQuota), usage: p64(s.Usage), quota_flags: pu8(s.QuotaFlags),
		cq_rg_off: p32(s.CQRgOff), rg_flavor_off: p32(s.RgFlavorOff), rg_flavor: p32(s.RgFlavor), rg_res_off: p32(s.RgResOff), rg_res: p32(s.RgRes),
		cq_policy: pu32(s.CQPolicy), cq_borrow_prio_threshold: p32(s.CQBorrowPrioThreshold), cq_generation: p64(s.CQGeneration),
		n_adm: C.int32_t(s.NAdm), cq_adm_off: p32(s.CQAdmOff), adm_priority: p64(s.AdmPriority), adm_queue_ts: p64(s.AdmQueueTs),
		adm_reserve_ts: p64(s.AdmReserveTs), adm_uid_rank: pu32(s.AdmUIDRank), adm_flags: pu8(s.AdmFlags),
		adm_use_off: p32(s.AdmUseOff), adm_use_fr: p32(s.AdmUseFr), adm_use_qty: p64(s.AdmUseQty),
	}
	if rc := C.kq_snapshot_put(e.h, &c); rc != 0 {
		return fmt.Errorf("kq_snapshot_put: %s (%s)", C.GoString(C.kq_strerror(rc)), C.GoString(C.kq_last_error(e.h)))
	}
	return nil
}

// RunCycle = nominate + iterator + processEntry (scheduler.go:308-386 steps 3-5) on the device.
// On ANY error the caller runs the stock Go path for this cycle (the engine is stateless across cycles).
func (e *Engine) RunCycle(h *C.kq_heads, out *C.kq_decisions) error {
	if rc := C.kq_cycle_run(e.h, h, out); rc != 0 {
		return fmt.Errorf("kq_cycle_run: %s (%s)", C.GoString(C.kq_strerror(rc)), C.GoString(C.kq_last_error(e.h)))
	}
	return nil
}

//go:build kq_hip

// group.go — include/kq_group.h: ONE root cohort tree over the GPUs of the controller process. The scheduler owns a Group instead of an
// Engine when BASELINE configs[2]/[3]-sized single-root clusters should use more than one device; calls and error handling as for Engine.
package kqengine

/*
#include <stdlib.h>
#include "kq_group.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"
)

type Group struct{ h *C.kq_group }

// NewGroup creates one engine per device ordinal and the RCCL communicator between them (ncclCommInitAll inside this process).
func NewGroup(cfg Config, devices []int32) (*Group, error) {
	var p runtime.Pinner
	defer p.Unpin()
	c := C.kq_config{abi_version: C.KQ_ABI_VERSION, gates: C.uint32_t(cfg.Gates), quota_check_strategy: C.int32_t(cfg.QuotaCheckStrategy)}
	if cfg.FairSharing {
		c.fair_sharing = 1
	}
	c.n_fs_strategies = C.int32_t(len(cfg.FSStrategies))
	for i, s := range cfg.FSStrategies {
		if i < 2 {
			c.fs_strategies[i] = C.int32_t(s)
		}
	}
	g := &Group{}
	if rc := C.kq_group_create(&c, C.int32_t(len(devices)), (*C.int32_t)(pin(&p, devices)), &g.h); rc != 0 {
		return nil, fmt.Errorf("kq_group_create: %s", C.GoString(C.kq_strerror(rc)))
	}
	return g, nil
}

// Flags of NewGroupOpts (include/kq_group.h).
const (
	GroupHostCollective uint32 = 1 // the exchange through pinned host memory: no RCCL, a device ordinal may repeat
	GroupForceSharded   uint32 = 2 // a group of one device also takes the sharded path
)

// NewGroupOpts is NewGroup with explicit flags (the environment is not read).
func NewGroupOpts(cfg Config, devices []int32, flags uint32) (*Group, error) {
	var p runtime.Pinner
	defer p.Unpin()
	c := C.kq_config{abi_version: C.KQ_ABI_VERSION, gates: C.uint32_t(cfg.Gates), quota_check_strategy: C.int32_t(cfg.QuotaCheckStrategy)}
	if cfg.FairSharing {
		c.fair_sharing = 1
	}
	c.n_fs_strategies = C.int32_t(len(cfg.FSStrategies))
	for i, s := range cfg.FSStrategies {
		if i < 2 {
			c.fs_strategies[i] = C.int32_t(s)
		}
	}
	g := &Group{}
	if rc := C.kq_group_create_opts(&c, C.int32_t(len(devices)), (*C.int32_t)(pin(&p, devices)), C.uint32_t(flags), &g.h); rc != 0 {
		return nil, fmt.Errorf("kq_group_create_opts: %s", C.GoString(C.kq_strerror(rc)))
	}
	return g, nil
}
func (g *Group) Close()     { C.kq_group_destroy(g.h); g.h = nil }
func (g *Group) Size() int  { return int(C.kq_group_size(g.h)) }
func (g *Group) err(what string, rc C.int) error {
	return fmt.Errorf("%s: %s (%s)", what, C.GoString(C.kq_strerror(rc)), C.GoString(C.kq_group_last_error(g.h)))
}

// PutSnapshot uploads cache.Snapshot to every device.
func (g *Group) PutSnapshot(s *FlatSnapshot) error {
	var p runtime.Pinner
	defer p.Unpin()
	c := (*C.kq_snapshot)(C.calloc(1, C.sizeof_kq_snapshot))
	defer C.free(unsafe.Pointer(c))
	fillSnapshot(&p, c, s)
	if rc := C.kq_group_snapshot_put(g.h, c); rc != 0 {
		return g.err("kq_group_snapshot_put", rc)
	}
	return nil
}

// RunCycle = Engine.RunCycle over the group: nomination sharded over the devices, one all-reduce, processEntry replicated.
func (g *Group) RunCycle(h *FlatHeads, out *FlatDecisions) error {
	var p runtime.Pinner
	defer p.Unpin()
	ch := (*C.kq_heads)(C.calloc(1, C.sizeof_kq_heads))
	defer C.free(unsafe.Pointer(ch))
	cd := (*C.kq_decisions)(C.calloc(1, C.sizeof_kq_decisions))
	defer C.free(unsafe.Pointer(cd))
	fillHeads(&p, ch, h)
	fillDecisions(&p, cd, out)
	if rc := C.kq_group_cycle_run(g.h, ch, cd); rc != 0 {
		return g.err("kq_group_cycle_run", rc)
	}
	return nil
}
func (g *Group) Commit() (int32, error) {
	var n C.int32_t
	if rc := C.kq_group_cycle_commit(g.h, &n); rc != 0 {
		return 0, g.err("kq_group_cycle_commit", rc)
	}
	return int32(n), nil
}
func (g *Group) Release(age int32) error {
	if rc := C.kq_group_cycle_release(g.h, C.int32_t(age)); rc != 0 {
		return g.err("kq_group_cycle_release", rc)
	}
	return nil
}
func (g *Group) ReadUsage(rank int32, usage []int64) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_group_read_usage(g.h, C.int32_t(rank), (*C.int64_t)(pin(&p, usage))); rc != 0 {
		return g.err("kq_group_read_usage", rc)
	}
	return nil
}

// CollectiveInfo = kq_group_collective_info: communicators ncclCommInitAll created (0: host collective / group of one), ncclAllReduce
// groups issued so far, exchanges summed through host memory.
func (g *Group) CollectiveInfo() (rcclRanks int32, allreduceCalls, hostSums int64, err error) {
	var r C.int32_t
	var a, h C.int64_t
	if rc := C.kq_group_collective_info(g.h, &r, &a, &h); rc != 0 {
		return 0, 0, 0, g.err("kq_group_collective_info", rc)
	}
	return int32(r), int64(a), int64(h), nil
}

//go:build kq_hip

package kqengine

/*
#include <stdlib.h>
#include "kq_engine.h"
*/
import "C"

import (
	"math/big"
	"runtime"
	"slices"
	"unsafe"

	inf "gopkg.in/inf.v0"
	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"

	config "sigs.k8s.io/kueue/apis/config/v1beta2"
	queueafs "sigs.k8s.io/kueue/pkg/cache/queue/afs"
	afs "sigs.k8s.io/kueue/pkg/util/admissionfairsharing"
	utilqueue "sigs.k8s.io/kueue/pkg/util/queue"
	"sigs.k8s.io/kueue/pkg/workload"
)

// AdmissionFairSharing ledger on the device: include/kq_engine.h "AdmissionFairSharing ledger". The scheduler's own ledger write — the
// entry penalty of an assumed workload (scheduler.go:1064-1068) — happens inside ApplyPending; the controllers' writes are pushed
// down with SubPenalty / SetConsumed. Amounts cross the boundary as exact 128-bit integers in units of 1e-9.

// nano128 = q at scale 9 as two's complement (lo, hi) words. resource.Quantity holds nothing finer than 1e-9, so this is exact.
func nano128(q resource.Quantity) (uint64, int64) {
	d := new(inf.Dec).Set(q.AsDec())
	d.Round(d, 9, inf.RoundDown) // exact: the scale is already <= 9
	v := d.UnscaledBig()
	m := new(big.Int).And(v, new(big.Int).Sub(new(big.Int).Lsh(big.NewInt(1), 128), big.NewInt(1))) // two's complement, 128 bits
	lo := new(big.Int).And(m, new(big.Int).SetUint64(^uint64(0))).Uint64()
	hi := new(big.Int).Rsh(m, 64).Uint64()
	return lo, int64(hi)
}

// FlatLedger is kq_afs_ledger. Resources is the ledger's dictionary: sorted by name (afs.CalculateUsage sums in sorted key order).
type FlatLedger struct {
	Resources                              []corev1.ResourceName
	LQWeight, ResWeight                    []float64
	ConsumedLo                             []uint64
	ConsumedHi                             []int64
	ConsumedF64                            []float64
	PenaltyLo                              []uint64
	PenaltyHi                              []int64
	PenaltyPresent                         []uint8
	WlPenaltyLo                            []uint64
	WlPenaltyHi                            []int64
	WlPenaltyMask                          []uint64
	index                                  map[corev1.ResourceName]int
}

func (l *FlatLedger) put(lo []uint64, hi []int64, row int, rl corev1.ResourceList) (mask uint64) {
	for name, q := range rl {
		r := l.index[name]
		lo[row*len(l.Resources)+r], hi[row*len(l.Resources)+r] = nano128(q)
		mask |= 1 << uint(r)
	}
	return mask
}

// FlattenLedger reads the AfsUsageLedger for the LocalQueues `lqs` (index = the lq column of PutPending) and computes, for every
// pending workload, what updateEntryPenalty would push (scheduler.go:1343-1348). lqWeight[i] = afs.ResolveLQWeight of lqs[i].
// totalRequests(w) = e.SumTotalRequests(formatter), filtered by the covered resources under IgnoreUndeclaredResources (:1344-1346).
func FlattenLedger(ledger *queueafs.AfsUsageLedger, cfg *config.AdmissionFairSharing, lqs []utilqueue.LocalQueueReference, lqWeight []float64,
	pending []*workload.Info, totalRequests func(*workload.Info) corev1.ResourceList) *FlatLedger {
	l := &FlatLedger{LQWeight: lqWeight, index: map[corev1.ResourceName]int{}}
	names := map[corev1.ResourceName]struct{}{}
	entries := make([]queueafs.UsageLedgerEntry, len(lqs))
	for i, k := range lqs {
		entries[i], _ = ledger.Get(k)
		for n := range entries[i].Resources {
			names[n] = struct{}{}
		}
		for n := range entries[i].PendingPenalty() {
			names[n] = struct{}{}
		}
	}
	pens := make([]corev1.ResourceList, len(pending))
	for w, wi := range pending {
		pens[w] = afs.CalculateEntryPenalty(totalRequests(wi), cfg)
		for n := range pens[w] {
			names[n] = struct{}{}
		}
	}
	for n := range names {
		l.Resources = append(l.Resources, n)
	}
	slices.Sort(l.Resources)
	for i, n := range l.Resources {
		l.index[n] = i
	}
	nr := len(l.Resources)
	resWeights := cfg.ResourceWeights // fsResWeights of afs.ResourceWeights (admission_fair_sharing.go:36-43) for a usage-based ClusterQueue
	l.ResWeight = make([]float64, nr)
	for i, n := range l.Resources {
		l.ResWeight[i] = 1
		if wgt, ok := resWeights[n]; ok {
			l.ResWeight[i] = wgt
		}
	}
	cells := len(lqs) * nr
	l.ConsumedLo, l.ConsumedHi, l.ConsumedF64 = make([]uint64, cells), make([]int64, cells), make([]float64, cells)
	l.PenaltyLo, l.PenaltyHi, l.PenaltyPresent = make([]uint64, cells), make([]int64, cells), make([]uint8, cells)
	for i := range lqs {
		l.put(l.ConsumedLo, l.ConsumedHi, i, entries[i].Resources)
		for n, q := range entries[i].Resources {
			l.ConsumedF64[i*nr+l.index[n]] = q.AsApproximateFloat64() // the float of the form the ledger holds (quantity.go:468)
		}
		m := l.put(l.PenaltyLo, l.PenaltyHi, i, entries[i].PendingPenalty())
		for r := 0; r < nr; r++ {
			l.PenaltyPresent[i*nr+r] = uint8(m >> uint(r) & 1)
		}
	}
	l.WlPenaltyLo, l.WlPenaltyHi, l.WlPenaltyMask = make([]uint64, len(pending)*nr), make([]int64, len(pending)*nr), make([]uint64, len(pending))
	for w := range pending {
		l.WlPenaltyMask[w] = l.put(l.WlPenaltyLo, l.WlPenaltyHi, w, pens[w])
	}
	return l
}

// PutLedger = kq_pending_afs_put (after PutPending with LocalQueue indices).
func (e *Engine) PutLedger(l *FlatLedger) error {
	var p runtime.Pinner
	defer p.Unpin()
	c := (*C.kq_afs_ledger)(C.calloc(1, C.sizeof_kq_afs_ledger))
	defer C.free(unsafe.Pointer(c))
	c.n_lq, c.n_res = C.int32_t(len(l.LQWeight)), C.int32_t(len(l.Resources))
	c.lq_weight, c.res_weight = (*C.double)(pin(&p, l.LQWeight)), (*C.double)(pin(&p, l.ResWeight))
	c.consumed_lo, c.consumed_hi = (*C.uint64_t)(pin(&p, l.ConsumedLo)), (*C.int64_t)(pin(&p, l.ConsumedHi))
	c.consumed_f64 = (*C.double)(pin(&p, l.ConsumedF64))
	c.penalty_lo, c.penalty_hi = (*C.uint64_t)(pin(&p, l.PenaltyLo)), (*C.int64_t)(pin(&p, l.PenaltyHi))
	c.penalty_present = (*C.uint8_t)(pin(&p, l.PenaltyPresent))
	c.wl_penalty_lo, c.wl_penalty_hi = (*C.uint64_t)(pin(&p, l.WlPenaltyLo)), (*C.int64_t)(pin(&p, l.WlPenaltyHi))
	c.wl_penalty_mask = (*C.uint64_t)(pin(&p, l.WlPenaltyMask))
	if rc := C.kq_pending_afs_put(e.h, c); rc != 0 {
		return e.err("kq_pending_afs_put", rc)
	}
	return nil
}

// SubPenalty = AfsUsageLedger.SubPenalty (entry_penalties.go:45) for pending-set workloads: the rollback of scheduler.go:1032 and the
// workload controller's deletions (workload_controller.go:1296,1475,1481) call it next to the host ledger's.
func (e *Engine) SubPenalty(wl []int32) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_pending_afs_sub_penalty(e.h, C.int32_t(len(wl)), (*C.int32_t)(pin(&p, wl))); rc != 0 {
		return e.err("kq_pending_afs_sub_penalty", rc)
	}
	return nil
}

// SetConsumed = a controller's rewrite of entry.Resources (LocalQueue reconciler decay; settlement workload_controller.go:1506-1528,
// settleWl[i] = the workload whose record folds in, or -1). rows[i] is the new Resources of LocalQueue lq[i].
func (e *Engine) SetConsumed(l *FlatLedger, lq []int32, rows []corev1.ResourceList, settleWl []int32) error {
	var p runtime.Pinner
	defer p.Unpin()
	nr := len(l.Resources)
	lo, hi, f := make([]uint64, len(lq)*nr), make([]int64, len(lq)*nr), make([]float64, len(lq)*nr)
	for i, rl := range rows {
		l.put(lo, hi, i, rl)
		for n, q := range rl {
			f[i*nr+l.index[n]] = q.AsApproximateFloat64()
		}
	}
	if rc := C.kq_pending_afs_set_consumed(e.h, C.int32_t(len(lq)), (*C.int32_t)(pin(&p, lq)), (*C.uint64_t)(pin(&p, lo)), (*C.int64_t)(pin(&p, hi)),
		(*C.double)(pin(&p, f)), (*C.int32_t)(pin(&p, settleWl))); rc != 0 {
		return e.err("kq_pending_afs_set_consumed", rc)
	}
	return nil
}

// SetWorkloadPenalties = what PushPenalty would record for pending workloads that arrived after PutLedger (AddPending / UpdatePending):
// rows[i] = afs.CalculateEntryPenalty(SumTotalRequests(wl[i]), cfg) (admission_fair_sharing.go:53-60), computed where the workload
// controller computes it today.
func (e *Engine) SetWorkloadPenalties(l *FlatLedger, wl []int32, rows []corev1.ResourceList) error {
	var p runtime.Pinner
	defer p.Unpin()
	nr := len(l.Resources)
	lo, hi, mask := make([]uint64, len(wl)*nr), make([]int64, len(wl)*nr), make([]uint64, len(wl))
	for i, rl := range rows {
		mask[i] = l.put(lo, hi, i, rl)
	}
	if rc := C.kq_pending_afs_wl_penalty(e.h, C.int32_t(len(wl)), (*C.int32_t)(pin(&p, wl)), (*C.uint64_t)(pin(&p, lo)), (*C.int64_t)(pin(&p, hi)),
		(*C.uint64_t)(pin(&p, mask))); rc != 0 {
		return e.err("kq_pending_afs_wl_penalty", rc)
	}
	return nil
}

// ReadLedgerUsage = afs.CalculateUsage of every LocalQueue as the next Heads() will see it (the visibility endpoint and the LocalQueue
// status read the same number, admission_fair_sharing.go:86-103).
func (e *Engine) ReadLedgerUsage(usage []float64) error {
	var p runtime.Pinner
	defer p.Unpin()
	if rc := C.kq_pending_afs_read(e.h, (*C.double)(pin(&p, usage)), nil, nil, nil, nil, nil, nil); rc != 0 {
		return e.err("kq_pending_afs_read", rc)
	}
	return nil
}

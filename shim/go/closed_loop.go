//go:build kq_hip

// NOT COMPILED IN THIS REPOSITORY (no Go toolchain in the build image): reviewed against kueue_amd/closed_loop.py, which the tests drive.
package kqengine

// closed_loop.go — twin of kueue_amd/closed_loop.py: SURVEY §8d's "run" for a population WITH preemption, i.e.
// (*Scheduler).schedule (scheduler.go:308-386) closed over the cache cycle after cycle, handed to the engine as ONE
// kq_snapshot_patch_rows(KQ_ROWS_FOLD_USAGE) per cycle:
//   - an entry that was assumed (scheduler.go:605 admit -> cache.AssumeWorkload -> clusterQueue.updateWorkloadUsage clusterqueue.go:594)
//     is an admitted ROW from the next snapshot on (its ClusterQueue, priority, queue / reservation time, uid, usage = Assignment.Usage);
//   - the targets of an entry in Preempt mode get the Evicted condition (preemption.go:201-270 IssuePreemptions): they stay admitted, marked
//     — first in every later candidate order — until their pods are gone (here: one cycle). The preemptor is back in its heap with
//     RequeueReasonPendingPreemption (ApplyPending) and is admitted once the quota is free;
//   - a workload whose time is up leaves: its row goes, its usage leaves the tree, the inadmissible workloads of that root cohort go
//     back to their heaps (QueueAssociatedInadmissibleWorkloadsAfter) — all three inside the engine's fold.
// In the real controller the three lists come from the cache's own events (AddOrUpdateWorkload / DeleteWorkload); this file is the
// self-contained driver the benchmark and the parity tests use, and shows which decision feeds which list.

const neverCycle = int64(1) << 62

// RowBook is what the driver remembers of the resident admitted table: per row its ClusterQueue, the cycle it finishes in and the
// cycle it was marked Evicted in. Place reproduces where kq_snapshot_patch_rows puts rows: kept rows keep their order inside their
// ClusterQueue, added rows land behind them in the order given.
type RowBook struct {
	NQ        int32
	CQ        []int32
	Finish    []int64
	EvictedAt []int64
}

func NewRowBook(s *FlatSnapshot) *RowBook {
	b := &RowBook{NQ: s.NCQ}
	for c := int32(0); c < s.NCQ; c++ {
		for r := s.CQAdmOff[c]; r < s.CQAdmOff[c+1]; r++ {
			b.CQ = append(b.CQ, c)
			b.Finish = append(b.Finish, neverCycle)
			ev := neverCycle
			if s.AdmFlags[r]&1 != 0 { // KQ_ADM_EVICTED
				ev = -1
			}
			b.EvictedAt = append(b.EvictedAt, ev)
		}
	}
	return b
}

// Place applies a patch to the book. remove: ascending old rows; addCQ / addFinish: the added rows in the order they were given.
func (b *RowBook) Place(remove []int32, addCQ []int32, addFinish []int64) {
	gone := make(map[int32]bool, len(remove))
	for _, r := range remove {
		gone[r] = true
	}
	kept := make([][]int, b.NQ) // old rows per ClusterQueue, in table order
	for r, c := range b.CQ {
		if !gone[int32(r)] {
			kept[c] = append(kept[c], r)
		}
	}
	added := make([][]int, b.NQ)
	for i, c := range addCQ {
		added[c] = append(added[c], i)
	}
	var cq []int32
	var fin, ev []int64
	for c := int32(0); c < b.NQ; c++ {
		for _, r := range kept[c] {
			cq, fin, ev = append(cq, c), append(fin, b.Finish[r]), append(ev, b.EvictedAt[r])
		}
		for _, i := range added[c] {
			cq, fin, ev = append(cq, c), append(fin, addFinish[i]), append(ev, neverCycle)
		}
	}
	b.CQ, b.Finish, b.EvictedAt = cq, fin, ev
}

// AssignmentRows builds the admitted rows of the heads `sel` of a cycle (RowPatch.Add*): usage = Assignment.Usage
// (flavorassigner.go:1017-1041) — per (podset, resource) that was given a flavor, the podset's request scaled to the admitted count
// (workload.go:317-340), the injected `pods` request (flavorassigner.go:743-749) = the count.
func AssignmentRows(s *FlatSnapshot, h *FlatHeads, d *FlatDecisions, sel []int, reserveTs int64, uidRank []uint32, p *RowPatch) {
	nR := s.NResource
	p.AddUseOff = append(p.AddUseOff[:0], 0)
	for k, i := range sel {
		use := map[int32]int64{}
		var order []int32
		for ps := h.PsOff[i]; ps < h.PsOff[i+1]; ps++ {
			cnt0, cnt := int64(h.PsCount[ps]), int64(d.PsCount[ps])
			req := map[int32]int64{}
			for e := h.PsReqOff[ps]; e < h.PsReqOff[ps+1]; e++ {
				req[h.ReqRes[e]] = h.ReqQty[e]
			}
			for r := int32(0); r < nR; r++ {
				f := d.Flavor[ps*nR+r]
				if f < 0 {
					continue
				}
				q := req[r]
				if cnt0 != 0 && cnt0 != cnt {
					q = (q / cnt0) * cnt
				}
				if r == s.PodsResource && podsCovered(s, h.CQ[i]) {
					q = cnt
				}
				fr := f*nR + r
				if _, ok := use[fr]; !ok {
					order = append(order, fr)
				}
				use[fr] += q
			}
		}
		p.AddCQ = append(p.AddCQ, h.CQ[i])
		p.AddPriority = append(p.AddPriority, h.Priority[i])
		p.AddQueueTs = append(p.AddQueueTs, h.QueueTs[i])
		p.AddReserveTs = append(p.AddReserveTs, reserveTs)
		p.AddUIDRank = append(p.AddUIDRank, uidRank[k])
		p.AddFlags = append(p.AddFlags, 0)
		for _, fr := range order {
			p.AddUseFr = append(p.AddUseFr, fr)
			p.AddUseQty = append(p.AddUseQty, use[fr])
		}
		p.AddUseOff = append(p.AddUseOff, int32(len(p.AddUseFr)))
	}
}

func podsCovered(s *FlatSnapshot, cq int32) bool {
	if s.PodsResource < 0 {
		return false
	}
	for g := s.CQRgOff[cq]; g < s.CQRgOff[cq+1]; g++ {
		for k := s.RgResOff[g]; k < s.RgResOff[g+1]; k++ {
			if s.RgRes[k] == s.PodsResource {
				return true
			}
		}
	}
	return false
}

// CyclePatch = kueue_amd/closed_loop.py cycle_patch: remove = the rows whose time is up + the rows an EARLIER cycle marked Evicted;
// add = the heads this cycle admitted; evict = its preemption targets that are still there and not marked yet.
func CyclePatch(b *RowBook, s *FlatSnapshot, clock int64, uidBase uint32, cycle int64, h *FlatHeads, d *FlatDecisions, headWl []int32) (p *RowPatch, admitted, preempting int) {
	p = &RowPatch{FoldUsage: true}
	gone := make([]bool, len(b.CQ))
	for r := range b.CQ {
		if b.Finish[r] <= cycle || (b.EvictedAt[r] != neverCycle && b.EvictedAt[r] < cycle) {
			gone[r] = true
			p.RemoveRows = append(p.RemoveRows, int32(r))
		}
	}
	if d == nil {
		return p, 0, 0
	}
	var adm []int
	var uid []uint32
	marked := map[int32]bool{}
	for i := 0; i < int(h.N); i++ {
		switch d.Action[i] {
		case 1: // KQ_ACT_ADMIT
			adm = append(adm, i)
			uid = append(uid, uidBase+uint32(headWl[i]))
		case 2: // KQ_ACT_PREEMPT
			preempting++
			for k := d.TgtOff[i]; k < d.TgtOff[i+1]; k++ {
				t := d.TgtAdm[k]
				if !gone[t] && b.EvictedAt[t] == neverCycle && !marked[t] {
					marked[t] = true
					p.EvictRows = append(p.EvictRows, t)
				}
			}
		}
	}
	if len(adm) > 0 {
		AssignmentRows(s, h, d, adm, clock, uid, p)
	}
	return p, len(adm), preempting
}

// PreemptionLoop drives one engine with the snapshot and the pending set resident. gather(headWl) returns the heads batch of the
// cycle with the STATIC columns the rows need (cq, priority, queue timestamp, podsets, requests) for the workloads Heads() popped.
type PreemptionLoop struct {
	E        *Engine
	S        *FlatSnapshot
	Book     *RowBook
	Hold     int64 // cycles an admitted workload runs; <= 0: it does not finish inside the run
	Clock    int64
	TickNs   int64
	UIDBase  uint32
	Out      *FlatDecisions
	HeadWl   []int32
	Gather   func(headWl []int32, cycle int64) *FlatHeads
	newIndex []int32
}

// Step runs one scheduling cycle and applies it: Heads() + the cycle + the requeue policy on the device, then the row patch.
func (l *PreemptionLoop) Step(cycle int64) (h *FlatHeads, d *FlatDecisions, err error) {
	n, _, err := l.E.Heads(cycle, nil, l.HeadWl)
	if err != nil {
		return nil, nil, err
	}
	if n > 0 {
		if err = l.E.RunPendingCycle(l.Out); err != nil {
			return nil, nil, err
		}
		d = l.Out
	}
	if err = l.E.ApplyPending(); err != nil {
		return nil, nil, err
	}
	var wl []int32
	for _, w := range l.HeadWl {
		if w >= 0 {
			wl = append(wl, w)
		}
	}
	if n > 0 {
		h = l.Gather(wl, cycle)
	}
	p, _, _ := CyclePatch(l.Book, l.S, l.Clock, l.UIDBase, cycle, h, d, wl)
	if len(p.RemoveRows)+len(p.AddCQ)+len(p.EvictRows) > 0 {
		for _, r := range p.EvictRows {
			l.Book.EvictedAt[r] = cycle
		}
		if cap(l.newIndex) < len(l.Book.CQ) {
			l.newIndex = make([]int32, len(l.Book.CQ))
		}
		if err = l.E.PatchRows(p, l.newIndex[:len(l.Book.CQ)]); err != nil {
			return h, d, err
		}
		fin := make([]int64, len(p.AddCQ))
		for i := range fin {
			if l.Hold > 0 {
				fin[i] = cycle + l.Hold
			} else {
				fin[i] = neverCycle
			}
		}
		l.Book.Place(p.RemoveRows, p.AddCQ, fin)
	}
	l.Clock += l.TickNs
	return h, d, nil
}

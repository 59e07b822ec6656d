// kq_pending.hpp — the pending side on the device: every pending workload of every ClusterQueue resident in
// HBM, `Heads()` as a segmented arg-min, the requeue policy driven by the cycle's decisions.
//
// Reference semantics followed (paths under /root/reference/pkg):
//   cache/queue/cluster_queue.go   baseCompareFunc :844 (sticky, priority desc, queue-order timestamp asc, UID asc)
//                                  Pop :657 (popCycle++)   RequeueIfNotPresent :826   requeueIfNotPresent :550
//                                  handleInadmissibleHash :606   preemptorWorkload :80-154   IsPreemptor :213   delete :495
//   cache/queue/inadmissible_workloads.go  queueInadmissibleWorkloads :149
//   cache/queue/manager.go         heads :922 (one Pop per ClusterQueue)
//   scheduler/scheduler.go         recordAssignment :281  markSkipped :248  markPreemptionOutcome :291  DeferredFit :459
//                                  requeueAndUpdate :1165
//   workload/workload.go           PendingFlavors :211
//
// Layout. The W workloads keep the columns of a heads batch (kq_heads): they never move. The heap order of a ClusterQueue
// is STATIC except for the sticky preemptor (priority, timestamp and UID do not change while a workload is pending), so the
// host sorts every ClusterQueue's workloads once (kq_pending_put) and the heap is a byte of state per workload:
//   Pop            = first ACTIVE position of the ClusterQueue's sorted segment (ballot over 64 positions per step), unless the
//                    sticky preemptor is ACTIVE — it precedes everything (baseCompareFunc :848-856);
//   requeue        = one state byte written; the bulk move of an equivalence class = one masked pass over the segment;
//   next cycle's resume state (LastAssignment) is kept per workload next to the static columns.
#pragma once

namespace kq {

enum { WL_ACTIVE = KQ_WL_ACTIVE, WL_INFLIGHT = KQ_WL_INFLIGHT, WL_INADMISSIBLE = KQ_WL_INADMISSIBLE, WL_GONE = KQ_WL_GONE };

struct DPend {
  int W, nq, nR, nfw;
  DHeads P;                  // static columns of the W workloads (flags / last_* here are the values at kq_pending_put)
  const uint32_t* uid;       // [W]
  const int32_t* cq_off;     // [nq+1] heap-ordered workloads of ClusterQueue c: ord[cq_off[c] .. cq_off[c+1])
  const int32_t* ord;        // [W]
  // mutable queue state
  uint8_t* state;            // [W] WL_*
  uint8_t* bulk;             // [W] the workload's requeue bulk-moved its equivalence class (its hash is in hashToBulkMoveReason,
                             //     cluster_queue.go:169-172) since the ClusterQueue's last queueInadmissibleWorkloads
  uint32_t* mflags;          // [W] KQ_HEAD_* (HAS_LAST_ASSIGNMENT evolves with the cycles)
  int32_t* last_tried;       // [n_ps_total * nR] LastAssignment.LastTriedFlavorIdx
  int64_t *last_gen, *last_cycle;
  uint64_t* last_hash;
  int32_t* pw;               // [nq] preemptorWorkload (workload id, -1 = none)
  uint8_t* pw_sticky;        // [nq]
  int64_t *pop_cycle, *qi_cycle;  // [nq] popCycle, queueInadmissibleCycle (-1 at start, cluster_queue.go:313)
  // the heads of the cycle in flight
  int32_t* head_wl;          // [nq] workload popped from ClusterQueue c, -1 = none
  int32_t* hd;               // [n_heads] workload of head h (canonical head order = ClusterQueue index ascending)
  int32_t* hreq;             // [n_heads] first request of head h in the gathered batch
  int32_t* counts;           // [4] n_heads, n_podsets, n_requests
  const uint8_t* cq_active;  // [nq] or null: statusChecker.ClusterQueueActive (manager.go:926)
  // AdmissionFairSharing ordering (queueOrderingFunc cluster_queue.go:880): LocalQueue of every workload (-1: none) and the
  // LocalQueues' fair-sharing usage (host-evaluated afs.CalculateUsage); lq == null: baseCompareFunc everywhere
  const int32_t* lq;         // [W] or null
  const double* lq_usage;    // [n_lq]
  // back-off (backoffWaitingTimeExpired cluster_queue.go:474): RequeueAt per workload (null: nobody backs off) and the queues' clock
  int64_t* requeue_at;       // [W] or null
  int64_t now;
};
KQ_DEV bool pend_backoff_expired(const DPend& D, int w) {
  if (!D.requeue_at) return true;
  const int64_t at = D.requeue_at[w];
  if (at == KQ_REQUEUE_NONE) return true;
  if (at == KQ_REQUEUE_BLOCKED) return false;   // Requeued condition False :475
  return D.now >= at;                           // Now().After(at) || Now().Equal(at) :483
}
// Go cmp.Compare(float64) as a sortable 64-bit key: NaN sorts before everything, then -Inf .. +Inf (-0 == +0)
KQ_DEV uint64_t afs_key(double v) {
  if (v != v) return 0;
  if (v == 0) v = 0;  // -0 -> +0
  uint64_t b;
#ifdef KQ_HOST_EMU
  memcpy(&b, &v, 8);
#else
  b = (uint64_t)__double_as_longlong(v);
#endif
  return ((b >> 63) ? ~b : (b | 0x8000000000000000ull)) + 1;  // + 1 keeps 0 for NaN (the largest key, +Inf, does not overflow)
}

// the gathered batch (same arrays as a kq_heads upload), written by pend_gather_head
struct DGather {
  int32_t* cq; int64_t* priority; int64_t* queue_ts; uint32_t* flags; int32_t* ps_off;
  int32_t *ps_count, *ps_min_count, *ps_req_off, *req_res; int64_t* req_qty; uint64_t* ps_flavor_ok; int32_t* ps_last_tried;
  int64_t *last_generation, *last_cycle; uint64_t *last_hash, *hash;
};

// ClusterQueue.Pop (cluster_queue.go:657-672) for ClusterQueue c — one wave
KQ_DEV void pend_pop(const DPend& D, int c) {
  const int lane = lane_id();
  if (D.cq_active && !D.cq_active[c]) { if (lane == 0) D.head_wl[c] = -1; return; }  // manager.go:926: no Pop at all
  int head = -1;
  const int pw = D.pw[c];
  const bool sticky = pw >= 0 && D.pw_sticky[c] && D.state[pw] == WL_ACTIVE;
  if (sticky && !D.lq) head = pw;  // stickyMatches sorts first (:848-856)
  if (head < 0 && !D.lq) {
    const int o0 = D.cq_off[c], o1 = D.cq_off[c + 1];
    for (int base = o0; base < o1 && head < 0; base += WAVE) {
      const int j = base + lane;
      const int w = j < o1 ? D.ord[j] : -1;
      const uint64_t m = wballot(w >= 0 && D.state[w] == WL_ACTIVE);
      if (m) head = wbcast(w, ffs64(m));
    }
  }
  if (head < 0 && D.lq) {
    // queueOrderingFunc (:880-904): LocalQueue usage FIRST, then baseCompareFunc = [sticky preemptor, the static base order (the
    // position in the sorted segment)] — the sticky workload only wins among equal usage keys. A wave arg-min over
    // (usage key, 0 for the sticky workload | 1 + position): 64 positions per step.
    const int o0 = D.cq_off[c], o1 = D.cq_off[c + 1];
    uint64_t best_k = ~0ull; uint32_t best_pos = 0xffffffffu;
    for (int base = o0; base < o1; base += WAVE) {
      const int j = base + lane;
      const int w = j < o1 ? D.ord[j] : -1;
      if (w >= 0 && D.state[w] == WL_ACTIVE) {
        const int l = D.lq[w];
        const uint64_t kk = l >= 0 ? afs_key(D.lq_usage[l]) : afs_key(0.0);
        const uint32_t pk = (sticky && w == pw) ? 0u : (uint32_t)(j - o0) + 1u;
        if (kk < best_k || (kk == best_k && pk < best_pos)) { best_k = kk; best_pos = pk; }
      }
    }
    const uint64_t mk = wmin_u64(best_k);
    const uint64_t mp = wmin_u64(best_k == mk ? (uint64_t)best_pos : ~0ull);
    if (mk != ~0ull && mp != 0xffffffffull) head = mp == 0 ? pw : D.ord[o0 + (int)mp - 1];
  }
  if (lane == 0) {
    D.pop_cycle[c] += 1;  // :670, also when the heap is empty
    if (head >= 0) D.state[head] = WL_INFLIGHT;
    D.head_wl[c] = head;
  }
}

// Compaction of the popped heads into batch positions: one workgroup, nthreads threads, chunks of `nthreads` ClusterQueues with
// a running carry. scan3 = LDS scratch [3][nthreads]. Head order = ClusterQueue index ascending.
KQ_DEV void pend_scan(const DPend& D, const DGather& G, int tid, int nthreads, int32_t* scan3) {
  int carry_h = 0, carry_p = 0, carry_r = 0;
  for (int base = 0; base < D.nq; base += nthreads) {
    const int c = base + tid;
    const int w = c < D.nq ? D.head_wl[c] : -1;
    int nps = 0, nreq = 0;
    if (w >= 0) { const int p0 = D.P.ps_off[w], p1 = D.P.ps_off[w + 1]; nps = p1 - p0; nreq = D.P.ps_req_off[p1] - D.P.ps_req_off[p0]; }
#ifdef KQ_HOST_EMU
    (void)scan3;
    if (w >= 0) { D.hd[carry_h] = w; G.ps_off[carry_h] = carry_p; D.hreq[carry_h] = carry_r; carry_h++; carry_p += nps; carry_r += nreq; }
#else
    int32_t* sh = scan3; int32_t* sp = scan3 + nthreads; int32_t* sr = scan3 + 2 * nthreads;
    sh[tid] = w >= 0 ? 1 : 0; sp[tid] = nps; sr[tid] = nreq;
    __syncthreads();
    for (int o = 1; o < nthreads; o <<= 1) {  // Hillis-Steele inclusive scan of the three counters
      const int a = tid >= o ? sh[tid - o] : 0, b = tid >= o ? sp[tid - o] : 0, d = tid >= o ? sr[tid - o] : 0;
      __syncthreads();
      sh[tid] += a; sp[tid] += b; sr[tid] += d;
      __syncthreads();
    }
    if (w >= 0) {
      const int h = carry_h + sh[tid] - 1;
      D.hd[h] = w; G.ps_off[h] = carry_p + sp[tid] - nps; D.hreq[h] = carry_r + sr[tid] - nreq;
    }
    carry_h += sh[nthreads - 1]; carry_p += sp[nthreads - 1]; carry_r += sr[nthreads - 1];
    __syncthreads();
#endif
  }
  if (tid == 0) {
    G.ps_off[carry_h] = carry_p;
    G.ps_req_off[carry_p] = carry_r;
    D.counts[0] = carry_h; D.counts[1] = carry_p; D.counts[2] = carry_r; D.counts[3] = 0;
  }
}

// One head's rows of the batch — one wave per head. workload.Info as the scheduler sees it this cycle: static columns from the
// store, LastAssignment from the per-workload resume state, IsPreemptor from the ClusterQueue's preemptorWorkload.
KQ_DEV void pend_gather_head(const DPend& D, const DGather& G, int h) {
  const int lane = lane_id();
  const int w = D.hd[h];
  const int c = D.P.cq[w];
  const int p0 = D.P.ps_off[w], nps = D.P.ps_off[w + 1] - p0;
  const int gp0 = G.ps_off[h];
  const int r0 = D.P.ps_req_off[p0], nreq = D.P.ps_req_off[p0 + nps] - r0;
  const int gr0 = D.hreq[h];
  if (lane == 0) {
    G.cq[h] = c; G.priority[h] = D.P.priority[w]; G.queue_ts[h] = D.P.queue_ts[w];
    uint32_t fl = D.mflags[w] & ~(uint32_t)KQ_HEAD_IS_PREEMPTOR;
    if (D.pw[c] == w) fl |= KQ_HEAD_IS_PREEMPTOR;  // IsPreemptor :213 (generation unchanged while pending)
    G.flags[h] = fl;
    G.last_generation[h] = D.last_gen[w]; G.last_cycle[h] = D.last_cycle[w]; G.last_hash[h] = D.last_hash[w]; G.hash[h] = D.P.hash[w];
  }
  for (int i = lane; i < nps; i += WAVE) {
    G.ps_count[gp0 + i] = D.P.ps_count[p0 + i]; G.ps_min_count[gp0 + i] = D.P.ps_min_count[p0 + i];
    G.ps_req_off[gp0 + i] = gr0 + (D.P.ps_req_off[p0 + i] - r0);
  }
  for (int i = lane; i < nreq; i += WAVE) { G.req_res[gr0 + i] = D.P.req_res[r0 + i]; G.req_qty[gr0 + i] = D.P.req_qty[r0 + i]; }
  for (int i = lane; i < nps * D.nfw; i += WAVE) G.ps_flavor_ok[(size_t)gp0 * D.nfw + i] = D.P.ps_flavor_ok[(size_t)p0 * D.nfw + i];
  for (int i = lane; i < nps * D.nR; i += WAVE) G.ps_last_tried[(size_t)gp0 * D.nR + i] = D.last_tried[(size_t)p0 * D.nR + i];
}

// Step 6 of schedule() for head h (scheduler.go:362-377): the queue side of requeueAndUpdate :1165 — or, for an admitted entry,
// the workload leaving the queue (assumeWorkload; the controller's delete, cluster_queue.go:495). One wave per head.
KQ_DEV void pend_apply_head(const DPend& D, const DSnap& S, const DOut& O, const DHeads& H, uint32_t gates, int64_t cycle, int h) {
  const int lane = lane_id();
  const int w = D.hd[h];
  const int c = D.P.cq[w];
  const int status = O.status[h], action = O.action[h], mode = O.mode[h], rq = O.requeue_reason[h];
  if (status == KQ_ST_ASSUMED) {
    if (lane == 0) { D.state[w] = WL_GONE; if (D.pw[c] == w) { D.pw[c] = -1; D.pw_sticky[c] = 0; } }
    return;
  }
  const bool strict = KQ_POL_STRICT_FIFO(S.cq_policy[c]) != 0;
  // next LastAssignment: recordAssignment :281 unless cleared by markPreemptionOutcome :291 (preemptions issued), DeferredFit
  // :459-464, or markSkipped without FlavorFungibilityPreserveScanProgress :248-254
  const bool nil_last = action == KQ_ACT_PREEMPT || mode == KQ_MODE_DEFERRED_FIT ||
                        (status == KQ_ST_SKIPPED && !(gates & KQ_GATE_PRESERVE_SCAN_PROGRESS));
  const int p0 = D.P.ps_off[w], nps = D.P.ps_off[w + 1] - p0, gp0 = H.ps_off[h];
  bool pend = false;
  for (int i = lane; i < nps * D.nR; i += WAVE) {
    const int t = nil_last ? -1 : O.tried_idx[(size_t)gp0 * D.nR + i];
    D.last_tried[(size_t)p0 * D.nR + i] = t;
    if (t != -1) pend = true;
  }
  const bool pending_flavors = wballot(pend) != 0;  // PendingFlavors workload.go:211-224
  if (lane != 0) { /* the bulk move below is wave-wide; scalars are written by lane 0 */ }
  if (lane == 0) {
    if (nil_last) D.mflags[w] &= ~(uint32_t)KQ_HEAD_HAS_LAST_ASSIGNMENT;
    else { D.mflags[w] |= KQ_HEAD_HAS_LAST_ASSIGNMENT; D.last_gen[w] = S.cq_gen[c]; D.last_cycle[w] = cycle; D.last_hash[w] = D.P.hash[w]; }
    if (rq == KQ_RQ_PENDING_PREEMPTION) { D.pw[c] = w; D.pw_sticky[c] = strict ? 0 : 1; }  // :558-563
  }
  // RequeueIfNotPresent :826-841
  const bool immediate = strict ? true : (rq == KQ_RQ_FAILED_AFTER_NOMINATION || rq == KQ_RQ_PENDING_PREEMPTION);
  // requeueIfNotPresent :568-575
  if (pend_backoff_expired(D, w) && (immediate || D.qi_cycle[c] >= D.pop_cycle[c] || pending_flavors)) { if (lane == 0) D.state[w] = WL_ACTIVE; return; }
  if (lane == 0) D.state[w] = WL_INADMISSIBLE;  // :585
  // :592-597 bulk move of the equivalence class (handleInadmissibleHash :606-621; BestEffortFIFO only)
  const uint64_t hash = D.P.hash[w];
  if ((gates & KQ_GATE_SCHEDULING_EQUIVALENCE_HASHING) && hash != 0 && !strict && (rq == KQ_RQ_NOFIT || rq == KQ_RQ_PREEMPTION_NO_CANDIDATES)) {
    if (lane == 0) D.bulk[w] = 1;  // c.hashToBulkMoveReason[hash] = reason :610
    for (int j = D.cq_off[c] + lane; j < D.cq_off[c + 1]; j += WAVE) {
      const int w2 = D.ord[j];
      if (w2 != w && D.state[w2] == WL_ACTIVE && D.P.hash[w2] == hash) D.state[w2] = WL_INADMISSIBLE;
    }
  }
}

// queueInadmissibleWorkloads (inadmissible_workloads.go:149-175) for ClusterQueue c — one wave
KQ_DEV void pend_queue_inadmissible(const DPend& D, int c) {
  const int lane = lane_id();
  if (lane == 0) D.qi_cycle[c] = D.pop_cycle[c];
  for (int j = D.cq_off[c] + lane; j < D.cq_off[c + 1]; j += WAVE) {
    const int w = D.ord[j];
    if (D.state[w] == WL_INADMISSIBLE && pend_backoff_expired(D, w)) D.state[w] = WL_ACTIVE;  // :167: a workload still backing off stays
    D.bulk[w] = 0;  // c.hashToBulkMoveReason = make(...) inadmissible_workloads.go:158
  }
}
// PushOrUpdate of a workload that was just appended (cluster_queue.go:419-425): BestEffortFIFO, hash known, class already bulk-moved
// => it joins the inadmissible workloads instead of the heap. One wave per new workload.
KQ_DEV void pend_add_fix(const DPend& D, const DSnap& S, int w) {
  const int lane = lane_id();
  const int c = D.P.cq[w];
  if (!pend_backoff_expired(D, w)) { if (lane == 0) D.state[w] = WL_INADMISSIBLE; return; }  // :414
  const uint64_t hash = D.P.hash[w];
  if (hash == 0 || KQ_POL_STRICT_FIFO(S.cq_policy[c]) != 0) return;
  bool hit = false;
  for (int j = D.cq_off[c] + lane; j < D.cq_off[c + 1]; j += WAVE) {
    const int w2 = D.ord[j];
    if (D.bulk[w2] && D.P.hash[w2] == hash) hit = true;
  }
  if (wballot(hit) != 0 && lane == 0) D.state[w] = WL_INADMISSIBLE;
}
// PushOrUpdate of a pending workload whose RequeueState / Requeued condition changed (cluster_queue.go:391-428): one wave per workload
KQ_DEV void pend_requeue_at(const DPend& D, const DSnap& S, const int32_t* list, const int64_t* at, int i) {
  const int lane = lane_id();
  const int w = list[i];
  if (lane == 0) D.requeue_at[w] = at[i];
  wsync();
  if (D.state[w] != WL_INADMISSIBLE) return;   // in flight: RequeueWorkload places it (:388); in the heap: stays (:414 needs GetActive == nil)
  if (lane == 0) D.state[w] = WL_ACTIVE;       // conditions changed => leaves the inadmissible set (:405) ...
  wsync();
  pend_add_fix(D, S, w);                       // ... unless it still backs off or its class is bulk-moved
}
// ClusterQueue.Delete :488-512 — one thread per workload
KQ_DEV void pend_delete(const DPend& D, const int32_t* list, int i) {
  const int w = list[i];
  const int c = D.P.cq[w];
  D.state[w] = WL_GONE;
  if (D.pw[c] == w) { D.pw[c] = -1; D.pw_sticky[c] = 0; }
}

// Workloads finishing free quota: the cache notifies the queues, which move the inadmissible workloads of the whole ROOT cohort of
// every affected ClusterQueue back to their heaps (QueueAssociatedInadmissibleWorkloadsAfter -> requeueWorkloadsCohort,
// inadmissible_workloads.go:112-147; a ClusterQueue without a cohort requeues itself). Two steps: the released admissions stamp their
// tree, then every ClusterQueue of a stamped tree runs queueInadmissibleWorkloads. `stamp` is unique per release, nothing is cleared.
KQ_DEV void pend_release_mark(const DSnap& S, int32_t* tree_stamp, const int32_t* cq, const int32_t* use_n, int n, int i, int32_t stamp) {
  if (i < n && use_n[i] > 0) tree_stamp[S.tree_of[cq[i]]] = stamp;
}
KQ_DEV void pend_release_requeue(const DPend& D, const DSnap& S, const int32_t* tree_stamp, int c, int32_t stamp) {
  if (tree_stamp[S.tree_of[c]] == stamp) pend_queue_inadmissible(D, c);
}

}  // namespace kq

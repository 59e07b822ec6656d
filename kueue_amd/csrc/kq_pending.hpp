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

// AdmissionFairSharing ledger (pkg/cache/queue/afs/usage_ledger.go:38-60): per (LocalQueue, resource) the consumed history and the
// aggregate of the pending entry penalties, exact 128-bit integers in units of 1e-9; per workload what PushPenalty records.
struct DAfs {
  int n_lq, n_res;           // n_res == 0: no ledger resident (DPend::lq_usage is the host's)
  const double *lq_weight, *res_weight;
  uint64_t* cons_lo; int64_t* cons_hi; double* cons_f64;     // [n_lq * n_res] entry.Resources
  uint64_t* pen_lo; int64_t* pen_hi; uint8_t* pen_present;   // [n_lq * n_res] entry.pendingPenalty
  uint64_t* wl_lo; int64_t* wl_hi; uint64_t* wl_mask;        // [W * n_res], [W]
  uint8_t* wl_rec;           // [W] penaltyRecords[wlKey] exists
  double* usage;             // [n_lq] afs.CalculateUsage, rewritten by afs_usage_lq
};

struct DPend {
  int W, nq, nR, nfw;
  DAfs A;
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
  uint8_t* pw_sticky;        // [nq] bit 0: isSticky; bit 1: the workload's object changed since the pointer was set (generation differs: IsPreemptor no more, stickyMatches still)
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

// ---- ledger arithmetic ----------------------------------------------------------------------------------------------------
struct I128 { uint64_t lo; int64_t hi; };
KQ_DEV I128 i128_add(I128 a, I128 b) { I128 r; r.lo = a.lo + b.lo; r.hi = (int64_t)((uint64_t)a.hi + (uint64_t)b.hi + (r.lo < a.lo ? 1u : 0u)); return r; }
KQ_DEV I128 i128_neg(I128 a) { I128 r; r.lo = ~a.lo + 1; r.hi = (int64_t)(~(uint64_t)a.hi + (r.lo == 0 ? 1u : 0u)); return r; }
KQ_DEV double afs_mul(double a, double b) {
#ifdef KQ_HOST_EMU
  volatile double r = a * b; return r;
#else
  return __dmul_rn(a, b);
#endif
}
KQ_DEV double afs_addd(double a, double b) {
#ifdef KQ_HOST_EMU
  volatile double r = a + b; return r;
#else
  return __dadd_rn(a, b);
#endif
}
// big.NewFloat(0).SetInt(unscaled).Float64() (quantity.go:472): the integer rounded to the nearest double, ties to even
KQ_DEV double i128_to_f64(I128 v) {
  const bool neg = v.hi < 0;
  if (neg) v = i128_neg(v);
  const uint64_t hi = (uint64_t)v.hi, lo = v.lo;
  double r;
  if (hi == 0) {
    if (lo < (1ull << 53)) r = (double)(int64_t)lo;
    else {
      const int lz = clz64(lo), sh = 11 - lz;          // keep 53 bits
      uint64_t m = lo >> sh; const uint64_t rem = lo & ((1ull << sh) - 1), half = 1ull << (sh - 1);
      if (rem > half || (rem == half && (m & 1))) m++;
      r = (double)(int64_t)m;                            // exact: m <= 2^53
      for (int i = 0; i < sh; i++) r = afs_mul(r, 2.0);  // exact scaling
    }
  } else {
    const int lz = clz64(hi), bits = 128 - lz, sh = bits - 53;  // sh in 12..75
    uint64_t m, rem_hi, rem_lo;                               // m = v >> sh; rem = v & (2^sh - 1)
    if (sh >= 64) { m = hi >> (sh - 64); rem_hi = sh == 64 ? 0 : (hi & ((1ull << (sh - 64)) - 1)); rem_lo = lo; }
    else { m = (hi << (64 - sh)) | (lo >> sh); rem_hi = 0; rem_lo = lo & ((1ull << sh) - 1); }
    // half = 2^(sh-1)
    const uint64_t half_hi = sh - 1 >= 64 ? 1ull << (sh - 65) : 0, half_lo = sh - 1 >= 64 ? 0 : 1ull << (sh - 1);
    const bool gt = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo), eq = rem_hi == half_hi && rem_lo == half_lo;
    if (gt || (eq && (m & 1))) m++;
    r = (double)(int64_t)m;
    for (int i = 0; i < sh; i++) r = afs_mul(r, 2.0);
  }
  return neg ? -r : r;
}
// Quantity.AsApproximateFloat64 of an amount held at scale 9: base * math.Pow10(-9) (quantity.go:482; Pow10(-9) = 1 / 1e9)
KQ_DEV double afs_nano_f64(I128 v) { return afs_mul(i128_to_f64(v), 1e-9); }

// entry.WithoutPenalty(wlKey) (usage_ledger.go:96-120): the record's amounts leave the aggregate, zero keys are dropped
KQ_DEV void afs_without(const DAfs& A, int l, int w) {
  if (!A.wl_rec[w]) return;
  A.wl_rec[w] = 0;
  const uint64_t mask = A.wl_mask[w];
  for (int r = 0; r < A.n_res; r++) {
    const size_t o = (size_t)l * A.n_res + r;
    I128 agg{A.pen_lo[o], A.pen_hi[o]};
    if ((mask >> r) & 1) {
      const size_t q = (size_t)w * A.n_res + r;
      agg = i128_add(agg, i128_neg(I128{A.wl_lo[q], A.wl_hi[q]}));
      A.pen_lo[o] = agg.lo; A.pen_hi[o] = agg.hi;
    }
    if (agg.lo == 0 && agg.hi == 0) A.pen_present[o] = 0;   // :113-117 (every zero key of the aggregate)
  }
}
// entry.withPenalty(wlKey, penalty) (usage_ledger.go:78-90)
KQ_DEV void afs_push(const DAfs& A, int l, int w) {
  afs_without(A, l, w);
  const uint64_t mask = A.wl_mask[w];
  for (int r = 0; r < A.n_res; r++) if ((mask >> r) & 1) {
    const size_t o = (size_t)l * A.n_res + r, q = (size_t)w * A.n_res + r;
    const I128 agg = i128_add(I128{A.pen_lo[o], A.pen_hi[o]}, I128{A.wl_lo[q], A.wl_hi[q]});
    A.pen_lo[o] = agg.lo; A.pen_hi[o] = agg.hi; A.pen_present[o] = 1;
  }
  A.wl_rec[w] = 1;
}
// afs.CalculateUsage(consumed, penalty, lqWeight, resWeights) (admission_fair_sharing.go:86-103) for LocalQueue l
KQ_DEV void afs_usage_lq(const DAfs& A, int l) {
  double usage = 0.0;
  for (int r = 0; r < A.n_res; r++) {   // sorted key order = dictionary order
    const size_t o = (size_t)l * A.n_res + r;
    // MergeResourceListKeepSum: a key of both lists is Quantity.Add (infDec at scale 9 once a penalty takes part), a key of
    // one list is copied
    const double v = A.pen_present[o] ? afs_nano_f64(i128_add(I128{A.cons_lo[o], A.cons_hi[o]}, I128{A.pen_lo[o], A.pen_hi[o]})) : A.cons_f64[o];
    usage = afs_addd(usage, afs_mul(A.res_weight[r], v));
  }
  const double lw = A.lq_weight[l];
  const double inf = __builtin_huge_val();   // math.Inf(1) :99
#ifdef KQ_HOST_EMU
  A.usage[l] = lw <= 0 ? inf : usage / lw;
#else
  A.usage[l] = lw <= 0 ? inf : __ddiv_rn(usage, lw);
#endif
}
// every LocalQueue's usage after a ledger upload (init_f64: the consumed amounts are in the scale-9 form)
KQ_DEV void afs_init_lq(const DAfs& A, int l, bool init_f64) {
  if (init_f64) for (int r = 0; r < A.n_res; r++) { const size_t o = (size_t)l * A.n_res + r; A.cons_f64[o] = afs_nano_f64(I128{A.cons_lo[o], A.cons_hi[o]}); }
  afs_usage_lq(A, l);
}
// AfsUsageLedger.SubPenalty (entry_penalties.go:45-52) for the listed workloads, in list order
KQ_DEV void afs_sub_list(const DPend& D, const int32_t* list, int n) {
  for (int i = 0; i < n; i++) {
    const int w = list[i], l = D.lq ? D.lq[w] : -1;
    if (l < 0 || !D.A.wl_rec[w]) continue;
    afs_without(D.A, l, w);
    afs_usage_lq(D.A, l);
  }
}
// entry.Resources of LocalQueue lq[i] rewritten by a controller; with settle[i] >= 0 that workload's record folds in the same write
// (workload_controller.go:1519-1526: remaining, penalty := old.WithoutPenalty(wlKey); Resources = newConsumed + penalty)
// A settle_wl that is not a workload of LocalQueue lq[i] is ignored: the reference's penaltyRecords are per entry (usage_ledger.go:113), a
// foreign key is simply not in that entry's map; here it would take another LocalQueue's penalty out of this row's aggregate.
KQ_DEV void afs_set_consumed(const DAfs& A, const int32_t* wl_lq, const int32_t* lq, const uint64_t* lo, const int64_t* hi, const double* f64, const int32_t* settle, int i) {
  const int l = lq[i];
  int w = settle ? settle[i] : -1;
  if (w >= 0 && wl_lq && wl_lq[w] != l) w = -1;
  uint64_t fold = 0;
  if (w >= 0 && A.wl_rec[w]) { fold = A.wl_mask[w]; afs_without(A, l, w); }
  for (int r = 0; r < A.n_res; r++) {
    const size_t o = (size_t)l * A.n_res + r, q = (size_t)i * A.n_res + r;
    I128 v{lo[q], hi[q]};
    double f = f64 ? f64[q] : afs_nano_f64(v);
    if ((fold >> r) & 1) { v = i128_add(v, I128{A.wl_lo[(size_t)w * A.n_res + r], A.wl_hi[(size_t)w * A.n_res + r]}); f = afs_nano_f64(v); }
    A.cons_lo[o] = v.lo; A.cons_hi[o] = v.hi; A.cons_f64[o] = f;
  }
  afs_usage_lq(A, l);
}

// the gathered batch (same arrays as a kq_heads upload), written by pend_gather_head
struct DGather {
  int32_t* cq; int64_t* priority; int64_t* queue_ts; uint32_t* flags; int32_t* ps_off;
  int32_t *ps_count, *ps_min_count, *ps_req_off, *req_res; int64_t* req_qty; uint64_t* ps_flavor_ok; int32_t* ps_last_tried;
  int64_t *last_generation, *last_cycle; uint64_t *last_hash, *hash;
  // workload slices (null unless the resident set holds the columns): kq_heads slice_row ... ps_slice_pods_qty
  int32_t *slice_row, *ps_slice_count, *req_slice_flavor, *ps_slice_pods_flavor; int64_t *req_slice_qty, *ps_slice_pods_qty;
  int32_t* ps_group;   // kq_heads.ps_group of the batch (PodSetGroupName groups), null while no resident workload has one
};

// ClusterQueue.Pop (cluster_queue.go:657-672) for ClusterQueue c — one wave
KQ_DEV void pend_pop(const DPend& D, int c) {
  const int lane = lane_id();
  if (D.cq_active && !D.cq_active[c]) { if (lane == 0) D.head_wl[c] = -1; return; }  // manager.go:926: no Pop at all
  int head = -1;
  const int pw = D.pw[c];
  const bool sticky = pw >= 0 && (D.pw_sticky[c] & 1) && D.state[pw] == WL_ACTIVE;
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
    if (D.pw[c] == w && !(D.pw_sticky[c] & 2)) fl |= KQ_HEAD_IS_PREEMPTOR;  // IsPreemptor :213 (strict: the generation the pointer was set with — kq_pending_update marks a change)
    G.flags[h] = fl;
    G.last_generation[h] = D.last_gen[w]; G.last_cycle[h] = D.last_cycle[w]; G.last_hash[h] = D.last_hash[w]; G.hash[h] = D.P.hash[w];
  }
  for (int i = lane; i < nps; i += WAVE) {
    G.ps_count[gp0 + i] = D.P.ps_count[p0 + i]; G.ps_min_count[gp0 + i] = D.P.ps_min_count[p0 + i];
    if (G.ps_group) G.ps_group[gp0 + i] = D.P.ps_group[p0 + i];
    G.ps_req_off[gp0 + i] = gr0 + (D.P.ps_req_off[p0 + i] - r0);
  }
  for (int i = lane; i < nreq; i += WAVE) { G.req_res[gr0 + i] = D.P.req_res[r0 + i]; G.req_qty[gr0 + i] = D.P.req_qty[r0 + i]; }
  for (int i = lane; i < nps * D.nfw; i += WAVE) G.ps_flavor_ok[(size_t)gp0 * D.nfw + i] = D.P.ps_flavor_ok[(size_t)p0 * D.nfw + i];
  for (int i = lane; i < nps * D.nR; i += WAVE) G.ps_last_tried[(size_t)gp0 * D.nR + i] = D.last_tried[(size_t)p0 * D.nR + i];
  if (G.slice_row) {   // ElasticJobsViaWorkloadSlices: the slice the head replaces and what that slice holds (workloadslicing.go:371)
    if (lane == 0) G.slice_row[h] = D.P.slice_row[w];
    for (int i = lane; i < nps; i += WAVE) {
      G.ps_slice_count[gp0 + i] = D.P.ps_slice_count[p0 + i];
      G.ps_slice_pods_flavor[gp0 + i] = D.P.ps_slice_pods_flavor[p0 + i]; G.ps_slice_pods_qty[gp0 + i] = D.P.ps_slice_pods_qty[p0 + i];
    }
    for (int i = lane; i < nreq; i += WAVE) { G.req_slice_flavor[gr0 + i] = D.P.req_slice_flavor[r0 + i]; G.req_slice_qty[gr0 + i] = D.P.req_slice_qty[r0 + i]; }
  }
}

// Step 6 of schedule() for head h (scheduler.go:362-377): the queue side of requeueAndUpdate :1165 — or, for an admitted entry,
// the workload leaving the queue (assumeWorkload; the controller's delete, cluster_queue.go:495). One wave per head.
KQ_DEV void pend_apply_head(const DPend& D, const DSnap& S, const DOut& O, const DHeads& H, uint32_t gates, int64_t cycle, int h) {
  const int lane = lane_id();
  if (h >= D.counts[0]) return;   // the grid is sized by a bound when the count never left the device (kq_pending_step)
  const int w = D.hd[h];
  const int c = D.P.cq[w];
  if (O.error && O.error[0] != 0) {  // the cycle failed on the device (capacity): its heads go back to the heap as if they had not been popped
    if (lane == 0) D.state[w] = WL_ACTIVE;
    return;
  }
  const int status = O.status[h], action = O.action[h], mode = O.mode[h], rq = O.requeue_reason[h];
  if (status == KQ_ST_ASSUMED) {
    if (lane == 0) {
      D.state[w] = WL_GONE; if (D.pw[c] == w) { D.pw[c] = -1; D.pw_sticky[c] = 0; }
      // assumeWorkload -> updateEntryPenalty(add) (scheduler.go:1064-1068, :1337-1350) when shouldApplyEntryPenalty (:1318-1335):
      // a ledger is resident, the ClusterQueue admits usage-based (the workload has a LocalQueue row), first reservation
      if (D.A.n_res > 0 && D.lq && D.lq[w] >= 0 && !(D.P.flags[w] & KQ_HEAD_HAS_QUOTA_RESERVATION)) {
        afs_push(D.A, D.lq[w], w);
        afs_usage_lq(D.A, D.lq[w]);   // <= 1 head per ClusterQueue and a LocalQueue feeds one ClusterQueue: nobody else writes this row
      }
    }
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
// ---- kq_pending_add: the arrivals merged into the heap orders on the device --------------------------------------------------------------
// baseCompareFunc's static part (cluster_queue.go:844-878 without the sticky term) inside one ClusterQueue: does workload a sort before b?
KQ_DEV bool pend_before(const DPend& D, int a, int b) {
  const int64_t pa = D.P.priority[a], pb = D.P.priority[b];
  if (pa != pb) return pa > pb;
  const int64_t ta = D.P.queue_ts[a], tb = D.P.queue_ts[b];
  if (ta != tb) return ta < tb;
  const uint32_t ua = D.uid[a], ub = D.uid[b];
  if (ua != ub) return ua < ub;
  return a < b;
}
// first position in list[lo, hi) (sorted by pend_before) whose workload does NOT sort before w = how many of them precede w
KQ_DEV int pend_rank_in(const DPend& D, const int32_t* list, int lo, int hi, int w) {
  int a = lo, b = hi;
  while (a < b) { const int m = (a + b) >> 1; if (pend_before(D, list[m], w)) a = m + 1; else b = m; }
  return a - lo;
}
// resident workload at heap position j: it keeps its place among the resident ones and moves back by the arrivals of earlier
// ClusterQueues (fresh_off[c]) and by those of its own ClusterQueue that sort before it
KQ_DEV void pend_merge_old(const DPend& D, const int32_t* ord_old, int32_t* ord_new, const int32_t* fresh, const int32_t* fresh_off, int j) {
  const int w = ord_old[j], c = D.P.cq[w];
  ord_new[j + fresh_off[c] + pend_rank_in(D, fresh, fresh_off[c], fresh_off[c + 1], w)] = w;
}
// arrival r of the sorted list: behind the resident workloads of its ClusterQueue that sort before it; r itself counts the arrivals in front
KQ_DEV void pend_merge_new(const DPend& D, const int32_t* ord_old, const int32_t* off_old, int32_t* ord_new, const int32_t* fresh, int r) {
  const int w = fresh[r], c = D.P.cq[w];
  ord_new[off_old[c] + pend_rank_in(D, ord_old, off_old[c], off_old[c + 1], w) + r] = w;
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
// PushOrUpdate :379-428 of a key that IS pending, with a new object: the replacement (index first + i) was appended and placed like an
// arrival (pend_add_fix); here the old record's part — one thread per workload:
//   old in the heap          PushOrUpdateActive :427 — the back-off (:414) and the bulk-moved classes (:421) only count when GetActive(key) == nil
//   old inadmissible         RemoveFromInadmissible :405, then what an arrival gets (already there)
//   gone already             a plain arrival
// The preemptor pointer holds a NAME (:109): it follows the key, stickyMatches (:124) still sees it, the strict match of IsPreemptor
// (:213, generation) does not. (AdmissionFairSharing: a penalty record only exists once a workload is assumed, i.e. gone from here; the
// replacement's entry-penalty amounts come through kq_pending_afs_wl_penalty like an arrival's.) Then the old record leaves.
KQ_DEV void pend_update_fix(const DPend& D, const int32_t* list, const uint8_t* same_gen, int first, int i) {
  const int old = list[i], w2 = first + i;
  const int c = D.P.cq[old];
  const bool same = D.P.cq[w2] == c;
  const uint8_t st = D.state[old];
  if (st == WL_GONE) return;
  if (same && st == WL_ACTIVE) D.state[w2] = WL_ACTIVE;
  if (D.pw[c] == old) {
    if (same) { D.pw[c] = w2; if (!(same_gen && same_gen[i])) D.pw_sticky[c] |= 2; }   // IsPreemptor :213 compares Obj.Generation
    else { D.pw[c] = -1; D.pw_sticky[c] = 0; }
  }
  D.state[old] = WL_GONE;
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

// kq_pending_oracle.cpp — CPU ORACLE for the pending side (pkg/cache/queue), the checker of kueue_amd/csrc/kq_pending.hpp.
//
// *** TEST INFRASTRUCTURE ONLY. ***  A single-threaded C++ restatement of the reference's ClusterQueue heap and requeue
// policy. Only tests/ and bench.py's cpu_baseline leg load it; the product never does.
//
// Parity status: PINNED BY TRANSCRIPTION (no Go toolchain here) — tests/golden/pending_queue.yaml holds the cases of the
// reference's own tests for these functions (cluster_queue_test.go: TestBestEffortFIFORequeueIfNotPresent :1380,
// TestStrictFIFORequeueIfNotPresent :1665, TestStrictFIFO :1546, TestQueueInadmissibleWorkloadsDuringScheduling :1285,
// TestRecordInadmissibleHash :1880, TestRequeueHashTriggerByReason :2043, TestClusterQueueImpl :1084), replayed op by op.
//
// Each function cites the reference file:line it follows (paths relative to /root/reference/pkg).
// Outside the boundary, as in the engine: RequeueState back-off (backoffWaitingTimeExpired is true), namespace selectors,
// PushOrUpdate against hashToBulkMoveReason for workloads arriving later, AdmissionFairSharing ordering.
#include "../include/kq_engine.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <vector>

namespace kqp {

struct WL {
  int cq; int64_t prio, ts; uint32_t uid; uint64_t hash;
  int state;  // KQ_WL_*
  // LastAssignment (workload.AssignmentClusterQueueState, workload.go:115-135)
  bool has_last; std::vector<int32_t> last_tried; int64_t last_gen, last_cycle; uint64_t last_hash;
  int ps0, nps;
  int lq;  // LocalQueue index, -1 = the ClusterQueue orders by baseCompareFunc only
  int64_t requeue_at = KQ_REQUEUE_NONE;  // RequeueState.RequeueAt (ns); KQ_REQUEUE_BLOCKED while the Requeued condition is False
};

struct CQ {
  bool strict;                       // queueingStrategy == StrictFIFO
  std::vector<int> members;          // every workload ever pushed (heap U inadmissible U inflight U gone)
  int pw = -1; bool pw_sticky = false;  // preemptorWorkload cluster_queue.go:80-154
  bool pw_gen_changed = false;          // the workload's object was replaced since set(): matches(strict) fails on the generation (:113-121)
  int64_t popCycle = 0, queueInadmissibleCycle = -1;  // :176-183, :313
  std::set<uint64_t> hashToBulkMoveReason;             // :169-172 (reasons themselves are status text)
};

struct Queue {
  int nR = 0; uint32_t gates = 0;
  std::vector<WL> wl;
  std::vector<CQ> cqs;
  std::vector<int> head_wl;          // heads of the cycle in flight, per ClusterQueue
  std::vector<double> lq_usage;      // ComputeLocalQueueFSUsage per LocalQueue (workload.go:492), host-evaluated
  int64_t now = 0;                   // c.clock.Now()
  // backoffWaitingTimeExpired cluster_queue.go:474-485
  bool backoffExpired(const WL& x) const {
    if (x.requeue_at == KQ_REQUEUE_BLOCKED) return false;
    if (x.requeue_at == KQ_REQUEUE_NONE) return true;
    return now >= x.requeue_at;
  }
  // Go cmp.Compare on float64: NaN < everything, -0 == +0
  static int cmpF(double a, double b) {
    const bool an = a != a, bn = b != b;
    if (an && bn) return 0;
    if (an) return -1;
    if (bn) return 1;
    return a < b ? -1 : (a > b ? 1 : 0);
  }

  // baseCompareFunc cluster_queue.go:844-876 : a before b
  bool before(const CQ& c, int a, int b) const {
    if (!lq_usage.empty()) {  // queueOrderingFunc cluster_queue.go:880-904 (enableAdmissionFs): usage first, baseCmp only on a tie
      const double ua = wl[a].lq >= 0 ? lq_usage[wl[a].lq] : 0.0, ub = wl[b].lq >= 0 ? lq_usage[wl[b].lq] : 0.0;
      const int c3 = cmpF(ua, ub);
      if (c3 != 0) return c3 < 0;
    }
    const bool as = c.pw_sticky && c.pw == a, bs = c.pw_sticky && c.pw == b;  // baseCompareFunc :848-856
    if (as != bs) return as;
    if (wl[a].prio != wl[b].prio) return wl[a].prio > wl[b].prio;
    if (wl[a].ts != wl[b].ts) return wl[a].ts < wl[b].ts;
    if (wl[a].uid != wl[b].uid) return wl[a].uid < wl[b].uid;
    return a < b;
  }
  // ClusterQueue.Pop :657-672 (the heap's minimum under baseCompareFunc)
  int Pop(int ci) {
    CQ& c = cqs[ci];
    c.popCycle++;
    int best = -1;
    for (int w : c.members) if (wl[w].state == KQ_WL_ACTIVE && (best < 0 || before(c, w, best))) best = w;
    if (best >= 0) wl[best].state = KQ_WL_INFLIGHT;
    return best;
  }
  bool IsPreemptor(int w) const { const CQ& c = cqs[wl[w].cq]; return c.pw == w && !c.pw_gen_changed; }  // :213 (strict match: name and generation)
  // handleInadmissibleHash :606-621
  int handleInadmissibleHash(CQ& c, uint64_t hash) {
    if (c.strict) return 0;
    c.hashToBulkMoveReason.insert(hash);
    int moved = 0;
    for (int w : c.members) if (wl[w].state == KQ_WL_ACTIVE && wl[w].hash == hash) { wl[w].state = KQ_WL_INADMISSIBLE; moved++; }
    return moved;
  }
  bool PendingFlavors(const WL& w) const {  // workload.go:211-224
    if (!w.has_last) return false;
    for (int32_t t : w.last_tried) if (t != -1) return true;
    return false;
  }
  // requeueIfNotPresent :550-601
  bool requeueIfNotPresent(int w, bool immediate, int reason) {
    WL& x = wl[w]; CQ& c = cqs[x.cq];
    if (reason == KQ_RQ_PENDING_PREEMPTION) { c.pw = w; c.pw_sticky = !c.strict; c.pw_gen_changed = false; }  // :558-563
    const bool inadm = x.state == KQ_WL_INADMISSIBLE;
    if (backoffExpired(x) && (immediate || c.queueInadmissibleCycle >= c.popCycle || PendingFlavors(x))) {  // :568-575
      if (x.state == KQ_WL_ACTIVE) return false;  // PushActiveIfNotPresent
      x.state = KQ_WL_ACTIVE;
      return true;
    }
    if (inadm) return false;                    // :577
    if (x.state == KQ_WL_ACTIVE) return false;  // :581
    x.state = KQ_WL_INADMISSIBLE;               // :585
    if ((gates & KQ_GATE_SCHEDULING_EQUIVALENCE_HASHING) && x.hash != 0 && (reason == KQ_RQ_NOFIT || reason == KQ_RQ_PREEMPTION_NO_CANDIDATES))
      handleInadmissibleHash(c, x.hash);        // :592-597
    return true;
  }
  // RequeueIfNotPresent :826-841 (NamespaceMismatch / PendingMigration / PreemptionFailed do not cross this boundary)
  bool RequeueIfNotPresent(int w, int reason) {
    const CQ& c = cqs[wl[w].cq];
    bool immediate;
    if (c.strict) immediate = true;
    else immediate = reason == KQ_RQ_FAILED_AFTER_NOMINATION || reason == KQ_RQ_PENDING_PREEMPTION;
    return requeueIfNotPresent(w, immediate, reason);
  }
  // delete :495-512
  void Delete(int w) {
    CQ& c = cqs[wl[w].cq];
    wl[w].state = KQ_WL_GONE;
    if (c.pw == w) { c.pw = -1; c.pw_sticky = false; c.pw_gen_changed = false; }
  }
  // queueInadmissibleWorkloads inadmissible_workloads.go:149-175
  int queueInadmissibleWorkloads(int ci) {
    CQ& c = cqs[ci];
    c.queueInadmissibleCycle = c.popCycle;
    c.hashToBulkMoveReason.clear();
    int moved = 0;
    for (int w : c.members) if (wl[w].state == KQ_WL_INADMISSIBLE && backoffExpired(wl[w])) { wl[w].state = KQ_WL_ACTIVE; moved++; }  // :167
    return moved;
  }
  // PushOrUpdate :379-428 of workload w with its present columns: where does it go?
  void place(int w) {
    WL& x = wl[w]; CQ& c = cqs[x.cq];
    if (!backoffExpired(x)) { x.state = KQ_WL_INADMISSIBLE; return; }                                               // :414
    if (!c.strict && x.hash != 0 && c.hashToBulkMoveReason.count(x.hash)) { x.state = KQ_WL_INADMISSIBLE; return; }  // :419-425
    x.state = KQ_WL_ACTIVE;
  }
};

}  // namespace kqp

using namespace kqp;

extern "C" {

// PushOrUpdate (cluster_queue.go:379) of every workload; cq_policy supplies the queueing strategy per ClusterQueue
void* kqp_create(const kq_pending* p, int32_t n_cq, int32_t n_resource, const uint32_t* cq_policy, uint32_t gates) {
  Queue* q = new Queue();
  q->nR = n_resource; q->gates = gates;
  q->cqs.resize(n_cq);
  for (int c = 0; c < n_cq; c++) q->cqs[c].strict = KQ_POL_STRICT_FIFO(cq_policy[c]) != 0;
  const kq_heads& h = p->w;
  q->wl.resize(h.n);
  for (int w = 0; w < h.n; w++) {
    WL& x = q->wl[w];
    x.cq = h.cq[w]; x.prio = h.priority[w]; x.ts = h.queue_ts[w]; x.uid = p->uid_rank ? p->uid_rank[w] : (uint32_t)w;
    x.hash = h.hash ? h.hash[w] : 0; x.state = KQ_WL_ACTIVE;
    x.ps0 = h.ps_off[w]; x.nps = h.ps_off[w + 1] - h.ps_off[w];
    x.has_last = (h.flags[w] & KQ_HEAD_HAS_LAST_ASSIGNMENT) != 0;
    x.last_tried.assign((size_t)x.nps * n_resource, -1);
    if (h.ps_last_tried) for (size_t i = 0; i < x.last_tried.size(); i++) x.last_tried[i] = h.ps_last_tried[(size_t)x.ps0 * n_resource + i];
    x.last_gen = h.last_generation ? h.last_generation[w] : 0; x.last_cycle = h.last_cycle ? h.last_cycle[w] : 0;
    x.last_hash = h.last_hash ? h.last_hash[w] : 0;
    x.lq = (p->lq && p->n_lq > 0) ? p->lq[w] : -1;
    x.requeue_at = p->requeue_at ? p->requeue_at[w] : KQ_REQUEUE_NONE;
    q->cqs[x.cq].members.push_back(w);
    q->place(w);
  }
  q->head_wl.assign(n_cq, -1);
  if (p->lq && p->n_lq > 0) q->lq_usage.assign(p->n_lq, 0.0);
  return q;
}
void kqp_destroy(void* q) { delete (Queue*)q; }

// PushOrUpdate (cluster_queue.go:379-428) of workloads that were not pending before; returns the index of the first one.
// A new workload goes to the heap unless its ClusterQueue is BestEffortFIFO, its hash is known and the class was bulk-moved (:419-425).
int kqp_add(void* qp, const kq_pending* p) {
  Queue& q = *(Queue*)qp;
  const kq_heads& h = p->w;
  const int first = (int)q.wl.size();
  int ps_base = 0;
  for (const WL& x : q.wl) ps_base = std::max(ps_base, x.ps0 + x.nps);
  for (int i = 0; i < h.n; i++) {
    WL x;
    x.cq = h.cq[i]; x.prio = h.priority[i]; x.ts = h.queue_ts[i]; x.uid = p->uid_rank ? p->uid_rank[i] : (uint32_t)(first + i);
    x.hash = h.hash ? h.hash[i] : 0;
    x.ps0 = ps_base + h.ps_off[i]; x.nps = h.ps_off[i + 1] - h.ps_off[i];
    x.has_last = (h.flags[i] & KQ_HEAD_HAS_LAST_ASSIGNMENT) != 0;
    x.last_tried.assign((size_t)x.nps * q.nR, -1);
    if (h.ps_last_tried) for (size_t j = 0; j < x.last_tried.size(); j++) x.last_tried[j] = h.ps_last_tried[(size_t)h.ps_off[i] * q.nR + j];
    x.last_gen = h.last_generation ? h.last_generation[i] : 0; x.last_cycle = h.last_cycle ? h.last_cycle[i] : 0;
    x.last_hash = h.last_hash ? h.last_hash[i] : 0;
    x.lq = (p->lq && p->n_lq > 0) ? p->lq[i] : -1;
    x.requeue_at = p->requeue_at ? p->requeue_at[i] : KQ_REQUEUE_NONE;
    x.state = KQ_WL_ACTIVE;
    q.wl.push_back(x);
    q.cqs[x.cq].members.push_back(first + i);
    q.place(first + i);
  }
  return first;
}

// PushOrUpdate (cluster_queue.go:379-428) of keys that ARE pending, each with a new object (the i-th workload of `more` replaces wl[i];
// a record is immutable here, so the replacement is a new index and the old one leaves). Returns the index of the first replacement.
int kqp_update(void* qp, int32_t n, const int32_t* wl, const kq_pending* more) {
  Queue& q = *(Queue*)qp;
  std::vector<int> was(n);
  for (int i = 0; i < n; i++) was[i] = q.wl[wl[i]].state;
  const int first = kqp_add(qp, more);   // what a key with GetActive(key) == nil gets: back-off :414, bulk-moved class :419-425, else the heap
  for (int i = 0; i < n; i++) {
    const int old = wl[i], w2 = first + i;
    if (was[i] == KQ_WL_GONE) continue;   // not pending any more: a plain arrival
    CQ& c = q.cqs[q.wl[old].cq];
    const bool same = q.wl[w2].cq == q.wl[old].cq;
    // in the heap: c.workloads.GetActive(key) != nil, so neither :414 nor :421 applies -> PushOrUpdateActive :427.
    // inadmissible: RemoveFromInadmissible :405 and on as above. (in flight :388 does not occur between cycles.)
    if (was[i] == KQ_WL_ACTIVE && same) q.wl[w2].state = KQ_WL_ACTIVE;
    if (c.pw == old) {
      // the pointer is a name (:109): stickyMatches :124 still; IsPreemptor :213 also wants the Generation the pointer was set with
      if (same) { c.pw = w2; if (!(more->same_generation && more->same_generation[i])) c.pw_gen_changed = true; }
      else { c.pw = -1; c.pw_sticky = false; c.pw_gen_changed = false; }   // left this ClusterQueue: Delete :506
    }
    q.wl[old].state = KQ_WL_GONE;
  }
  return first;
}

// manager.heads :922-947 in canonical ClusterQueue order. head_wl[n_cq]: popped workload or -1; returns the number of heads.
// For every head also: its flags / LastAssignment as the scheduler will see them (the caller builds the kq_heads batch).
int kqp_heads(void* qp, const uint8_t* cq_active, int32_t* head_wl) {
  Queue& q = *(Queue*)qp;
  int n = 0;
  for (size_t c = 0; c < q.cqs.size(); c++) {
    int w = -1;
    if (!cq_active || cq_active[c]) w = q.Pop((int)c);
    q.head_wl[c] = w; head_wl[c] = w;
    if (w >= 0) n++;
  }
  return n;
}
// per-workload view for building the heads batch: flags (HAS_LAST / IS_PREEMPTOR), last_* ; last_tried rows [nps * nR]
void kqp_workload(void* qp, int32_t w, uint32_t* flags_io, int32_t* last_tried, int64_t* last_gen, int64_t* last_cycle, uint64_t* last_hash) {
  Queue& q = *(Queue*)qp;
  const WL& x = q.wl[w];
  uint32_t f = *flags_io & ~(uint32_t)(KQ_HEAD_HAS_LAST_ASSIGNMENT | KQ_HEAD_IS_PREEMPTOR);
  if (x.has_last) f |= KQ_HEAD_HAS_LAST_ASSIGNMENT;
  if (q.IsPreemptor(w)) f |= KQ_HEAD_IS_PREEMPTOR;
  *flags_io = f;
  for (size_t i = 0; i < x.last_tried.size(); i++) last_tried[i] = x.has_last ? x.last_tried[i] : -1;
  *last_gen = x.last_gen; *last_cycle = x.last_cycle; *last_hash = x.last_hash;
}
// Step 6 of schedule() (scheduler.go:362-377) for the heads of the cycle: d = the cycle's decisions over the heads batch built
// from kqp_heads' pops (canonical order), h = that batch.
int kqp_apply(void* qp, const kq_heads* h, const kq_decisions* d, const int64_t* cq_generation) {
  Queue& q = *(Queue*)qp;
  int hi = 0;
  for (size_t c = 0; c < q.cqs.size(); c++) {
    const int w = q.head_wl[c];
    if (w < 0) continue;
    if (hi >= h->n || h->cq[hi] != (int)c) return KQ_EINVAL;
    WL& x = q.wl[w];
    const int status = d->status[hi], action = d->action[hi], mode = d->mode[hi], rq = d->requeue_reason[hi];
    if (status == KQ_ST_ASSUMED) { q.Delete(w); hi++; continue; }  // admitted: the workload leaves the queue
    // recordAssignment scheduler.go:281 ... cleared by markPreemptionOutcome :291, DeferredFit :459-464, markSkipped :248-254
    const bool nil_last = action == KQ_ACT_PREEMPT || mode == KQ_MODE_DEFERRED_FIT ||
                          (status == KQ_ST_SKIPPED && !(q.gates & KQ_GATE_PRESERVE_SCAN_PROGRESS));
    x.has_last = !nil_last;
    const int gp0 = h->ps_off[hi];
    for (size_t i = 0; i < x.last_tried.size(); i++) x.last_tried[i] = nil_last ? -1 : d->tried_idx[(size_t)gp0 * q.nR + i];
    if (!nil_last) { x.last_gen = cq_generation[c]; x.last_cycle = h->cycle; x.last_hash = x.hash; }
    q.RequeueIfNotPresent(w, rq);  // requeueAndUpdate :1179
    hi++;
  }
  for (auto& v : q.head_wl) v = -1;
  return hi == h->n ? KQ_OK : KQ_EINVAL;
}
int kqp_queue_inadmissible(void* qp, int32_t n, const int32_t* cq) {
  Queue& q = *(Queue*)qp;
  int moved = 0;
  if (!cq) { for (size_t c = 0; c < q.cqs.size(); c++) moved += q.queueInadmissibleWorkloads((int)c); }
  else for (int i = 0; i < n; i++) moved += q.queueInadmissibleWorkloads(cq[i]);
  return moved;
}
void kqp_set_lq_usage(void* qp, int32_t n, const double* usage) { Queue& q = *(Queue*)qp; for (int i = 0; i < n && i < (int)q.lq_usage.size(); i++) q.lq_usage[i] = usage[i]; }
void kqp_read_state(void* qp, uint8_t* state) { Queue& q = *(Queue*)qp; for (size_t w = 0; w < q.wl.size(); w++) state[w] = (uint8_t)q.wl[w].state; }

// ---- single operations, for the transcribed unit tests of the reference ----
int kqp_pop(void* qp, int32_t cq) { return ((Queue*)qp)->Pop(cq); }
int kqp_requeue(void* qp, int32_t w, int32_t reason, int32_t immediate_override /* -1: RequeueIfNotPresent; 0/1: requeueIfNotPresent(immediate) */) {
  Queue& q = *(Queue*)qp;
  return immediate_override < 0 ? q.RequeueIfNotPresent(w, reason) : q.requeueIfNotPresent(w, immediate_override != 0, reason);
}
void kqp_set_last(void* qp, int32_t w, int32_t has_last, const int32_t* last_tried) {
  Queue& q = *(Queue*)qp;
  q.wl[w].has_last = has_last != 0;
  for (size_t i = 0; i < q.wl[w].last_tried.size(); i++) q.wl[w].last_tried[i] = last_tried ? last_tried[i] : -1;
}
void kqp_set_state(void* qp, int32_t w, int32_t state) { ((Queue*)qp)->wl[w].state = state; }
void kqp_delete(void* qp, int32_t w) { ((Queue*)qp)->Delete(w); }
void kqp_set_clock(void* qp, int64_t now) { ((Queue*)qp)->now = now; }
// the controller changed RequeueState / the Requeued condition: PushOrUpdate of an existing workload whose conditions changed (:391-428)
void kqp_set_requeue_at(void* qp, int32_t n, const int32_t* wl, const int64_t* at) {
  Queue& q = *(Queue*)qp;
  for (int i = 0; i < n; i++) {
    WL& x = q.wl[wl[i]];
    x.requeue_at = at[i];
    if (x.state != KQ_WL_INADMISSIBLE) continue;  // in flight: skipped (:388); in the heap: PushOrUpdateActive
    q.place(wl[i]);
  }
}
void kqp_delete_list(void* qp, int32_t n, const int32_t* wl) { for (int i = 0; i < n; i++) ((Queue*)qp)->Delete(wl[i]); }
int kqp_handle_hash(void* qp, int32_t cq, uint64_t hash) { Queue& q = *(Queue*)qp; return hash ? q.handleInadmissibleHash(q.cqs[cq], hash) : 0; }
int kqp_is_sticky(void* qp, int32_t w) { Queue& q = *(Queue*)qp; const CQ& c = q.cqs[q.wl[w].cq]; return c.pw == w && c.pw_sticky; }
int kqp_has_hash(void* qp, int32_t cq, uint64_t hash) { return ((Queue*)qp)->cqs[cq].hashToBulkMoveReason.count(hash) ? 1 : 0; }

}  // extern "C"

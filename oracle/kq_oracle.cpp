// kq_oracle.cpp — CPU ORACLE for the kq_engine hot path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  This file is a single-threaded C++ restatement of the
// reference's Go algorithm (kubernetes-sigs/kueue, /root/reference) for the scheduling-cycle
// decision path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// it. The product path (kueue_amd/csrc) never links, loads or calls anything in this directory.
//
// Parity status: PINNED BY TRANSCRIPTION — the Go reference cannot be built here (no Go
// toolchain); this restatement is checked against golden vectors transcribed from the
// reference's own unit tests (tests/golden/*.json, see tests/golden/README.md for file:line).
//
// Each function cites the reference file:line it follows (paths relative to /root/reference/pkg).
// Determinism contract: Go map iteration / unstable sorts are replaced by the
// canonical orders stated next to each use.
#include "../include/kq_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../include/kq_cycle_tas.h"
#include "kq_tas_oracle.hpp"

namespace kqo {

static const int64_t I64MAX = std::numeric_limits<int64_t>::max();
static const int64_t I64MIN = std::numeric_limits<int64_t>::min();

// ---- util/math/math.go:27,39 -------------------------------------------------------------------
static inline int64_t SaturatingAdd(int64_t a, int64_t b) {
  if (b > 0 && a > I64MAX - b) return I64MAX;
  if (b < 0 && a < I64MIN - b) return I64MIN;
  return a + b;
}
static inline int64_t SaturatingSub(int64_t a, int64_t b) {
  if (b < 0 && a > I64MAX + b) return I64MAX;
  if (b > 0 && a < I64MIN + b) return I64MIN;
  return a - b;
}
// util/math/math.go:89-107
static inline int64_t SaturatingMul(int64_t a, int64_t b) {
  if (a == 0 || b == 0) return 0;
  if ((a == -1 && b == I64MIN) || (b == -1 && a == I64MIN)) return I64MAX;
  int64_t res = (int64_t)((uint64_t)a * (uint64_t)b);
  if (res / b != a) return ((a < 0) == (b < 0)) ? I64MAX : I64MIN;
  return res;
}

// ---- resources/amount.go:46-205 ------------------------------------------------------------------
struct Amount {
  int64_t v = 0;
  Amount() = default;
  explicit Amount(int64_t x) : v(x) {}
  bool isUnlimited() const { return v == I64MAX; }
  Amount Add(Amount b) const {  // :114
    if (isUnlimited() || b.isUnlimited()) return Amount(I64MAX);
    return Amount(SaturatingAdd(v, b.v));
  }
  Amount AddInt64(int64_t x) const {  // :122
    if (isUnlimited()) return *this;
    return Amount(SaturatingAdd(v, x));
  }
  Amount Sub(Amount b) const {  // :133
    if (isUnlimited() && b.isUnlimited()) return Amount(0);
    if (isUnlimited()) return Amount(I64MAX);
    if (b.isUnlimited()) return Amount(I64MIN);
    return Amount(SaturatingSub(v, b.v));
  }
  Amount SubInt64(int64_t x) const {  // :147
    if (isUnlimited()) return *this;
    return Amount(SaturatingSub(v, x));
  }
  int Cmp(Amount b) const {  // :156
    if (isUnlimited() && b.isUnlimited()) return 0;
    if (isUnlimited()) return 1;
    if (b.isUnlimited()) return -1;
    return v < b.v ? -1 : (v > b.v ? 1 : 0);
  }
  int CmpInt64(int64_t x) const {  // :174
    if (isUnlimited()) return 1;
    return v < x ? -1 : (v > x ? 1 : 0);
  }
};
static inline Amount MinAmount(Amount a, Amount b) { return a.Cmp(b) < 0 ? a : b; }  // :189
static inline Amount MaxAmount(Amount a, Amount b) { return a.Cmp(b) > 0 ? a : b; }  // :197

// FlavorResourceQuantities (resources/resource.go): map[FlavorResource]Amount. Canonical iteration
// order = ascending fr index.
typedef std::map<int, Amount> FRQ;
static inline Amount frq_get(const FRQ& m, int fr) {
  auto it = m.find(fr);
  return it == m.end() ? Amount(0) : it->second;
}

// ---- modes ---------------------------------------------------------------------------------------
enum { NoFit = 0, Preempt = 1, DeferredFit = 2, Fit = 3 };                     // flavorassigner.go:457
enum { pmNoFit = 0, pmNoCandidates = 1, pmPreempt = 2, pmReclaim = 3, pmFit = 4 };  // :523-531
enum { ppNoCandidates = 0, ppPreempt = 1, ppReclaim = 2 };                     // common/types.go:23
struct GranularMode {
  int pm;
  int64_t borrow;
};
static const int64_t MAXINT = I64MAX;  // math.MaxInt on 64-bit

// ---- probe of runFirstFsStrategy (tools/fs_pop_probe.py; VERDICT r04 item 1 "measure first"): histograms over every victim search
// of the calls made on this thread while it is enabled. Layout (int64 each):
//   [0] searches  [1] pops  [2] failed pops (-> retryCandidates)  [3] victims  [4] ClusterQueue visits (nextTarget results)
//   [5] visits ending with a victim  [6] visits exhausted without one  [7] visits skipped by fsStrategyUnsatisfiable
//   [8] unconditional victims (own ClusterQueue / within nominal)
//   [9] consecutive victims: same ClusterQueue  [10] same parent cohort  [11] same child of the root  [12] pairs counted
//   [13] consecutive nextTarget results: same parent cohort  [14] same child of the root  [15] pairs counted
//   [16] victims whose removal changes a usage cell on the preemptor's path  [17] victims that do not
//   [18..21] victims by height of the LCA above the target ClusterQueue (1 = its parent .. 3; 0 = own ClusterQueue)
//   [32..95]  failed pops in front of the victim of a visit (index = count, last bin = 63+)
//   [96..159] failed pops of a visit that ended exhausted
//   [160..223] length of a run of consecutive victims under the same child of the root (last bin 63+)
//   [224..287] candidates left in the ClusterQueue at the moment it is visited
static thread_local bool g_fs_probe_on = false;
static thread_local int64_t g_fs_probe[288];

// ---- bytes accounting (SURVEY §8d "algorithmic bytes") ------------------------------------------
struct Stats {
  int64_t cells = 0;         // fitsResourceQuota evaluations
  int64_t cell_bytes = 0;    // 40*(D+1) per cell
  int64_t head_io_bytes = 0; // requests in + assignment out + usage out
  int64_t entry_bytes = 0;   // fits re-check + addUsage
  int64_t victim_bytes = 0;  // candidate scan + simulated removals
  int64_t drs_bytes = 0;
  // victim/drs bytes of SimulatePreemption calls whose result the reference discards (the flavor was
  // already noFit, flavorassigner.go:1161): the engine does not run those searches at all.
  int64_t discarded_bytes = 0;
  // probe counters (kqo_jacobi_probe): victim searches, candidates they listed, rows they removed before the fill-back, fair pops
  int64_t searches = 0, cand_listed = 0, cand_removed = 0, fair_pops = 0;
  // what TestFairPreemptionSkipsUnsatisfiableTournament (preemption_fair_test.go:1194) counts through the V(4) log lines: target
  // ClusterQueues whose candidates the first strategy simulated (one collapsed entry per ClusterQueue, preemption.go:425-470), second
  // strategy evaluations (:512), ClusterQueues skipped by fsStrategyUnsatisfiable (:494)
  int64_t fs_first_cq_evals = 0, fs_second_evals = 0, fs_skipped_queues = 0;
  int64_t total() const { return cell_bytes + head_io_bytes + entry_bytes + victim_bytes + drs_bytes - discarded_bytes; }
};

// ---- the snapshot as the scheduler sees it -------------------------------------------------------
struct Snap {
  kq_config cfg;
  const kq_snapshot* s;
  int nq, nc, N, nF, nR, nfr;
  std::vector<int64_t> usage;    // mutable Usage planes (the per-cycle copy, snapshot.go:171)
  std::vector<uint8_t> removed;  // admitted row deleted from cq.Workloads (snapshot.go:62)
  std::vector<int> adm_cq;
  std::vector<int> depth;        // #ancestors
  Stats st;
  // Topology-Aware Scheduling (optional): one TASFlavorSnapshot per TAS flavor, shared by the ClusterQueues (snapshot.go:260)
  const kq_cycle_tas* T = nullptr;
  std::vector<tas::Snapshot*> tasS;
  std::vector<int> tasOfFlavor;  // flavor -> index into T->tas_flavor, -1 = not a TAS flavor
  int tasR = 0;
  bool tasUnsupported = false;
  int64_t tasFinds = 0, tasRecomputes = 0;
  void attachTAS(const kq_cycle_tas* t) {
    T = t;
    tasOfFlavor.assign(nF, -1);
    for (int i = 0; i < t->n_tas; i++) { tasS.push_back(tas::snapshot_new(&t->topo[i])); tasOfFlavor[t->tas_flavor[i]] = i; tasR = t->topo[i].n_resources; }
    // topo[i].tas_usage is the usage NOT attributable to an admitted row of the snapshot; the rows' own usage comes through the CSR
    // so that removing a row (preemption simulation) takes exactly what adding it put there (tas_flavor.go: the cache adds
    // workload.TASUsage() of every admitted workload when it builds the flavor snapshot)
    for (int row = 0; row < s->n_adm; row++) tasRow(row, true);
  }
  ~Snap() { for (auto* x : tasS) tas::snapshot_free(x); }
  Snap(const Snap&) = delete;
  // workload.Usage().TAS of an admitted row applied to the flavor snapshots (clusterqueue_snapshot.go:121-134)
  void tasRow(int row, bool add) {
    if (!T) return;
    for (int k = T->adm_off[row]; k < T->adm_off[row + 1]; k++)
      tas::usage_apply(*tasS[T->adm_tas[k]], T->adm_leaf[k], T->adm_count[k], T->adm_req + (size_t)k * tasR, add);
  }
  // Snapshot.SimulateWorkloadUsageRemoval (snapshot.go:80-97) of one admitted row, and its revert
  void RemoveRowUsage(int row) { RemoveUsage(adm_cq[row], admUsage(row)); tasRow(row, false); }
  void AddRowUsage(int row) { AddUsage(adm_cq[row], admUsage(row)); tasRow(row, true); }

  Snap(const kq_config& c, const kq_snapshot* sn) : cfg(c), s(sn) {
    nq = s->n_cq; nc = s->n_cohort; N = nq + nc; nF = s->n_flavor; nR = s->n_resource; nfr = nF * nR;
    usage.assign(s->usage, s->usage + (size_t)N * nfr);
    removed.assign(s->n_adm, 0);
    adm_cq.assign(s->n_adm, -1);
    for (int c2 = 0; c2 < nq; c2++)
      for (int r = s->cq_adm_off[c2]; r < s->cq_adm_off[c2 + 1]; r++) adm_cq[r] = c2;
    depth.assign(N, 0);
    for (int n = 0; n < N; n++) { int d = 0; for (int a = s->parent[n]; a >= 0; a = s->parent[a]) d++; depth[n] = d; }
  }
  bool gate(uint32_t g) const { return (cfg.gates & g) != 0; }
  bool isCQ(int n) const { return n < nq; }
  bool HasParent(int n) const { return s->parent[n] >= 0; }
  int Parent(int n) const { return s->parent[n]; }
  int Root(int n) const { while (s->parent[n] >= 0) n = s->parent[n]; return n; }
  size_t ix(int n, int fr) const { return (size_t)n * nfr + fr; }
  Amount Nominal(int n, int fr) const { return Amount(s->nominal[ix(n, fr)]); }
  bool hasBL(int n, int fr) const { return s->borrow_limit[ix(n, fr)] != KQ_NIL_LIMIT; }
  bool hasLL(int n, int fr) const { return s->lend_limit[ix(n, fr)] != KQ_NIL_LIMIT; }
  Amount BL(int n, int fr) const { return Amount(s->borrow_limit[ix(n, fr)]); }
  Amount LL(int n, int fr) const { return Amount(s->lend_limit[ix(n, fr)]); }
  Amount SubtreeQuota(int n, int fr) const { return Amount(s->subtree_quota[ix(n, fr)]); }
  Amount Usage(int n, int fr) const { return Amount(usage[ix(n, fr)]); }
  void setUsage(int n, int fr, Amount a) { usage[ix(n, fr)] = a.v; }
  bool inSubtreeQuota(int n, int fr) const { return s->quota_flags[ix(n, fr)] & KQ_QF_SUBTREE; }
  uint32_t policy(int cq) const { return s->cq_policy[cq]; }

  // resource_node.go:67-72
  Amount localQuota(int n, int fr) const {
    if (hasLL(n, fr)) return MaxAmount(Amount(0), SubtreeQuota(n, fr).Sub(LL(n, fr)));
    return Amount(0);
  }
  // resource_node.go:92-95
  Amount LocalAvailable(int n, int fr) const {
    return MaxAmount(Amount(0), localQuota(n, fr).Sub(Usage(n, fr)));
  }
  // resource_node.go:106-122
  Amount available(int n, int fr) const {
    if (!HasParent(n)) return SubtreeQuota(n, fr).Sub(Usage(n, fr));
    Amount parentAvailable = available(Parent(n), fr);
    if (hasBL(n, fr)) {
      Amount lq = localQuota(n, fr);
      Amount storedInParent = SubtreeQuota(n, fr).Sub(lq);
      Amount usedInParent = MaxAmount(Amount(0), Usage(n, fr).Sub(lq));
      Amount withMaxFromParent = storedInParent.Sub(usedInParent).Add(BL(n, fr));
      parentAvailable = MinAmount(withMaxFromParent, parentAvailable);
    }
    return LocalAvailable(n, fr).Add(parentAvailable);
  }
  // resource_node.go:129-140
  Amount potentialAvailable(int n, int fr) const {
    if (!HasParent(n)) return SubtreeQuota(n, fr);
    Amount avail = localQuota(n, fr).Add(potentialAvailable(Parent(n), fr));
    if (hasBL(n, fr)) {
      Amount maxWithBorrowing = SubtreeQuota(n, fr).Add(BL(n, fr));
      avail = MinAmount(maxWithBorrowing, avail);
    }
    return avail;
  }
  // resource_node.go:144-152
  void addUsage(int n, int fr, Amount val) {
    Amount localAvailable = LocalAvailable(n, fr);
    setUsage(n, fr, Usage(n, fr).Add(val));
    if (HasParent(n) && val.Cmp(localAvailable) > 0) addUsage(Parent(n), fr, val.Sub(localAvailable));
  }
  // resource_node.go:156-165
  void removeUsage(int n, int fr, Amount val) {
    Amount usageStoredInParent = Usage(n, fr).Sub(localQuota(n, fr));
    setUsage(n, fr, Usage(n, fr).Sub(val));
    if (usageStoredInParent.CmpInt64(0) <= 0 || !HasParent(n)) return;
    removeUsage(Parent(n), fr, MinAmount(val, usageStoredInParent));
  }
  // clusterqueue_snapshot.go:107-119 (TAS part out of scope here)
  void AddUsage(int cq, const FRQ& u) { for (auto& kv : u) addUsage(cq, kv.first, kv.second); }
  void RemoveUsage(int cq, const FRQ& u) { for (auto& kv : u) removeUsage(cq, kv.first, kv.second); }
  // clusterqueue_snapshot.go:167
  Amount Available(int cq, int fr) const { return MaxAmount(Amount(0), available(cq, fr)); }
  Amount PotentialAvailable(int cq, int fr) const { return potentialAvailable(cq, fr); }
  // clusterqueue_snapshot.go:155-161 / cohort_snapshot.go:90
  bool BorrowingWith(int n, int fr, Amount val) const {
    if (isCQ(n)) return Nominal(n, fr).Cmp(Usage(n, fr).Add(val)) < 0;
    return SubtreeQuota(n, fr).Cmp(Usage(n, fr).Add(val)) < 0;
  }
  bool Borrowing(int cq, int fr) const { return BorrowingWith(cq, fr, Amount(0)); }
  // clusterqueue_snapshot.go:134-141 (quota part)
  bool Fits(int cq, const FRQ& u) const {
    for (auto& kv : u) if (Available(cq, kv.first).Cmp(kv.second) < 0) return false;
    return true;
  }
  // resource_node.go:233-243
  bool QuantitiesFitInQuota(int n, const FRQ& requests, FRQ* remaining) const {
    bool fits = true;
    remaining->clear();
    for (auto& kv : requests) {
      if (SubtreeQuota(n, kv.first).Cmp(Usage(n, kv.first).Add(kv.second)) < 0) fits = false;
      (*remaining)[kv.first] = MaxAmount(Amount(0), kv.second.Sub(LocalAvailable(n, kv.first)));
    }
    return fits;
  }
  // resource_node.go:247-254
  bool IsWithinNominalInResources(int n, const std::set<int>& frs) const {
    for (int fr : frs) if (SubtreeQuota(n, fr).Cmp(Usage(n, fr)) < 0) return false;
    return true;
  }
  // hierarchy/cohort.go:44-50 — canonical (name-sorted) children
  int nChildCohorts(int cohort) const { int k = cohort - nq; return s->child_cohort_off[k + 1] - s->child_cohort_off[k]; }
  int childCohort(int cohort, int i) const { return s->child_cohort[s->child_cohort_off[cohort - nq] + i]; }
  int nChildCQs(int cohort) const { int k = cohort - nq; return s->child_cq_off[k + 1] - s->child_cq_off[k]; }
  int childCQ(int cohort, int i) const { return s->child_cq[s->child_cq_off[cohort - nq] + i]; }
  // classical/hierarchical_preemption.go:209-215
  int getNodeHeight(int cohort) const {
    int maxHeight = std::min(nChildCohorts(cohort) + nChildCQs(cohort), 1);
    for (int i = 0; i < nChildCohorts(cohort); i++) maxHeight = std::max(maxHeight, getNodeHeight(childCohort(cohort, i)) + 1);
    return maxHeight;
  }
  // classical/hierarchical_preemption.go:221-234
  std::pair<int, bool> FindHeightOfLowestSubtreeThatFits(int c, int fr, Amount val) const {
    if (!BorrowingWith(c, fr, val) || !HasParent(c)) return {0, HasParent(c)};
    Amount remaining = val.Sub(LocalAvailable(c, fr));
    for (int t = Parent(c); t >= 0; t = Parent(t)) {
      if (!BorrowingWith(t, fr, remaining)) return {getNodeHeight(t), HasParent(t)};
      remaining = remaining.Sub(LocalAvailable(t, fr));
    }
    return {getNodeHeight(Root(c)), false};
  }
  // cohort_snapshot.go:50-58: ChildCQs first, then recurse into child cohorts
  void SubtreeClusterQueues(int cohort, std::vector<int>* out) const {
    for (int i = 0; i < nChildCQs(cohort); i++) out->push_back(childCQ(cohort, i));
    for (int i = 0; i < nChildCohorts(cohort); i++) SubtreeClusterQueues(childCohort(cohort, i), out);
  }
  // workload.go:447 Info.Usage().Quota.Assigned for an admitted row
  FRQ admUsage(int row) const {
    FRQ u;
    for (int k = s->adm_use_off[row]; k < s->adm_use_off[row + 1]; k++) {
      int fr = s->adm_use_fr[k];
      u[fr] = frq_get(u, fr).AddInt64(s->adm_use_qty[k]);
    }
    return u;
  }
  // snapshot.go:60-74
  void RemoveWorkload(int row) { removed[row] = 1; tasRow(row, false); RemoveUsage(adm_cq[row], admUsage(row)); st.victim_bytes += 16 * (depth[adm_cq[row]] + 1) * (s->adm_use_off[row + 1] - s->adm_use_off[row]); }
  void AddWorkload(int row) { removed[row] = 0; tasRow(row, true); AddUsage(adm_cq[row], admUsage(row)); st.victim_bytes += 16 * (depth[adm_cq[row]] + 1) * (s->adm_use_off[row + 1] - s->adm_use_off[row]); }
  // resourcegroups.RGByResource (util/resourcegroups/resourcegroups.go:62)
  int RGByResource(int cq, int res) const {
    for (int g = s->cq_rg_off[cq]; g < s->cq_rg_off[cq + 1]; g++)
      for (int k = s->rg_res_off[g]; k < s->rg_res_off[g + 1]; k++)
        if (s->rg_res[k] == res) return g;
    return -1;
  }
  bool rgCovers(int g, int res) const {
    for (int k = s->rg_res_off[g]; k < s->rg_res_off[g + 1]; k++) if (s->rg_res[k] == res) return true;
    return false;
  }
};

// ---- fair sharing: cache/scheduler/fair_sharing.go --------------------------------------------------
struct DRS {
  double fairWeight = 1.0;
  double unweightedRatio = 0;
  int dominantResource = -1;  // "" == -1
  bool borrowing = false;
  std::vector<int> borrowedFRs;
  bool IsZero() const { return unweightedRatio == 0; }               // :66
  bool IsBorrowing() const { return borrowing; }                      // :72
  bool isWeightZero() const { return fairWeight == 0; }               // :88
  bool zeroWeightBorrows() const { return isWeightZero() && !IsZero(); }  // :145
  double PreciseWeightedShare() const {                               // :92
    if (IsZero()) return 0.0;
    if (isWeightZero()) return std::numeric_limits<double>::infinity();
    return unweightedRatio / fairWeight;
  }
  bool IsBorrowingOn(const FRQ& requested) const {                    // :78
    for (int fr : borrowedFRs) if (frq_get(requested, fr).CmpInt64(0) > 0) return true;
    return false;
  }
};
static inline DRS NegativeDRS() { DRS d; d.unweightedRatio = -1; d.fairWeight = 1.0; return d; }  // :58
// Go cmp.Compare for float64 (NaN < everything, NaN == NaN)
static inline int cmpFloat(double a, double b) {
  bool an = std::isnan(a), bn = std::isnan(b);
  if (an && bn) return 0;
  if (an) return -1;
  if (bn) return 1;
  return a < b ? -1 : (a > b ? 1 : 0);
}
// fair_sharing.go:112-123
static inline int CompareDRS(const DRS& a, const DRS& b) {
  if (a.zeroWeightBorrows() && b.zeroWeightBorrows()) return cmpFloat(a.unweightedRatio, b.unweightedRatio);
  if (a.zeroWeightBorrows()) return 1;
  if (b.zeroWeightBorrows()) return -1;
  return cmpFloat(a.PreciseWeightedShare(), b.PreciseWeightedShare());
}
// resource names are compared alphabetically for the dominant-resource tie-break (:176); the
// boundary does not carry names, so the snapshot's resource dictionary must be name-sorted
// (index order == alphabetical order). kueue_amd/api.py guarantees it.
// fair_sharing.go:186-200
static std::vector<Amount> calculateLendable(Snap& sn, int node) {
  int root = sn.Root(node);
  std::vector<Amount> lendable(sn.nR, Amount(0));
  for (int fr = 0; fr < sn.nfr; fr++) {
    if (!sn.inSubtreeQuota(root, fr)) continue;
    int res = fr % sn.nR;
    lendable[res] = lendable[res].Add(sn.potentialAvailable(node, fr));
  }
  return lendable;
}
// fair_sharing.go:149-182 (wlReq is always nil on the path; the tests pass one)
static DRS dominantResourceShare(Snap& sn, int node, const FRQ* wlReq = nullptr) {
  DRS drs;
  drs.fairWeight = sn.s->fair_weight[node];
  if (!sn.HasParent(node)) return drs;
  std::map<int, Amount> borrowing;  // resource -> amount
  std::vector<int> borrowedFRs;
  int frcount = 0;
  for (int fr = 0; fr < sn.nfr; fr++) {
    if (!sn.inSubtreeQuota(node, fr)) continue;
    frcount++;
    Amount amountBorrowed = (wlReq ? frq_get(*wlReq, fr) : Amount(0)).Add(sn.Usage(node, fr)).Sub(sn.SubtreeQuota(node, fr));
    if (amountBorrowed.CmpInt64(0) > 0) {
      int res = fr % sn.nR;
      borrowing[res] = (borrowing.count(res) ? borrowing[res] : Amount(0)).Add(amountBorrowed);
      borrowedFRs.push_back(fr);
    }
  }
  sn.st.drs_bytes += (int64_t)frcount * 24;
  if (borrowing.empty()) return drs;
  drs.borrowing = true;
  drs.borrowedFRs = borrowedFRs;
  std::vector<Amount> lendable = calculateLendable(sn, sn.Parent(node));
  sn.st.drs_bytes += (int64_t)frcount * 40 * (sn.depth[node] + 1);
  for (auto& kv : borrowing) {  // ascending resource index == alphabetical
    Amount lr = lendable[kv.first];
    if (lr.CmpInt64(0) > 0) {
      double ratio = (double)kv.second.v * 1000.0 / (double)lr.v;
      if (ratio > drs.unweightedRatio || (ratio == drs.unweightedRatio && kv.first < drs.dominantResource)) {
        drs.unweightedRatio = ratio;
        drs.dominantResource = kv.first;
      }
    }
  }
  return drs;
}

// ---- the pending workload (workload.Info + qcache.Head) --------------------------------------------
struct PodSetReq {
  int count, min_count;
  std::vector<std::pair<int, int64_t>> req;  // (resource, qty) in given order
  // the podset of the replaced workload slice at the same index (replaceWorkloadSlice.TotalRequests[psID]): Count, Flavors, Requests
  int slice_count = 0;
  std::map<int, std::pair<int, int64_t>> slice;  // resource -> (flavor or -1, request)
  int group = -1;                            // PodSetGroupName id (kq_heads.ps_group, else kq_cycle_tas.ps_group), -1 = none
};
struct Head {
  int idx, cq;
  int64_t priority, queue_ts;
  uint32_t flags;
  std::vector<PodSetReq> ps;
  int ps_base;                     // global podset index of ps[0]
  int slice_row = -1;              // admitted row of the workload slice this head replaces (workloadslicing.ReplacedWorkloadSlice), -1 = none
  bool has_last;                   // LastAssignment != nil
  std::vector<std::vector<int>> last_tried;  // [ps][res], -1 absent
  // NominationMapping (workload.go:262): [ps][res] -> flavor, empty when unset
  std::vector<std::map<int, int>> nomination;
  bool has_nomination = false;     // len(NominationMapping) > 0: readResourceToFlavorMapping makes an entry per assigned podset, even without flavors
  bool CanBePartiallyAdmitted() const {  // workload.go:636
    for (auto& p : ps) if (p.min_count >= 0 && p.count > p.min_count) return true;
    return false;
  }
};

struct FlavorAssignment { int flavor = -1; int mode = NoFit; int tried = 0; int borrow = 0; };
// one Status.reasons string as its operands (KQ_RSN_*, include/kq_engine.h): Status.appendf flavorassigner.go:349
struct Reason { int code, flavor, resource; int64_t a, b, c; };
// FlavorAssignmentAttempt (flavor_assigner_attempts.go:35-42) as far as Assignment.NoFitReason needs it: the flavor, the worst mode over the
// resources it was tried for, and its NoFitReason as the severity rank of reasonSeverity (flavorassigner.go:306-327): 0 "", 1
// TopologyPlacementFailed, 2 WaitingForQuota, 3 ExceedsMaxQuota, 4 NoMatchingFlavor — mostSevereReason is then max(). Collected only when
// FlavorAssigner.observe is set (the UnadmittedWorkloadsObservability gate): kqo_assign_attempts, the checker of TestIsNoFitDueToCapacityAndLimits.
enum { lblNone = 0, lblTopologyPlacementFailed = 1, lblWaitingForQuota = 2, lblExceedsMaxQuota = 3, lblNoMatchingFlavor = 4 };
struct Attempt { int flavor; int mode; int label; };
struct PodSetAssignment {
  std::vector<Attempt> attempts;             // FlavorAssignmentAttempts (sorted by flavor name in the reference; an unordered set here)
  std::map<int, FlavorAssignment> flavors;  // resource -> assignment (ResourceAssignment)
  int nreasons = 0;                          // len(Status.reasons)
  std::vector<Reason> reasons;               // Status.reasons, in append order
  bool err = false;
  int count = 0;
  int group = -1;                            // PodSetGroupName id of the podset (its Flavors entries are shared with the group's other members)
  std::vector<std::pair<int, int64_t>> requests;  // effective requests (incl. injected pods)
  // TopologyAssignment (flavorassigner.go:379): (leaf, count) on TAS flavor tasIdx
  bool hasTopo = false;
  int tasIdx = -1;
  tas::Assignment topo;
  std::vector<uint8_t> topoFlags;  // KQ_EX_* of topo's domains while it is still the admission's (second pass); empty afterwards
  // flavorassigner.go:386-404
  int RepresentativeMode() const {
    if (!err && nreasons == 0) return Fit;
    if (err) return NoFit;
    if (flavors.empty()) return NoFit;
    int mode = Fit;
    for (auto& kv : flavors) if (kv.second.mode < mode) mode = kv.second.mode;
    return mode;
  }
};
struct TasDomainUse { int tas, leaf; int32_t count; int ps; };  // workload.TopologyDomainRequests; SinglePodRequests = the podset's
struct Assignment {
  std::vector<PodSetAssignment> PodSets;
  int noFitLabel = lblNone;  // Assignment.NoFitReason (flavorassigner.go:82), set by resolveNoFitReason when observe is on
  int Borrowing = 0;
  FRQ Usage;  // Usage.Quota.Assigned
  std::vector<TasDomainUse> UsageTAS;  // Usage.TAS
  int rep = -1;
  // flavorassigner.go:211-229
  int RepresentativeMode() {
    if (PodSets.empty()) return NoFit;
    if (rep >= 0) return rep;
    int mode = Fit;
    for (auto& ps : PodSets) mode = std::min(mode, ps.RepresentativeMode());
    rep = mode;
    return mode;
  }
  // flavorassigner.go:109-114
  void SetRepresentativeMode(int mode) {
    rep = mode;
    for (auto& ps : PodSets) for (auto& kv : ps.flavors) kv.second.mode = mode;
  }
};

struct Target { int row; int reason; };

// ---- Topology-Aware Scheduling inside the cycle -------------------------------------------------------
// WorkloadTASRequests (clusterqueue_snapshot.go:202): TAS flavor -> podset indices, in podset order
typedef std::map<int, std::vector<int>> TasRequests;
struct TasFailure { bool failed = false; int ps = -1, tas = -1, status = 0; int32_t a = 0, b = 0; };
struct TasResult {
  std::map<int, std::pair<int, tas::PodSetResult>> byPodSet;  // ps -> (tas flavor, result)   (TASAssignmentsResult :409)
  // TASAssignmentsResult.Failure :411 ranges over a Go map; FindTopologyAssignmentsForFlavor stops at a flavor's first failing
  // podset, so there is one candidate per TAS flavor. Canonical choice here: the lowest TAS flavor (name order).
  TasFailure Failure() const {
    TasFailure f;
    for (auto& kv : byPodSet) {
      const tas::PodSetResult& r = kv.second.second;
      if (r.status == KQ_TAS_OK || r.status == KQ_TAS_SKIPPED) continue;
      if (!f.failed || kv.second.first < f.tas) { f.failed = true; f.ps = kv.first; f.tas = kv.second.first; f.status = r.status; f.a = r.a; f.b = r.b; }
    }
    return f;
  }
};

// PodSetAssignment.HasUnhealthyNode tas_flavorassigner.go:85: the names of Status.UnhealthyNodes are the leaves the admission's
// domains flag (KQ_EX_UNHEALTHY), on any podset of the head
static bool HasUnhealthyNode(Snap& sn, const Head& wl, const PodSetAssignment& psa) {
  if (!(wl.flags & KQ_HEAD_HAS_UNHEALTHY_NODES) || !sn.T->ps_ex_off) return false;
  for (size_t i = 0; i < psa.topo.size(); i++) {
    if (i < psa.topoFlags.size()) { if (psa.topoFlags[i] & KQ_EX_UNHEALTHY) return true; continue; }
    if (psa.topo[i].first < 0) continue;
    for (size_t q = 0; q < wl.ps.size(); q++)
      for (int j = sn.T->ps_ex_off[wl.ps_base + q]; j < sn.T->ps_ex_off[wl.ps_base + q + 1]; j++)
        if ((sn.T->ps_ex_flags[j] & KQ_EX_UNHEALTHY) && sn.T->ps_ex_leaf[j] == psa.topo[i].first) return true;
  }
  return false;
}
// tas_flavorassigner.go:37-83 (+ podSetTopologyRequest :92, onlyTASFlavor :142). MultiKueue / ProvisioningRequest delays and elastic
// slices are outside the boundary (the host keeps such workloads on the Go path).
static TasRequests WorkloadsTopologyRequests(Snap& sn, const Head& wl, Assignment& a) {
  TasRequests out;
  if (!sn.T) return out;
  for (size_t p = 0; p < wl.ps.size(); p++) {
    const bool explicitReq = (sn.T->ps_flags[wl.ps_base + p] & KQ_PS_TAS_EXPLICIT) != 0;
    const bool implied = !explicitReq && sn.T->cq_tas_only[wl.cq];  // isTASImplied :225
    if (!explicitReq && !implied) continue;                        // isTASRequested :231
    if (p >= a.PodSets.size()) continue;
    PodSetAssignment& psa = a.PodSets[p];
    if (psa.err) continue;
    if (psa.count == 0) continue;
    if (psa.hasTopo && !HasUnhealthyNode(sn, wl, psa)) continue;  // :50: already computed, and no failed node to replace
    std::set<int> flavors;  // onlyTASFlavor
    for (auto& kv : psa.flavors) if (sn.tasOfFlavor[kv.second.flavor] >= 0) flavors.insert(sn.tasOfFlavor[kv.second.flavor]);
    if (flavors.size() != 1) { psa.err = true; a.rep = -1; continue; }  // ErrNoTASFlavorAssigned / MultipleTASFlavorsAssignedError -> psError :290
    out[*flavors.begin()].push_back((int)p);
  }
  return out;
}

// ClusterQueueSnapshot.FindTopologyAssignmentsForWorkload clusterqueue_snapshot.go:204-237. TASHandleOverlappingFlavors
// (aggregatedDomainUsages across flavors that share hostname leaves) is not restated: one TAS flavor per workload, else *unsupported.
static TasResult FindTopologyAssignmentsForWorkload(Snap& sn, const Head& wl, const Assignment& a, const TasRequests& reqs, bool simulateEmpty, bool* unsupported) {
  TasResult out;
  if (reqs.size() > 1 && unsupported) *unsupported = true;
  for (auto& kv : reqs) {
    const int t = kv.first;
    const std::vector<int>& pss = kv.second;
    const int n = (int)pss.size(), R = sn.tasR, nt = sn.T->n_tas;
    std::vector<int32_t> wl_off = {0, n}, count(n), level(n), slice_size(n), slice_level(n), group(n);
    std::vector<uint8_t> kind(n);
    std::vector<int64_t> req((size_t)n * R);
    std::vector<int32_t> n_layers(n, 0), layer_level((size_t)n * KQ_TAS_MAX_LEVELS, -1), layer_size((size_t)n * KQ_TAS_MAX_LEVELS, 0);
    for (int i = 0; i < n; i++) {
      const int g = wl.ps_base + pss[i];
      if (sn.T->ps_n_layers) {
        n_layers[i] = sn.T->ps_n_layers[g];
        for (int j = 0; j < KQ_TAS_MAX_LEVELS; j++) {
          layer_level[(size_t)i * KQ_TAS_MAX_LEVELS + j] = sn.T->ps_layer_level[((size_t)g * nt + t) * KQ_TAS_MAX_LEVELS + j];
          layer_size[(size_t)i * KQ_TAS_MAX_LEVELS + j] = sn.T->ps_layer_size[(size_t)g * KQ_TAS_MAX_LEVELS + j];
        }
      }
      count[i] = a.PodSets[pss[i]].count;
      level[i] = sn.T->ps_level[(size_t)g * nt + t];
      slice_size[i] = sn.T->ps_slice_size[g];
      slice_level[i] = sn.T->ps_slice_level[(size_t)g * nt + t];
      group[i] = sn.T->ps_group[g];
      kind[i] = sn.T->ps_kind[g];
      for (int r = 0; r < R; r++) req[(size_t)i * R + r] = sn.T->ps_req[(size_t)g * R + r];
    }
    kq_tas_requests rq;
    memset(&rq, 0, sizeof rq);
    rq.n_workloads = 1; rq.wl_off = wl_off.data(); rq.single_pod_requests = req.data(); rq.count = count.data(); rq.level = level.data();
    rq.kind = kind.data(); rq.slice_size = slice_size.data(); rq.slice_level = slice_level.data(); rq.group = group.data();
    if (sn.T->ps_n_layers) { rq.n_layers = n_layers.data(); rq.layer_level = layer_level.data(); rq.layer_size = layer_size.data(); }
    // node feasibility of the podsets on this flavor (kq_cycle_tas.ps_mask: taints / tolerations, nodeSelector, required affinity —
    // the simulator's side of fillInCounts, tas_flavor_snapshot.go:955-963, :1893-1905), as the leaf_ok rows of the batch boundary
    std::vector<uint8_t> leaf_ok;
    if (sn.T->ps_mask) {
      const kq_tas_topology& tp = sn.T->topo[t];
      const int nL = tp.level_off[tp.n_levels] - tp.level_off[tp.n_levels - 1];
      bool any = false;
      for (int i = 0; i < n; i++) any = any || sn.T->ps_mask[(size_t)(wl.ps_base + pss[i]) * nt + t] >= 0;
      if (any) {
        leaf_ok.assign((size_t)n * nL, 1);
        for (int i = 0; i < n; i++) {
          const int row = sn.T->ps_mask[(size_t)(wl.ps_base + pss[i]) * nt + t];
          if (row >= 0) memcpy(leaf_ok.data() + (size_t)i * nL, sn.T->leaf_mask + (size_t)row * sn.T->mask_stride, (size_t)nL);
        }
        rq.leaf_ok = leaf_ok.data();
      }
    }
    std::vector<tas::PodSetResult> res;
    if ((wl.flags & KQ_HEAD_HAS_UNHEALTHY_NODES) && sn.T->ps_ex_off) {
      // tas_flavor_snapshot.go:608-633: every podset on its own through findReplacementAssignment :686, against Status.Admission's
      // TopologyAssignment (findPSA :747, not the assignment under construction); deleteDomain :693 takes UnhealthyNodes[0]'s domain
      // out and its pods are what is placed
      std::vector<uint8_t> isr(n, 0);
      std::vector<int32_t> xo(1, 0), xl, xc;
      for (int i = 0; i < n; i++) {
        const int g = wl.ps_base + pss[i];
        const int e0 = sn.T->ps_ex_off[g], e1 = sn.T->ps_ex_off[g + 1];
        isr[i] = e1 > e0;   // psa.TopologyAssignment != nil
        int32_t affected = 0;
        for (int j = e0; j < e1; j++) {
          if (sn.T->ps_ex_flags[j] & KQ_EX_FIRST) affected = sn.T->ps_ex_count[j];
          else { xl.push_back(sn.T->ps_ex_leaf[j]); xc.push_back(sn.T->ps_ex_count[j]); }
        }
        if (isr[i]) count[i] = affected;
        xo.push_back((int32_t)xl.size());
      }
      kq_tas_replacement x;
      x.is_replacement = isr.data(); x.ex_off = xo.data(); x.ex_leaf = xl.data(); x.ex_count = xc.data();
      tas::find_workload_replacement(*sn.tasS[t], &rq, &x, 0, n, &res);
      bool any = false;
      for (int i = 0; i < n; i++) any = any || isr[i];
      if (any) sn.tasFinds++;   // (the counter is a diagnostic: one per call that placed or refused something)
      for (int i = 0; i < n; i++) if (isr[i]) out.byPodSet[pss[i]] = {t, res[i]};   // (no result for a podset without a TopologyAssignment :612)
      continue;
    }
    tas::find_workload(*sn.tasS[t], &rq, 0, n, simulateEmpty, &res);
    sn.tasFinds++;
    for (int i = 0; i < n; i++) out.byPodSet[pss[i]] = {t, res[i]};
  }
  return out;
}

// Assignment.ComputeTASNetUsage flavorassigner.go:106-155: per domain, what the assignment holds beyond admittedDomainCounts :157 of
// Status.Admission (nothing for a pending workload)
static void ComputeTASNetUsage(Snap& sn, const Head& wl, Assignment& a) {
  a.UsageTAS.clear();
  for (size_t p = 0; p < a.PodSets.size(); p++) {
    const PodSetAssignment& psa = a.PodSets[p];
    if (!psa.hasTopo) continue;
    std::map<int, int32_t> accounted;
    if (sn.T->ps_ex_off)
      for (int j = sn.T->ps_ex_off[wl.ps_base + p]; j < sn.T->ps_ex_off[wl.ps_base + p + 1]; j++)
        if (sn.T->ps_ex_leaf[j] >= 0) accounted[sn.T->ps_ex_leaf[j]] += sn.T->ps_ex_count[j];
    for (auto& dc : psa.topo) {
      if (dc.first < 0) continue;   // a domain the snapshot does not hold: it is the admission's own (stale), unchanged
      const int32_t count = dc.second - (accounted.count(dc.first) ? accounted[dc.first] : 0);
      if (count > 0) a.UsageTAS.push_back({psa.tasIdx, dc.first, count, (int)p});
    }
  }
}
// Assignment.UpdateForTASResult flavorassigner.go:87-96
static void UpdateForTASResult(Snap& sn, const Head& wl, Assignment& a, const TasResult& result) {
  for (auto& kv : result.byPodSet) {
    PodSetAssignment& psa = a.PodSets[kv.first];
    const tas::PodSetResult& r = kv.second.second;
    psa.hasTopo = r.status == KQ_TAS_OK;  // TopologyAssignment = psResult.TopologyAssignment (nil on failure)
    psa.tasIdx = kv.second.first;
    psa.topo = r.status == KQ_TAS_OK ? r.domains : tas::Assignment();
    psa.topoFlags.clear();
  }
  ComputeTASNetUsage(sn, wl, a);
}
// Assignment.updateMode flavorassigner.go:192-198. ResourceAssignment maps resources to *FlavorAssignment and resolvePodSetFlavors (:917, FilterKeys)
// hands every member of a PodSetGroupName group the SAME pointers out of groupFlavors: the mode written through one member's entry is the mode of
// the other members' entries for that resource too.
static void updateModePS(Assignment& a, int ps, int mode) {
  for (auto& kv : a.PodSets[ps].flavors) kv.second.mode = mode;
  const int grp = a.PodSets[ps].group;
  if (grp >= 0)
    for (size_t q = 0; q < a.PodSets.size(); q++)
      if ((int)q != ps && a.PodSets[q].group == grp)
        for (auto& kv : a.PodSets[q].flavors) if (a.PodSets[ps].flavors.count(kv.first)) kv.second.mode = mode;
  a.rep = mode;
}
// Usage.TAS applied to / checked against the flavor snapshots (clusterqueue_snapshot.go:121-149)
static void tasUsageApply(Snap& sn, const Head& wl, const std::vector<TasDomainUse>& u, bool add) {
  for (auto& d : u) tas::usage_apply(*sn.tasS[d.tas], d.leaf, d.count, sn.T->ps_req + (size_t)(wl.ps_base + d.ps) * sn.tasR, add);
}
static bool tasUsageFits(Snap& sn, const Head& wl, const std::vector<TasDomainUse>& u) {
  for (auto& d : u) if (!tas::fits_domain(*sn.tasS[d.tas], d.leaf, d.count, sn.T->ps_req + (size_t)(wl.ps_base + d.ps) * sn.tasR)) return false;
  return true;
}

typedef std::function<std::pair<int, int>(int cq, const Head& wl, int fr, Amount quantity)> OracleFn;

// ---- flavor fungibility helpers ----------------------------------------------------------------------
// flavorassigner.go:536-576
static bool isPreferred(GranularMode a, GranularMode b, uint32_t pol) {
  if (a.pm == pmNoFit) return false;
  if (b.pm == pmNoFit) return true;
  bool aNo = a.pm == pmNoCandidates, bNo = b.pm == pmNoCandidates;
  if (aNo != bNo) return !aNo;
  auto borrowingOverPreemption = [&]() {
    if (a.pm != b.pm) return a.pm > b.pm;
    return a.borrow < b.borrow;
  };
  auto preemptionOverBorrowing = [&]() {
    if (a.borrow != b.borrow) return a.borrow < b.borrow;
    return a.pm > b.pm;
  };
  switch (KQ_POL_PREFERENCE(pol)) {
    case KQ_PREF_BORROWING_OVER_PREEMPTION: return borrowingOverPreemption();
    case KQ_PREF_PREEMPTION_OVER_BORROWING: return preemptionOverBorrowing();
  }
  return borrowingOverPreemption();
}
// flavorassigner.go:1263-1282
static bool shouldTryNextFlavor(GranularMode m, uint32_t pol) {
  if (m.pm == pmNoFit || m.pm == pmNoCandidates) return true;
  if ((m.pm == pmPreempt || m.pm == pmReclaim) && KQ_POL_PREEMPT_TRYNEXT(pol)) return true;
  if (m.borrow != 0 && KQ_POL_BORROW_TRYNEXT(pol)) return true;
  return false;
}
static int flavorAssignmentMode(int pm) {  // flavorassigner.go:604-619
  switch (pm) { case pmNoFit: return NoFit; case pmFit: return Fit; default: return Preempt; }
}

// ---- FlavorAssigner (flavorassigner.go:644-1384) -----------------------------------------------------
struct FlavorAssigner {
  Snap& sn;
  const kq_heads* H;
  const Head& wl;
  int cq;
  bool enableFairSharing;
  OracleFn oracle;
  bool observe = false;  // features.UnadmittedWorkloadsObservability: keep the attempts and their NoFitReason

  bool flavorOk(int psGlobal, int flavor) const {
    int nw = (sn.nF + 63) / 64;
    return (H->ps_flavor_ok[(size_t)psGlobal * nw + flavor / 64] >> (flavor % 64)) & 1;
  }
  // flavorassigner.go:1386-1389
  bool canPreemptWhileBorrowing() const {
    uint32_t p = sn.policy(cq);
    return KQ_POL_BORROW_WITHIN(p) != 0 || (enableFairSharing && (KQ_POL_RECLAIM(p) != KQ_POLICY_NEVER || KQ_POL_RECLAIM_UNSET(p)));
  }
  // workload.go:226-238
  int NextFlavorToTry(int ps, int res) const {
    if (!sn.gate(KQ_GATE_FLAVOR_FUNGIBILITY)) return 0;
    if (!wl.has_last || ps >= (int)wl.last_tried.size()) return 0;
    int idx = wl.last_tried[ps][res];
    if (idx < 0) return 0;  // absent, or stored -1 -> 0 either way
    return idx + 1;
  }
  // flavorassigner.go:1334-1384 ; returns (preemptionMode, borrow, hasStatus)
  struct FitRes { int pm; int borrow; bool status; Reason why; int label = lblNone; };  // label: Status.noFitReason (:1342-1343, :1361, :1378-1380)
  FitRes fitsResourceQuota(int fr, Amount assumedUsage, int64_t requestUsage) {
    sn.st.cells++;
    sn.st.cell_bytes += 40 * (sn.depth[cq] + 1);
    Amount available = sn.Available(cq, fr);
    Amount maxCapacity = sn.PotentialAvailable(cq, fr);
    Amount val = assumedUsage.AddInt64(requestUsage);
    const int fl = fr / sn.nR, rs = fr % sn.nR;
    if (val.Cmp(maxCapacity) > 0) return {pmNoFit, 0, true, {KQ_RSN_EXCEEDS_MAX_CAPACITY, fl, rs, assumedUsage.v, requestUsage, maxCapacity.v}, lblExceedsMaxQuota};  // :1353
    auto hb = sn.FindHeightOfLowestSubtreeThatFits(cq, fr, val);
    int borrow = hb.first;
    bool mayReclaimInHierarchy = hb.second;
    if (val.Cmp(available) <= 0) return {pmFit, borrow, false, {}, lblNone};
    const Reason more = {KQ_RSN_INSUFFICIENT_UNUSED, fl, rs, val.Sub(available).v, 0, 0};  // :1372
    if (sn.Nominal(cq, fr).Cmp(val) >= 0 || mayReclaimInHierarchy || canPreemptWhileBorrowing()) {
      auto r = oracle(cq, wl, fr, val);
      int mode;
      switch (r.first) { case ppNoCandidates: mode = pmNoCandidates; break; case ppPreempt: mode = pmPreempt; break; default: mode = pmReclaim; }
      return {mode, r.second, true, more, mode != pmNoFit ? lblNone : lblWaitingForQuota};
    }
    return {pmNoFit, borrow, true, more, lblWaitingForQuota};
  }
  // flavorassigner.go:1408-1414 / :1422-1430
  bool shouldRespectNominationMapping() const {
    // len(a.wl.NominationMapping) > 0 (flavorassigner.go:1403): the mapping has one (possibly empty) entry per podset of the nominated
    // assignment (scheduler.go:651-660) — a podset that was nominated without any flavor then skips EVERY flavor of the recomputation
    return wl.has_nomination && (sn.gate(KQ_GATE_RECOMPUTE_ON_OVERLAP) || (sn.T && !(sn.T->flags & KQ_CT_NO_RECOMPUTE)));
  }
  // flavorassigner.go:1065-1210 ; psIDs == {psi} (no TAS podset groups on this path)
  // returns assignments (empty => nil), nreasons; *statusNil true when Go returns a nil status
  std::map<int, FlavorAssignment> findFlavorForPodSets(int psi, const std::vector<std::pair<int, int64_t>>& requests,
                                                       int resName, const FRQ& assignmentUsage, int* nreasons, bool* statusNil,
                                                       std::vector<Reason>* why, std::vector<Attempt>* considered = nullptr,
                                                       const std::vector<int>* psIDs = nullptr /* the podset group; null = {psi} */) {
    *nreasons = 0; *statusNil = false;
    why->clear();
    if (considered) considered->clear();
    int g = sn.RGByResource(cq, resName);
    if (g < 0) { *nreasons = 1; why->push_back({KQ_RSN_RESOURCE_UNAVAILABLE, -1, resName, 0, 0, 0}); return {}; }
    std::vector<std::pair<int, int64_t>> filtered;  // filterRequestedResources :1391
    for (auto& rq : requests) if (sn.rgCovers(g, rq.first)) filtered.push_back(rq);
    std::map<int, FlavorAssignment> bestAssignment;
    bool haveBest = false;
    GranularMode bestMode = {pmNoFit, MAXINT};
    int f0 = sn.s->rg_flavor_off[g], nflv = sn.s->rg_flavor_off[g + 1] - f0;
    uint32_t pol = sn.policy(cq);
    int attemptedFlavorIdx = -1;
    int idx = NextFlavorToTry(psi, resName);
    bool respectNom = shouldRespectNominationMapping();
    // Accounting only (the decisions below are the reference's, simulation by simulation): the engine does not run the simulations whose
    // results cannot be observed (kq_device.hpp assign_flavors "dead simulations") — inside one of its passes of 64 / |requests| flavors,
    // every simulation once a flavor with all cells Fit is the best so far or stands anywhere in the pass (no FlavorFungibility, or
    // WhenCanPreempt = TryNextFlavor; not under PreemptionOverBorrowing). Their bytes are booked as discarded so that both byte counters agree.
    const int idxFirst = idx;
    std::vector<int64_t> flavorSim(std::max(nflv, 1), 0);
    std::vector<char> flavorAllFit(std::max(nflv, 1), 0);
    auto bookDead = [&](int scannedEnd) {   // flavors [idxFirst, scannedEnd) were looked at
      if (sn.gate(KQ_GATE_FLAVOR_FUNGIBILITY) && !KQ_POL_PREEMPT_TRYNEXT(pol)) return;
      if (KQ_POL_PREFERENCE(pol) == KQ_PREF_PREEMPTION_OVER_BORROWING) return;   // (the representative mode of a flavor with a simulated cell can be "Fit" there)
      const int nf = (int)filtered.size();
      if (nf == 0) return;
      const int fpp = 64 / nf;
      bool fitSeen = false;
      for (int p0 = idxFirst; p0 < scannedEnd; p0 += fpp) {
        const int p1 = std::min(p0 + fpp, scannedEnd);
        bool passFit = false;
        for (int j = p0; j < p1; j++) passFit |= flavorAllFit[j] != 0;
        if (fitSeen || passFit) for (int j = p0; j < p1; j++) sn.st.discarded_bytes += flavorSim[j];
        fitSeen |= passFit;
      }
    };
    for (; idx < nflv; idx++) {
      attemptedFlavorIdx = idx;
      int fName = sn.s->rg_flavor[f0 + idx];
      if (respectNom) {  // shouldSkipBasedOnNominationMapping :1422
        bool keep = false;
        for (int q : (psIDs ? *psIDs : std::vector<int>{psi})) {
          auto it = wl.nomination[q].find(resName);
          keep |= it != wl.nomination[q].end() && it->second == fName;
        }
        if (!keep) { (*nreasons)++; why->push_back({KQ_RSN_NOT_IN_NOMINATION, fName, resName, 0, 0, 0}); continue; }
      }
      bool eligible = flavorOk(wl.ps_base + psi, fName);
      if (psIDs) for (int q : *psIDs) eligible &= flavorOk(wl.ps_base + q, fName);  // checkFlavorForPodSets walks every podset of the group (:1234)
      if (!eligible) {  // checkFlavorForPodSets :1212 (host-evaluated)
        (*nreasons)++; why->push_back({KQ_RSN_FLAVOR_INELIGIBLE, fName, -1, 0, 0, 0});
        if (considered) considered->push_back({fName, NoFit, lblNoMatchingFlavor});  // :1106-1108 AddNoFitFlavorAttempt
        continue;
      }
      int flavorNoFitReason = lblNone;
      std::map<int, FlavorAssignment> assignments;
      GranularMode representativeMode = {pmFit, 0};
      for (auto& rq : filtered) {
        int fr = fName * sn.nR + rq.first;
        int64_t val = rq.second;
        if (wl.slice_row >= 0) {  // :1125-1145 — the same flavor as the slice it replaces, and only the delta is requested
          auto it = wl.ps[psi].slice.find(rq.first);
          const int originalFlavor = it == wl.ps[psi].slice.end() ? -1 : it->second.first;
          if (originalFlavor != fName) {
            representativeMode = {pmNoFit, MAXINT};  // worstGranularMode(); the `break` only leaves the psIDs loop: fitsResourceQuota still runs
            (*nreasons)++; why->push_back({KQ_RSN_SLICE_FLAVOR_MISMATCH, fName, rq.first, originalFlavor, 0, 0});
            flavorNoFitReason = std::max(flavorNoFitReason, (int)lblNoMatchingFlavor);  // :1136
          } else val -= it->second.second;
        }
        const bool discarded = representativeMode.pm == pmNoFit;
        const int64_t vb0 = sn.st.victim_bytes + sn.st.drs_bytes;
        FitRes r = fitsResourceQuota(fr, frq_get(assignmentUsage, fr), val);
        if (discarded) sn.st.discarded_bytes += sn.st.victim_bytes + sn.st.drs_bytes - vb0;
        else flavorSim[idx] += sn.st.victim_bytes + sn.st.drs_bytes - vb0;
        if (r.status) { (*nreasons)++; why->push_back(r.why); flavorNoFitReason = std::max(flavorNoFitReason, r.label); }  // :1155
        GranularMode mode = {r.pm, r.borrow};
        if (isPreferred(representativeMode, mode, pol)) representativeMode = mode;
        if (representativeMode.pm == pmNoFit) continue;  // closure "return" :1161
        FlavorAssignment fa; fa.flavor = fName; fa.mode = flavorAssignmentMode(r.pm); fa.borrow = r.borrow;
        assignments[rq.first] = fa;
      }
      if (considered) considered->push_back({fName, flavorAssignmentMode(representativeMode.pm), flavorNoFitReason});  // :1174
      flavorAllFit[idx] = representativeMode.pm == pmFit;
      if (sn.gate(KQ_GATE_FLAVOR_FUNGIBILITY)) {
        if (!shouldTryNextFlavor(representativeMode, pol)) {
          bestAssignment = assignments; haveBest = true; bestMode = representativeMode;
          break;
        }
        if (isPreferred(representativeMode, bestMode, pol)) { bestAssignment = assignments; haveBest = true; bestMode = representativeMode; }
      } else if (representativeMode.pm > bestMode.pm) {
        bestAssignment = assignments; haveBest = true; bestMode = representativeMode;
        if (bestMode.pm == pmFit) { flavorAllFit[idx] = 1; bookDead(idx + 1); *statusNil = true; return bestAssignment; }
      }
    }
    bookDead(std::min(idx + 1, nflv));
    if (sn.gate(KQ_GATE_FLAVOR_FUNGIBILITY)) {
      for (auto& kv : bestAssignment) kv.second.tried = (attemptedFlavorIdx == nflv - 1) ? -1 : attemptedFlavorIdx;
      if (bestMode.pm == pmFit) { *statusNil = true; return bestAssignment; }
    }
    (void)haveBest;
    return bestAssignment;
  }

  // flavorassigner.go:708-908. The podsets of one PodSetGroupName (:782-790) are ONE flavor scan over the sum of their requests; every member
  // then takes the group's flavors for the resources it requests itself (resolvePodSetFlavors :917-945; a member without requests keeps the
  // group's TAS flavors) and the group's Status. A podset outside any group is a group of one. Groups must be contiguous (orderedgroups
  // appends PodSets in group order; this restatement indexes PodSets by podset): an interleaved workload raises tasUnsupported. Workload
  // slices: single podsets only.
  Assignment assignFlavors(const std::vector<int>* counts) {
    Assignment a;
    const int P = (int)wl.ps.size();
    std::vector<PodSetReq> requests(P);
    std::vector<PodSetAssignment> psas(P);
    for (int i = 0; i < P; i++) {
      requests[i] = wl.ps[i];
      if (counts && !counts->empty()) {
        int nc2 = (*counts)[i];
        if (wl.ps[i].count != 0 && wl.ps[i].count != nc2) {
          for (auto& rq : requests[i].req) { rq.second = rq.second / (int64_t)wl.ps[i].count; rq.second = SaturatingMul(rq.second, (int64_t)nc2); }
          requests[i].count = nc2;
        }
      }
      PodSetReq& podSet = requests[i];
      if (sn.s->pods_resource >= 0 && sn.RGByResource(cq, sn.s->pods_resource) >= 0) {
        bool found = false;
        for (auto& rq : podSet.req) if (rq.first == sn.s->pods_resource) { rq.second = podSet.count; found = true; }
        if (!found) podSet.req.push_back({sn.s->pods_resource, (int64_t)podSet.count});
      }
      std::stable_sort(podSet.req.begin(), podSet.req.end(), [&](const std::pair<int, int64_t>& x, const std::pair<int, int64_t>& y) {
        return sn.s->resource_order[x.first] < sn.s->resource_order[y.first];
      });
      psas[i].count = podSet.count;
      psas[i].group = wl.ps[i].group;
      psas[i].requests = podSet.req;
      if (sn.T && sn.T->ps_adm_flavor) {  // :765-779, as in assignFlavors
        const int g = wl.ps_base + i;
        for (int r = 0; r < sn.nR; r++) {
          const int fl = sn.T->ps_adm_flavor[(size_t)g * sn.nR + r];
          if (fl >= 0) { FlavorAssignment fa; fa.flavor = fl; fa.mode = Fit; fa.tried = 0; fa.borrow = 0; psas[i].flavors[r] = fa; }
        }
        if (sn.T->ps_ex_off && sn.T->ps_ex_off[g + 1] > sn.T->ps_ex_off[g]) {
          psas[i].hasTopo = true;
          for (auto& kv : psas[i].flavors) if (sn.tasOfFlavor[kv.second.flavor] >= 0) psas[i].tasIdx = sn.tasOfFlavor[kv.second.flavor];
          for (int j = sn.T->ps_ex_off[g]; j < sn.T->ps_ex_off[g + 1]; j++) { psas[i].topo.push_back({sn.T->ps_ex_leaf[j], sn.T->ps_ex_count[j]}); psas[i].topoFlags.push_back(sn.T->ps_ex_flags[j]); }
        }
      }
    }
    auto groupOf = [&](int i) { return wl.ps[i].group; };
    for (int i = 0; i < P;) {
      std::vector<int> psIDs{i};
      const int gid = groupOf(i);
      int j = i + 1;
      if (gid >= 0) {
        while (j < P && groupOf(j) == gid) psIDs.push_back(j++);
        for (int q = j; q < P; q++) if (groupOf(q) == gid) sn.tasUnsupported = true;  // interleaved group names
      }
      if (wl.slice_row >= 0 && psIDs.size() > 1) sn.tasUnsupported = true;
      std::vector<std::pair<int, int64_t>> sum;  // requests.Add over the members, Requests.Iter order
      for (int q : psIDs) for (auto& rq : requests[q].req) {
        bool found = false;
        for (auto& x : sum) if (x.first == rq.first) { x.second = SaturatingAdd(x.second, rq.second); found = true; }
        if (!found) sum.push_back(rq);
      }
      std::stable_sort(sum.begin(), sum.end(), [&](const std::pair<int, int64_t>& x, const std::pair<int, int64_t>& y) {
        return sn.s->resource_order[x.first] < sn.s->resource_order[y.first];
      });
      sn.st.head_io_bytes += (int64_t)sum.size() * 8;
      std::map<int, FlavorAssignment> groupFlavors;
      for (int q : psIDs) for (auto& kv : psas[q].flavors) groupFlavors[kv.first] = kv.second;  // every prior-pass assignment (:801-804)
      bool groupNil = false;
      int groupReasons = 0;
      std::vector<Reason> groupWhy;
      std::vector<Attempt> attempts;
      for (auto& rq : sum) {
        const int resName = rq.first; const int64_t quantity = rq.second;
        if (sn.RGByResource(cq, resName) < 0) {
          if (quantity == 0) continue;
          if (sn.gate(KQ_GATE_QUOTA_CHECK_STRATEGY) && sn.cfg.quota_check_strategy == KQ_QUOTA_CHECK_IGNORE_UNDECLARED) continue;
        }
        if (groupFlavors.count(resName)) continue;
        int nre; bool statusNil;
        std::vector<Reason> why;
        std::vector<Attempt> considered;
        auto flavors = findFlavorForPodSets(psIDs[0], sum, resName, a.Usage, &nre, &statusNil, &why, observe ? &considered : nullptr, &psIDs);
        if (observe) mergeFlavorAttemptsForResource(attempts, considered, resName);
        if (flavors.empty() && !sum.empty()) { groupFlavors.clear(); groupNil = true; groupReasons = nre; groupWhy = why; break; }
        for (auto& kv : flavors) groupFlavors[kv.first] = kv.second;
        if (!statusNil) { groupReasons += nre; groupWhy.insert(groupWhy.end(), why.begin(), why.end()); }
      }
      bool failed = false;
      for (int q : psIDs) {
        PodSetAssignment& psa = psas[q];
        psa.flavors.clear();
        if (!groupNil) {  // resolvePodSetFlavors :917-945
          if (!requests[q].req.empty()) {
            for (auto& kv : groupFlavors) for (auto& rq : requests[q].req) if (rq.first == kv.first) { psa.flavors[kv.first] = kv.second; break; }
          } else if (gid >= 0) {
            for (auto& kv : groupFlavors) if (sn.T && sn.tasOfFlavor[kv.second.flavor] >= 0) psa.flavors[kv.first] = kv.second;  // tasFlavorsOnly :996
          }
        }
        psa.nreasons = groupReasons; psa.reasons = groupWhy; psa.attempts = attempts;
        for (auto& kv : psa.flavors) {  // Assignment.append :1017-1041
          if (kv.second.borrow > a.Borrowing) a.Borrowing = kv.second.borrow;
          const int fr = kv.second.flavor * sn.nR + kv.first;
          int64_t requestAmount = 0;
          for (auto& rq : requests[q].req) if (rq.first == kv.first) requestAmount = rq.second;
          if (wl.slice_row >= 0) { auto it = wl.ps[q].slice.find(kv.first); if (it != wl.ps[q].slice.end()) requestAmount -= it->second.second; }
          a.Usage[fr] = frq_get(a.Usage, fr).AddInt64(requestAmount);
        }
        sn.st.head_io_bytes += (int64_t)requests[q].req.size() * 8 + (int64_t)psa.flavors.size() * 16;
        failed |= !requests[q].req.empty() && psa.flavors.empty();
        a.PodSets.push_back(psa);
        a.rep = -1;
      }
      if (failed) { resolveNoFitReason(a); return a; }
      i = j;
    }
    if (a.RepresentativeMode() == NoFit) { resolveNoFitReason(a); return a; }
    if (sn.T) assignTAS(a);
    resolveNoFitReason(a);
    return a;
  }

  // flavor_assigner_attempts.go:88-122 + mergeFlavorAttempts :124-166 (mode, NoFitReason only)
  void mergeFlavorAttemptsForResource(std::vector<Attempt>& dst, const std::vector<Attempt>& src, int resName) {
    for (const Attempt& at : src) {
      bool found = false;
      for (Attempt& e : dst) if (e.flavor == at.flavor) { e.mode = std::min(e.mode, at.mode); e.label = std::max(e.label, at.label); found = true; break; }
      if (!found) dst.push_back(at);
    }
    if (src.empty()) return;
    const int g = sn.RGByResource(cq, resName);
    for (Attempt& e : dst) {
      bool inRG = false, present = false;
      for (int k = sn.s->rg_flavor_off[g]; k < sn.s->rg_flavor_off[g + 1]; k++) inRG |= sn.s->rg_flavor[k] == e.flavor;
      for (const Attempt& at : src) present |= at.flavor == e.flavor;
      if (inRG && !present) { e.mode = NoFit; e.label = lblNoMatchingFlavor; }  // "flavor %s does not provide resource %s"
    }
  }
  // flavorassigner.go:947-994
  void resolveNoFitReason(Assignment& a) {
    if (!observe || a.RepresentativeMode() != NoFit) return;
    int overall = lblNone;
    for (PodSetAssignment& ps : a.PodSets) {
      if (ps.RepresentativeMode() != NoFit) continue;
      if (ps.attempts.empty()) { overall = std::max(overall, (int)lblNoMatchingFlavor); continue; }
      std::map<int, int> rgMinReason;  // resource group -> the least severe blocker among its (alternative) flavors
      for (const Attempt& att : ps.attempts) {
        if (att.mode != NoFit) continue;
        // findRGIndicesByFlavor: every resource group of the ClusterQueue that lists the flavor (a flavor of an attempt is always listed)
        for (int g = sn.s->cq_rg_off[cq]; g < sn.s->cq_rg_off[cq + 1]; g++)
          for (int k = sn.s->rg_flavor_off[g]; k < sn.s->rg_flavor_off[g + 1]; k++)
            if (sn.s->rg_flavor[k] == att.flavor) {
              auto it = rgMinReason.find(g);
              if (it == rgMinReason.end() || att.label < it->second) rgMinReason[g] = att.label;
            }
      }
      int podSetReason = lblNone;
      for (auto& kv : rgMinReason) podSetReason = std::max(podSetReason, kv.second);  // across groups: co-requisites
      overall = std::max(overall, podSetReason);
    }
    a.noFitLabel = overall;
  }

  // flavorassigner.go:864-903
  void assignTAS(Assignment& assignment) {
    TasRequests tasRequests = WorkloadsTopologyRequests(sn, wl, assignment);
    if (assignment.RepresentativeMode() == Fit) {
      TasResult result = FindTopologyAssignmentsForWorkload(sn, wl, assignment, tasRequests, false, &sn.tasUnsupported);
      TasFailure failure = result.Failure();
      if (failure.failed) {
        PodSetAssignment& psa = assignment.PodSets[failure.ps];
        psa.nreasons++;  // psAssignment.reason(failure.Reason)
        psa.reasons.push_back({KQ_RSN_TAS_FAILURE, sn.T->tas_flavor[failure.tas], -1, failure.status, failure.a, failure.b});
        updateModePS(assignment, failure.ps, Preempt);
      } else {
        UpdateForTASResult(sn, wl, assignment, result);
      }
    }
    if (assignment.RepresentativeMode() == Preempt && !(wl.flags & KQ_HEAD_HAS_UNHEALTHY_NODES)) {  // :879 "Don't preempt other workloads if looking for a failed node replacement"
      TasResult result = FindTopologyAssignmentsForWorkload(sn, wl, assignment, tasRequests, true, &sn.tasUnsupported);
      TasFailure failure = result.Failure();
      if (failure.failed) {
        if (observe)  // markFlavorAttempt :413-421
          for (Attempt& at : assignment.PodSets[failure.ps].attempts)
            if (at.flavor == sn.T->tas_flavor[failure.tas]) { at.mode = NoFit; at.label = lblTopologyPlacementFailed; break; }
        updateModePS(assignment, failure.ps, NoFit);
      }
      else for (auto& kv : tasRequests) for (int ps : kv.second) updateModePS(assignment, ps, Preempt);  // updateModeForTASRequests :200
    }
  }
};

// ---- preemption (scheduler/preemption/*.go) ---------------------------------------------------------
struct PreemptionCtx {
  const Head* preemptor;
  int preemptorCQ;
  FRQ workloadUsage;
  std::set<int> frsNeedPreemption;
  TasRequests tasRequests;             // preemption.go:135-138
  const Assignment* assignment = nullptr;
};

struct Preemptor {
  Snap& sn;
  bool enableFairSharing;
  std::vector<int> fsStrategies;

  Preemptor(Snap& s) : sn(s) {
    enableFairSharing = s.cfg.fair_sharing != 0;
    if (s.cfg.n_fs_strategies <= 0) fsStrategies = {KQ_FS_LESS_THAN_OR_EQUAL_TO_FINAL_SHARE, KQ_FS_LESS_THAN_INITIAL_SHARE};  // preemption.go:364-366
    else for (int i = 0; i < s.cfg.n_fs_strategies && i < 2; i++) fsStrategies.push_back(s.cfg.fs_strategies[i]);
  }

  // candidate_generator.go:54-63
  bool WorkloadUsesResources(int row, const std::set<int>& frs) const {
    for (int k = sn.s->adm_use_off[row]; k < sn.s->adm_use_off[row + 1]; k++) if (frs.count(sn.s->adm_use_fr[k])) return true;
    return false;
  }
  // common/preemption_policy.go:27-42
  static bool satisfiesPreemptionPolicy(int64_t pp, int64_t pts, int64_t cp, int64_t cts, int policy) {  // effective priorities
    bool lowerPriority = pp > cp;
    if (policy == KQ_POLICY_LOWER_PRIORITY) return lowerPriority;
    if (policy == KQ_POLICY_LOWER_OR_NEWER_EQUAL) {
      bool newerEqual = (pp == cp) && pts < cts;
      return lowerPriority || newerEqual;
    }
    return policy == KQ_POLICY_ANY;
  }
  bool SatisfiesPreemptionPolicy(const Head& preemptor, int row, int policy) const {
    return satisfiesPreemptionPolicy(preemptor.priority, preemptor.queue_ts, sn.s->adm_priority[row], sn.s->adm_queue_ts[row], policy);
  }
  // common/ordering.go:42-83 (AFS branch off: out of scope)
  int CandidatesOrdering(int a, int b, int cq) const {
    bool ea = sn.s->adm_flags[a] & KQ_ADM_EVICTED, eb = sn.s->adm_flags[b] & KQ_ADM_EVICTED;
    if (ea != eb) return ea ? -1 : 1;
    bool aIn = sn.adm_cq[a] == cq, bIn = sn.adm_cq[b] == cq;
    if (bIn != aIn) return bIn ? -1 : 1;  // CompareBool(b==cq, a==cq)
    int64_t pa = sn.s->adm_priority[a], pb = sn.s->adm_priority[b];
    if (pa != pb) return pa < pb ? -1 : 1;
    int64_t ta = sn.s->adm_reserve_ts[a], tb = sn.s->adm_reserve_ts[b];
    if (ta != tb) return tb < ta ? -1 : 1;  // quotaReservationTime(b).Compare(a)
    uint32_t ua = sn.s->adm_uid_rank[a], ub = sn.s->adm_uid_rank[b];
    return ua < ub ? -1 : (ua > ub ? 1 : 0);
  }
  void sortCandidates(std::vector<int>& v, int cq) const {
    std::sort(v.begin(), v.end(), [&](int a, int b) { return CandidatesOrdering(a, b, cq) < 0; });
  }

  // ---------- classical ----------
  enum { vNever = 0, vWithinCQ, vHierarchicalReclaim, vReclaimWithoutBorrowing, vReclaimWhileBorrowing };
  struct candidateElem { int wl; int lca; int variant; };
  static int variantReason(int v) {  // hierarchical_preemption.go:46-58
    switch (v) {
      case vWithinCQ: return KQ_REASON_IN_CLUSTER_QUEUE;
      case vHierarchicalReclaim: return KQ_REASON_IN_COHORT_RECLAMATION;
      case vReclaimWhileBorrowing: return KQ_REASON_IN_COHORT_RECLAIM_WHILE_BORROWING;
      default: return KQ_REASON_IN_COHORT_RECLAMATION;
    }
  }
  // hierarchical_preemption.go:71-77
  bool IsBorrowingWithinCohortForbidden(int cq) const { return KQ_POL_BORROW_WITHIN(sn.policy(cq)) == 0; }
  // hierarchical_preemption.go:115-123
  bool isAboveBorrowingThreshold(int64_t cand, int64_t incoming, int cq) const {
    if (cand >= incoming) return true;
    if (!KQ_POL_HAS_THRESHOLD(sn.policy(cq))) return false;
    return cand > (int64_t)sn.s->cq_borrow_prio_threshold[cq];
  }
  // hierarchical_preemption.go:81-113
  int classifyPreemptionVariant(const PreemptionCtx& ctx, int row, bool haveHierarchicalAdvantage) const {
    if (!WorkloadUsesResources(row, ctx.frsNeedPreemption)) return vNever;
    uint32_t p = sn.policy(ctx.preemptorCQ);
    bool same = sn.adm_cq[row] == ctx.preemptorCQ;
    int policy = same ? KQ_POL_WITHIN_CQ(p) : KQ_POL_RECLAIM(p);
    if (!SatisfiesPreemptionPolicy(*ctx.preemptor, row, policy)) return vNever;
    if (same) return vWithinCQ;
    if (haveHierarchicalAdvantage) return vHierarchicalReclaim;
    if (IsBorrowingWithinCohortForbidden(ctx.preemptorCQ)) return vReclaimWithoutBorrowing;
    if (isAboveBorrowingThreshold(sn.s->adm_priority[row], ctx.preemptor->priority, ctx.preemptorCQ)) return vReclaimWithoutBorrowing;
    return vReclaimWhileBorrowing;
  }
  // hierarchical_preemption.go:132-147 ; cq.Workloads iteration -> canonical row order (lists are sorted after)
  void getCandidatesFromCQ(int cq, int lca, const PreemptionCtx& ctx, bool adv, std::vector<candidateElem>* out) {
    for (int row = sn.s->cq_adm_off[cq]; row < sn.s->cq_adm_off[cq + 1]; row++) {
      if (sn.removed[row]) continue;
      sn.st.victim_bytes += 32 + 12 * (sn.s->adm_use_off[row + 1] - sn.s->adm_use_off[row]);
      int v = classifyPreemptionVariant(ctx, row, adv);
      if (v == vNever) continue;
      out->push_back({row, lca, v});
    }
  }
  // hierarchical_preemption.go:179-207
  void collectCandidatesInSubtree(const PreemptionCtx& ctx, int currentCohort, int subtreeRoot, int skipSubtree, bool adv, std::vector<candidateElem>* result) {
    for (int i = 0; i < sn.nChildCohorts(currentCohort); i++) {
      int childCohort = sn.childCohort(currentCohort, i);
      if (childCohort == skipSubtree) continue;
      if (sn.IsWithinNominalInResources(childCohort, ctx.frsNeedPreemption)) continue;
      collectCandidatesInSubtree(ctx, childCohort, subtreeRoot, skipSubtree, adv, result);
    }
    for (int i = 0; i < sn.nChildCQs(currentCohort); i++) {
      int childCq = sn.childCQ(currentCohort, i);
      if (childCq == ctx.preemptorCQ) continue;
      if (!sn.IsWithinNominalInResources(childCq, ctx.frsNeedPreemption)) getCandidatesFromCQ(childCq, subtreeRoot, ctx, adv, result);
    }
  }
  // hierarchical_preemption.go:149-175
  void collectCandidatesForHierarchicalReclaim(const PreemptionCtx& ctx, std::vector<candidateElem>* hierarchy, std::vector<candidateElem>* priorityC) {
    int cq = ctx.preemptorCQ;
    if (!sn.HasParent(cq) || KQ_POL_RECLAIM(sn.policy(cq)) == KQ_POLICY_NEVER) return;
    int previousSubtreeRoot = -1;
    FRQ remainingRequests, next;
    bool hasHierarchicalAdvantage = sn.QuantitiesFitInQuota(cq, ctx.workloadUsage, &remainingRequests);
    for (int currentSubtreeRoot = sn.Parent(cq); currentSubtreeRoot >= 0; currentSubtreeRoot = sn.Parent(currentSubtreeRoot)) {
      std::vector<candidateElem>* list = hasHierarchicalAdvantage ? hierarchy : priorityC;
      collectCandidatesInSubtree(ctx, currentSubtreeRoot, currentSubtreeRoot, previousSubtreeRoot, hasHierarchicalAdvantage, list);
      bool fits = sn.QuantitiesFitInQuota(currentSubtreeRoot, remainingRequests, &next);
      remainingRequests = next;
      hasHierarchicalAdvantage = hasHierarchicalAdvantage || fits;
      previousSubtreeRoot = currentSubtreeRoot;
    }
  }
  // candidate_generator.go:136-158
  bool candidateIsValid(const PreemptionCtx& ctx, const candidateElem& c, bool borrow) const {
    int ccq = sn.adm_cq[c.wl];
    if (ctx.preemptorCQ == ccq) return true;
    if (borrow && c.variant == vReclaimWithoutBorrowing) return false;
    if (sn.IsWithinNominalInResources(ccq, ctx.frsNeedPreemption)) return false;
    for (int node = sn.Parent(ccq); node >= 0; node = sn.Parent(node)) {
      if (node == c.lca) break;
      if (sn.IsWithinNominalInResources(node, ctx.frsNeedPreemption)) return false;
    }
    return true;
  }
  // preemption.go:669-686
  bool workloadFits(const PreemptionCtx& ctx, bool allowBorrowing) {
    sn.st.victim_bytes += 40 * (int64_t)(sn.depth[ctx.preemptorCQ] + 1) * (int64_t)ctx.workloadUsage.size();
    for (auto& kv : ctx.workloadUsage) {
      if (!allowBorrowing && sn.BorrowingWith(ctx.preemptorCQ, kv.first, kv.second)) return false;
      if (kv.second.Cmp(sn.Available(ctx.preemptorCQ, kv.first)) > 0) return false;
    }
    if (ctx.tasRequests.empty()) return true;
    return !FindTopologyAssignmentsForWorkload(sn, *ctx.preemptor, *ctx.assignment, ctx.tasRequests, false, &sn.tasUnsupported).Failure().failed;
  }
  // preemption.go:341-354
  std::vector<Target> fillBackWorkloads(const PreemptionCtx& ctx, std::vector<Target> targets, bool allowBorrowing) {
    for (int i = (int)targets.size() - 2; i >= 0; i--) {
      sn.AddWorkload(targets[i].row);
      if (workloadFits(ctx, allowBorrowing)) {
        targets[i] = targets.back();
        targets.pop_back();
      } else {
        sn.RemoveWorkload(targets[i].row);
      }
    }
    return targets;
  }
  void restoreSnapshot(const std::vector<Target>& targets) { for (auto& t : targets) sn.AddWorkload(t.row); }  // :356
  // preemption.go:700-707
  bool queueUnderNominalInResourcesNeedingPreemption(const PreemptionCtx& ctx) const {
    for (int fr : ctx.frsNeedPreemption) if (sn.Nominal(ctx.preemptorCQ, fr).Cmp(sn.Usage(ctx.preemptorCQ, fr)) <= 0) return false;
    return true;
  }
  // preemption.go:714-721
  bool queueWithinNominalInResourcesNeedingPreemption(const PreemptionCtx& ctx) const {
    for (int fr : ctx.frsNeedPreemption) if (sn.Borrowing(ctx.preemptorCQ, fr)) return false;
    return true;
  }
  // preemption.go:284-339 + candidate_generator.go:78-121
  std::vector<Target> classicalPreemptions(const PreemptionCtx& ctx) {
    std::vector<candidateElem> sameQueue, hierarchy, priorityC;
    if (KQ_POL_WITHIN_CQ(sn.policy(ctx.preemptorCQ)) != KQ_POLICY_NEVER) getCandidatesFromCQ(ctx.preemptorCQ, -1, ctx, false, &sameQueue);
    collectCandidatesForHierarchicalReclaim(ctx, &hierarchy, &priorityC);
    auto sortL = [&](std::vector<candidateElem>& v) {
      std::sort(v.begin(), v.end(), [&](const candidateElem& a, const candidateElem& b) { return CandidatesOrdering(a.wl, b.wl, ctx.preemptorCQ) < 0; });
    };
    sortL(sameQueue); sortL(priorityC); sortL(hierarchy);
    auto split = [&](const std::vector<candidateElem>& v, std::vector<candidateElem>* ev, std::vector<candidateElem>* nev) {
      size_t k = 0;
      while (k < v.size() && (sn.s->adm_flags[v[k].wl] & KQ_ADM_EVICTED)) k++;
      ev->assign(v.begin(), v.begin() + k); nev->assign(v.begin() + k, v.end());
    };
    std::vector<candidateElem> eh, nh, ep, np, es, ns, all;
    split(hierarchy, &eh, &nh); split(priorityC, &ep, &np); split(sameQueue, &es, &ns);
    for (auto* l : {&eh, &ep, &es, &nh, &np, &ns}) all.insert(all.end(), l->begin(), l->end());
    sn.st.searches++; sn.st.cand_listed += (int64_t)all.size();
    bool NoCandidateFromOtherQueues = hierarchy.empty() && priorityC.empty();
    bool NoCandidateForHierarchicalReclaim = hierarchy.empty();
    bool forbidden = IsBorrowingWithinCohortForbidden(ctx.preemptorCQ);
    std::vector<bool> attempts;
    if (NoCandidateFromOtherQueues || (forbidden && !queueUnderNominalInResourcesNeedingPreemption(ctx))) attempts = {true};
    else if (forbidden && NoCandidateForHierarchicalReclaim) attempts = {false, true};
    else attempts = {true, false};
    for (bool borrowing : attempts) {
      std::vector<Target> targets;
      for (size_t runIndex = 0; runIndex < all.size(); runIndex++) {
        const candidateElem& c = all[runIndex];
        if (!candidateIsValid(ctx, c, borrowing)) continue;
        sn.RemoveWorkload(c.wl);
        targets.push_back({c.wl, variantReason(c.variant)});
        if (workloadFits(ctx, borrowing)) {
          sn.st.cand_removed += (int64_t)targets.size();
          targets = fillBackWorkloads(ctx, targets, borrowing);
          restoreSnapshot(targets);
          return targets;
        }
      }
      restoreSnapshot(targets);
    }
    return {};
  }

  // ---------- fair sharing (preemption.go:381-631, fairsharing/*.go) ----------
  struct Ordering {  // fairsharing/ordering.go:46-90
    int preemptorCq;
    std::set<int> preemptorAncestors;
    std::map<int, std::vector<int>> clusterQueueToTarget;  // cq -> rows (in given order)
    std::set<int> prunedClusterQueues, prunedCohorts;
    bool hasWorkload(int cq) const { auto it = clusterQueueToTarget.find(cq); return it != clusterQueueToTarget.end() && !it->second.empty(); }
    int PopWorkload(int cq) { auto& v = clusterQueueToTarget[cq]; int h = v.front(); v.erase(v.begin()); return h; }
  };
  Ordering MakeClusterQueueOrdering(int cq, const std::vector<int>& candidates) {
    Ordering t; t.preemptorCq = cq;
    for (int a = sn.Parent(cq); a >= 0; a = sn.Parent(a)) t.preemptorAncestors.insert(a);
    for (int row : candidates) t.clusterQueueToTarget[sn.adm_cq[row]].push_back(row);
    return t;
  }
  // fairsharing/ordering.go:144-226 ; returns target cq or -1
  int nextTarget(Ordering& t, int cohort) {
    int highestCq = -1; DRS highestCqDrs = NegativeDRS();
    for (int i = 0; i < sn.nChildCQs(cohort); i++) {
      int cq = sn.childCQ(cohort, i);
      if (t.prunedClusterQueues.count(cq)) continue;
      DRS drs = dominantResourceShare(sn, cq);
      if ((!drs.IsBorrowing() && cq != t.preemptorCq) || !t.hasWorkload(cq)) {
        t.prunedClusterQueues.insert(cq);
      } else if (CompareDRS(drs, highestCqDrs) == 0) {
        int newCandWl = t.clusterQueueToTarget[cq][0];
        int currentCandWl = t.clusterQueueToTarget[highestCq][0];
        if (CandidatesOrdering(newCandWl, currentCandWl, t.preemptorCq) < 0) highestCq = cq;
      } else if (CompareDRS(drs, highestCqDrs) == 1) {
        highestCqDrs = drs; highestCq = cq;
      }
    }
    int highestCohort = -1; DRS highestCohortDrs = NegativeDRS();
    for (int i = 0; i < sn.nChildCohorts(cohort); i++) {
      int ch = sn.childCohort(cohort, i);
      if (t.prunedCohorts.count(ch)) continue;
      DRS drs = dominantResourceShare(sn, ch);
      if (!drs.IsBorrowing() && !t.preemptorAncestors.count(ch)) t.prunedCohorts.insert(ch);
      else if (CompareDRS(drs, highestCohortDrs) >= 0) { highestCohortDrs = drs; highestCohort = ch; }
    }
    if (highestCohort < 0 && highestCq < 0) { t.prunedCohorts.insert(cohort); return -1; }
    if (CompareDRS(highestCohortDrs, highestCqDrs) >= 0) return nextTarget(t, highestCohort);
    return highestCq;
  }
  // least_common_ancestor.go:27-58 ; returns nodes (almostLCA of preemptor, of target)
  std::pair<int, int> getAlmostLCAs(const Ordering& t, int targetCq) const {
    int lca = -1;
    for (int a = sn.Parent(targetCq); a >= 0; a = sn.Parent(a)) if (t.preemptorAncestors.count(a)) { lca = a; break; }
    auto almost = [&](int cq) { int aLca = cq; for (int a = sn.Parent(cq); a >= 0; a = sn.Parent(a)) { if (a == lca) return aLca; aLca = a; } return aLca; };
    return {almost(t.preemptorCq), almost(targetCq)};
  }
  static bool strategyPass(int strategy, const DRS& preemptorNew, const DRS& targetOld, const DRS& targetNew) {  // strategy.go:41,46
    if (strategy == KQ_FS_LESS_THAN_OR_EQUAL_TO_FINAL_SHARE) return CompareDRS(preemptorNew, targetNew) <= 0;
    return CompareDRS(preemptorNew, targetOld) < 0;
  }
  // preemption.go:690-695
  bool workloadFitsForFairSharing(const PreemptionCtx& ctx) {
    sn.RemoveUsage(ctx.preemptorCQ, ctx.workloadUsage);
    bool res = workloadFits(ctx, true);
    sn.AddUsage(ctx.preemptorCQ, ctx.workloadUsage);
    return res;
  }
  // The Iter() protocol of ordering.go:92-127 folded into a "next" call. Returns target cq or -1 when done.
  int orderingNext(Ordering& t) {
    if (!sn.HasParent(t.preemptorCq)) {
      if (!t.prunedClusterQueues.count(t.preemptorCq) && t.hasWorkload(t.preemptorCq)) return t.preemptorCq;
      return -1;
    }
    int root = sn.Root(t.preemptorCq);
    while (!t.prunedCohorts.count(root)) {
      int target = nextTarget(t, root);
      if (target < 0) continue;
      return target;
    }
    return -1;
  }
  // preemption.go:384-470
  // (probe helpers: read-only)
  int probeTopChild(int cq) const { int prev = cq; for (int a = sn.Parent(cq); a >= 0; a = sn.Parent(a)) { if (sn.Parent(a) < 0) return prev; prev = a; } return prev; }
  std::vector<int64_t> probePathCells(const PreemptionCtx& ctx) const {
    std::vector<int64_t> v;
    for (int a = ctx.preemptorCQ; a >= 0; a = sn.Parent(a)) for (auto& kv : ctx.workloadUsage) v.push_back(sn.usage[(size_t)a * sn.nfr + kv.first]);
    return v;
  }
  bool probeReachesPreemptor(const PreemptionCtx& ctx, const std::vector<int64_t>& before) const { return before != probePathCells(ctx); }
  bool runFirstFsStrategy(const PreemptionCtx& ctx, const std::vector<int>& candidates, int strategy, std::vector<Target>* targets, std::vector<int>* retryCandidates) {
    Ordering ordering = MakeClusterQueueOrdering(ctx.preemptorCQ, candidates);
    bool preemptorWithinNominal = sn.gate(KQ_GATE_FS_PREEMPT_WITHIN_NOMINAL) && queueWithinNominalInResourcesNeedingPreemption(ctx);
    const bool P = g_fs_probe_on;
    int64_t* H = g_fs_probe;
    int lastVictimCq = -1, lastVisitCq = -1, run = 0;
    auto bin = [](int64_t v) { return (int)(v > 63 ? 63 : v); };
    auto victim = [&](int cq, const std::vector<int64_t>& before) {
      H[3]++;
      if (lastVictimCq >= 0) {
        H[12]++;
        if (cq == lastVictimCq) H[9]++;
        if (sn.Parent(cq) == sn.Parent(lastVictimCq)) H[10]++;
        if (probeTopChild(cq) == probeTopChild(lastVictimCq)) { H[11]++; run++; } else { H[160 + bin(run)]++; run = 1; }
      } else run = 1;
      lastVictimCq = cq;
      H[probeReachesPreemptor(ctx, before) ? 16 : 17]++;
      int h = 0;
      if (cq != ctx.preemptorCQ) { auto al = getAlmostLCAs(ordering, cq); h = 1; for (int a = cq; a != al.second; a = sn.Parent(a)) h++; }
      H[18 + (h > 3 ? 3 : h)]++;
    };
    if (P) H[0]++;
    bool result = false;
    for (int candCQ = orderingNext(ordering); candCQ >= 0; candCQ = orderingNext(ordering)) {
      if (P) {
        H[4]++;
        if (lastVisitCq >= 0) { H[15]++; if (sn.Parent(candCQ) == sn.Parent(lastVisitCq)) H[13]++; if (probeTopChild(candCQ) == probeTopChild(lastVisitCq)) H[14]++; }
        lastVisitCq = candCQ;
        H[224 + bin((int64_t)ordering.clusterQueueToTarget[candCQ].size())]++;
      }
      std::vector<int64_t> before;
      if (candCQ == ctx.preemptorCQ) {  // InClusterQueuePreemption
        int candWl = ordering.PopWorkload(candCQ);
        if (P) before = probePathCells(ctx);
        sn.RemoveWorkload(candWl);
        targets->push_back({candWl, KQ_REASON_IN_CLUSTER_QUEUE});
        if (P) { H[1]++; H[8]++; H[5]++; victim(candCQ, before); }
        if (workloadFitsForFairSharing(ctx)) { result = true; break; }
        continue;
      }
      if (preemptorWithinNominal) {
        int candWl = ordering.PopWorkload(candCQ);
        if (P) before = probePathCells(ctx);
        sn.RemoveWorkload(candWl);
        targets->push_back({candWl, KQ_REASON_IN_COHORT_RECLAMATION});
        if (P) { H[1]++; H[8]++; H[5]++; victim(candCQ, before); }
        if (workloadFitsForFairSharing(ctx)) { result = true; break; }
        continue;
      }
      auto al = getAlmostLCAs(ordering, candCQ);
      DRS preemptorNewShare = dominantResourceShare(sn, al.first), targetOldShare = dominantResourceShare(sn, al.second);
      // fsStrategyUnsatisfiable :494-497
      if (std::isinf(preemptorNewShare.PreciseWeightedShare()) && preemptorNewShare.PreciseWeightedShare() > 0 &&
          !(std::isinf(targetOldShare.PreciseWeightedShare()) && targetOldShare.PreciseWeightedShare() > 0)) {
        while (ordering.hasWorkload(candCQ)) { retryCandidates->push_back(ordering.PopWorkload(candCQ)); if (P) { H[1]++; H[2]++; } }
        sn.st.fs_skipped_queues++;
        if (P) H[7]++;
        continue;
      }
      sn.st.fs_first_cq_evals++;
      int fails = 0; bool got = false, fit = false;
      while (ordering.hasWorkload(candCQ)) {
        int candWl = ordering.PopWorkload(candCQ);
        if (P) H[1]++;
        // ComputeTargetShareAfterRemoval target.go:67-73
        FRQ u = sn.admUsage(candWl);
        sn.RemoveUsage(candCQ, u);
        DRS targetNewShare = dominantResourceShare(sn, getAlmostLCAs(ordering, candCQ).second);
        sn.AddUsage(candCQ, u);
        if (strategyPass(strategy, preemptorNewShare, targetOldShare, targetNewShare)) {
          if (P) before = probePathCells(ctx);
          sn.RemoveWorkload(candWl);
          targets->push_back({candWl, KQ_REASON_IN_COHORT_FAIR_SHARING});
          got = true;
          if (P) { H[5]++; H[32 + bin(fails)]++; victim(candCQ, before); }
          if (workloadFitsForFairSharing(ctx)) fit = true;
          break;
        } else {
          retryCandidates->push_back(candWl);
          fails++;
          if (P) H[2]++;
        }
      }
      if (P && !got) { H[6]++; H[96 + bin(fails)]++; }
      if (fit) { result = true; break; }
    }
    if (P && run > 0) H[160 + bin(run)]++;
    return result;
  }
  // preemption.go:501-534
  bool runSecondFsStrategy(const std::vector<int>& retryCandidates, const PreemptionCtx& ctx, std::vector<Target>* targets) {
    Ordering ordering = MakeClusterQueueOrdering(ctx.preemptorCQ, retryCandidates);
    for (int candCQ = orderingNext(ordering); candCQ >= 0; candCQ = orderingNext(ordering)) {
      auto al = getAlmostLCAs(ordering, candCQ);
      DRS preemptorNewShare = dominantResourceShare(sn, al.first), targetOldShare = dominantResourceShare(sn, al.second);
      bool passed = CompareDRS(preemptorNewShare, targetOldShare) < 0;
      sn.st.fs_second_evals++;
      int candWl = ordering.PopWorkload(candCQ);
      if (passed) {
        sn.RemoveWorkload(candWl);
        targets->push_back({candWl, KQ_REASON_IN_COHORT_FAIR_SHARING});
        if (workloadFitsForFairSharing(ctx)) return true;
      }
      ordering.prunedClusterQueues.insert(candCQ);  // DropQueue
    }
    return false;
  }
  // preemption.go:633-667
  std::vector<int> findCandidates(const PreemptionCtx& ctx) {
    std::vector<int> candidates;
    int cq = ctx.preemptorCQ; uint32_t p = sn.policy(cq);
    auto forPolicy = [&](int c, int policy) {  // findCandidatesForPolicy :599-627
      for (int row = sn.s->cq_adm_off[c]; row < sn.s->cq_adm_off[c + 1]; row++) {
        if (sn.removed[row]) continue;
        sn.st.victim_bytes += 32 + 12 * (sn.s->adm_use_off[row + 1] - sn.s->adm_use_off[row]);
        if (!SatisfiesPreemptionPolicy(*ctx.preemptor, row, policy)) continue;
        if (!WorkloadUsesResources(row, ctx.frsNeedPreemption)) continue;
        candidates.push_back(row);
      }
    };
    if (KQ_POL_WITHIN_CQ(p) != KQ_POLICY_NEVER) forPolicy(cq, KQ_POL_WITHIN_CQ(p));
    if (sn.HasParent(cq) && KQ_POL_RECLAIM(p) != KQ_POLICY_NEVER) {
      std::vector<int> cqs; sn.SubtreeClusterQueues(sn.Root(cq), &cqs);
      for (int cohortCQ : cqs) {
        bool borrowing = false;  // cqIsBorrowing :657-667
        if (sn.HasParent(cohortCQ)) for (int fr : ctx.frsNeedPreemption) if (sn.Borrowing(cohortCQ, fr)) { borrowing = true; break; }
        if (cq == cohortCQ || !borrowing) continue;
        forPolicy(cohortCQ, KQ_POL_RECLAIM(p));
      }
    }
    return candidates;
  }
  // preemption.go:536-597
  std::vector<Target> fairPreemptions(const PreemptionCtx& ctx) {
    std::vector<int> candidates = findCandidates(ctx);
    if (candidates.empty()) return {};
    sortCandidates(candidates, ctx.preemptorCQ);
    sn.st.searches++; sn.st.cand_listed += (int64_t)candidates.size();
    sn.AddUsage(ctx.preemptorCQ, ctx.workloadUsage);  // SimulateUsageAddition :557
    std::vector<Target> targets; std::vector<int> retry;
    bool fits = runFirstFsStrategy(ctx, candidates, fsStrategies[0], &targets, &retry);
    if (!fits && fsStrategies.size() > 1) fits = runSecondFsStrategy(retry, ctx, &targets);
    sn.RemoveUsage(ctx.preemptorCQ, ctx.workloadUsage);  // revertSimulation
    sn.st.cand_removed += (int64_t)targets.size();
    sn.st.fair_pops += (int64_t)targets.size() + (int64_t)retry.size();
    if (!fits) { restoreSnapshot(targets); return {}; }
    targets = fillBackWorkloads(ctx, targets, true);
    restoreSnapshot(targets);
    return targets;
  }
  // preemption.go:157-162
  std::vector<Target> getTargets(const PreemptionCtx& ctx) {
    if (enableFairSharing) return fairPreemptions(ctx);
    return classicalPreemptions(ctx);
  }
  // preemption_oracle.go:43-85
  std::pair<int, int> SimulatePreemption(int cq, const Head& wl, int fr, Amount quantity) {
    PreemptionCtx ctx; ctx.preemptor = &wl; ctx.preemptorCQ = wl.cq; ctx.frsNeedPreemption = {fr}; ctx.workloadUsage[fr] = quantity;
    std::vector<Target> candidates = getTargets(ctx);
    if (candidates.empty()) return {ppNoCandidates, sn.FindHeightOfLowestSubtreeThatFits(cq, fr, quantity).first};
    // SimulateWorkloadUsageRemoval snapshot.go:80-100
    for (auto& c : candidates) sn.RemoveRowUsage(c.row);
    int borrowAfter = sn.FindHeightOfLowestSubtreeThatFits(cq, fr, quantity).first;
    for (auto& c : candidates) sn.AddRowUsage(c.row);
    for (auto& c : candidates) if (sn.adm_cq[c.row] == cq) return {ppPreempt, borrowAfter};
    return {ppReclaim, borrowAfter};
  }
  // Assignment.TotalRequestsFor flavorassigner.go:267-296 (no workload slices)
  FRQ TotalRequestsFor(const Head& wl, Assignment& a) const {
    FRQ usage;
    for (size_t i = 0; i < wl.ps.size(); i++) {
      // (an assignment that stopped early — atLeastOnePodsAssignmentFailed :848 — holds fewer PodSets than the workload; with a Preempt
      // RepresentativeMode the reference indexes past them here and panics. Only a stale bookmark past the last flavor gets there: a podset
      // with requests, an empty scan and an empty Status. The restatement stops at the PodSets there are; the engine reads "no flavor".)
      if (i >= a.PodSets.size()) break;
      const PodSetReq& ps = wl.ps[i];
      int newCount = a.PodSets[i].count;
      if (wl.slice_row >= 0) newCount = ps.count - ps.slice_count;  // :265-267
      for (auto rq : ps.req) {
        int64_t q = rq.second;
        if (ps.count != 0 && ps.count != newCount) { q = q / (int64_t)ps.count; q = SaturatingMul(q, (int64_t)newCount); }
        if (q == 0) continue;
        auto it = a.PodSets[i].flavors.find(rq.first);
        if (it == a.PodSets[i].flavors.end()) continue;  // IgnoreUndeclared path; BlockUndeclared never gets here with a Preempt assignment
        int fr = it->second.flavor * sn.nR + rq.first;
        usage[fr] = frq_get(usage, fr).AddInt64(q);
      }
    }
    return usage;
  }
  // preemption.go:132-155 + flavorResourcesNeedPreemption :586-597
  std::vector<Target> GetTargets(const Head& wl, Assignment& a) {
    PreemptionCtx ctx; ctx.preemptor = &wl; ctx.preemptorCQ = wl.cq;
    for (auto& ps : a.PodSets) for (auto& kv : ps.flavors) if (kv.second.mode == Preempt) ctx.frsNeedPreemption.insert(kv.second.flavor * sn.nR + kv.first);
    ctx.workloadUsage = TotalRequestsFor(wl, a);
    if (sn.T) { ctx.tasRequests = WorkloadsTopologyRequests(sn, wl, a); ctx.assignment = &a; }
    return getTargets(ctx);
  }
};

// ---- scheduler (scheduler/scheduler.go) -----------------------------------------------------------
struct Entry {
  Head head;
  Assignment assignment;
  std::vector<Target> preemptionTargets;
  int status = KQ_ST_NOT_NOMINATED;
  int requeueReason = KQ_RQ_GENERIC;
  int skip = KQ_SKIP_NONE;
  int action = KQ_ACT_NONE;
  int nominatedMode = NoFit;
  int finalMode = NoFit;
  int order = -1;
  // what processEntry did to the state the entries share (read by kqo_jacobi_probe only): the usage it added to the ClusterQueue
  // and whether its targets joined preemptedWorkloads; recomputed = 1 overlap recomputation, 2 TAS recomputation
  FRQ effUsage;
  bool effInsert = false;
  int recomputed = 0;
};

// PodSetReducer (podset_reducer.go:28-86): Search finds the largest counts between PodSets[*].Count and *MinimumCount that
// pass fits(); sort.Search(totalDelta + 1, f) = smallest i in [0, n) with f(i) true, n if none. The last call of fits might
// not be a successful one, hence lastGoodIdx. fits(counts) records its own result when it returns true.
template <class F> static bool podSetReducerSearch(const std::vector<int>& fullCounts, const std::vector<int>& minCounts, F fits) {
  const int P = (int)fullCounts.size();
  std::vector<int> deltas(P);
  int totalDelta = 0;
  for (int i = 0; i < P; i++) { deltas[i] = fullCounts[i] - (minCounts[i] >= 0 ? minCounts[i] : fullCounts[i]); totalDelta += deltas[i]; }  // ptr.Deref(MinCount, Count)
  if (totalDelta == 0) return false;
  int lastGoodIdx = 0;
  std::vector<int> current(P);
  auto f = [&](int si) {
    for (int i = 0; i < P; i++) current[i] = fullCounts[i] - (int)((int64_t)deltas[i] * (int64_t)si / (int64_t)totalDelta);  // fillPodSetSizesForSearchIndex
    const bool ok = fits(current);
    if (ok) lastGoodIdx = si;
    return ok;
  };
  int lo = 0, hi = totalDelta + 1;
  while (lo < hi) { const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (!f(mid)) lo = mid + 1; else hi = mid; }
  return lo == lastGoodIdx;
}

struct Scheduler {
  Snap& sn;
  const kq_heads* H;
  Preemptor preemptor;
  int64_t schedulingCycle;
  OracleFn stubOracle;  // tests only (flavorassigner_test.go:159-176)

  Scheduler(Snap& s, const kq_heads* h) : sn(s), H(h), preemptor(s), schedulingCycle(h->cycle) {}

  Head loadHead(int i) const {
    Head h; h.idx = i; h.cq = H->cq[i]; h.priority = H->priority[i]; h.queue_ts = H->queue_ts[i]; h.flags = H->flags[i];
    h.ps_base = H->ps_off[i];
    h.has_last = (h.flags & KQ_HEAD_HAS_LAST_ASSIGNMENT) != 0;
    for (int p = H->ps_off[i]; p < H->ps_off[i + 1]; p++) {
      PodSetReq ps; ps.count = H->ps_count[p]; ps.min_count = H->ps_min_count ? H->ps_min_count[p] : -1;
      for (int k = H->ps_req_off[p]; k < H->ps_req_off[p + 1]; k++) ps.req.push_back({H->req_res[k], H->req_qty[k]});
      if (H->slice_row && H->slice_row[i] >= 0 && sn.gate(KQ_GATE_ELASTIC_JOBS)) {
        ps.slice_count = H->ps_slice_count ? H->ps_slice_count[p] : 0;
        for (int k = H->ps_req_off[p]; k < H->ps_req_off[p + 1]; k++)
          ps.slice[H->req_res[k]] = {H->req_slice_flavor ? H->req_slice_flavor[k] : -1, H->req_slice_qty ? H->req_slice_qty[k] : 0};
        if (sn.s->pods_resource >= 0 && !ps.slice.count(sn.s->pods_resource))
          ps.slice[sn.s->pods_resource] = {H->ps_slice_pods_flavor ? H->ps_slice_pods_flavor[p] : -1, H->ps_slice_pods_qty ? H->ps_slice_pods_qty[p] : 0};
      }
      ps.group = H->ps_group ? H->ps_group[p] : (sn.T && sn.T->ps_group ? sn.T->ps_group[p] : -1);
      h.ps.push_back(ps);
      std::vector<int> lt(sn.nR, -1);
      if (H->ps_last_tried) for (int r = 0; r < sn.nR; r++) lt[r] = H->ps_last_tried[(size_t)p * sn.nR + r];
      h.last_tried.push_back(lt);
    }
    h.nomination.assign(h.ps.size(), {});
    if (H->slice_row && sn.gate(KQ_GATE_ELASTIC_JOBS)) h.slice_row = H->slice_row[i];
    return h;
  }
  // scheduler.go:840-856
  bool lastAssignmentOutdated(const Head& h) const {
    int i = h.idx;
    if (sn.gate(KQ_GATE_PRESERVE_SCAN_PROGRESS)) {
      uint64_t lh = H->last_hash ? H->last_hash[i] : 0, ch = H->hash ? H->hash[i] : 0;
      if (!(lh == 0 || ch == 0 || lh == ch)) return true;  // MatchesSchedulingShape workload.go:204
      if (schedulingCycle - (H->last_cycle ? H->last_cycle[i] : 0) <= 1) return false;
    }
    return sn.s->cq_generation[h.cq] > (H->last_generation ? H->last_generation[i] : 0);
  }
  OracleFn makeOracle() {
    if (stubOracle) return stubOracle;
    return [this](int cq, const Head& wl, int fr, Amount q) { return preemptor.SimulatePreemption(cq, wl, fr, q); };
  }
  // scheduler.go:880-924
  void getInitialAssignments(Head& wl0, Assignment* outA, std::vector<Target>* outT) {
    // ReplacedWorkloadSlice looks the old slice up in queue.Workloads (workloadslicing.go:371): inside an overlap recomputation
    // (SimulateWorkloadRemoval of the other preemptions' victims, scheduler.go:726-727) a slice that was preempted is not there
    // any more and the head is assigned like any other workload
    Head wl = wl0;
    if (wl.slice_row >= 0 && sn.removed[wl.slice_row]) wl.slice_row = -1;
    FlavorAssigner fa{sn, H, wl, wl.cq, sn.cfg.fair_sharing != 0, makeOracle()};
    Assignment full = fa.assignFlavors(nullptr);
    int arm = full.RepresentativeMode();
    // workloadslicing.ReplacedWorkloadSlice :883: the replaced slice is a target from the start (Target without a reason)
    std::vector<Target> sliceTargets;
    if (wl.slice_row >= 0) sliceTargets.push_back({wl.slice_row, KQ_REASON_REPLACED_SLICE});
    // (should the victim search name the slice itself, the reference's list holds it twice and FindReplacedSliceTarget :402 takes the
    // first entry out; here the row is listed once, under the search's reason: it is evicted either way — documented in kq_engine.h)
    auto withSlice = [&](const std::vector<Target>& t) {
      std::vector<Target> r;
      bool dup = false;
      for (auto& x : t) if (!sliceTargets.empty() && x.row == sliceTargets[0].row) dup = true;
      if (!dup) r = sliceTargets;
      r.insert(r.end(), t.begin(), t.end());
      return r;
    };
    if (arm == Fit) { *outA = full; *outT = sliceTargets; return; }
    if (arm == Preempt) {
      std::vector<Target> t = preemptor.GetTargets(wl, full);
      if (!t.empty()) { *outA = full; *outT = withSlice(t); return; }
    }
    if (sn.gate(KQ_GATE_PARTIAL_ADMISSION) && wl.CanBePartiallyAdmitted()) {
      int P = (int)wl.ps.size();
      std::vector<int> fullCounts(P), minCounts(P);
      for (int i = 0; i < P; i++) { fullCounts[i] = wl.ps[i].count; minCounts[i] = wl.ps[i].min_count; }
      Assignment lastA; std::vector<Target> lastT;
      bool found = podSetReducerSearch(fullCounts, minCounts, [&](const std::vector<int>& current) {
        Assignment a = fa.assignFlavors(&current);
        int mode = a.RepresentativeMode();
        if (mode == Fit) { lastA = a; lastT.clear(); return true; }
        if (mode == Preempt) {
          std::vector<Target> t = preemptor.GetTargets(wl, a);
          if (!t.empty()) { lastA = a; lastT = t; return true; }
        }
        return false;
      });
      if (found) { *outA = lastA; *outT = withSlice(lastT); return; }
    }
    *outA = full; outT->clear();
  }
  // scheduler.go:821-838 + recordAssignment :281-286
  void getAssignments(Entry& e) {
    if (e.head.has_last && lastAssignmentOutdated(e.head)) e.head.has_last = false;
    getInitialAssignments(e.head, &e.assignment, &e.preemptionTargets);
    if (sn.T) updateAssignmentForTAS(e.head, e.assignment, e.preemptionTargets);
    // recordAssignment: LastAssignment = &assignment.LastState
    e.head.has_last = true;
    for (size_t p = 0; p < e.head.ps.size(); p++) {
      std::vector<int> lt(sn.nR, -1);
      if (p < e.assignment.PodSets.size()) for (auto& kv : e.assignment.PodSets[p].flavors) lt[kv.first] = kv.second.tried;
      e.head.last_tried[p] = lt;
    }
    if (e.assignment.PodSets.size() < e.head.ps.size()) e.head.last_tried.resize(e.assignment.PodSets.size());
  }
  // scheduler.go:941-985
  void updateAssignmentForTAS(const Head& wl, Assignment& assignment, const std::vector<Target>& targets) {
    if (assignment.RepresentativeMode() != Preempt) return;
    if (wl.flags & KQ_HEAD_UNHEALTHY_ASSIGNMENT) return;  // :952 !HasTopologyAssignmentWithUnhealthyNode
    bool anyExplicit = false;
    for (size_t p = 0; p < wl.ps.size(); p++) if (sn.T->ps_flags[wl.ps_base + p] & KQ_PS_TAS_EXPLICIT) anyExplicit = true;
    if (!anyExplicit && !sn.T->cq_tas_only[wl.cq]) return;
    TasRequests tasRequests = WorkloadsTopologyRequests(sn, wl, assignment);
    TasResult tasResult;
    if (!targets.empty()) {
      for (auto& t : targets) sn.RemoveRowUsage(t.row);  // SimulateWorkloadUsageRemoval
      tasResult = FindTopologyAssignmentsForWorkload(sn, wl, assignment, tasRequests, false, &sn.tasUnsupported);
      for (auto& t : targets) sn.AddRowUsage(t.row);
    } else {
      tasResult = FindTopologyAssignmentsForWorkload(sn, wl, assignment, tasRequests, true, &sn.tasUnsupported);
    }
    UpdateForTASResult(sn, wl, assignment, tasResult);
  }
  // scheduler.go:647 assignmentUsage -> netUsage :785-794
  FRQ assignmentUsage(const Entry& e) const {
    if (e.head.flags & KQ_HEAD_HAS_QUOTA_RESERVATION) return {};
    return e.assignment.Usage;
  }
  // scheduler.go:771-777 ; canonical removal order = ascending admitted row
  enum { FitsCheckOk = 0, FitsCheckNoQuota = 1, FitsCheckNoTAS = 2 };  // clusterqueue_snapshot.go:42-51
  int fitsCheck(const Entry& e, int cq, const FRQ& usage, const std::set<int>& preempted, const std::vector<Target>& newTargets) {
    std::set<int> merged = preempted;
    for (auto& t : newTargets) merged.insert(t.row);
    for (int row : merged) sn.RemoveRowUsage(row);
    sn.st.entry_bytes += (int64_t)usage.size() * 40 * (sn.depth[cq] + 1);
    int res = sn.Fits(cq, usage) ? FitsCheckOk : FitsCheckNoQuota;  // ClusterQueueSnapshot.Fits :136-150
    if (res == FitsCheckOk && sn.T && !tasUsageFits(sn, e.head, e.assignment.UsageTAS)) res = FitsCheckNoTAS;
    for (int row : merged) sn.AddRowUsage(row);
    return res;
  }
  static bool hasAny(const std::set<int>& p, const std::vector<Target>& t) { for (auto& x : t) if (p.count(x.row)) return true; return false; }
  // scheduler.go:707-769
  bool updateAssignmentIfNeeded(Entry& e, int cq, const std::set<int>& preempted, FRQ* usageOut) {
    FRQ usage = assignmentUsage(e);
    int fc = fitsCheck(e, cq, usage, preempted, e.preemptionTargets);
    const bool needsTASRecompute = fc == FitsCheckNoTAS && !(sn.T->flags & KQ_CT_NO_RECOMPUTE);  // TASRecomputeAssignmentWithinSchedulingCycle
    bool needsOverlapRecompute = hasAny(preempted, e.preemptionTargets) && sn.gate(KQ_GATE_RECOMPUTE_ON_OVERLAP);
    if (!needsOverlapRecompute && !needsTASRecompute) { *usageOut = usage; return fc == FitsCheckOk; }
    if (!needsOverlapRecompute) {  // case needsTASRecompute :728
      sn.tasRecomputes++;
      e.recomputed = 2;
      e.head.has_last = false;
      e.head.nomination.assign(e.head.ps.size(), {});
      e.head.has_nomination = !e.assignment.PodSets.empty();
      for (size_t p = 0; p < e.assignment.PodSets.size(); p++) for (auto& kv : e.assignment.PodSets[p].flavors) e.head.nomination[p][kv.first] = kv.second.flavor;
      getAssignments(e);
      usage = assignmentUsage(e);
      fc = fitsCheck(e, cq, usage, preempted, e.preemptionTargets);
      e.head.nomination.assign(e.head.ps.size(), {}); e.head.has_nomination = false;
      *usageOut = usage;
      return fc == FitsCheckOk;
    }
    e.recomputed = 1;
    std::vector<int> victims(preempted.begin(), preempted.end());
    // The engine keeps a second usage plane without the cycle's victims instead of removing and
    // re-adding them here, so this traffic is excluded from the comparable byte count.
    int64_t vb0 = sn.st.victim_bytes;
    for (int row : victims) sn.RemoveWorkload(row);  // SimulateWorkloadRemoval snapshot.go:105-114
    sn.st.discarded_bytes += sn.st.victim_bytes - vb0;
    e.head.has_last = false;
    // readResourceToFlavorMapping :651-660
    e.head.nomination.assign(e.head.ps.size(), {});
    e.head.has_nomination = !e.assignment.PodSets.empty();
    for (size_t p = 0; p < e.assignment.PodSets.size(); p++) for (auto& kv : e.assignment.PodSets[p].flavors) e.head.nomination[p][kv.first] = kv.second.flavor;
    getAssignments(e);
    vb0 = sn.st.victim_bytes;
    for (int row : victims) sn.AddWorkload(row);
    sn.st.discarded_bytes += sn.st.victim_bytes - vb0;
    if (e.assignment.RepresentativeMode() == Fit) e.assignment.SetRepresentativeMode(DeferredFit);
    usage = assignmentUsage(e);
    fc = fitsCheck(e, cq, usage, preempted, e.preemptionTargets);
    e.head.nomination.assign(e.head.ps.size(), {}); e.head.has_nomination = false;
    *usageOut = usage;
    return fc == FitsCheckOk;
  }
  // scheduler.go:796-814
  FRQ quotaResourcesToReserve(Entry& e, int cq) {
    if (e.assignment.RepresentativeMode() != Preempt) return e.assignment.Usage;
    FRQ reserved;
    for (auto& kv : e.assignment.Usage) {
      int fr = kv.first; Amount usage = kv.second;
      if (e.assignment.Borrowing > 0) {
        if (!sn.hasBL(cq, fr)) reserved[fr] = usage;
        else reserved[fr] = MinAmount(usage, sn.Nominal(cq, fr).Add(sn.BL(cq, fr)).Sub(sn.Usage(cq, fr)));
      } else {
        reserved[fr] = MaxAmount(Amount(0), MinAmount(usage, sn.Nominal(cq, fr).Sub(sn.Usage(cq, fr))));
      }
    }
    return reserved;
  }
  // scheduler.go:392-523
  void processEntry(Entry& e, std::set<int>& preemptedWorkloads) {
    int cq = e.head.cq;
    FRQ usage;
    bool fitsOk = updateAssignmentIfNeeded(e, cq, preemptedWorkloads, &usage);
    int mode = e.assignment.RepresentativeMode();
    e.finalMode = mode;
    if (sn.T && !(sn.T->flags & KQ_CT_NO_FAIL_FAST) && (e.head.flags & KQ_HEAD_UNHEALTHY_ASSIGNMENT) && mode != Fit) {  // :425-428 handleFailedTASReplacement :522
      e.status = KQ_ST_EVICTED; e.action = KQ_ACT_EVICT;
      return;
    }
    if (mode == NoFit) { e.requeueReason = KQ_RQ_NOFIT; return; }
    if (mode == Preempt) {
      if (e.preemptionTargets.empty()) {
        e.requeueReason = KQ_RQ_PREEMPTION_NO_CANDIDATES;
        // reserveCapacityForUnreclaimablePreempt :538-543 ; CanAlwaysReclaim policy.go:27
        bool canAlwaysReclaim = KQ_POL_RECLAIM(sn.policy(cq)) == KQ_POLICY_ANY;
        if (!canAlwaysReclaim || (sn.gate(KQ_GATE_PRIORITIZE_PREEMPTORS) && (e.head.flags & KQ_HEAD_IS_PREEMPTOR))) {
          FRQ r = (e.head.flags & KQ_HEAD_HAS_QUOTA_RESERVATION) ? FRQ() : quotaResourcesToReserve(e, cq);  // resourcesToReserve :780 via netUsage
          sn.AddUsage(cq, r);
          e.effUsage = r;
          if (sn.T) tasUsageApply(sn, e.head, e.assignment.UsageTAS, true);  // resourcesToReserve -> netUsage :785-794 carries Usage.TAS
          sn.st.entry_bytes += (int64_t)r.size() * 8 * (sn.depth[cq] + 1);
        }
        return;
      }
    }
    if (mode == DeferredFit) {
      e.requeueReason = KQ_RQ_PENDING_PREEMPTION;
      e.head.has_last = false;
      sn.AddUsage(cq, usage);
      e.effUsage = usage;
      if (sn.T) tasUsageApply(sn, e.head, e.assignment.UsageTAS, true);
      sn.st.entry_bytes += (int64_t)usage.size() * 8 * (sn.depth[cq] + 1);
      return;
    }
    if (hasAny(preemptedWorkloads, e.preemptionTargets)) { e.status = KQ_ST_SKIPPED; e.skip = KQ_SKIP_OVERLAP; return; }
    if (!fitsOk) { e.status = KQ_ST_SKIPPED; e.skip = KQ_SKIP_NO_LONGER_FITS; return; }
    for (auto& t : e.preemptionTargets) preemptedWorkloads.insert(t.row);
    e.effInsert = true;
    sn.AddUsage(cq, usage);
    e.effUsage = usage;
    if (sn.T) tasUsageApply(sn, e.head, e.assignment.UsageTAS, true);
    sn.st.entry_bytes += (int64_t)usage.size() * 8 * (sn.depth[cq] + 1);
    if (mode == Preempt) {
      // issuePreemptions :563 -> markPreemptionOutcome :291 (all evictions assumed to succeed)
      e.action = KQ_ACT_PREEMPT;
      e.requeueReason = KQ_RQ_PENDING_PREEMPTION;
      return;
    }
    e.status = KQ_ST_ASSUMED;  // markNominated + admit -> assumeWorkload -> markAssumed
    e.action = KQ_ACT_ADMIT;
  }

  // scheduler.go:1110-1163 ; unstable sort w/ ties -> canonical: stable sort
  std::vector<int> classicalOrder(std::vector<Entry>& entries) {
    std::vector<int> ord(entries.size());
    for (size_t i = 0; i < ord.size(); i++) ord[i] = (int)i;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
      Entry &a = entries[x], &b = entries[y];
      bool aq = a.head.flags & KQ_HEAD_HAS_QUOTA_RESERVATION, bq = b.head.flags & KQ_HEAD_HAS_QUOTA_RESERVATION;
      if (aq != bq) return aq;
      if (sn.gate(KQ_GATE_PRIORITIZE_PREEMPTORS)) {
        bool ap = a.head.flags & KQ_HEAD_IS_PREEMPTOR, bp = b.head.flags & KQ_HEAD_IS_PREEMPTOR;
        if (ap != bp) return ap;
      }
      if (a.assignment.Borrowing != b.assignment.Borrowing) return a.assignment.Borrowing < b.assignment.Borrowing;
      if (sn.gate(KQ_GATE_PRIORITY_SORTING_IN_COHORT) && a.head.priority != b.head.priority) return a.head.priority > b.head.priority;
      return a.head.queue_ts < b.head.queue_ts;
    });
    return ord;
  }

  // ---- fair sharing iterator: fair_sharing_iterator.go ----
  struct DrsKey { int cohort; int entry; bool operator<(const DrsKey& o) const { return cohort != o.cohort ? cohort < o.cohort : entry < o.entry; } };
  std::map<DrsKey, DRS> drsValues;
  std::map<int, FRQ> requestedFRs;
  // fair_sharing_iterator.go:227-263
  void computeDRS(int rootCohort, std::vector<Entry>& entries, const std::map<int, int>& cqToEntry) {
    drsValues.clear(); requestedFRs.clear();
    std::vector<int> cqs; sn.SubtreeClusterQueues(rootCohort, &cqs);
    for (int cq : cqs) {
      auto it = cqToEntry.find(cq);
      if (it == cqToEntry.end()) continue;
      Entry& e = entries[it->second];
      FRQ usage = assignmentUsage(e);
      sn.AddUsage(cq, usage);
      if (sn.gate(KQ_GATE_FS_PRIORITIZE_NON_BORROWING)) requestedFRs[it->second] = usage;
      DRS d = dominantResourceShare(sn, cq);
      for (int anc = sn.Parent(cq); anc >= 0; anc = sn.Parent(anc)) {
        drsValues[{anc, it->second}] = d;
        d = dominantResourceShare(sn, anc);
      }
      sn.RemoveUsage(cq, usage);
    }
  }
  // fair_sharing_iterator.go:176-221
  bool less(std::vector<Entry>& entries, int a, int b, int parentCohort) {
    Entry &ea = entries[a], &eb = entries[b];
    if (sn.gate(KQ_GATE_PRIORITIZE_PREEMPTORS)) {
      bool ap = ea.head.flags & KQ_HEAD_IS_PREEMPTOR, bp = eb.head.flags & KQ_HEAD_IS_PREEMPTOR;
      if (ap != bp) return ap;
    }
    DRS aDrs = drsValues.count({parentCohort, a}) ? drsValues[{parentCohort, a}] : DRS{0, 0, -1, false, {}};
    DRS bDrs = drsValues.count({parentCohort, b}) ? drsValues[{parentCohort, b}] : DRS{0, 0, -1, false, {}};
    if (sn.gate(KQ_GATE_FS_PRIORITIZE_NON_BORROWING)) {
      bool aB = aDrs.IsBorrowingOn(requestedFRs[a]), bB = bDrs.IsBorrowingOn(requestedFRs[b]);
      if (aB != bB) return !aB;
    }
    int c = CompareDRS(aDrs, bDrs);
    if (c != 0) return c == -1;
    if (sn.gate(KQ_GATE_PRIORITY_SORTING_IN_COHORT) && ea.head.priority != eb.head.priority) return ea.head.priority > eb.head.priority;
    return ea.head.queue_ts < eb.head.queue_ts;
  }
  // fair_sharing_iterator.go:125-163 ; returns entry index or -1
  int runTournament(int cohort, std::vector<Entry>& entries, const std::map<int, int>& cqToEntry) {
    std::vector<int> candidates;
    for (int i = 0; i < sn.nChildCohorts(cohort); i++) { int c = runTournament(sn.childCohort(cohort, i), entries, cqToEntry); if (c >= 0) candidates.push_back(c); }
    for (int i = 0; i < sn.nChildCQs(cohort); i++) { auto it = cqToEntry.find(sn.childCQ(cohort, i)); if (it != cqToEntry.end()) candidates.push_back(it->second); }
    if (candidates.empty()) return -1;
    int best = candidates[0];
    for (size_t k = 1; k < candidates.size(); k++) if (less(entries, candidates[k], best, cohort)) best = candidates[k];
    return best;
  }
  // fair_sharing_iterator.go:47-119 ; getCq: canonical = lowest CQ index remaining.
  // pop() reads the snapshot as mutated by the entries processed so far (scheduler.go:358-359 calls
  // processEntry between pops), so iteration and processing are interleaved here exactly like there.
  void fairIterate(std::vector<Entry>& entries, std::set<int>& preemptedWorkloads) {
    std::map<int, int> cqToEntry;
    for (size_t i = 0; i < entries.size(); i++) cqToEntry[entries[i].head.cq] = (int)i;  // later entry of same CQ overwrites (:58-60)
    int pos = 0;
    while (!cqToEntry.empty()) {
      int cq = cqToEntry.begin()->first;
      int w;
      if (!sn.HasParent(cq)) { w = cqToEntry[cq]; }
      else {
        int root = sn.Root(cq);
        computeDRS(root, entries, cqToEntry);
        w = runTournament(root, entries, cqToEntry);
      }
      cqToEntry.erase(entries[w].head.cq);
      entries[w].order = pos++;
      processEntry(entries[w], preemptedWorkloads);
    }
  }

  // scheduler.go:308-386 steps 3-5
  void schedule(std::vector<Entry>& entries, bool nominate_only = false) {
    entries.clear();
    for (int i = 0; i < H->n; i++) {  // nominate :665-705 (gatekeeping branches are host-side)
      Entry e; e.head = loadHead(i);
      getAssignments(e);
      e.nominatedMode = e.assignment.RepresentativeMode();
      entries.push_back(std::move(e));
    }
    if (nominate_only) { for (auto& e : entries) e.finalMode = e.nominatedMode; return; }
    std::set<int> preemptedWorkloads;
    if (sn.cfg.fair_sharing) {
      fairIterate(entries, preemptedWorkloads);
    } else {
      std::vector<int> ord = classicalOrder(entries);
      int pos = 0;
      for (int i : ord) { entries[i].order = pos++; processEntry(entries[i], preemptedWorkloads); }
    }
    // entries dropped by the fair-sharing map (duplicate CQ) are never processed; finalMode = nominated
    for (auto& e : entries) if (e.order < 0) e.finalMode = e.nominatedMode;
  }
};

// resource_node.go:183-230 updateCohortResourceNode / accumulateFromChild
static void deriveCohort(const kq_snapshot* s, int nq, int nfr, int cohort, std::vector<int64_t>& sq, std::vector<int64_t>& us, std::vector<uint8_t>& fl) {
  int k = cohort - nq;
  auto ix = [&](int n, int fr) { return (size_t)n * nfr + fr; };
  for (int fr = 0; fr < nfr; fr++) {
    sq[ix(cohort, fr)] = 0; us[ix(cohort, fr)] = 0; fl[ix(cohort, fr)] &= ~KQ_QF_SUBTREE;
    if (fl[ix(cohort, fr)] & KQ_QF_QUOTA) { sq[ix(cohort, fr)] = s->nominal[ix(cohort, fr)]; fl[ix(cohort, fr)] |= KQ_QF_SUBTREE; }
  }
  auto localQuota = [&](int n, int fr) {
    int64_t ll = s->lend_limit[ix(n, fr)];
    if (ll != KQ_NIL_LIMIT) return MaxAmount(Amount(0), Amount(sq[ix(n, fr)]).Sub(Amount(ll)));
    return Amount(0);
  };
  auto accumulate = [&](int child) {
    for (int fr = 0; fr < nfr; fr++) {
      if (fl[ix(child, fr)] & KQ_QF_SUBTREE) {
        Amount delta = Amount(sq[ix(child, fr)]).Sub(localQuota(child, fr));
        sq[ix(cohort, fr)] = Amount(sq[ix(cohort, fr)]).Add(delta).v;
        fl[ix(cohort, fr)] |= KQ_QF_SUBTREE;
      }
      // Usage map keys: every fr with a usage entry; a zero entry contributes max(0, 0-lq)=0
      Amount d = MaxAmount(Amount(0), Amount(us[ix(child, fr)]).Sub(localQuota(child, fr)));
      us[ix(cohort, fr)] = Amount(us[ix(cohort, fr)]).Add(d).v;
    }
  };
  for (int i = s->child_cohort_off[k]; i < s->child_cohort_off[k + 1]; i++) { deriveCohort(s, nq, nfr, s->child_cohort[i], sq, us, fl); accumulate(s->child_cohort[i]); }
  for (int i = s->child_cq_off[k]; i < s->child_cq_off[k + 1]; i++) {
    int cq = s->child_cq[i];
    for (int fr = 0; fr < nfr; fr++) {  // updateClusterQueueResourceNode :167-173
      fl[ix(cq, fr)] &= ~KQ_QF_SUBTREE; sq[ix(cq, fr)] = 0;
      if (fl[ix(cq, fr)] & KQ_QF_QUOTA) { sq[ix(cq, fr)] = s->nominal[ix(cq, fr)]; fl[ix(cq, fr)] |= KQ_QF_SUBTREE; }
    }
    accumulate(cq);
  }
}

}  // namespace kqo

// =================================== C entry points (tests / cpu_baseline) ==========================
using namespace kqo;

static void writeDecisions(Snap& sn, const kq_heads* h, std::vector<Entry>& entries, kq_decisions* out, int* rc) {
  int nR = sn.nR;
  int ntgt = 0, nrsn = 0;
  for (int i = 0; i < h->n; i++) {
    Entry& e = entries[i];
    if (out->status) out->status[i] = (uint8_t)e.status;
    if (out->action) out->action[i] = (uint8_t)e.action;
    if (out->nominated_mode) out->nominated_mode[i] = (uint8_t)e.nominatedMode;
    if (out->mode) out->mode[i] = (uint8_t)e.finalMode;
    if (out->requeue_reason) {
      int rq = e.requeueReason;
      if (e.status != KQ_ST_NOT_NOMINATED && e.status != KQ_ST_ASSUMED && rq == KQ_RQ_GENERIC) rq = KQ_RQ_FAILED_AFTER_NOMINATION;  // scheduler.go:1167-1170
      out->requeue_reason[i] = (uint8_t)rq;
    }
    if (out->skip) out->skip[i] = (uint8_t)e.skip;
    if (out->borrowing) out->borrowing[i] = e.assignment.Borrowing;
    if (out->order) out->order[i] = e.order;
    for (int p = h->ps_off[i]; p < h->ps_off[i + 1]; p++) {
      int lp = p - h->ps_off[i];
      for (int r = 0; r < nR; r++) {
        size_t k = (size_t)p * nR + r;
        if (out->flavor) out->flavor[k] = -1;
        if (out->res_mode) out->res_mode[k] = NoFit;
        if (out->tried_idx) out->tried_idx[k] = -1;
      }
      if (out->ps_count) out->ps_count[p] = h->ps_count[p];
      if (lp < (int)e.assignment.PodSets.size()) {
        PodSetAssignment& psa = e.assignment.PodSets[lp];
        if (out->ps_count) out->ps_count[p] = psa.count;
        for (auto& kv : psa.flavors) {
          size_t k = (size_t)p * nR + kv.first;
          if (out->flavor) out->flavor[k] = kv.second.flavor;
          if (out->res_mode) out->res_mode[k] = (uint8_t)kv.second.mode;
          if (out->tried_idx) out->tried_idx[k] = kv.second.tried;
        }
      }
    }
    if (out->rsn_cap > 0 && out->rsn_off) {
      out->rsn_off[i] = nrsn;
      for (size_t lp = 0; lp < e.assignment.PodSets.size(); lp++)
        for (const Reason& r : e.assignment.PodSets[lp].reasons) {
          if (nrsn >= out->rsn_cap) { *rc = KQ_ECAPACITY; continue; }
          out->rsn_code[nrsn] = (uint8_t)r.code; out->rsn_podset[nrsn] = (uint8_t)lp; out->rsn_flavor[nrsn] = (int16_t)r.flavor; out->rsn_resource[nrsn] = (int16_t)r.resource;
          out->rsn_a[nrsn] = r.a; out->rsn_b[nrsn] = r.b; out->rsn_c[nrsn] = r.c;
          nrsn++;
        }
    }
    if (out->tgt_off) out->tgt_off[i] = ntgt;
    // canonical target order inside an entry: ascending admitted row (the reference tests compare sets)
    std::vector<Target> ts = e.preemptionTargets;
    std::sort(ts.begin(), ts.end(), [](const Target& a, const Target& b) { return a.row < b.row; });
    for (auto& t : ts) {
      if (ntgt >= out->tgt_cap) { *rc = KQ_ECAPACITY; continue; }
      if (out->tgt_adm) out->tgt_adm[ntgt] = t.row;
      if (out->tgt_reason) out->tgt_reason[ntgt] = (uint8_t)t.reason;
      ntgt++;
    }
  }
  if (out->tgt_off) out->tgt_off[h->n] = ntgt;
  if (out->rsn_cap > 0 && out->rsn_off) out->rsn_off[h->n] = nrsn;
}

extern "C" {

// One scheduling cycle on the CPU. stats[0..6] (optional): cells, cell_bytes, head_io, entry, victim, drs, discarded
int kqo_cycle_run(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, kq_decisions* out, int64_t* stats, int64_t* usage_after) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  std::vector<Entry> entries;
  sch.schedule(entries);
  int rc = KQ_OK;
  writeDecisions(sn, h, entries, out, &rc);
  if (stats) { stats[0] = sn.st.cells; stats[1] = sn.st.cell_bytes; stats[2] = sn.st.head_io_bytes; stats[3] = sn.st.entry_bytes; stats[4] = sn.st.victim_bytes; stats[5] = sn.st.drs_bytes; stats[6] = sn.st.discarded_bytes; }
  if (usage_after) memcpy(usage_after, sn.usage.data(), sn.usage.size() * sizeof(int64_t));
  return rc;
}

// One scheduling cycle with Topology-Aware Scheduling inside it (include/kq_cycle_tas.h). tstats[0..2] (optional): TAS placements
// computed, TAS recomputations inside processEntry, 1 when the cycle met a case outside the restated path.
int kqo_cycle_run_tas(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, const kq_cycle_tas* t, kq_decisions* out, kq_cycle_tas_out* tout,
                      int64_t* tstats) {
  Snap sn(*cfg, s);
  sn.attachTAS(t);
  Scheduler sch(sn, h);
  std::vector<Entry> entries;
  sch.schedule(entries);
  int rc = KQ_OK;
  writeDecisions(sn, h, entries, out, &rc);
  int nd = 0;
  tout->dom_off[0] = 0;
  for (int i = 0; i < h->n; i++) {
    Entry& e = entries[i];
    for (int p = h->ps_off[i]; p < h->ps_off[i + 1]; p++) {
      const int lp = p - h->ps_off[i];
      tout->ps_tas[p] = -1;
      if (lp < (int)e.assignment.PodSets.size() && e.assignment.PodSets[lp].hasTopo) {
        const PodSetAssignment& psa = e.assignment.PodSets[lp];
        tout->ps_tas[p] = psa.tasIdx;
        for (auto& dc : psa.topo) {
          if (nd >= tout->dom_cap) { rc = KQ_ECAPACITY; break; }
          tout->dom_leaf[nd] = dc.first; tout->dom_count[nd] = dc.second; nd++;
        }
      }
      tout->dom_off[p + 1] = nd;
    }
  }
  if (tout->tas_usage_after) {
    // the leaf usage after the cycle = the input usage + every Usage.TAS the cycle added: replay on a fresh copy is not needed,
    // the flavor snapshots carry it; exported through a find-independent accessor
    size_t o = 0;
    for (int i = 0; i < t->n_tas; i++) o += tas::snapshot_export_usage(*sn.tasS[i], tout->tas_usage_after + o);
  }
  if (tstats) { tstats[0] = sn.tasFinds; tstats[1] = sn.tasRecomputes; tstats[2] = sn.tasUnsupported ? 1 : 0; }
  return rc;
}

// Scheduler.nominate (scheduler.go:665-705) for every head, nothing else: the checker of kq_nominate_run_resident.
int kqo_nominate_run(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, kq_decisions* out, int64_t* stats) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  std::vector<Entry> entries;
  sch.schedule(entries, true);
  int rc = KQ_OK;
  writeDecisions(sn, h, entries, out, &rc);
  if (stats) { stats[0] = sn.st.cells; stats[1] = sn.st.cell_bytes; stats[2] = sn.st.head_io_bytes; stats[3] = sn.st.entry_bytes; stats[4] = sn.st.victim_bytes; stats[5] = sn.st.drs_bytes; stats[6] = sn.st.discarded_bytes; }
  return rc;
}

// FlavorAssigner.Assign for head `hi` with an optional STUB preemption oracle, as
// TestAssignFlavors drives it (flavorassigner_test.go:159-176,3641-3652): stub_fr[k] ->
// (stub_poss[k], stub_borrow[k]); any other fr -> (Preempt, 0). n_stub < 0 selects the real oracle.
// counts: optional per-podset counts (partial admission).
static int assignImpl(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, const kq_cycle_tas* t, int hi, const int32_t* counts,
               int n_stub, const int32_t* stub_fr, const int32_t* stub_poss, const int32_t* stub_borrow,
               int32_t* flavor, uint8_t* res_mode, int32_t* tried_idx, int32_t* res_borrow,
               int32_t* rep_mode, int32_t* borrowing, int64_t* usage_fr, int32_t* ps_nreasons,
               int32_t rsn_cap, int32_t* rsn_n, int32_t* rsn_rec, int64_t* rsn_abc, int32_t* ps_err);

int kqo_assign(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, int hi, const int32_t* counts,
               int n_stub, const int32_t* stub_fr, const int32_t* stub_poss, const int32_t* stub_borrow,
               int32_t* flavor, uint8_t* res_mode, int32_t* tried_idx, int32_t* res_borrow,
               int32_t* rep_mode, int32_t* borrowing, int64_t* usage_fr /* [n_fr] dense, 0 if absent */, int32_t* ps_nreasons,
               int32_t rsn_cap, int32_t* rsn_n, int32_t* rsn_rec /* [rsn_cap][4]: podset, code, flavor, resource */, int64_t* rsn_abc /* [rsn_cap][3] */) {
  return assignImpl(cfg, s, h, nullptr, hi, counts, n_stub, stub_fr, stub_poss, stub_borrow, flavor, res_mode, tried_idx, res_borrow, rep_mode, borrowing,
                    usage_fr, ps_nreasons, rsn_cap, rsn_n, rsn_rec, rsn_abc, nullptr);
}
// kqo_assign with the TAS half of Assign (t != NULL: assignTAS runs — WorkloadsTopologyRequests, the placements) and PodSetAssignment.Status.err
// per podset (ps_err, may be NULL): what TestAssignFlavors_LeaderWorkerSetTASFlavor reads (flavorassigner_test.go:6036-6050)
int kqo_assign_tas(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, const kq_cycle_tas* t, int hi,
                   int n_stub, const int32_t* stub_fr, const int32_t* stub_poss, const int32_t* stub_borrow,
                   int32_t* flavor, uint8_t* res_mode, int32_t* tried_idx, int32_t* rep_mode, int64_t* usage_fr, int32_t* ps_nreasons, int32_t* ps_err,
                   int32_t rsn_cap, int32_t* rsn_n, int32_t* rsn_rec, int64_t* rsn_abc) {
  int32_t borrowing = 0;
  return assignImpl(cfg, s, h, t, hi, nullptr, n_stub, stub_fr, stub_poss, stub_borrow, flavor, res_mode, tried_idx, nullptr, rep_mode, &borrowing,
                    usage_fr, ps_nreasons, rsn_cap, rsn_n, rsn_rec, rsn_abc, ps_err);
}
} // extern "C"
static int assignImpl(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, const kq_cycle_tas* t, int hi, const int32_t* counts,
               int n_stub, const int32_t* stub_fr, const int32_t* stub_poss, const int32_t* stub_borrow,
               int32_t* flavor, uint8_t* res_mode, int32_t* tried_idx, int32_t* res_borrow,
               int32_t* rep_mode, int32_t* borrowing, int64_t* usage_fr, int32_t* ps_nreasons,
               int32_t rsn_cap, int32_t* rsn_n, int32_t* rsn_rec, int64_t* rsn_abc, int32_t* ps_err) {
  Snap sn(*cfg, s);
  if (t) sn.attachTAS(t);
  Scheduler sch(sn, h);
  Head wl = sch.loadHead(hi);
  OracleFn orc;
  if (n_stub >= 0) {
    orc = [=](int, const Head&, int fr, Amount) -> std::pair<int, int> {
      for (int k = 0; k < n_stub; k++) if (stub_fr[k] == fr) return {stub_poss[k], stub_borrow[k]};
      return {ppPreempt, 0};
    };
  } else {
    orc = sch.makeOracle();
  }
  FlavorAssigner fa{sn, h, wl, wl.cq, cfg->fair_sharing != 0, orc};
  std::vector<int> cv;
  if (counts) cv.assign(counts, counts + wl.ps.size());
  Assignment a = fa.assignFlavors(counts ? &cv : nullptr);
  int P = (int)wl.ps.size();
  for (int p = 0; p < P; p++) {
    for (int r = 0; r < sn.nR; r++) { size_t k = (size_t)p * sn.nR + r; flavor[k] = -1; res_mode[k] = NoFit; tried_idx[k] = -1; if (res_borrow) res_borrow[k] = 0; }
    if (ps_nreasons) ps_nreasons[p] = 0;
    if (p < (int)a.PodSets.size()) {
      if (ps_nreasons) ps_nreasons[p] = a.PodSets[p].nreasons;
      for (auto& kv : a.PodSets[p].flavors) { size_t k = (size_t)p * sn.nR + kv.first; flavor[k] = kv.second.flavor; res_mode[k] = (uint8_t)kv.second.mode; tried_idx[k] = kv.second.tried; if (res_borrow) res_borrow[k] = kv.second.borrow; }
    }
  }
  *rep_mode = a.RepresentativeMode();
  *borrowing = a.Borrowing;
  if (usage_fr) { for (int fr = 0; fr < sn.nfr; fr++) usage_fr[fr] = 0; for (auto& kv : a.Usage) usage_fr[kv.first] = kv.second.v; }
  if (rsn_n) {
    int n = 0;
    for (size_t p = 0; p < a.PodSets.size(); p++)
      for (const Reason& r : a.PodSets[p].reasons) {
        if (n < rsn_cap) { rsn_rec[4 * n] = (int)p; rsn_rec[4 * n + 1] = r.code; rsn_rec[4 * n + 2] = r.flavor; rsn_rec[4 * n + 3] = r.resource; rsn_abc[3 * n] = r.a; rsn_abc[3 * n + 1] = r.b; rsn_abc[3 * n + 2] = r.c; }
        n++;
      }
    *rsn_n = n;
  }
  if (ps_err) for (int p = 0; p < P; p++) ps_err[p] = p < (int)a.PodSets.size() && a.PodSets[p].err ? 1 : 0;
  return KQ_OK;
}
extern "C" {


// Assign with features.UnadmittedWorkloadsObservability on: the FlavorAssignmentAttempts of every podset — (podset, flavor, mode, label), label
// = the severity rank of the attempt's NoFitReason (lbl* above) — and Assignment.NoFitReason, as TestIsNoFitDueToCapacityAndLimits reads them
// (flavorassigner_test.go:5772-5793). t may be NULL (no TAS flavors); stub as in kqo_assign.
int kqo_assign_attempts(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, const kq_cycle_tas* t, int hi,
                        int n_stub, const int32_t* stub_fr, const int32_t* stub_poss, const int32_t* stub_borrow,
                        int32_t att_cap, int32_t* att_n, int32_t* att_rec /* [att_cap][4] */, int32_t* no_fit_label, int32_t* rep_mode) {
  Snap sn(*cfg, s);
  if (t) sn.attachTAS(t);
  Scheduler sch(sn, h);
  Head wl = sch.loadHead(hi);
  OracleFn orc;
  if (n_stub >= 0) {
    orc = [=](int, const Head&, int fr, Amount) -> std::pair<int, int> {
      for (int k = 0; k < n_stub; k++) if (stub_fr[k] == fr) return {stub_poss[k], stub_borrow[k]};
      return {ppPreempt, 0};
    };
  } else {
    orc = sch.makeOracle();
  }
  FlavorAssigner fa{sn, h, wl, wl.cq, cfg->fair_sharing != 0, orc};
  fa.observe = true;
  Assignment a = fa.assignFlavors(nullptr);
  int n = 0;
  for (size_t p = 0; p < a.PodSets.size(); p++)
    for (const Attempt& at : a.PodSets[p].attempts) {
      if (n < att_cap) { att_rec[4 * n] = (int)p; att_rec[4 * n + 1] = at.flavor; att_rec[4 * n + 2] = at.mode; att_rec[4 * n + 3] = at.label; }
      n++;
    }
  *att_n = n;
  *no_fit_label = a.noFitLabel;
  *rep_mode = a.RepresentativeMode();
  return sn.tasUnsupported ? KQ_EUNSUPPORTED : KQ_OK;
}

// Preemptor.GetTargets for head `hi` given an explicit assignment (flavor + mode per (podset,resource)),
// as TestPreemption drives it (preemption_test.go:4093-4170). Checks the snapshot is restored exactly.
// probe of runFirstFsStrategy (see g_fs_probe): on != 0 clears and enables, on == 0 disables; out (may be null) receives the 288 counters
void kqo_fs_probe(int on, int64_t* out288) {
  if (out288) for (int i = 0; i < 288; i++) out288[i] = g_fs_probe[i];
  if (on) for (int i = 0; i < 288; i++) g_fs_probe[i] = 0;
  g_fs_probe_on = on != 0;
}
static thread_local int64_t g_last_fs_counters[3] = {0, 0, 0};
// the three log-line counters of the last kqo_get_targets on this thread (see Stats::fs_first_cq_evals)
void kqo_last_fs_counters(int64_t* out3) { for (int i = 0; i < 3; i++) out3[i] = g_last_fs_counters[i]; }
int kqo_get_targets(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, int hi,
                    const int32_t* flavor, const uint8_t* res_mode, int32_t cap, int32_t* tgt_adm, uint8_t* tgt_reason, int32_t* n_out) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  Head wl = sch.loadHead(hi);
  Assignment a;
  for (size_t p = 0; p < wl.ps.size(); p++) {
    PodSetAssignment psa; psa.count = wl.ps[p].count; psa.nreasons = 1;
    for (int r = 0; r < sn.nR; r++) { size_t k = p * sn.nR + r; if (flavor[k] >= 0) { FlavorAssignment f; f.flavor = flavor[k]; f.mode = res_mode[k]; psa.flavors[r] = f; } }
    a.PodSets.push_back(psa);
  }
  std::vector<int64_t> before = sn.usage;
  std::vector<Target> t = sch.preemptor.GetTargets(wl, a);
  g_last_fs_counters[0] = sn.st.fs_first_cq_evals; g_last_fs_counters[1] = sn.st.fs_second_evals; g_last_fs_counters[2] = sn.st.fs_skipped_queues;
  if (before != sn.usage) return KQ_EINVAL;  // snapshot-restoration invariant (preemption_test.go:4172)
  for (auto x : sn.removed) if (x) return KQ_EINVAL;
  std::sort(t.begin(), t.end(), [](const Target& x, const Target& y) { return x.row < y.row; });
  *n_out = (int)t.size();
  for (size_t i = 0; i < t.size() && (int)i < cap; i++) { tgt_adm[i] = t[i].row; tgt_reason[i] = (uint8_t)t[i].reason; }
  return (int)t.size() > cap ? KQ_ECAPACITY : KQ_OK;
}

// Recompute SubtreeQuota (all nodes), cohort Usage and the SUBTREE flag from Quotas + CQ usage.
int kqo_derive(const kq_snapshot* s, int64_t* subtree_quota, int64_t* usage, uint8_t* quota_flags) {
  int nq = s->n_cq, N = nq + s->n_cohort, nfr = s->n_flavor * s->n_resource;
  std::vector<int64_t> sq((size_t)N * nfr, 0), us(s->usage, s->usage + (size_t)N * nfr);
  std::vector<uint8_t> fl(s->quota_flags, s->quota_flags + (size_t)N * nfr);
  for (int cq = 0; cq < nq; cq++) for (int fr = 0; fr < nfr; fr++) {  // parentless CQs too
    size_t k = (size_t)cq * nfr + fr;
    fl[k] &= ~KQ_QF_SUBTREE; sq[k] = 0;
    if (fl[k] & KQ_QF_QUOTA) { sq[k] = s->nominal[k]; fl[k] |= KQ_QF_SUBTREE; }
  }
  for (int c = nq; c < N; c++) if (s->parent[c] < 0) deriveCohort(s, nq, nfr, c, sq, us, fl);
  memcpy(subtree_quota, sq.data(), sq.size() * 8);
  memcpy(usage, us.data(), us.size() * 8);
  memcpy(quota_flags, fl.data(), fl.size());
  return KQ_OK;
}

// Quota math probes: what[0]=Available what[1]=PotentialAvailable what[2]=LocalAvailable
// what[3]=borrow height for `val` what[4]=mayReclaim
int kqo_quota_probe(const kq_config* cfg, const kq_snapshot* s, int cq, int fr, int64_t val, int64_t* what) {
  Snap sn(*cfg, s);
  what[0] = sn.Available(cq, fr).v; what[1] = sn.PotentialAvailable(cq, fr).v; what[2] = sn.LocalAvailable(cq, fr).v;
  auto hb = sn.FindHeightOfLowestSubtreeThatFits(cq, fr, Amount(val));
  what[3] = hb.first; what[4] = hb.second;
  return KQ_OK;
}

// Apply a sequence of AddWorkload(+row) / RemoveWorkload(-(row+1)) to the snapshot and return usage.
// Closed loop (checker for kq_cycle_commit / kq_cycle_release): run the cycle, then fold the usage of every admitted
// workload into the snapshot's ORIGINAL usage (cache side of assumeWorkload: clusterqueue.go:594 -> resource_node.go:144).
// The (cq, fr, qty) triples that were added come back so that a later release can be replayed with kqo_usage_apply.
int kqo_cycle_commit(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, int64_t* usage_out, int32_t* n_admitted,
                     int32_t cap, int32_t* t_cq, int32_t* t_fr, int64_t* t_qty, int32_t* n_triples) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  std::vector<Entry> entries;
  sch.schedule(entries);
  Snap fresh(*cfg, s);
  int na = 0, nt = 0;
  for (auto& e : entries) {
    if (e.action != KQ_ACT_ADMIT) continue;
    na++;
    FRQ u = sch.assignmentUsage(e);
    fresh.AddUsage(e.head.cq, u);
    for (auto& kv : u) { if (nt >= cap) return KQ_ECAPACITY; t_cq[nt] = e.head.cq; t_fr[nt] = kv.first; t_qty[nt] = kv.second.v; nt++; }
  }
  memcpy(usage_out, fresh.usage.data(), fresh.usage.size() * 8);
  *n_admitted = na; *n_triples = nt;
  return KQ_OK;
}
int kqo_usage_apply(const kq_config* cfg, const kq_snapshot* s, int n, const int32_t* cq, const int32_t* fr, const int64_t* qty, int add, int64_t* usage_out) {
  Snap sn(*cfg, s);
  for (int i = 0; i < n; i++) { if (add) sn.addUsage(cq[i], fr[i], Amount(qty[i])); else sn.removeUsage(cq[i], fr[i], Amount(qty[i])); }
  memcpy(usage_out, sn.usage.data(), sn.usage.size() * 8);
  return KQ_OK;
}

int kqo_apply_ops(const kq_config* cfg, const kq_snapshot* s, int n_ops, const int32_t* ops, int64_t* usage_out) {
  Snap sn(*cfg, s);
  for (int i = 0; i < n_ops; i++) { if (ops[i] >= 0) sn.AddWorkload(ops[i]); else sn.RemoveWorkload(-ops[i] - 1); }
  memcpy(usage_out, sn.usage.data(), sn.usage.size() * 8);
  return KQ_OK;
}

// dominantResourceShare(node): ratio (unweighted), weighted precise share, rounded share
// (roundedWeightedShare fair_sharing.go:133-141), dominant resource, borrowing flag.
int kqo_drs(const kq_config* cfg, const kq_snapshot* s, int node, const int64_t* wl_req /* [n_fr] or NULL */, double* unweighted, double* precise, int64_t* rounded, int32_t* dominant, int32_t* borrowing) {
  Snap sn(*cfg, s);
  FRQ req;
  if (wl_req) for (int fr = 0; fr < sn.nfr; fr++) if (wl_req[fr] != 0) req[fr] = Amount(wl_req[fr]);
  DRS d = dominantResourceShare(sn, node, wl_req ? &req : nullptr);
  *unweighted = d.unweightedRatio; *precise = d.PreciseWeightedShare();
  *rounded = d.zeroWeightBorrows() ? I64MAX : (int64_t)std::ceil(d.PreciseWeightedShare());
  *dominant = d.dominantResource; *borrowing = d.borrowing;
  return KQ_OK;
}

// calculateLendable(node) per resource
int kqo_lendable(const kq_config* cfg, const kq_snapshot* s, int node, int64_t* out) {
  Snap sn(*cfg, s);
  auto l = calculateLendable(sn, node);
  for (int r = 0; r < sn.nR; r++) out[r] = l[r].v;
  return KQ_OK;
}

int kqo_is_preferred(int a_pm, int64_t a_borrow, int b_pm, int64_t b_borrow, uint32_t policy) {
  return isPreferred({a_pm, a_borrow}, {b_pm, b_borrow}, policy) ? 1 : 0;
}

// Hooks that replay the reference's small table tests directly against the restated functions
// (TestResourcesToReserve scheduler_test.go:8692, TestLastAssignmentOutdated :9216, TestEntryOrdering :6793,
//  TestCandidatesOrdering preemption_test.go:4613).
// quotaResourcesToReserve (scheduler.go:796-814) for head 0 of `h` with the given representative mode / borrowing and
// assignment usage; reserved[] is dense over flavor-resources (entries not in the usage stay 0).
int kqo_resources_to_reserve(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, int mode, int borrowing,
                             int n, const int32_t* fr, const int64_t* qty, int64_t* reserved) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  Entry e; e.head = sch.loadHead(0);
  e.assignment.PodSets.resize(1);
  for (int i = 0; i < n; i++) {
    e.assignment.Usage[fr[i]] = Amount(qty[i]);
    FlavorAssignment fa; fa.flavor = fr[i] / sn.nR; fa.mode = mode; fa.borrow = borrowing;
    e.assignment.PodSets[0].flavors[fr[i] % sn.nR] = fa;
  }
  e.assignment.Borrowing = borrowing;
  e.assignment.SetRepresentativeMode(mode);  // the test harness sets it the same way (scheduler_test.go:8839-8850)
  FRQ r = sch.quotaResourcesToReserve(e, e.head.cq);
  for (int f = 0; f < sn.nfr; f++) reserved[f] = 0;
  for (auto& kv : r) reserved[kv.first] = kv.second.v;
  return KQ_OK;
}
// lastAssignmentOutdated (scheduler.go:840-856) for head 0
int kqo_last_assignment_outdated(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  Head hd = sch.loadHead(0);
  return sch.lastAssignmentOutdated(hd) ? 1 : 0;
}
// makeClassicalIterator (scheduler.go:1110-1163): order[i] = head index at position i, given each head's Borrows()
int kqo_entry_order(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, const int32_t* borrowing, int32_t* order) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  std::vector<Entry> entries(h->n);
  for (int i = 0; i < h->n; i++) { entries[i].head = sch.loadHead(i); entries[i].assignment.Borrowing = borrowing[i]; }
  std::vector<int> ord = sch.classicalOrder(entries);
  for (int i = 0; i < h->n; i++) order[i] = ord[i];
  return KQ_OK;
}
// CandidatesOrdering (preemption/common/ordering.go:42-83): the admitted rows `rows` sorted for a preemptor in `cq`
int kqo_candidates_order(const kq_config* cfg, const kq_snapshot* s, int cq, int n, const int32_t* rows, int32_t* out) {
  Snap sn(*cfg, s);
  Preemptor p(sn);
  std::vector<int> v(rows, rows + n);
  p.sortCandidates(v, cq);
  for (int i = 0; i < n; i++) out[i] = v[i];
  return KQ_OK;
}

// TargetClusterQueueOrdering.Iter() (preemption/fairsharing/ordering.go:92-226) driven like the reference's
// TestMakeClusterQueueOrdering (ordering_test.go:255-268): every yielded target is either dropped (DropQueue) or loses its first
// candidate (PopWorkload), as actions[i] says (1 = drop; beyond n_actions: pop). Returns the yielded ClusterQueues in order.
int kqo_cq_ordering(const kq_config* cfg, const kq_snapshot* s, int32_t preemptor_cq, int32_t n_cand, const int32_t* cand_rows,
                    int32_t n_actions, const uint8_t* actions, int32_t cap, int32_t* out_cq, int32_t* out_n) {
  Snap sn(*cfg, s);
  Preemptor p(sn);
  std::vector<int> cands(cand_rows, cand_rows + n_cand);
  Preemptor::Ordering ord = p.MakeClusterQueueOrdering(preemptor_cq, cands);
  int n = 0;
  for (int cq = p.orderingNext(ord); cq >= 0; cq = p.orderingNext(ord)) {
    if (n >= cap) return KQ_ECAPACITY;
    out_cq[n] = cq;
    if (n < n_actions && actions[n]) ord.prunedClusterQueues.insert(cq);  // DropQueue ordering.go:129-131
    else ord.PopWorkload(cq);
    n++;
  }
  *out_n = n;
  return KQ_OK;
}

// resources.Amount arithmetic (pkg/resources/amount.go:114-186) for the reference's TestAmountArithmetic known answers.
// op: 0 Add, 1 AddInt64, 2 Sub, 3 SubInt64, 4 Cmp, 5 CmpInt64
int kqo_amount_op(int32_t op, int64_t a, int64_t b, int64_t* out) {
  const Amount x(a), y(b);
  switch (op) {
    case 0: *out = x.Add(y).v; break;
    case 1: *out = x.AddInt64(b).v; break;
    case 2: *out = x.Sub(y).v; break;
    case 3: *out = x.SubInt64(b).v; break;
    case 4: *out = x.Cmp(y); break;
    case 5: *out = x.CmpInt64(b); break;
    default: return KQ_EINVAL;
  }
  return KQ_OK;
}

// PodSetReducer.Search with the predicate of the reference's TestSearch (podset_reducer_test.go:127-134): sum(counts) <= limit
int kqo_podset_reducer_search(int32_t n, const int32_t* counts, const int32_t* min_counts, int32_t count_limit, int32_t* out_count, int32_t* out_found) {
  std::vector<int> full(counts, counts + n), mins(min_counts, min_counts + n);
  int64_t last = 0;
  const bool found = podSetReducerSearch(full, mins, [&](const std::vector<int>& cur) {
    int64_t total = 0;
    for (int v : cur) total += v;
    if (total <= count_limit) { last = total; return true; }
    return false;
  });
  *out_count = (int32_t)last; *out_found = found ? 1 : 0;
  return 0;
}
// SatisfiesPreemptionPolicy (preemption/common/preemption_policy.go:27-42) on effective priorities and queue-order timestamps
int kqo_satisfies_preemption_policy(int64_t preemptor_priority, int64_t preemptor_ts, int64_t candidate_priority, int64_t candidate_ts, int32_t policy) {
  return Preemptor::satisfiesPreemptionPolicy(preemptor_priority, preemptor_ts, candidate_priority, candidate_ts, policy) ? 1 : 0;
}

// ---- measurement, not a restatement: how deep is the dependency chain of processEntry? -----------------------------------------
// Runs the cycle's nomination once, then (a) the reference's sequential walk (scheduler.go:356-360) recording per entry what it
// did, and (b) a JACOBI iteration over the same entries: in round r every entry is processed against the state the outcomes of
// round r-1 leave in front of it (base usage + the usage the earlier entries added + the rows they preempted), in the order the
// iterator yields on that state (classical: the static sort; fair sharing: the DRS tournament on the speculative state). Entry
// 0 is exact in round 0; an entry is final once everything in front of it is final and unchanged, so the final prefix grows by
// at least one per round and the fixed point is the sequential result. rounds[r] = {final prefix after round r, entries whose
// outcome changed against round r-1, first changed position}. per_entry[i] = {position, nominated targets, overlap/TAS
// recompute flag, final targets, final status, action, searches of its processEntry, candidates listed, rows removed before the
// fill-back, round in which its outcome last changed, recompute against (cycle-start usage, true preempted set) differs? 0/1/-1}.
struct ProbeOutcome {
  bool valid = false;
  int status = 0, skip = 0, finalMode = 0, action = 0, requeue = 0;
  std::vector<int> targets;
  FRQ effUsage; bool effInsert = false;
  bool same(const ProbeOutcome& o) const {
    if (valid != o.valid || status != o.status || skip != o.skip || finalMode != o.finalMode || action != o.action || requeue != o.requeue || effInsert != o.effInsert || targets != o.targets) return false;
    if (effUsage.size() != o.effUsage.size()) return false;
    auto a = effUsage.begin(); auto b = o.effUsage.begin();
    for (; a != effUsage.end(); ++a, ++b) if (a->first != b->first || a->second.v != b->second.v) return false;
    return true;
  }
};
static ProbeOutcome probeOutcomeOf(const Entry& e) {
  ProbeOutcome o; o.valid = true; o.status = e.status; o.skip = e.skip; o.finalMode = e.finalMode; o.action = e.action; o.requeue = e.requeueReason;
  for (auto& t : e.preemptionTargets) o.targets.push_back(t.row);
  std::sort(o.targets.begin(), o.targets.end());
  o.effUsage = e.effUsage; o.effInsert = e.effInsert;
  return o;
}
int kqo_jacobi_probe(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, int32_t max_rounds, int32_t* per_entry /* [n][12] */,
                     int32_t* rounds /* [max_rounds][3] */, int32_t* n_rounds, int32_t* converged) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  std::vector<Entry> nominated;
  sch.schedule(nominated, true);
  const int n = h->n;
  const bool fair = sn.cfg.fair_sharing != 0;
  const std::vector<int64_t> baseUsage = sn.usage;
  std::vector<int> classical;
  if (!fair) classical = sch.classicalOrder(nominated);
  // the iterator on the current state: classical = position in the static order; fair = getCq + computeDRS + tournament
  auto nextEntry = [&](std::vector<Entry>& ents, std::map<int, int>& cqToEntry, int pos) -> int {
    if (!fair) return classical[pos];
    int cq = cqToEntry.begin()->first, w;
    if (!sn.HasParent(cq)) w = cqToEntry[cq];
    else { int root = sn.Root(cq); sch.computeDRS(root, ents, cqToEntry); w = sch.runTournament(root, ents, cqToEntry); }
    cqToEntry.erase(ents[w].head.cq);
    return w;
  };
  auto freshMap = [&](std::map<int, int>& m) { m.clear(); if (fair) for (int i = 0; i < n; i++) m[nominated[i].head.cq] = i; };
  const int npos = [&] { std::map<int, int> m; freshMap(m); return fair ? (int)m.size() : n; }();
  // (a) the sequential walk
  std::vector<ProbeOutcome> truth(n);
  std::vector<int> trueSeq;
  for (int i = 0; i < n; i++) for (int k = 0; k < 12; k++) per_entry[(size_t)i * 12 + k] = -1;
  {
    std::vector<Entry> ents = nominated;
    std::map<int, int> m; freshMap(m);
    std::set<int> preempted;
    for (int pos = 0; pos < npos; pos++) {
      const int i = nextEntry(ents, m, pos);
      trueSeq.push_back(i);
      int32_t* pe = per_entry + (size_t)i * 12;
      pe[0] = pos; pe[1] = (int)ents[i].preemptionTargets.size();
      // (c) would the outcome be the same against {cycle-start usage, the true preempted set}? — i.e. does only the preempted
      // set carry the dependency, or also the usage the earlier entries added
      int differs = -1;
      const bool overlap = Scheduler::hasAny(preempted, ents[i].preemptionTargets);
      if (overlap) {
        std::vector<int64_t> saveU = sn.usage;
        sn.usage = baseUsage;
        Entry e2 = ents[i]; std::set<int> p2 = preempted;
        sch.processEntry(e2, p2);
        ProbeOutcome o2 = probeOutcomeOf(e2);
        sn.usage = saveU;
        Entry e3 = ents[i]; std::set<int> p3 = preempted;
        std::vector<int64_t> saveU3 = sn.usage;
        sch.processEntry(e3, p3);
        sn.usage = saveU3;
        differs = o2.same(probeOutcomeOf(e3)) ? 0 : 1;
      }
      const int64_t s0 = sn.st.searches, c0 = sn.st.cand_listed, r0 = sn.st.cand_removed;
      sch.processEntry(ents[i], preempted);
      truth[i] = probeOutcomeOf(ents[i]);
      pe[2] = ents[i].recomputed; pe[3] = (int)ents[i].preemptionTargets.size(); pe[4] = ents[i].status; pe[5] = ents[i].action;
      pe[6] = (int)(sn.st.searches - s0); pe[7] = (int)(sn.st.cand_listed - c0); pe[8] = (int)(sn.st.cand_removed - r0);
      pe[10] = differs;
    }
  }
  // (b) Jacobi rounds
  std::vector<ProbeOutcome> prev(n), cur(n);
  std::vector<int> prevSeq;
  *n_rounds = 0; *converged = 0;
  for (int r = 0; r < max_rounds; r++) {
    sn.usage = baseUsage;
    std::fill(sn.removed.begin(), sn.removed.end(), 0);
    std::vector<Entry> ents = nominated;
    std::map<int, int> m; freshMap(m);
    std::set<int> preempted;
    std::vector<int> seq;
    for (int pos = 0; pos < npos; pos++) {
      const int i = nextEntry(ents, m, pos);
      seq.push_back(i);
      std::vector<int64_t> saveU = sn.usage;
      std::set<int> p2 = preempted;
      Entry e = nominated[i];
      sch.processEntry(e, p2);
      cur[i] = probeOutcomeOf(e);
      sn.usage = saveU;
      if (prev[i].valid) {  // the state the NEXT entries see comes from last round's outcome of this one
        if (prev[i].effInsert) for (int row : prev[i].targets) preempted.insert(row);
        sn.AddUsage(e.head.cq, prev[i].effUsage);
      }
    }
    int changed = 0, first = npos;
    for (int pos = 0; pos < npos; pos++) {
      const int i = seq[pos];
      const bool ch = (r > 0 && prevSeq[pos] != i) || !cur[i].same(prev[i]);
      if (ch) { changed++; if (pos < first) first = pos; per_entry[(size_t)i * 12 + 9] = r; }
    }
    rounds[3 * r] = first >= npos ? npos : first + 1; rounds[3 * r + 1] = changed; rounds[3 * r + 2] = first;
    *n_rounds = r + 1;
    prev = cur; prevSeq = seq;
    if (changed == 0) { *converged = 1; break; }
  }
  if (*converged) {  // the fixed point must be the sequential result
    for (int pos = 0; pos < npos; pos++) if (prevSeq[pos] != trueSeq[pos] || !prev[trueSeq[pos]].same(truth[trueSeq[pos]])) return KQ_EINVAL;
  } else {
    // how much of the sequential result has the last round reached?
    int ok = 0;
    for (int pos = 0; pos < npos; pos++) { if (prevSeq[pos] != trueSeq[pos] || !prev[trueSeq[pos]].same(truth[trueSeq[pos]])) break; ok++; }
    per_entry[11] = ok;
  }
  return KQ_OK;
}

// entryComparer.less (fair_sharing_iterator.go:176-221) with INJECTED DRS values, as TestEntryComparerLess (scheduler_test.go:8411-8683)
// drives it: heads 0 / 1 of `h` are a / b (queue_ts, KQ_HEAD_IS_PREEMPTOR, priority). kind[i]: -1 no entry in drsValues (the map's
// zero value DRS{}), 0 DRS{} stored explicitly, 1 schdcache.BorrowingDRS(fr 0) (fair_sharing_test_util.go:22), 2 schdcache.NegativeDRS()
// (fair_sharing.go:58). req[i]: requestedFRs[entry][fr 0] (0 = no entry).
int kqo_entry_less(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, const int32_t* kind, const int64_t* req) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  std::vector<Entry> entries(2);
  for (int i = 0; i < 2; i++) entries[i].head = sch.loadHead(i);
  const int cohort = sn.N;  // any key: the comparison only looks the two entries up under it
  for (int i = 0; i < 2; i++) {
    DRS d;
    d.fairWeight = 0;                                   // Go zero value
    if (kind[i] == 1) { d.fairWeight = 1.0; d.borrowing = true; d.borrowedFRs = {0}; }
    if (kind[i] == 2) d = NegativeDRS();
    if (kind[i] >= 0) sch.drsValues[{cohort, i}] = d;
    if (req[i] > 0) sch.requestedFRs[i][0] = Amount(req[i]);
  }
  return sch.less(entries, 0, 1, cohort) ? 1 : 0;
}
// scheduler.fits (scheduler.go:771-777) as TestFitsDedupsOverlappingVictims (scheduler_test.go:9379-9450) calls it: the incoming usage of
// ClusterQueue `cq` against the snapshot with `preempted` ∪ `targets` removed ONCE each. Returns the FitsCheckResult (0 Ok, 1 NoQuota).
int kqo_fits_check(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, int32_t cq, int32_t n_use, const int32_t* fr, const int64_t* qty,
                   int32_t n_pre, const int32_t* pre_rows, int32_t n_tgt, const int32_t* tgt_rows) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  Entry e;
  FRQ usage;
  for (int i = 0; i < n_use; i++) usage[fr[i]] = Amount(qty[i]);
  std::set<int> preempted(pre_rows, pre_rows + n_pre);
  std::vector<Target> targets;
  for (int i = 0; i < n_tgt; i++) targets.push_back({tgt_rows[i], 0});
  const std::vector<int64_t> before = sn.usage;
  const int rc = sch.fitsCheck(e, cq, usage, preempted, targets);
  if (before != sn.usage) return KQ_EINVAL;
  return rc;
}

// Assignment.TotalRequestsFor (flavorassigner.go:267-296, with a replaced workload slice :265) as TestAssignment_TotalRequestsFor
// (flavorassigner_test.go:4645) drives it: head `hi`, the assignment's per-podset counts and the flavor of every (podset, resource) cell
// (-1: no flavor for that resource). usage_fr: dense [n_fr], 0 where absent.
int kqo_total_requests_for(const kq_config* cfg, const kq_snapshot* s, const kq_heads* h, int32_t hi, const int32_t* counts, const int32_t* flavor, int64_t* usage_fr) {
  Snap sn(*cfg, s);
  Scheduler sch(sn, h);
  Head wl = sch.loadHead(hi);
  Assignment a;
  for (size_t p = 0; p < wl.ps.size(); p++) {
    PodSetAssignment psa; psa.count = counts[p];
    for (int r = 0; r < sn.nR; r++) if (flavor[p * sn.nR + r] >= 0) { FlavorAssignment f; f.flavor = flavor[p * sn.nR + r]; f.mode = Fit; psa.flavors[r] = f; }
    a.PodSets.push_back(psa);
  }
  FRQ u = sch.preemptor.TotalRequestsFor(wl, a);
  for (int fr = 0; fr < sn.nfr; fr++) usage_fr[fr] = 0;
  for (auto& kv : u) usage_fr[kv.first] = kv.second.v;
  return KQ_OK;
}
}  // extern "C"

// kq_tas_cycle.hpp — Topology-Aware Scheduling INSIDE the scheduling cycle (include/kq_cycle_tas.h, kq_cycle_run_tas).
//
// The cycle's own code (kq_device.hpp) is compiled a second time with KQ_TAS_CYCLE defined (translation unit
// kq_tas_cycle_kernel.hip, kernels k_nominate_tas / k_process_tas); the hooks it then calls are defined here:
//   assign_flavors      -> tc_assign_tas          flavorassigner.go:864-903 (assignTAS)
//   get_targets         -> tc_search_begin / tc_search_row / tc_search_fits / tc_search_end   preemption.go:135-138, :669-684
//   get_assignments     -> tc_update_assignment   scheduler.go:941-985 (updateAssignmentForTAS)
//   publish_assignment  -> tc_publish             the head's TopologyAssignments for processEntry and the host
//   k_process_tas       -> process_entry_tas      scheduler.go:392-523 + :707-769 with the TAS side of Fits / AddUsage
// The placement itself is kq_tas_device.hpp's t_workload (FindTopologyAssignmentsForFlavor :578), called in-wave on a request block
// the wave builds from the head's current flavor assignment. The ordinary cycle's kernels do not contain any of this.
//
// Leaf usage of a TAS flavor ([n_leaves][R] int64) exists in three planes: `base` (cycle start: tas_usage + every admitted row's
// TopologyDomainRequests — what nominate reads), `work` (processEntry's snapshot) and `np` (work minus the rows preempted so far this
// cycle: what scheduler.fits and an overlap recomputation see). A victim search runs on a private copy per wave slot.
#pragma once
#include "../../include/kq_cycle_tas.h"
#include "kq_tas_device.hpp"

namespace kq {

constexpr int TC_P = KQ_MAXPS;
constexpr int TC_ML = KQ_TAS_MAX_LEVELS;
// a slot's int32 request / result block (one FindTopologyAssignmentsForFlavor call in flight per wave)
enum {
  TQ_WLOFF = 0, TQ_COUNT = 2, TQ_LEVEL = TQ_COUNT + TC_P, TQ_SSIZE = TQ_LEVEL + TC_P, TQ_SLEVEL = TQ_SSIZE + TC_P, TQ_GROUP = TQ_SLEVEL + TC_P,
  TQ_NLAY = TQ_GROUP + TC_P, TQ_LLEVEL = TQ_NLAY + TC_P, TQ_LSIZE = TQ_LLEVEL + TC_P * TC_ML, TQ_STATUS = TQ_LSIZE + TC_P * TC_ML,
  TQ_OPA = TQ_STATUS + TC_P, TQ_OPB = TQ_OPA + TC_P, TQ_DPOS = TQ_OPB + TC_P, TQ_DN = TQ_DPOS + TC_P, TQ_MISC = TQ_DN + TC_P,   // misc: pool_used, error, bytes (64 bit), then the timing builds' cycle counters
  TQ_LO = TQ_MISC + 24, TQ_HI = TQ_LO + TC_P,   // second pass: the leaf range of a replacement (global block only)
  TQ_MASK = TQ_HI + TC_P, TQ_WORDS = TQ_MASK + TC_P   // node feasibility: the request's row of TCyc::leaf_mask, -1 = none (global block only)
};
static_assert(TQ_MISC % 2 == 0, "the misc words hold an aligned 64-bit byte counter");
// where the fields of a request block start: the slot's block in global memory holds TC_P podsets, the block k_process_tas keeps in LDS one
struct TQOff { int count, level, ssize, slevel, group, nlay, llevel, lsize, status, opa, opb, dpos, dn, misc; };
KQ_DEV TQOff tq_full() { return TQOff{TQ_COUNT, TQ_LEVEL, TQ_SSIZE, TQ_SLEVEL, TQ_GROUP, TQ_NLAY, TQ_LLEVEL, TQ_LSIZE, TQ_STATUS, TQ_OPA, TQ_OPB, TQ_DPOS, TQ_DN, TQ_MISC}; }
KQ_DEV TQOff tq_one() { return TQOff{2, 3, 4, 5, 6, 7, 8, 8 + TC_ML, 8 + 2 * TC_ML, 9 + 2 * TC_ML, 10 + 2 * TC_ML, 11 + 2 * TC_ML, 12 + 2 * TC_ML, 14 + 2 * TC_ML}; }
// the LDS block of k_process_tas behind the placement's working state (tas_lds_layout): request block of one podset, its flags, its
// per-pod requests, the wave's two domain stores of TX_DCAP entries each
constexpr int TX_QWORDS = 14 + 2 * TC_ML + 24, TX_DCAP = 64;
static_assert((14 + 2 * TC_ML) % 2 == 0, "the misc words hold an aligned 64-bit byte counter");
constexpr size_t TX_Q = 0, TX_QU = TX_Q + (size_t)TX_QWORDS * 4, TX_QS = TX_QU + 16, TX_DL = TX_QS + (size_t)KQ_TAS_MAXR * 8, TX_DC = TX_DL + (size_t)2 * TX_DCAP * 4,
                 TX_BYTES = (TX_DC + (size_t)2 * TX_DCAP * 4 + 15) & ~(size_t)15;
static_assert(TX_QS % 8 == 0 && TX_Q % 8 == 0, "alignment of the LDS request block");

struct TCyc {
  uint32_t flags;
  int n_tas, R, slots;
  const int32_t* tas_of_flavor;   // [nF] index into the TAS flavors, -1 = not one
  const TK* tk;                   // [n_tas] topology + per-slot scratch (X); T.tas_usage = the base plane
  int64_t* const* work;           // [n_tas]
  int64_t* const* np;             // [n_tas]
  int64_t* const* priv;           // [n_tas] [slots][n_leaves * R]
  const uint8_t* cq_tas_only;
  const int32_t *adm_off, *adm_tas, *adm_leaf, *adm_count;
  const int64_t* adm_req;
  const uint8_t *ps_flags, *ps_kind;
  const int32_t *ps_level, *ps_slice_size, *ps_slice_level, *ps_group;
  const int64_t* ps_req;
  const int32_t *ps_n_layers, *ps_layer_level, *ps_layer_size;
  int32_t* q_i32;                 // [slots][TQ_WORDS]
  uint8_t* q_u8;                  // [slots][TC_P + 8] kind per request, simulate-empty flag
  int64_t* q_spr;                 // [slots][TC_P][R]
  int32_t *d_leaf, *d_count;      // [slots][2][d_cap]: half 0 = the assignment the wave keeps, half 1 = output of the find in flight
  int d_cap;
  int32_t *h_tas, *h_pos, *h_n;   // [n_ps] the published TopologyAssignment of every podset: TAS flavor (-1 = none), segment of the pool
  int32_t *pool_leaf, *pool_count, *pool_used;
  int pool_cap;
  long long* stats;               // [4] placements, recomputations, unsupported
  int32_t* tree_state;            // [n_tree][12] processEntry's per-tree state while one wave walks all trees in entry order
  // Request classes of k_process_tas: phase 1 of a placement depends on (per-pod requests, slice size / level) only, so the podsets of a
  // cycle fall into a handful of classes (6 at cfg 5). Their phase-1 tables over the WORK plane are kept resident and patched after
  // every AddUsage (an admission touches <= a few dozen leaves and their ancestors), so that a recomputation starts phase 2 directly.
  int ncls;
  const int32_t* ps_class;        // [n_ps] class of the podset's request, -1 = none (podset group, inner layers, table full)
  const int64_t* cls_req;         // [ncls][R]
  const int32_t* cls_ssize;       // [ncls]
  const int32_t* cls_slevel;      // [ncls][n_tas]
  int32_t* const* cls_tab;        // [n_tas] -> [5][ncls][D]: podCount, sliceCount, podCountWithLeader, sliceCountWithLeader, leaderCount
  // The same tables for an EMPTY cluster (WithSimulateEmpty :555: usage ignored): the placement a Preempt-mode assignment reserves with
  // (flavorassigner.go:889-897, scheduler.go:966-975). They depend on the free capacity alone — computed once per cycle, never patched.
  // In a saturated closed loop every recomputed entry runs two such placements, and each paid a full phase 1 over all leaves: 61 of the
  // 233 us of an entry (profiles/r05j_prof_tas_closed.txt). Working copies: slots c.slots + c.ncls + cls.
  int32_t* const* cls_tab_e;
  long long* const* cls_bytes;    // [n_tas] -> [ncls] algorithmic bytes of one phase 1 of the class
  uint8_t* cls_ok;                // [n_tas][ncls] the class's slice parameters are valid on this flavor
  const int32_t* const* par;      // [n_tas] -> [D] parent domain (global id), -1 at level 0
  int32_t* const* cflag;          // [n_tas] -> [ncls][D] owner marks of the incremental update (all zero between two updates)
  // The second pass (kq_cycle_tas.ps_adm_flavor / ps_ex_*): heads that hold an admission. All null when no head does.
  const int32_t* ps_adm_flavor;   // [n_ps][nR] Status.Admission's flavors: Assign keeps them (flavorassigner.go:768-774)
  const uint8_t* sp_kind;         // [n_ps] SP_HAS_EX: the admission holds a TopologyAssignment for the podset; SP_UNHEALTHY: it names an unhealthy node
  // what findReplacementAssignment :686 derives from the admission and the topology alone, done by the host (kq_host.hpp sp_prepare):
  // [n_ps][SP_W] = {status (KQ_TAS_OK, or the failure known up front: KQ_TAS_STALE :695, KQ_TAS_BAD_SLICE_SIZE :701), operand a, operand b,
  // tr.Count = the pods of the deleted domain :693, slice size and slice level after the rewrite :703-722, 1 when rewritten (no inner layers left),
  // [lo, hi) = the leaves below requiredReplacementDomain :759, leaf of the deleted domain (-1 = not a leaf of the snapshot)}
  const int32_t* sp_req;
  // node feasibility rows (kq_cycle_tas.ps_mask / leaf_mask): null = every podset may use every leaf. A masked podset has no request class.
  const int32_t* ps_mask;         // [n_ps][n_tas] row, -1 = every leaf
  const uint8_t* leaf_mask;       // [n_masks][mask_stride]
  int mask_stride;
  int32_t* sp_del_out;            // [n_ps] pods the replacement put back on the deleted domain's leaf: the admission accounts for them
                                  // already (ComputeTASNetUsage flavorassigner.go:131-141), so they are kept out of the wave's usage list;
                                  // -1 = the podset lost the admission's assignment to a failed result
};
constexpr int SP_W = 12, SP_STATUS = 0, SP_OPA = 1, SP_OPB = 2, SP_COUNT = 3, SP_SSIZE = 4, SP_SLEVEL = 5, SP_NLAY = 6, SP_LO = 7, SP_HI = 8, SP_DEL = 9;
constexpr uint8_t SP_HAS_EX = 1, SP_UNHEALTHY = 2;

#ifdef KQ_TAS_CYCLE
// ---- usage planes --------------------------------------------------------------------------------------------------------------------
KQ_DEV int64_t* tc_plane(const TCyc& c, int t, int which, int slot) {
  const TTopo& T = c.tk[t].T;
  if (which == 0) return T.tas_usage;
  if (which == 1) return c.work[t];
  if (which == 2) return c.np[t];
  return c.priv[t] + (size_t)slot * T.n_leaves * T.R;
}
// workload.TASUsage() of an admitted row applied to one plane of every TAS flavor (clusterqueue_snapshot.go:121-134). Lanes = the row's
// domain entries; two entries of a row may name the same leaf (two podsets), hence the atomics — callers fence before reading.
KQ_NOINLINE void tc_row_apply(const TCyc& c, int row, bool add, int which, int slot) {
  for (int e = c.adm_off[row] + lane_id(); e < c.adm_off[row + 1]; e += WAVE) {
    const int t = c.adm_tas[e];
    const TTopo& T = c.tk[t].T;
    int64_t* pl = tc_plane(c, t, which, slot);
    const int64_t cnt = c.adm_count[e];
    for (int r = 0; r < T.R; r++) {
      const int64_t q = c.adm_req[(size_t)e * T.R + r];
      const int64_t v = (q > 0 ? q : 0) * cnt + (r == T.pods ? cnt : 0);
      if (v) atomic_add_i64((long long*)&pl[(size_t)c.adm_leaf[e] * T.R + r], (long long)(add ? v : -v));
    }
  }
  wsync();
}
// TASFlavorSnapshot.Fits :433 for one TopologyDomainRequests on `pl` (agent-scope loads: the cells are updated with L2 atomics)
KQ_DEV bool tc_fits_dom(const TTopo& T, const int64_t* pl, int leaf, int32_t count, const int64_t* spr) {
  if (T.R <= 4) {   // the usual width: every operand (requests, capacity, usage) is in flight before the first division
    int64_t q[4], cap[4], used[4];
    #pragma unroll
    for (int r = 0; r < 4; r++) {
      const bool in = r < T.R;
      q[r] = in ? spr[r] : 0;
      cap[r] = in ? T.free_cap[(size_t)leaf * T.R + r] : 0;
      used[r] = in ? (int64_t)ag_load_u64((const uint64_t*)(pl + (size_t)leaf * T.R + r)) : 0;
    }
    bool have = false; int32_t result = 0;
    #pragma unroll
    for (int r = 0; r < 4; r++) {
      if (q[r] == 0) continue;
      int32_t cc = 0x7fffffff;
      if (q[r] > 0) cc = (int32_t)i64max(0, i64min(t_div(cap[r] - used[r], q[r]), 0x7fffffff));
      if (!have || cc < result) { result = cc; have = true; }
    }
    return (have ? result : 0) >= count;
  }
  bool have = false; int32_t result = 0;
  for (int r = 0; r < T.R; r++) {
    if (spr[r] == 0) continue;
    int32_t cc = 0x7fffffff;
    if (spr[r] > 0) {
      const int64_t used = (int64_t)ag_load_u64((const uint64_t*)(pl + (size_t)leaf * T.R + r));
      cc = (int32_t)i64max(0, i64min(t_div(T.free_cap[(size_t)leaf * T.R + r] - used, spr[r]), 0x7fffffff));
    }
    if (!have || cc < result) { result = cc; have = true; }
  }
  return (have ? result : 0) >= count;
}

// ---- request classes: resident phase-1 tables over the work plane -----------------------------------------------------------------
KQ_DEV TLeafArgs tc_class_args(const TCyc& c, int t, int cls, int32_t* tab) {
  const size_t D = c.tk[t].T.D, n = (size_t)c.ncls * D;
  int32_t* b = tab + (size_t)cls * D;
  return TLeafArgs{b, b + n, b + 2 * n, b + 3 * n, b + 4 * n, nullptr, c.cls_req + (size_t)cls * c.R, nullptr, nullptr, 0, 0,
                   c.cls_slevel[(size_t)cls * c.n_tas + t], c.cls_ssize[cls]};
}
// the working copy of class cls: the per-slot state arrays of slot c.slots + cls (t_workload starts phase 2 from it and puts the class
// table's values back into the domains it consumed)
KQ_DEV TLeafArgs tc_class_work_args(const TCyc& c, int t, int cls) {
  const TK& tk = c.tk[t];
  const size_t o = (size_t)(c.slots + cls) * tk.T.D;
  return TLeafArgs{tk.X.pc + o, tk.X.sc + o, tk.X.pcwl + o, tk.X.scwl + o, tk.X.lc + o, nullptr, c.cls_req + (size_t)cls * c.R, nullptr, nullptr, 0, 0,
                   c.cls_slevel[(size_t)cls * c.n_tas + t], c.cls_ssize[cls]};
}
// phase 1 of class cls on flavor t over the work plane, from scratch (one wave; before k_process_tas walks)
KQ_DEV void tc_class_init(const TCyc& c, int t, int cls) {
  TK tk = c.tk[t];
  tk.T.tas_usage = c.work[t];
  tk.mail = nullptr;
  const int lane = lane_id();
  const TTopo& T = tk.T;
  const TLeafArgs a = tc_class_args(c, t, cls, c.cls_tab[t]);
  const TLeafArgs b = tc_class_work_args(c, t, cls);
  const bool ok = a.sliceSize > 0 && a.sliceLevelIdx >= 0 && a.sliceLevelIdx < T.L;
  if (lane == 0) { c.cls_ok[(size_t)t * c.ncls + cls] = ok ? 1 : 0; c.cls_bytes[t][cls] = 0; }
  if (!ok) return;
  TState s{};
  s.pc = a.pc; s.sc = a.sc; s.pcwl = a.pcwl; s.scwl = a.scwl; s.lc = a.lc;
  TParams st{};
  st.count = 1; st.leaderCount = 0; st.sliceSize = a.sliceSize; st.sliceLevelIdx = a.sliceLevelIdx; st.requestedLevelIdx = 0;
  st.simulateEmpty = false; st.hasAssumed = false; st.req = a.req; st.leaderReq = nullptr; st.leafOk = nullptr;
  t_fill_in_counts(tk, s, st, c.cls_bytes[t] + cls);
  wsync();
  for (int d = lane; d < T.D; d += WAVE) { b.pc[d] = a.pc[d]; b.sc[d] = a.sc[d]; b.pcwl[d] = a.pcwl[d]; b.scwl[d] = a.scwl[d]; b.lc[d] = a.lc[d]; }
  int32_t* meta = tk.X.meta + (size_t)(c.slots + cls) * 4;
  if (lane == 0) { meta[0] = 0; meta[1] = 0; meta[2] = cls; }
  wsync();
  if (c.cls_tab_e) {   // the empty-cluster table and its working copy
    const TLeafArgs ae = tc_class_args(c, t, cls, c.cls_tab_e[t]);
    const size_t o = (size_t)(c.slots + c.ncls + cls) * T.D;
    TState se{};
    se.pc = ae.pc; se.sc = ae.sc; se.pcwl = ae.pcwl; se.scwl = ae.scwl; se.lc = ae.lc;
    st.simulateEmpty = true;
    long long scratch_bytes = 0;   // (a class's phase-1 bytes are charged from cls_bytes: the same whatever the usage)
    t_fill_in_counts(tk, se, st, &scratch_bytes);
    wsync();
    for (int d = lane; d < T.D; d += WAVE) { tk.X.pc[o + d] = ae.pc[d]; tk.X.sc[o + d] = ae.sc[d]; tk.X.pcwl[o + d] = ae.pcwl[d]; tk.X.scwl[o + d] = ae.scwl[d]; tk.X.lc[o + d] = ae.lc[d]; }
    int32_t* me = tk.X.meta + (size_t)(c.slots + c.ncls + cls) * 4;
    if (lane == 0) { me[0] = 0; me[1] = 0; me[2] = cls; }
    wsync();
  }
}
// AddUsage changed the work plane on the leaves of entry e's TopologyAssignments: every class table (and its working copy) follows.
// A class carries no leader and no inner layers, so in its table podCountWithLeader == podCount, sliceCountWithLeader == sliceCount,
// leaderCount == 0 on every level, an upper domain's podCount is the SUM of its children's (fillInCountsHelper :1930 with minPodDiff 0),
// and its sliceCount is podCount / sliceSize on the slice level, the sum of the children's above it, 0 below it. Hence deltas:
// lanes = (class, touched leaf): the leaf's counts are recomputed from the plane (CountIn), the difference of its podCount is added to
// every ancestor; then the slice-level ancestors (one owner lane each) turn their new podCount into a sliceCount and send that
// difference up. Bit-identical to running phase 1 again.
KQ_DEV void tc_class_update(const K& k, int ps_base, int nps, bool lds_on, int lds_bytes) {
  const TCyc& c = *k.tc;
  if (c.ncls == 0) return;
  const int lane = lane_id();
  for (int p = 0; p < nps; p++) {
    const int g = ps_base + p, t = c.h_tas[g];
    if (t < 0 || c.h_n[g] == 0) continue;
    TTopo T = c.tk[t].T;
    T.tas_usage = c.work[t];
    const int32_t* par = c.par[t]; int32_t* flag = c.cflag[t];
    const int n = c.h_n[g], pos = c.h_pos[g], items = n * c.ncls;
    // (a placement that keeps its working state in LDS fills it from the table every time: no working copies to maintain)
    const bool copies = !(lds_on && TX_BYTES + tas_lds_layout(T.D, c.tk[t].X.max_set).total <= (size_t)lds_bytes);
    // A: leaves, podCount deltas up the tree (and the sliceCount deltas when the leaves are the slice level)
    for (int i = lane; i < items; i += WAVE) {
      const int cls = i / n, j = i % n;
      if (!c.cls_ok[(size_t)t * c.ncls + cls]) continue;
      const TLeafArgs a = tc_class_args(c, t, cls, c.cls_tab[t]);
      const int leaf = c.pool_leaf[pos + j], d = T.leaf_base + leaf;
      const int32_t old_pc = a.pc[d], old_sc = a.sc[d];
      t_leaf_counts_any(T, a, leaf, T.n_leaves);   // exactly this leaf
      const int32_t dpc = a.pc[d] - old_pc, dsc = a.sc[d] - old_sc;
      for (int x = par[d]; x >= 0; x = par[x]) {
        if (dpc) { atomic_add_i32(&a.pc[x], dpc); atomic_add_i32(&a.pcwl[x], dpc); }
        if (dsc) { atomic_add_i32(&a.sc[x], dsc); atomic_add_i32(&a.scwl[x], dsc); }
      }
    }
    wsync();
    // B: slice level above the leaves
    for (int i = lane; i < items; i += WAVE) {
      const int cls = i / n, j = i % n;
      if (!c.cls_ok[(size_t)t * c.ncls + cls]) continue;
      const TLeafArgs a = tc_class_args(c, t, cls, c.cls_tab[t]);
      const int sl = a.sliceLevelIdx;
      if (sl >= T.L - 1) continue;
      int x = T.leaf_base + c.pool_leaf[pos + j];
      for (int l = T.L - 1; l > sl; l--) x = par[x];
      if (atomic_add_i32(&flag[(size_t)cls * T.D + x], 1) != 0) continue;   // another lane owns this (class, domain)
      const int32_t npc = (int32_t)ag_load_u32((const uint32_t*)&a.pc[x]);
      const int32_t nsc = npc / a.sliceSize, dsc = nsc - a.sc[x];
      a.sc[x] = nsc; a.scwl[x] = nsc;
      if (dsc) for (int y = par[x]; y >= 0; y = par[y]) { atomic_add_i32(&a.sc[y], dsc); atomic_add_i32(&a.scwl[y], dsc); }
    }
    wsync();
    // C: the touched chains into the working copies (and rewritten in place: what the atomics left in L2 becomes what plain loads see)
    for (int i = lane; i < items; i += WAVE) {
      const int cls = i / n, j = i % n;
      if (!c.cls_ok[(size_t)t * c.ncls + cls]) continue;
      const TLeafArgs a = tc_class_args(c, t, cls, c.cls_tab[t]);
      const TLeafArgs b = tc_class_work_args(c, t, cls);
      int lvl = T.L - 1;
      for (int x = T.leaf_base + c.pool_leaf[pos + j]; x >= 0; x = par[x], lvl--) {
        const int32_t v0 = (int32_t)ag_load_u32((const uint32_t*)&a.pc[x]), v1 = (int32_t)ag_load_u32((const uint32_t*)&a.sc[x]);
        a.pc[x] = v0; a.pcwl[x] = v0; a.sc[x] = v1; a.scwl[x] = v1;
        if (copies) { b.pc[x] = v0; b.pcwl[x] = v0; b.sc[x] = v1; b.scwl[x] = v1; b.lc[x] = 0; }
        if (lvl == a.sliceLevelIdx) flag[(size_t)cls * T.D + x] = 0;
      }
    }
    wsync();
  }
}

KQ_DEV int tc_flavor_of(const TCyc& c, int nF, int t) {  // the ResourceFlavor of TAS flavor t (inverse of tas_of_flavor)
  for (int f = 0; f < nF; f++) if (c.tas_of_flavor[f] == t) return f;
  return -1;
}

// ---- Assign's TAS step -------------------------------------------------------------------------------------------------------------------
KQ_DEV int tc_adm_flavor(const K& k, int psg, int res) {
  const TCyc& c = *k.tc;
  return c.ps_adm_flavor ? c.ps_adm_flavor[(size_t)psg * k.S.nR + res] : -1;
}
KQ_DEV bool tc_is_tas_flavor(const K& k, int flavor) { return k.tc->tas_of_flavor[flavor] >= 0; }   // a key of cq.TASFlavors (tasFlavorsOnly, flavorassigner.go:996)
// a fresh Assign starts from the admission's TopologyAssignment again (flavorassigner.go:777-779)
KQ_DEV void tc_sp_reset(const K& k, int psg) { if (k.tc->sp_del_out && lane_id() == 0) k.tc->sp_del_out[psg] = 0; }
KQ_DEV void tc_reset(Wave& w) {
  if (lane_id() == 0) { w.ta.t = -1; w.ta.nreq = 0; w.ta.af_early = 0; w.ta.err_mask = 0; w.ta.has_mask = 0; w.ta.kept_used = 0; w.ta.srch = 0; w.ta.em_ps = -1; w.ta.req_valid = 0; }
}
// Assignment.updateMode flavorassigner.go:192-198 for podset p; the usage entries' modes (flavorResourcesNeedPreemption reads them)
// are re-derived from the cells: an entry is as weak as the weakest (podset, resource) behind it
KQ_NOINLINE void tc_update_mode(const K& k, Wave& w, int p, int mode) {
  const int nR = k.S.nR;
  const size_t o = (size_t)(w.ps_base + p) * nR;
  for (int r = lane_id(); r < nR; r += WAVE) if (k.O.flavor[o + r] >= 0) k.O.res_mode[o + r] = (uint8_t)mode;
  // ResourceAssignment holds *FlavorAssignment, and resolvePodSetFlavors (:917, FilterKeys) hands every member of a PodSetGroupName group the
  // SAME pointers out of groupFlavors: the mode written through p's entry is the mode of the other members' entries for that resource
  if (const int32_t* grp = k.H.ps_group) {
    const int gid = grp[w.ps_base + p];
    for (int q = 0; gid >= 0 && q < w.nps; q++) {
      if (q == p || grp[w.ps_base + q] != gid) continue;
      const size_t oq = (size_t)(w.ps_base + q) * nR;
      for (int r = lane_id(); r < nR; r += WAVE) if (k.O.flavor[o + r] >= 0 && k.O.flavor[oq + r] >= 0) k.O.res_mode[oq + r] = (uint8_t)mode;
    }
  }
  wsync();
  // (lanes = the head's (podset, resource) cells, one load each, a wave minimum per usage entry: this was nuse x nps x nR dependent loads
  // on lane 0, twice per recomputed entry of a saturated cycle)
  const int cells = w.nps * nR;
  if (cells <= WAVE) {
    const int i = lane_id();
    int fr = -1, rm = M_FIT;
    if (i < cells) {
      const size_t c = (size_t)w.ps_base * nR + i;
      const int fl = k.O.flavor[c];
      if (fl >= 0) { fr = fl * nR + (i % nR); rm = k.O.res_mode[c]; }
    }
    for (int u = 0; u < w.nuse; u++) {
      const int m = (int)wmin_u64((uint64_t)(fr == w.use_fr[u] ? rm : M_FIT));
      if (lane_id() == 0) w.use_mode[u] = (uint8_t)m;
    }
  } else {
    for (int u = 0; u < w.nuse; u++) {
      int m = M_FIT;
      for (int i = lane_id(); i < cells; i += WAVE) {
        const size_t c = (size_t)w.ps_base * nR + i;
        const int fl = k.O.flavor[c];
        if (fl >= 0 && fl * nR + (i % nR) == w.use_fr[u] && k.O.res_mode[c] < m) m = k.O.res_mode[c];
      }
      m = (int)wmin_u64((uint64_t)m);
      if (lane_id() == 0) w.use_mode[u] = (uint8_t)m;
    }
  }
  if (lane_id() == 0) w.rep_mode = mode;
  wsync();
}
// tas_flavorassigner.go:37-83 WorkloadsTopologyRequests (+ onlyTASFlavor :142) on the head's current flavor assignment
KQ_NOINLINE void tc_requests(const K& k, Wave& w) {
  const TCyc& c = *k.tc;
  // (Assign, GetTargets and updateAssignmentForTAS each ask — three times per recomputed entry; the answer only moves with the flavor
  // assignment (tc_reset) and with the podsets that hold a TopologyAssignment (tc_keep_result))
  if (w.ta.req_valid) return;
  if (lane_id() == 0) {
    w.ta.req_valid = 1;
    const int nR = k.S.nR;
    w.ta.t = -1; w.ta.nreq = 0;
    bool two_flavors = false;
    for (int p = 0; p < w.nps && p < TC_P; p++) {
      const int g = w.ps_base + p;
      const bool explicitReq = (c.ps_flags[g] & KQ_PS_TAS_EXPLICIT) != 0;
      if (!explicitReq && !c.cq_tas_only[w.cq]) continue;                       // isTASRequested :231
      if (w.ta.af_early && p >= w.ta.af_early) continue;                        // behind the podset Assign stopped at: not in assignment.PodSets
      if ((w.ta.err_mask >> p) & 1) continue;
      if (k.O.ps_count[g] == 0) continue;
      if ((w.ta.has_mask >> p) & 1) continue;
      if (c.sp_kind) {
        // second pass: a podset whose admission holds a TopologyAssignment is placed again only to replace a failed node (:50); in the
        // replacement branch a podset without one gets no result at all (findPSA tas_flavor_snapshot.go:612)
        const uint8_t sk = c.sp_kind[g];
        if ((sk & SP_HAS_EX) && !((w.hflags & KQ_HEAD_HAS_UNHEALTHY_NODES) && (sk & SP_UNHEALTHY))) continue;
      }
      int first = -1; bool many = false;
      for (int r = 0; r < nR; r++) {
        const int fl = k.O.flavor[(size_t)g * nR + r];
        const int t = fl >= 0 ? c.tas_of_flavor[fl] : -1;
        if (t < 0) continue;
        if (first < 0) first = t; else if (t != first) many = true;
      }
      if (first < 0 || many) { w.ta.err_mask |= 1u << p; w.rep_mode = M_NOFIT; continue; }  // psError :290 -> RepresentativeMode NoFit
      if (w.ta.t >= 0 && w.ta.t != first) { two_flavors = true; continue; }
      w.ta.t = first;
      if (c.sp_kind && (w.hflags & KQ_HEAD_HAS_UNHEALTHY_NODES) && !(c.sp_kind[g] & SP_HAS_EX)) continue;   // requested, but the replacement branch has no result for it
      w.ta.req_ps[w.ta.nreq++] = (uint8_t)p;
    }
    // the podsets of one workload on two TAS flavors: the placement (one TAS flavor per wave) does not take them. With a psError on some podset
    // the assignment is NoFit and no placement is ever asked for (flavorassigner.go:866, :879; scheduler.go:941): nothing to refuse then
    if (two_flavors && !w.ta.err_mask) { if (*k.O.error == 0) *k.O.error = KQ_EUNSUPPORTED; if (c.stats) c.stats[2] = 1; }
  }
  wsync();
}
struct TcFail { bool failed; int ps, status; int32_t a, b; };
// ClusterQueueSnapshot.FindTopologyAssignmentsForWorkload clusterqueue_snapshot.go:204-237 for the requests in w.ta on plane `which`;
// the domains land in half 1 of the slot's store. Failure = TASAssignmentsResult.Failure :411 (first failing podset).
// the wave's domain stores: half 0 = the assignment it keeps, half 1 = output of the find in flight
KQ_DEV void tc_dstore(const TCyc& c, const Wave& w, int slot, int half, int32_t** leaf, int32_t** count, int* cap) {
  if (w.ta.d_lds) { *leaf = (int32_t*)(w.ta.lds + TX_DL) + half * TX_DCAP; *count = (int32_t*)(w.ta.lds + TX_DC) + half * TX_DCAP; *cap = TX_DCAP; }
  else { *leaf = c.d_leaf + ((size_t)slot * 2 + half) * c.d_cap; *count = c.d_count + ((size_t)slot * 2 + half) * c.d_cap; *cap = c.d_cap; }
}
// the request / result block of the wave's last find
KQ_DEV int32_t* tc_qblock(const TCyc& c, const Wave& w, int slot, TQOff* qo) {
  if (w.ta.q_lds) { *qo = tq_one(); return (int32_t*)(w.ta.lds + TX_Q); }
  *qo = tq_full();
  return c.q_i32 + (size_t)slot * TQ_WORDS;
}
KQ_NOINLINE TcFail tc_find(const K& k, Wave& w, int slot, bool simulateEmpty, int which) {
  const TCyc& c = *k.tc;
  TcFail f{false, -1, 0, 0, 0};
  const int n = w.ta.nreq;
  if (n == 0) return f;
  KQ_T0();
  const int t = w.ta.t;
  const int lane = lane_id();
  if (lane == 0) w.ta.em_ps = -1;   // (whatever the block and the store half held is about to be overwritten)
  // processEntry on the work plane, one podset of a request class: start from the class's resident phase 1, and (k_process_tas) keep the
  // request block and the placement's working state in LDS
  // (k_process_tas: the static half of the block came with the entry's prefetched header when it has one podset on the one TAS flavor)
  // second pass, a failed node to replace (tas_flavor_snapshot.go:608-633): every podset on its own through findReplacementAssignment :686
  // — the request the host rewrote (TCyc::sp_req), never on an empty cluster (:723), no class table, no prefetched block
  const bool repl = c.sp_kind != nullptr && (w.hflags & KQ_HEAD_HAS_UNHEALTHY_NODES) != 0;
  if (repl) simulateEmpty = false;
  int n_place = n;   // the requests in front of the first one whose failure is known up front (:694-702)
  if (repl) for (int i = n - 1; i >= 0; i--) if (c.sp_req[(size_t)(w.ps_base + w.ta.req_ps[i]) * SP_W + SP_STATUS] != KQ_TAS_OK) n_place = i;
  const TPre* qp = (!repl && w.ta.cur_pre && w.ta.cur_pre->q_ok && n == 1 && w.ta.req_ps[0] == 0 && t == 0) ? w.ta.cur_pre : nullptr;
  int cls = -1;
  const bool empty_tab = simulateEmpty && c.cls_tab_e != nullptr;
  if (!repl && which == 1 && (!simulateEmpty || empty_tab) && n == 1 && c.ncls > 0) {
    if (qp) cls = qp->q_cls_ok ? qp->q_cls : -1;
    else {
      cls = c.ps_class[w.ps_base + w.ta.req_ps[0]];
      if (cls >= 0 && !c.cls_ok[(size_t)t * c.ncls + cls]) cls = -1;
    }
  }
  const TK& tk0 = c.tk[t];
  const bool st_lds = cls >= 0 && w.ta.lds && TX_BYTES + tas_lds_layout(tk0.T.D, tk0.X.max_set).total <= (size_t)w.ta.lds_bytes;
  wsync();
  if (lane == 0) w.ta.q_lds = st_lds ? 1 : 0;
  wsync();
  TQOff qo;
  int32_t* qi = tc_qblock(c, w, slot, &qo);
  const bool masks = c.ps_mask != nullptr && !st_lds;   // (a masked podset has no class: its block is the slot's global one)
  uint8_t* qu = st_lds ? (uint8_t*)(w.ta.lds + TX_QU) : c.q_u8 + (size_t)slot * (TC_P + 8);
  int64_t* qs = st_lds ? (int64_t*)(w.ta.lds + TX_QS) : c.q_spr + (size_t)slot * TC_P * c.R;
  uint8_t* qsim = st_lds ? qu + 8 : qu + TC_P;
  const bool layered = c.ps_n_layers != nullptr;
  if (lane == 0) {
    qi[TQ_WLOFF] = 0; qi[TQ_WLOFF + 1] = n_place;
    if (qp) {
      qi[qo.count] = k.O.ps_count[w.ps_base];
      qi[qo.level] = qp->q_level; qi[qo.ssize] = qp->q_ssize; qi[qo.slevel] = qp->q_slevel; qi[qo.group] = qp->q_group;
      qu[0] = (uint8_t)qp->q_kind;
      for (int r = 0; r < c.R; r++) qs[r] = qp->q_req[r];
    } else
    for (int i = 0; i < n; i++) {
      const int g = w.ps_base + w.ta.req_ps[i];
      qi[qo.count + i] = k.O.ps_count[g];
      qi[qo.level + i] = c.ps_level[(size_t)g * c.n_tas + t];
      qi[qo.ssize + i] = c.ps_slice_size[g];
      qi[qo.slevel + i] = c.ps_slice_level[(size_t)g * c.n_tas + t];
      qi[qo.group + i] = c.ps_group[g];
      qu[i] = c.ps_kind[g];
      for (int r = 0; r < c.R; r++) qs[(size_t)i * c.R + r] = c.ps_req[(size_t)g * c.R + r];
      if (layered) {
        qi[qo.nlay + i] = c.ps_n_layers[g];
        for (int j = 0; j < TC_ML; j++) {
          qi[qo.llevel + i * TC_ML + j] = c.ps_layer_level[((size_t)g * c.n_tas + t) * TC_ML + j];
          qi[qo.lsize + i * TC_ML + j] = c.ps_layer_size[(size_t)g * TC_ML + j];
        }
      }
      if (repl) {
        const int32_t* sp = c.sp_req + (size_t)g * SP_W;
        qi[qo.count + i] = sp[SP_COUNT]; qi[qo.ssize + i] = sp[SP_SSIZE]; qi[qo.slevel + i] = sp[SP_SLEVEL];
        qi[qo.group + i] = -1;   // (:609: the podsets of a group one by one, none of them a leader)
        if (layered && sp[SP_NLAY]) qi[qo.nlay + i] = 0;
        qi[TQ_LO + i] = sp[SP_LO]; qi[TQ_HI + i] = sp[SP_HI];
      }
    }
    if (masks) for (int i = 0; i < n; i++) qi[TQ_MASK + i] = c.ps_mask[(size_t)(w.ps_base + w.ta.req_ps[i]) * c.n_tas + t];
    *qsim = simulateEmpty ? 1 : 0;
#ifdef KQ_HOST_EMU
    for (int i = 0; i < n; i++) { qi[qo.status + i] = 0x5a5a5a5a; qi[qo.dn + i] = 0x5a5a5a5a; qi[qo.dpos + i] = 0x5a5a5a5a; }   // (the emulation: nothing may be read from the block of the find before)
#endif
    for (int i = 0; i < 24; i++) qi[qo.misc + i] = 0;
    if (c.stats) atomic_add_i64(c.stats, 1);
  }
  wsync();
  TK tk = c.tk[t];
  tk.T.tas_usage = tc_plane(c, t, which, slot);
  tk.Q.n_wl = 1; tk.Q.wl_off = qi + TQ_WLOFF; tk.Q.sim_empty = qsim; tk.Q.spr = qs;
  tk.Q.count = qi + qo.count; tk.Q.level = qi + qo.level; tk.Q.kind = qu; tk.Q.slice_size = qi + qo.ssize; tk.Q.slice_level = qi + qo.slevel;
  tk.Q.group = qi + qo.group;
  tk.Q.leaf_ok = masks ? c.leaf_mask : nullptr; tk.Q.leaf_ok_idx = masks ? qi + TQ_MASK : nullptr; tk.Q.leaf_ok_stride = c.mask_stride;
  tk.Q.leaf_lo = repl ? qi + TQ_LO : nullptr; tk.Q.leaf_hi = repl ? qi + TQ_HI : nullptr;
  tk.Q.n_layers = layered ? qi + qo.nlay : nullptr; tk.Q.layer_level = layered ? qi + qo.llevel : nullptr; tk.Q.layer_size = layered ? qi + qo.lsize : nullptr;
  tk.O.status = qi + qo.status; tk.O.op_a = qi + qo.opa; tk.O.op_b = qi + qo.opb; tk.O.dom_pos = qi + qo.dpos; tk.O.dom_n = qi + qo.dn;
  tk.O.layer_fit = nullptr;
  tc_dstore(c, w, slot, 1, &tk.O.pool_leaf, &tk.O.pool_count, &tk.O.pool_cap);
  tk.O.pool_used = qi + qo.misc; tk.O.error = qi + qo.misc + 1; tk.O.bytes = (long long*)(qi + qo.misc + 2);
  tk.C.n = 0;
  tk.mail = w.ta.mail;
  tk.lds = nullptr;
  int xslot = slot;
  if (cls >= 0) {
    const int g = w.ps_base + w.ta.req_ps[0];
    const size_t nD = (size_t)c.ncls * tk.T.D;
    int32_t* tab = simulateEmpty ? c.cls_tab_e[t] : c.cls_tab[t];
    tk.C.n = c.ncls; tk.C.wl_class = c.ps_class + g; tk.C.order = nullptr; tk.C.leader = nullptr; tk.C.workers = nullptr;   // (leader == nullptr: t_workload keeps all five count arrays of a global-row state)
    tk.C.pc = tab; tk.C.sc = tab + nD; tk.C.pcwl = tab + 2 * nD; tk.C.scwl = tab + 3 * nD; tk.C.lc = tab + 4 * nD;
    tk.C.bytes = c.cls_bytes[t];
    xslot = c.slots + (simulateEmpty ? c.ncls : 0) + cls;
    if (st_lds) tk.lds = w.ta.lds + TX_BYTES;   // the working state in LDS
    if (lane == 0 && c.stats) atomic_add_i64(c.stats + 3, 1);
  }
  if (which != 0) KQ_TS(k, 47);   // request block + argument block of the placement
  if (n_place > 0) t_workload(tk, xslot, 0);
  wsync();
#ifndef KQ_TAS_NO_PREFETCH
  if (w.ta.mail && w.ta.pf_pos >= 0) {
    // nothing is posted to the helper waves for the rest of this entry: wave 1 fetches the next entry's header meanwhile
    TLeafJob& j = *w.ta.mail;
    t_post_begin(j);
    if (lane == 0) { j.pf_next = w.ta.pf_pos; j.cmd = 6; w.ta.pf_pos = -1; }
    bsync();
#ifdef KQ_HOST_EMU
    t_prefetch_entry(j);
#endif
  }
#endif
  if (which != 0) KQ_TS(k, 45);   // (timing builds, processEntry only) the placement; 46 = its phase 1
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  if (lane == 0 && which != 0) {
    atomic_add_i64((long long*)k.prof + 46, *(long long*)(qi + qo.misc + 4));
    for (int j = 0; j < 8; j++) atomic_add_i64((long long*)k.prof + 51 + j, *(long long*)(qi + qo.misc + 6 + 2 * j));   // TPROF segments of the placement
  }
#endif
  // (the placement's own algorithmic bytes, qi[misc + 2], are not added to the cycle's counter: SURVEY 8d's accounting of the quota
  // cycle does not include them, and neither does the oracle's)
  if (lane == 0 && qi[qo.misc + 1] != 0 && *k.O.error == 0) *k.O.error = qi[qo.misc + 1];
  wsync();
  for (int i = 0; i < n_place; i++) {
    const int st = qi[qo.status + i];
    if (st != KQ_TAS_OK && st != KQ_TAS_SKIPPED) { f.failed = true; f.ps = w.ta.req_ps[i]; f.status = st; f.a = qi[qo.opa + i]; f.b = qi[qo.opb + i]; break; }
    if (repl && st == KQ_TAS_OK && qi[qo.dn + i] == 0) { f.failed = true; f.ps = w.ta.req_ps[i]; f.status = KQ_TAS_NO_REPLACEMENT; break; }   // :727
  }
  if (!f.failed && n_place < n) {
    const int32_t* sp = c.sp_req + (size_t)(w.ps_base + w.ta.req_ps[n_place]) * SP_W;
    f.failed = true; f.ps = w.ta.req_ps[n_place]; f.status = sp[SP_STATUS]; f.a = sp[SP_OPA]; f.b = sp[SP_OPB];
  }
  if (repl && f.failed) {
    // tc_keep_result reads the status words (updateAssignmentForTAS keeps the result of a failed find too: UpdateForTASResult takes the
    // assignment away): the failures found here, and the requests behind the failing one, are written where the placement's are
    wsync();
    if (lane == 0) {
      bool behind = false;
      for (int i = 0; i < n; i++) {
        if (behind) { qi[qo.status + i] = KQ_TAS_SKIPPED; qi[qo.dn + i] = 0; }
        else if ((int)w.ta.req_ps[i] == f.ps) { qi[qo.status + i] = f.status; qi[qo.opa + i] = f.a; qi[qo.opb + i] = f.b; qi[qo.dn + i] = 0; behind = true; }
      }
    }
    wsync();
  }
  if (lane == 0 && simulateEmpty && n == 1 && !f.failed && qi[qo.status] == KQ_TAS_OK) { w.ta.em_ps = w.ta.req_ps[0]; w.ta.em_t = t; w.ta.em_count = k.O.ps_count[w.ps_base + w.ta.req_ps[0]]; }
  wsync();
  return f;
}
// Assignment.UpdateForTASResult flavorassigner.go:87-96 with the result of the last tc_find: a podset whose placement succeeded keeps its
// domains (copied into half 0 of the store), the others lose theirs
KQ_NOINLINE void tc_keep_result(const K& k, Wave& w, int slot) {
  const TCyc& c = *k.tc;
  TQOff qo;
  const int32_t* qi = tc_qblock(c, w, slot, &qo);
  int32_t *fl, *fc, *kl, *kc; int dcap;
  tc_dstore(c, w, slot, 1, &fl, &fc, &dcap);
  tc_dstore(c, w, slot, 0, &kl, &kc, &dcap);
  for (int i = 0; i < w.ta.nreq; i++) {
    const int p = w.ta.req_ps[i];
    const bool ok = qi[qo.status + i] == KQ_TAS_OK;
    const int pos = qi[qo.dpos + i], n = ok ? qi[qo.dn + i] : 0;
    const int at = w.ta.kept_used;
    if (ok && at + n > dcap) { set_error(k, KQ_ECAPACITY); return; }
    int kept = n;
    if (c.sp_kind && (w.hflags & KQ_HEAD_HAS_UNHEALTHY_NODES)) {
      // a replacement: the wave keeps the entry's net usage (ComputeTASNetUsage flavorassigner.go:106-155) — the replacement's pods,
      // except those that went back onto the deleted domain's leaf, which the admission accounts for already (at most as many as it
      // held there). The host merges the admission's other domains in for the caller (mergeTopologyAssignments :2072).
      const int g = w.ps_base + p;
      const int del = c.sp_req[(size_t)g * SP_W + SP_DEL];
      int o = 0; int32_t back = 0;   // (every lane walks the few domains; lane 0 writes: Wave::counts is Assign's under partial admission)
      for (int j = 0; j < n; j++) {
        const int32_t lf = fl[pos + j], cn = fc[pos + j];
        if (del >= 0 && lf == del) { back += cn; continue; }
        if (lane_id() == 0) { kl[at + o] = lf; kc[at + o] = cn; }
        o++;
      }
      if (lane_id() == 0) c.sp_del_out[g] = ok ? back : -1;   // -1: UpdateForTASResult took the admission's assignment away (a failed result, flavorassigner.go:90)
      kept = o;
      wsync();
    } else {
      for (int j = lane_id(); j < n; j += WAVE) { kl[at + j] = fl[pos + j]; kc[at + j] = fc[pos + j]; }
      wsync();
    }
    if (lane_id() == 0) {
      if (ok) { w.ta.has_mask |= 1u << p; w.ta.pos[p] = at; w.ta.n[p] = kept; w.ta.kept_used = at + kept; }
      else w.ta.has_mask &= ~(1u << p);
      w.ta.req_valid = 0;
    }
    wsync();
  }
}
// flavorassigner.go:864-903
KQ_DEV void tc_assign_tas(const K& k, Wave& w, int slot) {
  KQ_T0();
  tc_requests(k, w);
  if (w.ta.plane != 0) KQ_TS(k, 48);   // (timing builds, processEntry only) WorkloadsTopologyRequests
  if (w.rep_mode == M_FIT) {
    const TcFail f = tc_find(k, w, slot, false, w.ta.plane);
    if (w.ta.plane != 0) KQ_TS(k, 49); // the find (47 + 45 inside it)
    if (f.failed) {
      // psAssignment.reason(failure.Reason) :875: the operands of the message (KQ_RSN_TAS_FAILURE); a Fit assignment holds no other reason
      if (lane_id() == 0) {
        rsn_push(k, w, KQ_RSN_TAS_FAILURE, f.ps, tc_flavor_of(*k.tc, k.S.nF, w.ta.t), -1, f.status, f.a, f.b);
        if (k.O.rsn_win > 0) k.O.rsn_n[w.h] = w.rsn_over ? -w.nrsn : w.nrsn;
      }
      wsync();
      tc_update_mode(k, w, f.ps, M_PREEMPT);
    } else tc_keep_result(k, w, slot);
    if (w.ta.plane != 0) KQ_TS(k, 50); // keep the result
  }
  if (w.rep_mode == M_PREEMPT && !(w.hflags & KQ_HEAD_HAS_UNHEALTHY_NODES)) {   // :879 "Don't preempt other workloads if looking for a failed node replacement"
    const TcFail f = tc_find(k, w, slot, true, w.ta.plane);
    if (f.failed) tc_update_mode(k, w, f.ps, M_NOFIT);
    else for (int i = 0; i < w.ta.nreq; i++) tc_update_mode(k, w, w.ta.req_ps[i], M_PREEMPT);  // updateModeForTASRequests :200
  }
}

// ---- GetTargets with TAS requests (preemption.go:135-138): the walk runs on a private copy of the leaf usage ---------------------
KQ_NOINLINE void tc_copy_plane(const TCyc& c, int t, int from, int slot) {
  const TTopo& T = c.tk[t].T;
  const int64_t* src = tc_plane(c, t, from, slot);
  int64_t* dst = tc_plane(c, t, 3, slot);
  for (int i = lane_id(); i < T.n_leaves * T.R; i += WAVE) dst[i] = (int64_t)ag_load_u64((const uint64_t*)(src + i));
  wsync();
}
// The private copy is made when the walk first touches it (srch == 2: asked for, not made yet): a search that finds no candidate —
// every Preempt-mode entry of a ClusterQueue without a preemption policy, i.e. all of the closed loop of cfg5-cycle — never does, and
// the copy (131 KB read + written by ONE wave at cfg 5) was most of what such an entry's GetTargets cost inside processEntry.
KQ_DEV void tc_search_begin(const K& k, Wave& w, int slot) {
  tc_requests(k, w);
  if (w.ta.nreq == 0) return;
  if (lane_id() == 0) w.ta.srch = 2;
  wsync();
}
KQ_DEV void tc_search_copy(const K& k, Wave& w, int slot) {
  if (w.ta.srch != 2) return;
  for (int t = 0; t < k.tc->n_tas; t++) tc_copy_plane(*k.tc, t, w.ta.plane, slot);
  if (lane_id() == 0) w.ta.srch = 1;
  wsync();
}
KQ_DEV void tc_search_end(Wave& w) { if (lane_id() == 0) w.ta.srch = 0; wsync(); }
KQ_DEV void tc_search_row(const K& k, Wave& w, int slot, int row, bool add) { if (w.ta.srch) { tc_search_copy(k, w, slot); tc_row_apply(*k.tc, row, add, 3, slot); } }
KQ_DEV bool tc_search_fits(const K& k, Wave& w, int slot) {  // preemption.go:676-684
  if (!w.ta.srch) return true;
  tc_search_copy(k, w, slot);
  return !tc_find(k, w, slot, false, 3).failed;
}

// ---- scheduler.go:941-985 updateAssignmentForTAS ------------------------------------------------------------------------------------
KQ_DEV void tc_update_assignment(const K& k, Wave& w, int slot, const int32_t* trow, int nt) {
  const TCyc& c = *k.tc;
  if (w.rep_mode != M_PREEMPT) return;
  if (w.hflags & KQ_HEAD_UNHEALTHY_ASSIGNMENT) return;   // scheduler.go:952 !HasTopologyAssignmentWithUnhealthyNode
  bool any = c.cq_tas_only[w.cq] != 0;
  for (int p = 0; p < w.nps && !any; p++) if (c.ps_flags[w.ps_base + p] & KQ_PS_TAS_EXPLICIT) any = true;
  if (!any) return;
  tc_requests(k, w);
  if (nt > 0) {
    if (w.ta.nreq > 0) {
      for (int t = 0; t < c.n_tas; t++) tc_copy_plane(c, t, w.ta.plane, slot);
      for (int i = 0; i < nt; i++) tc_row_apply(c, trow[i], false, 3, slot);   // SimulateWorkloadUsageRemoval
    }
    tc_find(k, w, slot, false, 3);
  } else if (w.ta.nreq == 1 && w.ta.em_ps == (int)w.ta.req_ps[0] && w.ta.em_t == w.ta.t && w.ta.em_count == k.O.ps_count[w.ps_base + w.ta.req_ps[0]]) {
    // the simulate-empty placement of this very request is still in the request block and the store half (see TAW::em_ps): the call the
    // reference makes here returns it again; it is counted, not computed
    if (lane_id() == 0 && c.stats) atomic_add_i64(c.stats, 1);
    wsync();
  } else {
    tc_find(k, w, slot, true, w.ta.plane);
  }
  if (w.ta.nreq > 0) tc_keep_result(k, w, slot);
}

// the head's TopologyAssignments for processEntry and the host
KQ_NOINLINE void tc_publish(const K& k, Wave& w, int slot) {
  const TCyc& c = *k.tc;
  int32_t *kl, *kc; int dcap;
  tc_dstore(c, w, slot, 0, &kl, &kc, &dcap);
  int total = 0;
  for (int p = 0; p < w.nps && p < TC_P; p++) if ((w.ta.has_mask >> p) & 1) total += w.ta.n[p];
  int base = 0;
  if (lane_id() == 0) {
    if (w.ta.pool_own) { base = total > 0 ? w.ta.pool_next : 0; w.ta.pool_next += total; *c.pool_used = w.ta.pool_next; }   // (k_process_tas: the only writer)
    else base = total > 0 ? atomic_add_i32(c.pool_used, total) : 0;
    if (base + total > c.pool_cap) { if (*k.O.error == 0) *k.O.error = KQ_ECAPACITY; base = -1; }
    w.counts[0] = base;
    w.ta.pub_lds = (w.ta.d_lds && base >= 0) ? 1 : 0;   // the published domains are still in store half 0 (tc_entry_fits / tc_entry_add)
  }
  // k_process_tas (the pool is the wave's own): only LDS words are exchanged here, and a full fence would wait for the acknowledgement
  // of the global stores above — a round trip each
  const bool own = w.ta.pool_own != 0;
  if (own) wsync_lds(); else wsync();
  base = w.counts[0];
  const bool in_lds = w.ta.pub_lds != 0;
  if (own) wsync_lds(); else wsync();
  int at = base;
  for (int p = 0; p < w.nps && p < TC_P; p++) {
    const int g = w.ps_base + p;
    const bool has = base >= 0 && ((w.ta.has_mask >> p) & 1);
    if (lane_id() == 0) { c.h_tas[g] = has ? w.ta.t : -1; c.h_pos[g] = has ? at : 0; c.h_n[g] = has ? w.ta.n[p] : 0; }
    if (has) {
      for (int j = lane_id(); j < w.ta.n[p]; j += WAVE) { c.pool_leaf[at + j] = kl[w.ta.pos[p] + j]; c.pool_count[at + j] = kc[w.ta.pos[p] + j]; }
      at += w.ta.n[p];
    }
  }
  // what was just written to global memory (h_*, the pool) is read by this wave again only when the domains are NOT kept in LDS; its
  // other readers — helper wave 2's class-table patch, the host — come behind the fences of AddUsage / the kernel's end
  if (own && in_lds) wsync_lds(); else wsync();
}

// ---- processEntry ------------------------------------------------------------------------------------------------------------------------
// Usage.TAS of entry e (Assignment.ComputeTASNetUsage flavorassigner.go:106-155, pending workloads) against / onto plane `which`:
// every (podset, domain) on its own (clusterqueue_snapshot.go:136-149)
// the TopologyAssignment entry processing sees for podset p: what the head published (global memory), or — k_process_tas after a
// recomputation of this very entry — the same domains where they still are, in store half 0 (LDS)
struct TcPub { int t, n; const int32_t *leaf, *count; };
KQ_DEV TcPub tc_pub(const TCyc& c, const Wave& w, int p) {
  if (w.ta.pub_lds) {
    const bool has = p < TC_P && ((w.ta.has_mask >> p) & 1);
    int32_t *kl, *kc; int cap;
    tc_dstore(c, w, 0, 0, &kl, &kc, &cap);
    return TcPub{has ? w.ta.t : -1, has ? w.ta.n[p] : 0, kl + (has ? w.ta.pos[p] : 0), kc + (has ? w.ta.pos[p] : 0)};
  }
  const int g = w.ps_base + p, t = c.h_tas[g];
  return TcPub{t, t >= 0 ? c.h_n[g] : 0, c.pool_leaf + c.h_pos[g], c.pool_count + c.h_pos[g]};
}
KQ_DEV bool tc_entry_fits(const K& k, const Wave& w, int which) {
  const TCyc& c = *k.tc;
  bool bad = false;
  for (int p = 0; p < w.nps; p++) {
    const TcPub a = tc_pub(c, w, p);
    if (a.t < 0) continue;
    const int g = w.ps_base + p;
    const TTopo& T = c.tk[a.t].T;
    const int64_t* pl = tc_plane(c, a.t, which, 0);
    for (int j = lane_id(); j < a.n; j += WAVE) {
      const int32_t cnt = a.count[j];
      if (cnt > 0 && !tc_fits_dom(T, pl, a.leaf[j], cnt, c.ps_req + (size_t)g * c.R)) bad = true;
    }
  }
  return wballot(bad) == 0;
}
KQ_DEV void tc_entry_add(const K& k, const Wave& w) {  // updateTASUsage :267 on the work plane and on the one without the preempted rows
  const TCyc& c = *k.tc;
  for (int p = 0; p < w.nps; p++) {
    const TcPub a = tc_pub(c, w, p);
    const int g = w.ps_base + p, t = a.t;
    if (t < 0) continue;
    const TTopo& T = c.tk[t].T;
    for (int j = lane_id(); j < a.n; j += WAVE) {
      const int64_t cnt = a.count[j];
      const int leaf = a.leaf[j];
      if (cnt <= 0) continue;
      for (int r = 0; r < T.R; r++) {
        const int64_t q = c.ps_req[(size_t)g * c.R + r];
        const int64_t v = (q > 0 ? q : 0) * cnt + (r == T.pods ? cnt : 0);
        if (!v) continue;
        atomic_add_i64((long long*)&c.work[t][(size_t)leaf * T.R + r], (long long)v);
        atomic_add_i64((long long*)&c.np[t][(size_t)leaf * T.R + r], (long long)v);
      }
    }
    wsync();
  }
#ifdef KQ_TAS_NO_ASYNC
  const bool async_ok = false;
#else
  const bool async_ok = true;
#endif
  if (async_ok && w.ta.mail && k.tc->ncls > 0) {
    // The class tables are read next by the copy job of the next placement, a barrier away: helper wave 2 patches them while the leader
    // goes on (one barrier to post, nobody waits). A later AddUsage that lands on the plane before wave 2 has read a leaf is simply seen
    // by this patch already (a patch recomputes its leaves from the plane and sends the difference to the table's old values up).
    TLeafJob& j = *w.ta.mail;
    t_post_begin(j);
    if (lane_id() == 0) { j.early_cls = -1; j.cu_ps_base = w.ps_base; j.cu_nps = w.nps; j.cu_lds_on = w.ta.lds != nullptr; j.cu_lds_bytes = w.ta.lds_bytes; j.cmd = 7; }   // (the tables move: whatever rows sit in LDS are stale)
    bsync();
#ifdef KQ_HOST_EMU
    t_class_update_job(j);
#endif
  } else tc_class_update(k, w.ps_base, w.nps, w.ta.lds != nullptr, w.ta.lds_bytes);
}
// scheduler.fits :771-777 -> ClusterQueueSnapshot.Fits :136-150: 0 = fits, 1 = no quota, 2 = no TAS capacity
KQ_DEV int tc_fits_check(const K& k, Wave& w, const int32_t* trows, int nt, bool quota_usage, int tree) {
  const TCyc& c = *k.tc;
  KQ_T0();
  const bool qfits = entry_fits(k, w, trows, nt, quota_usage, tree);
  KQ_TS(k, 38);   // (timing builds) scheduler.fits: the quota half, both calls of an entry
  if (!qfits) return 1;
  bool any = false;
  for (int p = 0; p < w.nps; p++) { const TcPub a = tc_pub(c, w, p); if (a.t >= 0 && a.n > 0) any = true; }
  if (!any) return 0;
  for (int i = 0; i < nt; i++) if (!k.preempted[trows[i]]) tc_row_apply(c, trows[i], false, 2, 0);
  const bool ok = tc_entry_fits(k, w, 2);
  for (int i = nt - 1; i >= 0; i--) if (!k.preempted[trows[i]]) tc_row_apply(c, trows[i], true, 2, 0);
  KQ_TS(k, 39);   // scheduler.fits: the leaf half
  return ok ? 0 : 2;
}

// the per-tree state of processEntry (Wave::np_broken, n_pre, broken) while ONE wave walks the entries of every tree in entry order:
// a TAS flavor's leaves are shared by ClusterQueues of different root cohorts (snapshot.go:260), so the trees cannot run side by side
KQ_DEV void tc_tree_switch(const K& k, Wave& w, int from, int to, const TPre* pre = nullptr) {
  int32_t* st = (w.ta.mail && k.S.n_tree <= KQ_TAS_TS_TREES) ? w.ta.mail->tstate : k.tc->tree_state;   // (the job block's copy starts zeroed, as the host's does)
  if (lane_id() == 0) {
    if (from >= 0) {
      int32_t* a = st + (size_t)from * 12;
      a[0] = w.np_broken; a[1] = w.n_pre;
      for (int i = 0; i < 4; i++) { a[2 + 2 * i] = (int32_t)(w.broken[i] & 0xffffffffu); a[3 + 2 * i] = (int32_t)(w.broken[i] >> 32); }
    }
    const int32_t* b = st + (size_t)to * 12;
    w.np_broken = b[0]; w.n_pre = b[1];
    for (int i = 0; i < 4; i++) w.broken[i] = (uint64_t)(uint32_t)b[2 + 2 * i] | ((uint64_t)(uint32_t)b[3 + 2 * i] << 32);
    w.pc_on = 0;
    w.pc_ncq = pre ? pre->tree_ncq : k.S.tree_cq_off[to + 1] - k.S.tree_cq_off[to];
    w.pc_ncoh = (pre ? pre->tree_nn : k.S.tree_node_off[to + 1] - k.S.tree_node_off[to]) - w.pc_ncq;
    w.pc_region_bytes = 0;
  }
  wsync();
}

// helper wave 2 of k_process_tas: the class tables follow the AddUsage the leader just made
KQ_DEV void t_class_update_job(TLeafJob& job) {
  tc_class_update(*(const K*)job.pf_k, job.cu_ps_base, job.cu_nps, job.cu_lds_on != 0, job.cu_lds_bytes);
}
// helper wave 1 of k_process_tas: the header of the entry at iterator position job.pf_next into job.pre[position & 1]
KQ_DEV void t_prefetch_entry(TLeafJob& job) {
  const K& k = *(const K*)job.pf_k;
  const DSnap& S = k.S; const DOut& O = k.O; const DHeads& H = k.H;
  const int pos = job.pf_next, lane = lane_id();
  TPre& r = job.pre[pos & 1];
  const int e = k.order_idx[pos];
  const int cq = H.cq[e];
  // what hangs on e and on cq, one item per lane (a loop over the items: the 1-lane emulation walks it)
  for (int it = lane; it < 10 + KQ_MAXD; it += WAVE) {
    switch (it) {
      case 0: r.e = e; r.cq = cq; r.prio = H.priority[e]; r.ts = H.queue_ts[e]; break;
      case 1: r.hflags = H.flags[e]; break;
      case 2: { const int a = H.ps_off[e]; r.ps_base = a; r.nps = H.ps_off[e + 1] - a; break; }
      case 3: r.slice_row = (H.slice_row && gate(k, KQ_GATE_ELASTIC_JOBS)) ? H.slice_row[e] : -1; break;
      case 4: r.borrowing = O.borrowing[e]; break;
      case 5: r.nominated_mode = O.nominated_mode[e]; break;
      case 6: r.tgt_n = O.tgt_n[e]; r.tgt_pos = O.tgt_pos[e]; break;
      case 7: r.pol = S.cq_policy[cq]; break;
      case 8: r.plen = S.plen[cq]; break;
      case 9: { const int tr = S.tree_of[cq]; r.tree = tr; r.tree_ncq = S.tree_cq_off[tr + 1] - S.tree_cq_off[tr]; r.tree_nn = S.tree_node_off[tr + 1] - S.tree_node_off[tr]; break; }
      default: {
        const int i = it - 10;
        const int nd = S.path[(size_t)cq * KQ_MAXD + i];
        r.path[i] = nd;
        r.node_local[i] = (i < S.plen[cq] && nd >= 0) ? S.node_local[nd] : 0;
      }
    }
  }
  {  // the request block's static half (tc_find), when there is one podset and one TAS flavor
    const TCyc& c = *k.tc;
    const int g = H.ps_off[e];
    const bool one = c.n_tas == 1 && H.ps_off[e + 1] - g == 1 && c.R <= KQ_TAS_PF_R && c.ps_n_layers == nullptr;
    if (lane == 0) r.q_ok = one ? 1 : 0;
    if (one) {
      for (int it = lane; it < 7 + c.R; it += WAVE) {
        switch (it) {
          case 0: { const int cls = c.ncls > 0 ? c.ps_class[g] : -1; r.q_cls = cls; r.q_cls_ok = cls >= 0 ? c.cls_ok[cls] : 0; break; }
          case 1: r.q_level = c.ps_level[g]; break;
          case 2: r.q_ssize = c.ps_slice_size[g]; break;
          case 3: r.q_slevel = c.ps_slice_level[g]; break;
          case 4: r.q_group = c.ps_group[g]; break;
          case 5: r.q_kind = c.ps_kind[g]; break;
          case 6: break;
          default: r.q_req[it - 7] = c.ps_req[(size_t)g * c.R + (it - 7)];
        }
      }
    }
  }
  const int nu = O.use_n[e];
  if (lane == 0) r.nuse = nu;
  for (int u = lane; u < nu && u < KQ_TAS_PF_MAXU; u += WAVE) { r.use_fr[u] = O.use_fr[(size_t)e * KQ_MAXU + u]; r.use_qty[u] = O.use_qty[(size_t)e * KQ_MAXU + u]; }
  wsync();
  if (lane == 0) *(volatile int*)&r.ready_for = nu <= KQ_TAS_PF_MAXU ? pos : -1;
  wsync();
}

// A recomputation on the work plane that will place one podset of a request class (the usual entry of a TAS cycle): its class's rows
// are copied into LDS by the helper waves while the leader runs the flavor assignment (cmd 8, split-phase; t_class_to_lds joins it).
// One TAS flavor only: which flavor a podset lands on is the assignment's to decide.
KQ_DEV void tc_early_copy(const K& k, Wave& w, bool overlap) {
  const TCyc& c = *k.tc;
  if (overlap || !w.ta.mail || !w.ta.lds || c.n_tas != 1 || c.ncls == 0 || w.nps != 1 || w.ta.mail->nw < 2) return;
  const int cls = c.ps_class[w.ps_base];
  if (cls < 0 || !c.cls_ok[cls]) return;
  const TK& tk0 = c.tk[0];
  const TLdsLay l = tas_lds_layout(tk0.T.D, tk0.X.max_set);
  if (TX_BYTES + l.total > (size_t)w.ta.lds_bytes) return;
  TLeafJob& j = *w.ta.mail;
  t_post_begin(j);
  unsigned char* base = w.ta.lds + TX_BYTES;
  if (lane_id() == 0) {
    const int32_t* tab = c.cls_tab[0];
    j.a.pc = const_cast<int32_t*>(tab) + (size_t)cls * tk0.T.D; j.a.sc = const_cast<int32_t*>(tab) + (size_t)c.ncls * tk0.T.D + (size_t)cls * tk0.T.D;
    j.cp_pc = (int32_t*)(base + l.pc); j.cp_sc = (int32_t*)(base + l.sc); j.cp_n = tk0.T.D;
    j.early_cls = cls; j.early_pending = 1; j.cmd = 8;
  }
  bsync();
#ifdef KQ_HOST_EMU
  for (int wv = 1; wv < j.nw; wv++) t_helper_step(j, wv, j.nw);
#endif
}

// scheduler.go:392-523 for entry e at iterator position pos; the generic path of process_entry with the TAS side of
// updateAssignmentIfNeeded (:707-769), Fits and AddUsage
KQ_NOINLINE void process_entry_tas(const K& k, Wave& w, int e, int pos, int slot, int tree, const TPre* pre = nullptr) {
  const DSnap& S = k.S; const DOut& O = k.O; const TCyc& c = *k.tc;
  const int lane = lane_id();
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  const long long _h0 = clock64();
#endif
  int nt, tpos;
  if (lane == 0) { w.ta.pub_lds = 0; w.ta.cur_pre = pre; }   // what the head published came from k_nominate_tas: global memory
  if (pre) {
    // the header came with helper wave 1 (LDS): load_head + the nomination without a global access
    if (lane == 0) {
      w.h = e; w.cq = pre->cq; w.prio = pre->prio; w.ts = pre->ts; w.hflags = pre->hflags; w.pol = pre->pol;
      w.ps_base = pre->ps_base; w.nps = pre->nps; w.plen = pre->plen;
      for (int i = 0; i < KQ_MAXD; i++) w.path[i] = pre->path[i];
      w.has_last = (pre->hflags & KQ_HEAD_HAS_LAST_ASSIGNMENT) ? 1 : 0;
      w.bytes = 0;
      w.slice_row = pre->slice_row;
      if (w.nps > KQ_MAXPS) *k.O.error = KQ_EUNSUPPORTED;
      w.nuse = pre->nuse; w.borrowing = pre->borrowing; w.rep_mode = pre->nominated_mode;
      O.order[e] = pos;
      for (int i = 1; i < pre->plen; i++) w.path_coh[i] = pre->node_local[i] - w.pc_ncq;
    }
    for (int u = lane; u < pre->nuse; u += WAVE) { w.use_fr[u] = pre->use_fr[u]; w.use_qty[u] = pre->use_qty[u]; }
    nt = pre->tgt_n; tpos = pre->tgt_pos;
    wsync();
  } else {
    load_head(k, w, e);
    if (lane == 0) {
      w.nuse = O.use_n[e];
      for (int u = 0; u < w.nuse; u++) { w.use_fr[u] = O.use_fr[(size_t)e * KQ_MAXU + u]; w.use_qty[u] = O.use_qty[(size_t)e * KQ_MAXU + u]; }
      w.borrowing = O.borrowing[e];
    }
    wsync();
    if (lane == 0) {
      w.rep_mode = O.nominated_mode[e];
      O.order[e] = pos;
      for (int i = 1; i < w.plen; i++) w.path_coh[i] = S.node_local[w.path[i]] - w.pc_ncq;
    }
    wsync();
    nt = O.tgt_n[e]; tpos = O.tgt_pos[e];
  }
  const bool quota_usage = !(w.hflags & KQ_HEAD_HAS_QUOTA_RESERVATION);
  if (w.rep_mode != M_NOFIT) cert_unverifiable(k, tree);
  const int32_t* trows = O.pool_row + tpos;
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  if (lane == 0) { atomic_add_i64((long long*)k.prof + 36, clock64() - _h0); if (pre) atomic_add_i64((long long*)k.prof + 34, 1); }   // the entry's header; headers that came prefetched
#endif
  auto has_any = [&]() { bool a = false; for (int t = 0; t < nt; t++) if (k.preempted[trows[t]]) a = true; return a; };
  KQ_T0();
  int fc = tc_fits_check(k, w, trows, nt, quota_usage, tree);
  KQ_TS(k, 40);   // (timing builds) head + first fits
  int mode = w.rep_mode;
  const bool overlap = has_any() && gate(k, KQ_GATE_RECOMPUTE_ON_OVERLAP);
  const bool tas_recompute = fc == 2 && !(c.flags & KQ_CT_NO_RECOMPUTE);
  if (overlap || tas_recompute) {
    if (overlap && np_exact_mode(w)) np_rebuild(k, w, tree, false);
    if (!overlap && lane == 0 && c.stats) atomic_add_i64(c.stats + 1, 1);
    if (lane == 0) { w.has_last = 0; w.ta.plane = overlap ? 2 : 1; }
    tc_early_copy(k, w, overlap);
    for (int i = lane; i < w.nps * S.nR; i += WAVE) k.X.nom[(size_t)slot * KQ_MAXPS * S.nR + i] = O.flavor[(size_t)w.ps_base * S.nR + i];
    wsync();
    // overlap: SimulateWorkloadRemoval(victimsOfOtherPreemptions) = the planes without the preempted rows; TAS only: the snapshot as it is
    const int64_t* plane = overlap ? k.usage_np : k.usage_work;
    const uint8_t* removed = overlap ? k.preempted : nullptr;
    KQ_TS(k, 41);
    Search s = get_assignments(k, w, slot, plane, removed, true);
    KQ_TS(k, 42);   // the recomputation (45 / 46 are inside it)
    publish_assignment(k, w, s, e);
    KQ_TS(k, 37);   // publish
    trows = O.pool_row + O.tgt_pos[e];
    nt = O.tgt_n[e];
    mode = w.rep_mode;
    if (overlap && mode == M_FIT) {  // SetRepresentativeMode(DeferredFit) flavorassigner.go:109-114
      mode = M_DEFERRED;
      for (int i = lane; i < w.nps * S.nR; i += WAVE) {
        const size_t o = (size_t)w.ps_base * S.nR + i;
        if (O.flavor[o] >= 0) O.res_mode[o] = M_DEFERRED;
      }
    }
    wsync();
    fc = tc_fits_check(k, w, trows, nt, quota_usage, tree);
    KQ_TS(k, 43);   // publish + second fits
  }
  const bool fits_ok = fc == 0;
  int status = KQ_ST_NOT_NOMINATED, action = KQ_ACT_NONE, rq = KQ_RQ_GENERIC, skip = KQ_SKIP_NONE;
  bool done = false;
  if ((w.hflags & KQ_HEAD_UNHEALTHY_ASSIGNMENT) && !(c.flags & KQ_CT_NO_FAIL_FAST) && mode != M_FIT) {
    // TASFailedNodeReplacementFailFast scheduler.go:425-428: no replacement for the failed node -> handleFailedTASReplacement :522
    status = KQ_ST_EVICTED; action = KQ_ACT_EVICT; done = true;
  }
  if (!done && mode == M_NOFIT) { rq = KQ_RQ_NOFIT; done = true; }
  if (!done && mode == M_PREEMPT && nt == 0) {
    rq = KQ_RQ_PREEMPTION_NO_CANDIDATES;
    const bool can_always_reclaim = KQ_POL_RECLAIM(w.pol) == KQ_POLICY_ANY;
    if (!can_always_reclaim || (gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS) && (w.hflags & KQ_HEAD_IS_PREEMPTOR))) {
      if (quota_usage) {
        if (lane == 0)
          for (int u = 0; u < w.nuse; u++) {
            const int fr = w.use_fr[u];
            w.s_qty[u] = reserve_amount(w.use_qty[u], S.nominal[ix(S, w.cq, fr)], S.bl[ix(S, w.cq, fr)], k.usage_work[ix(S, w.cq, fr)], w.borrowing);
            if (w.s_qty[u] < 0) mark_broken(w, fr);
          }
        wsync();
        entry_add_usage(k, w, w.s_qty);
      }
      tc_entry_add(k, w);   // resourcesToReserve -> netUsage :785-794 carries Usage.TAS
    }
    done = true;
  }
  if (!done && mode == M_DEFERRED) {
    rq = KQ_RQ_PENDING_PREEMPTION;
    if (quota_usage) entry_add_usage(k, w, w.use_qty);
    tc_entry_add(k, w);
    done = true;
  }
  if (!done && has_any()) { status = KQ_ST_SKIPPED; skip = KQ_SKIP_OVERLAP; done = true; }
  if (!done && !fits_ok) { status = KQ_ST_SKIPPED; skip = KQ_SKIP_NO_LONGER_FITS; done = true; }
  if (!done) {
    // preemptedWorkloads.Insert(targets): the rows leave the np planes (quota and leaf usage) for the rest of the cycle
    int fresh = 0;
    for (int base = 0; base < nt; base += WAVE) {
      const int t = base + lane;
      bool isnew = false;
      if (t < nt) {
        const int row = trows[t];
        if (!k.preempted[row]) {
          isnew = true;
          k.preempted[row] = 4;   // new in this pass (np_apply_targets skips nothing below: restricted = false)
          atomic_add_i32(&k.cq_rm_bytes[S.adm_cq[row]], 32 + 12 * (S.adm_use_off[row + 1] - S.adm_use_off[row]));
        } else if (k.preempted[row] == 1) k.preempted[row] = 3;
      }
      fresh += popc64(wballot(isnew));
    }
    if (lane == 0) w.n_pre += fresh;
    wsync();
    for (int t = 0; t < nt; t++) if (k.preempted[trows[t]] == 4) tc_row_apply(c, trows[t], false, 2, 0);
    for (int t = lane; t < nt; t += WAVE) if (k.preempted[trows[t]] == 4) k.preempted[trows[t]] = 1;
    wsync();
    np_apply_targets(k, w, trows, nt, false, false, tree);
    for (int t = lane; t < nt; t += WAVE) if (k.preempted[trows[t]] == 3) k.preempted[trows[t]] = 1;
    wsync();
    KQ_TS(k, 62);   // (timing builds) preemptedWorkloads.Insert
    if (quota_usage) entry_add_usage(k, w, w.use_qty);
    KQ_TS(k, 63);   // AddUsage on the quota planes
    tc_entry_add(k, w);
    KQ_TS(k, 61);   // leaf usage + the class tables
    if (mode == M_PREEMPT) { action = KQ_ACT_PREEMPT; rq = KQ_RQ_PENDING_PREEMPTION; }
    else { status = KQ_ST_ASSUMED; action = KQ_ACT_ADMIT; }
  }
  write_entry_result(k, w, e, status, action, rq, skip, mode);
  if (lane == 0) atomic_add_i64(O.stat_bytes, (long long)w.bytes);
  wsync();
  KQ_TS(k, 44);   // usage added, result written
}

// k_process_tas under fair sharing: ONE wave plays the fair-sharing iterator (fair_sharing_iterator.go:47-263) of every root tree,
// interleaved the way the reference's single iterator does — with the canonical getCq the lowest ClusterQueue
// index still in the map names the tree that pops next. TAS leaves are shared across root cohorts (snapshot.go:260), so the trees cannot
// run side by side as they do in k_process_fair. Per-tree iterator state lives in the scratch rows of slot = tree, exactly as there:
// cqToEntry, the cached tournament keys with their stale levels and costs, the cohort winners. Same pops, same byte count.
KQ_DEV void process_all_fair_tas(const K& k, Wave& w, int slot) {
  const DSnap& S = k.S; const DOut& O = k.O; const DHeads& H = k.H;
  const int lane = lane_id();
  const int n = hn(H);
  const bool want_bon = gate(k, KQ_GATE_FS_PRIORITIZE_NON_BORROWING);
  const bool fs_plain = fs_plain_now(k);
  auto ents = [&](int t) { return k.X.cq_ent + (size_t)t * k.X.max_tree_cqs; };
  for (int t = 0; t < S.n_tree; t++) {
    const int nqs = S.tree_cq_off[t + 1] - S.tree_cq_off[t], nn = S.tree_node_off[t + 1] - S.tree_node_off[t];
    int32_t* cq_ent = ents(t);
    uint8_t* stale = k.X.fs_stale + (size_t)t * k.X.max_tree_cqs;
    int32_t* cost = k.X.fs_cost + (size_t)t * k.X.max_tree_cqs * KQ_MAXD;
    int32_t* win = k.X.fs_win + (size_t)t * k.X.max_tree_nodes;
    for (int i = lane; i < nqs; i += WAVE) { cq_ent[i] = -1; stale[i] = 0; }
    for (int i = lane; i < nqs * KQ_MAXD; i += WAVE) cost[i] = 0;
    for (int i = lane; i < nn; i += WAVE) win[i] = -1;
    if (lane == 0) { k.X.fs_sum[t] = 0; int32_t* ctl = k.X.fs_ctl + (size_t)t * 4; ctl[0] = 0; ctl[1] = -1; ctl[2] = 0; ctl[3] = 0; }
  }
  wsync();
  // cqToEntry: the last head of a CQ wins (:58-60)
  for (int h = lane; h < n; h += WAVE) { const int c = H.cq[h]; atomic_max_i32(&ents(S.tree_of[c])[S.cq_local[c]], h); }
  wsync();
  int cur = -1, pos = 0;
  for (int c = 0; c < S.nq;) {
    const int t = S.tree_of[c];
    int32_t* cq_ent = ents(t);
    if (cq_ent[S.cq_local[c]] < 0) { c++; continue; }   // getCq: the lowest ClusterQueue index still in the map
    if (t != cur) { tc_tree_switch(k, w, cur, t); cur = t; }
    const int q0 = S.tree_cq_off[t], nqs = S.tree_cq_off[t + 1] - q0;
    const int n0 = S.tree_node_off[t], nn = S.tree_node_off[t + 1] - n0;
    if (nn == 1) {  // ClusterQueue without Cohort: its workload is simply returned (:71-78)
      const int e = cq_ent[0];
      wsync();
      if (lane == 0) cq_ent[0] = -1;
      wsync();
      process_entry_tas(k, w, e, pos++, slot, t);
      continue;
    }
    int32_t* win = k.X.fs_win + (size_t)t * k.X.max_tree_nodes;
    uint64_t* fkeys = k.X.fs_keys + (size_t)t * k.X.max_tree_cqs * KQ_MAXD * 4;
    uint8_t* stale = k.X.fs_stale + (size_t)t * k.X.max_tree_cqs;
    int32_t* cost = k.X.fs_cost + (size_t)t * k.X.max_tree_cqs * KQ_MAXD;
    long long* sum = k.X.fs_sum + t;
    int32_t* ctl = k.X.fs_ctl + (size_t)t * 4;
    // ---- computeDRS (:227-263): the levels of every remaining entry that the tree's last pop made stale ----
    {
      const int xc = ctl[1];
      const bool changed = ctl[2] != 0;
      int64_t delta = 0;
      for (int i = lane; i < nqs; i += WAVE) {
        const int en = cq_ent[i];
        if (en < 0) continue;
        const int cq = S.tree_cqs[q0 + i];
        const int32_t* path = S.path + (size_t)cq * KQ_MAXD;
        const int plen = S.plen[cq];
        int from = stale[i];
        if (changed) {  // lowest level of this path that is also on the popped entry's path
          const int32_t* xp = S.path + (size_t)xc * KQ_MAXD;
          const int xl = S.plen[xc];
          int tt = 0;
          while (tt < plen - 1 && tt < xl - 1 && path[plen - 1 - tt - 1] == xp[xl - 1 - tt - 1]) tt++;
          const int lvl = plen - 1 - tt;
          if (lvl < from) from = lvl;
        }
        if (from + 1 >= plen) { if (from != 255) stale[i] = 255; continue; }
        PE pe{&k, &w, path, 0, O.use_fr + (size_t)en * KQ_MAXU, O.use_qty + (size_t)en * KQ_MAXU,
              (H.flags[en] & KQ_HEAD_HAS_QUOTA_RESERVATION) ? 0 : O.use_n[en]};  // netUsage scheduler.go:785-794
        for (int l = from; l + 1 < plen; l++) {
          pe.level = l;
          int64_t lb = 0;
          DRSv d = fs_plain ? drs_entry_level(k, w, path, l, pe.ufr, pe.uqty, pe.nu, want_bon, &lb)
                            : drs_of(S, path[l], pe, &lb, pe.ufr, pe.uqty, want_bon ? pe.nu : 0);
          const size_t o = (size_t)i * KQ_MAXD + l;
          const FsKey key = fs_make_key(k, en, d);
          fkeys[o * 4 + 0] = key.k1; fkeys[o * 4 + 1] = key.k2; fkeys[o * 4 + 2] = key.k3; fkeys[o * 4 + 3] = key.k4;
          delta += lb - cost[o];
          cost[o] = (int32_t)lb;
        }
        stale[i] = 255;
      }
      const int64_t tot = wsum_i64(delta);
      if (lane == 0 && tot) *sum += tot;
      wsync();
    }
    // ---- the tournaments (:125-163): all of them at the tree's first pop, afterwards the cohorts on the last popped entry's path ----
    if (!ctl[3]) {
      for (int d = KQ_MAXD - 1; d >= 0; d--)
        for (int i = nqs; i < nn; i++) {
          const int x = S.tree_nodes[n0 + i];
          if (S.depth[x] != d) continue;
          const int b = tournament_cohort(k, t, x, win, cq_ent);
          if (lane == 0) win[i] = b;
          wsync();
        }
    } else {
      const int xc = ctl[1];
      for (int l = 1; l < S.plen[xc]; l++) {
        const int x = S.path[(size_t)xc * KQ_MAXD + l];
        const int b = tournament_cohort(k, t, x, win, cq_ent);
        if (lane == 0) win[S.node_local[x]] = b;
        wsync();
      }
    }
    const int root = S.path[(size_t)S.tree_cqs[q0] * KQ_MAXD + S.plen[S.tree_cqs[q0]] - 1];
    const int e = win[S.node_local[root]];
    const int ec = H.cq[e], ei = S.cq_local[ec];
    wsync();
    if (lane == 0) {
      atomic_add_i64(O.stat_bytes, *sum);  // the reference evaluates every remaining (entry, level) on every pop
      long long mine = 0;
      for (int l = 0; l + 1 < S.plen[ec]; l++) mine += cost[(size_t)ei * KQ_MAXD + l];
      *sum -= mine;
      cq_ent[ei] = -1;
      w.usage_dirty = 0;
    }
    wsync();
    process_entry_tas(k, w, e, pos++, slot, t);
    if (fs_plain && w.usage_dirty) {  // the rows of the popped entry's path changed: refresh their sums
      const int pl = S.plen[ec];
      PW pw{&k, &w};
      for (int j = lane; j < pl * S.nR; j += WAVE) {
        const int nd = S.path[(size_t)ec * KQ_MAXD + j / S.nR], r = j % S.nR;
        int64_t sm; int ps;
        node_sums(S, nd, pw, r, &sm, &ps);
        k.X.bs_sum[(size_t)nd * S.nR + r] = sm;
        w.cell_borrow[j] = ps;
      }
      wsync();
      for (int j = lane; j < pl; j += WAVE) {
        int ps = 0;
        for (int r = 0; r < S.nR; r++) ps += w.cell_borrow[j * S.nR + r];
        k.X.bs_pos[S.path[(size_t)ec * KQ_MAXD + j]] = ps;
      }
      wsync();
    }
    if (lane == 0) { ctl[1] = ec; ctl[2] = w.usage_dirty; ctl[3] = 1; }
    wsync();
  }
}

// k_process_tas: one wave walks every entry in iterator order
KQ_DEV void process_all_tas(const K& k, Wave& w, int slot, TLeafJob* mail, unsigned char* lds, int lds_bytes) {
  const int n = hn(k.H);
  if (lane_id() == 0) {
    w.ta.mail = mail; w.ta.lds = lds_bytes > 0 ? lds : nullptr; w.ta.lds_bytes = lds_bytes; w.ta.pf_pos = -1;
    w.ta.cur_pre = nullptr; w.ta.q_lds = 0; w.ta.pub_lds = 0; w.ta.d_lds = (lds_bytes >= (int)TX_BYTES && k.tc->d_cap <= TX_DCAP) ? 1 : 0;
    w.ta.pool_own = 1; w.ta.pool_next = *k.tc->pool_used; w.ta.req_valid = 0; w.ta.em_ps = -1;
    if (mail) {
      mail->pf_k = &k; mail->pf_next = -1; mail->pre[0].ready_for = -1; mail->pre[1].ready_for = -1; mail->early_pending = 0; mail->early_cls = -1;
      for (int i = 0; i < KQ_TAS_TS_TREES * 12; i++) mail->tstate[i] = 0;
    }
    w.pc_on = 0; w.pc_lds = nullptr; w.np_broken = 0; w.n_pre = 0; w.broken[0] = w.broken[1] = w.broken[2] = w.broken[3] = 0;
    w.cs_lds = nullptr; w.cs_lds_bytes = 0; w.help_on = 0; w.mono_break = 0; w.ta.plane = 1; w.ta.srch = 0;
  }
  wsync();
  if (k.C.fair_sharing) { process_all_fair_tas(k, w, slot); return; }
  int cur = -1;
  KQ_T0();
  for (int i = 0; i < n; i++) {
    // the entry's header: from helper wave 1 if it was fetched while the previous entry finished, else from global memory
    const TPre* pre = nullptr;
    if (mail) {
      const TPre* cand = &mail->pre[i & 1];
      if (*(volatile const int*)&cand->ready_for == i) pre = cand;
      wsync();
    }
    const int e = pre ? pre->e : k.order_idx[i];
    const int tree = pre ? pre->tree : k.S.tree_of[k.H.cq[e]];
    if (lane_id() == 0) w.ta.pf_pos = (mail && i + 1 < n) ? i + 1 : -1;
    if (tree != cur) { tc_tree_switch(k, w, cur, tree, pre); cur = tree; }
    KQ_TS(k, 35);   // (timing builds) position -> entry, tree switch
    process_entry_tas(k, w, e, i, slot, tree, pre);
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
    _t0 = clock64();
#endif
  }
}

// planes at the start of a cycle: base = tas_usage + the admitted rows' usage (tas_flavor.go: the cache adds workload.TASUsage() of every
// admitted workload when it builds the flavor snapshot); one thread per domain entry of the CSR
KQ_DEV void tc_base_cell(const TCyc& c, int e) {
  const int t = c.adm_tas[e];
  const TTopo& T = c.tk[t].T;
  const int64_t cnt = c.adm_count[e];
  for (int r = 0; r < T.R; r++) {
    const int64_t q = c.adm_req[(size_t)e * T.R + r];
    const int64_t v = (q > 0 ? q : 0) * cnt + (r == T.pods ? cnt : 0);
    if (v) atomic_add_i64((long long*)&T.tas_usage[(size_t)c.adm_leaf[e] * T.R + r], (long long)v);
  }
}

#endif  // KQ_TAS_CYCLE

}  // namespace kq

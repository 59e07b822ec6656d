// kq_tas_device.hpp — Topology-Aware Scheduling on gfx950: FindTopologyAssignmentsForFlavor
// (pkg/cache/scheduler/tas_flavor_snapshot.go:578) for a batch of workloads, one wavefront per workload.
//
//   phase 1  fillLeafCounts :1899 / CountInWithLimitingResource (pkg/resources/requests.go:195): lanes = leaves, the leaf
//            capacity / usage rows are read coalesced ([leaf][R] int64); fillInCountsHelper :1930: level by level,
//            lanes = domains (children of a domain are a contiguous id range because domains are numbered in the
//            lexicographic order of their levelValues).
//   phase 2  the reference sorts domain slices and walks them sequentially with early exits. Here a sorted slice is a
//            LAZY view: packed 128-bit sort keys are built once per slice (lanes = elements), elements are produced
//            in order on demand by a wave arg-min over "key > last key" (only as many as the greedy loops consume —
//            domains whose state is all zero can never change a decision and are left out), and the "best fit over
//            the rest of the slice" scans (findBestFitDomainBy :1307) are wave min-reductions over the same keys.
//
// Everything is integer work on int32 counts / int64 capacities; bound by L2/HBM reads of the leaf table.
// The same code compiles for the 1-lane CPU emulation used by the CPU test-suite (KQ_HOST_EMU).
#pragma once
#include "../../include/kq_tas.h"
#include "kq_device.hpp"
#include <cmath>   // (the entropy order of balanced placement: frexp / log on the host build; the device build has HIP's)

namespace kq {

struct TTopo {
  int L, R, pods, profile_mixed, D, n_leaves, leaf_base;
  int balanced;                // features.TASBalancedPlacement (KQ_TAS_F_BALANCED_PLACEMENT): kq_tas_find only
  int level_off[KQ_TAS_MAX_LEVELS + 1];
  const int32_t* child_first;  // [D] global id of the first child (children are contiguous), -1 for leaves
  const int32_t* child_cnt;    // [D]
  const int64_t* free_cap;     // [n_leaves][R]
  int64_t* tas_usage;          // [n_leaves][R]
};
struct TReq {
  int n_wl;
  const int32_t* wl_off;
  const uint8_t* sim_empty;
  const int64_t* spr;
  const int32_t *count, *level;
  const uint8_t* kind;
  const int32_t *slice_size, *slice_level, *group;
  const uint8_t* leaf_ok;
  // kq_cycle_run_tas: the requests share mask ROWS — request i uses row leaf_ok_idx[i] of leaf_ok (stride leaf_ok_stride), -1 = every
  // leaf; NULL = the batch layout, one row of n_leaves per request
  const int32_t* leaf_ok_idx = nullptr;
  int leaf_ok_stride = 0;
  const int32_t *n_layers, *layer_level, *layer_size;  // TASMultiLayerTopology: NULL = single layer everywhere
  // kq_tas_find_elastic: assumed usage a workload starts with — the previous pods of its elastic slice (handleScaleUp
  // tas_elastic_workloads.go:97-106): CSR per workload of (leaf, count, podset whose SinglePodRequests the pods carry); NULL = none
  const int32_t *seed_off, *seed_leaf, *seed_count, *seed_ps;
  // the leaves below a required replacement domain (tas_flavor_snapshot.go:1902-1905), per podset request: [leaf_lo, leaf_hi) in leaf
  // indices; NULL = every leaf. (kq_cycle_run_tas's second pass; the batch entry point kq_tas_find_replacement passes a leaf_ok row.)
  const int32_t *leaf_lo, *leaf_hi;
};
struct TOut {
  int32_t *status, *op_a, *op_b, *dom_pos, *dom_n;  // per podset request; dom_pos = offset into the pool
  int32_t* layer_fit;  // [n][KQ_TAS_MAX_LEVELS] or NULL
  int32_t *pool_leaf, *pool_count;
  int32_t pool_cap;
  int32_t* pool_used;  // [1] atomic
  int32_t* error;      // [1]
  long long* bytes;    // [1]
};
struct TScratch {  // per wave slot
  int32_t *pc, *sc, *pcwl, *scwl, *lc;  // [slots][D] domainState :55
  int32_t *set, *arr, *cur, *nxt;       // [slots][n_leaves] id lists: the slice being sorted, its materialised prefix, currFitDomain x2
  uint64_t *k0, *k1;                    // [slots][n_leaves] sort keys of `set`
  int64_t* assumed;                     // [slots][n_leaves][R] assumedUsage of the workload (:586)
  int32_t* log;                         // [slots][max_set] domains whose state phase 2 changed
  int32_t* meta;                        // [slots][4] log length, state no longer restorable, class the state was copied from
  int32_t max_set;
  // balanced placement (tas_balanced_placement.go; T.balanced): per-slot scratch of bal_stride int32 words — two saved copies of the
  // domain state, four domain lists of the widest level, the dynamic programme's table (bal_dp entries of two words)
  int32_t* bal; long long bal_stride; long long bal_dp; int32_t bal_w;
};
// Phase 1 only depends on (requests, leader requests, simulate-empty, slice size / level) — not on the pod count, the
// requested level or required / preferred / unconstrained. Workloads with one podset group and no feasibility mask are
// grouped into request classes on the host; phase 1 runs once per class (k_tas_classes) and a workload starts phase 2
// from its class's table. Phase 2 changes the state of the few domains it consumes: those are logged and restored.
struct TClass {
  int n;
  const int32_t *workers, *leader;  // [n] representative podset requests (leader = -1: none)
  const uint8_t* sim_empty;         // [n]
  int32_t *pc, *sc, *pcwl, *scwl, *lc;  // [n][D]
  long long* bytes;                 // [n] algorithmic bytes of one phase 1 of the class
  const int32_t* wl_class;          // [n_wl] class of the workload, -1 = private phase 1
  const int32_t* order;             // [n_wl] workloads sorted by class: a slot walks a contiguous piece
};
struct TLeafJob;
// k_process_tas only: the working state of a class-path placement in the workgroup's LDS instead of the slot's global rows. A lone wave
// pays an L2 round trip (1 500-2 000 cycles) per dependent access to its scratch rows, and a placement is a chain of them: sweep, reduce,
// write the winner, read it back. In LDS the same chain costs ~100 cycles a link. What lives there: the counts (podCount, sliceCount
// of every domain: a class has no leader, so the ...WithLeader arrays ARE these arrays and there is no leaderCount array at all,
// TState::nolead), the sort keys of the slice being walked, its materialised prefix, currFitDomain x2 and the meta words. The slice's
// id list (set: written once, read by index) and the log stay in global memory. 148.5 KB for 4096 leaves, next to 8.8 KB of static LDS.
struct TLdsLay { size_t k0, k1, pc, sc, arr, cur, nxt, meta, total; };
KQ_HD TLdsLay tas_lds_layout(int D, int max_set) {
  TLdsLay l{};
  size_t o = 0;
  auto take = [&](size_t b) { const size_t at = o; o += (b + 15) & ~(size_t)15; return at; };
  l.k0 = take((size_t)max_set * 8); l.k1 = take((size_t)max_set * 8);
  l.pc = take((size_t)D * 4); l.sc = take((size_t)D * 4);
  l.arr = take(((size_t)max_set + 1) * 4); l.cur = take((size_t)max_set * 4); l.nxt = take((size_t)max_set * 4); l.meta = take(16);
  l.total = o;
  return l;
}
struct TK { TTopo T; TReq Q; TOut O; TScratch X; TClass C; TLeafJob* mail; unsigned char* lds; };  // mail: phase 1 shared with the helper waves of the workgroup (null: none); lds: see TLds (null: the slot's global rows)

struct TState {  // per-slot pointers
  int32_t *pc, *sc, *pcwl, *scwl, *lc, *set, *arr, *cur, *nxt;
  uint64_t *k0, *k1;
  int64_t* assumed;
  int32_t *log, *meta;
  int logcap;
  bool nolead;   // leaderCount is 0 everywhere and stays 0: no lc array
  TLeafJob* coop;   // the state is in LDS and the workgroup has helper waves: long slices are swept together (null: alone)
  int coop_min;
  bool lazy_keys;   // the slot's rows are global memory: t_find_level writes a level's id list and sort keys only when something reads them
  int bal_slot;     // the slot's index (TScratch::bal)
};
// lane 0 records a domain whose state it is about to change
KQ_DEV void t_touch(const TState& s, int d) {
  const int n = s.meta[0];
  if (n < s.logcap) { s.log[n] = d; s.meta[0] = n + 1; } else s.meta[1] = 1;
}
// The arrays of a state in LDS are reached through generic pointers: a `flat_*` access, which waits for the wave's outstanding GLOBAL
// traffic as well (one counter pair) — the fire-and-forget stores of the id lists and the log would be back on the chain. Told that the
// pointers are LDS, the compiler emits `ds_*`; that takes a second instantiation of the placement (LDS = true) next to the one every
// other caller uses.
#if !defined(KQ_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
#define KQ_TAS_IS_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#else
#define KQ_TAS_IS_LDS(p) do {} while (0)   // (the emulation, and hipcc's host pass)
#endif
KQ_DEV void t_assume_lds(const TState& s) {
  KQ_TAS_IS_LDS(s.pc); KQ_TAS_IS_LDS(s.sc); KQ_TAS_IS_LDS(s.pcwl); KQ_TAS_IS_LDS(s.scwl); KQ_TAS_IS_LDS(s.k0); KQ_TAS_IS_LDS(s.k1);
  KQ_TAS_IS_LDS(s.arr); KQ_TAS_IS_LDS(s.cur); KQ_TAS_IS_LDS(s.nxt); KQ_TAS_IS_LDS(s.meta);
}
// leaderCount of domain d: a state without leaders (TState::nolead: the LDS working copy of a request class) holds no such array
KQ_DEV int32_t t_lc(const TState& s, int d) { return s.nolead ? 0 : s.lc[d]; }
KQ_DEV void t_set_lc(const TState& s, int d, int32_t v) { if (!s.nolead) s.lc[d] = v; }
KQ_DEV TState tas_state(const TK& k, int slot) {
  TState s;
  const size_t D = k.T.D, M = k.X.max_set;
  s.pc = k.X.pc + slot * D; s.sc = k.X.sc + slot * D; s.pcwl = k.X.pcwl + slot * D; s.scwl = k.X.scwl + slot * D; s.lc = k.X.lc + slot * D;
  s.set = k.X.set + slot * M; s.arr = k.X.arr + slot * M; s.cur = k.X.cur + slot * M; s.nxt = k.X.nxt + slot * M;
  s.k0 = k.X.k0 + slot * M; s.k1 = k.X.k1 + slot * M;
  s.assumed = k.X.assumed + (size_t)slot * k.T.n_leaves * k.T.R;
  s.log = k.X.log + slot * M; s.meta = k.X.meta + (size_t)slot * 4; s.logcap = (int)M;
  s.nolead = false; s.coop = nullptr; s.coop_min = 0; s.bal_slot = slot; s.lazy_keys = k.lds == nullptr;
  if (k.lds) {
    const TLdsLay l = tas_lds_layout(k.T.D, (int)M);
    s.k0 = (uint64_t*)(k.lds + l.k0); s.k1 = (uint64_t*)(k.lds + l.k1);
    s.pc = (int32_t*)(k.lds + l.pc); s.sc = (int32_t*)(k.lds + l.sc);
    s.pcwl = s.pc; s.scwl = s.sc;   // no leader in a class: the placement never writes the ...WithLeader counts, and the table holds the same values
    s.lc = nullptr; s.nolead = true;
    s.arr = (int32_t*)(k.lds + l.arr); s.cur = (int32_t*)(k.lds + l.cur); s.nxt = (int32_t*)(k.lds + l.nxt); s.meta = (int32_t*)(k.lds + l.meta);
  }
  return s;
}

// timing builds: cycles of a placement's segments, in the words behind the byte counter (O.bytes[2 + id]); read by tc_find
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU) && defined(KQ_TAS_CYCLE)   // (the cycle's request block has the room; the batch kernel's counter is one word)
#define TPROF0() long long _tp = clock64()
#define TPROF(k, id) do { const long long _tq = clock64(); if (lane_id() == 0) (k).O.bytes[2 + (id)] += _tq - _tp; _tp = _tq; } while (0)
#else
#define TPROF0() do {} while (0)
#define TPROF(k, id) do {} while (0)
#endif
struct TParams {  // topologyAssignmentParameters :473 + requirements :461
  int32_t count, leaderCount, sliceSize;
  int requestedLevelIdx, sliceLevelIdx;
  bool required, unconstrained, simulateEmpty, hasLeader, hasAssumed;
  const int64_t* req;        // [R] SinglePodRequests of the workers (pods added on the fly)
  const int64_t* leaderReq;  // [R] or NULL
  const uint8_t* leafOk;
  int leafLo, leafHi;   // the leaves below the required replacement domain (:1902): [leafLo, leafHi); leafHi <= 0 = every leaf
  // TASMultiLayerTopology: sliceSizeAtLevel :474 (0 = no entry), the constraint list for multiLayerNotFitMessage :2030
  int32_t sizeAt[KQ_TAS_MAX_LEVELS + 1];
  int nLayers;                // len(multiLayerConstraints) :482, 0 unless sliceSizeAtLevel has an entry
  const int32_t *layerLevel, *layerSize;
};

// sizeAt[l] with static indices only, so that the table stays in registers (a dynamically indexed member would put TParams into scratch)
KQ_DEV int32_t t_size_at(const TParams& p, int l) {
  int32_t v = 0;
  #pragma unroll
  for (int i = 0; i <= KQ_TAS_MAX_LEVELS; i++) if (i == l) v = p.sizeAt[i];
  return v;
}
KQ_DEV bool t_lfc(const TK& k, bool unconstrained) { return unconstrained && k.T.profile_mixed; }  // useLeastFreeCapacityAlgorithm :1468

// floor(a / b) as Go's int64 division gives it, for the operands CountIn meets: gfx950 has no 64-bit integer divide (the compiler
// expands one into a ~150-instruction routine, and phase 1 does twelve per leaf: 80 % of a placement's time, profiles/r03l_prof_tas_cycle.txt),
// but it has an IEEE fp64 divide. For 0 < a, b < 2^52 both operands are exact doubles, the correctly rounded quotient is within one
// of the true one, and the exact integer correction below makes it exact. Anything else takes the generic division.
KQ_DEV int64_t t_div(int64_t a, int64_t b) {
  if (a > 0 && b > 0 && a < ((int64_t)1 << 52) && b < ((int64_t)1 << 52)) {
    int64_t q = (int64_t)((double)a / (double)b);
    int64_t r = a - q * b;
    while (r < 0) { q--; r += b; }
    while (r >= b) { q++; r -= b; }
    return q;
  }
  return a / b;
}
// requests.go:195-232 on remaining capacity rem[] (registers of the lane)
KQ_DEV int32_t t_count_in(const TK& k, const int64_t* req, const int64_t* rem) {
  bool have = false;
  int32_t result = 0;
  for (int r = 0; r < k.T.R; r++) {
    int64_t q = req[r] + (r == k.T.pods ? 1 : 0);  // resources.OnePodRequest :905
    if (q == 0) continue;
    int64_t c = t_div(rem[r], q);
    int32_t cnt = (int32_t)i64max(0, i64min(c, 0x7fffffff));
    if (!have || cnt < result) { result = cnt; have = true; }
  }
  return have ? result : 0;
}

constexpr int KQ_TAS_MAXR = 32;   // resources of a TAS flavor's leaves (the reference's BenchmarkSchedulerTAS uses 30)

// Phase 1 over the leaves (fillLeafCounts :1899) as a job any wave of the workgroup can take a stripe of: the leader of k_process_tas
// posts it in LDS and its helper waves fill their stripes (t_leaf_helper); everywhere else the wave does it alone.
struct TLeafArgs {
  int32_t *pc, *sc, *pcwl, *scwl, *lc;
  const int64_t *assumed, *req, *leaderReq;
  const uint8_t* leafOk;
  int simulateEmpty, hasAssumed, sliceLevelIdx;
  int32_t sliceSize;
  int leafLo, leafHi;   // [leafLo, leafHi) of the leaves, leafHi <= 0 (what a shorter initializer list leaves) = every leaf
};
// one sweep over a slice of domains (t_view_first_fit): what it is after, and what a wave found in its share of the elements
struct TSweepArgs { int n, order, id0; bool lfc, by_order; int32_t needed, leaderCount; int which; bool store; };   // store: the keys of every element are written for the t_get calls that may follow
struct TSweepRes { uint64_t m0, m1, g, g0, g1; };   // slice order minimum; (count, slice order) minimum over the holders (g = ~0: none)
// k_process_tas, classical order: what the leader needs to start on the NEXT entry (its head, its nomination), fetched by helper wave 1
// while the leader finishes the current one. None of it changes while the kernel runs: an entry's nomination outputs are only rewritten
// by its own processEntry. (A dozen dependent global round trips per entry otherwise: 20 us between two entries at cfg 5.)
constexpr int KQ_TAS_PF_MAXU = 16, KQ_TAS_PF_R = 8;
struct TPre {
  int ready_for;   // iterator position the record was made for, -1: none
  int e, tree, cq, plen, ps_base, nps, slice_row, nuse, borrowing, nominated_mode, tgt_n, tgt_pos;
  int tree_ncq, tree_nn;   // ClusterQueues / nodes of the entry's tree (what a tree switch asks the snapshot for)
  // the static half of the request block of a find for the entry's ONLY podset on the cycle's ONLY TAS flavor (q_ok = 0: anything else)
  int q_ok, q_cls, q_cls_ok, q_level, q_ssize, q_slevel, q_group, q_kind;
  int64_t q_req[KQ_TAS_PF_R];
  uint32_t hflags, pol;
  int64_t prio, ts;
  int32_t path[KQ_MAXD], node_local[KQ_MAXD];
  int32_t use_fr[KQ_TAS_PF_MAXU];   // (an assignment with more usage entries is not prefetched)
  int64_t use_qty[KQ_TAS_PF_MAXU];
};
constexpr int KQ_TAS_TS_TREES = 16;   // processEntry's per-tree state of that many trees fits the job block (LDS); more: global memory
struct TLeafJob {
  TTopo T;
  TLeafArgs a;
  int cmd;            // 0 idle, 1 phase-1 job posted, 2 quit, 3 copy job posted (a.pc / a.sc = the class table's rows, cp_* = the LDS arrays),
                      // 6 wave 1 fetches the header of entry pf_next, 7 wave 2 patches the class tables (one barrier each, nobody waits for them),
                      // 8 the helper waves copy a class table into LDS and wait for the leader at the second barrier
  int nw;             // waves of the workgroup sharing the job (set once by the kernel that owns the helpers)
  int coop_min;       // slices at least this long are swept by every wave of the workgroup (two barriers: ~1 us; set once, 1024 unless a test says otherwise)
  long long bytes;    // helpers add their share
  int32_t *cp_pc, *cp_sc;
  int cp_n;
  // cmd 4: a fused sweep (t_sweep_part) / cmd 5: the arg-min over the keys not yet produced (t_get) — of a state in LDS, shared by the waves
  TState sw_s;
  TSweepArgs sw_a;
  TSweepRes sw_r[8];
  int ar_n, ar_started, ar_skip;
  uint64_t ar_c0, ar_c1, ar_m0[8], ar_m1[8];
  const void* pf_k;   // the kernel's K block
  int pf_next;        // iterator position to fetch next, -1: none
  TPre pre[2];        // [position & 1]
  int cu_ps_base, cu_nps, cu_lds_on, cu_lds_bytes;   // cmd 7: the podsets whose TopologyAssignments were just added to the work plane
  int32_t tstate[KQ_TAS_TS_TREES * 12];              // TCyc::tree_state of a cycle with few trees
  // cmd 8: the copy job of the NEXT placement posted split-phase at the start of a recomputation — the helper waves copy the class's rows
  // while the leader runs the flavor assignment, and wait at the job's second barrier; the leader joins it when the placement starts
  // (t_class_to_lds), or before it posts anything else (t_post_begin). early_cls: the class whose rows the LDS arrays hold (-1: none)
  int early_pending, early_cls;
};
// A class's phase-1 rows (global memory, patched by L2 atomics: agent-scope loads, so that no stale line of this CU's vector cache is
// read) into the LDS working copy; thread tid of nthreads. Eight 8-byte loads per array in flight per thread: 4168 domains are one round
// trip for 256 threads.
KQ_DEV void t_class_copy(const int32_t* spc, const int32_t* ssc, int32_t* dpc, int32_t* dsc, int n, int tid, int nthreads) {
  const int32_t* src[2] = {spc, ssc};
  int32_t* dst[2] = {dpc, dsc};
  #pragma unroll
  for (int a = 0; a < 2; a++) {
    const int h = (int)(((uintptr_t)src[a] >> 2) & 1);   // one leading element up to 8-byte alignment
    const int n2 = n > h ? (n - h) / 2 : 0;
    if (tid == 0 && h && n > 0) dst[a][0] = (int32_t)ag_load_u32((const uint32_t*)src[a]);
    if (tid == 0 && n > h && ((n - h) & 1)) dst[a][n - 1] = (int32_t)ag_load_u32((const uint32_t*)src[a] + n - 1);
    const uint64_t* s2 = (const uint64_t*)(src[a] + h);
    constexpr int U = 8;
    for (int b = tid; b < n2; b += nthreads * U) {
      uint64_t v[U];
      #pragma unroll
      for (int q = 0; q < U; q++) { const int i = b + q * nthreads; v[q] = i < n2 ? ag_load_u64(s2 + i) : 0; }
      #pragma unroll
      for (int q = 0; q < U; q++) {
        const int i = b + q * nthreads;
        if (i < n2) { dst[a][h + 2 * i] = (int32_t)(uint32_t)v[q]; dst[a][h + 2 * i + 1] = (int32_t)(uint32_t)(v[q] >> 32); }
      }
    }
  }
}
// leaves first, first + stride, ...: U leaves per step so that the next one's rows are in flight while the first one divides.
// RM = 4 keeps the per-pod requests (incl. the pod itself; 0 = not requested) in registers; the general case reads them per use.
template <int RM, int U> KQ_DEV int64_t t_leaf_counts(const TTopo& T, const TLeafArgs& a, int first, int stride) {
  int64_t lb = 0;
  const bool at = T.L - 1 == a.sliceLevelIdx;
  const bool lead = a.leaderReq != nullptr;
  auto reqv = [&](const int64_t* rq, int r) -> int64_t { return r < T.R ? rq[r] + (r == T.pods ? 1 : 0) : 0; };
  // CountIn (requests.go:195-232): the minimum over the requested resources of floor(remaining / request), 0 when nothing is requested
  auto count = [&](const int64_t* rq, const int64_t (&rem)[RM]) -> int32_t {
    bool have = false;
    int32_t result = 0;
    #pragma unroll
    for (int r = 0; r < RM; r++) {
      const int64_t q = reqv(rq, r);
      if (q == 0) continue;
      const int32_t cnt = (int32_t)i64max(0, i64min(t_div(rem[r], q), 0x7fffffff));
      if (!have || cnt < result) { result = cnt; have = true; }
    }
    return have ? result : 0;
  };
  for (int leaf0 = first; leaf0 < T.n_leaves; leaf0 += U * stride) {
    int64_t rem[U][RM];
    bool ok[U];
    #pragma unroll
    for (int u = 0; u < U; u++) {
      const int leaf = leaf0 + u * stride;
      ok[u] = leaf < T.n_leaves && (a.leafHi <= 0 || (leaf >= a.leafLo && leaf < a.leafHi)) && (!a.leafOk || a.leafOk[leaf]);
      #pragma unroll
      for (int r = 0; r < RM; r++) {
        rem[u][r] = 0;
        if (r < T.R && ok[u]) {
          int64_t v = T.free_cap[(size_t)leaf * T.R + r];
          if (!a.simulateEmpty) v -= T.tas_usage[(size_t)leaf * T.R + r];  // remainingCapacityForLeaf :1884
          if (a.hasAssumed) v -= a.assumed[(size_t)leaf * T.R + r];
          rem[u][r] = v;
        }
      }
    }
    #pragma unroll
    for (int u = 0; u < U; u++) {
      const int leaf = leaf0 + u * stride;
      if (leaf >= T.n_leaves) continue;
      const int d = T.leaf_base + leaf;
      int32_t pc = 0, pcwl = 0, lc = 0;
      if (ok[u]) {
        pc = count(a.req, rem[u]);
        if (lead && count(a.leaderReq, rem[u]) > 0) {
          lc = 1;
          #pragma unroll
          for (int r = 0; r < RM; r++) rem[u][r] -= reqv(a.leaderReq, r);
        }
        pcwl = lc ? count(a.req, rem[u]) : pc;   // without a leader on the leaf the capacity is what it was
        lb += (int64_t)T.R * 16 + 24;
      }
      a.pc[d] = pc; a.pcwl[d] = pcwl; if (a.lc) a.lc[d] = lc;   // (no leaderCount array: a state without leaders, TState::nolead)
      a.sc[d] = at ? pc / a.sliceSize : 0;
      a.scwl[d] = at ? pcwl / a.sliceSize : 0;
    }
  }
  return lb;
}
KQ_DEV int64_t t_leaf_counts_any(const TTopo& T, const TLeafArgs& a, int first, int stride) {
  if (T.R <= 4) return t_leaf_counts<4, 2>(T, a, first, stride);
  return T.R <= 16 ? t_leaf_counts<16, 1>(T, a, first, stride) : t_leaf_counts<KQ_TAS_MAXR, 1>(T, a, first, stride);
}
// helper waves of a workgroup whose wave 0 posts phase-1 jobs (k_process_tas): wave `wv` of `nw`
KQ_DEV void t_sweep_help(TLeafJob& job, int wv, int nw);
KQ_DEV void t_argmin_help(TLeafJob& job, int wv, int nw);
// before the leader posts a job: a split-phase copy still waiting at its second barrier is joined first
KQ_DEV void t_post_begin(TLeafJob& j) {
  if (j.early_pending) {
    bsync();
    if (lane_id() == 0) j.early_pending = 0;
    wsync();
  }
}
// one posted job, wave wv's share of it
KQ_DEV void t_helper_step(TLeafJob& job, int wv, int nw) {
  if (job.cmd == 8) {
    if (nw > 1) t_class_copy(job.a.pc, job.a.sc, job.cp_pc, job.cp_sc, job.cp_n, (wv - 1) * WAVE + lane_id(), (nw - 1) * WAVE);
  } else if (job.cmd == 3) {
    t_class_copy(job.a.pc, job.a.sc, job.cp_pc, job.cp_sc, job.cp_n, wv * WAVE + lane_id(), nw * WAVE);
  } else if (job.cmd == 4) {
    t_sweep_help(job, wv, nw);
  } else if (job.cmd == 5) {
    t_argmin_help(job, wv, nw);
  } else {
    const int64_t lb = wsum_i64(t_leaf_counts_any(job.T, job.a, wv * WAVE + lane_id(), nw * WAVE));
    if (lane_id() == 0 && lb) atomic_add_i64(&job.bytes, (long long)lb);
  }
}
KQ_DEV void t_prefetch_entry(TLeafJob& job);   // kq_tas_cycle.hpp
KQ_DEV void t_class_update_job(TLeafJob& job);
KQ_DEV void t_leaf_helper(TLeafJob& job, int wv, int nw) {
  for (;;) {
    bsync();
    if (job.cmd == 2) break;
    if (job.cmd == 6) {
#ifdef KQ_TAS_CYCLE
      if (wv == 1) t_prefetch_entry(job);
#endif
      continue;
    }
    if (job.cmd == 7) {
#ifdef KQ_TAS_CYCLE
      if (wv == 2) t_class_update_job(job);
#endif
      continue;
    }
    t_helper_step(job, wv, nw);
    bsync();
  }
}
// the 1-lane emulation has no concurrent waves: the leader plays the helpers' shares itself, right after posting a job
#ifdef KQ_HOST_EMU
#define KQ_TAS_EMU_HELPERS(j) do { for (int _wv = 1; _wv < (j).nw; _wv++) t_helper_step((j), _wv, (j).nw); } while (0)
#else
#define KQ_TAS_EMU_HELPERS(j) do {} while (0)
#endif
// the leader's side of a copy job: with helper waves the rows are split over the workgroup, alone the wave copies them itself
KQ_DEV void t_class_to_lds(const TK& k, const TState& s, int cls) {
  const TTopo& T = k.T;
  const int32_t* spc = k.C.pc + (size_t)cls * T.D; const int32_t* ssc = k.C.sc + (size_t)cls * T.D;
  if (k.mail) {
    TLeafJob& j = *k.mail;
    if (j.early_pending && j.early_cls == cls && j.cp_pc == s.pc && j.a.pc == spc) {   // (same class AND same table: the empty-cluster tables share the class numbers)
      // the rows came while the flavor assignment ran: the helper waves are at the job's second barrier
      bsync();
      if (lane_id() == 0) { j.early_pending = 0; j.early_cls = -1; }   // (the placement consumes the copy)
      wsync();
      return;
    }
    t_post_begin(j);
    if (lane_id() == 0) { j.early_cls = -1; j.a.pc = (int32_t*)spc; j.a.sc = (int32_t*)ssc; j.cp_pc = s.pc; j.cp_sc = s.sc; j.cp_n = T.D; j.cmd = 3; }
    bsync();
    KQ_TAS_EMU_HELPERS(j);
    t_class_copy(spc, ssc, s.pc, s.sc, T.D, lane_id(), j.nw * WAVE);
    bsync();
  } else {
    t_class_copy(spc, ssc, s.pc, s.sc, T.D, lane_id(), WAVE);
    wsync();
  }
}

// fillInCounts :1800 = fillLeafCounts for every feasible leaf + fillInCountsHelper roll-up
KQ_DEV void t_fill_in_counts(const TK& k, const TState& s, const TParams& p, long long* bytes_out) {
  const TTopo& T = k.T;
  const int lane = lane_id();
  for (int d = lane; d < T.leaf_base; d += WAVE) { s.pc[d] = 0; s.sc[d] = 0; s.pcwl[d] = 0; s.scwl[d] = 0; t_set_lc(s, d, 0); }
  int64_t lb = 0;
  {
    const TLeafArgs a{s.pc, s.sc, s.pcwl, s.scwl, s.lc, s.assumed, p.req, p.leaderReq, p.leafOk, p.simulateEmpty ? 1 : 0, p.hasAssumed ? 1 : 0, p.sliceLevelIdx, p.sliceSize, p.leafLo, p.leafHi};
    if (k.mail) {
      TLeafJob& j = *k.mail;
      t_post_begin(j);
      if (lane == 0) { j.T = T; j.a = a; j.bytes = 0; j.cmd = 1; }
      bsync();
      KQ_TAS_EMU_HELPERS(j);
      lb = t_leaf_counts_any(j.T, j.a, lane, j.nw * WAVE);
      bsync();
      if (lane == 0) lb += j.bytes;
    } else {
      lb = t_leaf_counts_any(T, a, lane, WAVE);
    }
  }
  wsync();
  const bool leaderRequired = p.leaderCount > 0;
  for (int level = T.L - 2; level >= 0; level--) {
    for (int d = T.level_off[level] + lane; d < T.level_off[level + 1]; d += WAVE) {
      int32_t childrenCapacity = 0, sliceCapacity = 0, minPodDiff = 0x7fffffff, minSliceDiff = 0x7fffffff, leaderCount = 0;
      bool contributor = false;
      const int c0 = T.child_first[d], cn = T.child_cnt[d];
      const int32_t innerSize = t_size_at(p, level + 1);  // a child at a constrained level only contributes whole inner slices (:1950-1967)
      for (int c = c0; c < c0 + cn; c++) {
        int32_t cpc = s.pc[c], cpcwl = s.pcwl[c];
        const int32_t csc = s.sc[c], cscwl = s.scwl[c], clc = t_lc(s, c);
        if (innerSize > 0) { cpc = (cpc / innerSize) * innerSize; cpcwl = (cpcwl / innerSize) * innerSize; }
        childrenCapacity += cpc;
        sliceCapacity += csc;
        if (!leaderRequired || clc > 0) {
          contributor = true;
          if (cpc - cpcwl < minPodDiff) minPodDiff = cpc - cpcwl;
          if (csc - cscwl < minSliceDiff) minSliceDiff = csc - cscwl;
        }
        if (clc > leaderCount) leaderCount = clc;
      }
      int32_t pcwl = 0, scwl = 0;
      if (contributor) { pcwl = childrenCapacity - minPodDiff; scwl = sliceCapacity - minSliceDiff; }
      if (level == p.sliceLevelIdx) { sliceCapacity = childrenCapacity / p.sliceSize; scwl = pcwl / p.sliceSize; }
      s.pc[d] = childrenCapacity; s.pcwl[d] = pcwl; t_set_lc(s, d, leaderCount); s.sc[d] = sliceCapacity; s.scwl[d] = scwl;
      lb += 24;
    }
    wsync();
  }
  const int64_t tot = wsum_i64(lb);
  if (lane == 0) atomic_add_i64(bytes_out, (long long)(tot + (int64_t)T.R * 8));
}

// ---- lazily sorted slice of domains -------------------------------------------------------------------
// A slice of domains in the order the reference would have sorted it (sortedDomains :1770, sortedDomainsWithLeader
// :1731) or, for currFitDomain, in the order it was built. The order is a 128-bit key per element, ascending.
enum { ORD_LIST = 0, ORD_PLAIN = 1, ORD_LEADER = 2 };
constexpr uint64_t KQ_TAS_EXCLUDED = 1ull << 63;  // bit 63 of k0 is free: the leaderCount field is < 2^31
struct TView {
  int n;                 // elements of s.set
  int order; bool lfc;
  int mat;               // materialised prefix length (s.arr)
  uint64_t c0, c1;       // key of the last element taken from the stream
  bool started;
  int skip;              // element moved to the front by prioritizeLeaderDomain (it sits in arr already), -1 none
  bool virt;             // no stored keys: every look builds an element's key from the state (a slice nothing changes while it is walked)
};
KQ_DEV int t_dom(uint64_t k1) { return (int)(uint32_t)(k1 & 0xffffffffu); }
KQ_DEV void t_key_of(const TState& s, int d, int pos, int order, bool lfc, uint64_t* k0, uint64_t* k1) {
  if (order == ORD_LIST) { *k0 = 0; *k1 = ((uint64_t)(uint32_t)pos << 32) | (uint32_t)d; return; }
  const bool wl = order == ORD_LEADER;
  const uint32_t lcv = wl ? (uint32_t)t_lc(s, d) : 0u;
  const uint32_t scv = (uint32_t)(wl ? s.scwl[d] : s.sc[d]);
  const uint32_t pcv = (uint32_t)(wl ? s.pcwl[d] : s.pc[d]);
  // leaderCount descending, sliceCount descending (BestFit) / ascending (LeastFreeCapacity), podCount ascending, levelValues
  *k0 = ((uint64_t)(0x7fffffffu - lcv) << 32) | (uint64_t)(lfc ? scv : 0x7fffffffu - scv);
  *k1 = ((uint64_t)pcv << 32) | (uint32_t)d;
}
KQ_DEV bool t_key_lt(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1) { return a0 < b0 || (a0 == b0 && a1 < b1); }
KQ_DEV bool t_all_zero(const TState& s, int d) { return (s.pc[d] | s.sc[d] | s.pcwl[d] | s.scwl[d] | t_lc(s, d)) == 0; }

// s.set[0..n) must hold the slice. In the sorted orders, domains with an all-zero state are dropped: in every loop of
// phase 2 they neither change a remaining count nor survive into the assignment (buildTopologyAssignmentForLevels :1690
// drops podCount == 0), and they can never win a best-fit scan.
KQ_DEV TView t_view(const TK& k, const TState& s, int n, int order, bool unconstrained) {
  TView v;
  v.n = n; v.order = order; v.lfc = t_lfc(k, unconstrained); v.mat = 0; v.c0 = 0; v.c1 = 0; v.started = false; v.skip = -1; v.virt = false;
  for (int i = lane_id(); i < n; i += WAVE) {
    const int d = s.set[i];
    uint64_t a, b;
    t_key_of(s, d, i, order, v.lfc, &a, &b);
    if (order != ORD_LIST && t_all_zero(s, d)) a |= KQ_TAS_EXCLUDED;  // keeps its place in the order, never produced
    s.k0[i] = a; s.k1[i] = b;
  }
  wsync();
  return v;
}
// wave arg-min over the elements accepted by `sel`, ranked by sel.rank (default: the slice order: the winner is
// t_dom(*o1)); false if no element is accepted
template <class SEL> KQ_DEV void t_argmin_part(const TState& s, int n, const SEL& sel, bool with_excluded, int tid, int nth, uint64_t* m0o, uint64_t* m1o,
                                               bool virt = false, int order = ORD_PLAIN, bool lfc = false) {
  uint64_t b0 = ~0ull, b1 = ~0ull;
  // a lone wave is latency-bound: fetch the keys of UNR strided elements before looking at any of them
  constexpr int UNR = 8;
  for (int base = tid; base < n; base += nth * UNR) {
    uint64_t k0v[UNR], k1v[UNR];
    if (virt) {   // the keys t_view would have stored, from the (unchanged) state
      int dv[UNR];
      #pragma unroll
      for (int q = 0; q < UNR; q++) { const int i = base + q * nth; dv[q] = i < n ? s.set[i] : -1; }
      #pragma unroll
      for (int q = 0; q < UNR; q++) {
        k0v[q] = KQ_TAS_EXCLUDED; k1v[q] = 0;
        if (dv[q] < 0) continue;
        uint64_t a, b;
        t_key_of(s, dv[q], base + q * nth, order, lfc, &a, &b);
        if (order != ORD_LIST && t_all_zero(s, dv[q])) a |= KQ_TAS_EXCLUDED;
        k0v[q] = a; k1v[q] = b;
      }
    } else {
    #pragma unroll
    for (int q = 0; q < UNR; q++) {
      const int i = base + q * nth;
      k0v[q] = i < n ? s.k0[i] : KQ_TAS_EXCLUDED;
      k1v[q] = i < n ? s.k1[i] : 0;
    }
    }
    #pragma unroll
    for (int q = 0; q < UNR; q++) {
      if (base + q * nth >= n) continue;
      uint64_t a0 = k0v[q];
      const uint64_t a1 = k1v[q];
      if (a0 & KQ_TAS_EXCLUDED) { if (!with_excluded) continue; a0 &= ~KQ_TAS_EXCLUDED; }
      if (!sel.take(a0, a1)) continue;
      uint64_t x0 = a0, x1 = a1;
      sel.rank(&x0, &x1);
      if (t_key_lt(x0, x1, b0, b1)) { b0 = x0; b1 = x1; }
    }
  }
  const uint64_t m0 = wmin_u64(b0);
  *m0o = m0;
  *m1o = wmin_u64(b0 == m0 ? b1 : ~0ull);
}
template <class SEL> KQ_DEV bool t_argmin(const TState& s, const TView& v, const SEL& sel, uint64_t* o0, uint64_t* o1, bool with_excluded = false) {
  uint64_t m0, m1;
  t_argmin_part(s, v.n, sel, with_excluded, lane_id(), WAVE, &m0, &m1, v.virt, v.order, v.lfc);
  if (m0 == ~0ull && m1 == ~0ull) return false;
  *o0 = m0; *o1 = m1;
  return true;
}
// elements of the slice that have not been produced yet
struct SelRest {
  bool started; uint64_t c0, c1; int skip;
  KQ_MDEV bool rest(uint64_t a0, uint64_t a1) const { return t_dom(a1) != skip && (!started || t_key_lt(c0, c1, a0, a1)); }
  KQ_MDEV bool take(uint64_t a0, uint64_t a1) const { return rest(a0, a1); }
  KQ_MDEV void rank(uint64_t*, uint64_t*) const {}
};
// domains[i]: materialise the slice up to index i; -1 past the end
KQ_DEV bool t_argmin_rest_coop(const TState& s, const TView& v, const SelRest& sel, uint64_t* o0, uint64_t* o1);
KQ_DEV int t_get(const TState& s, TView& v, int i) {
  while (v.mat <= i) {
    SelRest sel{v.started, v.c0, v.c1, v.skip};
    uint64_t o0 = 0, o1 = 0;
    if (!(s.coop && v.n >= s.coop_min ? t_argmin_rest_coop(s, v, sel, &o0, &o1) : t_argmin(s, v, sel, &o0, &o1))) return -1;
    const int d = t_dom(o1);
    v.started = true; v.c0 = o0; v.c1 = o1;
    if (lane_id() == 0) s.arr[v.mat] = d;
    v.mat++;
    wsync();
  }
  return s.arr[i];
}
KQ_DEV int32_t t_count_of(const TState& s, int d, int which) {  // 0 podCount, 1 podCountWithLeader, 2 sliceCount, 3 sliceCountWithLeader
  return which == 0 ? s.pc[d] : which == 1 ? s.pcwl[d] : which == 2 ? s.sc[d] : s.scwl[d];
}
// findBestFitDomainBy :1307 over domains[from:] : the first (slice order) domain with the smallest count >= needed
struct SelFitCount {  // rank = the count itself
  const TState* s; SelRest r; int32_t needed, leaderCount, below; int which;
  KQ_MDEV bool take(uint64_t a0, uint64_t a1) const {
    if (!r.rest(a0, a1)) return false;
    const int d = t_dom(a1);
    if (t_lc(*s, d) < leaderCount) return false;
    const int32_t c = t_count_of(*s, d, which);
    return c >= needed && c < below;
  }
  KQ_MDEV void rank(uint64_t* x0, uint64_t* x1) const { *x0 = (uint64_t)(uint32_t)t_count_of(*s, t_dom(*x1), which); *x1 &= 0xffffffffull; }
};
struct SelFitEq {  // the first element, in slice order, having exactly `want`
  const TState* s; SelRest r; int32_t want, leaderCount; int which;
  KQ_MDEV bool take(uint64_t a0, uint64_t a1) const {
    if (!r.rest(a0, a1)) return false;
    const int d = t_dom(a1);
    return t_lc(*s, d) >= leaderCount && t_count_of(*s, d, which) == want;
  }
  KQ_MDEV void rank(uint64_t*, uint64_t*) const {}
};
KQ_DEV int t_best_fit(const TState& s, TView& v, int from, int32_t needed, int which, int32_t leaderCount) {
  const int first = t_get(s, v, from);
  if (first < 0) return -1;
  // the materialised part arr[from..mat) is short: walk it in order, then reduce over what the stream still holds
  int best = -1; int32_t bestCount = 0x7fffffff;
  for (int i = from; i < v.mat; i++) {
    const int d = s.arr[i];
    if (t_lc(s, d) < leaderCount) continue;
    const int32_t c = t_count_of(s, d, which);
    if (c >= needed && c < bestCount) { best = d; bestCount = c; }
  }
  const SelRest rest{v.started, v.c0, v.c1, v.skip};
  SelFitCount sc{&s, rest, needed, leaderCount, bestCount, which};
  uint64_t m0 = 0, m1 = 0;
  if (t_argmin(s, v, sc, &m0, &m1)) {
    SelFitEq se{&s, rest, (int32_t)m0, leaderCount, which};
    uint64_t e0 = 0, e1 = 0;
    if (t_argmin(s, v, se, &e0, &e1)) best = t_dom(e1);  // strictly smaller count than anything in the materialised part
  }
  return best >= 0 ? best : first;
}
KQ_DEV int t_best_fit_pods(const TState& s, TView& v, int from, int32_t count, int32_t leaderCount) {  // findBestFitDomain :1276
  return t_best_fit(s, v, from, count, leaderCount > 0 ? 1 : 0, leaderCount);
}
KQ_DEV int t_best_fit_slices(const TState& s, TView& v, int from, int32_t sliceCount, int32_t leaderCount) {  // :1293
  return t_best_fit(s, v, from, sliceCount, leaderCount > 0 ? 3 : 2, leaderCount);
}

// prioritizeLeaderDomain :1533 — the first leader-capable domain whose leader penalty fits in the slack moves to the front
// t_view + domains[0] + the best-fit scan from position 0 in ONE sweep over the slice. A lone wave pays a fixed latency per sweep (a
// round trip to its scratch rows, two or three wave reductions, a fence) whatever the slice's length, and a placement used to make
// four of them per level: the keys, the arg-min, the smallest count >= needed, its first holder (profiles/r03n_prof_tas_cycle.txt:
// 52 us of phase 2 per recomputation, mostly that latency). Here every element's keys are built and stored for the t_get calls that
// may follow, and two minima are kept on the way: the slice order (= domains[0]) and (count, slice order) over the elements that hold
// `needed` with `leaderCount` leaders (= findBestFitDomainBy :1307 over domains[0:]).
//   *first  domains[0], -1 when the slice has nothing but all-zero domains
//   *fit    the best-fit domain among the holders, -1 when nothing holds `needed`
// The view comes back with domains[0] materialised, exactly as after t_view + t_get(.., 0).
// id0 >= 0: the slice is the contiguous id range [id0, id0 + n) (a whole level, the children of one domain) — s.set holds the same
// values but is not read here, so the caller needs no fence between writing it and this sweep.
// elements tid, tid + nth, ... of the slice; the result is the wave's (every lane holds it)
KQ_DEV TSweepRes t_sweep_part(const TState& s, const TSweepArgs& a, int tid, int nth) {
  const int n = a.n, order = a.order, id0 = a.id0, which = a.which;
  uint64_t b0 = ~0ull, b1 = ~0ull;               // slice order
  uint64_t f0 = ~0ull, f1 = ~0ull, fc = ~0ull;   // (count, slice order) over the holders
  // a lone wave is latency-bound: the rows of UNR strided elements are in flight before any of them is looked at
  constexpr int UNR = 8;
  for (int base = tid; base < n; base += nth * UNR) {
    int dv[UNR]; int32_t pcv[UNR], scv[UNR], pwv[UNR], swv[UNR], lcv[UNR];
    #pragma unroll
    for (int q = 0; q < UNR; q++) { const int i = base + q * nth; dv[q] = i < n ? (id0 >= 0 ? id0 + i : s.set[i]) : -1; }
    #pragma unroll
    for (int q = 0; q < UNR; q++) {
      const int d = dv[q];
      if (d < 0) { pcv[q] = scv[q] = pwv[q] = swv[q] = lcv[q] = 0; continue; }
      pcv[q] = s.pc[d]; scv[q] = s.sc[d];
      if (s.nolead) { pwv[q] = pcv[q]; swv[q] = scv[q]; lcv[q] = 0; }   // (the ...WithLeader arrays are the same arrays)
      else { pwv[q] = s.pcwl[d]; swv[q] = s.scwl[d]; lcv[q] = s.lc[d]; }
    }
    #pragma unroll
    for (int q = 0; q < UNR; q++) {
      const int i = base + q * nth, d = dv[q];
      if (d < 0) continue;
      uint64_t ka, kb;
      if (order == ORD_LIST) { ka = 0; kb = ((uint64_t)(uint32_t)i << 32) | (uint32_t)d; }
      else {
        const bool wl = order == ORD_LEADER;
        const uint32_t l = wl ? (uint32_t)lcv[q] : 0u, sc2 = (uint32_t)(wl ? swv[q] : scv[q]), pc2 = (uint32_t)(wl ? pwv[q] : pcv[q]);
        ka = ((uint64_t)(0x7fffffffu - l) << 32) | (uint64_t)(a.lfc ? sc2 : 0x7fffffffu - sc2);
        kb = ((uint64_t)pc2 << 32) | (uint32_t)d;
      }
      const bool zero = order != ORD_LIST && (pcv[q] | scv[q] | pwv[q] | swv[q] | lcv[q]) == 0;
      if (a.store) { s.k0[i] = zero ? (ka | KQ_TAS_EXCLUDED) : ka; s.k1[i] = kb; }
      if (zero) continue;
      if (t_key_lt(ka, kb, b0, b1)) { b0 = ka; b1 = kb; }
      const int32_t c = which == 0 ? pcv[q] : which == 1 ? pwv[q] : which == 2 ? scv[q] : swv[q];
      if (lcv[q] >= a.leaderCount && c >= a.needed) {
        const uint64_t cu = a.by_order ? 0 : (uint64_t)(uint32_t)c;
        if (cu < fc || (cu == fc && t_key_lt(ka, kb, f0, f1))) { fc = cu; f0 = ka; f1 = kb; }
      }
    }
  }
  TSweepRes r;
  r.m0 = wmin_u64(b0);
  r.m1 = wmin_u64(b0 == r.m0 ? b1 : ~0ull);
  r.g = wmin_u64(fc);
  r.g0 = ~0ull; r.g1 = ~0ull;
  if (r.g != ~0ull) {
    r.g0 = wmin_u64(fc == r.g ? f0 : ~0ull);
    r.g1 = wmin_u64(fc == r.g && f0 == r.g0 ? f1 : ~0ull);
  }
  return r;
}
KQ_DEV void t_sweep_merge(TSweepRes& r, const TSweepRes& o) {
  if (t_key_lt(o.m0, o.m1, r.m0, r.m1)) { r.m0 = o.m0; r.m1 = o.m1; }
  if (o.g < r.g || (o.g == r.g && t_key_lt(o.g0, o.g1, r.g0, r.g1))) { r.g = o.g; r.g0 = o.g0; r.g1 = o.g1; }
}
KQ_DEV TSweepRes t_sweep_coop(const TState& s, const TSweepArgs& a);
// by_order: the holders are ranked by the slice order alone — LeastFreeCapacity's "first domain, ascending, that holds everything"
// (:1364-1376) instead of BestFit's smallest count.
KQ_DEV TView t_view_first_fit(const TK& k, const TState& s, int n, int order, bool unconstrained, int32_t needed, int which, int32_t leaderCount,
                              int* first, int* fit, int id0 = -1, bool by_order = false, bool store = true) {
  TView v;
  v.n = n; v.order = order; v.lfc = t_lfc(k, unconstrained); v.mat = 0; v.c0 = 0; v.c1 = 0; v.started = false; v.skip = -1; v.virt = false;
  const TSweepArgs a{n, order, id0, v.lfc, by_order, needed, leaderCount, which, store};
  // a long slice of a state in LDS is shared with the workgroup's helper waves (k_process_tas): each wave a quarter of the elements
  const TSweepRes r = (s.coop && n >= s.coop_min && id0 >= 0) ? t_sweep_coop(s, a) : t_sweep_part(s, a, lane_id(), WAVE);
  *first = -1; *fit = -1;
  if (r.g != ~0ull) *fit = t_dom(r.g1);
  if (!(r.m0 == ~0ull && r.m1 == ~0ull)) {
    *first = t_dom(r.m1);
    v.started = true; v.c0 = r.m0; v.c1 = r.m1; v.mat = 1;
    if (lane_id() == 0) s.arr[0] = *first;
  }
  wsync();
  return v;
}
// ---- long slices of a state in LDS, shared with the helper waves (k_process_tas) ----
KQ_DEV void t_sweep_help(TLeafJob& job, int wv, int nw) {
  const TState s = job.sw_s;
  t_assume_lds(s);
  const TSweepRes r = t_sweep_part(s, job.sw_a, wv * WAVE + lane_id(), nw * WAVE);
  if (lane_id() == 0) job.sw_r[wv] = r;
}
KQ_DEV TSweepRes t_sweep_coop(const TState& s, const TSweepArgs& a) {
  TLeafJob& j = *s.coop;
  t_post_begin(j);
  if (lane_id() == 0) { j.sw_s = s; j.sw_a = a; j.cmd = 4; }
  bsync();
  KQ_TAS_EMU_HELPERS(j);
  TSweepRes r = t_sweep_part(s, a, lane_id(), j.nw * WAVE);
  bsync();
  for (int wv = 1; wv < j.nw; wv++) t_sweep_merge(r, j.sw_r[wv]);
  return r;
}
KQ_DEV void t_argmin_help(TLeafJob& job, int wv, int nw) {
  const SelRest sel{job.ar_started != 0, job.ar_c0, job.ar_c1, job.ar_skip};
  uint64_t m0, m1;
  const TState s = job.sw_s;
  t_assume_lds(s);
  t_argmin_part(s, job.ar_n, sel, false, wv * WAVE + lane_id(), nw * WAVE, &m0, &m1);
  if (lane_id() == 0) { job.ar_m0[wv] = m0; job.ar_m1[wv] = m1; }
}
KQ_DEV bool t_argmin_rest_coop(const TState& s, const TView& v, const SelRest& sel, uint64_t* o0, uint64_t* o1) {
  TLeafJob& j = *s.coop;
  t_post_begin(j);
  if (lane_id() == 0) { j.sw_s = s; j.ar_n = v.n; j.ar_started = sel.started ? 1 : 0; j.ar_c0 = sel.c0; j.ar_c1 = sel.c1; j.ar_skip = sel.skip; j.cmd = 5; }
  bsync();
  KQ_TAS_EMU_HELPERS(j);
  uint64_t m0, m1;
  t_argmin_part(s, v.n, sel, false, lane_id(), j.nw * WAVE, &m0, &m1);
  bsync();
  for (int wv = 1; wv < j.nw; wv++) if (t_key_lt(j.ar_m0[wv], j.ar_m1[wv], m0, m1)) { m0 = j.ar_m0[wv]; m1 = j.ar_m1[wv]; }
  if (m0 == ~0ull && m1 == ~0ull) return false;
  *o0 = m0; *o1 = m1;
  return true;
}
struct SelLeaderElig {
  const TState* s; int32_t leaderCount, availableCapacity, requiredCapacity; bool slices;
  KQ_MDEV bool take(uint64_t, uint64_t a1) const {
    const int d = t_dom(a1);
    if (t_lc(*s, d) < leaderCount) return false;
    const int32_t pen = slices ? s->sc[d] - s->scwl[d] : s->pc[d] - s->pcwl[d];
    return availableCapacity - pen >= requiredCapacity;
  }
  KQ_MDEV void rank(uint64_t*, uint64_t*) const {}
};
KQ_DEV void t_prioritize_leader(const TK& k, const TState& s, TView& v, int32_t count, int32_t leaderCount, int32_t sliceSize, bool slicesEnabled) {
  (void)k;
  if (leaderCount == 0) return;
  int64_t avail = 0;  // all-zero domains add nothing; "len(domains) < 2" cannot reorder anything either way
  for (int i = lane_id(); i < v.n; i += WAVE) {
    if (s.k0[i] & KQ_TAS_EXCLUDED) continue;
    const int d = s.set[i];
    avail += slicesEnabled ? s.sc[d] : s.pc[d];
  }
  const int32_t availableCapacity = (int32_t)wsum_i64(avail);
  const int32_t requiredCapacity = slicesEnabled ? count / sliceSize : count;
  SelLeaderElig se{&s, leaderCount, availableCapacity, requiredCapacity, slicesEnabled};
  uint64_t f0 = 0, f1 = 0;
  if (!t_argmin(s, v, se, &f0, &f1)) return;
  const int f = t_dom(f1);
  const int head = t_get(s, v, 0);
  if (head == f) return;
  const bool inside = v.started && !t_key_lt(v.c0, v.c1, f0, f1);  // f already sits in the materialised prefix
  wsync();
  if (lane_id() == 0) {
    int pos = v.mat;
    if (inside) { pos = 0; while (s.arr[pos] != f) pos++; }
    for (int i = pos; i >= 1; i--) s.arr[i] = s.arr[i - 1];
    s.arr[0] = f;
  }
  if (!inside) { v.mat++; v.skip = f; }
  wsync();
}

// consumeWithLeadersGeneric :1486 ; *domain = domains[i]; returns completed
KQ_DEV bool t_consume_with_leaders(const TK& k, const TState& s, TView& v, int i, int* domain, int32_t* remainingPrimary, int32_t* remainingLeaderCount,
                                   bool unconstrained, int32_t sliceSize, bool slices) {
  int d = *domain;
  int32_t* withLeader = slices ? s.scwl : s.pcwl;
  int32_t* primary = slices ? s.sc : s.pc;
  if (!t_lfc(k, unconstrained) && withLeader[d] >= *remainingPrimary && t_lc(s, d) >= *remainingLeaderCount)
    d = slices ? t_best_fit_slices(s, v, i, *remainingPrimary, *remainingLeaderCount) : t_best_fit_pods(s, v, i, *remainingPrimary, *remainingLeaderCount);
  *domain = d;
  bool completed;
  if (withLeader[d] >= *remainingPrimary && t_lc(s, d) >= *remainingLeaderCount) {
    wsync();
    if (lane_id() == 0) { t_touch(s, d); primary[d] = *remainingPrimary; t_set_lc(s, d, *remainingLeaderCount); s.pc[d] = *remainingPrimary * sliceSize; }
    completed = true;
  } else {
    int32_t wl = withLeader[d], lcv = t_lc(s, d);
    if (wl > *remainingPrimary) wl = *remainingPrimary;
    if (lcv > *remainingLeaderCount) lcv = *remainingLeaderCount;
    wsync();
    if (lane_id() == 0) { t_touch(s, d); withLeader[d] = wl; t_set_lc(s, d, lcv); primary[d] = wl; s.pc[d] = wl * sliceSize; }
    *remainingLeaderCount -= lcv;
    *remainingPrimary -= wl;
    completed = false;
  }
  wsync();
  return completed;
}

// updateCountsToMinimumGeneric :1575 over the slice s.set[0..n) in `order`; result domains appended to out[out_n..)
// returns the new length of out, -1 on the reference's "unexpected remainingCount" path
KQ_DEV int t_update_counts(const TK& k, const TState& s, int n, int order, int32_t count, int32_t leaderCount, int32_t sliceSize,
                           bool unconstrained, bool slices, int32_t* out, int out_n, int32_t recompute = 0, int id0 = -1) {
  const bool bestfit = !t_lfc(k, unconstrained);
  // no leader, no inner-layer recount, BestFit: keys, domains[0] and the first best-fit scan in one sweep
  const bool fuse = leaderCount == 0 && recompute <= 1 && bestfit;
  int first0 = -1, fit0 = -1;
  TView v;
  if (fuse && n == 1) {
    // one domain: no order to establish, no reduction to make
    const int d = id0 >= 0 ? id0 : s.set[0];
    const int32_t need = slices ? count / sliceSize : count;
    const bool zero = order != ORD_LIST && t_all_zero(s, d);
    first0 = zero ? -1 : d;
    fit0 = (!zero && (slices ? s.sc[d] : s.pc[d]) >= need) ? d : -1;
    v.n = 1; v.order = order; v.lfc = false; v.skip = -1;
    v.mat = zero ? 0 : 1; v.started = !zero; v.c0 = ~0ull; v.c1 = ~0ull;   // nothing left in the stream either way
    if (zero) { if (lane_id() == 0) { s.k0[0] = KQ_TAS_EXCLUDED; s.k1[0] = (uint32_t)d; } }
    else if (lane_id() == 0) { s.arr[0] = d; s.k0[0] = 0; s.k1[0] = (uint32_t)d; }
    wsync();
  } else if (fuse) v = t_view_first_fit(k, s, n, order, unconstrained, slices ? count / sliceSize : count, slices ? 2 : 0, 0, &first0, &fit0, id0);
  else { if (id0 >= 0) wsync(); v = t_view(k, s, n, order, unconstrained); }
  if (recompute > 1) {
    // an inner slice layer (:1060-1070): the children were sorted with the slice counts phase 1 left (the keys above), only then
    // are the counts recomputed for this layer's size
    for (int i = lane_id(); i < n; i += WAVE) { const int d = s.set[i]; s.sc[d] = s.pc[d] / recompute; s.scwl[d] = s.pcwl[d] / recompute; }
    if (lane_id() == 0) s.meta[1] = 1;
    wsync();
  }
  t_prioritize_leader(k, s, v, count, leaderCount, sliceSize, slices);
  int32_t remainingPrimary = slices ? count / sliceSize : count;
  int32_t remainingLeaderCount = leaderCount;
  for (int i = 0;; i++) {
    const bool pre = fuse && i == 0;   // (remainingPrimary is still what the sweep was given)
    int dom = pre ? first0 : t_get(s, v, i);
    if (dom < 0) break;
    if (remainingLeaderCount > 0) {
      const bool completed = t_consume_with_leaders(k, s, v, i, &dom, &remainingPrimary, &remainingLeaderCount, unconstrained, slices ? sliceSize : 1, slices);
      if (lane_id() == 0) out[out_n] = dom;
      out_n++;
      wsync();
      if (completed) return out_n;
      continue;
    }
    if (slices) {
      if (bestfit && s.sc[dom] >= remainingPrimary) dom = pre ? (fit0 >= 0 ? fit0 : dom) : t_best_fit_slices(s, v, i, remainingPrimary, 0);
      const int32_t scv = s.sc[dom];
      wsync();
      if (scv >= remainingPrimary) {
        if (lane_id() == 0) { t_touch(s, dom); t_set_lc(s, dom, 0); s.pc[dom] = remainingPrimary * sliceSize; s.sc[dom] = remainingPrimary; out[out_n] = dom; }
        out_n++;
        wsync();
        return out_n;
      }
      if (lane_id() == 0) { t_touch(s, dom); t_set_lc(s, dom, 0); s.pc[dom] = scv * sliceSize; out[out_n] = dom; }
      remainingPrimary -= scv;
      out_n++;
      wsync();
      continue;
    }
    if (bestfit && s.pc[dom] >= remainingPrimary) dom = pre ? (fit0 >= 0 ? fit0 : dom) : t_best_fit_pods(s, v, i, remainingPrimary, 0);
    const int32_t pcv = s.pc[dom];
    wsync();
    if (pcv >= remainingPrimary) {
      if (lane_id() == 0) { t_touch(s, dom); t_set_lc(s, dom, 0); s.pc[dom] = remainingPrimary; out[out_n] = dom; }
      out_n++;
      wsync();
      return out_n;
    }
    if (lane_id() == 0) { t_touch(s, dom); t_set_lc(s, dom, 0); out[out_n] = dom; }
    remainingPrimary -= pcv;
    out_n++;
    wsync();
  }
  // only all-zero domains are left: the reference walks them without effect, then either it had nothing left to place
  // or it reports "unexpected remainingCount" and returns nil
  return (remainingPrimary <= 0 && remainingLeaderCount <= 0) ? out_n : -1;
}

struct TFail { int status; int32_t a, b; };
struct SelLast {  // the last element of the slice: arg-max by complementing the key (the caller un-complements the id)
  KQ_MDEV bool take(uint64_t, uint64_t) const { return true; }
  KQ_MDEV void rank(uint64_t* x0, uint64_t* x1) const { *x0 = ~*x0 - 1; *x1 = ~*x1; }
};
struct SelAll {
  KQ_MDEV bool take(uint64_t, uint64_t) const { return true; }
  KQ_MDEV void rank(uint64_t*, uint64_t*) const {}
};
// sortedDomain[0] / sortedDomain[len-1] of the reference's slice, all-zero domains included (they only matter for
// the numbers a failure message quotes)
KQ_DEV int t_true_first(const TState& s, const TView& v) {
  uint64_t a = 0, b = 0;
  return t_argmin(s, v, SelAll{}, &a, &b, true) ? t_dom(b) : -1;
}
KQ_DEV int t_true_last(const TState& s, const TView& v) {
  uint64_t a = 0, b = 0;
  return t_argmin(s, v, SelLast{}, &a, &b, true) ? t_dom(~b) : -1;
}

// (the balanced placement is defined behind t_find_level's helpers, see below)
// findLevelWithFitDomains :1336 ; on success the fitting domains are in s.cur[0..*nfit)
KQ_DEV TFail t_find_level(const TK& k, const TState& s, const TParams& st, int* fitLevel, int* nfit) {
  const TTopo& T = k.T;
  const int32_t sliceCount = st.count / st.sliceSize;
  const bool lfc = t_lfc(k, st.unconstrained);
  for (int searchLevelIdx = st.requestedLevelIdx;; searchLevelIdx--) {
    *fitLevel = searchLevelIdx;  // also the level a KQ_TAS_NOT_FIT below refers to (notFitReason :1354)
    const int n = T.level_off[searchLevelIdx + 1] - T.level_off[searchLevelIdx];
    if (n == 0) return TFail{KQ_TAS_NO_LEVEL, 0, 0};
    // A slot whose rows are global memory (k_tas_find) writes the level's id list and sort keys only when something is going to read them:
    // a level search that ends with the sweep's own answer (one domain holds everything; nothing does and the level above is next) never
    // does, and an unconstrained request sweeps every leaf — 20 B written per leaf and workload, 2.28 GB per 50 000-workload launch at
    // 4096 leaves (profiles/r05w). Nothing changes the state between the sweep and need_keys(): the keys are the ones the sweep saw.
    const bool lazy = s.lazy_keys;
    const int lvl_id0 = T.level_off[searchLevelIdx];
    if (!lazy) for (int i = lane_id(); i < n; i += WAVE) s.set[i] = lvl_id0 + i;
    bool have_keys = !lazy;
    auto need_keys = [&]() {
      if (have_keys) return;
      have_keys = true;
      for (int i = lane_id(); i < n; i += WAVE) {
        const int d = lvl_id0 + i;
        uint64_t a, b;
        t_key_of(s, d, i, ORD_LEADER, lfc, &a, &b);
        if (t_all_zero(s, d)) a |= KQ_TAS_EXCLUDED;
        s.set[i] = d; s.k0[i] = a; s.k1[i] = b;
      }
      wsync();
    };
    int fitDomain = -1, topDomain = -1;
    TView v;
    TPROF0();
    // keys, sortedDomain[0] and the domain the level's search is after — BestFit: findBestFitDomainBy over the whole level;
    // LeastFreeCapacity: the first domain in ascending order that holds everything — in one sweep (its closing fence publishes s.set as well)
    v = t_view_first_fit(k, s, n, ORD_LEADER, st.unconstrained, sliceCount, st.leaderCount > 0 ? 3 : 2, st.leaderCount, &topDomain, &fitDomain,
                         lvl_id0, lfc, !lazy);
    TPROF(k, 3);   // (timing builds) the level's sweep
    if (topDomain < 0) {
      // every domain of the level has an all-zero state: whatever sortedDomain[0] is, it holds nothing
      if (sliceCount == 0 && st.leaderCount == 0) {
        if (lane_id() == 0) s.cur[0] = T.level_off[searchLevelIdx];
        *fitLevel = searchLevelIdx; *nfit = 1;
        wsync();
        return TFail{KQ_TAS_OK, 0, 0};
      }
      if (st.required || searchLevelIdx == 0 || st.unconstrained) return TFail{KQ_TAS_NOT_FIT, 0, sliceCount};
      continue;
    }
    if (!lfc && s.scwl[topDomain] >= sliceCount && t_lc(s, topDomain) >= st.leaderCount && fitDomain >= 0) topDomain = fitDomain;   // findBestFitDomain :1293 over the whole level
    if (lfc) {
      if (fitDomain >= 0) {
        if (lane_id() == 0) s.cur[0] = fitDomain;
        *fitLevel = searchLevelIdx; *nfit = 1;
        wsync();
        return TFail{KQ_TAS_OK, 0, 0};
      }
      if (st.required) {
        need_keys();
        const int last = t_true_last(s, v);
        return TFail{KQ_TAS_NOT_FIT, last >= 0 ? s.pc[last] : 0, sliceCount};
      }
    }
    if (s.scwl[topDomain] < sliceCount || t_lc(s, topDomain) < st.leaderCount) {
      if (st.required) { need_keys(); return TFail{KQ_TAS_NOT_FIT, s.sc[t_true_first(s, v)], sliceCount}; }
      if (searchLevelIdx > 0 && !st.unconstrained) continue;
      // the LeastFreeCapacity histogram below reads the counts alone and builds its own slice: no keys of the whole level either
      const bool histo = lfc && st.leaderCount == 0 && n > 128 && k.X.max_set >= 128;
      if (!histo) need_keys();
      int nres = 0;
      int32_t remainingSliceCount = sliceCount, remainingLeaderCount = st.leaderCount;
      t_prioritize_leader(k, s, v, st.count, st.leaderCount, st.sliceSize, true);
      int idx = 0;
      for (; remainingLeaderCount > 0; idx++) {
        int domain = t_get(s, v, idx);
        if (domain < 0 || t_lc(s, domain) <= 0) break;
        if (!lfc && s.scwl[domain] >= remainingSliceCount) domain = t_best_fit_slices(s, v, idx, remainingSliceCount, remainingLeaderCount);
        if (lane_id() == 0) s.cur[nres] = domain;
        nres++;
        remainingLeaderCount -= t_lc(s, domain);
        remainingSliceCount -= s.scwl[domain];
      }
      if (remainingLeaderCount > 0) return TFail{KQ_TAS_NOT_FIT, st.leaderCount - remainingLeaderCount, sliceCount};
      wsync();
      if (remainingSliceCount > 0) {
        // sortedDomains(sortedDomain[idx:]): the rest of the slice re-sorted by worker capacity
        int m = n;
        if (idx > 0) {
          m = 0;
          for (int base = 0; base < n; base += WAVE) {
            const int i = base + lane_id();
            bool keep = false; int d = 0;
            if (i < n) {
              d = T.level_off[searchLevelIdx] + i;
              keep = true;
              for (int q = 0; q < idx; q++) if (s.arr[q] == d) keep = false;
            }
            const uint64_t mk = wballot(keep);
            const int my = m + popc64(mk & ((lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()))));
            if (keep) s.nxt[my] = d;
            m += popc64(mk);
          }
          wsync();
          for (int i = lane_id(); i < m; i += WAVE) s.set[i] = s.nxt[i];
          wsync();
        }
        if (lfc && idx == 0 && m > 128 && k.X.max_set >= 128) {  // 64 int64 bins live in s.nxt
          // LeastFreeCapacity walks the slice in ascending sliceCount until the capacity adds up (:1425-1436). Every
          // domain whose sliceCount lies below the value at which the running sum reaches the target is taken whole, in
          // any order (none of them can hold the remainder on its own, so updateCountsToMinimumGeneric consumes each of
          // them entirely); only the domains AT that value need the slice order. One histogram pass finds the value.
          constexpr int NB = 64;  // sliceCount values >= NB - 1 share the last bin (its members are ordered by the view)
          long long* bins = (long long*)s.nxt;
          for (int b = lane_id(); b < NB; b += WAVE) bins[b] = 0;
          wsync();
          const int lvl0 = T.level_off[searchLevelIdx];   // (idx == 0: the slice is still the whole level, s.set[i] = lvl0 + i)
          for (int i = lane_id(); i < m; i += WAVE) {
            const int32_t scv = s.sc[lvl0 + i];
            if (scv > 0) atomic_add_i64(&bins[scv < NB - 1 ? scv : NB - 1], (long long)scv);
          }
          wsync();
          int tb = -1;
          int64_t cum = 0;
          for (int b = 0; b < NB; b++) {
            const int64_t v = bins[b];
            if (cum + v >= (int64_t)remainingSliceCount) { tb = b; break; }
            cum += v;
          }
          if (tb < 0) return TFail{KQ_TAS_NOT_FIT, (int32_t)cum, sliceCount};
          // below the threshold -> results; at the threshold -> the slice the loop below still has to order
          int m2 = 0;
          for (int base = 0; base < m; base += WAVE) {
            const int i = base + lane_id();
            int d = 0, bin = -1;
            if (i < m) { d = lvl0 + i; const int32_t scv = s.sc[d]; bin = scv > 0 ? (scv < NB - 1 ? scv : NB - 1) : -1; }
            const uint64_t lo = wballot(bin >= 0 && bin < tb), at = wballot(bin == tb);
            const uint64_t below_me = (lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()));
            if (bin >= 0 && bin < tb) s.cur[nres + popc64(lo & below_me)] = d;
            if (bin == tb) s.arr[m2 + popc64(at & below_me)] = d;
            nres += popc64(lo);
            m2 += popc64(at);
          }
          wsync();
          for (int i = lane_id(); i < m2; i += WAVE) s.set[i] = s.arr[i];
          wsync();
          remainingSliceCount -= (int32_t)cum;
          m = m2;
          TPROF(k, 7);   // LeastFreeCapacity: the histogram threshold (everything of the level between the sweep and here)
        }
        // (a global-row state, behind the histogram: the domains at the threshold value are often hundreds, a few of them are consumed, and
        // nothing changes the state while they are walked — no keys are stored for them)
        TView r;
        if (lazy && histo) { r.n = m; r.order = ORD_PLAIN; r.lfc = lfc; r.mat = 0; r.c0 = 0; r.c1 = 0; r.started = false; r.skip = -1; r.virt = true; }
        else r = t_view(k, s, m, ORD_PLAIN, st.unconstrained);
        for (int i = 0; remainingSliceCount > 0; i++) {
          int domain = t_get(s, r, i);
          if (domain < 0) break;
          if (!lfc && s.sc[domain] >= remainingSliceCount) domain = t_best_fit_slices(s, r, i, remainingSliceCount, 0);
          if (lane_id() == 0) s.cur[nres] = domain;
          nres++;
          remainingSliceCount -= s.sc[domain];
        }
        if (remainingSliceCount > 0) return TFail{KQ_TAS_NOT_FIT, sliceCount - remainingSliceCount, sliceCount};
      }
      *fitLevel = searchLevelIdx; *nfit = nres;
      wsync();
      return TFail{KQ_TAS_OK, 0, 0};
    }
    if (lane_id() == 0) s.cur[0] = topDomain;
    *fitLevel = searchLevelIdx; *nfit = 1;
    wsync();
    return TFail{KQ_TAS_OK, 0, 0};
  }
}

// multiLayerNotFitMessage :2030 as operands: a = the domain of the failing level with the most slices (ties: the lower canonical
// index = the lower domain id), b = that level; lf[i] = countSlicesInSubtree :2019 of that domain for layer i. Domains are numbered in
// canonical order with contiguous children, so the descendants of a domain on a lower level are one index range.
KQ_DEV TFail t_not_fit_layers(const TK& k, const TState& s, const TParams& st, int lvl, int32_t* lf0, int32_t* lf1) {
  const TTopo& T = k.T;
  const int lane = lane_id();
  const int b0 = T.level_off[lvl], n = T.level_off[lvl + 1] - b0;
  uint64_t key = ~0ull;
  for (int i = lane; i < n; i += WAVE) {
    const uint64_t kk = ((uint64_t)(uint32_t)(0x7fffffff - s.sc[b0 + i]) << 32) | (uint32_t)i;
    if (kk < key) key = kk;
  }
  key = wmin_u64(key);
  const int best = n > 0 ? (int)(uint32_t)(key & 0xffffffffu) : -1;
  for (int i = 0; i < KQ_TAS_MAX_LEVELS; i++) {
    int32_t fit = 0;
    if (best >= 0 && i < st.nLayers && st.layerLevel[i] >= lvl) {
      int lo = b0 + best, hi = lo + 1;
      for (int l = lvl; l < st.layerLevel[i]; l++) { const int nlo = T.child_first[lo], nhi = T.child_first[hi - 1] + T.child_cnt[hi - 1]; lo = nlo; hi = nhi; }
      int64_t part = 0;
      for (int d = lo + lane; d < hi; d += WAVE) part += s.pc[d] / st.layerSize[i];
      fit = (int32_t)wsum_i64(part);
    }
    if (lane == 0) { if (lf0) lf0[i] = fit; if (lf1) lf1[i] = fit; }
  }
  return TFail{KQ_TAS_NOT_FIT_LAYERS, best, lvl};
}

// The gate is off by default: its code is compiled into kernel variants of their own (KQ_TAS_BAL: k_tas_find_bal, k_nominate_tas_bal,
// k_process_tas_bal — chosen by the host when a topology carries KQ_TAS_F_BALANCED_PLACEMENT), so that the kernels everybody runs keep their
// registers and stay without scratch (k_tas_find: 0 B; with the routine as a callee it had 392 B and ran 7 % slower, profiles/r05t_*).
#ifdef KQ_TAS_BAL
// ---- tas_balanced_placement.go (features.TASBalancedPlacement; preferred requests only, tas_flavor_snapshot.go:1012) ------------------------
// A gate that is off by default: the whole of it runs on lane 0 over the slot's own scratch (TScratch::bal) — plain loops, insertion
// sorts, a dynamic programme over a table of back-pointers. The placement's other steps (phase 1 before it, updateCountsToMinimum and the
// descent after it) are the wave-wide ones. Every candidate set of the reference works on cloneDomains :347 of the subtrees; here the
// state arrays are saved / restored as a whole, which is the same for the subtrees a candidate touches.
struct TBal {
  int32_t *orig, *best;           // [5 * D] pc | sc | pcwl | scwl | lc
  int32_t *la, *lb, *lc, *ld;     // [W] domain lists
  int32_t* dp; long long dp_cap;  // [dp_cap][2]: (domain | -2 = the empty placement | -1 = unset, previous state's index)
  long long bytes;                // algorithmic bytes of the roll-ups (fillInCountsHelper: 24 per inner domain)
  bool overflow;                  // the table does not hold the request: KQ_EUNSUPPORTED
};
KQ_DEV TBal t_bal_of(const TK& k, int slot) {
  TBal b;
  int32_t* p = k.X.bal + (size_t)slot * k.X.bal_stride;
  const size_t D = k.T.D, W = k.X.bal_w;
  b.orig = p; b.best = p + 5 * D; b.la = p + 10 * D; b.lb = b.la + W; b.lc = b.lb + W; b.ld = b.lc + W; b.dp = b.ld + W;
  b.dp_cap = k.X.bal_dp; b.bytes = 0; b.overflow = false;
  return b;
}
KQ_DEV void t_bal_save(const TK& k, const TState& s, int32_t* dst) {
  const int D = k.T.D;
  for (int d = 0; d < D; d++) { dst[d] = s.pc[d]; dst[D + d] = s.sc[d]; dst[2 * D + d] = s.pcwl[d]; dst[3 * D + d] = s.scwl[d]; dst[4 * D + d] = t_lc(s, d); }
}
KQ_DEV void t_bal_load(const TK& k, const TState& s, const int32_t* src) {
  const int D = k.T.D;
  for (int d = 0; d < D; d++) { s.pc[d] = src[d]; s.sc[d] = src[D + d]; s.pcwl[d] = src[2 * D + d]; s.scwl[d] = src[3 * D + d]; t_set_lc(s, d, src[4 * D + d]); }
}
// sortedDomainsWithLeader :1731 / sortedDomains :1770 (BestFit: these callers pass unconstrained = false), in place
KQ_DEV bool t_bal_before(const TState& s, int a, int b, bool withLeader) {
  if (withLeader) {
    if (t_lc(s, a) != t_lc(s, b)) return t_lc(s, b) < t_lc(s, a);
    if (s.scwl[a] != s.scwl[b]) return s.scwl[b] < s.scwl[a];
    if (s.pcwl[a] != s.pcwl[b]) return s.pcwl[a] < s.pcwl[b];
    return a < b;
  }
  if (s.sc[a] != s.sc[b]) return s.sc[b] < s.sc[a];
  if (s.pc[a] != s.pc[b]) return s.pc[a] < s.pc[b];
  return a < b;
}
KQ_DEV void t_bal_sort(const TState& s, int32_t* l, int n, bool withLeader) {
  for (int i = 1; i < n; i++) {
    const int x = l[i];
    int j = i - 1;
    while (j >= 0 && t_bal_before(s, x, l[j], withLeader)) { l[j + 1] = l[j]; j--; }
    l[j + 1] = x;
  }
}
struct TGreedy { bool fits; int32_t selected; int lastWithLeader, last; };
// evaluateGreedyAssignment :28 over the list l[0..n) (tmp: a list of the same length)
KQ_DEV TGreedy t_bal_greedy(const TState& s, const int32_t* l, int n, int32_t* tmp, int32_t sliceCount, int32_t leaderCount) {
  TGreedy g{false, 0, -1, -1};
  for (int i = 0; i < n; i++) tmp[i] = l[i];
  int32_t remainingSlice = sliceCount, remainingLeader = leaderCount;
  int idx = 0;
  if (leaderCount > 0) {
    t_bal_sort(s, tmp, n, true);
    for (; remainingLeader > 0 && idx < n && t_lc(s, tmp[idx]) > 0; idx++) {
      g.selected++; g.lastWithLeader = tmp[idx];
      remainingLeader -= t_lc(s, tmp[idx]);
      remainingSlice -= s.scwl[tmp[idx]];
    }
  }
  t_bal_sort(s, tmp + idx, n - idx, false);
  if (remainingLeader > 0) return TGreedy{false, 0, -1, -1};
  for (int i = idx; remainingSlice > 0 && i < n && s.sc[tmp[i]] > 0; i++) {
    g.selected++; g.last = tmp[i];
    remainingSlice -= s.sc[tmp[i]];
  }
  if (remainingSlice > 0) return TGreedy{false, 0, -1, -1};
  g.fits = true;
  return g;
}
KQ_DEV int32_t t_bal_threshold(const TState& s, int32_t sliceCount, const TGreedy& g) {   // balanceThresholdValue :64
  int32_t t = sliceCount / g.selected;
  if (g.lastWithLeader >= 0 && s.scwl[g.lastWithLeader] < t) t = s.scwl[g.lastWithLeader];
  if (g.last >= 0 && s.sc[g.last] < t) t = s.sc[g.last];
  return t;
}
// math.Log as Go computes it on amd64 (src/math/log.go: the pure-Go port of FreeBSD's e_log.c — there is no assembly stub outside s390x):
// the libm of the host and HIP's device log are correctly rounded to < 1 ulp too, but not bit-identical to it, and a 1-ulp difference in
// calculateDomainsEntropy can flip compareDomainCapacityAndEntropy between two near-equal domains. x is a finite positive fraction here
// (frac of Frexp, in [0.5, 1)); no fused multiply-add (-ffp-contract=off; GOAMD64=v1 does not fuse either).
KQ_DEV double t_go_log(double x) {
  const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10,
               L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
               L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01, L7 = 1.479819860511658591e-01;
  int ki; double f1 = ::frexp(x, &ki);
  if (f1 < 0.70710678118654752440 /* Sqrt2 / 2 */) { f1 *= 2; ki--; }
  const double f = f1 - 1, k = (double)ki;
  const double s = f / (2 + f), s2 = s * s, s4 = s2 * s2;
  const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
  const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
  const double R = t1 + t2, hfsq = 0.5 * f * f;
  return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}
// math.Log2 (log2.go): Frexp, then Log(frac) * (1 / Ln2) + exp
KQ_DEV double t_bal_log2(double x) {
  int e; const double frac = ::frexp(x, &e);
  if (frac == 0.5) return (double)(e - 1);
  return t_go_log(frac) * (1.0 / 0.693147180559945309417232121458176568) + (double)e;
}
KQ_DEV double t_bal_entropy(const TK& k, const TState& s, int d) {   // calculateDomainsEntropy :189 of d's children
  const int c0 = k.T.child_first[d], cn = c0 >= 0 ? k.T.child_cnt[d] : 0;
  if (cn == 0) return 0.0;
  int32_t total = 0;
  for (int c = c0; c < c0 + cn; c++) total += s.pc[c];
  if (total == 0) return 0.0;
  double e = 0.0; const double tf = (double)total;
  for (int c = c0; c < c0 + cn; c++) if (s.pc[c] > 0) { const double pI = (double)s.pc[c] / tf; e += -pI * t_bal_log2(pI); }
  return e;
}
KQ_DEV bool t_bal_before_entropy(const TK& k, const TState& s, int a, int b) {   // compareDomainCapacityAndEntropy :214 < 0
  if (t_lc(s, a) != t_lc(s, b)) return t_lc(s, b) < t_lc(s, a);
  if (s.scwl[a] != s.scwl[b]) return s.scwl[b] < s.scwl[a];
  const double ae = t_bal_entropy(k, s, a), be = t_bal_entropy(k, s, b);
  if (be > ae) return false;
  if (be < ae) return true;
  return a < b;
}
// selectOptimalDomainSetToFit :79: the result in out[0..return), -1 = nil. The Go maps keyed by (leaders left, pods left) are one table
// of back-pointers here: "first writer wins" with the keys walked in ascending order, as slices.Sorted(maps.Keys(..)) does.
KQ_DEV int t_bal_select(const TK& k, const TState& s, TBal& b, const int32_t* l, int n, int32_t* ord, int32_t* tmp, int32_t* out,
                        int32_t sliceCount, int32_t leaderCount, int32_t sliceSize, bool byEntropy) {
  const TGreedy g = t_bal_greedy(s, l, n, tmp, sliceCount, leaderCount);
  if (!g.fits) return -1;
  const int optimal = g.selected;
  for (int i = 0; i < n; i++) ord[i] = l[i];
  for (int i = 1; i < n; i++) {   // slices.SortFunc by the entropy order / by levelValues
    const int x = ord[i];
    int j = i - 1;
    while (j >= 0 && (byEntropy ? t_bal_before_entropy(k, s, x, ord[j]) : x < ord[j])) { ord[j + 1] = ord[j]; j--; }
    ord[j + 1] = x;
  }
  int32_t maxPc = 1;
  for (int i = 0; i < n; i++) { if (s.pc[ord[i]] > maxPc) maxPc = s.pc[ord[i]]; if (s.pcwl[ord[i]] > maxPc) maxPc = s.pcwl[ord[i]]; }
  // pods left: from P0 down. A state with nothing left (no leader, pods <= 0) is final, so without a leader the lowest key is
  // 1 - maxPc; while a leader is still to be placed a state goes on below zero: at most `optimal` picks of maxPc each
  const long long P0 = (long long)sliceCount * sliceSize, minP = leaderCount > 0 ? P0 - (long long)optimal * maxPc - 1 : 1 - (long long)maxPc,
                  R = P0 - minP + 1, Lr = (long long)leaderCount + 1;
  const long long cells = (long long)(optimal + 1) * Lr * R;
  if (cells > b.dp_cap || P0 > 0x3fffffff) { b.overflow = true; return -1; }
  for (long long c = 0; c < cells; c++) b.dp[2 * c] = -1;
  auto at = [&](int i, long long ld, long long pods) { return ((long long)i * Lr + ld) * R + (pods - minP); };
  b.dp[2 * at(0, leaderCount, P0)] = -2;
  for (int q = 0; q < n; q++) {
    const int d = ord[q];
    const int32_t dlc = t_lc(s, d), dpc = s.pc[d], dpcwl = s.pcwl[d], dsc = s.sc[d];
    for (int i = optimal; i > 0; i--)
      for (long long bl = 0; bl < Lr; bl++)
        for (long long bp = minP; bp <= P0; bp++) {
          const long long from = at(i - 1, bl, bp);
          if (b.dp[2 * from] == -1) continue;
          if (bl <= 0 && bp <= 0) continue;
          if (bl > 0 && dlc > 0) {   // pick this domain with leader
            const long long al = bl - dlc, ap = bp - dpcwl;
            if (al >= 0 && ap >= minP) { const long long to = at(i, al, ap); if (b.dp[2 * to] == -1) { b.dp[2 * to] = d; b.dp[2 * to + 1] = (int32_t)from; } }
          }
          if (dsc > 0) {             // pick this domain without leader
            const long long ap = bp - dpc;
            if (ap >= minP) { const long long to = at(i, bl, ap); if (b.dp[2 * to] == -1) { b.dp[2 * to] = d; b.dp[2 * to + 1] = (int32_t)from; } }
          }
        }
  }
  long long bestIdx = -1;
  for (long long pods = 0; pods >= minP; pods--) { const long long c = at(optimal, 0, pods); if (b.dp[2 * c] != -1) { bestIdx = c; break; } }
  if (bestIdx < 0) return -1;
  int m = 0;
  for (long long c = bestIdx; b.dp[2 * c] != -2; c = b.dp[2 * c + 1]) out[m++] = b.dp[2 * c];
  for (int i = 0; i < m / 2; i++) { const int t = out[i]; out[i] = out[m - 1 - i]; out[m - 1 - i] = t; }
  return m;
}
// placeSlicesOnDomainsBalanced :151 on the list l[0..n): the chosen domains, in sortedDomainsWithLeader order, in out; -1 = a failure reason
KQ_DEV int t_bal_place(const TK& k, const TState& s, TBal& b, const int32_t* l, int n, int32_t* ord, int32_t* tmp, int32_t* out,
                       int32_t sliceCount, int32_t leaderCount, int32_t sliceSize, int32_t threshold) {
  const int m = t_bal_select(k, s, b, l, n, ord, tmp, out, sliceCount, leaderCount, sliceSize, false);
  if (m < 0) return -1;
  if (sliceCount < (int32_t)m * threshold) return -1;
  t_bal_sort(s, out, m, true);
  int32_t extraLeft = sliceCount - (int32_t)m * threshold, leadersLeft = leaderCount, take = 0;
  for (int i = 0; i < m; i++) {
    const int d = out[i];
    if (leadersLeft > 0) { take = s.scwl[d] - threshold < extraLeft ? s.scwl[d] - threshold : extraLeft; t_set_lc(s, d, 1); leadersLeft--; }
    else if (extraLeft > 0) { take = s.sc[d] - threshold < extraLeft ? s.sc[d] - threshold : extraLeft; t_set_lc(s, d, 0); }
    else { t_set_lc(s, d, 0); take = 0; }
    s.pc[d] = (threshold + take) * sliceSize;
    s.sc[d] = threshold + take;
    s.scwl[d] = s.sc[d];
    s.pcwl[d] = s.pc[d] - t_lc(s, d);
    extraLeft -= take;
  }
  if (extraLeft > 0 || leadersLeft > 0) return -1;
  return m;
}
// the id range of d's subtree at depth `dep` below d's own level (children of consecutive domains are consecutive)
KQ_DEV void t_bal_range(const TK& k, int d, int dep, int* lo, int* hi) {
  int a = d, z = d + 1;
  for (int i = 0; i < dep && z > a; i++) {
    int na = -1, nz = -1;
    for (int x = a; x < z; x++) if (k.T.child_first[x] >= 0 && k.T.child_cnt[x] > 0) { if (na < 0) na = k.T.child_first[x]; nz = k.T.child_first[x] + k.T.child_cnt[x]; }
    if (na < 0) { a = z = 0; break; }
    a = na; z = nz;
  }
  *lo = a; *hi = z;
}
KQ_DEV void t_bal_clear(const TK& k, const TState& s, int d, int level, bool leaderOnly) {   // clearState :324 / clearLeaderCapacity :332
  for (int dep = 0; level + dep < k.T.L; dep++) {
    int lo, hi;
    t_bal_range(k, d, dep, &lo, &hi);
    for (int x = lo; x < hi; x++) {
      if (!leaderOnly) { s.pc[x] = 0; s.sc[x] = 0; }
      s.pcwl[x] = 0; s.scwl[x] = 0; t_set_lc(s, x, 0);
    }
  }
}
KQ_DEV void t_bal_prune_node(const TK& k, const TState& s, int d, int level, int32_t threshold, bool leaderRequired) {   // :365
  if (s.sc[d] < threshold) { t_bal_clear(k, s, d, level, false); return; }
  if (leaderRequired && t_lc(s, d) > 0 && s.scwl[d] < threshold) t_bal_clear(k, s, d, level, true);
}
// fillInCountsHelper :1930 for the subtree of d (level `level`), without inner slice layers (pruneDomainsBelowThreshold passes nil)
KQ_DEV void t_bal_rollup(const TK& k, const TState& s, TBal& b, int d, int level, int32_t sliceSize, int sliceLevelIdx, bool leaderRequired) {
  for (int lv = k.T.L - 1; lv >= level; lv--) {
    int lo, hi;
    t_bal_range(k, d, lv - level, &lo, &hi);
    for (int x = lo; x < hi; x++) {
      const int c0 = k.T.child_first[x], cn = c0 >= 0 ? k.T.child_cnt[x] : 0;
      if (cn == 0) {
        if (lv == sliceLevelIdx) { s.sc[x] = s.pc[x] / sliceSize; s.scwl[x] = s.pcwl[x] / sliceSize; }
        continue;
      }
      int32_t childrenCapacity = 0, sliceCapacity = 0, minPodDiff = 0x7fffffff, minSliceDiff = 0x7fffffff, leaderCount = 0;
      bool contributor = false;
      for (int c = c0; c < c0 + cn; c++) {
        childrenCapacity += s.pc[c];
        sliceCapacity += s.sc[c];
        if (!leaderRequired || t_lc(s, c) > 0) {
          contributor = true;
          if (s.pc[c] - s.pcwl[c] < minPodDiff) minPodDiff = s.pc[c] - s.pcwl[c];
          if (s.sc[c] - s.scwl[c] < minSliceDiff) minSliceDiff = s.sc[c] - s.scwl[c];
        }
        if (t_lc(s, c) > leaderCount) leaderCount = t_lc(s, c);
      }
      s.pc[x] = childrenCapacity;
      int32_t scwl = 0;
      if (contributor) { s.pcwl[x] = childrenCapacity - minPodDiff; scwl = sliceCapacity - minSliceDiff; } else s.pcwl[x] = 0;
      t_set_lc(s, x, leaderCount);
      if (lv == sliceLevelIdx) { sliceCapacity = s.pc[x] / sliceSize; scwl = s.pcwl[x] / sliceSize; }
      s.sc[x] = sliceCapacity; s.scwl[x] = scwl;
      b.bytes += 24;
    }
  }
}
// pruneDomainsBelowThreshold :377 over the sibling range [a, a + n) at `level`
KQ_DEV void t_bal_prune(const TK& k, const TState& s, TBal& b, int a, int n, int32_t threshold, int32_t sliceSize, int sliceLevelIdx, int level, bool leaderRequired) {
  for (int d = a; d < a + n; d++) {
    const int c0 = k.T.child_first[d], cn = c0 >= 0 ? k.T.child_cnt[d] : 0;
    for (int c = c0; c < c0 + cn; c++) t_bal_prune_node(k, s, c, level + 1, threshold, leaderRequired);
  }
  for (int d = a; d < a + n; d++) {
    t_bal_rollup(k, s, b, d, level, sliceSize, sliceLevelIdx, leaderRequired);
    t_bal_prune_node(k, s, d, level, threshold, leaderRequired);
  }
}
// :1012-1024 of findTopologyAssignment: findBestDomainsForBalancedPlacement :235 + applyBalancedPlacementAlgorithm :296. Lane 0 only.
// true: s.cur[0..*ncur) holds currFitDomain at *fitLevel and the state is the chosen candidate's; false: the state is the original
// one and findLevelWithFitDomains takes over ("falling back to Best Fit").
// (not inlined: a gate that is off by default must not grow the placement every kernel carries)
KQ_NOINLINE bool t_balanced_lane0(const TK& k, const TState& s, const TParams& p, TBal& b, int* fitLevel, int* ncur) {
  const TTopo& T = k.T;
  const int32_t sliceCount = p.count / p.sliceSize;
  const bool leaderRequired = p.leaderCount > 0;
  const int req = p.requestedLevelIdx;
  t_bal_save(k, s, b.orig);
  int32_t bestThreshold = 0, bestDomainCount = 0;
  int bestA = -1, bestN = 0;
  const int nsets = req == 0 ? 1 : T.level_off[req] - T.level_off[req - 1];
  for (int si = 0; si < nsets; si++) {
    int a, n;
    if (req == 0) { a = T.level_off[0]; n = T.level_off[1] - T.level_off[0]; }
    else { const int h = T.level_off[req - 1] + si; a = T.child_first[h]; n = a >= 0 ? T.child_cnt[h] : 0; }
    if (si > 0) t_bal_load(k, s, b.orig);   // cloneDomains: every candidate starts from the shared state
    if (n <= 0) continue;
    // getLowerLevelDomains :317
    int la = a, ln = n;
    if (req < p.sliceLevelIdx) { int lo, hi; lo = -1; hi = -1; for (int d = a; d < a + n; d++) if (T.child_first[d] >= 0 && T.child_cnt[d] > 0) { if (lo < 0) lo = T.child_first[d]; hi = T.child_first[d] + T.child_cnt[d]; } la = lo < 0 ? 0 : lo; ln = lo < 0 ? 0 : hi - lo; }
    for (int i = 0; i < ln; i++) b.la[i] = la + i;
    const TGreedy g = t_bal_greedy(s, b.la, ln, b.lb, sliceCount, p.leaderCount);
    if (!g.fits || g.selected == 0) continue;   // (selected == 0: an empty request, which the reference never sends here)
    int32_t threshold = t_bal_threshold(s, sliceCount, g);
    int32_t withLeaderReservation = threshold;
    if (p.leaderCount > 0 && g.last >= 0 && s.scwl[g.last] < withLeaderReservation) withLeaderReservation = s.scwl[g.last];
    if (threshold < bestThreshold) continue;
    t_bal_prune(k, s, b, a, n, threshold, p.sliceSize, p.sliceLevelIdx, req, leaderRequired);
    for (int i = 0; i < n; i++) b.la[i] = a + i;
    TGreedy after = t_bal_greedy(s, b.la, n, b.lb, sliceCount, p.leaderCount);
    if (!after.fits && withLeaderReservation < threshold) {
      if (withLeaderReservation <= 0 || withLeaderReservation < bestThreshold) continue;
      threshold = withLeaderReservation;
      t_bal_load(k, s, b.orig);
      t_bal_prune(k, s, b, a, n, threshold, p.sliceSize, p.sliceLevelIdx, req, leaderRequired);
      after = t_bal_greedy(s, b.la, n, b.lb, sliceCount, p.leaderCount);
    }
    if (!after.fits) continue;
    if (threshold > bestThreshold || (threshold == bestThreshold && after.selected < bestDomainCount)) {
      bestThreshold = threshold; bestDomainCount = after.selected; bestA = a; bestN = n;
      t_bal_save(k, s, b.best);
    }
  }
  if (bestThreshold <= 0) { t_bal_load(k, s, b.orig); return false; }
  t_bal_load(k, s, b.best);
  // applyBalancedPlacementAlgorithm :296
  int n = bestN;
  for (int i = 0; i < n; i++) b.la[i] = bestA + i;
  int32_t* curr = b.la;
  if (req < p.sliceLevelIdx) {
    const int m = t_bal_select(k, s, b, b.la, n, b.lb, b.lc, b.ld, sliceCount, p.leaderCount, p.sliceSize, true);
    if (m < 0) { t_bal_load(k, s, b.orig); return false; }
    n = 0;
    for (int i = 0; i < m; i++) { const int d = b.ld[i], c0 = T.child_first[d], cn = c0 >= 0 ? T.child_cnt[d] : 0; for (int c = c0; c < c0 + cn; c++) b.la[n++] = c; }   // lowerLevelDomains
    *fitLevel = req + 1;
  } else *fitLevel = req;
  const int m = t_bal_place(k, s, b, curr, n, b.lb, b.lc, b.ld, sliceCount, p.leaderCount, p.sliceSize, bestThreshold);
  if (m < 0) { t_bal_load(k, s, b.orig); return false; }
  for (int i = 0; i < m; i++) s.cur[i] = b.ld[i];
  *ncur = m;
  return true;
}
#endif  // KQ_TAS_BAL

// findTopologyAssignment :886. On success the leaves of the assignment are in s.cur[0..*nfit) with their pod / leader
// counts in s.pc / s.lc.
KQ_DEV TFail t_find_assignment(const TK& k, const TState& s, const TParams& st, int* nfit, bool have_counts) {
  const TTopo& T = k.T;
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  const long long _p0 = clock64();
#endif
  if (!have_counts) t_fill_in_counts(k, s, st, k.O.bytes);
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  if (lane_id() == 0) k.O.bytes[1] += clock64() - _p0;   // (timing builds: cycles of phase 1, in the word behind the byte counter)
#endif
  int fitLevelIdx = 0, ncur = 0;
  TPROF0();
  bool balanced = false;
#ifdef KQ_TAS_BAL
  if (T.balanced && !st.required && !st.unconstrained && k.X.bal) {   // tas_flavor_snapshot.go:1012-1024
    // (lane 0 alone; its verdict travels through the head of the slot's scratch. The class table's state, if the placement started from
    // one, is no longer restorable by the log: the next workload copies it again)
    TBal b = t_bal_of(k, s.bal_slot);
    int32_t* verdict = b.dp;   // [4] behind the routine: used, fitLevel, ncur, error
    wsync();
    if (lane_id() == 0) {
      int fl = 0, nc = 0;
      const bool used = t_balanced_lane0(k, s, st, b, &fl, &nc);
      s.meta[1] = 1;
      atomic_add_i64(k.O.bytes, b.bytes);
      verdict[0] = used ? 1 : 0; verdict[1] = fl; verdict[2] = nc; verdict[3] = b.overflow ? 1 : 0;
      if (b.overflow && *k.O.error == 0) *k.O.error = KQ_EUNSUPPORTED;
    }
    wsync();
    balanced = verdict[0] != 0; fitLevelIdx = verdict[1]; ncur = verdict[2];
    wsync();
  }
#endif
  TFail f{KQ_TAS_OK, 0, 0};
  if (!balanced) f = t_find_level(k, s, st, &fitLevelIdx, &ncur);
  TPROF(k, 0);   // findLevelWithFitDomains
  if (f.status == KQ_TAS_NOT_FIT && st.nLayers > 0) { *nfit = fitLevelIdx; return f; }  // the caller turns it into the per-layer form
  if (f.status != KQ_TAS_OK) return f;
  // phase 2b :1041 — currFitDomain in the order findLevelWithFitDomains built it
  for (int i = lane_id(); i < ncur; i += WAVE) s.set[i] = s.cur[i];
  const int only = ncur == 1 ? s.cur[0] : -1;   // the usual case: one fitting domain, nothing to order
  if (only < 0) wsync();
  int nout = t_update_counts(k, s, ncur, ORD_LIST, st.count, st.leaderCount, st.sliceSize, st.unconstrained, true, s.nxt, 0, 0, only);
  if (nout < 0) { *nfit = 0; return TFail{KQ_TAS_OK, 0, 0}; }
  for (int i = lane_id(); i < nout; i += WAVE) s.cur[i] = s.nxt[i];
  wsync();
  ncur = nout;
  TPROF(k, 1);   // the fit level's own domains
  int level = fitLevelIdx;
  const int stop = balanced ? level : ((T.L - 1) < st.sliceLevelIdx ? (T.L - 1) : st.sliceLevelIdx);   // (:1034 "&& !useBalancedPlacement")
  for (; level < stop; level++) {
    // sortedDomains(lowerLevelDomains(currFitDomain))
    int m = 0, id0 = -1;
    for (int j = 0; j < ncur; j++) {
      const int d = s.cur[j], c0 = T.child_first[d], cn = T.child_cnt[d];
      for (int i = lane_id(); i < cn; i += WAVE) s.set[m + i] = c0 + i;
      if (ncur == 1) id0 = c0;   // the children of one domain are one id range: the sweep needs neither s.set nor a fence before it
      m += cn;
    }
    if (id0 < 0) wsync();
    nout = t_update_counts(k, s, m, ORD_PLAIN, st.count, st.leaderCount, st.sliceSize, st.unconstrained, true, s.nxt, 0, 0, id0);
    if (nout < 0) { *nfit = 0; return TFail{KQ_TAS_OK, 0, 0}; }
    for (int i = lane_id(); i < nout; i += WAVE) s.cur[i] = s.nxt[i];
    wsync();
    ncur = nout;
  }
  TPROF(k, 2);   // levels down to the slice level
  for (; level < T.L - 1; level++) {
    int32_t sliceSizeOnLevel = st.sliceSize;
    if (level >= st.sliceLevelIdx) { const int32_t sz = t_size_at(st, level + 1); sliceSizeOnLevel = sz > 0 ? sz : 1; }  // :1049-1057
    nout = 0;
    for (int j = 0; j < ncur; j++) {
      const int d = s.cur[j], c0 = T.child_first[d], cn = T.child_cnt[d];
      for (int i = lane_id(); i < cn; i += WAVE) s.set[i] = c0 + i;
      wsync();
      const int32_t dpc = s.pc[d], dlc = t_lc(s, d);
      nout = t_update_counts(k, s, cn, ORD_PLAIN, dpc, dlc, sliceSizeOnLevel, st.unconstrained, sliceSizeOnLevel > 1, s.nxt, nout, sliceSizeOnLevel);
      if (nout < 0) { *nfit = 0; return TFail{KQ_TAS_OK, 0, 0}; }
    }
    for (int i = lane_id(); i < nout; i += WAVE) s.cur[i] = s.nxt[i];
    wsync();
    ncur = nout;
  }
  *nfit = ncur;
  return TFail{KQ_TAS_OK, 0, 0};
}

// buildAssignment :1701 for `which` (0 workers: podCount, 1 leader: leaderCount) into the pool, leaves ascending
KQ_DEV void t_emit(const TK& k, const TState& s, int ncur, int which, int ps) {
  const TTopo& T = k.T; const TOut& O = k.O;
  const int lane = lane_id();
  // how many entries, and each entry's rank among them (the lists are short: pods of one podset)
  int total = 0;
  for (int base = 0; base < ncur; base += WAVE) {
    const int i = base + lane;
    const bool in = i < ncur && (which == 0 ? s.pc[s.cur[i]] : t_lc(s, s.cur[i])) > 0;
    total += popc64(wballot(in));
  }
  int pos = 0;
  if (lane == 0) {
    pos = total > 0 ? atomic_add_i32(O.pool_used, total) : 0;
    if (pos + total > O.pool_cap) { if (*O.error == 0) *O.error = KQ_ECAPACITY; O.dom_pos[ps] = 0; O.dom_n[ps] = 0; }
    else { O.dom_pos[ps] = pos; O.dom_n[ps] = total; }
    s.arr[0] = pos;  // broadcast through memory
  }
  wsync();
  pos = s.arr[0];
  wsync();
  if (pos + total > O.pool_cap) return;
  for (int i = lane; i < ncur; i += WAVE) {
    const int d = s.cur[i];
    const int32_t c = which == 0 ? s.pc[d] : t_lc(s, d);
    if (c <= 0) continue;
    int rank = 0;
    for (int j = 0; j < ncur; j++) {
      const int e = s.cur[j];
      if ((which == 0 ? s.pc[e] : t_lc(s, e)) > 0 && e < d) rank++;
    }
    O.pool_leaf[pos + rank] = d - T.leaf_base;
    O.pool_count[pos + rank] = c;
  }
  wsync();
}

// FindTopologyAssignmentsForFlavor :578 for workload w
template <bool LDS> KQ_DEV void t_workload_t(const TK& k, int slot, int w) {
  const TTopo& T = k.T; const TReq& Q = k.Q; const TOut& O = k.O;
  TState s0 = tas_state(k, slot);
  if (LDS && k.mail) { s0.coop = k.mail; s0.coop_min = k.mail->coop_min; }
  if (LDS) t_assume_lds(s0);
  TState s = s0;   // (the global rows of k_tas_find: the ...WithLeader arrays alias the plain ones while a class without a leader is being placed, below)
  const int lane = lane_id();
  const int p0 = Q.wl_off[w], p1 = Q.wl_off[w + 1];
  // more than one group => later groups see the usage assumed for the earlier ones (:654-656)
  int ngroups = 0;
  for (int p = p0; p < p1; p++) {
    bool firstOfGroup = true;
    if (Q.group[p] >= 0) for (int q = p0; q < p; q++) if (Q.group[q] == Q.group[p]) firstOfGroup = false;
    if (firstOfGroup) ngroups++;
  }
  const int sd0 = Q.seed_off ? Q.seed_off[w] : 0, sd1 = Q.seed_off ? Q.seed_off[w + 1] : 0;
  const bool seeded = sd1 > sd0;
  const bool track = ngroups > 1 || seeded;
  TPROF0();
  if (lane == 0) s.meta[0] = 0;
  if (track) { for (int i = lane; i < T.n_leaves * T.R; i += WAVE) s.assumed[i] = 0; wsync(); }
  if (seeded) {   // addAssumedUsage :734 of the previous pods (one lane per resource: the entries of a workload may name a leaf twice)
    for (int r = lane; r < T.R; r += WAVE)
      for (int j = sd0; j < sd1; j++) {
        const int64_t c = Q.seed_count[j];
        s.assumed[(size_t)Q.seed_leaf[j] * T.R + r] += Q.spr[(size_t)Q.seed_ps[j] * T.R + r] * c + (r == T.pods ? c : 0);
      }
    wsync();
  }
  bool failed = false, hasAssumed = seeded;
  for (int p = p0; p < p1; p++) {
    bool firstOfGroup = true;
    if (Q.group[p] >= 0) for (int q = p0; q < p; q++) if (Q.group[q] == Q.group[p]) firstOfGroup = false;
    if (!firstOfGroup) continue;
    int second = -1, members = 1;
    if (Q.group[p] >= 0) for (int q = p + 1; q < p1; q++) if (Q.group[q] == Q.group[p]) { if (second < 0) second = q; members++; }
    auto set_all = [&](int st, int a, int b) {
      if (lane == 0) {
        O.status[p] = st; O.op_a[p] = a; O.op_b[p] = b; O.dom_pos[p] = 0; O.dom_n[p] = 0;
        if (Q.group[p] >= 0) for (int q = p + 1; q < p1; q++) if (Q.group[q] == Q.group[p]) { O.status[q] = st; O.op_a[q] = a; O.op_b[q] = b; O.dom_pos[q] = 0; O.dom_n[q] = 0; }
      }
    };
    if (failed) { set_all(KQ_TAS_SKIPPED, 0, 0); continue; }
    if (members > 2) { set_all(KQ_TAS_UNSUPPORTED, 0, 0); failed = true; continue; }
    // findLeaderAndWorkers :668
    int workers = p, leader = -1;
    if (second >= 0) { leader = second; if (Q.count[leader] > Q.count[workers]) { leader = p; workers = second; } }
    TParams st;
    st.count = Q.count[workers]; st.leaderCount = leader >= 0 ? 1 : 0; st.sliceSize = Q.slice_size[workers];
    st.requestedLevelIdx = Q.level[workers]; st.sliceLevelIdx = Q.slice_level[workers];
    st.required = Q.kind[workers] == KQ_TAS_REQUIRED; st.unconstrained = Q.kind[workers] == KQ_TAS_UNCONSTRAINED;
    st.simulateEmpty = Q.sim_empty && Q.sim_empty[w]; st.hasLeader = leader >= 0; st.hasAssumed = hasAssumed;
    st.req = Q.spr + (size_t)workers * T.R; st.leaderReq = leader >= 0 ? Q.spr + (size_t)leader * T.R : nullptr;
    st.leafOk = !Q.leaf_ok ? nullptr : !Q.leaf_ok_idx ? Q.leaf_ok + (size_t)workers * T.n_leaves
              : Q.leaf_ok_idx[workers] >= 0 ? Q.leaf_ok + (size_t)Q.leaf_ok_idx[workers] * Q.leaf_ok_stride : nullptr;
    st.leafLo = Q.leaf_lo ? Q.leaf_lo[workers] : 0; st.leafHi = Q.leaf_lo ? Q.leaf_hi[workers] : 0;
    #pragma unroll
    for (int l = 0; l <= KQ_TAS_MAX_LEVELS; l++) st.sizeAt[l] = 0;
    st.nLayers = 0; st.layerLevel = nullptr; st.layerSize = nullptr;
    TFail f{KQ_TAS_OK, 0, 0};
    int ncur = 0;
    if (st.sliceSize <= 0) f = TFail{KQ_TAS_BAD_SLICE_SIZE, 0, 0};
    else if (st.requestedLevelIdx < 0 || st.requestedLevelIdx >= T.L || st.sliceLevelIdx < 0 || st.sliceLevelIdx >= T.L) f = TFail{KQ_TAS_NO_LEVEL, 0, 0};
    else if (st.requestedLevelIdx > st.sliceLevelIdx) f = TFail{KQ_TAS_SLICE_ABOVE, 0, 0};
    else {
      // buildSliceSizeAtLevel :1123 over the inner layers (layer 0 is sliceSize / sliceLevelIdx)
      const int nl = Q.n_layers ? Q.n_layers[workers] : 0;
      if (nl > 1) {
        const int32_t* ll = Q.layer_level + (size_t)workers * KQ_TAS_MAX_LEVELS;
        const int32_t* ls = Q.layer_size + (size_t)workers * KQ_TAS_MAX_LEVELS;
        int32_t prevSize = st.sliceSize; int prevLevel = st.sliceLevelIdx;
        for (int i = 1; i < nl && i < KQ_TAS_MAX_LEVELS && f.status == KQ_TAS_OK; i++) {
          if (ll[i] < 0 || ll[i] >= T.L) f = TFail{KQ_TAS_BAD_LAYER, i, 0};
          else if (ll[i] <= prevLevel) f = TFail{KQ_TAS_BAD_LAYER, i, 1};
          else if (ls[i] <= 0 || prevSize % ls[i] != 0) f = TFail{KQ_TAS_BAD_LAYER, i, 2};
          else {
            #pragma unroll
            for (int l = 0; l <= KQ_TAS_MAX_LEVELS; l++) if (l > prevLevel && l <= ll[i]) st.sizeAt[l] = ls[i];
            prevSize = ls[i]; prevLevel = ll[i];
          }
        }
        if (f.status == KQ_TAS_OK) { st.nLayers = nl < KQ_TAS_MAX_LEVELS ? nl : KQ_TAS_MAX_LEVELS; st.layerLevel = ll; st.layerSize = ls; }
      }
    }
    if (f.status == KQ_TAS_OK) {
      const int cls = (k.C.n > 0 && !seeded) ? k.C.wl_class[w] : -1;   // (a seeded workload's phase 1 sees its own assumed usage: no shared table)
      bool have = false;
      if (!LDS) {
        // a class without a leader: podCountWithLeader / sliceCountWithLeader hold the plain counts in its table, the placement never writes
        // them apart and leaderCount is 0 everywhere — the slot keeps two arrays of the five (what the LDS state of k_process_tas does):
        // 33 KB instead of 83 KB copied per class change at 4096 leaves, 8 B instead of 20 B read per domain and sweep
        // (k_tas_find's classes; the cycle's per-class slots are patched in place by its class updates and keep every array. A private
        // phase 1 without a leader fills the same values into both pairs — leaf :352, roll-up with minPodDiff = 0 — and is used by this
        // placement alone: two arrays there as well, 33 KB instead of 83 KB written per placement of k_nominate_tas at 4096 leaves. Not with
        // the balanced placement, whose saved copies hold five arrays.)
        const bool nl = cls >= 0 ? (k.C.leader && k.C.leader[cls] < 0) : (leader < 0 && !T.balanced);
        s.nolead = nl; s.pcwl = nl ? s0.pc : s0.pcwl; s.scwl = nl ? s0.sc : s0.scwl; s.lc = nl ? nullptr : s0.lc;
      }
      if (cls >= 0) {
        // start from the class's phase-1 table; the slot keeps it between workloads of the same class (the LDS copy is filled every
        // time: the table moves with every AddUsage, and the copy is one round trip of the whole workgroup)
        if (LDS) {
          t_class_to_lds(k, s, cls);
          if (lane == 0) { s.meta[1] = 1; s.meta[2] = cls; }   // (nothing to put back afterwards)
        } else if (s.meta[2] != cls || s.meta[1]) {
          wsync();
          for (int d = lane; d < T.D; d += WAVE) {
            const size_t o = (size_t)cls * T.D + d;
            s.pc[d] = k.C.pc[o]; s.sc[d] = k.C.sc[o];
            if (!s.nolead) { s.pcwl[d] = k.C.pcwl[o]; s.scwl[d] = k.C.scwl[o]; s.lc[d] = k.C.lc[o]; }
          }
          if (lane == 0) { s.meta[1] = 0; s.meta[2] = cls; }
        }
        if (lane == 0) { s.meta[0] = 0; atomic_add_i64(O.bytes, k.C.bytes[cls]); }
        wsync();
        have = true;
      } else if (lane == 0) {
        s.meta[2] = -1;
      }
      TPROF(k, 4);   // everything of t_workload before the search
      f = t_find_assignment(k, s, st, &ncur, have);
      TPROF(k, 5);   // t_find_assignment (0-3 are inside)
      if (f.status == KQ_TAS_NOT_FIT && st.nLayers > 0)
        f = t_not_fit_layers(k, s, st, ncur, O.layer_fit ? O.layer_fit + (size_t)workers * KQ_TAS_MAX_LEVELS : nullptr,
                             O.layer_fit && leader >= 0 ? O.layer_fit + (size_t)leader * KQ_TAS_MAX_LEVELS : nullptr);
    }
    if (f.status != KQ_TAS_OK) { set_all(f.status, f.a, f.b); failed = true; continue; }
    if (lane == 0) { O.status[workers] = KQ_TAS_OK; O.op_a[workers] = 0; O.op_b[workers] = 0; if (leader >= 0) { O.status[leader] = KQ_TAS_OK; O.op_a[leader] = 0; O.op_b[leader] = 0; } }
    // addAssumedUsage :734 before the leader's copies of the counts are consumed
    if (track) {
      for (int i = lane; i < ncur; i += WAVE) {
        const int d = s.cur[i], leaf = d - T.leaf_base;
        for (int r = 0; r < T.R; r++) {
          int64_t add = Q.spr[(size_t)workers * T.R + r] * (int64_t)s.pc[d] + (r == T.pods ? s.pc[d] : 0);
          if (leader >= 0) add += Q.spr[(size_t)leader * T.R + r] * (int64_t)t_lc(s, d) + (r == T.pods ? t_lc(s, d) : 0);
          s.assumed[(size_t)leaf * T.R + r] += add;
        }
      }
      hasAssumed = true;
      wsync();
    }
    if (leader >= 0) t_emit(k, s, ncur, 1, leader);
    t_emit(k, s, ncur, 0, workers);
    TPROF(k, 6);   // status words + buildAssignment
  }
  // put the class table's values back into the domains phase 2 consumed
  if (s.meta[2] >= 0 && !s.meta[1]) {
    const int cls = s.meta[2], nlog = s.meta[0];
    wsync();
    for (int i = lane; i < nlog; i += WAVE) {
      const int d = s.log[i];
      const size_t o = (size_t)cls * T.D + d;
      s.pc[d] = k.C.pc[o]; s.sc[d] = k.C.sc[o];
      if (!s.nolead) { s.pcwl[d] = k.C.pcwl[o]; s.scwl[d] = k.C.scwl[o]; s.lc[d] = k.C.lc[o]; }
    }
    wsync();
  }
}
KQ_DEV void t_workload(const TK& k, int slot, int w) {
  if (k.lds) t_workload_t<true>(k, slot, w); else t_workload_t<false>(k, slot, w);
}
// phase 1 of request class c into the class tables
KQ_DEV void t_class(const TK& k, int c) {
  const TTopo& T = k.T; const TReq& Q = k.Q;
  TState s{};
  s.pc = k.C.pc + (size_t)c * T.D; s.sc = k.C.sc + (size_t)c * T.D; s.pcwl = k.C.pcwl + (size_t)c * T.D;
  s.scwl = k.C.scwl + (size_t)c * T.D; s.lc = k.C.lc + (size_t)c * T.D;
  const int workers = k.C.workers[c], leader = k.C.leader[c];
  TParams st{};
  st.count = Q.count[workers]; st.leaderCount = leader >= 0 ? 1 : 0; st.sliceSize = Q.slice_size[workers];
  st.sliceLevelIdx = Q.slice_level[workers]; st.requestedLevelIdx = Q.level[workers];
  st.simulateEmpty = k.C.sim_empty[c] != 0; st.hasAssumed = false;
  st.req = Q.spr + (size_t)workers * T.R; st.leaderReq = leader >= 0 ? Q.spr + (size_t)leader * T.R : nullptr;
  st.leafOk = nullptr; st.leafLo = 0; st.leafHi = 0;
  if (lane_id() == 0) k.C.bytes[c] = 0;
  wsync();
  if (st.sliceSize <= 0 || st.sliceLevelIdx < 0 || st.sliceLevelIdx >= T.L) return;  // the workloads of the class fail before phase 1
  t_fill_in_counts(k, s, st, k.C.bytes + c);
}

}  // namespace kq

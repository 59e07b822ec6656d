// kq_fs.hpp — the fair-sharing victim search (preemption.go:381-631) with its whole working state in the workgroup's LDS.
//
// fair_search (kq_device.hpp) keeps a private copy of the tree's usage plane in HBM scratch and walks it with one dependent
// global access after the other: at cfg 4f (one tree of 1111 nodes, 40 000 admitted rows, ~2400 pops per search) a pop costs
// ~50 000 cycles, 60 % of it in snapshot.RemoveWorkload / AddWorkload chains (profiles/r02k_prof_fair_1000.txt). Here:
//
//  * DRS only reads the per-node borrowed sums (psum[node][resource], count of borrowed cells): they live in LDS together with
//    the CACHED share of every node (dval: the quantity CompareDRS orders by), refreshed for the <= 4 nodes of a row's path
//    after every usage change. A tournament level of nextTarget (ordering.go:144-226) is then one LDS read per child and a
//    wave arg-max instead of nR fp64 divisions per child and a serial fold.
//  * Candidate queues are bitmaps over a static "position order" (kq_prep.hpp FsScan/FsApply: candidates grouped by
//    ClusterQueue, inside a ClusterQueue in pop order): class 1 = candidates, class 2 = retryCandidates. PopWorkload is a
//    find-first-set, findCandidates a ballot per 64 positions; nothing is rescanned.
//  * Usage cells: the columns (flavor-resources) of the flavors the preemptor needs are cached in LDS for all nodes — with
//    one flavor per admitted workload those are the only cells a search writes. Other columns fall back to the private HBM
//    plane, copied lazily the first time a row touches one.
//  * A row's Remove/AddWorkload gathers every constant it needs (localQuota / SubtreeQuota of the <= 4 flavor-resources x
//    <= 4 path levels, lendable and weight of the path nodes) in ONE round of independent loads from tables in tree-node
//    order; the chain of removeUsage / addUsage (resource_node.go:144-165) then runs on registers and LDS.
//
// Same sequence of operations as fair_search, hence the same targets, the same private state where callers read it, and the
// same algorithmic byte count (an "evaluation" of a cached share is charged what computing it costs the reference). Trees or
// searches outside the preconditions (deeper than FS_LV levels, rows with more than FS_RFR flavor-resources, non-plain
// amounts, more than FS_PCN preemptor cells) return false and fair_search runs. Bit-exact: tests run both and compare.
#pragma once

namespace kq {

constexpr int FS_PCN = 64;      // (in-use slot, level) cells of the preemptor's context
constexpr int FS_NCMAX = 16;    // usage columns cached per search
constexpr int FS_RC = FS_RFR * FS_LV;
constexpr int FS_BQ = 64;       // ClusterQueues of one cohort handled as a batch (fs_batch_first / fs_batch_second)
constexpr int FS_BC = 128;      // candidates evaluated per batch

struct Fs {
  const K* k; Wave* w; Search* s;
  int nn, nqs, ncoh, nR, nfr, n0, q0, row0, nrows, mw, nc, npc, wli, plen, tree;
  bool wcopied;
  // state
  int64_t* psum; double* dval; int32_t* ppos; uint8_t* nflag; uint64_t *m1, *m2, *mq; int8_t* colslot; int64_t* col;
  // static tables of the tree
  int32_t* posoff; int16_t *kid, *koff, *knc, *knh, *c0, *c1, *par; int8_t* plv;   // plv: level of the node on the preemptor's path, -1 = not on it
  // context of the row being applied / of the preemptor
  int32_t* rc_ptr; int64_t *rc_lq, *rc_sqb, *rc_lend; double* rc_wt;   // rc_ptr / pc_ptr: cell codes, see fs_cell
  int64_t* td; int32_t* tp; double* tx; uint64_t* tout;
  // what an evaluation (commit = false) computed, kept for the commit that follows it when the candidate passes: new cell values and
  // touched levels per chain, new borrowed sums per (level, resource), new borrowed-cell counts / shares / zero-weight flags per level
  int64_t *ev_nv, *ev_sum; int32_t *ev_t, *ev_np; double* ev_v; uint8_t* ev_z;   // staging of one row operation: borrowed-amount deltas / borrowed-cell count deltas per (flavor-resource, level), share terms per (level, resource), result
  int32_t* pc_ptr; int64_t *pc_lq, *pc_sq, *pc_sqb, *pc_bl, *pc_lend; double* pc_wt; int32_t* pc_u;
  int32_t* tpos;
  // per ClusterQueue of the tree: the almost-LCAs with the preemptor (fs_lcas), filled once per search. (Round 5 also tried the candidate's
  // record and the quota constants of its path in ONE round trip through an LDS landing area instead of fs_row_load + fs_row_ctx's two
  // dependent ones: the staging cost more than the round trip it saved — "row load + context" 1 086 -> 1 329 ms of the cfg 4f process
  // kernel, profiles/r05d_prof_fair_cfg4f_process_only.txt — and was taken out again. So were two more attempts on the same visit, each
  // measured with the segment timers of the -DKQ_PROF build: nextTarget's answer kept per cohort until something below it moves
  // ("ordering.next" 1 637 -> 1 931 ms: the bookkeeping of the kept answers and their invalidation cost more than the scans it saved,
  // profiles/r05e_prof_fair_cfg4f_nexttarget_cache.txt), and a per-row record of the quota constants so that the row's record and its
  // constants leave in one round trip (A/B on one box: 1 208 vs 1 216 ms, profiles/r05g_prof_fair_cfg4f_rowc_{on,off}.txt — the two
  // dependent loads were never the cost of that segment).)
  int32_t* cq_lca;
  // the first tcap targets (row, reason, position) live in what LDS the state left over: a target pushed or moved is then no global
  // store, and every fence of the walk waits for the wave's outstanding global stores (fs_tget / fs_tset; flushed to trow / treason /
  // tpos when the search ends)
  int tcap; int32_t *lt_row, *lt_pos; uint8_t* lt_rs;
  // batch of one cohort's ClusterQueues (fs_batch_*): unsorted (child order) and sorted (visiting order) views, candidate list
  int16_t *bq_c, *bq_ord; uint64_t *bq_k, *bq_h; uint8_t* bq_z;
  int16_t *br_c, *br_ap, *br_at, *br_n, *br_off; uint8_t* br_fl;
  int32_t* cl_pos; uint8_t* cl_rk;
  // fields of the Search (it lives in the caller's frame)
  int64_t* W; const int64_t* usage; const uint8_t* removed; int32_t* trow; uint8_t* treason;
};
struct FsRow { int64_t qty[FS_RFR]; int fr[FS_RFR]; int res[FS_RFR]; int lp[FS_LV]; int plen, row, cbytes, rowbytes, nent; uint32_t hkey; };   // nent: entries to look at (CS_RFR, a wide row FS_RFR)

// every array of the search sits in the workgroup's LDS (fs_setup checks it): tell the compiler, so that the accesses are ds_* instead
// of flat_* — in every function that receives the Fs by reference (the facts do not travel across a call)
KQ_DEV void fs_assume_lds(const Fs& f) {
#if !defined(KQ_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
  #define FS_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
  FS_LDS(f.psum); FS_LDS(f.dval); FS_LDS(f.ppos); FS_LDS(f.nflag); FS_LDS(f.m1); FS_LDS(f.m2); FS_LDS(f.mq); FS_LDS(f.colslot); FS_LDS(f.col);
  FS_LDS(f.posoff); FS_LDS(f.kid); FS_LDS(f.koff); FS_LDS(f.knc); FS_LDS(f.knh); FS_LDS(f.c0); FS_LDS(f.c1); FS_LDS(f.par); FS_LDS(f.plv);
  FS_LDS(f.rc_ptr); FS_LDS(f.rc_lq); FS_LDS(f.rc_sqb); FS_LDS(f.rc_lend); FS_LDS(f.rc_wt); FS_LDS(f.td); FS_LDS(f.tp); FS_LDS(f.tx); FS_LDS(f.tout);
  FS_LDS(f.ev_nv); FS_LDS(f.ev_sum); FS_LDS(f.ev_t); FS_LDS(f.ev_np); FS_LDS(f.ev_v); FS_LDS(f.ev_z);
  FS_LDS(f.pc_ptr); FS_LDS(f.pc_lq); FS_LDS(f.pc_sq); FS_LDS(f.pc_sqb); FS_LDS(f.pc_bl); FS_LDS(f.pc_lend); FS_LDS(f.pc_wt); FS_LDS(f.pc_u);
  FS_LDS(f.bq_c); FS_LDS(f.bq_ord); FS_LDS(f.bq_k); FS_LDS(f.bq_h); FS_LDS(f.bq_z); FS_LDS(f.br_c); FS_LDS(f.br_ap); FS_LDS(f.br_at); FS_LDS(f.br_n); FS_LDS(f.br_off);
  FS_LDS(f.br_fl); FS_LDS(f.cl_pos); FS_LDS(f.cl_rk);
  FS_LDS(f.cq_lca);
  if (f.tcap > 0) { FS_LDS(f.lt_row); FS_LDS(f.lt_pos); FS_LDS(f.lt_rs); }
  #undef FS_LDS
#endif
  (void)f;
}
#ifdef KQ_HOST_EMU
static inline
#else
__host__ __device__ inline
#endif
size_t fs_bytes(int nn, int nqs, int nR, int nfr, int mw, int ncols) {
  auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const int ncoh = nn - nqs;
  size_t b = 0;
  b += al(FS_RC * 8) * 3 + al(FS_LV * KQ_MAXR * 8) + al(FS_LV * 8);                               // row context
  b += al(FS_RC * 8) + al(FS_RC * 4) + al(FS_LV * KQ_MAXR * 8) + al(4 * 8);                        // staging of a row operation
  b += al(FS_RC * 8) + al(FS_LV * KQ_MAXR * 8) + al(FS_RFR * 4) + al(FS_LV * 4) + al(FS_LV * 8) + al(FS_LV);   // an evaluation's results (ev_*)
  b += al(FS_PCN * 8) * 5 + al(FS_PCN * 4) + al(FS_LV * KQ_MAXR * 8) + al(FS_LV * 8);              // preemptor context
  b += al(FS_NCMAX * 8);                                                                            // column pointers
  b += al(nn) * 2 + al((size_t)nn * 8) + al((size_t)nn * 4) + al((size_t)nn * 2) * 4;              // nflag plv dval ppos c0 c1 kid par
  b += al((size_t)(ncoh > 0 ? ncoh : 1) * 2) * 3 + al((size_t)(nqs + 1) * 4) + al(nfr);            // koff knc knh posoff colslot
  b += al((size_t)nn * nR * 8) + al((size_t)mw * 8) * 2;                                           // psum m1 m2
  b += al((size_t)nn * 8) * (size_t)ncols;
  b += al(FS_BQ * 2) * 2 + al(FS_BQ * 8) * 2 + al(FS_BQ);                                          // bq_c bq_ord bq_k bq_h bq_z
  b += al(FS_BQ * 2) * 4 + al((FS_BQ + 1) * 2) + al(FS_BQ);                                        // br_*
  b += al(FS_BC * 4) + al(FS_BC);                                                                  // cl_*
  b += al((size_t)nqs * 4);   // cq_lca
  return b + 256;
}

// Under C.fs_plain every finite amount is below 2^50; "unlimited" constants are clipped to 2^60 when they are gathered, so that
// the chains below run on plain int64 arithmetic: a clipped value stays far above every finite one through the <= FS_LV
// additions / subtractions of a chain, hence every comparison and every finite result equals what resources.Amount's
// saturating operations (a_add / a_sub) give — at a tenth of the instructions (a wave64 VALU instruction costs 4 cycles
// whatever the number of active lanes, and these chains are nothing but 64-bit VALU work).
constexpr int64_t FS_CAP = (int64_t)1 << 60;
KQ_DEV int64_t fs_cap(int64_t v) { return v > FS_CAP ? FS_CAP : v; }
KQ_DEV uint64_t fs_bits(double v) { union { double d; uint64_t u; } x; x.d = v; return x.u; }
// total order of Go's cmp.Compare on float64 (NaN lowest, -0 == +0) as an unsigned key
KQ_DEV uint64_t fs_okey(double v) {
  if (v != v) return 0;
  if (v == 0) v = 0.0;
  const uint64_t u = fs_bits(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
// CompareDRS (fair_sharing.go:112-123) on (zero-weight-borrows, key of ratio / PreciseWeightedShare)
KQ_DEV int fs_cmp(int za, uint64_t ka, int zb, uint64_t kb) {
  if (za != zb) return za ? 1 : -1;
  return ka < kb ? -1 : (ka > kb ? 1 : 0);
}
KQ_DEV uint64_t wmax_u64(uint64_t v) { return ~wmin_u64(~v); }
template <class T> KQ_DEV T fs_sel4(const T* a, int i) { T v = a[0]; if (i == 1) v = a[1]; if (i == 2) v = a[2]; if (i == 3) v = a[3]; return v; }
template <class T> KQ_DEV T fs_sel8(const T* a, int i) { const T lo = fs_sel4(a, i & 3), hi = fs_sel4(a + 4, i & 3); return i < 4 ? lo : hi; }
static_assert(CS_RFR == 4 && FS_RFR == 8 && FS_LV == 4, "fs_sel4 on the levels and on a record's entries, fs_sel8 on a row's entries");
// the usage entries of the row of a position record, as the search sees them: the record's own, then a wide row's other ones
struct FsEnt { int64_t qty[FS_RFR]; int fr[FS_RFR]; int res[FS_RFR]; };   // res = the entry's resource index (fr % nR), -1 = unused entry
KQ_DEV FsEnt fs_entries(const DSnap& S, const FsApply& a) {
  FsEnt e;
  #pragma unroll
  for (int u = 0; u < CS_RFR; u++) {
    e.qty[u] = a.qty[u]; e.fr[u] = a.fr[u]; e.res[u] = a.fr[u] >= 0 ? (int)a.res[u] : -1;
    e.qty[CS_RFR + u] = 0; e.fr[CS_RFR + u] = -1; e.res[CS_RFR + u] = -1;
  }
  if (a.wide) {
    const AdmRecX x = S.adm_recx[a.row];
    #pragma unroll
    for (int u = 0; u < CS_RFX; u++) { e.qty[CS_RFR + u] = x.qty[u]; e.fr[CS_RFR + u] = x.fr[u]; e.res[CS_RFR + u] = x.fr[u] >= 0 ? x.fr[u] % S.nR : -1; }
  }
  return e;
}

// first set position of bitmap m inside [a, b), -1 if none
KQ_DEV int fs_first(const uint64_t* m, int a, int b) {
  if (a >= b) return -1;
  const int wa = a >> 6, wb = (b - 1) >> 6;
  for (int wi = wa; wi <= wb; wi++) {
    uint64_t x = m[wi];
    if (wi == wa) x &= ~0ull << (a & 63);
    if (wi == wb) { const int e = (b - 1) & 63; if (e < 63) x &= (2ull << e) - 1; }
    if (x) return wi * 64 + ffs64(x);
  }
  return -1;
}
// nflag: 1 pruned (prunedClusterQueues / prunedCohorts), 2 zero weight but borrowing, 4 the ClusterQueue contributes candidates,
//        8 the ClusterQueue's queue of the class under iteration is not empty
KQ_DEV bool fs_cq_has(const Fs& f, int li) { return (f.nflag[li] & 8) != 0; }
KQ_DEV void fs_has_update(const Fs& f, int li) {  // one lane
  const bool has = fs_first(f.mq, f.posoff[li], f.posoff[li + 1]) >= 0;
  f.nflag[li] = (uint8_t)((f.nflag[li] & ~8) | (has ? 8 : 0));
}

// A usage cell of the private state is addressed by a code: >= 0 an index into the cached columns (LDS), < 0 the complement of
// an index into the private HBM plane.
KQ_DEV int fs_cell(const Fs& f, int li, int fr) {
  const int sl = f.colslot[fr];
  return sl >= 0 ? sl * f.nn + li : ~(li * f.nfr + fr);
}
KQ_DEV int64_t fs_ld(const Fs& f, int code) { return code >= 0 ? f.col[code] : f.W[~code]; }
KQ_DEV void fs_st(const Fs& f, int code, int64_t v) { if (code >= 0) f.col[code] = v; else f.W[~code] = v; }
struct FsTgt { int32_t row, pos; uint8_t reason; };
KQ_DEV FsTgt fs_tget(const Fs& f, int i) { return i < f.tcap ? FsTgt{f.lt_row[i], f.lt_pos[i], f.lt_rs[i]} : FsTgt{f.trow[i], f.tpos[i], f.treason[i]}; }
KQ_DEV int fs_tpos(const Fs& f, int i) { return i < f.tcap ? f.lt_pos[i] : f.tpos[i]; }
KQ_DEV void fs_tset(const Fs& f, int i, const FsTgt& t) {
  if (i < f.tcap) { f.lt_row[i] = t.row; f.lt_pos[i] = t.pos; f.lt_rs[i] = t.reason; }
  else { f.trow[i] = t.row; f.tpos[i] = t.pos; f.treason[i] = t.reason; }
}
// the search is over: the nt targets it keeps, where the callers read them
KQ_DEV void fs_tflush(const Fs& f, int nt) {
  const int m = nt < f.tcap ? nt : f.tcap;
  for (int i = lane_id(); i < m; i += WAVE) { f.trow[i] = f.lt_row[i]; f.treason[i] = f.lt_rs[i]; }
  wsync();
}
// the private HBM plane serves the columns that are not cached: copied from the search's starting state on first use
KQ_DEV void fs_ensure_w(Fs& f) {
  if (f.wcopied) return;
  const DSnap& S = f.k->S;
  for (int i = lane_id(); i < f.nn * f.nfr; i += WAVE) f.W[i] = f.usage[ix(S, S.tree_nodes[f.n0 + i / f.nfr], i % f.nfr)];
  wsync();
  f.wcopied = true;
}

// cached share of node li from its borrowed sums: dominantResourceShare :149-182 + PreciseWeightedShare :92
KQ_DEV void fs_refresh(const Fs& f, int li, const int64_t* lend, double weight) {
  double ratio = 0;
  for (int r = 0; r < f.nR; r++) {
    const int64_t sum = f.psum[(size_t)li * f.nR + r];
    if (sum <= 0) continue;
    const int64_t lr = lend[r];
    if (lr > 0) { const double x = (double)sum * 1000.0 / (double)lr; if (x > ratio) ratio = x; }
  }
  const bool zwb = weight == 0 && ratio != 0;
  double v = ratio;
  if (!zwb) v = ratio == 0 ? 0.0 : (weight == 0 ? __builtin_inf() : ratio / weight);
  f.dval[li] = v;
  f.nflag[li] = (uint8_t)((f.nflag[li] & ~2) | (zwb ? 2 : 0));
}
KQ_DEV int64_t fs_cost(const Fs& f, int li) { return (int64_t)f.c0[li] + (f.ppos[li] > 0 ? (int64_t)f.c1[li] : 0); }
KQ_DEV bool fs_pos_inf(const Fs& f, int li) { return (f.nflag[li] & 2) || f.dval[li] == __builtin_inf(); }

// the candidate scan streams the whole tree once per search: keep it from evicting the records the pops re-read (L2 is 4 MB per XCD)
KQ_DEV FsScan fs_scan_load(const FsScan* p) {
#if defined(KQ_HOST_EMU) || !defined(__HIP_DEVICE_COMPILE__)
  return *p;
#else
  typedef long long v2 __attribute__((ext_vector_type(2)));
  const v2* q = (const v2*)p;
  const v2 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
  FsScan r;
  r.prio = a.x; r.qts = a.y;
  const uint64_t w0 = (uint64_t)b.x, w1 = (uint64_t)b.y;
  r.fr[0] = (int16_t)w0; r.fr[1] = (int16_t)(w0 >> 16); r.fr[2] = (int16_t)(w0 >> 32); r.fr[3] = (int16_t)(w0 >> 48);
  r.row = (int32_t)w1; r.cql = (int16_t)(w1 >> 32); r.cbytes = (uint16_t)(w1 >> 48);
  return r;
#endif
}
// ---- the row under Remove/AddWorkload ---------------------------------------------------------------------------------------
KQ_DEV FsRow fs_row_load(const Fs& f, int p) {
  const DSnap& S = f.k->S;
  const FsApply ap = S.fs_apply[(size_t)f.row0 + p];
  FsRow r;
  const FsEnt en = fs_entries(S, ap);
  #pragma unroll
  for (int e = 0; e < FS_RFR; e++) { r.qty[e] = en.qty[e]; r.fr[e] = en.fr[e]; r.res[e] = en.res[e]; }
  r.nent = ap.wide ? FS_RFR : CS_RFR;
  #pragma unroll
  for (int l = 0; l < FS_LV; l++) r.lp[l] = ap.lp[l];
  r.plen = ap.plen; r.row = ap.row; r.cbytes = ap.cbytes; r.hkey = ap.hkey;
  r.rowbytes = 16 * (int)ap.plen * (((int)ap.cbytes - 32) / 12);
  return r;
}
// constants of every (flavor-resource, level) cell of the row and of its path nodes: one round of independent loads
KQ_DEV void fs_row_ctx(Fs& f, const FsRow& r) {
  const DSnap& S = f.k->S;
  bool miss = false;
  #pragma unroll
  for (int e = 0; e < FS_RFR; e++) if (r.fr[e] >= 0 && f.colslot[r.fr[e]] < 0) miss = true;
  if (miss) fs_ensure_w(f);
  for (int c = lane_id(); c < r.nent * FS_LV; c += WAVE) {
    const int u = c / FS_LV, i = c % FS_LV;
    const int fr = fs_sel8(r.fr, u), li = fs_sel4(r.lp, i);
    if (fr < 0 || i >= r.plen) continue;
    const FsQ q = S.fs_q[(size_t)(f.n0 + li) * f.nfr + fr];
    f.rc_lq[c] = fs_cap(q.lq); f.rc_sqb[c] = fs_cap(q.sqb);
    f.rc_ptr[c] = fs_cell(f, li, fr);
  }
  for (int j = lane_id(); j < FS_LV * KQ_MAXR; j += WAVE) {
    const int i = j / KQ_MAXR, rr = j % KQ_MAXR;
    if (i >= r.plen || rr >= f.nR) continue;
    f.rc_lend[j] = S.fs_lend[(size_t)(f.n0 + fs_sel4(r.lp, i)) * f.nR + rr];
  }
  for (int i = lane_id(); i < r.plen; i += WAVE) f.rc_wt[i] = S.fs_weight[f.n0 + fs_sel4(r.lp, i)];
  wsync();
}
// one usage chain on gathered cells: removeUsage (resource_node.go:156-165) / addUsage (:144-152) of `val` along `plen` levels.
// Outputs per level: the new value, what the node's borrowed amount (max(0, usage - SubtreeQuota)) and its count of borrowed
// cells change by. `write`: store the new values.
struct FsChainOut { int64_t d[FS_LV]; int dp[FS_LV]; };
KQ_DEV FsChainOut fs_chain(const Fs& f, const int32_t* ptr, const int64_t* lq, const int64_t* sqb, int plen, int64_t val, bool add, bool write,
                           int64_t* nv_out = nullptr, int32_t* t_out = nullptr) {
  int64_t v[FS_LV], nv[FS_LV];
  bool t[FS_LV];
  #pragma unroll
  for (int i = 0; i < FS_LV; i++) { v[i] = 0; nv[i] = 0; t[i] = false; if (i < plen) v[i] = fs_ld(f, ptr[i]); }
  bool go = true;
  #pragma unroll
  for (int i = 0; i < FS_LV; i++) {
    if (!go || i >= plen) continue;
    const int64_t uu = v[i];
    t[i] = true;
    if (add) {
      const int64_t la = i64max(0, lq[i] - uu);
      nv[i] = uu + val;
      if (i + 1 < plen && val > la) val = val - la; else go = false;
    } else {
      const int64_t stored = uu - lq[i];
      nv[i] = uu - val;
      if (stored <= 0 || i + 1 >= plen) go = false; else val = i64min(val, stored);
    }
  }
  FsChainOut o;
  #pragma unroll
  for (int i = 0; i < FS_LV; i++) {
    o.d[i] = 0; o.dp[i] = 0;
    if (!t[i]) continue;
    const int64_t ob = i64max(0, v[i] - sqb[i]), nb = i64max(0, nv[i] - sqb[i]);
    o.d[i] = nb - ob;
    o.dp[i] = (nb > 0 ? 1 : 0) - (ob > 0 ? 1 : 0);
    if (write) fs_st(f, ptr[i], nv[i]);
  }
  if (nv_out) {
    int tm = 0;
    #pragma unroll
    for (int i = 0; i < FS_LV; i++) { nv_out[i] = nv[i]; if (t[i]) tm |= 1 << i; }
    *t_out = tm;
  }
  return o;
}
// The borrowed sums / cached shares of `plen` path nodes after the per-chain deltas staged in f.td / f.tp (nch chains, chain u
// belongs to resource res(u)): one lane per (level, resource) adds the deltas and evaluates the share term, one lane per level
// folds the terms (dominantResourceShare :149-182). commit = false: nothing is stored, only the share of node `at` is returned
// (ComputeTargetShareAfterRemoval target.go:67-73 without touching the state). Returns z | borrowing << 1 of `at`, *key = its key.
template <class RES>
KQ_DEV int fs_nodes_update(const Fs& f, const int* lp, int plen, const int64_t* lend, const double* wt, int nch, const RES& res, bool commit, int at, uint64_t* key) {
  const int lane = lane_id();
  for (int j = lane; j < FS_LV * KQ_MAXR; j += WAVE) {
    const int i = j / KQ_MAXR, rr = j % KQ_MAXR;
    if (i >= plen || rr >= f.nR) continue;
    const int li = fs_sel4(lp, i);
    int64_t d = 0;
    for (int u = 0; u < nch; u++) if (res(u) == rr) d += f.td[u * FS_LV + i];
    const int64_t sum = f.psum[(size_t)li * f.nR + rr] + d;
    if (commit && d != 0) f.psum[(size_t)li * f.nR + rr] = sum;
    if (!commit) f.ev_sum[j] = sum;
    double x = 0;
    if (sum > 0) { const int64_t lr = lend[i * KQ_MAXR + rr]; if (lr > 0) x = (double)sum * 1000.0 / (double)lr; }
    f.tx[i * KQ_MAXR + rr] = x;
  }
  wsync_lds();
  for (int i = lane; i < plen; i += WAVE) {
    const int li = fs_sel4(lp, i);
    int dp = 0;
    for (int u = 0; u < nch; u++) dp += f.tp[u * FS_LV + i];
    const int np = f.ppos[li] + dp;
    double ratio = 0;
    for (int r = 0; r < f.nR; r++) { const double x = f.tx[i * KQ_MAXR + r]; if (x > ratio) ratio = x; }
    const double weight = wt[i];
    const bool zwb = weight == 0 && ratio != 0;
    double v = ratio;
    if (!zwb) v = ratio == 0 ? 0.0 : (weight == 0 ? __builtin_inf() : ratio / weight);
    if (commit) {
      if (dp) f.ppos[li] = np;
      f.dval[li] = v;
      f.nflag[li] = (uint8_t)((f.nflag[li] & ~2) | (zwb ? 2 : 0));
    } else { f.ev_np[i] = np; f.ev_v[i] = v; f.ev_z[i] = zwb ? 1 : 0; }
    if (li == at) { f.tout[0] = fs_okey(v); f.tout[1] = (uint64_t)((zwb ? 1 : 0) | (np > 0 ? 2 : 0)); }
  }
  wsync_lds();
  if (at >= 0) { *key = f.tout[0]; return (int)f.tout[1]; }
  return 0;
}
struct FsRowRes { int r[FS_RFR]; KQ_MDEV int operator()(int u) const { return fs_sel8(r, u); } };
// snapshot.RemoveWorkload / AddWorkload of the row whose context is loaded. commit = false evaluates the share of node `at`
// after the operation without changing anything.
KQ_DEV int fs_row_apply(const Fs& f, const FsRow& r, bool add, bool commit, bool count, int at = -1, uint64_t* key = nullptr) {
  for (int u = lane_id(); u < r.nent; u += WAVE) {
    const int fr = fs_sel8(r.fr, u);
    FsChainOut o;
    #pragma unroll
    for (int i = 0; i < FS_LV; i++) { o.d[i] = 0; o.dp[i] = 0; }
    if (fr >= 0) o = fs_chain(f, f.rc_ptr + u * FS_LV, f.rc_lq + u * FS_LV, f.rc_sqb + u * FS_LV, r.plen, fs_sel8(r.qty, u), add, commit,
                              commit ? nullptr : f.ev_nv + u * FS_LV, commit ? nullptr : f.ev_t + u);
    else if (!commit) f.ev_t[u] = 0;
    #pragma unroll
    for (int i = 0; i < FS_LV; i++) { f.td[u * FS_LV + i] = o.d[i]; f.tp[u * FS_LV + i] = o.dp[i]; }
  }
  wsync();
  const int z = fs_nodes_update(f, r.lp, r.plen, f.rc_lend, f.rc_wt, r.nent, FsRowRes{{r.res[0], r.res[1], r.res[2], r.res[3], r.res[4], r.res[5], r.res[6], r.res[7]}}, commit, at, key);
  if (count && lane_id() == 0) f.w->bytes += r.rowbytes;
  return z;
}

// RemoveWorkload of the row whose removal has just been EVALUATED (fs_row_apply with commit = false, same row, nothing in between): the
// evaluation computed every new value — cells, borrowed sums, counts, shares — and only kept the share it was asked for; the commit
// writes what it left in ev_* instead of computing it a second time.
KQ_DEV void fs_row_commit_evaluated(const Fs& f, const FsRow& r) {
  const int lane = lane_id();
  for (int u = lane; u < r.nent; u += WAVE) {
    const int tm = f.ev_t[u];
    #pragma unroll
    for (int i = 0; i < FS_LV; i++) if (tm & (1 << i)) fs_st(f, f.rc_ptr[u * FS_LV + i], f.ev_nv[u * FS_LV + i]);
  }
  for (int j = lane; j < FS_LV * KQ_MAXR; j += WAVE) {
    const int i = j / KQ_MAXR, rr = j % KQ_MAXR;
    if (i >= r.plen || rr >= f.nR) continue;
    f.psum[(size_t)fs_sel4(r.lp, i) * f.nR + rr] = f.ev_sum[j];   // (unchanged sums are written back as they were)
  }
  for (int i = lane; i < r.plen; i += WAVE) {
    const int li = fs_sel4(r.lp, i);
    f.ppos[li] = f.ev_np[i];
    f.dval[li] = f.ev_v[i];
    f.nflag[li] = (uint8_t)((f.nflag[li] & ~2) | (f.ev_z[i] ? 2 : 0));
  }
  if (lane == 0) f.w->bytes += r.rowbytes;
  wsync();
}

// ---- the preemptor ----------------------------------------------------------------------------------------------------------
KQ_DEV void fs_pc_setup(Fs& f) {
  const K& k = *f.k; Wave& w = *f.w; const DSnap& S = k.S;
  if (lane_id() == 0) { int n = 0; for (int u = 0; u < w.ns; u++) if (w.s_inu[u]) f.pc_u[n++] = u; }
  wsync();
  bool miss = false;
  for (int j = 0; j < f.npc; j++) if (f.colslot[w.s_fr[f.pc_u[j]]] < 0) miss = true;
  if (miss) fs_ensure_w(f);
  for (int c = lane_id(); c < f.npc * FS_LV; c += WAVE) {
    const int j = c / FS_LV, i = c % FS_LV;
    if (i >= f.plen) continue;
    const int fr = w.s_fr[f.pc_u[j]], n = w.path[i], li = w.cs_pl[i];
    const size_t o = ix(S, n, fr);
    f.pc_lq[c] = fs_cap(local_quota(S, n, fr)); f.pc_sq[c] = fs_cap(S.sq[o]); f.pc_bl[c] = S.bl[o] == KQ_NIL_LIMIT ? KQ_NIL_LIMIT : fs_cap(S.bl[o]);
    f.pc_sqb[c] = (S.qflags[o] & KQ_QF_SUBTREE) ? fs_cap(S.sq[o]) : FS_CAP;
    f.pc_ptr[c] = fs_cell(f, li, fr);
  }
  for (int j = lane_id(); j < FS_LV * KQ_MAXR; j += WAVE) {
    const int i = j / KQ_MAXR, rr = j % KQ_MAXR;
    if (i < f.plen && rr < f.nR) f.pc_lend[j] = S.fs_lend[(size_t)(f.n0 + w.cs_pl[i]) * f.nR + rr];
  }
  for (int i = lane_id(); i < f.plen; i += WAVE) f.pc_wt[i] = S.fs_weight[f.n0 + w.cs_pl[i]];
  wsync();
}
// cq.SimulateUsageAddition(workloadUsage) / its revert (preemption.go:557, :690-695)
KQ_DEV void fs_pc_apply(const Fs& f, bool add) {
  const Wave& w = *f.w;
  int lp[FS_LV];
  #pragma unroll
  for (int i = 0; i < FS_LV; i++) lp[i] = w.cs_pl[i];
  // the staging cells serve CS_RFR chains at a time
  for (int base = 0; base < f.npc; base += CS_RFR) {
    const int nch = f.npc - base < CS_RFR ? f.npc - base : CS_RFR;
    for (int q = lane_id(); q < CS_RFR; q += WAVE) {
      FsChainOut o;
      #pragma unroll
      for (int i = 0; i < FS_LV; i++) { o.d[i] = 0; o.dp[i] = 0; }
      const int j = base + q;
      if (q < nch) o = fs_chain(f, f.pc_ptr + j * FS_LV, f.pc_lq + j * FS_LV, f.pc_sqb + j * FS_LV, f.plen, w.s_qty[f.pc_u[j]], add, true);
      #pragma unroll
      for (int i = 0; i < FS_LV; i++) { f.td[q * FS_LV + i] = o.d[i]; f.tp[q * FS_LV + i] = o.dp[i]; }
    }
    wsync();
    int rs[CS_RFR];
    #pragma unroll
    for (int q = 0; q < CS_RFR; q++) rs[q] = q < nch ? w.s_fr[f.pc_u[base + q]] % f.nR : -1;
    fs_nodes_update(f, lp, f.plen, f.pc_lend, f.pc_wt, nch, FsRowRes{{rs[0], rs[1], rs[2], rs[3], -1, -1, -1, -1}}, true, -1, nullptr);
  }
}
// Available (resource_node.go:106-122) of one in-use slot from its gathered path cells
KQ_DEV int64_t fs_avail(const int64_t* v, const int64_t* lq, const int64_t* sq, const int64_t* bl, int plen) {
  int64_t a = 0;
  #pragma unroll
  for (int i = FS_LV - 1; i >= 0; i--) {
    if (i >= plen) continue;
    if (i == plen - 1) { a = sq[i] - v[i]; continue; }
    if (bl[i] != KQ_NIL_LIMIT) a = i64min((sq[i] - lq[i]) - i64max(0, v[i] - lq[i]) + bl[i], a);
    a = i64max(0, lq[i] - v[i]) + a;
  }
  return a;
}
// workloadFits(allowBorrowing = true) preemption.go:669-686. without_own: workloadFitsForFairSharing :690-695 — the reference
// removes the preemptor's simulated usage, tests, and adds it again; on plain amounts addUsage(removeUsage(x)) == x level by
// level (what removeUsage passes up, min(val, usage - localQuota), is exactly what addUsage passes up afterwards), so the
// removal is evaluated on registers and nothing is written.
KQ_DEV bool fs_fits(const Fs& f, bool without_own) {
  Wave& w = *f.w;
  bool bad = false;
  for (int j = lane_id(); j < f.npc; j += WAVE) {
    int64_t v[FS_LV];
    #pragma unroll
    for (int i = 0; i < FS_LV; i++) { v[i] = 0; if (i < f.plen) v[i] = fs_ld(f, f.pc_ptr[j * FS_LV + i]); }
    const int64_t qty = w.s_qty[f.pc_u[j]];
    if (without_own) {
      int64_t val = qty;
      bool go = true;
      #pragma unroll
      for (int i = 0; i < FS_LV; i++) {
        if (!go || i >= f.plen) continue;
        const int64_t stored = v[i] - f.pc_lq[j * FS_LV + i];
        v[i] = v[i] - val;
        if (stored <= 0 || i + 1 >= f.plen) go = false; else val = i64min(val, stored);
      }
    }
    if (qty > i64max(0, fs_avail(v, f.pc_lq + j * FS_LV, f.pc_sq + j * FS_LV, f.pc_bl + j * FS_LV, f.plen))) bad = true;
  }
  if (lane_id() == 0) w.bytes += 40 * (int64_t)f.plen * f.npc;
  return wballot(bad) == 0;
}
KQ_DEV bool fs_fits_fs(const Fs& f) { return fs_fits(f, true); }

// ---- TargetClusterQueueOrdering ---------------------------------------------------------------------------------------------
// CandidatesOrdering of the queue heads of two ClusterQueues (common/ordering.go:42-83), as in f_head_key
KQ_DEV uint64_t fs_head_key(const Fs& f, int li) {
  const int p = fs_first(f.mq, f.posoff[li], f.posoff[li + 1]);
  const uint32_t k32 = p >= 0 ? f.k->S.fs_apply[(size_t)f.row0 + p].hkey : 0xffffffffu;
  return ((uint64_t)(k32 >> 31) << 33) | ((uint64_t)(li == f.wli ? 1 : 0) << 32) | (uint64_t)(k32 & 0x7fffffffu);
}
// nextTarget (ordering.go:144-226) from the tree-local cohort `root`; the target ClusterQueue (tree-local) or -1
KQ_DEV int fs_next_target(const Fs& f, int root) {
  Wave& w = *f.w;
  const int lane = lane_id();
  CSTAT(11, 1);
  int cohort = root;
  int64_t lb = 0;
  int result = -1;
  const uint64_t NEGK = fs_okey(-1.0);
  for (;;) {
    CSTAT(12, 1);
    const int xc = cohort - f.nqs;
    const int k0 = f.koff[xc], nkc = f.knc[xc], nkh = f.knh[xc];
    int best_cq = -1, bz = 0; uint64_t bk = NEGK;
    for (int base = 0; base < nkc; base += WAVE) {
      const int j = base + lane;
      const int c = j < nkc ? f.kid[k0 + j] : -1;
      bool elig = false; int z = 0; uint64_t key = 0;
      if (c >= 0 && !(f.nflag[c] & 1)) {
        lb += fs_cost(f, c);
        if ((f.ppos[c] <= 0 && c != f.wli) || !fs_cq_has(f, c)) f.nflag[c] |= 1;
        else { elig = true; z = (f.nflag[c] & 2) ? 1 : 0; key = fs_okey(f.dval[c]); }
      }
      uint64_t m = wballot(elig);
      if (!m) continue;
      const uint64_t mz = wballot(elig && z);
      if (mz) m = mz;
      const bool in = ((m >> lane) & 1) != 0;
      const uint64_t mx = wmax_u64(in ? key : 0);
      uint64_t tie = wballot(in && key == mx);
      if (tie & (tie - 1)) {  // equal shares: the ClusterQueue whose head candidate comes first
        const bool it = ((tie >> lane) & 1) != 0;
        const uint64_t hk = it ? fs_head_key(f, c) : ~0ull;
        const uint64_t mn = wmin_u64(hk);
        tie = wballot(it && hk == mn);
      }
      const int b = ffs64(tie);
      const int cb = wbcast_u(c, b);
      const int zb = mz ? 1 : 0;
      const int cmp = fs_cmp(zb, mx, bz, bk);
      if (cmp > 0 || (cmp == 0 && best_cq >= 0 && fs_head_key(f, cb) < fs_head_key(f, best_cq))) { best_cq = cb; bz = zb; bk = mx; }
    }
    int best_co = -1, hz = 0; uint64_t hk2 = NEGK;
    for (int base = 0; base < nkh; base += WAVE) {
      const int j = base + lane;
      const int ch = j < nkh ? f.kid[k0 + nkc + j] : -1;
      bool elig = false; int z = 0; uint64_t key = 0;
      if (ch >= 0 && !(f.nflag[ch] & 1)) {
        lb += fs_cost(f, ch);
        if (f.ppos[ch] <= 0 && f.plv[ch] < 1) f.nflag[ch] |= 1;
        else { elig = true; z = (f.nflag[ch] & 2) ? 1 : 0; key = fs_okey(f.dval[ch]); }
      }
      uint64_t m = wballot(elig);
      if (!m) continue;
      const uint64_t mz = wballot(elig && z);
      if (mz) m = mz;
      const bool in = ((m >> lane) & 1) != 0;
      const uint64_t mx = wmax_u64(in ? key : 0);
      const uint64_t tie = wballot(in && key == mx);
      const int b = 63 - clz64(tie);  // `>=` in the reference's fold: the last of equal cohorts wins
      const int hb = wbcast_u(ch, b);
      const int zb = mz ? 1 : 0;
      if (fs_cmp(zb, mx, hz, hk2) >= 0) { best_co = hb; hz = zb; hk2 = mx; }
    }
    wsync();
    if (best_co < 0 && best_cq < 0) {
      if (lane == 0) f.nflag[cohort] |= 1;
      wsync();
      result = -1;
      break;
    }
    if (fs_cmp(hz, hk2, bz, bk) >= 0) { cohort = best_co; continue; }
    result = best_cq;
    break;
  }
  const int64_t tot = wsum_i64(lb);
  if (lane == 0) w.bytes += tot;
  return result;
}
// TargetClusterQueueOrdering.Iter (ordering.go:92-127) as a "next" call
KQ_DEV int fs_ordering_next(const Fs& f) {
  if (f.plen <= 1) {
    if (!(f.nflag[f.wli] & 1) && fs_cq_has(f, f.wli)) return f.wli;
    return -1;
  }
  const int root = f.w->cs_pl[f.plen - 1];
  while (!(f.nflag[root] & 1)) {
    const int t = fs_next_target(f, root);
    if (t >= 0) return t;
  }
  return -1;
}
// PopWorkload (ordering.go:84-90) from class bitmap `from`; `to`: the class the row moves to (null: none). Returns the position.
KQ_DEV int fs_pop(const Fs& f, int li, uint64_t* from, uint64_t* to) {
  CSTAT(13, 1);
  int p = 0;
  if (lane_id() == 0) {  // one lane finds the head, removes it and looks whether anything is left behind it
    const int b = f.posoff[li + 1];
    p = fs_first(from, f.posoff[li], b);
    from[p >> 6] &= ~(1ull << (p & 63));
    if (to) to[p >> 6] |= 1ull << (p & 63);
    const bool has = fs_first(from, p + 1, b) >= 0;
    f.nflag[li] = (uint8_t)((f.nflag[li] & ~8) | (has ? 8 : 0));
  }
  p = wuniform_i32(p);
  wsync_lds();
  return p;
}
KQ_DEV bool fs_push_target(Fs& f, int* nt, int row, int p, int reason) {
  if (*nt >= f.k->X.tgt_cap) { set_error(*f.k, KQ_ECAPACITY); return false; }
  if (lane_id() == 0) fs_tset(f, *nt, FsTgt{row, p, (uint8_t)reason});  // read back after later fences only
  (*nt)++;
  return true;
}

// ---- set-up -----------------------------------------------------------------------------------------------------------------
KQ_DEV bool fs_setup(Search& s, Fs& f) {
  const K& k = *s.k; Wave& w = *s.w; const DSnap& S = k.S;
  const int lane = lane_id();
  if (!k.C.fair_sharing || !k.C.fs_on || !S.fs_ok || !fs_plain_now(k) || w.plen > FS_LV || S.nR > KQ_MAXR) return false;
  const int tree = S.tree_of[w.cq];
  if (!S.fs_ok[tree]) return false;
  f.k = &k; f.w = &w; f.s = &s;
  f.n0 = S.tree_node_off[tree]; f.nn = S.tree_node_off[tree + 1] - f.n0;
  f.q0 = S.tree_cq_off[tree]; f.nqs = S.tree_cq_off[tree + 1] - f.q0;
  f.ncoh = f.nn - f.nqs; f.nR = S.nR; f.nfr = S.nfr;
  f.row0 = S.tree_row_off[tree]; f.nrows = S.tree_row_off[tree + 1] - f.row0;
  f.mw = (f.nrows + 63) / 64 + 1;
  f.plen = w.plen; f.wli = S.cq_local[w.cq]; f.tree = tree;
  f.wcopied = false;
  int npc = 0;
  for (int u = 0; u < w.ns; u++) npc += w.s_inu[u] ? 1 : 0;
  if (npc * FS_LV > FS_PCN) return false;
  // the chains run on plain int64 (FS_CAP): the preemptor's own quantities must be plain too (the snapshot's are: C.fs_plain)
  for (int u = 0; u < w.ns; u++) if (w.s_qty[u] < 0 || w.s_qty[u] >= ((int64_t)1 << 50)) return false;
  f.npc = npc;
  // columns: every resource of the flavors the preemptor needs, then of the flavors it uses
  int colfr[FS_NCMAX];
  int nc = 0;
  for (int pass = 0; pass < 2; pass++)
    for (int u = 0; u < w.ns; u++) {
      if (pass == 0 ? !w.s_need[u] : (w.s_need[u] || !w.s_inu[u])) continue;
      const int fl = w.s_fr[u] / S.nR;
      for (int r = 0; r < S.nR; r++) {
        const int fr = fl * S.nR + r;
        bool have = false;
        #pragma unroll
        for (int q = 0; q < FS_NCMAX; q++) if (q < nc && colfr[q] == fr) have = true;
        if (have || nc >= FS_NCMAX) continue;
        #pragma unroll
        for (int q = 0; q < FS_NCMAX; q++) if (q == nc) colfr[q] = fr;
        nc++;
      }
    }
  f.nc = nc;
  const size_t need = fs_bytes(f.nn, f.nqs, f.nR, f.nfr, f.mw, nc);
  unsigned char* spill = k.X.cs ? k.X.cs + (size_t)s.slot * (size_t)k.X.cs_bytes : nullptr;
  const bool all_lds = w.cs_lds && (size_t)w.cs_lds_bytes >= need;
#ifdef KQ_HOST_EMU
  if (!all_lds && (!spill || (size_t)k.X.cs_bytes < need)) return false;  // the emulation also places (part of) the state in the spill space
#else
  if (!all_lds) return false;  // on the device the formulation only pays with the whole state in LDS
#endif
  CsCarve cv{w.cs_lds, w.cs_lds ? w.cs_lds + w.cs_lds_bytes : nullptr, spill};
  f.rc_ptr = (int32_t*)cv.take(FS_RC * 8); f.rc_lq = (int64_t*)cv.take(FS_RC * 8); f.rc_sqb = (int64_t*)cv.take(FS_RC * 8);
  f.rc_lend = (int64_t*)cv.take(FS_LV * KQ_MAXR * 8); f.rc_wt = (double*)cv.take(FS_LV * 8);
  f.td = (int64_t*)cv.take(FS_RC * 8); f.tp = (int32_t*)cv.take(FS_RC * 4); f.tx = (double*)cv.take(FS_LV * KQ_MAXR * 8); f.tout = (uint64_t*)cv.take(4 * 8);
  f.ev_nv = (int64_t*)cv.take(FS_RC * 8); f.ev_sum = (int64_t*)cv.take(FS_LV * KQ_MAXR * 8); f.ev_t = (int32_t*)cv.take(FS_RFR * 4); f.ev_np = (int32_t*)cv.take(FS_LV * 4);
  f.ev_v = (double*)cv.take(FS_LV * 8); f.ev_z = (uint8_t*)cv.take(FS_LV);
  f.pc_ptr = (int32_t*)cv.take(FS_PCN * 8); f.pc_lq = (int64_t*)cv.take(FS_PCN * 8); f.pc_sq = (int64_t*)cv.take(FS_PCN * 8);
  f.pc_sqb = (int64_t*)cv.take(FS_PCN * 8); f.pc_bl = (int64_t*)cv.take(FS_PCN * 8); f.pc_u = (int32_t*)cv.take(FS_PCN * 4);
  f.pc_lend = (int64_t*)cv.take(FS_LV * KQ_MAXR * 8); f.pc_wt = (double*)cv.take(FS_LV * 8);
  (void)cv.take(FS_NCMAX * 8);
  f.nflag = (uint8_t*)cv.take(f.nn); f.dval = (double*)cv.take((size_t)f.nn * 8); f.ppos = (int32_t*)cv.take((size_t)f.nn * 4);
  f.c0 = (int16_t*)cv.take((size_t)f.nn * 2); f.c1 = (int16_t*)cv.take((size_t)f.nn * 2); f.kid = (int16_t*)cv.take((size_t)f.nn * 2); f.par = (int16_t*)cv.take((size_t)f.nn * 2); f.plv = (int8_t*)cv.take(f.nn);
  const size_t nco = (size_t)(f.ncoh > 0 ? f.ncoh : 1);
  f.koff = (int16_t*)cv.take(nco * 2); f.knc = (int16_t*)cv.take(nco * 2); f.knh = (int16_t*)cv.take(nco * 2);
  f.posoff = (int32_t*)cv.take((size_t)(f.nqs + 1) * 4); f.colslot = (int8_t*)cv.take(f.nfr);
  f.psum = (int64_t*)cv.take((size_t)f.nn * f.nR * 8);
  f.m1 = (uint64_t*)cv.take((size_t)f.mw * 8); f.m2 = (uint64_t*)cv.take((size_t)f.mw * 8);
  f.mq = f.m1;
  f.tpos = s.cand;
  f.W = s.W; f.usage = s.usage; f.removed = s.removed; f.trow = s.trow; f.treason = s.treason;
  f.col = (int64_t*)cv.take((size_t)f.nn * 8 * (size_t)(nc > 0 ? nc : 1));
  f.bq_c = (int16_t*)cv.take(FS_BQ * 2); f.bq_ord = (int16_t*)cv.take(FS_BQ * 2); f.bq_k = (uint64_t*)cv.take(FS_BQ * 8); f.bq_h = (uint64_t*)cv.take(FS_BQ * 8);
  f.bq_z = (uint8_t*)cv.take(FS_BQ);
  f.br_c = (int16_t*)cv.take(FS_BQ * 2); f.br_ap = (int16_t*)cv.take(FS_BQ * 2); f.br_at = (int16_t*)cv.take(FS_BQ * 2); f.br_n = (int16_t*)cv.take(FS_BQ * 2);
  f.br_off = (int16_t*)cv.take((FS_BQ + 1) * 2); f.br_fl = (uint8_t*)cv.take(FS_BQ);
  f.cl_pos = (int32_t*)cv.take(FS_BC * 4); f.cl_rk = (uint8_t*)cv.take(FS_BC);
  f.cq_lca = (int32_t*)cv.take((size_t)f.nqs * 4);
  f.tcap = 0; f.lt_row = nullptr; f.lt_pos = nullptr; f.lt_rs = nullptr;
  if (all_lds) {
    const size_t left = (size_t)(cv.ae - cv.a);
    int cap = (int)(left / 9) & ~15;
    if (cap > 4096) cap = 4096;
#ifdef KQ_HOST_EMU
    if ((w.h & 1) && cap > 64) cap = 64;   // (tests: lists that straddle the LDS part and the global part)
#endif
    if (cap >= 64) { f.tcap = cap; f.lt_row = (int32_t*)cv.take((size_t)cap * 4); f.lt_pos = (int32_t*)cv.take((size_t)cap * 4); f.lt_rs = (uint8_t*)cv.take(cap); }
  }
  fs_assume_lds(f);
  for (int i = lane; i < f.nfr; i += WAVE) f.colslot[i] = -1;
  wsync();
  #pragma unroll
  for (int q = 0; q < FS_NCMAX; q++) {
    if (q >= nc) continue;
    if (lane == 0) f.colslot[colfr[q]] = (int8_t)q;
  }
  // tree-local ids of the preemptor's path
  if (lane == 0) for (int i = 0; i < w.plen; i++) w.cs_pl[i] = S.node_local[w.path[i]];
  wsync();
  return true;
}
KQ_DEV void fs_lcas(const Fs& f, int cand, int* ap, int* at);
// state of the search = the plane it starts from
KQ_DEV void fs_init(Fs& f) {
  const K& k = *f.k; const DSnap& S = k.S;
  const int lane = lane_id();
  for (int i = lane; i < f.nn; i += WAVE) {
    f.nflag[i] = 0; f.plv[i] = -1;
    f.c0[i] = S.fs_c0[f.n0 + i]; f.c1[i] = S.fs_c1[f.n0 + i]; f.kid[i] = S.fs_kid[f.n0 + i]; f.par[i] = S.fs_par[f.n0 + i];
    if (i >= f.nqs) { f.koff[i - f.nqs] = S.fs_koff[f.n0 + i]; f.knc[i - f.nqs] = S.fs_knc[f.n0 + i]; f.knh[i - f.nqs] = S.fs_knh[f.n0 + i]; }
  }
  wsync();
  for (int i = lane; i < f.plen; i += WAVE) f.plv[f.w->cs_pl[i]] = (int8_t)i;
  for (int i = lane; i <= f.nqs; i += WAVE) f.posoff[i] = S.fs_posoff[(size_t)f.q0 + f.tree + i];
  for (int i = lane; i < f.mw; i += WAVE) { f.m1[i] = 0; f.m2[i] = 0; }
  for (int q = 0; q < f.nc; q++) {
    int64_t* cp = f.col + (size_t)q * f.nn;
    int fr = 0;
    for (int x = 0; x < f.nfr; x++) if (f.colslot[x] == q) fr = x;
    for (int i = lane; i < f.nn; i += WAVE) cp[i] = f.usage[ix(S, S.tree_nodes[f.n0 + i], fr)];
  }
  if (f.usage == k.usage) {  // cycle-start plane: k_fs_sums already reduced it
    for (int i = lane; i < f.nn * f.nR; i += WAVE) f.psum[i] = k.X.bu_sum[(size_t)S.tree_nodes[f.n0 + i / f.nR] * f.nR + i % f.nR];
    for (int i = lane; i < f.nn; i += WAVE) f.ppos[i] = k.X.bu_pos[S.tree_nodes[f.n0 + i]];
  } else {
    for (int i = lane; i < f.nn * f.nR; i += WAVE) f.psum[i] = 0;
    for (int i = lane; i < f.nn; i += WAVE) f.ppos[i] = 0;
    wsync();
    for (int i = lane; i < f.nn * f.nfr; i += WAVE) {
      const int li = i / f.nfr, fr = i % f.nfr;
      const int64_t b = a_sub(f.usage[ix(S, S.tree_nodes[f.n0 + li], fr)], S.fs_q[(size_t)(f.n0 + li) * f.nfr + fr].sqb);
      if (b > 0) { atomic_add_i64((long long*)&f.psum[(size_t)li * f.nR + fr % f.nR], (long long)b); atomic_add_i32(&f.ppos[li], 1); }
    }
  }
  wsync();
  for (int i = lane; i < f.nn; i += WAVE) fs_refresh(f, i, S.fs_lend + (size_t)(f.n0 + i) * f.nR, S.fs_weight[f.n0 + i]);
  // per ClusterQueue: its almost-LCAs with the preemptor (least_common_ancestor.go:27-58) — a visit read them through three dependent LDS
  // round trips (602 -> 378 ms of the 6.3 s the first strategy takes at cfg 4f, profiles/r04k_* vs r05d_prof_fair_cfg4f_process_only.txt)
  for (int q = lane; q < f.nqs; q += WAVE) {
    int ap = f.wli, at = q;
    fs_lcas(f, q, &ap, &at);
    f.cq_lca[q] = (int32_t)(((uint32_t)(uint16_t)(int16_t)at << 16) | (uint32_t)(uint16_t)(int16_t)ap);
  }
  wsync();
}
KQ_DEV void fs_lca_get(const Fs& f, int q, int* ap, int* at) { const uint32_t v = (uint32_t)f.cq_lca[q]; *ap = (int16_t)(v & 0xffff); *at = (int16_t)(v >> 16); }

// getAlmostLCAs (least_common_ancestor.go:27-58): the nodes just below the lowest common ancestor on the preemptor's and on the
// target ClusterQueue's path (tree-local ids), from the tree's parent table
KQ_DEV void fs_lcas(const Fs& f, int cand, int* ap, int* at) {
  const Wave& w = *f.w;
  int prev = cand, cur = f.par[cand];
  for (int j = 1; j < FS_LV && cur >= 0; j++) {
    const int l = f.plv[cur];
    if (l >= 1) { *ap = w.cs_pl[l - 1]; *at = prev; return; }
    prev = cur; cur = f.par[cur];
  }
  *ap = w.cs_pl[f.plen - 1]; *at = prev;
}


// ---- a cohort's ClusterQueues as a batch ------------------------------------------------------------------------------------------
// Between two RemoveWorkload commits nothing the tournament reads changes: a candidate that fails the strategy goes to retryCandidates
// and the state stays as it was (ComputeTargetShareAfterRemoval restores it, target.go:67-73). So after nextTarget has returned the
// ClusterQueue `cand` of cohort X, the CALLS THAT FOLLOW are known in advance as long as candidates keep failing: X's non-pruned
// ClusterQueues come back one after the other in descending share order (ordering.go:152-176: max DRS, zero-weight borrowers first,
// ties by the queue heads' CandidatesOrdering), each has all its candidates popped, fails, and is pruned by the next call. Instead of
// ~3000 dependent instructions per pop and ~1200 per call, the candidates of all those ClusterQueues are evaluated at once — one lane
// each: the row's record, the quota constants of its cells, the removeUsage chain up to the target's almost-LCA, the share there — and
// the first one that passes (in visiting order) is found with a ballot. Everything in front of it is moved to retryCandidates in bulk,
// the ClusterQueues in front of its own are pruned, and the algorithmic bytes of the calls / visits / evaluations that did not
// happen one by one are charged in closed form:
//   call t (t = 2 ..) returns the ClusterQueue of rank t - 1; it reads every non-pruned child of the cohorts from the root down to X:
//   above X those sets are what call 1 left (U_up), at X it is A' minus the ranks 0 .. t - 3 (rank t - 2 is exhausted but only this
//   call notices and prunes it).
// The batch ends in front of the preemptor's own ClusterQueue (an unconditional removal, preemption.go:399-410), in front of a
// ClusterQueue whose share does not beat X's best child cohort (nextTarget descends there instead, ordering.go:219), and at FS_BC
// candidates. Returns -1: not applicable, nothing was touched (the caller takes `cand` alone, the walk's way); 0: every candidate of
// the batch failed; 1: *pos_out passed (it has been popped; the caller commits it).
KQ_DEV uint64_t fs_range_bits(const uint64_t* m, int wi, int a, int b) {  // the bits of word wi of bitmap m inside [a, b)
  if (a >= b) return 0;
  const int wa = a >> 6, wb = (b - 1) >> 6;
  if (wi < wa || wi > wb) return 0;
  uint64_t x = m[wi];
  if (wi == wa) x &= ~0ull << (a & 63);
  if (wi == wb) { const int e = (b - 1) & 63; if (e < 63) x &= (2ull << e) - 1; }
  return x;
}
KQ_DEV int fs_range_count(const uint64_t* m, int a, int b) {
  int n = 0;
  if (a >= b) return 0;
  for (int wi = a >> 6; wi <= ((b - 1) >> 6); wi++) n += popc64(fs_range_bits(m, wi, a, b));
  return n;
}
// the share of node `at` (on the row's path) after RemoveWorkload of the row at position p, nothing written: the one-lane form of
// fs_row_apply(commit = false) + fs_nodes_update for that node. *flags = zero-weight-borrows | (borrowed cells > 0) << 1.
KQ_DEV uint64_t fs_eval_removed(const Fs& f, int p, int at, int* flags) {
  const DSnap& S = f.k->S;
  const FsApply a = S.fs_apply[(size_t)f.row0 + p];
  const int plen = a.plen;
  int ia = 0;
  #pragma unroll
  for (int i = 0; i < FS_LV; i++) if (i < plen && (int)a.lp[i] == at) ia = i;
  const FsEnt en = fs_entries(S, a);
  int64_t du[FS_RFR]; int dpu[FS_RFR];
  #pragma unroll
  for (int u = 0; u < FS_RFR; u++) {
    du[u] = 0; dpu[u] = 0;
    const int fr = en.fr[u];
    if (fr < 0) continue;
    int64_t val = en.qty[u];
    bool go = true;
    #pragma unroll
    for (int i = 0; i < FS_LV; i++) {
      if (!go || i >= plen || i > ia) continue;
      const int li = a.lp[i];
      const FsQ q = S.fs_q[(size_t)(f.n0 + li) * f.nfr + fr];
      const int64_t lq = fs_cap(q.lq), sqb = fs_cap(q.sqb);
      const int64_t uu = fs_ld(f, fs_cell(f, li, fr));
      const int64_t stored = uu - lq, nv = uu - val;
      if (i == ia) {
        const int64_t ob = i64max(0, uu - sqb), nb = i64max(0, nv - sqb);
        du[u] = nb - ob; dpu[u] = (nb > 0 ? 1 : 0) - (ob > 0 ? 1 : 0);
      }
      if (stored <= 0 || i + 1 >= plen) go = false; else val = i64min(val, stored);
    }
  }
  double ratio = 0;
  for (int rr = 0; rr < f.nR; rr++) {
    int64_t d = 0;
    #pragma unroll
    for (int u = 0; u < FS_RFR; u++) if (en.res[u] == rr) d += du[u];
    const int64_t sum = f.psum[(size_t)at * f.nR + rr] + d;
    double x = 0;
    if (sum > 0) { const int64_t lr = S.fs_lend[(size_t)(f.n0 + at) * f.nR + rr]; if (lr > 0) x = (double)sum * 1000.0 / (double)lr; }
    if (x > ratio) ratio = x;
  }
  int np = f.ppos[at];
  #pragma unroll
  for (int u = 0; u < FS_RFR; u++) np += dpu[u];
  const double weight = S.fs_weight[f.n0 + at];
  const bool zwb = weight == 0 && ratio != 0;
  double v = ratio;
  if (!zwb) v = ratio == 0 ? 0.0 : (weight == 0 ? __builtin_inf() : ratio / weight);
  *flags = (zwb ? 1 : 0) | (np > 0 ? 2 : 0);
  return fs_okey(v);
}
// sum of fs_cost over the non-pruned children (ClusterQueues and cohorts) of tree-local cohort X: what one nextTarget level reads
KQ_DEV int64_t fs_level_cost(const Fs& f, int X) {
  const int xc = X - f.nqs;
  const int k0 = f.koff[xc], nk = f.knc[xc] + f.knh[xc];
  int64_t c = 0;
  for (int j = lane_id(); j < nk; j += WAVE) { const int ch = f.kid[k0 + j]; if (!(f.nflag[ch] & 1)) c += fs_cost(f, ch); }
  return wsum_i64(c);
}
// `second`: runSecondFsStrategy's form (preemption.go:501-534): no candidate simulation, one pop per ClusterQueue, DropQueue at once.
// (force-inlined: passing the Fs block by reference to a function that is not inlined sends the whole block to scratch memory in the
// caller as well — every f.psum / f.dval / ... access of the search became a scratch load, 14.1 instead of 11.5 s per cfg 4f cycle in
// k_process_fair with the batch switched OFF, profiles/r04c_bench_cfg4f_nobatch.json)
KQ_DEV int fs_batch(Fs& f, int cand, int strategy0, bool second, int* pos_out) {
  const K& k = *f.k; Wave& w = *f.w; const DSnap& S = k.S;
  const int lane = lane_id();
  fs_assume_lds(f);
  const int X = f.par[cand];
  if (X < f.nqs) return -1;
  const int xc = X - f.nqs;
  const int k0 = f.koff[xc], nkc = f.knc[xc], nkh = f.knh[xc];
  // ---- X's non-pruned ClusterQueues (all eligible: the call that returned `cand` has just pruned the others), child order ----
  int nq = 0;
  for (int base = 0; base < nkc; base += WAVE) {
    const int j = base + lane;
    const int c = j < nkc ? f.kid[k0 + j] : -1;
    const bool in = c >= 0 && !(f.nflag[c] & 1);
    const uint64_t m = wballot(in);
    const int o = nq + popc64(m & ((1ull << lane) - 1));
    if (in && o < FS_BQ) { f.bq_c[o] = (int16_t)c; f.bq_k[o] = fs_okey(f.dval[c]); f.bq_z[o] = (f.nflag[c] & 2) ? 1 : 0; }
    nq += popc64(m);
  }
  if (nq > FS_BQ || nq < 1) return -1;
  // X's best child cohort: a ClusterQueue is only returned while its share is strictly above it (ordering.go:219 `>= 0` descends)
  int hz = 0; uint64_t hk2 = fs_okey(-1.0); bool has_co = false;
  for (int base = 0; base < nkh; base += WAVE) {
    const int j = base + lane;
    const int ch = j < nkh ? f.kid[k0 + nkc + j] : -1;
    const bool in = ch >= 0 && !(f.nflag[ch] & 1);
    uint64_t m = wballot(in);
    if (!m) continue;
    const int z = in && (f.nflag[ch] & 2) ? 1 : 0;
    const uint64_t mz = wballot(in && z);
    if (mz) m = mz;
    const bool cmpin = ((m >> lane) & 1) != 0;
    const uint64_t mx = wmax_u64(cmpin ? fs_okey(f.dval[ch]) : 0);
    const int zb = mz ? 1 : 0;
    if (!has_co || fs_cmp(zb, mx, hz, hk2) >= 0) { hz = zb; hk2 = mx; }
    has_co = true;
  }
  wsync_lds();
  // ---- visiting order: zero-weight borrowers first, share descending, ties by the queue heads (ordering.go:158-166) ----
  bool tie = false;
  for (int i = lane; i < nq; i += WAVE)
    for (int j = 0; j < nq; j++) if (j != i && f.bq_z[j] == f.bq_z[i] && f.bq_k[j] == f.bq_k[i]) tie = true;
  if (wballot(tie)) {
    for (int i = lane; i < nq; i += WAVE) f.bq_h[i] = fs_head_key(f, f.bq_c[i]);
  } else {
    for (int i = lane; i < nq; i += WAVE) f.bq_h[i] = 0;
  }
  wsync_lds();
  for (int i = lane; i < nq; i += WAVE) {
    const int zi = f.bq_z[i]; const uint64_t ki = f.bq_k[i], hi = f.bq_h[i];
    int r = 0;
    for (int j = 0; j < nq; j++) {
      if (j == i) continue;
      const int zj = f.bq_z[j]; const uint64_t kj = f.bq_k[j], hj = f.bq_h[j];
      const bool before = zj != zi ? zj > zi : (kj != ki ? kj > ki : (hj != hi ? hj < hi : j < i));
      if (before) r++;
    }
    f.bq_ord[r] = (int16_t)i;
  }
  wsync_lds();
  if ((int)f.bq_c[f.bq_ord[0]] != cand) return -1;  // (cannot happen: rank 0 is what nextTarget has just returned)
  // ---- per rank: the almost-LCAs and both shares (least_common_ancestor.go:27-58), fsStrategyUnsatisfiable, the candidates ----
  const uint64_t* mq = second ? f.m2 : f.m1;
  for (int r = lane; r < nq; r += WAVE) {
    const int i = f.bq_ord[r], q = f.bq_c[i];
    int ap = f.wli, at = q;
    fs_lca_get(f, q, &ap, &at);
    const int pz = (f.nflag[ap] & 2) ? 1 : 0, tz = (f.nflag[at] & 2) ? 1 : 0;
    const uint64_t pk = fs_okey(f.dval[ap]), tk = fs_okey(f.dval[at]);
    const bool unsat = !second && fs_pos_inf(f, ap) && !fs_pos_inf(f, at);
    // the batch stops in front of: the preemptor's own ClusterQueue (first strategy), a ClusterQueue that does not beat the best cohort
    const bool stop = (!second && q == f.wli) || (has_co && fs_cmp(hz, hk2, f.bq_z[i], f.bq_k[i]) >= 0);
    const bool passed2 = fs_cmp(pz, pk, tz, tk) < 0;  // second strategy: LessThanInitialShare on the shares as they are
    f.br_c[r] = (int16_t)q; f.br_ap[r] = (int16_t)ap; f.br_at[r] = (int16_t)at;
    f.br_fl[r] = (uint8_t)((pz ? 1 : 0) | (tz ? 2 : 0) | (unsat ? 4 : 0) | (stop ? 8 : 0) | (passed2 ? 16 : 0));
    f.br_n[r] = (int16_t)fs_range_count(mq, f.posoff[q], f.posoff[q + 1]);
  }
  wsync_lds();
  // how many ranks the batch takes (uniform, serial over <= 64 entries), and the offsets of their candidates
  int kk = 0, T = 0;
  for (int r = 0; r < nq; r++) {
    if (f.br_fl[r] & 8) break;
    const int n = second ? 0 : ((f.br_fl[r] & 4) ? 0 : (int)f.br_n[r]);
    if (T + n > FS_BC) break;
    if (lane == 0) f.br_off[r] = (int16_t)T;
    T += n; kk++;
  }
  if (kk < 1) return -1;
  if (lane == 0) f.br_off[kk] = (int16_t)T;
  wsync_lds();
  int pass_rank = -1, pass_idx = -1, pass_pos = -1;
  int64_t eval_bytes = 0;
  if (!second) {
    // ---- candidate list in visiting order, then the evaluation, 64 (one in the emulation) at a time ----
    bool miss = false;
    for (int r = lane; r < kk; r += WAVE) {
      if (f.br_fl[r] & 4) continue;
      const int q = f.br_c[r], a0 = f.posoff[q], b0 = f.posoff[q + 1];
      int o = f.br_off[r];
      if (a0 < b0)
        for (int wi = a0 >> 6; wi <= ((b0 - 1) >> 6); wi++) {
          uint64_t x = fs_range_bits(f.m1, wi, a0, b0);
          while (x) { const int b = ffs64(x); x &= x - 1; f.cl_pos[o] = wi * 64 + b; f.cl_rk[o] = (uint8_t)r; o++; }
        }
    }
    wsync_lds();
    for (int i = lane; i < T; i += WAVE) {
      const FsEnt en = fs_entries(S, S.fs_apply[(size_t)f.row0 + f.cl_pos[i]]);
      #pragma unroll
      for (int e = 0; e < FS_RFR; e++) if (en.fr[e] >= 0 && f.colslot[en.fr[e]] < 0) miss = true;
    }
    if (wballot(miss)) fs_ensure_w(f);
    for (int base = 0; base < T && pass_idx < 0; base += WAVE) {
      const int i = base + lane;
      bool pass = false;
      int cost = 0;   // what the walk charges this evaluation (the cached share of `at`, fs_cost's two parts)
      if (i < T) {
        const int r = f.cl_rk[i], at = f.br_at[r];
        int fl = 0;
        const uint64_t nk = fs_eval_removed(f, f.cl_pos[i], at, &fl);
        cost = (int)f.c0[at] + ((fl & 2) ? (int)f.c1[at] : 0);
        const int pz = f.br_fl[r] & 1, tz = (f.br_fl[r] >> 1) & 1;
        const uint64_t pk = fs_okey(f.dval[f.br_ap[r]]);
        pass = strategy0 == KQ_FS_LESS_THAN_OR_EQUAL_TO_FINAL_SHARE ? fs_cmp(pz, pk, fl & 1, nk) <= 0 : fs_cmp(pz, pk, tz, fs_okey(f.dval[at])) < 0;
      }
      const uint64_t pm = wballot(pass);
      int upto = T - base < WAVE ? T - base : WAVE;   // evaluations of this chunk that the walk would have made
      if (pm) { const int b = ffs64(pm); pass_idx = base + b; upto = b + 1; }
      eval_bytes += wsum_i64(lane < upto ? (int64_t)cost : 0);
    }
    if (pass_idx >= 0) { pass_rank = f.cl_rk[pass_idx]; pass_pos = f.cl_pos[pass_idx]; }
    CSTAT(13, pass_idx >= 0 ? pass_idx + 1 : T);
  } else {
    // second strategy: the first ClusterQueue in visiting order whose target share is above the preemptor's
    for (int base = 0; base < kk && pass_rank < 0; base += WAVE) {
      const int r = base + lane;
      const uint64_t pm = wballot(r < kk && (f.br_fl[r] & 16));
      if (pm) pass_rank = base + ffs64(pm);
    }
    if (pass_rank >= 0) pass_pos = fs_first(f.m2, f.posoff[f.br_c[pass_rank]], f.posoff[f.br_c[pass_rank] + 1]);
    CSTAT(13, pass_rank >= 0 ? pass_rank + 1 : kk);
  }
  // ---- what the calls / visits that did not run one by one are charged ----
  const int visited = pass_rank >= 0 ? pass_rank + 1 : kk;       // ClusterQueues nextTarget returned (the first by the real call)
  const int R = visited - 1;                                     // virtual calls
  int64_t bytes = eval_bytes;
  {
    int64_t vb = 0;
    for (int r = lane; r < visited; r += WAVE) vb += fs_cost(f, f.br_ap[r]) + fs_cost(f, f.br_at[r]);
    bytes += wsum_i64(vb);
  }
  if (R > 0) {
    int64_t up = 0;
    for (int A = f.par[X]; A >= f.nqs; A = f.par[A]) up += fs_level_cost(f, A);
    const int64_t Ap = fs_level_cost(f, X);
    // first strategy: rank s is pruned by the call after the one that finds it exhausted; second: DropQueue prunes it at its own visit
    int64_t sub = 0;
    for (int s = lane; s < visited; s += WAVE) {
      const int times = second ? (R - s) : (R - 1 - s);
      if (times > 0) sub += fs_cost(f, f.br_c[s]) * times;
    }
    bytes += (int64_t)R * (up + Ap) - wsum_i64(sub);
    CSTAT(11, R);
  }
  if (lane == 0) w.bytes += bytes;
  CSTAT(27, 1); CSTAT(28, R); if (second) CSTAT(29, 1);
  // ---- effects ----
  if (!second) {
    // candidates in front of the passing one (all of them if none passed) become retryCandidates
    const int nmove = pass_idx >= 0 ? pass_idx : T;
    for (int i = lane; i < nmove; i += WAVE) {
      const int p = f.cl_pos[i];
      atomic_and_u64(&f.m1[p >> 6], ~(1ull << (p & 63)));
      atomic_or_u64(&f.m2[p >> 6], 1ull << (p & 63));
    }
    wsync_lds();
    // fsStrategyUnsatisfiable :494-497: the whole queue goes over without a simulation
    for (int r = 0; r < visited; r++) {
      if (!(f.br_fl[r] & 4)) continue;
      const int q = f.br_c[r], a0 = f.posoff[q], b0 = f.posoff[q + 1];
      if (a0 < b0)
        for (int wi = (a0 >> 6) + lane; wi <= ((b0 - 1) >> 6); wi += WAVE) {
          const uint64_t mv = fs_range_bits(f.m1, wi, a0, b0);
          CSTAT(13, popc64(mv));
          atomic_or_u64(&f.m2[wi], mv); atomic_and_u64(&f.m1[wi], ~mv);
        }
      wsync_lds();
    }
    if (pass_idx >= 0 && lane == 0) f.m1[pass_pos >> 6] &= ~(1ull << (pass_pos & 63));  // PopWorkload of the one that passed
    wsync_lds();
    for (int r = lane; r < visited; r += WAVE) {
      const int q = f.br_c[r];
      if (r == pass_rank) {
        const bool has = fs_first(f.m1, f.posoff[q], f.posoff[q + 1]) >= 0;
        f.nflag[q] = (uint8_t)((f.nflag[q] & ~8) | (has ? 8 : 0));
      } else {
        // exhausted; the call that returned the next rank has pruned it — the last one of a batch without a pass is still unnoticed
        const bool pruned = pass_rank >= 0 || r < visited - 1;
        f.nflag[q] = (uint8_t)((f.nflag[q] & ~8) | (pruned ? 1 : 0));
      }
    }
  } else {
    // one pop per visited ClusterQueue (ordering.PopWorkload :84-90), then DropQueue :129-131
    for (int r = lane; r < visited; r += WAVE) {
      const int q = f.br_c[r];
      const int p = fs_first(f.m2, f.posoff[q], f.posoff[q + 1]);
      if (p >= 0) atomic_and_u64(&f.m2[p >> 6], ~(1ull << (p & 63)));
      f.nflag[q] |= 1;
    }
  }
  wsync_lds();
  if (pass_rank < 0) return 0;
  *pos_out = pass_pos;
  return 1;
}


// fillBackWorkloads (preemption.go:341-354, allowBorrowing = true) without walking it: the reference adds the targets back newest-first,
// keeps one out of the set when the preemptor still fits, removes it again otherwise. A probe that fails leaves the state as it was (on
// plain amounts removeUsage undoes addUsage level by level), and usage only grows with every probe that is kept back — so the probes of
// 64 targets are EVALUATED at once against the state as it is (one lane each: the addUsage chain of the row, Available on the
// preemptor's path with the cells the chain reaches replaced), the lanes in front of the first one that fits have failed for good, that
// one is committed (swap-delete, as the reference), and only the lanes behind it are evaluated again. Returns the new target count.
KQ_DEV bool fs_probe_fits(const Fs& f, int p) {  // would the preemptor still fit with the row at position p added back? (one lane)
  const DSnap& S = f.k->S; const Wave& w = *f.w;
  const FsApply a = S.fs_apply[(size_t)f.row0 + p];
  const FsEnt en = fs_entries(S, a);
  const int rplen = a.plen;
  bool bad = false;
  for (int j = 0; j < f.npc; j++) {
    const int frj = w.s_fr[f.pc_u[j]];
    int64_t add[FS_LV];
    #pragma unroll
    for (int i = 0; i < FS_LV; i++) add[i] = 0;
    #pragma unroll
    for (int e = 0; e < FS_RFR; e++) {
      if (en.fr[e] != frj) continue;
      int64_t val = en.qty[e];
      bool go = true;
      #pragma unroll
      for (int h = 0; h < FS_LV; h++) {  // addUsage resource_node.go:144-152
        if (!go || h >= rplen) continue;
        const int li = a.lp[h];
        const int64_t uu = fs_ld(f, fs_cell(f, li, frj));
        const int64_t lq = fs_cap(S.fs_q[(size_t)(f.n0 + li) * f.nfr + frj].lq);
        const int pl = f.plv[li];
        #pragma unroll
        for (int i = 0; i < FS_LV; i++) if (i == pl) add[i] = val;
        const int64_t la = i64max(0, lq - uu);
        if (h + 1 < rplen && val > la) val = val - la; else go = false;
      }
    }
    int64_t v[FS_LV];
    #pragma unroll
    for (int i = 0; i < FS_LV; i++) { v[i] = 0; if (i < f.plen) v[i] = fs_ld(f, f.pc_ptr[j * FS_LV + i]) + add[i]; }
    if (w.s_qty[f.pc_u[j]] > i64max(0, fs_avail(v, f.pc_lq + j * FS_LV, f.pc_sq + j * FS_LV, f.pc_bl + j * FS_LV, f.plen))) bad = true;
  }
  return !bad;
}
KQ_DEV int fs_fillback_batch(Fs& f, int nt, int64_t* tbytes) {
  Wave& w = *f.w; const DSnap& S = f.k->S;
  const int lane = lane_id();
  fs_assume_lds(f);
  const int64_t fits_bytes = 40 * (int64_t)f.plen * f.npc;
  int64_t bytes = 0;
  int t = nt - 2;
  // Where the fill-back keeps most targets back (a recomputation inside processEntry removes ~750 rows and keeps ~24 of them,
  // profiles/r04f_cfg4f_jacobi_probe.txt) the probes succeed one after the other, and a probe the walk's way is two dependent global
  // round trips (the row's record, then the quota constants of its cells) plus the bookkeeping of shares nobody reads any more:
  // 5.2 us per probe, 2.6 of the 10.5 s of k_process_fair (profiles/r04g_prof_fair_cfg4f_process_only.txt). So the operands of FS_FG
  // probes are fetched together — a lane per (row, usage entry) — and parked in LDS; the probes then run one after the other on LDS
  // alone: AddWorkload as the usage chains only (the borrowed sums and cached shares are dead once the strategies are over), the fit
  // test, and RemoveWorkload again where it fails. Same cells written in the same order, same bytes charged.
  constexpr int FS_FG = 16, FS_FC = FS_FG * FS_RFR;
  const size_t st_bytes = (size_t)FS_FC * FS_LV * 8 + (size_t)FS_FC * FS_LV * 4 + (size_t)FS_FC * 8 + (size_t)FS_FC * 4 + (size_t)FS_FG * 8;
  const bool staged = (size_t)f.nn * f.nR * 8 >= st_bytes;   // the staging area lies over the borrowed sums
  int64_t* st_lq = f.psum; int32_t* st_ptr = (int32_t*)(st_lq + FS_FC * FS_LV); int64_t* st_qty = (int64_t*)(st_ptr + FS_FC * FS_LV);
  int32_t* st_fr = (int32_t*)(st_qty + FS_FC); int32_t* st_plen = st_fr + FS_FC; int32_t* st_rb = st_plen + FS_FG;
  while (t >= 0) {
    if (staged) {
      const int nb = t + 1 < FS_FG ? t + 1 : FS_FG;
      {
        bool miss = false;
        for (int c = lane; c < FS_FC; c += WAVE) {
          const int rr = c / FS_RFR, u = c % FS_RFR;
          if (rr < nb) { const int fr = fs_sel8(fs_entries(S, S.fs_apply[(size_t)f.row0 + fs_tpos(f, t - rr)]).fr, u); if (fr >= 0 && f.colslot[fr] < 0) miss = true; }
        }
        if (wballot(miss)) fs_ensure_w(f);
      }
      for (int c = lane; c < FS_FC; c += WAVE) {
        const int rr = c / FS_RFR, u = c % FS_RFR;
        const bool in = rr < nb;
        const FsApply a = S.fs_apply[(size_t)f.row0 + fs_tpos(f, in ? t - rr : t)];
        const FsEnt en = fs_entries(S, a);
        const int fr = in ? fs_sel8(en.fr, u) : -1;
        int64_t lqv[FS_LV];
        #pragma unroll
        for (int i = 0; i < FS_LV; i++) lqv[i] = (fr >= 0 && i < (int)a.plen) ? S.fs_q[(size_t)(f.n0 + a.lp[i]) * f.nfr + fr].lq : 0;
        #pragma unroll
        for (int i = 0; i < FS_LV; i++) {
          st_lq[c * FS_LV + i] = fs_cap(lqv[i]);
          st_ptr[c * FS_LV + i] = (fr >= 0 && i < (int)a.plen) ? fs_cell(f, a.lp[i], fr) : 0;
        }
        st_qty[c] = fr >= 0 ? fs_sel8(en.qty, u) : 0;
        st_fr[c] = fr;
        if (u == 0 && in) { st_plen[rr] = a.plen; st_rb[rr] = 16 * (int)a.plen * (((int)a.cbytes - 32) / 12); }
      }
      wsync();
      CSTAT(18, 1);   // staged fill-back rounds
      bool failed = false;
      int done = 0;
      for (int rr = 0; rr < nb; rr++) {
        const int rplen = st_plen[rr], rb = st_rb[rr];
        for (int u = lane; u < FS_RFR; u += WAVE) {
          const int c = rr * FS_RFR + u;
          if (st_fr[c] >= 0) (void)fs_chain(f, st_ptr + c * FS_LV, st_lq + c * FS_LV, st_lq + c * FS_LV, rplen, st_qty[c], true, true);
        }
        wsync();
        if (lane == 0) w.bytes += rb;
        done = rr + 1;
        if (fs_fits(f, false)) {
          const int tt = t - rr;
          if (lane == 0) fs_tset(f, tt, fs_tget(f, nt - 1));
          nt--;
          *tbytes -= rb;
          wsync();
          continue;
        }
        for (int u = lane; u < FS_RFR; u += WAVE) {
          const int c = rr * FS_RFR + u;
          if (st_fr[c] >= 0) (void)fs_chain(f, st_ptr + c * FS_LV, st_lq + c * FS_LV, st_lq + c * FS_LV, rplen, st_qty[c], false, true);
        }
        wsync();
        if (lane == 0) w.bytes += rb;
        failed = true;
        break;
      }
      t -= done;
      if (!failed) continue;
    } else
    // one probe the walk's way
    {
      const FsRow r = fs_row_load(f, fs_tpos(f, t));
      fs_row_ctx(f, r);
      fs_row_apply(f, r, true, true, true);
      if (fs_fits(f, false)) {
        if (lane == 0) fs_tset(f, t, fs_tget(f, nt - 1));
        nt--;
        *tbytes -= r.rowbytes;
        wsync();
        t--;
        continue;
      }
      fs_row_apply(f, r, false, true, true);
      t--;
    }
    // it failed: skip over the failures that follow, 64 probes per round, up to the next probe that fits (the walk's step above takes it)
    while (t >= 0) {
      const int tt = t - lane;
      const bool in = tt >= 0;
      const int p = fs_tpos(f, in ? tt : 0);
      const FsApply a = S.fs_apply[(size_t)f.row0 + p];
      const FsEnt en = fs_entries(S, a);
      bool miss = false;
      #pragma unroll
      for (int e = 0; e < FS_RFR; e++) if (in && en.fr[e] >= 0 && f.colslot[en.fr[e]] < 0) miss = true;
      if (wballot(miss)) fs_ensure_w(f);
      const uint64_t fm = wballot(in && fs_probe_fits(f, p));
      const int span = t + 1 < WAVE ? t + 1 : WAVE;
      const int nfail = fm ? ffs64(fm) : span;
      // a failed probe: AddWorkload, the fit test, RemoveWorkload again
      const int64_t rowbytes = 16 * (int64_t)a.plen * (((int)a.cbytes - 32) / 12);
      bytes += wsum_i64(lane < nfail ? 2 * rowbytes + fits_bytes : 0);
      t -= nfail;
      if (fm) break;
    }
  }
  if (lane == 0) w.bytes += bytes;
  return nt;
}

// fairPreemptions (preemption.go:536-597): same contract as fair_search. false: preconditions not met, nothing was done.
KQ_NOINLINE bool fair_search_lds(Search& s) {
  const K& k = *s.k; Wave& w = *s.w; const DSnap& S = k.S;
  const int lane = lane_id();
  const bool same_on = KQ_POL_WITHIN_CQ(w.pol) != KQ_POLICY_NEVER;
  const bool other_on = w.plen > 1 && KQ_POL_RECLAIM(w.pol) != KQ_POLICY_NEVER;
  if (!same_on && !other_on) { w.ntgt = 0; return true; }
  s.tree = S.tree_of[w.cq];
  s.row0 = S.tree_row_off[s.tree];
  s.nrows = S.tree_row_off[s.tree + 1] - s.row0;
  if (s.nrows == 0) { w.ntgt = 0; return true; }
  Fs f;
  if (!fs_setup(s, f)) return false;
  w.ntgt = 0;
  KQ_T0();
  KQ_LZERO(w);
  fs_init(f);
  KQ_TS(k, 41);
  // ---- findCandidates (:633-667) ----
  for (int i = lane; i < f.nqs; i += WAVE) {  // does the ClusterQueue contribute candidates?
    const int c = S.tree_cqs[f.q0 + i];
    bool take;
    if (i == f.wli) take = same_on;
    else {
      take = false;
      if (other_on)
        for (int u = 0; u < w.ns; u++)  // cqIsBorrowing :657-667
          if (w.s_need[u] && S.nominal[ix(S, c, w.s_fr[u])] < f.usage[ix(S, c, w.s_fr[u])]) take = true;
    }
    if (take) f.nflag[i] |= 4;
  }
  wsync();
  {
    // Round 6: the candidates come from the (tree, flavor-resource) buckets of the flavor-resources that need preemption (Prep::frb /
    // frec: every row using that flavor-resource, the policy operands beside it) instead of a scan over every row of the tree — a
    // candidate uses one of them by definition (:606, :618). At BASELINE configs[3] a bucket holds 1/16 of the 40 k rows; the scan was a
    // third of a search (profiles/r06i_*). The candidate records the reference reads are those of every workload of a contributing
    // ClusterQueue, whatever it uses: charged per ClusterQueue (cq_row_bytes minus the rows preempted so far), as the classical search does.
    int64_t cbytes = 0;
    for (int i = lane; i < f.nqs; i += WAVE)
      if (f.nflag[i] & 4) { const int c = S.tree_cqs[f.q0 + i]; cbytes += S.cq_row_bytes[c] - (f.removed ? k.cq_rm_bytes[c] : 0); }
    const int policy_same = KQ_POL_WITHIN_CQ(w.pol), policy_other = KQ_POL_RECLAIM(w.pol);
    // a record's verdict (findCandidatesForPolicy :599-627)
    auto cand_ok = [&](const CsRec& rc) -> bool {
      if (!(f.nflag[rc.cql] & 4)) return false;
      if (f.removed && f.removed[rc.row]) return false;
      const int policy = rc.cql == f.wli ? policy_same : policy_other;
      const bool lower = w.prio > rc.prio;
      bool ok = policy == KQ_POLICY_ANY;
      if (policy == KQ_POLICY_LOWER_PRIORITY) ok = lower;
      if (policy == KQ_POLICY_LOWER_OR_NEWER_EQUAL) ok = lower || (w.prio == rc.prio && w.ts < rc.qts);
      return ok;
    };
    for (int u = 0; u < w.ns; u++) {
      if (!w.s_need[u]) continue;
      bool dup = false;
      for (int v = 0; v < u; v++) if (w.s_need[v] && w.s_fr[v] == w.s_fr[u]) dup = true;
      if (dup) continue;
      const size_t b = (size_t)f.tree * S.nfr + w.s_fr[u];
      const int j0 = S.frb_off[b], j1 = S.frb_off[b + 1];
      // four records of a lane in flight together, then the bytes / positions that depend on them
      constexpr int UN = 4;
      for (int base = j0; base < j1; base += WAVE * UN) {
        CsRec rc[UN]; bool in[UN], ok[UN];
        #pragma unroll
        for (int q = 0; q < UN; q++) { const int j = base + q * WAVE + lane; in[q] = j < j1; rc[q] = S.frec[in[q] ? j : j1 - 1]; }
        #pragma unroll
        for (int q = 0; q < UN; q++) ok[q] = in[q] && cand_ok(rc[q]);
        int pos[UN];
        #pragma unroll
        for (int q = 0; q < UN; q++) pos[q] = ok[q] ? S.adm_rec[rc[q].row].fs_pos : 0;
        #pragma unroll
        for (int q = 0; q < UN; q++) if (ok[q]) atomic_or_u64(&f.m1[pos[q] >> 6], 1ull << (pos[q] & 63));
      }
    }
    const int64_t tot = wsum_i64(cbytes);
    if (lane == 0) w.bytes += tot;
  }
  wsync();
  int ncand = 0;
  for (int i = lane; i < f.nqs; i += WAVE) fs_has_update(f, i);
  wsync();
  for (int base = 0; base < f.nqs; base += WAVE) { const int i = base + lane; ncand += popc64(wballot(i < f.nqs && fs_cq_has(f, i))); }
  CSTAT(14, 1); CSTAT(15, ncand);
  KQ_TS(k, 42);
  if (ncand == 0) return true;
  fs_pc_setup(f);
  fs_pc_apply(f, true);  // SimulateUsageAddition :557
  int nt = 0;
  int64_t tbytes = 0;  // what the final bookkeeping of the targets is charged (16 B per level and usage entry)
  bool fits = false;
  const int strategy0 = k.C.n_fs > 0 ? k.C.fs[0] : KQ_FS_LESS_THAN_OR_EQUAL_TO_FINAL_SHARE;
  const bool have_second = k.C.n_fs > 0 ? k.C.n_fs > 1 : true;
  // ---- runFirstFsStrategy :384-470 ----
  {
    bool within_nominal = false;
    if (gate(k, KQ_GATE_FS_PREEMPT_WITHIN_NOMINAL)) {  // queueWithinNominalInResourcesNeedingPreemption :714-721
      within_nominal = true;
      for (int u = 0; u < w.ns; u++)
        if (w.s_need[u] && S.nominal[ix(S, w.cq, w.s_fr[u])] < fs_ld(f, fs_cell(f, f.wli, w.s_fr[u]))) within_nominal = false;
    }
    int streak = 0;   // ClusterQueues visited since the last victim
    while (!fits) {
      KQ_A0();
      const int cand = fs_ordering_next(f);
      KQ_LS(w, 0);
      if (cand < 0) break;
      if (cand == f.wli || within_nominal) {
        const int p = fs_pop(f, cand, f.m1, nullptr);
        const FsRow r = fs_row_load(f, p);
        fs_row_ctx(f, r);
        fs_row_apply(f, r, false, true, true);
        if (!fs_push_target(f, &nt, r.row, p, cand == f.wli ? KQ_REASON_IN_CLUSTER_QUEUE : KQ_REASON_IN_COHORT_RECLAMATION)) { w.ntgt = 0; return true; }
        tbytes += r.rowbytes;
        if (fs_fits_fs(f)) fits = true;
        streak = 0;
        continue;
      }
      // The batch pays when candidates FAIL (it skips over them); where most of them pass — a recomputation inside processEntry, whose
      // snapshot is no longer over-committed — a batch per victim costs twice the walk's single evaluation (k_process_fair 15.7 s with the
      // batch always on against 10.6 s without, profiles/r04e_*). So a visit takes the batch only while the walk is in a losing streak:
      // the previous ClusterQueue was exhausted without a victim.
      if ((k.C.fs_batch & 1) && (streak > 0 || (k.C.fs_batch & 16))) {
        int bp = -1;
        const int br = fs_batch(f, cand, strategy0, false, &bp);
        KQ_LS(w, 1);
        if (br == 0) continue;
        if (br == 1) {
          streak = 0;
          const FsRow r = fs_row_load(f, bp);
          fs_row_ctx(f, r);
          fs_row_apply(f, r, false, true, true);
          KQ_LS(w, 4);
          if (!fs_push_target(f, &nt, r.row, bp, KQ_REASON_IN_COHORT_FAIR_SHARING)) { w.ntgt = 0; return true; }
          tbytes += r.rowbytes;
          if (fs_fits_fs(f)) fits = true;
          KQ_LS(w, 6);
          continue;
        }
      }
      int ap = f.wli, at = cand;
      fs_lca_get(f, cand, &ap, &at);
      const int pz = (f.nflag[ap] & 2) ? 1 : 0, tz = (f.nflag[at] & 2) ? 1 : 0;
      const uint64_t pk = fs_okey(f.dval[ap]), tk = fs_okey(f.dval[at]);
      if (lane == 0) w.bytes += fs_cost(f, ap) + fs_cost(f, at);
      if (fs_pos_inf(f, ap) && !fs_pos_inf(f, at)) {  // fsStrategyUnsatisfiable :494-497: the whole queue goes to retryCandidates
        const int a = f.posoff[cand], b = f.posoff[cand + 1];
        wsync();
        for (int wi = (a >> 6) + lane; wi <= ((b - 1) >> 6); wi += WAVE) {
          uint64_t mask = ~0ull;
          if (wi == (a >> 6)) mask &= ~0ull << (a & 63);
          if (wi == ((b - 1) >> 6)) { const int e = (b - 1) & 63; if (e < 63) mask &= (2ull << e) - 1; }
          const uint64_t mv = f.m1[wi] & mask;
          CSTAT(13, popc64(mv));
          f.m2[wi] |= mv; f.m1[wi] &= ~mask;
        }
        if (lane == 0) f.nflag[cand] &= ~8;
        wsync();
        streak++;
        continue;
      }
      KQ_LS(w, 7);
      streak++;   // (reset below if this ClusterQueue yields a victim)
      while (fs_cq_has(f, cand)) {
        const int p = fs_pop(f, cand, f.m1, nullptr);
        KQ_LS(w, 1);
        const FsRow r = fs_row_load(f, p);
        fs_row_ctx(f, r);
        KQ_LS(w, 2);
        // ComputeTargetShareAfterRemoval target.go:67-73: RemoveWorkload, share of the target's side, AddWorkload. The state after
        // the pair is the state before it (plain amounts, see fs_fits), so the removal is only evaluated.
        uint64_t nk = 0;
        const int zb = fs_row_apply(f, r, false, false, false, at, &nk);
        KQ_LS(w, 3);
        const int nz = zb & 1;
        if (lane == 0) w.bytes += (int64_t)f.c0[at] + ((zb & 2) ? (int64_t)f.c1[at] : 0);
        const bool pass = strategy0 == KQ_FS_LESS_THAN_OR_EQUAL_TO_FINAL_SHARE ? fs_cmp(pz, pk, nz, nk) <= 0 : fs_cmp(pz, pk, tz, tk) < 0;  // strategy.go:41,46
        if (pass) {
          fs_row_commit_evaluated(f, r);
          KQ_LS(w, 4);
          if (!fs_push_target(f, &nt, r.row, p, KQ_REASON_IN_COHORT_FAIR_SHARING)) { w.ntgt = 0; return true; }
          tbytes += r.rowbytes;
          if (fs_fits_fs(f)) fits = true;
          KQ_LS(w, 6);
          streak = 0;
          break;
        }
        if (lane == 0) f.m2[p >> 6] |= 1ull << (p & 63);  // retryCandidates
        wsync();
      }
    }
  }
  KQ_TS(k, 43);
  // ---- runSecondFsStrategy :501-534 ----
  if (!fits && have_second) {
    f.mq = f.m2;
    for (int i = lane; i < f.nn; i += WAVE) { f.nflag[i] &= ~1; if (i < f.nqs) fs_has_update(f, i); }
    wsync();
    int streak2 = 0;
    while (!fits) {
      const int cand = fs_ordering_next(f);
      if (cand < 0) break;
      if ((k.C.fs_batch & 2) && (streak2 > 0 || (k.C.fs_batch & 16))) {
        int bp = -1;
        const int br = fs_batch(f, cand, strategy0, true, &bp);
        if (br == 0) continue;
        if (br == 1) {
          streak2 = 0;
          const FsRow r = fs_row_load(f, bp);
          fs_row_ctx(f, r);
          fs_row_apply(f, r, false, true, true);
          if (!fs_push_target(f, &nt, r.row, bp, KQ_REASON_IN_COHORT_FAIR_SHARING)) { w.ntgt = 0; return true; }
          tbytes += r.rowbytes;
          if (fs_fits_fs(f)) fits = true;
          continue;
        }
      }
      int ap = f.wli, at = cand;
      fs_lca_get(f, cand, &ap, &at);
      const bool passed = fs_cmp((f.nflag[ap] & 2) ? 1 : 0, fs_okey(f.dval[ap]), (f.nflag[at] & 2) ? 1 : 0, fs_okey(f.dval[at])) < 0;
      if (lane == 0) w.bytes += fs_cost(f, ap) + fs_cost(f, at);
      const int p = fs_pop(f, cand, f.m2, nullptr);
      streak2 = passed ? 0 : streak2 + 1;
      if (passed) {
        const FsRow r = fs_row_load(f, p);
        fs_row_ctx(f, r);
        fs_row_apply(f, r, false, true, true);
        if (!fs_push_target(f, &nt, r.row, p, KQ_REASON_IN_COHORT_FAIR_SHARING)) { w.ntgt = 0; return true; }
        tbytes += r.rowbytes;
        if (fs_fits_fs(f)) fits = true;
      }
      if (lane == 0) f.nflag[cand] |= 1;  // DropQueue
      wsync();
    }
  }
  fs_pc_apply(f, false);  // revertSimulation
  KQ_TS(k, 44);
  bool as_started = false;
  if (!fits) {
    if (lane == 0) w.bytes += tbytes;
    // restoreSnapshot :356 — the private copy is dropped, but callers read it on the preemptor's path. With every row back the state is
    // the one the search started from (plain amounts: usage is a function of the SET of rows present), so nothing is walked back.
    if (k.C.fs_batch & 8) as_started = true;
    else for (int t = 0; t < nt; t++) { const FsRow r = fs_row_load(f, fs_tpos(f, t)); fs_row_ctx(f, r); fs_row_apply(f, r, true, true, false); }
    w.ntgt = 0;
    KQ_TS(k, 45);
  } else {
    CSTAT(16, 1); CSTAT(17, nt);
    // fillBackWorkloads :341-354 with allowBorrowing = true
    if (k.C.fs_batch & 4) nt = fs_fillback_batch(f, nt, &tbytes);
    else
    for (int t = nt - 2; t >= 0; t--) {
      const FsRow r = fs_row_load(f, fs_tpos(f, t));
      fs_row_ctx(f, r);
      fs_row_apply(f, r, true, true, true);
      if (fs_fits(f, false)) {
        if (lane == 0) fs_tset(f, t, fs_tget(f, nt - 1));
        nt--;
        tbytes -= r.rowbytes;
        wsync();
      } else {
        fs_row_apply(f, r, false, true, true);
      }
    }
    w.ntgt = nt;
    fs_tflush(f, nt);
    if (lane == 0) w.bytes += tbytes;
    KQ_TS(k, 46);
  }
  KQ_LFLUSH(k, w, 47);
  // callers read the private plane on the preemptor's path (find_height in simulate_preemption)
  for (int c = lane; c < f.npc * FS_LV; c += WAVE) {
    const int j = c / FS_LV, i = c % FS_LV;
    if (i >= f.plen) continue;
    const int fr = w.s_fr[f.pc_u[j]];
    f.W[(size_t)w.cs_pl[i] * f.nfr + fr] = as_started ? f.usage[ix(S, w.path[i], fr)] : fs_ld(f, f.pc_ptr[c]);
  }
  wsync();
  return true;
}

}  // namespace kq

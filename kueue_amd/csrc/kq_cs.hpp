// kq_cs.hpp — the classical victim search (preemption.go:284-354) as segmented scans over the candidate list.
//
// The reference walks the ordered candidates one by one: re-check validity on the mutated snapshot
// (classical/candidate_generator.go:136-158), RemoveWorkload, workloadFits, stop at the first fit, then fillBackWorkloads. On
// a device that is ~3000 dependent steps per search (cfg 4: 2500 candidates visited, ~370 removed, ~370 fill-back probes), 38
// searches per head. This file computes the SAME sequence without walking it:
//
//  * removeUsage (resource_node.go:156-165) moves a parent by exactly what the child stops storing in it,
//    B(n) = max(0, usage(n) - localQuota(n)); so the private state of a search is a function of the SET of removed rows, and
//    while candidates are only removed every usage cell only goes down. "ClusterQueue / cohort within nominal" therefore flips
//    once: every node below the preemptor's path has a death time, after which the candidates under it are skipped.
//  * a candidate's validity only reads nodes strictly below its lowest common ancestor with the preemptor, and only removals
//    inside such a node's subtree move it. Per level (ClusterQueue, its cohort, ...), the candidates of a node are a contiguous
//    segment of a static "level order" of the flavor-resource's bucket (kq_prep.hpp CsEnt), time-ordered inside the segment:
//    one SEGMENTED PREFIX SUM of the removed quantities gives the node's usage before every candidate, hence the death point
//    (ballot) and what each removal passes on to the parent. Levels are processed bottom-up; lanes = candidates.
//  * the first fit is a prefix problem in global candidate order: per path level, the prefix sum of what arrives there gives
//    the preemptor path's usage after every removal; every lane evaluates workloadFits (preemption.go:669-686) for its
//    prefix, first set ballot bit = the reference's stopping point.
//  * fillBackWorkloads probes the targets newest-first; probes are evaluated 64 at a time against the current state (adding
//    usage back only makes later probes harder, so a failed probe stays failed) and only accepted ones are applied in order.
//
// Bit-exact with the walk (tests: emulation and GPU against the oracle, with this path forced on and off). Anything outside
// its preconditions (fair sharing, > CS_NS slots, need flavor-resources with different row sets, deep trees, rows with many
// flavor-resources, non-plain amounts) returns false and the generic walk in kq_device.hpp runs instead.
#pragma once

namespace kq {

struct CsCtx {
  const K* k; Wave* w; Search* s;
  int ns, M, Mp, nn, n0, boff, Mt, plen, levels, ufirst;
  int64_t* W;            // [ns][nn]  private usage, slot-major
  int64_t* dB[CS_NS];    // [Mp] per slot: what the candidate's removal takes out of the node one level up (level by level, in place)
  int64_t *sqT, *lqT;    // [ns][nn]  SubtreeQuota / localQuota of every node of the tree for the slots; null: read the planes
  uint16_t *tord, *tinv, *rb;  // [Mp] candidate at time t / time of candidate j (0xffff = never) / AdmRec::rowbytes
  uint16_t* lstL[CS_LEVELS];   // [Mp] per level: the level-order positions of the candidates inside a time prefix (cs_gather)
  int nL[CS_LEVELS]; int lst_T; // their lengths; the prefix they were gathered for (-1: none)
  uint8_t *alive, *cls, *att;  // [Mp] still removable / class byte (classical_search) / path level the candidate's branch hangs off
  uint8_t* cqi;          // [tree ClusterQueues] Search::cqinfo
  uint32_t needm, inum;
};

#ifdef KQ_HOST_EMU
static inline
#else
__host__ __device__ inline
#endif
size_t cs_bytes(int ns, int M, int nn, int nqs, bool tables) {  // everything a search with ns slots allocates
  const size_t Mp = ((size_t)M + 63) & ~(size_t)63;
  return (size_t)ns * nn * 8 * (tables ? 3 : 1) + (size_t)ns * Mp * 8 + Mp * 2 * (3 + CS_LEVELS) + Mp * 3 + (((size_t)nqs + 15) & ~(size_t)15) + 512;
}
// Arrays are placed in the workgroup's LDS region in order of heat while they fit, the rest in the wave slot's HBM spill space:
// the byte arrays and the private usage first, then the quota tables, then the per-slot quantity arrays.
struct CsCarve {
  unsigned char *a, *ae, *b;  // LDS cursor / end, spill cursor
  KQ_MDEV void* take(size_t bytes, bool* in_lds = nullptr) {
    bytes = (bytes + 15) & ~(size_t)15;
    if (a && a + bytes <= ae) { void* p = a; a += bytes; if (in_lds) *in_lds = true; return p; }
    void* p = b; b += bytes; if (in_lds) *in_lds = false; return p;
  }
};
KQ_DEV int64_t cs_rec_qty(const DSnap& S, const AdmRec& r, int row, int fr) {
  int64_t q = 0;
  #pragma unroll
  for (int e = 0; e < CS_RFR; e++) if (r.fr[e] == fr) q = r.qty[e];
  if (r.flags & 2u) {   // a wide row (kq_prep.hpp AdmRecX): rare, one more record
    const AdmRecX x = S.adm_recx[row];
    #pragma unroll
    for (int e = 0; e < CS_RFX; e++) if (x.fr[e] == fr) q = x.qty[e];
  }
  return q;
}
KQ_DEV int64_t cs_B(int64_t u, int64_t lq) { return i64max(0, a_sub(u, lq)); }
// workloadFits for one slot given the usage of the preemptor's path (preemption.go:669-686, resource_node.go:106-122)
// (arrays indexed by slot / level are fully unrolled with guards everywhere in this file: a runtime index would put them in
// scratch, and this compiler mis-selects the private->flat cast of a pointer into such an array on gfx950)
struct PathU { int64_t v[CS_LEVELS + 1]; };
KQ_DEV bool cs_fits_slot(const Wave& w, int u, int plen, const PathU uv, bool allow_borrow) {
  const int64_t val = w.s_qty[u];
  if (!allow_borrow && w.cs_nom[u] < a_add(uv.v[0], val)) return false;
  int64_t a = 0;
  #pragma unroll
  for (int i = CS_LEVELS; i >= 0; i--) {
    if (i >= plen) continue;
    if (i == plen - 1) { a = a_sub(w.cs_sq[u][i], uv.v[i]); continue; }
    const int64_t lq = w.cs_lq[u][i], blv = w.cs_bl[u][i];
    if (blv != KQ_NIL_LIMIT) a = i64min(a_add(a_sub(a_sub(w.cs_sq[u][i], lq), i64max(0, a_sub(uv.v[i], lq))), blv), a);
    a = a_add(i64max(0, a_sub(lq, uv.v[i])), a);
  }
  return !(val > i64max(0, a));
}
// tree-local node -> level on the preemptor's path, -1 = not on it
KQ_DEV int cs_path_level(const Wave& w, int plen, int node_local) {
  int r = -1;
  for (int i = 0; i < plen; i++) if (w.cs_pl[i] == node_local) r = i;
  return r;
}

// A time prefix (limit_t in front of the last candidate: the lazy first rounds of cs_run, the finalising passes) only involves the
// candidates removed at or before limit_t: their level-order positions, for every level at once — one light pass over the bucket (8 bytes
// per entry and level, the levels' loads in flight together) instead of one per level and pass: the gathers were most of a
// recomputation's level passes (profiles/r06l_prof_loop_cfg4c_feasible_process_only.txt). The lists serve every pass up to limit_t: the
// finalising passes (limit = the stopping time) reuse the prefix round's. Segments (same node) stay contiguous and time-ordered.
KQ_DEV void cs_gather(CsCtx& c, int limit_t) {
  const DSnap& S = c.k->S;
  const int lane = lane_id(), M = c.M;
  int n[CS_LEVELS];
  #pragma unroll
  for (int l = 0; l < CS_LEVELS; l++) n[l] = 0;
  for (int base = 0; base < M; base += WAVE) {
    const int q = base + lane;
    int jd[CS_LEVELS], nd[CS_LEVELS];
    #pragma unroll
    for (int l = 0; l < CS_LEVELS; l++) {
      jd[l] = 0; nd[l] = -1;
      if (l < c.levels && q < M) { const CsEnt* e = S.frl[l] + c.boff + q; jd[l] = e->jd; nd[l] = e->node; }
    }
    #pragma unroll
    for (int l = 0; l < CS_LEVELS; l++) {
      if (l >= c.levels) continue;
      const bool act = nd[l] >= 0 && (int)c.tinv[jd[l] & 0xffffff] <= limit_t;
      const uint64_t m = wballot(act);
      if (act) c.lstL[l][n[l] + popc64(m & ((1ull << lane) - 1))] = (uint16_t)q;
      n[l] += popc64(m);
    }
  }
  #pragma unroll
  for (int l = 0; l < CS_LEVELS; l++) c.nL[l] = n[l];
  c.lst_T = limit_t;
  wsync();
}

// One bottom-up level: the nodes at depth dd below the root (passes run from the deepest level up to 1; the root is on every
// preemptor's path). limit_t: only candidates removed at or before that time count. finalize: leave every node's usage after
// its last counted removal in W (second run, after the stopping time is known); otherwise mark the candidates that meet a dead
// node (alive = 0). A candidate enters at the depth of its ClusterQueue with its own quantities and carries, from then on, what
// its removal takes out of the node one level up.
KQ_DEV void cs_level_pass(CsCtx& c, int dd, int limit_t, bool finalize) {
  const K& k = *c.k; const DSnap& S = k.S; Wave& w = *c.w;
  const int lane = lane_id(), ns = c.ns, M = c.M;
  const CsEnt* ents = S.frl[dd - 1] + c.boff;
  int carry_node = -2;
  int64_t carry[CS_NS];
  #pragma unroll
  for (int u = 0; u < CS_NS; u++) carry[u] = 0;
  // a time prefix runs over the gathered candidates alone (cs_gather); a list gathered for a longer prefix serves too: the entries
  // behind limit_t take part with zero quantities (`al` below) and change nothing
  const bool packed = limit_t + 1 < c.Mt;
  int n = M;
  const uint16_t* lst = nullptr;
  if (packed) {
    if (c.lst_T < limit_t) cs_gather(c, limit_t);
    n = c.nL[dd - 1]; lst = c.lstL[dd - 1];
    if (n == 0) return;
  }
  auto pos_of = [&](int i) -> int { const int ii = i < n ? i : n - 1; return packed ? (int)lst[ii] : ii; };
  // the entries are static and read front to back: the next chunk's load is in flight while this one is processed
  CsEnt e_next = ents[pos_of(lane)];
  for (int base = 0; base < n; base += WAVE) {
    const int q = base + lane;
    const bool in = q < n;
    const CsEnt e = e_next;
    e_next = ents[pos_of(q + WAVE)];
    const int node_prev = wshift_up_i32(e.node);
    const bool head = in && (lane == 0 ? e.node != carry_node : e.node != node_prev);
    const uint64_t H = wballot(head);
    const uint64_t below = H & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
    const int hl = below ? 63 - clz64(below) : -1;
    const int plv = e.node >= 0 ? cs_path_level(w, c.plen, e.node) : -1;
    const bool part = in && e.node >= 0 && plv < 0;
    const int j = e.jd & 0xffffff;
    const bool enters = in && e.node >= 0 && (e.jd >> 24) == dd;  // the node is the row's ClusterQueue
    const bool al = in && e.node >= 0 && c.alive[j] != 0 && (int)c.tinv[j] <= limit_t;
    int64_t d[CS_NS];
    {
      AdmRec r{};
      if (enters && al && ns > 1) r = S.adm_rec[e.row];  // the other slots' quantities; the bucket's own one travels with the entry
      #pragma unroll
      for (int u = 0; u < CS_NS; u++)
        d[u] = (u < ns && al) ? (enters ? (u == c.ufirst ? e.qty : cs_rec_qty(S, r, e.row, w.s_fr[u])) : c.dB[u][j]) : 0;
    }
    bool within = true;   // IsWithinNominalInResources on the state before this candidate (resource_node.go:247-254)
    int64_t ua[CS_NS], outv[CS_NS];
    #pragma unroll
    for (int u = 0; u < CS_NS; u++) {
      if (u >= ns) { ua[u] = 0; outv[u] = 0; continue; }
      int64_t sqv = 0, lq = 0, u0 = 0;
      if (part) {
        if (c.sqT) { sqv = c.sqT[(size_t)u * c.nn + e.node]; lq = c.lqT[(size_t)u * c.nn + e.node]; }
        else {
          const size_t o = ix(S, e.gnode, w.s_fr[u]);
          sqv = S.sq[o];
          const int64_t llv = S.ll[o];
          lq = llv != KQ_NIL_LIMIT ? i64max(0, a_sub(sqv, llv)) : 0;
        }
        u0 = c.W[(size_t)u * c.nn + e.node];
      }
      const int64_t P = wprefix_incl_i64(d[u]);
      const int64_t Pex = P - d[u];
      const int64_t hb = wshfl_i64(Pex, hl < 0 ? 0 : hl);
      const int64_t ex = hl < 0 ? carry[u] + Pex : Pex - hb;
      const int64_t inc = ex + d[u];
      const int64_t ub = a_sub(u0, ex);
      ua[u] = a_sub(u0, inc);
      if (((c.needm >> u) & 1) && sqv < ub) within = false;
      outv[u] = cs_B(ub, lq) - cs_B(ua[u], lq);
      carry[u] = wbcast_u(inc, WAVE - 1);
    }
    carry_node = wbcast_u(e.node, WAVE - 1);
    const bool dead = part && within;
    if (!finalize && dead && in && c.alive[j]) c.alive[j] = 0;
    #pragma unroll
    for (int u = 0; u < CS_NS; u++) {
      if (u >= ns || !in) continue;
      if (part) { if (al && !dead) c.dB[u][j] = outv[u]; }
      else if (enters && (!finalize || (int)c.tinv[j] <= limit_t)) c.dB[u][j] = d[u];  // the preemptor's own ClusterQueue: straight onto the path
    }
    if (finalize) {
      // the node's usage after its last counted removal: written by the last entry of the segment
      const int nsh = wshift_down_i32(e.node);
      const int nfirst = wbcast_u(e_next.node, 0);  // first entry of the next chunk
      int nxt = -9;
      if (q + 1 < n) nxt = lane == WAVE - 1 ? nfirst : nsh;
      #pragma unroll
      for (int u = 0; u < CS_NS; u++) if (u < ns && part && nxt != e.node) c.W[(size_t)u * c.nn + e.node] = ua[u];
    }
    wsync();
  }
}

// Runs the search for the slots prepared in w (after classical_search has filled adv_at / cqinfo and charged the candidate
// records). Returns false when the preconditions do not hold (nothing has been changed then).
KQ_DEV bool cs_run(Search& s, bool same_on, bool other_on) {
  const K& k = *s.k; Wave& w = *s.w; const DSnap& S = k.S;
  const int lane = lane_id();
  if (!k.C.cs_on || k.C.fair_sharing || !S.cs_ok || !S.cs_ok[s.tree]) return false;
  const int ns = w.ns, plen = w.plen;
  if (ns > CS_NS || ns < 1 || plen > CS_LEVELS + 1) return false;
  // candidates = the rows using a flavor-resource that needs preemption: one bucket when all of those hold the same rows
  int first = -1, M = 0, boff = 0;
  uint32_t needm = 0, inum = 0;
  for (int u = 0; u < ns; u++) {
    if (w.s_inu[u]) inum |= 1u << u;
    if (!w.s_need[u]) continue;
    needm |= 1u << u;
    const size_t b = (size_t)s.tree * S.nfr + w.s_fr[u];
    const int m = S.frb_off[b + 1] - S.frb_off[b];
    if (first < 0) { first = u; M = m; boff = S.frb_off[b]; }
    else if (m != M || S.frb_sig[b] != S.frb_sig[(size_t)s.tree * S.nfr + w.s_fr[first]]) return false;
  }
  if (first < 0 || M == 0 || M > 0xfff0) return false;
  const int n0 = S.tree_node_off[s.tree], nn = S.tree_node_off[s.tree + 1] - n0;
  const int q0 = S.tree_cq_off[s.tree], nqs = S.tree_cq_off[s.tree + 1] - q0;
  if (!k.X.cs || cs_bytes(ns, M, nn, nqs, false) > (size_t)k.X.cs_bytes) return false;
  CsCtx c;
  c.k = &k; c.w = &w; c.s = &s; c.ns = ns; c.M = M; c.Mp = (M + 63) & ~63; c.nn = nn; c.n0 = n0; c.boff = boff; c.plen = plen;
  c.levels = S.tree_depth[s.tree] < CS_LEVELS ? S.tree_depth[s.tree] : CS_LEVELS;
  c.needm = needm; c.inum = inum; c.ufirst = first;
  {
    CsCarve cv{w.cs_lds, w.cs_lds ? w.cs_lds + w.cs_lds_bytes : nullptr, k.X.cs + (size_t)s.slot * k.X.cs_bytes};
    c.alive = (uint8_t*)cv.take(c.Mp); c.cls = (uint8_t*)cv.take(c.Mp); c.att = (uint8_t*)cv.take(c.Mp);
    c.tord = (uint16_t*)cv.take((size_t)c.Mp * 2); c.tinv = (uint16_t*)cv.take((size_t)c.Mp * 2); c.rb = (uint16_t*)cv.take((size_t)c.Mp * 2);
    for (int l = 0; l < CS_LEVELS; l++) c.lstL[l] = (uint16_t*)cv.take((size_t)c.Mp * 2);
    c.lst_T = -1;
    c.cqi = (uint8_t*)cv.take(nqs);
    c.W = (int64_t*)cv.take((size_t)ns * nn * 8);
    // the quota tables only pay off next to the arithmetic: in LDS or not at all (the planes are L2-resident anyway)
    c.sqT = c.lqT = nullptr;
    if (cv.a && cv.a + 2 * (((size_t)ns * nn * 8 + 15) & ~(size_t)15) <= cv.ae) { c.sqT = (int64_t*)cv.take((size_t)ns * nn * 8); c.lqT = (int64_t*)cv.take((size_t)ns * nn * 8); }
    #pragma unroll
    for (int u = 0; u < CS_NS; u++) c.dB[u] = u < ns ? (int64_t*)cv.take((size_t)c.Mp * 8) : nullptr;
  }
  CSTAT(20, 1);
  KQ_T0();
  // ---- private copy of the tree's usage (and quotas) for the slots; constants of the preemptor's path ----
  for (int i = lane; i < nn * ns; i += WAVE) {
    const int u = i / nn, ln = i % nn;
    const size_t o = ix(S, S.tree_nodes[n0 + ln], w.s_fr[u]);
    c.W[i] = s.usage[o];
    if (c.sqT) {
      const int64_t sqv = S.sq[o], llv = S.ll[o];
      c.sqT[i] = sqv; c.lqT[i] = llv != KQ_NIL_LIMIT ? i64max(0, a_sub(sqv, llv)) : 0;
    }
  }
  for (int i = lane; i < nqs; i += WAVE) c.cqi[i] = s.cqinfo[i];
  for (int i = lane; i < ns * plen; i += WAVE) {
    const int u = i / plen, l = i % plen, n = w.path[l];
    const size_t o = ix(S, n, w.s_fr[u]);
    const int64_t sqv = S.sq[o], llv = S.ll[o];
    w.cs_sq[u][l] = sqv; w.cs_bl[u][l] = S.bl[o];
    w.cs_lq[u][l] = llv != KQ_NIL_LIMIT ? i64max(0, a_sub(sqv, llv)) : 0;
    w.cs_u0[u][l] = s.usage[o];
    if (l == 0) w.cs_nom[u] = S.nominal[o];
    if (u == 0) w.cs_pl[l] = S.node_local[n];
  }
  wsync();
  KQ_TS(k, 35);  // scan search: private usage / quota tables / path constants
  // ---- classify the bucket once (hierarchical_preemption.go:81-113); class byte = 1 + list + 3 * not-evicted + 8 * variant ----
  const int32_t* rows = S.frbr + boff;
  const CsRec* recs = S.frec + boff;
  const int own_cql = S.cq_local[w.cq];
  int cnt[6] = {0, 0, 0, 0, 0, 0};
  CsRec r_next = recs[lane < M ? lane : M - 1];
  for (int base = 0; base < M; base += WAVE) {
    const int j = base + lane;
    uint8_t cb = 0, at = 0;
    const CsRec r = r_next;
    { const int jn = j + WAVE; r_next = recs[jn < M ? jn : M - 1]; }
    if (j < M) {
      const int row = r.row;
      const bool same = r.cql == own_cql;
      int level = 0;
      bool ok = !row_removed(s, row);
      if (ok) {
        if (same) ok = same_on;
        else { const uint8_t info = c.cqi[r.cql]; ok = info != 0; level = info - 1; }
      }
      if (ok) {
        const int policy = same ? KQ_POL_WITHIN_CQ(w.pol) : KQ_POL_RECLAIM(w.pol);
        const bool lower = w.prio > r.prio;
        if (policy == KQ_POLICY_LOWER_PRIORITY) ok = lower;
        else if (policy == KQ_POLICY_LOWER_OR_NEWER_EQUAL) ok = lower || (w.prio == r.prio && w.ts < r.qts);
        else ok = policy == KQ_POLICY_ANY;
      }
      if (ok) {
        int list, v;
        if (same) { list = 2; v = V_WITHIN_CQ; }
        else if (w.adv_at[level]) { list = 0; v = V_HIER; }
        else {
          list = 1;
          if (KQ_POL_BORROW_WITHIN(w.pol) == 0 || r.prio >= w.prio || (KQ_POL_HAS_THRESHOLD(w.pol) && r.prio > (int64_t)S.cq_thr[w.cq])) v = V_RECLAIM_NO_BORROW;
          else v = V_RECLAIM_BORROW;
        }
        const int ev = (r.flags & 1u) ? 0 : 1;
        cb = (uint8_t)(1 + (ev * 3 + list) + 8 * v);
        at = (uint8_t)level;
      }
      c.cls[j] = cb; c.att[j] = at; c.rb[j] = (uint16_t)r.rowbytes;
    }
    for (int p = 0; p < 6; p++) cnt[p] += popc64(wballot(cb != 0 && ((cb - 1) & 7) == p));
  }
  (void)other_on;
  const int Mt = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4] + cnt[5];
  c.Mt = Mt;
  w.ntgt = 0;
  if (Mt == 0) return true;
  wsync();
  {  // time order = the six lists one after another, rank order inside a list (preemption.go:299-309)
    int run[6], acc = 0;
    for (int p = 0; p < 6; p++) { run[p] = acc; acc += cnt[p]; }
    for (int base = 0; base < M; base += WAVE) {
      const int j = base + lane;
      const uint8_t cb = j < M ? c.cls[j] : 0;
      for (int p = 0; p < 6; p++) {
        const bool mine = cb != 0 && ((cb - 1) & 7) == p;
        const uint64_t m = wballot(mine);
        if (mine) { const int t = run[p] + popc64(m & ((1ull << lane) - 1)); c.tord[t] = (uint16_t)j; c.tinv[j] = (uint16_t)t; }
        run[p] += popc64(m);
      }
      if (j < M && cb == 0) c.tinv[j] = 0xffff;
    }
  }
  wsync();
  KQ_TS(k, 36);  // scan search: classification + time order
  const bool no_hier = cnt[0] + cnt[3] == 0, no_other = no_hier && (cnt[1] + cnt[4] == 0);
  const bool forbidden = KQ_POL_BORROW_WITHIN(w.pol) == 0;
  bool under_nominal;  // queueUnderNominalInResourcesNeedingPreemption preemption.go:700-707
  {
    bool nb = false;
    for (int u = lane; u < ns; u += WAVE)
      if (w.s_need[u] && S.nominal[ix(S, w.cq, w.s_fr[u])] <= s.usage[ix(S, w.cq, w.s_fr[u])]) nb = true;
    under_nominal = wballot(nb) == 0;
  }
  int nattempt; bool attempts[2];
  if (no_other || (forbidden && !under_nominal)) { nattempt = 1; attempts[0] = true; attempts[1] = true; }
  else if (forbidden && no_hier) { nattempt = 2; attempts[0] = false; attempts[1] = true; }
  else { nattempt = 2; attempts[0] = true; attempts[1] = false; }
  int n_inuse = 0;
  for (int u = 0; u < ns; u++) n_inuse += w.s_inu[u] ? 1 : 0;
  const int64_t fits_bytes = 40 * (int64_t)plen * n_inuse;
  for (int at = 0; at < nattempt; at++) {
    const bool borrowing = attempts[at];
    CSTAT(3, 1);
    // The walk stops at the first candidate whose removal makes the preemptor fit, and nothing behind that candidate has any influence
    // on what happens in front of it: the death points and the first fit are computed for a PREFIX of the time order (128 candidates,
    // then 4 x as many, ...) and the search stops at the first prefix that holds the stopping point. An overlap recomputation in
    // processEntry typically needs a handful of victims out of thousands of candidates (profiles/r04a_cfg4c_jacobi_probe.txt).
    int tstar = -1, nt = 0; int64_t rb_removed = 0;
    int cN = 0; int64_t cRB = 0;
    // (only where few victims are expected: a recomputation inside processEntry, s.removed set. A nomination against an over-committed
    // cycle-start snapshot removes hundreds of candidates before the first fit: growing prefixes would cost a third more than one pass.)
    const bool lazy = k.C.cs_lazy == 2 || (k.C.cs_lazy == 1 && s.removed != nullptr);
    for (int T = lazy ? (Mt < 128 ? Mt : 128) : Mt;; T = (T * 4 < Mt ? T * 4 : Mt)) {
    CSTAT(30, 1); if (T < Mt) CSTAT(31, 1);
    for (int base = 0; base < T; base += WAVE) {
      const int t = base + lane;
      if (t < T) { const int j = c.tord[t]; const uint8_t cb = c.cls[j]; c.alive[j] = (cb != 0 && !(borrowing && (cb >> 3) == V_RECLAIM_NO_BORROW)) ? 1 : 0; }
    }
    wsync();
    for (int dd = c.levels; dd >= 1; dd--) cs_level_pass(c, dd, T - 1, false);
    if (c.levels == 0) {  // ClusterQueues without a cohort: only same-queue candidates, straight onto the path
      for (int base = 0; base < M; base += WAVE) {
        const int j = base + lane;
        if (j < M) {
          const AdmRec r = S.adm_rec[rows[j]];
          #pragma unroll
          for (int u = 0; u < CS_NS; u++) if (u < ns) c.dB[u][j] = cs_rec_qty(S, r, rows[j], w.s_fr[u]);
        }
      }
      wsync();
    }
    KQ_TS(k, 37);  // scan search: alive flags + level passes of the prefix
    // ---- first fit in time order ----
    int64_t cA[CS_NS][CS_LEVELS + 1];
    #pragma unroll
    for (int u = 0; u < CS_NS; u++) {
      #pragma unroll
      for (int l = 0; l <= CS_LEVELS; l++) cA[u][l] = 0;
    }
    cN = 0; cRB = 0;
    tstar = -1; nt = 0; rb_removed = 0;
    for (int base = 0; base < T && tstar < 0; base += WAVE) {
      const int t = base + lane;
      const bool in = t < T;
      const int j = in ? c.tord[t] : 0;
      const bool al = in && c.alive[j] != 0;
      const int L = c.att[j];
      bool fit = al;
      PathU uvs[CS_NS];
      #pragma unroll
      for (int u = 0; u < CS_NS; u++) {
        if (u >= ns) continue;
        const int64_t d = al ? c.dB[u][j] : 0;
        int64_t A[CS_LEVELS + 1];
        #pragma unroll
        for (int l = 0; l <= CS_LEVELS; l++) {
          A[l] = cA[u][l];
          if (l >= plen || wballot(al && L == l) == 0) continue;
          const int64_t P = wprefix_incl_i64(L == l ? d : 0);
          A[l] = cA[u][l] + P;
          cA[u][l] = wbcast_u(A[l], WAVE - 1);
        }
        PathU uv;
        uv.v[0] = a_sub(w.cs_u0[u][0], A[0]);
        #pragma unroll
        for (int l = 1; l <= CS_LEVELS; l++) {
          if (l >= plen) { uv.v[l] = 0; continue; }
          const int64_t b0 = cs_B(w.cs_u0[u][l - 1], w.cs_lq[u][l - 1]), bt = cs_B(uv.v[l - 1], w.cs_lq[u][l - 1]);
          uv.v[l] = a_sub(a_sub(w.cs_u0[u][l], A[l]), b0 - bt);
        }
        uvs[u] = uv;
        if (w.s_inu[u] && !cs_fits_slot(w, u, plen, uv, borrowing)) fit = false;
      }
      const int na = cN + wprefix_incl_i32(al ? 1 : 0);
      const int64_t rbs = cRB + wprefix_incl_i64(al ? (int64_t)c.rb[j] : 0);
      cN = wbcast_u(na, WAVE - 1); cRB = wbcast_u(rbs, WAVE - 1);
      const uint64_t fm = wballot(fit);
      if (fm) {
        const int b = ffs64(fm);
        tstar = base + b;
        nt = wbcast_u(na, b);
        rb_removed = wbcast_u(rbs, b);
        #pragma unroll
        for (int u = 0; u < CS_NS; u++) {
          #pragma unroll
          for (int l = 0; l <= CS_LEVELS; l++) if (lane == b && u < ns && l < plen) w.cs_uf[u][l] = uvs[u].v[l];
        }
      }
    }
    wsync();
    KQ_TS(k, 38);  // scan search: first fit
    if (tstar >= 0 || T >= Mt) break;
    }  // next, longer prefix
    CSTAT(5, tstar >= 0 ? nt : cN);
    if (tstar < 0) {  // every candidate removed, never fit: restoreSnapshot adds them all back (preemption.go:333)
      if (lane == 0) w.bytes += 2 * cRB + (int64_t)cN * fits_bytes;
      continue;
    }
    if (nt > k.X.tgt_cap) { set_error(k, KQ_ECAPACITY); w.ntgt = 0; return true; }
    if (lane == 0) w.bytes += rb_removed + (int64_t)nt * fits_bytes;
    // ---- the state the walk has reached: every node's usage after the removals up to the stopping time ----
    for (int dd = c.levels; dd >= 1; dd--) cs_level_pass(c, dd, tstar, true);
    for (int i = lane; i < ns * plen; i += WAVE) { const int u = i / plen, l = i % plen; c.W[(size_t)u * nn + w.cs_pl[l]] = w.cs_uf[u][l]; }
    // ---- targets in time order ----
    {
      int run = 0;
      for (int base = 0; base <= tstar; base += WAVE) {
        const int t = base + lane;
        const bool in = t <= tstar;
        const int j = in ? c.tord[t] : 0;
        const bool al = in && c.alive[j] != 0;
        const uint64_t m = wballot(al);
        if (al) {
          const int i = run + popc64(m & ((1ull << lane) - 1));
          s.trow[i] = rows[j];
          s.treason[i] = (uint8_t)variant_reason(c.cls[j] >> 3);
        }
        run += popc64(m);
      }
    }
    wsync();
    KQ_TS(k, 39);  // scan search: finalising passes + target list
    // ---- fillBackWorkloads (preemption.go:341-354): probes newest-first, 64 at a time ----
    uint8_t* keep = c.alive;  // per target: still removed
    for (int i = lane; i < nt; i += WAVE) keep[i] = 1;
    int64_t fb_bytes = 0;   // lane-0 meaningful
    wsync();
    for (int hi = nt - 2; hi >= 0; hi -= WAVE) {
      const int i = hi - lane;
      const bool in = i >= 0;
      const int row = in ? s.trow[i] : s.trow[0];
      const AdmRec r = S.adm_rec[row];
      const int cq = r.cq;
      const int cplen = S.plen[cq];
      int chain[CS_LEVELS + 1];
      int64_t clq[CS_NS][CS_LEVELS + 1];
      int hc = cplen, L = 0;   // chain level that is on the preemptor's path, and which path level that is
      #pragma unroll
      for (int h = 0; h < CS_LEVELS + 1; h++) {
        const int n = S.path[(size_t)cq * KQ_MAXD + (h < cplen ? h : 0)];
        chain[h] = S.node_local[n];
        #pragma unroll
        for (int u = 0; u < CS_NS; u++) {
          if (u >= ns) { clq[u][h] = 0; continue; }
          const size_t o = ix(S, n, w.s_fr[u]);
          const int64_t sqv = S.sq[o], llv = S.ll[o];
          clq[u][h] = llv != KQ_NIL_LIMIT ? i64max(0, a_sub(sqv, llv)) : 0;
        }
        if (h < cplen && h < hc) { const int pl = cs_path_level(w, plen, chain[h]); if (pl >= 0) { hc = h; L = pl; } }
      }
      uint64_t pending = wballot(in);
      int64_t nu[CS_NS][CS_LEVELS + 1];
      int nchg[CS_NS];
      while (pending) {
        bool fit = in && ((pending >> lane) & 1);
        if (fit) {
          #pragma unroll
          for (int u = 0; u < CS_NS; u++) {
            if (u >= ns) { nchg[u] = 0; continue; }
            // addUsage of the row (resource_node.go:144-152) on the private state, kept in registers
            int64_t v = cs_rec_qty(S, r, row, w.s_fr[u]);
            int nc = 0;
            bool go = true;
            #pragma unroll
            for (int h = 0; h <= CS_LEVELS; h++) {
              nu[u][h] = 0;
              if (!go || h >= cplen) continue;
              const int64_t un = c.W[(size_t)u * nn + chain[h]];
              const int64_t la = i64max(0, a_sub(clq[u][h], un));
              nu[u][h] = a_add(un, v);
              nc = h + 1;
              if (h + 1 < cplen && v > la) v = a_sub(v, la); else go = false;
            }
            nchg[u] = nc;
            if (!w.s_inu[u]) continue;
            PathU uv;
            #pragma unroll
            for (int l = 0; l <= CS_LEVELS; l++) {
              uv.v[l] = 0;
              if (l >= plen) continue;
              const int h = hc + (l - L);
              int64_t x = c.W[(size_t)u * nn + w.cs_pl[l]];
              #pragma unroll
              for (int hh = 0; hh <= CS_LEVELS; hh++) if (l >= L && hh == h && hh < nc) x = nu[u][hh];
              uv.v[l] = x;
            }
            if (!cs_fits_slot(w, u, plen, uv, borrowing)) fit = false;
          }
        }
        const uint64_t fm = wballot(fit) & pending;
        if (!fm) break;
        const int b = ffs64(fm);   // lane 0 holds the newest target: the first probe that fits is the one the walk accepts next
        #pragma unroll
        for (int u = 0; u < CS_NS; u++) {
          #pragma unroll
          for (int h = 0; h <= CS_LEVELS; h++) if (lane == b && u < ns && h < nchg[u]) c.W[(size_t)u * nn + chain[h]] = nu[u][h];
        }
        if (lane == b) keep[i] = 0;
        wsync();
        pending &= b == 63 ? 0ull : ~((2ull << b) - 1);
      }
      // bytes: every probe adds the row and tests the fit; a rejected one is removed again
      const int64_t rbv = in ? (int64_t)r.rowbytes : 0;
      const int64_t tot = wsum_i64(in ? (rbv + fits_bytes + (keep[i] ? rbv : 0)) : 0);
      fb_bytes += tot;
    }
    wsync();
    // ---- compact the kept targets; the reference re-adds them (restoreSnapshot :356): those writes are traffic too ----
    {
      int run = 0;
      int64_t restore = 0;
      for (int base = 0; base < nt; base += WAVE) {
        const int i = base + lane;
        const bool kp = i < nt && keep[i] != 0;
        const int row = i < nt ? s.trow[i] : 0;
        const uint8_t rs = i < nt ? s.treason[i] : 0;
        const uint64_t m = wballot(kp);
        restore += wsum_i64(kp ? (int64_t)S.adm_rec[row].rowbytes : 0);
        wsync();
        if (kp) { const int o = run + popc64(m & ((1ull << lane) - 1)); s.trow[o] = row; s.treason[o] = rs; }
        run += popc64(m);
        wsync();
      }
      w.ntgt = run;
      if (lane == 0) w.bytes += fb_bytes + restore;
    }
    CSTAT(6, 1); CSTAT(7, w.ntgt);
    KQ_TS(k, 63);  // scan search: fill-back + compaction of the targets
    // the private state with exactly the targets removed, for the caller (find_height reads the preemptor's path)
    for (int i = lane; i < ns * plen; i += WAVE) {
      const int u = i / plen, l = i % plen;
      s.W[(size_t)w.cs_pl[l] * ns + u] = c.W[(size_t)u * nn + w.cs_pl[l]];
    }
    wsync();
    return true;
  }
  w.ntgt = 0;
  return true;
}

}  // namespace kq

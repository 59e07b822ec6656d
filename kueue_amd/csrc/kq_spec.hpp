// kq_spec.hpp — processEntry (scheduler.go:392-523) for the PLAIN entries of a root-cohort tree as speculative parallel rounds.
//
// The reference walks the ordered entries one by one: entry i is admitted iff it fits the usage its admitted predecessors left
// (scheduler.go:356-375, :771-777; cq.AddUsage :486). With x_i in {0,1} the admission of entry i that is the recurrence
//     x_i = fits(i, base + sum_{j<i} x_j * contribution_j)
// which has exactly one solution (induction over i). Two facts make it parallel:
//   * Available (resource_node.go:106-122) never grows when usage grows, and what addUsage (:144-152) passes up a level never
//     shrinks when the usage below grows. So with a set A of entries known to be admitted and a set R known to be rejected,
//     evaluating every undecided entry against "A and every undecided predecessor admitted" (the OVER world) under-estimates its
//     fit, and against "only A admitted" (the UNDER world) over-estimates it: fits in OVER => admitted, does not fit in UNDER =>
//     rejected. The first undecided entry in order sees the same usage in both worlds, so every round decides at least one
//     entry per connected group of flavor-resource columns; in practice a cfg 3 cycle (1000 heads, root row binding) is done in
//     4-8 rounds.
//   * the usage a cell (cohort node, flavor-resource) holds when entry i takes its turn is base + the sum of what the admitted
//     predecessors under that node pushed into it — a SEGMENTED PREFIX SUM over the entries in iterator order, one segment per
//     cell. Cells of different depth depend on each other only bottom-up (the amount entry j pushes into its k-th ancestor depends
//     on the usage of the levels below at j's turn), so one round is: for every cohort depth, deepest first, one segmented scan
//     over the (entry, flavor-resource) items arranged by (cell, iterator position).
// One 512-thread workgroup per tree; lanes = items (8 per thread); arrangements are built once per window with an LDS radix
// sort (stable, 4-bit digits, packed 16-bit counters scanned on the DPP network); both worlds travel through the scans together.
//
// Closed forms (plain operands: everything < 2^50, checked; otherwise the entry is left to the serial kernel). With u_k the usage
// of path level k at the entry's turn (0 = ClusterQueue), t_k = max(0, localQuota_k - u_k), E_j = t_0 + .. + t_{j-1}:
//     Available(cq) = min_j (E_j + c_j - u_j)          c_j = sq_j + bl_j below the root (no limit: a large sentinel), sq_j at the root
//     addUsage: level j receives max(0, val - E_j)     (level 0: val)
// (derivation in kq_device.hpp, core_run_quad). Per item and cohort depth d the window keeps K_d = c_d - base_d - val and
// T_d = lq_d - base_d; with P the exclusive prefix of the cell: fits at d  <=>  P <= E + K_d ; t_d = max(0, T_d - P).
//
// Entries the rounds do not take: preemption targets, more than FU flavor-resources / FD path levels, a second head of the same
// ClusterQueue in the batch, operands that are not small, negative reservations (scheduler.go:806). The rounds stop in front of
// the first such entry of the tree and K::spec_resume tells the serial kernel (process_tree) where to take over — also when the
// rounds fail to converge within SP_PMAX passes (everything in front of the first undecided entry is final by then).
#pragma once

namespace kq {

constexpr int SP_NT = 512;                 // threads of the workgroup
constexpr int SP_IPT = 10;                 // items per thread
constexpr int SP_MAXI = SP_NT * SP_IPT;    // items of a window
constexpr int SP_MAXE = 1024;              // entries of a window
constexpr int SP_MAXS = FD - 1;            // cohort depths on the fast path
constexpr int SP_NW = SP_NT / 64;          // waves
constexpr int SP_PMAX = 40;                // rounds per window before the undecided tail goes back to the serial kernel
constexpr int SP_VPAD = SP_MAXI + SP_NT;   // scan arrays are indexed p + p / SP_IPT: a thread's consecutive int64 start SP_IPT + 1 (odd) elements apart
constexpr int SP_SLOTS = 256;              // workgroups of a launch (each loops over trees): one region of K::spec_kt each
static_assert(SP_IPT % 2 == 0 && SP_IPT <= 16, "odd stride of the padded scan arrays; head flags of a thread are one 16-bit mask");

enum { SPC_DONE = 0, SPC_FIT = 1, SPC_FORCED = 2 };                 // entry class: no items / speculative / unconditional AddUsage (reservation)
enum { SPS_UNKNOWN = 0, SPS_ADMIT = 1, SPS_REJECT = 2, SPS_DROP = 3 };  // DROP: behind the truncation point, handed back

#ifdef KQ_HOST_EMU
static int g_spec_off = 0;       // tests: 1 = every tree goes to the serial kernel
static int g_spec_maxe = SP_MAXE, g_spec_maxi = SP_MAXI, g_spec_pmax = SP_PMAX;  // tests: small windows / early truncation
#endif

struct SpecLds {
  // uniform control words (read by everybody after a barrier)
  int32_t cursor, win_start, n_ent, n_items, cut, cut_ent, cut_items, closed, stop, abort_, n_unknown, first_unknown, final_, trunc, resume;
  int32_t n_act[SP_MAXS];
  int32_t needK[SP_MAXS], needT[SP_MAXS];  // some item's term of Available can bind at this depth / some item has local quota left there
  int32_t w_a[SP_NW], w_b[SP_NW], tot_a, tot_b;
  int64_t w_L[SP_NW], w_U[SP_NW];
  int32_t w_f[SP_NW];
  uint64_t w_c[SP_NW][4];
  int32_t dbase[16];
  int64_t bytes;
  int32_t s_a[SP_NT], s_b[SP_NT];
  int32_t ent[SP_MAXE], pos[SP_MAXE];
  uint16_t item0[SP_MAXE + 2];
  uint8_t st[SP_MAXE], cls[SP_MAXE], okL[SP_MAXE], okU[SP_MAXE];
  uint16_t desc[SP_MAXI];                  // entry << 3 | slot
  uint16_t hfm[SP_MAXS][SP_NT + 1];        // heads of the cells' segments: bit r of [d][t] = position t * SP_IPT + r starts a cell
  union {
    struct { int64_t vL[SP_VPAD], vU[SP_VPAD]; } v;
    struct { uint16_t ka[SP_MAXI], kb[SP_MAXI], pa[SP_MAXI], pb[SP_MAXI], pF[SP_MAXI], kd[SP_MAXI], ipos[SP_MAXI]; int32_t kc[SP_MAXI]; } s;
  };
};

// What a thread keeps of its items in registers: the request and the packed positions / ids. The per-depth constants K_d / T_d and
// E_0 live in the workgroup's region of K::spec_kt (L2-resident) and are loaded where a round needs them: with everything in registers
// the kernel needs 23 registers per item — three quarters of the CU's register file for a 4096-item window.
struct alignas(16) SpecItem {      // as staged in K::spec_kt (32 bytes)
  int64_t val, E0;                 // request (or reservation); t_0 = max(0, localQuota_0 - usage_0)
  uint32_t p01, p2e;               // positions in the arrangements: depth 0 | depth 1 << 16; depth 2 | (window entry | slot << 10 | act << 13) << 16
  uint32_t pad[2];
};
static_assert(sizeof(SpecItem) == 32, "staged in K::spec_kt as 4 words");
struct SpecReg { int64_t val; uint32_t p01, p2e; };
struct SpecThread { SpecReg it[SP_IPT]; int64_t eL[SP_IPT], eU[SP_IPT]; };  // eL / eU: E of the OVER / UNDER world while a round climbs the depths
KQ_DEV int sp_pos(const SpecReg& x, int d) { return d == 0 ? (int)(x.p01 & 0xffffu) : (d == 1 ? (int)(x.p01 >> 16) : (int)(x.p2e & 0xffffu)); }
KQ_DEV void sp_set_pos(SpecReg& x, int d, int p) {
  if (d == 0) x.p01 = (x.p01 & 0xffff0000u) | (uint32_t)p; else if (d == 1) x.p01 = (x.p01 & 0xffffu) | ((uint32_t)p << 16); else x.p2e = (x.p2e & 0xffff0000u) | (uint32_t)p;
}
KQ_DEV int sp_ent(const SpecReg& x) { return (int)((x.p2e >> 16) & 0x3ffu); }
KQ_DEV int sp_slot(const SpecReg& x) { return (int)((x.p2e >> 26) & 7u); }
KQ_DEV int sp_act(const SpecReg& x) { return (int)(x.p2e >> 29); }
static_assert(SP_MAXS <= 3 && SP_MAXE <= 1024 && FU <= 8, "SpecItem packs three positions, the window entry, the slot and three act bits");
// K_d / T_d of item q in the workgroup's region of K::spec_kt
KQ_DEV size_t sp_kt(int d, int which, int q) { return ((size_t)(d * 2 + which)) * SP_MAXI + q; }
constexpr size_t SP_KT_CONST = (size_t)SP_MAXS * 2 * SP_MAXI;          // K_d / T_d
constexpr size_t SP_KT_WORDS = SP_KT_CONST + (size_t)SP_MAXI * 4;     // + the items themselves (32 bytes each)

KQ_DEV int sp_vidx(int p) { return p + p / SP_IPT; }
KQ_DEV int sp_bits(int n) { int b = 0; while ((1 << b) < n) b++; return b; }  // ceil(log2(n)), 0 for n <= 1

// ---- workgroup primitives. Device: all SP_NT threads call them (they contain barriers and end with one). Emulation: one call does
// ---- the whole array serially (same LDS layout, so the phases around them are shared).
#ifndef KQ_HOST_EMU
// exclusive prefix sums of L.s_a / L.s_b over the threads, totals in L.tot_a / L.tot_b
KQ_DEV void sp_scan2(SpecLds& L, int tid) {
  const int a = L.s_a[tid], b = L.s_b[tid];
  const int ia = wprefix_incl_i32(a), ib = wprefix_incl_i32(b);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 63) { L.w_a[wave] = ia; L.w_b[wave] = ib; }
  __syncthreads();
  int ca = 0, cb = 0;
  for (int w2 = 0; w2 < wave; w2++) { ca += L.w_a[w2]; cb += L.w_b[w2]; }
  L.s_a[tid] = ca + ia - a; L.s_b[tid] = cb + ib - b;
  if (tid == SP_NT - 1) { L.tot_a = ca + ia; L.tot_b = cb + ib; }
  __syncthreads();
}
KQ_DEV uint32_t sp_field(const uint64_t* c, int d) { return (uint32_t)((c[d >> 2] >> ((d & 3) * 16)) & 0xffffu); }
// one stable radix pass (4-bit digit at `shift`) of (key, payload) over the first n positions: src -> dst
KQ_DEV void sp_radix_pass(SpecLds& L, const uint16_t* sk, const uint16_t* sp, uint16_t* dk, uint16_t* dp, int n, int shift, int tid) {
  const int base = tid * SP_IPT;
  uint16_t key[SP_IPT], pay[SP_IPT];
  uint64_t lc[4] = {0, 0, 0, 0};
  uint32_t lr[SP_IPT];
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int p = base + r;
    key[r] = p < n ? sk[p] : (uint16_t)0xffff; pay[r] = p < n ? sp[p] : (uint16_t)0;
    const int d = (key[r] >> shift) & 15;
    const uint64_t inc = 1ull << ((d & 3) * 16);
    const int q = d >> 2;
    const uint64_t cur = q == 0 ? lc[0] : (q == 1 ? lc[1] : (q == 2 ? lc[2] : lc[3]));
    lr[r] = (uint32_t)((cur >> ((d & 3) * 16)) & 0xffffu);
    lc[0] += q == 0 ? inc : 0; lc[1] += q == 1 ? inc : 0; lc[2] += q == 2 ? inc : 0; lc[3] += q == 3 ? inc : 0;
  }
  uint64_t in[4];
  #pragma unroll
  for (int q = 0; q < 4; q++) in[q] = (uint64_t)wprefix_incl_i64((int64_t)lc[q]);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 63) { L.w_c[wave][0] = in[0]; L.w_c[wave][1] = in[1]; L.w_c[wave][2] = in[2]; L.w_c[wave][3] = in[3]; }
  __syncthreads();
  if (wave == 0) {  // digit bases: total per digit over the waves, exclusive prefix over the 16 digits
    int t = 0;
    if (lane < 16) for (int w2 = 0; w2 < SP_NW; w2++) t += (int)((L.w_c[w2][lane >> 2] >> ((lane & 3) * 16)) & 0xffffu);
    const int inc = wprefix_incl_i32(t);
    if (lane < 16) L.dbase[lane] = inc - t;
  }
  uint64_t ex[4];
  #pragma unroll
  for (int q = 0; q < 4; q++) { ex[q] = in[q] - lc[q]; for (int w2 = 0; w2 < wave; w2++) ex[q] += L.w_c[w2][q]; }
  __syncthreads();
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int p = base + r;
    if (p >= n) continue;
    const int d = (key[r] >> shift) & 15;
    const int dst = L.dbase[d] + (int)sp_field(ex, d) + (int)lr[r];
    dk[dst] = key[r]; dp[dst] = pay[r];
  }
  __syncthreads();
}
// segmented EXCLUSIVE prefix sums of (vL, vU) over positions [0, n) with the head flags of depth d, in place. Two sweeps over the
// thread's positions (aggregate, then write-back) instead of keeping 4 x SP_IPT values in registers next to the caller's item state.
KQ_DEV void sp_segscan(SpecLds& L, int d, int n, int tid) {
  const int base = tid * SP_IPT, vb = tid * (SP_IPT + 1);
  const uint32_t fb = base < n ? L.hfm[d][tid] : 0u;
  int64_t accL = 0, accU = 0;
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const bool in = base + r < n;
    if ((fb >> r) & 1u) { accL = 0; accU = 0; }
    accL += in ? L.v.vL[vb + r] : 0; accU += in ? L.v.vU[vb + r] : 0;
  }
  // inclusive segmented scan of the thread aggregates (accL, accU, any head) over the wave: (f1,v1) o (f2,v2) = (f1|f2, f2 ? v2 : v1+v2)
  uint64_t vl = (uint64_t)accL, vu = (uint64_t)accU;
  int f = fb != 0 ? 1 : 0;
#define SP_STEP(CTRL, RM) { const uint64_t sl = dpp0_u64<CTRL, RM>(vl), su = dpp0_u64<CTRL, RM>(vu); \
    const int sf = __builtin_amdgcn_update_dpp(0, f, CTRL, RM, 0xf, false); if (!f) { vl += sl; vu += su; } f |= sf; }
  SP_STEP(0x111, 0xf) SP_STEP(0x112, 0xf) SP_STEP(0x114, 0xf) SP_STEP(0x118, 0xf) SP_STEP(0x142, 0xa) SP_STEP(0x143, 0xc)
#undef SP_STEP
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 63) { L.w_L[wave] = (int64_t)vl; L.w_U[wave] = (int64_t)vu; L.w_f[wave] = f; }
  // what the lanes before this one carry in: the inclusive value of lane - 1 (wave_shr:1), nothing for lane 0
  uint64_t cl = dpp_u64<0x138>(vl), cu = dpp_u64<0x138>(vu);
  int cf = __builtin_amdgcn_update_dpp(f, f, 0x138, 0xf, 0xf, false);
  if (lane == 0) { cl = 0; cu = 0; cf = 0; }
  __syncthreads();
  if (!cf) {  // no head in the earlier lanes of this wave: the earlier waves' open segment continues here
    for (int w2 = wave - 1; w2 >= 0; w2--) { cl += (uint64_t)L.w_L[w2]; cu += (uint64_t)L.w_U[w2]; if (L.w_f[w2]) break; }
  }
  int64_t runL = (int64_t)cl, runU = (int64_t)cu;  // running sum since the last head (carry included until the thread's first head)
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    if (base + r >= n) continue;
    if ((fb >> r) & 1u) { runL = 0; runU = 0; }
    const int64_t xl = L.v.vL[vb + r], xu = L.v.vU[vb + r];
    L.v.vL[vb + r] = runL; L.v.vU[vb + r] = runU;
    runL += xl; runU += xu;
  }
  __syncthreads();
}
#else
static void sp_scan2(SpecLds& L, int) {
  int a = 0, b = 0;
  for (int t = 0; t < SP_NT; t++) { const int xa = L.s_a[t], xb = L.s_b[t]; L.s_a[t] = a; L.s_b[t] = b; a += xa; b += xb; }
  L.tot_a = a; L.tot_b = b;
}
static void sp_radix_pass(SpecLds&, const uint16_t* sk, const uint16_t* sp, uint16_t* dk, uint16_t* dp, int n, int shift, int) {
  int cnt[17] = {0};
  for (int p = 0; p < n; p++) cnt[((sk[p] >> shift) & 15) + 1]++;
  for (int d = 0; d < 16; d++) cnt[d + 1] += cnt[d];
  for (int p = 0; p < n; p++) { const int d = (sk[p] >> shift) & 15; dk[cnt[d]] = sk[p]; dp[cnt[d]] = sp[p]; cnt[d]++; }
}
static void sp_segscan(SpecLds& L, int d, int n, int) {
  int64_t al = 0, au = 0;
  for (int p = 0; p < n; p++) {
    if ((L.hfm[d][p / SP_IPT] >> (p % SP_IPT)) & 1) { al = 0; au = 0; }
    const int64_t xl = L.v.vL[sp_vidx(p)], xu = L.v.vU[sp_vidx(p)];
    L.v.vL[sp_vidx(p)] = al; L.v.vU[sp_vidx(p)] = au;
    al += xl; au += xu;
  }
}
#endif

// everything uniform for the tree
struct SpecCtx {
  int tree, n, nfr, D;          // D: cohort depths handled (min(tree depth, SP_MAXS))
  int maxe, maxi, pmax;
  int dbits[SP_MAXS], dcnt[SP_MAXS], frbits;
  bool prio_preemptors;
};

// quotaResourcesToReserve (scheduler.go:796-814) of slot u of a Preempt-mode entry without targets; *odd: an operand the plain
// arithmetic below does not cover
KQ_DEV int64_t sp_reserve(const K& k, const PRec& r, int u, int nfr, bool* odd) {
  const int64_t qty = r.qty[u], uw0 = r.uw0[u], nominal = r.nominal[u], bl0 = k.S.bl[(size_t)r.cq * nfr + r.fr[u]];
  if (r.borrowing > 0) {
    if (bl0 == KQ_NIL_LIMIT || bl0 == I64MAX || nominal == I64MAX) return qty;
    if ((uint64_t)nominal >= (uint64_t)PLAIN_LIMIT || (uint64_t)bl0 >= (uint64_t)PLAIN_LIMIT) { *odd = true; return 0; }
    return i64min(qty, (nominal + bl0) - uw0);
  }
  if (nominal == I64MAX) return i64max(0, qty);
  if ((uint64_t)nominal >= (uint64_t)PLAIN_LIMIT) { *odd = true; return 0; }
  return i64max(0, i64min(qty, nominal - uw0));
}

// ---- phases (one call per thread; a barrier follows each) -----------------------------------------------------------------------
// Window building, one chunk of SP_NT iterator positions: which of them are this tree's, what they need.
// s_a = 1 for an entry of the tree, s_b = its items.  Entry classes and the reasons to stop in front of an entry ("bad").
KQ_DEV void sp_chunk_classify(const K& k, const SpecCtx& c, SpecLds& L, int tid, int* o_e, int* o_cls, int* o_bad) {
  const int p = L.cursor + tid;
  int mine = 0, items = 0, e = -1, cls = SPC_DONE, bad = 0;
  if (p < c.n) {
    e = k.order_idx[p];
    const int cq = k.H.cq[e];
    if (k.S.tree_of[cq] == c.tree) {
      mine = 1;
      const PRec& r = k.grec[e];
      const int mode = r.mode, nuse = r.nuse;
      uint32_t cb = 0;  // cbig of the cells the entry touches (cells beyond nuse / plen keep stale bytes: mask by the loop bounds)
      if (!r.slow_static) for (int u = 0; u < nuse; u++) for (int i = 0; i < r.plen; i++) cb |= r.cbig[u][i];
      if (r.slow_static || k.cq_heads[cq] > 1 || (cb & 2)) bad = 1;
      else if (mode == M_NOFIT || nuse == 0) cls = SPC_DONE;
      else if (mode == M_FIT) { cls = SPC_FIT; items = nuse; }
      else if (mode == M_PREEMPT) {  // no targets: reserveCapacityForUnreclaimablePreempt scheduler.go:538-543
        const bool can_always_reclaim = KQ_POL_RECLAIM(r.pol) == KQ_POLICY_ANY;
        if (!can_always_reclaim || (c.prio_preemptors && (r.flags & KQ_HEAD_IS_PREEMPTOR))) {
          cls = SPC_FORCED; items = nuse;
          for (int u = 0; u < nuse; u++) {
            bool odd = false;
            const int64_t v = sp_reserve(k, r, u, c.nfr, &odd);
            // a NEGATIVE reservation (scheduler.go:806 has no max(0, .)) only lowers the ClusterQueue's own cell — nothing is passed up —
            // but it voids the incremental usage_np of the serial kernel once rows are preempted in the tree: leave it to that kernel
            // whenever preemption is possible at all
            if (odd || (v < 0 && k.C.any_preempt)) bad = 1;
          }
        }
      } else bad = 1;
    }
  }
  L.s_a[tid] = mine; L.s_b[tid] = items;
  *o_e = e; *o_cls = cls; *o_bad = bad;
}
// after the scan: window slot / first item of every entry of the chunk; the window closes in front of the first entry that is bad
// or does not fit any more
KQ_DEV void sp_chunk_place(const SpecCtx& c, SpecLds& L, int tid, int mine, int items, int bad, int* o_j, int* o_i0) {
  const int j = L.n_ent + L.s_a[tid], i0 = L.n_items + L.s_b[tid];
  *o_j = j; *o_i0 = i0;
  if (mine && (bad || j >= c.maxe || i0 + items > c.maxi)) atomic_min_i32(&L.cut, L.cursor + tid);
}
KQ_DEV void sp_chunk_commit(const SpecCtx& c, SpecLds& L, int tid, int mine, int e, int cls, int bad, int items, int j, int i0) {
  const int p = L.cursor + tid;
  if (mine && p < L.cut) { L.ent[j] = e; L.pos[j] = p; L.cls[j] = (uint8_t)cls; L.item0[j] = (uint16_t)i0; }
  if (mine && p == L.cut) { L.cut_ent = j; L.cut_items = i0; L.stop = bad ? 1 : 0; }
  (void)items; (void)c;
}

// one item's constants (and the static level-0 test), staged in the workgroup's region of K::spec_kt. Returns false when an operand is
// outside what the classification let through (cannot happen): the window is abandoned before anything was written.
KQ_DEV bool sp_item_load(const K& k, const SpecCtx& c, SpecLds& L, int64_t* kt, int q) {
  const DSnap& S = k.S;
  const int dsc = L.desc[q], j = dsc >> 3, u = dsc & 7;
  const PRec& r = k.grec[L.ent[j]];
  const int plen = r.plen, fr = r.fr[u], cq = r.cq;
  SpecItem x;
  x.p01 = 0; x.pad[0] = x.pad[1] = 0;
  int act = 0;
  bool ok = true;
  const int64_t uw0 = r.uw0[u], qty = r.qty[u];
  // Unlimited constants (resources.Amount: absorbing). With finite usage, a level whose SubtreeQuota, localQuota or borrowing limit is
  // Unlimited never binds (its term of Available is Unlimited), and an Unlimited localQuota keeps everything local: LocalAvailable is
  // Unlimited, nothing is passed up (resource_node.go:92-152). Both are the closed form with a large constant.
  constexpr int64_t C_NOLIMIT = (int64_t)1 << 60, T_INF = (int64_t)1 << 59;
  int64_t val = qty;
  if (L.cls[j] == SPC_FORCED) {
    bool odd = false;
    val = sp_reserve(k, r, u, c.nfr, &odd);
    if (odd) ok = false;                                                // (classified as bad already: cannot happen)
    if (val < 0 && k.cert_flags) k.cert_flags[c.tree] = 1;              // outside the sharding certificate, as in the serial core
  }
  x.val = val;
  {
    const int64_t lq0 = r.lq[u][0];
    x.E0 = lq0 == I64MAX ? T_INF : i64max(0, lq0 - uw0);
    int64_t c0 = r.ccv[u][0];
    if (r.cbig[u][0] & 1) {
      const int64_t sq0 = r.sqv[u][0], bl0 = S.bl[(size_t)cq * c.nfr + fr];
      const bool root = plen == 1;
      c0 = (sq0 == I64MAX || lq0 == I64MAX || (!root && (bl0 == KQ_NIL_LIMIT || bl0 == I64MAX))) ? C_NOLIMIT : (root ? sq0 : sq0 + bl0);
    }
    if (L.cls[j] == SPC_FIT && qty > 0 && c0 - uw0 < qty) L.st[j] = SPS_REJECT;  // level-0 term of Available (static: one head per ClusterQueue)
  }
  #pragma unroll
  for (int d = 0; d < SP_MAXS; d++) {
    const int i = plen - 1 - d;
    if (d >= c.D || i < 1) continue;
    const int node = S.path[(size_t)cq * KQ_MAXD + i];
    const size_t o = (size_t)node * c.nfr + fr;
    const int64_t base = k.usage_work[o];
    act |= 1 << d;
    int64_t cc = r.ccv[u][i];
    const int64_t lqv = r.lq[u][i];
    if (r.cbig[u][i] & 1) {
      const int64_t sqv = r.sqv[u][i], blv = S.bl[o];
      const bool root = i == plen - 1;
      cc = (sqv == I64MAX || lqv == I64MAX || (!root && (blv == KQ_NIL_LIMIT || blv == I64MAX))) ? C_NOLIMIT : (root ? sqv : sqv + blv);
    }
    const int64_t Tv = lqv == I64MAX ? T_INF : lqv - base;
    kt[sp_kt(d, 0, q)] = cc - base - val; kt[sp_kt(d, 1, q)] = Tv;
    // the prefix P of a cell is >= 0: a level's term can only bind with a finite limit (and only undecided entries are tested); local
    // quota is only left while T > 0. A depth where neither holds for any item needs no scan before the last round.
    if (cc != C_NOLIMIT && cc != QC_NOLIMIT && L.cls[j] == SPC_FIT) L.needK[d] = 1;
    if (Tv > 0) L.needT[d] = 1;
  }
  x.p2e = ((uint32_t)j | (uint32_t)u << 10 | (uint32_t)act << 13) << 16;
  ((SpecItem*)(kt + SP_KT_CONST))[q] = x;
  return ok;
}

// ---- one depth of a round, around the scan -----------------------------------------------------------------------------------------
// before the scan: what every item pushes into its cell of depth d in the two worlds (addUsage: max(0, val - E))
KQ_DEV void sp_stage_push(SpecLds& L, const SpecThread& ts, int d, int n_items, int tid) {
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int q = tid * SP_IPT + r;
    const SpecReg& x = ts.it[r];
    if (q < n_items && (sp_act(x) >> d & 1)) {
      const int st = L.st[sp_ent(x)];
      const int vp = sp_vidx(sp_pos(x, d));
      L.v.vL[vp] = (st == SPS_UNKNOWN || st == SPS_ADMIT) ? i64max(0, x.val - ts.eL[r]) : 0;
      L.v.vU[vp] = st == SPS_ADMIT ? i64max(0, x.val - ts.eU[r]) : 0;
    }
  }
}
// after the scan: the level's term of Available for undecided entries (NEEDK), the local quota left (NEEDT); last round (FIN): the
// cells' usage and the certificate's slack
template <bool NEEDK, bool NEEDT, bool FIN>
KQ_DEV void sp_stage_pull(const K& k, const SpecCtx& c, SpecLds& L, const int64_t* kt, SpecThread& ts, int d, int na, int n_items, int tid) {
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int q = tid * SP_IPT + r;
    const SpecReg& x = ts.it[r];
    if (q >= n_items || !(sp_act(x) >> d & 1)) continue;
    const int p = sp_pos(x, d), en = sp_ent(x);
    const int64_t PL = L.v.vL[sp_vidx(p)], PU = L.v.vU[sp_vidx(p)];
    const int st = L.st[en];
    if (NEEDK || FIN) {
      const int64_t Kd = kt[sp_kt(d, 0, q)];
      if (NEEDK && st == SPS_UNKNOWN && x.val > 0) {
        if (PL > ts.eL[r] + Kd) L.okL[en] = 0;
        if (PU > ts.eU[r] + Kd) L.okU[en] = 0;
      }
      if (FIN) {
        const PRec& rec = k.grec[L.ent[en]];
        const int i = rec.plen - 1 - d, fr = rec.fr[sp_slot(x)];
        if (st == SPS_ADMIT && d == 0 && L.cls[en] == SPC_FIT && k.root_margin)  // sharding certificate: slack of the root term (K::root_margin)
          cert_min(k.root_margin + (size_t)c.tree * c.nfr + fr, (long long)(ts.eL[r] + Kd - PL));
        if (p == na - 1 || ((L.hfm[d][(p + 1) / SP_IPT] >> ((p + 1) % SP_IPT)) & 1)) {  // last item of the cell: the cell's usage after the window
          const int64_t add = PL + (st == SPS_ADMIT ? i64max(0, x.val - ts.eL[r]) : 0);
          if (add != 0) {
            const size_t o = (size_t)k.S.path[(size_t)rec.cq * KQ_MAXD + i] * c.nfr + fr;
            k.usage_work[o] += add; k.usage_np[o] += add;
          }
        }
      }
    }
    if (NEEDT) {
      const int64_t Td = kt[sp_kt(d, 1, q)];
      ts.eL[r] += i64max(0, Td - PL);
      ts.eU[r] += i64max(0, Td - PU);
    }
  }
}
KQ_DEV void sp_stage_pull_any(const K& k, const SpecCtx& c, SpecLds& L, const int64_t* kt, SpecThread& ts, int d, int na, int n_items, int tid, bool needk, bool needt, bool fin) {
  if (fin) { if (needt) sp_stage_pull<true, true, true>(k, c, L, kt, ts, d, na, n_items, tid); else sp_stage_pull<true, false, true>(k, c, L, kt, ts, d, na, n_items, tid); }
  else if (needk && needt) sp_stage_pull<true, true, false>(k, c, L, kt, ts, d, na, n_items, tid);
  else if (needk) sp_stage_pull<true, false, false>(k, c, L, kt, ts, d, na, n_items, tid);
  else sp_stage_pull<false, true, false>(k, c, L, kt, ts, d, na, n_items, tid);
}

// results of an entry (scheduler.go:392-523 for entries without targets) and its algorithmic bytes
KQ_DEV int64_t sp_entry_result(const K& k, SpecLds& L, int j) {
  const int e = L.ent[j];
  const PRec& r = k.grec[e];
  const int mode = r.mode, cls = L.cls[j], st = L.st[j];
  int status = KQ_ST_NOT_NOMINATED, action = KQ_ACT_NONE, rq = KQ_RQ_GENERIC, skip = KQ_SKIP_NONE;
  bool added = false;
  if (mode == M_NOFIT) rq = KQ_RQ_NOFIT;
  else if (mode == M_PREEMPT) { rq = KQ_RQ_PREEMPTION_NO_CANDIDATES; added = cls == SPC_FORCED; }
  else if (cls == SPC_DONE || st == SPS_ADMIT) { status = KQ_ST_ASSUMED; action = KQ_ACT_ADMIT; added = cls == SPC_FIT; }
  else { status = KQ_ST_SKIPPED; skip = KQ_SKIP_NO_LONGER_FITS; rq = KQ_RQ_FAILED_AFTER_NOMINATION; }  // scheduler.go:1167-1170
  const DOut& O = k.O;
  O.status[e] = (uint8_t)status; O.action[e] = (uint8_t)action; O.requeue_reason[e] = (uint8_t)rq; O.skip[e] = (uint8_t)skip; O.mode[e] = (uint8_t)mode;
  O.order[e] = L.pos[j];
  return r.nuse > 0 ? (int64_t)r.nuse * r.plen * (40 + (added ? 8 : 0)) : 0;
}

#ifdef KQ_HOST_EMU
#define SP_PHASE(...) for (int tid = 0; tid < SP_NT; tid++) { SpecThread& ts = tsv[tid]; (void)ts; __VA_ARGS__ }
#define SP_BLOCK(...) { const int tid = 0; (void)tid; __VA_ARGS__ }
#define SP_TSDECL std::vector<SpecThread> tsv(SP_NT); struct ChunkRegs { int e, cls, bad, j, i0, mine, items; }; std::vector<ChunkRegs> crv(SP_NT);
#define SP_CR crv[tid]
#else
#define SP_PHASE(...) { __VA_ARGS__ } __syncthreads();
#define SP_BLOCK(...) { __VA_ARGS__ }
#define SP_TSDECL SpecThread ts; struct ChunkRegs { int e, cls, bad, j, i0, mine, items; } cr_;
#define SP_CR cr_
#endif

// One root-cohort tree. Device: called by all SP_NT threads of the workgroup with their tid. Emulation: called once (tid ignored).
// kt: this workgroup's region of K::spec_kt (SP_KT_WORDS int64).
KQ_DEV void spec_tree(const K& k, int tree, SpecLds& L, int64_t* kt, int tid_arg) {
  const DSnap& S = k.S;
  SpecCtx c;
  c.tree = tree; c.n = k.H.n; c.nfr = S.nfr;
  c.D = S.tree_depth[tree] < SP_MAXS ? S.tree_depth[tree] : SP_MAXS;
  c.maxe = SP_MAXE; c.maxi = SP_MAXI; c.pmax = SP_PMAX;
#ifdef KQ_HOST_EMU
  c.maxe = g_spec_maxe; c.maxi = g_spec_maxi; c.pmax = g_spec_pmax;
#endif
  c.frbits = sp_bits(c.nfr);
  for (int d = 0; d < SP_MAXS; d++) { c.dcnt[d] = d < c.D ? S.tree_dcnt[(size_t)tree * KQ_MAXD + d] : 0; c.dbits[d] = sp_bits(c.dcnt[d] + 1); }
  c.prio_preemptors = gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS);
  SP_TSDECL
#ifndef KQ_HOST_EMU
  const int tid = tid_arg;
  const bool off = c.nfr > 0xfff0;   // sort keys are 16 bits wide
#else
  (void)tid_arg;
  const bool off = g_spec_off || c.nfr > 0xfff0;
#endif
  SP_PHASE( if (tid == 0) { L.cursor = 0; L.resume = -1; L.bytes = 0; L.stop = 0; } )
  if (off) { SP_PHASE( if (tid == 0) k.spec_resume[tree] = 0; ) return; }
  int64_t my_bytes = 0;  // (device: per thread; emulation: the one caller)
  const SpecItem* staged = (const SpecItem*)(kt + SP_KT_CONST);
  for (;;) {  // windows
    // ---- build the window: chunks of SP_NT iterator positions until it is full, a bad entry shows up or the order ends
    SP_PHASE( if (tid == 0) { L.n_ent = 0; L.n_items = 0; L.closed = 0; L.stop = 0; L.abort_ = 0; L.win_start = L.cursor; L.trunc = -1; }
              if (tid < SP_MAXS) { L.needK[tid] = 0; L.needT[tid] = 0; L.n_act[tid] = 0; } )
    while (!L.closed) {
      SP_PHASE( sp_chunk_classify(k, c, L, tid, &SP_CR.e, &SP_CR.cls, &SP_CR.bad); SP_CR.mine = L.s_a[tid]; SP_CR.items = L.s_b[tid];
                if (tid == 0) { L.cut = 0x7fffffff; L.cut_ent = -1; } )
      SP_BLOCK( sp_scan2(L, tid); )
      SP_PHASE( sp_chunk_place(c, L, tid, SP_CR.mine, SP_CR.items, SP_CR.bad, &SP_CR.j, &SP_CR.i0); )
      SP_PHASE( sp_chunk_commit(c, L, tid, SP_CR.mine, SP_CR.e, SP_CR.cls, SP_CR.bad, SP_CR.items, SP_CR.j, SP_CR.i0); )
      SP_PHASE( if (tid == 0) {
        if (L.cut != 0x7fffffff) { L.n_ent = L.cut_ent; L.n_items = L.cut_items; L.cursor = L.cut; L.closed = 1; }
        else { L.n_ent += L.tot_a; L.n_items += L.tot_b; L.cursor += SP_NT; if (L.cursor >= c.n) { L.cursor = c.n; L.closed = 1; } }
      } )
    }
    const int n_ent = L.n_ent, n_items = L.n_items;
    const bool last_window = L.stop || L.cursor >= c.n;
    SP_PHASE( )  // everybody has read the control words before the next window rewrites them
    if (n_ent == 0) {
      if (last_window) break;
      SP_PHASE( if (tid == 0) L.resume = L.cursor; )  // a window too small for one entry (tests only): hand the rest back
      break;
    }
    int rounds = 0;
    // ---- entries: state; items: descriptors, constants (item by item — every item is ~20 dependent loads — staged in the region)
    SP_PHASE( for (int j = tid; j < n_ent; j += SP_NT) {
      const int cls = L.cls[j];
      L.st[j] = cls == SPC_FIT ? SPS_UNKNOWN : SPS_ADMIT;
      const int i0 = L.item0[j];
      if (cls != SPC_DONE) { const int nu = k.grec[L.ent[j]].nuse; for (int u = 0; u < nu; u++) L.desc[i0 + u] = (uint16_t)(j << 3 | u); }
    } )
    SP_PHASE( _Pragma("nounroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; if (q < n_items && !sp_item_load(k, c, L, kt, q)) L.abort_ = 1; } )
    if (L.abort_) {  // nothing of this window has been written: the serial kernel takes the tree over at its first entry
      SP_PHASE( if (tid == 0) { L.resume = L.win_start; if (k.spec_stats) atomic_add_i32(&k.spec_stats[6], 1); } )
      break;
    }
    SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r;
      if (q < n_items) { ts.it[r].val = staged[q].val; ts.it[r].p01 = 0; ts.it[r].p2e = staged[q].p2e; } else { ts.it[r].val = 0; ts.it[r].p01 = 0; ts.it[r].p2e = 0; } } )
    if (n_items > 0 && c.D > 0) {
      // ---- arrangements: F = items by (flavor-resource, iterator position); per depth d: F stably sorted by the rank of the item's
      // ---- cohort of depth d (items without one last)
      uint16_t *ka = L.s.ka, *kb = L.s.kb, *pa = L.s.pa, *pb = L.s.pb;
      SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; if (q < n_items) {
        const PRec& rec = k.grec[L.ent[sp_ent(ts.it[r])]]; ka[q] = (uint16_t)rec.fr[sp_slot(ts.it[r])]; pa[q] = (uint16_t)q; } } )
      for (int sh = 0; sh < c.frbits; sh += 4) {
        SP_BLOCK( sp_radix_pass(L, ka, pa, kb, pb, n_items, sh, tid); )
        uint16_t* t1 = ka; ka = kb; kb = t1; t1 = pa; pa = pb; pb = t1;
      }
      SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r; if (p < n_items) L.s.pF[p] = pa[p]; } )
      for (int d = 0; d < c.D; d++) {
        // keys of depth d by item, then along F
        SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; if (q < n_items) {
          const SpecReg& x = ts.it[r];
          int key = c.dcnt[d];
          if (sp_act(x) >> d & 1) { const PRec& rec = k.grec[L.ent[sp_ent(x)]]; key = S.drank[S.path[(size_t)rec.cq * KQ_MAXD + (rec.plen - 1 - d)]]; }
          L.s.kd[q] = (uint16_t)key; } } )
        uint16_t *xa = L.s.ka, *xb = L.s.kb, *ya = L.s.pa, *yb = L.s.pb;
        SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r; if (p < n_items) { const int q = L.s.pF[p]; xa[p] = L.s.kd[q]; ya[p] = (uint16_t)q; } } )
        for (int sh = 0; sh < c.dbits[d]; sh += 4) {
          SP_BLOCK( sp_radix_pass(L, xa, ya, xb, yb, n_items, sh, tid); )
          uint16_t* t1 = xa; xa = xb; xb = t1; t1 = ya; ya = yb; yb = t1;
        }
        SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r; if (p < n_items) L.s.ipos[ya[p]] = (uint16_t)p; } )
        SP_PHASE( int na = 0; _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; if (q < n_items) {
          SpecReg& x = ts.it[r];
          const int p = L.s.ipos[q];
          sp_set_pos(x, d, p);
          int cell = -1;
          if (sp_act(x) >> d & 1) { const PRec& rec = k.grec[L.ent[sp_ent(x)]]; cell = rec.uoff[sp_slot(x)][rec.plen - 1 - d]; na++; }
          L.s.kc[p] = cell; } }
          if (na) atomic_add_i32(&L.n_act[d], na); )
        SP_PHASE( uint32_t m = 0; _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r;
          if (p < n_items && (p == 0 || L.s.kc[p] != L.s.kc[p - 1])) m |= 1u << r; }
          L.hfm[d][tid] = (uint16_t)m; if (tid == 0) L.hfm[d][SP_NT] = 1; )
      }
      // ---- rounds
      SP_PHASE( if (tid == 0) L.final_ = 0; )
      for (int pass = 0;; pass++) {
        const bool fin = L.final_ != 0;
        rounds++;
        SP_PHASE( for (int j = tid; j < n_ent; j += SP_NT) { L.okL[j] = 1; L.okU[j] = 1; }
                  _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; const int64_t e0 = q < n_items ? staged[q].E0 : 0; ts.eL[r] = e0; ts.eU[r] = e0; }
                  if (tid == 0) { L.n_unknown = 0; L.first_unknown = 0x7fffffff; } )
        for (int d = c.D - 1; d >= 0; d--) {
          const int na = L.n_act[d];
          const bool needk = L.needK[d] != 0, needt = L.needT[d] != 0;
          if (na == 0 || !(fin || needk || needt)) continue;  // nothing at this depth can bind or hold anything back: no scan before the last round
          SP_PHASE( sp_stage_push(L, ts, d, n_items, tid); )
          SP_BLOCK( sp_segscan(L, d, na, tid); )
          SP_PHASE( sp_stage_pull_any(k, c, L, kt, ts, d, na, n_items, tid, needk, needt, fin); )
        }
        if (fin) break;
        SP_PHASE( int nu = 0, fu = 0x7fffffff;
          for (int j = tid; j < n_ent; j += SP_NT) if (L.st[j] == SPS_UNKNOWN) {
            if (L.okL[j]) L.st[j] = SPS_ADMIT; else if (!L.okU[j]) L.st[j] = SPS_REJECT; else { nu++; if (j < fu) fu = j; }
          }
          if (nu) { atomic_add_i32(&L.n_unknown, nu); atomic_min_i32(&L.first_unknown, fu); } )
        // converged: one more round with both worlds equal gives the exact usage. No convergence within pmax rounds: everything in
        // front of the first undecided entry is final; the rest goes back to the serial kernel.
        const int nu = L.n_unknown, fu = L.first_unknown;
        const bool give_up = nu != 0 && pass + 1 >= c.pmax;
        SP_PHASE( if (give_up) for (int j = tid; j < n_ent; j += SP_NT) if (j >= fu) L.st[j] = SPS_DROP;
                  if (tid == 0 && (nu == 0 || give_up)) { L.final_ = 1; if (give_up) L.trunc = fu; } )
      }
    }
    // ---- results, ClusterQueue-level cells
    const int keep = L.trunc >= 0 ? L.trunc : n_ent;
    SP_PHASE( for (int j = tid; j < keep; j += SP_NT) {
      if (c.D == 0 && L.st[j] == SPS_UNKNOWN) L.st[j] = SPS_ADMIT;  // no cohort: the level-0 test was the whole of Available
      my_bytes += sp_entry_result(k, L, j);
    } )
    SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; if (q < n_items) {
      const SpecReg& x = ts.it[r];
      const int en = sp_ent(x);
      if (en < keep && L.st[en] == SPS_ADMIT) {
        const PRec& rec = k.grec[L.ent[en]];
        const size_t o = (size_t)rec.cq * c.nfr + rec.fr[sp_slot(x)];
        const int64_t nv = rec.uw0[sp_slot(x)] + x.val;
        k.usage_work[o] = nv; k.usage_np[o] = nv; k.cq_dirty[rec.cq] = 1;
      } } }
      if (tid == 0 && k.spec_stats) {
        atomic_add_i32(&k.spec_stats[0], 1); atomic_add_i32(&k.spec_stats[1], rounds); atomic_add_i32(&k.spec_stats[2], keep);
        atomic_add_i32(&k.spec_stats[4], n_items); atomic_max_i32(&k.spec_stats[5], rounds); if (keep < n_ent) atomic_add_i32(&k.spec_stats[7], 1);
      } )
    if (keep < n_ent) { SP_PHASE( if (tid == 0) L.resume = L.pos[keep]; ) break; }
    if (last_window) break;
  }
  // the serial kernel continues from here (an entry the rounds do not take, or nothing: position n)
#ifdef KQ_HOST_EMU
  { const int tid = 0;
#else
  {
#endif
    if (my_bytes) atomic_add_i64((long long*)&L.bytes, (long long)my_bytes);
#ifndef KQ_HOST_EMU
    __syncthreads();
#endif
    if (tid == 0) {
      const int res = L.resume >= 0 ? L.resume : (L.stop ? L.cursor : c.n);
      k.spec_resume[tree] = res;
      if (res < c.n && k.spec_stats) atomic_add_i32(&k.spec_stats[3], 1);
      if (L.bytes) atomic_add_i64(k.O.stat_bytes, (long long)L.bytes);
    }
#ifndef KQ_HOST_EMU
    __syncthreads();  // the workgroup's next tree re-initialises the control words
#endif
  }
}

}  // namespace kq

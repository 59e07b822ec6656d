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
// (derivation in kq_device.hpp, core_run_quad). Level 0 is static (one head per ClusterQueue): E_1 = t_0 is folded into the constants.
// Per item and cohort depth d: K_d = c_d - base_d - val + t_0, T_d = lq_d - base_d, push = max(0, val - t_0); with P the exclusive prefix
// of the cell and E the sum of the t of the cohort levels below:  fits at d <=> P <= E + K_d ;  t_d = max(0, T_d - P) ;  the entry
// pushes max(0, push - E) into the cell. k_records (kq_device.hpp rec_fill_static) computes K, T, push for every cell of every head on
// the whole chip; the workgroup here only gathers them. A depth where no limit is finite and no local quota is left (the usual case
// for all but one or two depths) needs neither arrangement nor scan: nothing there can bind or hold anything back.
//
// Entries the rounds do not take: preemption targets, more than FU flavor-resources / FD path levels, a second head of the same
// ClusterQueue in the batch, operands that are not small, negative reservations (scheduler.go:806). The rounds stop in front of
// the first such entry of the tree and K::spec_resume tells the serial kernel (process_tree) where to take over — also when the
// rounds fail to converge within SP_PMAX passes (everything in front of the first undecided entry is final by then).
#pragma once

namespace kq {

// 1024 threads: 5 items per thread. Measured at cfg 3 (round 6, gpurun_out/r07b/time_spec.txt): the process interval 0.152 ms with 512 threads
// (10 items per thread, 256 VGPRs, 113 spilled), 0.142 ms with 1024 (128 VGPRs, 95 spilled).
#ifndef KQ_SPEC_NT
#define KQ_SPEC_NT 1024
#endif
constexpr int SP_NT = KQ_SPEC_NT;          // threads of the workgroup
constexpr int SP_IPT = 5120 / SP_NT;       // items per thread
constexpr int SP_MAXI = SP_NT * SP_IPT;    // items of a window
constexpr int SP_MAXE = 1024;              // entries of a window
constexpr int SP_MAXS = FD - 1;            // cohort depths on the fast path
constexpr int SP_NW = SP_NT / 64;          // waves
constexpr int SP_PMAX = 40;                // rounds per window before the undecided tail goes back to the serial kernel
constexpr int SP_PADV = (SP_IPT % 2 == 0) ? 1 : 0;  // a thread's SP_IPT consecutive int64 of the scan arrays start an ODD number of elements apart
constexpr int SP_VPAD = SP_MAXI + SP_PADV * SP_NT;  // (bank conflicts otherwise): even SP_IPT => index p + p / SP_IPT
constexpr int SP_SLOTS = 256;              // workgroups of a launch (each loops over trees): one region of K::spec_kt each
static_assert(SP_IPT * SP_NT == 5120 && SP_IPT <= 16, "head flags of a thread are one 16-bit mask");

enum { SPC_DONE = 0, SPC_FIT = 1, SPC_FORCED = 2 };                 // entry class: no items / speculative / unconditional AddUsage (reservation)
enum { SPS_UNKNOWN = 0, SPS_ADMIT = 1, SPS_REJECT = 2, SPS_DROP = 3 };  // DROP: behind the truncation point, handed back

#ifdef KQ_HOST_EMU
static thread_local int g_spec_off = 0;       // tests: 1 = every tree goes to the serial kernel
static thread_local int g_spec_maxe = SP_MAXE, g_spec_maxi = SP_MAXI, g_spec_pmax = SP_PMAX;  // tests: small windows / early truncation
#endif

struct SpecLds {
  // uniform control words (read by everybody after a barrier)
  int32_t cursor, win_start, n_ent, n_items, cut, cut_ent, cut_items, cut_bad, closed, stop, trunc, resume, windows;
  int32_t n_unknown[2], first_unknown[2];  // by round parity: a round's decide phase counts into one pair and clears the other
  int32_t n_act[SP_MAXS];
  int32_t needK[SP_MAXS], needT[SP_MAXS];  // some item's term of Available can bind at this depth / some item has local quota left there
  int32_t w_a[SP_NW], w_b[SP_NW], tot_a, tot_b;
  int64_t w_L[SP_NW], w_U[SP_NW];
  int32_t w_f[SP_NW];
  uint64_t w_c[SP_NW][4];
  int32_t dbase[16];
  int64_t bytes;
  int32_t s_a[SP_NT], s_b[SP_NT];
  int32_t ent[SP_MAXE], pos[SP_MAXE], cqp[SP_MAXE];   // head, iterator position, ClusterQueue | path length << 28
  uint16_t item0[SP_MAXE + 2];
  uint8_t st[SP_MAXE], cls[SP_MAXE], okL[SP_MAXE], okU[SP_MAXE];
  uint16_t desc[SP_MAXI];                  // entry << 3 | slot
  uint16_t hfm[SP_MAXS][SP_NT + 1];        // heads of the cells' segments: bit r of [d][t] = position t * SP_IPT + r starts a cell
  union {
    struct { int64_t vL[SP_VPAD], vU[SP_VPAD]; } v;
    struct { uint16_t ka[SP_MAXI], kb[SP_MAXI], pa[SP_MAXI], pb[SP_MAXI], pF[SP_MAXI], kd[SP_MAXI], ipos[SP_MAXI]; int32_t kc[SP_MAXI]; } s;
  };
};

// What a thread keeps of an item in registers. The per-depth constants K_d / T_d are staged in the workgroup's region of K::spec_kt
// (L2-resident) and loaded where a round needs them: with everything in registers the kernel would need 23 registers per item —
// three quarters of the CU's register file for a 4096-item window.
struct SpecReg {
  int64_t push;        // what the entry pushes out of its ClusterQueue: max(0, val - t_0)
  uint32_t p01, p2e;   // padded positions in the arrangements: depth 0 | depth 1 << 16; depth 2 | (window entry | slot << 10 | act << 13) << 16
  int32_t cidx;        // (head * FU + slot) * FD + path length - 1: cell index of depth d in K::spec_K / spec_T / spec_o is cidx - d
};
struct SpecThread {
  SpecReg it[SP_IPT];
  int64_t eL[SP_IPT], eU[SP_IPT];  // E of the OVER / UNDER world while a round climbs the depths (only while some depth has local quota left)
  int64_t Kd[SP_IPT], Td[SP_IPT];  // the current depth's constants: requested before the scan, used after it
};
KQ_DEV int sp_vpos(const SpecReg& x, int d) { return d == 0 ? (int)(x.p01 & 0xffffu) : (d == 1 ? (int)(x.p01 >> 16) : (int)(x.p2e & 0xffffu)); }
KQ_DEV void sp_set_vpos(SpecReg& x, int d, int p) {
  if (d == 0) x.p01 = (x.p01 & 0xffff0000u) | (uint32_t)p; else if (d == 1) x.p01 = (x.p01 & 0xffffu) | ((uint32_t)p << 16); else x.p2e = (x.p2e & 0xffff0000u) | (uint32_t)p;
}
KQ_DEV int sp_ent(const SpecReg& x) { return (int)((x.p2e >> 16) & 0x3ffu); }
KQ_DEV int sp_slot(const SpecReg& x) { return (int)((x.p2e >> 26) & 7u); }
KQ_DEV int sp_act(const SpecReg& x) { return (int)(x.p2e >> 29); }
static_assert(SP_MAXS <= 3 && SP_MAXE <= 1024 && FU <= 8 && SP_VPAD <= 65536, "SpecReg packs three padded positions, the window entry, the slot and three act bits");
// K_d / T_d of item q in the workgroup's region of K::spec_kt
KQ_DEV size_t sp_kt(int d, int which, int q) { return ((size_t)(d * 2 + which)) * SP_MAXI + q; }
constexpr size_t SP_KT_WORDS = (size_t)SP_MAXS * 2 * SP_MAXI;

KQ_DEV int sp_vidx(int p) { return SP_PADV ? p + p / SP_IPT : p; }
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
KQ_DEV uint32_t sp_field(uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, int d) {  // (selects, not an indexed array: that would live in scratch)
  const int q = d >> 2;
  const uint64_t c = q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3));
  return (uint32_t)((c >> ((d & 3) * 16)) & 0xffffu);
}
// one stable radix pass (4-bit digit at `shift`) of (key, payload) over the first n positions: src -> dst
KQ_DEV void sp_radix_pass(SpecLds& L, const uint16_t* sk, const uint16_t* sp, uint16_t* dk, uint16_t* dp, int n, int shift, int tid) {
  const int base = tid * SP_IPT;
  uint32_t key[SP_IPT], pay[SP_IPT], lr[SP_IPT];
  uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int p = base + r;
    key[r] = p < n ? sk[p] : 0xffffu; pay[r] = p < n ? sp[p] : 0u;
    const int d = (key[r] >> shift) & 15;
    const uint64_t inc = 1ull << ((d & 3) * 16);
    const int q = d >> 2;
    lr[r] = sp_field(c0, c1, c2, c3, d);
    c0 += q == 0 ? inc : 0; c1 += q == 1 ? inc : 0; c2 += q == 2 ? inc : 0; c3 += q == 3 ? inc : 0;
  }
  const uint64_t i0 = (uint64_t)wprefix_incl_i64((int64_t)c0), i1 = (uint64_t)wprefix_incl_i64((int64_t)c1);
  const uint64_t i2 = (uint64_t)wprefix_incl_i64((int64_t)c2), i3 = (uint64_t)wprefix_incl_i64((int64_t)c3);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 63) { L.w_c[wave][0] = i0; L.w_c[wave][1] = i1; L.w_c[wave][2] = i2; L.w_c[wave][3] = i3; }
  __syncthreads();
  if (wave == 0) {  // digit bases: total per digit over the waves, exclusive prefix over the 16 digits
    int t = 0;
    if (lane < 16) for (int w2 = 0; w2 < SP_NW; w2++) t += (int)((L.w_c[w2][lane >> 2] >> ((lane & 3) * 16)) & 0xffffu);
    const int inc = wprefix_incl_i32(t);
    if (lane < 16) L.dbase[lane] = inc - t;
  }
  uint64_t e0 = i0 - c0, e1 = i1 - c1, e2 = i2 - c2, e3 = i3 - c3;
  for (int w2 = 0; w2 < wave; w2++) { e0 += L.w_c[w2][0]; e1 += L.w_c[w2][1]; e2 += L.w_c[w2][2]; e3 += L.w_c[w2][3]; }
  __syncthreads();
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int p = base + r;
    if (p >= n) continue;
    const int d = (key[r] >> shift) & 15;
    const int dst = L.dbase[d] + (int)sp_field(e0, e1, e2, e3, d) + (int)lr[r];
    dk[dst] = (uint16_t)key[r]; dp[dst] = (uint16_t)pay[r];
  }
  __syncthreads();
}
// segmented EXCLUSIVE prefix sums of (vL, vU) over positions [0, n) with the head flags of depth d, in place. Two sweeps over the
// thread's positions (aggregate, then write-back) instead of keeping 4 x SP_IPT values in registers next to the caller's item state.
KQ_DEV void sp_segscan(SpecLds& L, int d, int n, int tid) {
  const int base = tid * SP_IPT, vb = tid * (SP_IPT + SP_PADV);
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

// ---- phases (one call per thread; a barrier follows each) -----------------------------------------------------------------------
// Window building, one chunk of SP_NT iterator positions: which of them are this tree's, what they need.
// s_a = 1 for an entry of the tree, s_b = its items.  Entry classes and the reasons to stop in front of an entry ("bad").
// flags (out): OR of the entry's cell flags (PRec::cbig): bit 2 + d*0 .. see sp_chunk_commit.
// What the rounds need to know of an entry, written per ITERATOR POSITION when the order is scattered (k_order_scatter): window
// building then reads one 16-byte record per position instead of walking order -> head -> ClusterQueue -> record -> cell flags.
// w: class (bits 0-1) | bad << 2 | items << 3 (4 bits) | need << 8 (bits 0-2 needK of depth d, 3-5 needT of depth d, 6 level-0 test fails)
struct SpecHdr { int32_t e, cqp, tree; uint32_t w; };
KQ_DEV SpecHdr spec_hdr_of(const K& k, int e) {
  const int cq = k.H.cq[e];
  const PRec& r = k.grec[e];
  const int mode = r.mode, nuse = r.nuse, plen = r.plen;
  int items = 0, cls = SPC_DONE, bad = 0;
  uint32_t need = 0;
  uint32_t cb = 0;  // flags of the cells the entry touches (cells beyond nuse / plen keep stale bytes: the loop bounds mask them)
  if (!r.slow_static)
    for (int u = 0; u < nuse; u++)
      for (int i = 0; i < plen; i++) {
        const uint32_t f = r.cbig[u][i];
        cb |= f;
        const int d = plen - 1 - i;
        if (i >= 1 && d < SP_MAXS) need |= ((f >> 2) & 1u) << d | ((f >> 3) & 1u) << (3 + d);
      }
  if (cb & 16) need |= 64;
  if (r.slow_static || k.cq_heads[cq] > 1 || (cb & 2)) bad = 1;
  else if (mode == M_NOFIT || nuse == 0) cls = SPC_DONE;
  else if (mode == M_FIT) { cls = SPC_FIT; items = nuse; }
  else if (mode == M_PREEMPT) {  // no targets: reserveCapacityForUnreclaimablePreempt scheduler.go:538-543
    const bool can_always_reclaim = KQ_POL_RECLAIM(r.pol) == KQ_POLICY_ANY;
    if (!can_always_reclaim || (gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS) && (r.flags & KQ_HEAD_IS_PREEMPTOR))) {
      cls = SPC_FORCED; items = nuse;
      // a NEGATIVE reservation (scheduler.go:806 has no max(0, .)) only lowers the ClusterQueue's own cell — nothing is passed up —
      // but it voids the incremental usage_np of the serial kernel once rows are preempted in the tree: leave it to that kernel
      // whenever preemption is possible at all. (bit 7 of need: tells the window to flag the sharding certificate)
      if (cb & 32) { if (k.C.any_preempt) bad = 1; else need |= 128; }
    }
  } else bad = 1;
  if (cls != SPC_FIT) need &= 0xb8u;  // only Fit entries are tested against Available
  SpecHdr h;
  h.e = e; h.cqp = cq | plen << 28; h.tree = k.S.tree_of[cq];
  h.w = (uint32_t)cls | (uint32_t)bad << 2 | (uint32_t)items << 3 | need << 8;
  return h;
}
struct SpecPos { int e, cls, bad, cqp, mine, items, j, i0; uint32_t need; };
KQ_DEV void sp_chunk_classify(const K& k, const SpecCtx& c, int p, SpecPos& o) {
  o.e = -1; o.cls = SPC_DONE; o.bad = 0; o.cqp = 0; o.mine = 0; o.items = 0; o.need = 0; o.j = 0; o.i0 = 0;
  if (p >= c.n) return;
  const SpecHdr h = k.spec_hdr[p];
  if (h.tree != c.tree) return;
  o.e = h.e; o.cqp = h.cqp; o.mine = 1;
  o.cls = (int)(h.w & 3u); o.bad = (int)((h.w >> 2) & 1u); o.items = o.bad ? 0 : (int)((h.w >> 3) & 15u); o.need = h.w >> 8;
}
// after the scan: window slot / first item of the entry; the window closes in front of the first entry that is bad or does not fit
KQ_DEV void sp_chunk_place(const SpecCtx& c, SpecLds& L, int p, SpecPos& o, int j, int i0) {
  o.j = j; o.i0 = i0;
  if (o.mine && (o.bad || j >= c.maxe || i0 + o.items > c.maxi)) atomic_min_i32(&L.cut, p);
}
KQ_DEV void sp_chunk_commit(const K& k, const SpecCtx& c, SpecLds& L, int p, const SpecPos& o) {
  if (!o.mine) return;
  const int j = o.j, i0 = o.i0, cls = o.cls;
  if (p < L.cut) {
    L.ent[j] = o.e; L.pos[j] = p; L.cqp[j] = o.cqp; L.cls[j] = (uint8_t)cls; L.item0[j] = (uint16_t)i0;
    L.st[j] = cls == SPC_FIT ? ((o.need & 64) ? SPS_REJECT : SPS_UNKNOWN) : SPS_ADMIT;
    if (cls != SPC_DONE) {
      for (int d = 0; d < SP_MAXS; d++) { if ((o.need >> d) & 1) L.needK[d] = 1; if ((o.need >> (3 + d)) & 1) L.needT[d] = 1; }
      for (int u = 0; u < o.items; u++) L.desc[i0 + u] = (uint16_t)(j << 3 | u);
      if ((o.need & 128) && k.cert_flags) k.cert_flags[c.tree] = 1;  // a negative reservation is outside the sharding certificate, as in the serial core
    }
  }
  if (p == L.cut) { L.cut_ent = j; L.cut_items = i0; L.cut_bad = o.bad; }  // the window ends in front of this entry
}

// ---- one depth of a round, around the scan -----------------------------------------------------------------------------------------
// before the scan: what every item pushes into its cell of depth d in the two worlds (addUsage: max(0, push - E)); the depth's constants
// are requested here so that their L2 round trip overlaps the scan. ANYT: some depth of the window has local quota left (E can be > 0).
template <bool ANYT, bool LOADK, bool LOADT>
KQ_DEV void sp_stage_push(const K& k, SpecLds& L, const int64_t* kt, SpecThread& ts, int d, int n_items, int tid, bool first) {
  // the constants: k_records' (first window of the tree) or the staged, gain-corrected copies (later windows)
  const int64_t* Kp = first ? k.spec_K : kt + sp_kt(d, 0, 0);
  const int64_t* Tp = first ? k.spec_T : kt + sp_kt(d, 1, 0);
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int q = tid * SP_IPT + r;
    const SpecReg& x = ts.it[r];
    if (q < n_items && (sp_act(x) >> d & 1)) {
      const int ci = first ? x.cidx - d : q;
      if (LOADK) ts.Kd[r] = Kp[ci];
      if (LOADT) ts.Td[r] = Tp[ci];
      const int st = L.st[sp_ent(x)];
      const int vp = sp_vpos(x, d);
      const int64_t pl = ANYT ? i64max(0, x.push - ts.eL[r]) : x.push, pu = ANYT ? i64max(0, x.push - ts.eU[r]) : x.push;
      L.v.vL[vp] = (st == SPS_UNKNOWN || st == SPS_ADMIT) ? pl : 0;
      L.v.vU[vp] = st == SPS_ADMIT ? pu : 0;
    }
  }
}
// after the scan: the level's term of Available for undecided entries (NEEDK), the local quota left (NEEDT); last round (FIN): the
// cell's usage (every admitted item adds what it pushed: atomics, so that depths without a scan need no arrangement either) and the
// sharding certificate's slack
template <bool ANYT, bool NEEDK, bool NEEDT, bool FIN>
KQ_DEV void sp_stage_pull(const K& k, const SpecCtx& c, SpecLds& L, SpecThread& ts, int d, int n_items, int tid) {
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int q = tid * SP_IPT + r;
    const SpecReg& x = ts.it[r];
    if (q >= n_items || !(sp_act(x) >> d & 1)) continue;
    const int vp = sp_vpos(x, d), en = sp_ent(x);
    const int64_t PL = L.v.vL[vp], PU = L.v.vU[vp];
    const int st = L.st[en];
    const int64_t eL = ANYT ? ts.eL[r] : 0, eU = ANYT ? ts.eU[r] : 0;
    if (NEEDK && st == SPS_UNKNOWN) {
      if (PL > eL + ts.Kd[r]) L.okL[en] = 0;
      if (PU > eU + ts.Kd[r]) L.okU[en] = 0;
    }
    if (FIN && st == SPS_ADMIT) {
      const int o = k.spec_o[x.cidx - d];
      if (NEEDK && d == 0 && L.cls[en] == SPC_FIT && k.root_margin)  // sharding certificate: slack of the root term (K::root_margin)
        cert_min(k.root_margin + (size_t)c.tree * c.nfr + (o % c.nfr), (long long)(eL + ts.Kd[r] - PL));
      const int64_t add = i64max(0, x.push - eL);
      if (add != 0) { atomic_add_i64((long long*)k.usage_work + o, (long long)add); atomic_add_i64((long long*)k.usage_np + o, (long long)add); }
    }
    if (NEEDT) {
      ts.eL[r] += i64max(0, ts.Td[r] - PL);
      ts.eU[r] += i64max(0, ts.Td[r] - PU);
    }
  }
}
// last round, a depth without a scan: every admitted item adds what it pushes (nothing at this depth holds anything back)
template <bool ANYT>
KQ_DEV void sp_stage_add(const K& k, SpecLds& L, SpecThread& ts, int d, int n_items, int tid) {
  #pragma unroll
  for (int r = 0; r < SP_IPT; r++) {
    const int q = tid * SP_IPT + r;
    const SpecReg& x = ts.it[r];
    if (q >= n_items || !(sp_act(x) >> d & 1) || L.st[sp_ent(x)] != SPS_ADMIT) continue;
    const int64_t add = ANYT ? i64max(0, x.push - ts.eL[r]) : x.push;
    if (add > 0) { const int o = k.spec_o[x.cidx - d]; atomic_add_i64((long long*)k.usage_work + o, (long long)add); atomic_add_i64((long long*)k.usage_np + o, (long long)add); }
  }
}
template <bool ANYT>
KQ_DEV void sp_stage_push_any(const K& k, SpecLds& L, const int64_t* kt, SpecThread& ts, int d, int n_items, int tid, bool loadk, bool loadt, bool first) {
  if (loadk && loadt) sp_stage_push<ANYT, true, true>(k, L, kt, ts, d, n_items, tid, first);
  else if (loadk) sp_stage_push<ANYT, true, false>(k, L, kt, ts, d, n_items, tid, first);
  else sp_stage_push<ANYT, false, true>(k, L, kt, ts, d, n_items, tid, first);
}
template <bool ANYT>
KQ_DEV void sp_stage_pull_any(const K& k, const SpecCtx& c, SpecLds& L, SpecThread& ts, int d, int n_items, int tid, bool needk, bool needt, bool fin) {
  if (fin) {
    if (needk && needt) sp_stage_pull<ANYT, true, true, true>(k, c, L, ts, d, n_items, tid);
    else if (needk) sp_stage_pull<ANYT, true, false, true>(k, c, L, ts, d, n_items, tid);
    else sp_stage_pull<ANYT, false, true, true>(k, c, L, ts, d, n_items, tid);
  } else if (needk && needt) sp_stage_pull<ANYT, true, true, false>(k, c, L, ts, d, n_items, tid);
  else if (needk) sp_stage_pull<ANYT, true, false, false>(k, c, L, ts, d, n_items, tid);
  else sp_stage_pull<ANYT, false, true, false>(k, c, L, ts, d, n_items, tid);
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

// optional section timing (-DKQ_SPEC_PROF, device only): wall-clock ticks (100 MHz) of thread 0 into K::prof[32 + id]
#if defined(KQ_SPEC_PROF) && !defined(KQ_HOST_EMU)
#define SP_T0() long long _sp_t = wall_clock64()
#define SP_T(id) do { const long long _n = wall_clock64(); if (tid == 0) atomic_add_i64((long long*)k.prof + 32 + (id), _n - _sp_t); _sp_t = _n; } while (0)
#else
#define SP_T0() do {} while (0)
#define SP_T(id) do {} while (0)
#endif
#ifdef KQ_HOST_EMU
#define SP_PHASE(...) for (int tid = 0; tid < SP_NT; tid++) { SpecThread& ts = tsv[tid]; (void)ts; __VA_ARGS__ }
#define SP_BLOCK(...) { const int tid = 0; (void)tid; __VA_ARGS__ }
#define SP_TSDECL std::vector<SpecThread> tsv(SP_NT); struct ChunkRegs { SpecPos a, b; }; std::vector<ChunkRegs> crv(SP_NT);
#define SP_CR crv[tid]
#else
#define SP_PHASE(...) { __VA_ARGS__ } __syncthreads();
#define SP_BLOCK(...) { __VA_ARGS__ }
#define SP_TSDECL SpecThread ts; struct ChunkRegs { SpecPos a, b; } cr_;
#define SP_CR cr_
#endif

// the rounds of one window. ANYT: some depth has local quota left (E must be carried). Returns the number of rounds.
// Barriers per round: per scanning depth push | scan (2) | pull, then decide. Whether the next round is the last one is decided by
// every thread from the same LDS words (no flag to publish); the counters are double-buffered by round parity so that clearing them
// never races with a slow reader.
template <bool ANYT>
KQ_DEV int sp_rounds(const K& k, const SpecCtx& c, SpecLds& L, const int64_t* kt,
#ifdef KQ_HOST_EMU
                     std::vector<SpecThread>& tsv,
#else
                     SpecThread& ts, int tid,
#endif
                     int n_ent, int n_items, bool first) {
  int rounds = 0;
  SP_T0();
  SP_PHASE( for (int j = tid; j < n_ent; j += SP_NT) { L.okL[j] = 1; L.okU[j] = 1; }
            if (tid == 0) { L.n_unknown[0] = 0; L.first_unknown[0] = 0x7fffffff; L.n_unknown[1] = 0; L.first_unknown[1] = 0x7fffffff; } )
  bool fin = false;
  for (int pass = 0;; pass++) {
    const int par = pass & 1;
    rounds++;
    if (ANYT) {
#ifdef KQ_HOST_EMU
      for (int t = 0; t < SP_NT; t++) for (int r = 0; r < SP_IPT; r++) { tsv[t].eL[r] = 0; tsv[t].eU[r] = 0; }
#else
      _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { ts.eL[r] = 0; ts.eU[r] = 0; }
#endif
    }
    for (int d = c.D - 1; d >= 0; d--) {
      const int na = L.n_act[d];
      const bool needk = L.needK[d] != 0, needt = L.needT[d] != 0;
      if (na == 0) continue;
      if (!(needk || needt)) {  // nothing at this depth can bind or hold anything back: no scan; the last round adds the usage
        if (fin) { SP_PHASE( sp_stage_add<ANYT>(k, L, ts, d, n_items, tid); ) }
        continue;
      }
      SP_T(4);
      SP_PHASE( sp_stage_push_any<ANYT>(k, L, kt, ts, d, n_items, tid, needk, needt, first); )
      SP_T(5);  // push
      SP_BLOCK( sp_segscan(L, d, na, tid); )
      SP_T(6);  // scan
      SP_PHASE( sp_stage_pull_any<ANYT>(k, c, L, ts, d, n_items, tid, needk, needt, fin); )
      SP_T(7);  // pull
    }
    if (fin) break;
    SP_PHASE( int nu = 0, fu = 0x7fffffff;
      for (int j = tid; j < n_ent; j += SP_NT) {
        if (L.st[j] == SPS_UNKNOWN) { if (L.okL[j]) L.st[j] = SPS_ADMIT; else if (!L.okU[j]) L.st[j] = SPS_REJECT; else { nu++; if (j < fu) fu = j; } }
        L.okL[j] = 1; L.okU[j] = 1;
      }
      if (tid == 0) { L.n_unknown[par ^ 1] = 0; L.first_unknown[par ^ 1] = 0x7fffffff; }
      if (nu) { atomic_add_i32(&L.n_unknown[par], nu); atomic_min_i32(&L.first_unknown[par], fu); } )
    // converged: one more round with both worlds equal gives the exact usage. No convergence within pmax rounds: everything in
    // front of the first undecided entry is final; the rest goes back to the serial kernel.
    const int nu = L.n_unknown[par], fu = L.first_unknown[par];
    if (nu == 0) { fin = true; continue; }
    if (pass + 1 >= c.pmax) {
      SP_PHASE( for (int j = tid; j < n_ent; j += SP_NT) if (j >= fu) L.st[j] = SPS_DROP;
                if (tid == 0) L.trunc = fu; )
      fin = true;
    }
  }
  SP_T(4);  // rounds: everything outside push / scan / pull
  return rounds;
}

// One root-cohort tree. Device: called by all SP_NT threads of the workgroup with their tid. Emulation: called once (tid ignored).
// kt: this workgroup's region of K::spec_kt (SP_KT_WORDS int64).
KQ_DEV void spec_tree(const K& k, int tree, SpecLds& L, int64_t* kt, int tid_arg) {
  const DSnap& S = k.S;
  SpecCtx c;
  c.tree = tree; c.n = hn(k.H); c.nfr = S.nfr;
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
  SP_PHASE( if (tid == 0) { L.cursor = 0; L.resume = -1; L.bytes = 0; L.stop = 0; L.windows = 0; } )
  if (off) { SP_PHASE( if (tid == 0) k.spec_resume[tree] = 0; ) return; }
  int64_t my_bytes = 0;  // (device: per thread; emulation: the one caller)
  SP_T0();
  for (;;) {  // windows
    // ---- build the window: chunks of SP_NT iterator positions until it is full, a bad entry shows up or the order ends
    SP_PHASE( if (tid == 0) { L.n_ent = 0; L.n_items = 0; L.closed = 0; L.stop = 0; L.win_start = L.cursor; L.trunc = -1; }
              if (tid < SP_MAXS) { L.needK[tid] = 0; L.needT[tid] = 0; L.n_act[tid] = 0; } )
    while (!L.closed) {  // a chunk = 2 * SP_NT iterator positions, two consecutive ones per thread (their loads overlap)
      SP_PHASE( const int p0 = L.cursor + 2 * tid;
                sp_chunk_classify(k, c, p0, SP_CR.a); sp_chunk_classify(k, c, p0 + 1, SP_CR.b);
                L.s_a[tid] = SP_CR.a.mine + SP_CR.b.mine; L.s_b[tid] = SP_CR.a.items + SP_CR.b.items;
                if (tid == 0) L.cut = 0x7fffffff; )
      SP_BLOCK( sp_scan2(L, tid); )
      SP_PHASE( const int p0 = L.cursor + 2 * tid, j = L.n_ent + L.s_a[tid], i0 = L.n_items + L.s_b[tid];
                sp_chunk_place(c, L, p0, SP_CR.a, j, i0); sp_chunk_place(c, L, p0 + 1, SP_CR.b, j + SP_CR.a.mine, i0 + SP_CR.a.items); )
      SP_PHASE( const int p0 = L.cursor + 2 * tid; sp_chunk_commit(k, c, L, p0, SP_CR.a); sp_chunk_commit(k, c, L, p0 + 1, SP_CR.b); )
      SP_PHASE( if (tid == 0) {
        if (L.cut != 0x7fffffff) { L.n_ent = L.cut_ent; L.n_items = L.cut_items; L.cursor = L.cut; L.closed = 1; L.stop = L.cut_bad ? 1 : 0; }
        else { L.n_ent += L.tot_a; L.n_items += L.tot_b; L.cursor += 2 * SP_NT; if (L.cursor >= c.n) { L.cursor = c.n; L.closed = 1; } }
      } )
    }
    SP_T(1);  // window building
    const int n_ent = L.n_ent, n_items = L.n_items;
    const bool last_window = L.stop || L.cursor >= c.n;
    const bool first_window = L.windows == 0;
    bool anyt = false;
    for (int d = 0; d < c.D; d++) anyt = anyt || L.needT[d] != 0;
    SP_PHASE( if (tid == 0) L.windows++; )  // (and: everybody has read the control words before the next window rewrites them)
    if (n_ent == 0) {
      if (last_window) break;
      SP_PHASE( if (tid == 0) L.resume = L.cursor; )  // a window too small for one entry (tests only): hand the rest back
      break;
    }
    int rounds = 0;
    // ---- items: registers; the constants of the depths that need a scan, staged in the workgroup's region. They were computed by
    // ---- k_records against the usage at the start of the cycle; windows after the first subtract what the cell has gained since
    SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r;
      SpecReg& x = ts.it[r];
      x.push = 0; x.p01 = 0; x.p2e = 0; x.cidx = 0;
      if (q < n_items) {
        const int dsc = L.desc[q], j = dsc >> 3, u = dsc & 7, e = L.ent[j];
        const int plen = (int)((uint32_t)L.cqp[j] >> 28);
        int act = 0;
        for (int d = 0; d < c.D; d++) if (plen - 1 - d >= 1) act |= 1 << d;
        x.cidx = (e * FU + u) * FD + plen - 1;
        x.p2e = ((uint32_t)j | (uint32_t)u << 10 | (uint32_t)act << 13) << 16;
        x.push = k.spec_push[(size_t)e * FU + u];
        L.s.ka[q] = (uint16_t)k.O.use_fr[(size_t)e * KQ_MAXU + u]; L.s.pa[q] = (uint16_t)q;  // keys of the first arrangement (by flavor-resource)
      } } )
    if (!first_window) SP_PHASE( for (int d = 0; d < c.D; d++) { if (!(L.needK[d] || L.needT[d])) continue;
      _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; const SpecReg& x = ts.it[r];
        if (q < n_items && (sp_act(x) >> d & 1)) {
          int64_t Kv = k.spec_K[x.cidx - d], Tv = k.spec_T[x.cidx - d];
          { const int o = k.spec_o[x.cidx - d]; const int64_t gain = k.usage_work[o] - k.usage[o]; Kv -= gain; if (Tv != SP_T_INF) Tv -= gain; }
          kt[sp_kt(d, 0, q)] = Kv; kt[sp_kt(d, 1, q)] = Tv;
        } } } )
    SP_T(2);  // item registers, staged constants
    if (n_items > 0 && c.D > 0) {
      // ---- arrangements, only for the depths that scan: F = items by (flavor-resource, iterator position); per depth d: F stably
      // ---- sorted by the rank of the item's cohort of depth d (items without one last)
      bool any_need = false;
      const uint16_t* fkeys = L.s.ka;
      for (int d = 0; d < c.D; d++) any_need = any_need || L.needK[d] || L.needT[d];
      if (any_need) {
        uint16_t *ka = L.s.ka, *kb = L.s.kb, *pa = L.s.pa, *pb = L.s.pb;
        for (int sh = 0; sh < c.frbits; sh += 4) {
          SP_BLOCK( sp_radix_pass(L, ka, pa, kb, pb, n_items, sh, tid); )
          uint16_t* t1 = ka; ka = kb; kb = t1; t1 = pa; pa = pb; pb = t1;
        }
        SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r; if (p < n_items) L.s.pF[p] = pa[p]; } )
        fkeys = ka;  // (read by depth 0 below, before the deeper depths reuse the key buffers)
      }
      for (int d = 0; d < c.D; d++) {
        if (!(L.needK[d] || L.needT[d])) {  // no scan at this depth: only the number of items that have it (0: nothing to add in the last round)
          SP_PHASE( int na = 0; _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { if (tid * SP_IPT + r < n_items && (sp_act(ts.it[r]) >> d & 1)) na++; }
                    if (na) atomic_add_i32(&L.n_act[d], na); )
          continue;
        }
        if (d == 0) {
          // every item has the root and the root is one node: F (sorted by flavor-resource) is the arrangement, a cell starts where the key changes
          const uint16_t *fk = fkeys, *fp = L.s.pF;
          SP_PHASE( uint32_t m = 0; _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r; if (p < n_items) {
            L.s.ipos[fp[p]] = (uint16_t)p; if (p == 0 || fk[p] != fk[p - 1]) m |= 1u << r; } }
            L.hfm[0][tid] = (uint16_t)m; if (tid == 0) L.n_act[0] = n_items; )
          SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; if (q < n_items) sp_set_vpos(ts.it[r], 0, sp_vidx(L.s.ipos[q])); } )
          continue;
        }
        // keys of depth d by item, then along F
        SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; if (q < n_items) {
          const SpecReg& x = ts.it[r];
          int key = c.dcnt[d];
          if (sp_act(x) >> d & 1) key = S.drank[k.spec_o[x.cidx - d] / c.nfr];
          L.s.kd[q] = (uint16_t)key; } } )
        uint16_t *xa = L.s.ka, *xb = L.s.kb, *ya = L.s.pa, *yb = L.s.pb;
        SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r; if (p < n_items) { const int q = L.s.pF[p]; xa[p] = L.s.kd[q]; ya[p] = (uint16_t)q; } } )
        if (c.dcnt[d] > 1 || d > 0)
          for (int sh = 0; sh < c.dbits[d]; sh += 4) {
            SP_BLOCK( sp_radix_pass(L, xa, ya, xb, yb, n_items, sh, tid); )
            uint16_t* t1 = xa; xa = xb; xb = t1; t1 = ya; ya = yb; yb = t1;
          }
        SP_PHASE( _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r; if (p < n_items) L.s.ipos[ya[p]] = (uint16_t)p; } )
        SP_PHASE( int na = 0; _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int q = tid * SP_IPT + r; if (q < n_items) {
          SpecReg& x = ts.it[r];
          const int p = L.s.ipos[q];
          sp_set_vpos(x, d, sp_vidx(p));
          int cell = -1;
          if (sp_act(x) >> d & 1) { cell = k.spec_o[x.cidx - d]; na++; }
          L.s.kc[p] = cell; } }
          if (na) atomic_add_i32(&L.n_act[d], na); )
        SP_PHASE( uint32_t m = 0; _Pragma("unroll") for (int r = 0; r < SP_IPT; r++) { const int p = tid * SP_IPT + r;
          if (p < n_items && (p == 0 || L.s.kc[p] != L.s.kc[p - 1])) m |= 1u << r; }
          L.hfm[d][tid] = (uint16_t)m; )
      }
      SP_T(3);  // arrangements (radix sorts)
      // ---- rounds
#ifdef KQ_HOST_EMU
      rounds = anyt ? sp_rounds<true>(k, c, L, kt, tsv, n_ent, n_items, first_window) : sp_rounds<false>(k, c, L, kt, tsv, n_ent, n_items, first_window);
#else
      rounds = anyt ? sp_rounds<true>(k, c, L, kt, ts, tid, n_ent, n_items, first_window) : sp_rounds<false>(k, c, L, kt, ts, tid, n_ent, n_items, first_window);
#endif
      SP_T0();
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
        const int cq = L.cqp[en] & 0x0fffffff, plen = (int)((uint32_t)L.cqp[en] >> 28);
        const int o0 = k.spec_o[x.cidx - (plen - 1)];   // the ClusterQueue's own cell
        const int64_t nv = k.spec_nv[(size_t)L.ent[en] * FU + sp_slot(x)];
        k.usage_work[o0] = nv; k.usage_np[o0] = nv; k.cq_dirty[cq] = 1;
      } } }
      if (tid == 0 && k.spec_stats) {
        atomic_add_i32(&k.spec_stats[0], 1); atomic_add_i32(&k.spec_stats[1], rounds); atomic_add_i32(&k.spec_stats[2], keep);
        atomic_add_i32(&k.spec_stats[4], n_items); atomic_max_i32(&k.spec_stats[5], rounds); if (keep < n_ent) atomic_add_i32(&k.spec_stats[7], 1);
      } )
    SP_T(8);  // results
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

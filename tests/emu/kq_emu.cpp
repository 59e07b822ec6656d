// kq_emu.cpp — TEST-ONLY build of the engine's device logic with a 1-lane "wave" on the CPU.
//
// Compiles kueue_amd/csrc/kq_device.hpp + kq_host.hpp with -DKQ_HOST_EMU so that the CPU test
// suite (-m "not gpu") can check the engine's control logic (flavor scan, victim search, entry
// processing, host orchestration, buffer sizing) against the oracle without a GPU. It exports
// kqe_* symbols, lives under tests/, and is never loaded by the kueue_amd package: the product
// path has no CPU implementation and fails loudly without the HIP library.
#define KQ_HOST_EMU 1
#define KQ_TAS_CYCLE 1   // the TAS hooks of the cycle are always compiled in here (inert while K::tc is null)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../kueue_amd/csrc/kq_host.hpp"
#include "../../kueue_amd/csrc/kq_tas_host.hpp"

namespace kq {
struct EmuBackend {
  // Device memory is not zeroed by hipMalloc: the emulation hands out POISONED memory (0xA5 bytes; KQE_ZERO_ALLOC=1 restores zeroes), so
  // that code which only works on zero-initialised buffers fails here and not first on the GPU (round 4: the entries of DOut::use_fr past
  // use_n are whatever the allocator left there; an address built from one of them faulted on the MI355X while every CPU test passed).
  void* alloc(size_t n) {
    static const bool zero = getenv("KQE_ZERO_ALLOC") != nullptr;
    static const int poison = getenv("KQE_POISON") ? (int)strtol(getenv("KQE_POISON"), nullptr, 16) : 0xA5;   // (0xA5 words are negative: "absent" to most checks; 01 / 7f make them large valid-looking indices)
    void* p = malloc(n ? n : 1);
    if (p) memset(p, zero ? 0 : poison, n);
    return p;
  }
  void free(void* p) { ::free(p); }
  void* alloc_host(size_t n) { return calloc(n, 1); }
  void free_host(void* p) { ::free(p); }
  void h2d(void* d, const void* h, size_t n) { if (n) memcpy(d, h, n); }   // (n == 0 may come with null pointers: empty tables)
  void d2h(void* h, const void* d, size_t n) { if (n) memcpy(h, d, n); }
  void d2d(void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); }
  void memset(void* d, int v, size_t n) { if (n) ::memset(d, v, n); }
  int sync() { return KQ_OK; }
  const char* error() { return ""; }
  int rot = 0, nom_rot = 0, spec_rot = 0, spec_fixed = -1;
  int max_slots() { return 7; }  // small on purpose: exercises the grid-stride loop over heads
  long long bal_dp() { return (long long)1 << 16; }   // (small: the table is allocated and poisoned per call)
  bool tas_bal = false;          // (the HIP backend picks the _bal kernels by it; the emulation is compiled with KQ_TAS_BAL throughout)
  size_t lds_budget() { return 150 * 1024; }
  int help_blocks(int) { return 1; }  // no helper runs in the emulation, but the leader runs every other task the way one would
  void timer_mark(int) {}
  void stage_select(int) {}
  void step_begin(int) {}
  void step_end() {}
  void usage_levels_side(const DSnap& S, int64_t* usage, int max_depth) { launch_usage_levels(S, usage, max_depth); }
  void usage_join() {}
  void side_fence() {}
  void d2h_side(void* h, const void* d, size_t n) { memcpy(h, d, n); }
  void side_done() {}
  void stage_mark() {}
  int stage_wait() { return KQ_OK; }
  double timer_ms(int, int) { return 0; }
  void launch_tas_classes(const TK& k) { for (int c = 0; c < k.C.n; c++) t_class(k, c); }
  void launch_tas_find(const TK& k, int slots) {
    const int per = (k.Q.n_wl + slots - 1) / slots;
    for (int slot = 0; slot < slots; slot++)
      for (int i = slot * per; i < (slot + 1) * per && i < k.Q.n_wl; i++) t_workload(k, slot, k.C.order[i]);
  }
  void launch_tas_usage(const TTopo& T, int n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int add) { for (int i = 0; i < n; i++) t_usage_cell(T, i, leaf, count, spr, add); }
  void launch_tas_admit(const TTopo& T, const TAdmit& A) { t_admit_seq(T, A); }
  void launch_tas_delta(const TTopo& T, int n, const uint8_t* sel, const int32_t* dom_off, const int32_t* dom_leaf, const int32_t* dom_count, const int64_t* spr, int64_t* plane) {
    for (int p = 0; p < n; p++) t_delta_cell(T, p, sel, dom_off, dom_leaf, dom_count, spr, plane);
  }
  void launch_tas_plane_add(const TTopo& T, const int64_t* plane, int sign) { for (size_t i = 0; i < (size_t)T.n_leaves * T.R; i++) t_plane_add_cell(T, i, plane, sign); }
  void launch_tas_overflow(const TTopo& T, const int64_t* plane, uint8_t* over, int32_t* n_over) { for (int l = 0; l < T.n_leaves; l++) t_overflow_cell(T, l, plane, over, n_over); }
  void launch_tas_excl(const TTopo& T, const TExcl& E) { for (int s = 0; s < E.n_sel; s++) for (int l = 0; l < T.n_leaves; l++) t_excl_cell(T, E, s, l); }
  void launch_tas_fits(const TTopo& T, int n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int32_t* flag) { for (int i = 0; i < n; i++) t_fits_cell(T, i, leaf, count, spr, flag); }
  K last_k{};
  void launch_commit_mask(int n, int32_t* use_n_out, int32_t* cq_out, int32_t* fr_out, int64_t* qty_out, int32_t* count) {
    for (int i = 0; i < n * KQ_MAXU; i++) commit_keep_cell(last_k, i, use_n_out, cq_out, fr_out, qty_out, count);
  }
  void launch_commit_cells(const DSnap& S, const DCommit& c, bool add) {
    // alternate between the two procedures: both leave the ClusterQueue cells right (the second one also updates the cohorts,
    // which the deferred re-derivation overwrites)
    if (rot++ & 1) for (int i = 0; i < c.n * KQ_MAXU; i++) commit_cq_cell(c, S, i / KQ_MAXU, i % KQ_MAXU, add);
    else for (int t = 0; t < S.n_tree; t++) commit_tree(S, c, t, add);
  }
  void launch_commit_trees(const DSnap& S, const DCommit& c, bool add) { for (int t = 0; t < S.n_tree; t++) commit_tree(S, c, t, add); }
  void launch_usage_levels(const DSnap& S, int64_t* usage, int max_depth) {
    for (int dep = max_depth; dep >= 0; dep--)
      for (int i = 0; i < S.nc * S.nfr; i++) if (S.depth[S.nq + i / S.nfr] == dep) derive_usage_cell(S, usage, S.nq + i / S.nfr, i % S.nfr);
  }
  static constexpr bool FUSE_PREP_K = false;
  void launch_prep_k(const DPrep& p, const K&) { launch_prep(p); }
  void launch_prep(const DPrep& p) { for (int o = 0; o < p.n; o++) for (uint32_t i = 0; i < p.op[o].words; i++) prep_word(p, o, i); }
  void launch_derive(const DSnap& S, const DDerive& d, int max_depth) {
    for (int i = 0; i < S.nq * S.nfr; i++) derive_cq_cell(S, d, i / S.nfr, i % S.nfr);
    for (int dep = max_depth; dep >= 0; dep--)
      for (int i = 0; i < S.nc * S.nfr; i++) if (S.depth[S.nq + i / S.nfr] == dep) derive_cohort_cell(S, d, S.nq + i / S.nfr, i % S.nfr);
  }
  void launch_fs_sums(const K& k) {
    for (int n = 0; n < k.S.N; n++) for (int r = 0; r < k.S.nR; r++) fs_sums_cell(k, n, r);
    for (int n = 0; n < k.S.N; n++) fs_pos_node(k, n);
  }
  void launch_pend_heads(const DPend& D, const DGather& G) {
    for (int c = 0; c < D.nq; c++) pend_pop(D, c);
    pend_scan(D, G, 0, 1, nullptr);
    for (int h = 0; h < D.counts[0]; h++) pend_gather_head(D, G, h);
  }
  void launch_step_commit_apply(const DSnap& S, const DCommit& c, const DPend& D, uint32_t gates, int64_t cycle) {
    for (int i = 0; i < c.n * KQ_MAXU; i++) commit_fused_cell(last_k, S, c, i);
    for (int h = 0; h < c.n; h++) pend_apply_head(D, S, last_k.O, last_k.H, gates, cycle, h);
  }
  void launch_step_release(const DSnap& S, const DCommit& c, const DPend& D, int32_t* tree_stamp, int32_t stamp) {
    for (int i = 0; i < c.n * KQ_MAXU; i++) commit_cq_cell(c, S, i / KQ_MAXU, i % KQ_MAXU, false);
    for (int i = 0; i < c.n; i++) pend_release_mark(S, tree_stamp, c.cq, c.use_n, c.n, i, stamp);
    for (int q = 0; q < D.nq; q++) pend_release_requeue(D, S, tree_stamp, q, stamp);
  }
  void launch_afs_usage(const DPend& D, bool init_f64) { for (int l = 0; l < D.A.n_lq; l++) afs_init_lq(D.A, l, init_f64); }
  void launch_afs_sub(const DPend& D, const int32_t* list, int n) { afs_sub_list(D, list, n); }
  void launch_afs_set_consumed(const DPend& D, const int32_t* lq, const uint64_t* lo, const int64_t* hi, const double* f64, const int32_t* settle, int n) {
    for (int i = 0; i < n; i++) afs_set_consumed(D.A, D.lq, lq, lo, hi, f64, settle, i);
  }
  void launch_pend_apply(const DPend& D, const DSnap& S, const DOut& O, const DHeads& H, uint32_t gates, int64_t cycle, int n) {
    for (int h = 0; h < n; h++) pend_apply_head(D, S, O, H, gates, cycle, h);
  }
  void launch_pend_merge(const DPend& D, const int32_t* ord_old, const int32_t* off_old, int32_t* ord_new, int32_t* off_new,
                         const int32_t* fresh, const int32_t* fresh_off, int W0, int n) {
    for (int c = 0; c <= D.nq; c++) off_new[c] = off_old[c] + fresh_off[c];
    for (int j = 0; j < W0; j++) pend_merge_old(D, ord_old, ord_new, fresh, fresh_off, j);
    for (int r = 0; r < n; r++) pend_merge_new(D, ord_old, off_old, ord_new, fresh, r);
  }
  void launch_pend_add_fix(const DPend& D, const DSnap& S, int first, int n) { for (int i = 0; i < n; i++) pend_add_fix(D, S, first + i); }
  void launch_pend_requeue_at(const DPend& D, const DSnap& S, const int32_t* list, const int64_t* at, int n) { for (int i = 0; i < n; i++) pend_requeue_at(D, S, list, at, i); }
  void launch_pend_update_fix(const DPend& D, const int32_t* list, const uint8_t* same_gen, int first, int n) { for (int i = 0; i < n; i++) pend_update_fix(D, list, same_gen, first, i); }
  void launch_pend_delete(const DPend& D, const int32_t* list, int n) { for (int i = 0; i < n; i++) pend_delete(D, list, i); }
  void launch_pend_qi(const DPend& D, const int32_t* list, int n) { for (int i = 0; i < n; i++) pend_queue_inadmissible(D, list ? list[i] : i); }
  void launch_pend_release(const DPend& D, const DSnap& S, int32_t* tree_stamp, const int32_t* cq, const int32_t* use_n, int n, int32_t stamp) {
    for (int i = 0; i < n; i++) pend_release_mark(S, tree_stamp, cq, use_n, n, i, stamp);
    for (int c = 0; c < D.nq; c++) pend_release_requeue(D, S, tree_stamp, c, stamp);
  }
  void launch_usage_delta(int64_t* out, const int64_t* work, const int64_t* start, size_t n) { for (size_t i = 0; i < n; i++) usage_delta_cell(out, work, start, i); }
  void launch_usage_add(int64_t* usage, const int64_t* delta, size_t n, int sign, int32_t* big) { for (size_t i = 0; i < n; i++) usage_add_cell(usage, delta, i, sign, big); }
  void launch_nominate(const K& k, int slots, size_t lds, bool full_pass) {
    std::vector<int64_t> region(lds / 8 + 8);
    // every other launch skips the lean first pass, so that the full pass also sees the heads the lean one would have finished
    const bool lean = !full_pass || (nom_rot++ & 1) == 0;
    if (lean) { for (int slot = 0; slot < slots; slot++) { Wave w{}; for (int h = slot; h < hn(k.H); h += slots) nominate_head_lean(k, w, h); } }
    else { for (int h = 0; h < hn(k.H); h++) k.defer_list[h] = h; *k.defer_count = hn(k.H); }
    // the simulations the lean pass / the emit rounds listed (K::sim_*), dealt to two serial "waves" with both placements of the search arrays
    if (full_pass && k.sim_nscan)
      for (int round = 0; round <= SIM_ROUNDS; round++) {
        if (round > 0) { Wave w{}; for (int i = 0; i < *k.defer_count; i++) nominate_head_emit(k, w, k.defer_list[i]); }
        const int keep = k.sim_ctl[0];
        const int start = round == 0 ? 0 : k.sim_ctl[SIMC_START + round];
        for (int wave = 0; wave < 2; wave++) {
          const int slot = wave % std::max(slots, 1);
          Wave w{};
          if (lds && ((wave + rot) & 1)) { w.cs_lds = (unsigned char*)region.data(); w.cs_lds_bytes = (int)lds; }
          if (wave == 0 && keep - start >= 2) {   // the first half (a wave's last, failing pull takes a ticket: handed back here — on the device the list is empty by then)
            k.sim_ctl[0] = start + (keep - start) / 2;
            sim_worker(k, w, slot, round);
            k.sim_ctl[SIMC_TICKET + round] -= 1; k.sim_ctl[0] = keep;
          } else sim_worker(k, w, slot, round);   // the last "wave" drains the list (and leaves the next round's start behind)
        }
      }
    const int nd = *k.defer_count;
    if (!full_pass) { if (nd != 0) { fprintf(stderr, "kq_emu: the lean pass deferred %d heads but the host skipped the full pass\n", nd); abort(); } return; }
    for (int slot = 0; slot < slots; slot++) {
      Wave w{};
      // alternate between "LDS" and the spill space so that both placements of the search arrays are exercised
      if (lds && ((slot + rot) & 1)) { w.cs_lds = (unsigned char*)region.data(); w.cs_lds_bytes = (int)lds; }
      for (int i = slot; i < nd; i += slots) nominate_head(k, w, k.defer_list[i], slot);
    }
  }
  void launch_shard_export(const K& k, size_t nps_total, int rsn_win) {
    for (int h = 0; h < k.H.n; h++) shard_export_head(k, h, nps_total);
    shard_export_misc(k, nps_total, rsn_win);
    for (int t = 0; t < k.shard.pool_cap; t++) shard_export_pool(k, nps_total, rsn_win, t);
  }
  void launch_shard_import(const K& k, size_t nps_total, int rsn_win) {
    for (int h = 0; h < k.H.n; h++) shard_import_head(k, h, nps_total);
    shard_import_misc(k, nps_total);
    for (int t = 0; t < k.shard.world * k.shard.pool_cap; t++) shard_import_pool(k, nps_total, rsn_win, t);
  }
  static constexpr bool FUSE_RECORDS_ORDER = false;
  void launch_records_order(const K& k, int32_t* order_idx, int32_t* rank) { launch_records(k); launch_order(k, order_idx, rank); }
  void launch_records(const K& k) {
    pack_counts(k);
    for (int e = 0; e < hn(k.H); e++) for (int c = 0; c < FU * FD; c++) rec_fill_static(k, e, c);
  }
  void launch_order(const K& k, int32_t* order_idx, int32_t*) {
    for (int i = 0; i < hn(k.H); i++) {
      int rank = 0;
      for (int j = 0; j < hn(k.H); j++) if (j != i && entry_before(k, j, i)) rank++;
      order_idx[rank] = i;
      if (k.spec_hdr) k.spec_hdr[rank] = spec_hdr_of(k, i);
    }
  }
  void launch_process(const K& k, int n_tree, size_t) {
    // alternate between an LDS-sized cache and none so both code paths are exercised
    // rotate LDS budgets so the chunked/LDS-resident, chunked/HBM-rows and unchunked paths are all exercised
    std::vector<int64_t> lds(160 * 1024 / 8);
    const size_t budgets[3] = {lds.size() * 8, sizeof(PRec) * CH * NBUF + 64, 0};
    // the speculative rounds first (kq_spec.hpp), with rotating window sizes / round limits so that multi-window trees, truncation
    // (the undecided tail goes back to the serial kernel) and "rounds off" are all exercised; then the serial kernel from K::spec_resume
    {
      static thread_local SpecLds sl;
      const int variant = spec_fixed >= 0 ? spec_fixed : spec_rot++ % 5;
      g_spec_off = variant == 4;
      g_spec_maxe = variant == 1 ? 3 : (variant == 2 ? 17 : SP_MAXE);
      g_spec_maxi = variant == 1 ? 16 : (variant == 2 ? 40 : SP_MAXI);
      g_spec_pmax = variant == 3 ? 2 : (variant == 2 ? 3 : SP_PMAX);
      for (int t = 0; t < n_tree; t++) spec_tree(k, t, sl, k.spec_kt + (size_t)(t % SP_SLOTS) * SP_KT_WORDS, 0);
    }
    for (int t = 0; t < n_tree; t++) { Wave w{}; g_emu_pipeline = ((t + rot) % 2); process_tree(k, w, t, t, lds.data(), budgets[(t + rot) % 3], 0, 1); }
    g_emu_pipeline = 0;
    rot++;
    last_k = k;
  }
  // admitted-row structures on the "device" (kq_rows.hpp)
  void launch_rows(const DRows& R, int op, int n) { for (int i = 0; i < n; i++) rows_cell(R, op, i, true); }
  void sort_pairs(uint64_t*& key, int32_t*& val, uint64_t*&, int32_t*&, int n, int bits) {   // (in place here; only the low `bits` bits order)
    std::vector<int> idx(n);
    for (int i = 0; i < n; i++) idx[i] = i;
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return (key[a] & mask) < (key[b] & mask); });
    std::vector<uint64_t> k2(n); std::vector<int32_t> v2(n);
    for (int i = 0; i < n; i++) { k2[i] = key[idx[i]]; v2[i] = val[idx[i]]; }
    for (int i = 0; i < n; i++) { key[i] = k2[i]; val[i] = v2[i]; }
  }
  void scan_excl(const int32_t* in, int32_t* out, int n) { int32_t acc = 0; for (int i = 0; i < n; i++) { const int32_t v = in[i]; out[i] = acc; acc += v; } }
  // kq_cycle_run_tas (kq_tas_cycle.hpp)
  void launch_tas_base(const TCyc* c, int n) { for (int e = 0; e < n; e++) tc_base_cell(*c, e); }
  void launch_tas_cycle_classes(const TCyc* c, int n) { for (int i = 0; i < n; i++) tc_class_init(*c, i / c->ncls, i % c->ncls); }
  void launch_nominate_tas(const K& k, int slots) {
    for (int slot = 0; slot < slots; slot++) { Wave w{}; for (int h = slot; h < hn(k.H); h += slots) nominate_head(k, w, h, slot); }
  }
  // (the placement's working state in "LDS" unless KQ_TAS_LDS_OFF is set; poisoned, as a workgroup's LDS holds anything at launch)
  void launch_process_tas(const K& k, size_t lds_want) {
    Wave w{};
    std::vector<unsigned char> lds(getenv("KQ_TAS_LDS_OFF") ? 0 : lds_want, 0xa5);
    // (helper waves: the leader plays their shares itself, KQ_TAS_EMU_HELPERS; every other launch runs without them)
    static thread_local int launches = 0;
    TLeafJob job{};
    job.nw = 4; job.coop_min = getenv("KQ_TAS_COOP_MIN") ? atoi(getenv("KQ_TAS_COOP_MIN")) : 8;
    process_all_tas(k, w, 0, (launches++ & 1) ? nullptr : &job, lds.data(), (int)lds.size());
    last_k = k;
  }
  void launch_process_fair(const K& k, int n_tree, size_t, size_t, int32_t* rank) {
    std::vector<int64_t> lds(160 * 1024 / 8);
    // (a recomputation's victim search borrows the region: whole state in "LDS" / almost none of it / no region at all)
    const size_t budgets[3] = {lds.size() * 8, sizeof(PRec) + 2048, 0};
    // (the iterator's state in "LDS" for two launches out of three)
    std::vector<unsigned char> iter(fiter_bytes(k.X.max_tree_nodes, k.X.max_tree_cqs) + 64);
    for (int t = 0; t < n_tree; t++) {
      Wave w{};
      const bool lds_iter = (t + rot) % 3 != 1;
      process_tree_fair(k, w, t, t, lds.data(), budgets[(t + rot) % 3], 0, 1, lds_iter ? iter.data() : nullptr, lds_iter ? iter.size() : 0);
    }
    rot++;
    for (int i = 0; i < hn(k.H); i++) rank[i] = k.X.fs_key[i] >= 0 ? fair_rank(k, i, 0, hn(k.H)) : 0;
    for (int i = 0; i < hn(k.H); i++) if (k.X.fs_key[i] >= 0) k.O.order[i] = rank[i];
    last_k = k;
  }
};
}  // namespace kq

typedef kq::EngineT<kq::EmuBackend> EmuEngine;

extern "C" {
// the device's resources.Amount arithmetic (kq_device.hpp a_add / a_addi / a_sub; Cmp is a plain signed compare because Unlimited is
// INT64_MAX) for the reference's known answers. op: 0 Add, 1 AddInt64, 2 Sub, 4 Cmp, 5 CmpInt64
int kqe_amount_op(int32_t op, int64_t a, int64_t b, int64_t* out) {
  switch (op) {
    case 0: *out = kq::a_add(a, b); break;
    case 1: *out = kq::a_addi(a, b); break;
    case 2: *out = kq::a_sub(a, b); break;
    case 4: case 5: *out = a < b ? -1 : (a > b ? 1 : 0); break;
    default: return KQ_EINVAL;
  }
  return KQ_OK;
}
// the device's CountIn (kq_tas_device.hpp t_count_in) on dense [R] vectors; no pods pseudo-resource
int kqe_tas_count_in(int32_t R, const int64_t* req, const int64_t* cap, int32_t* out) {
  kq::TK k{};
  k.T.R = R; k.T.pods = -1;
  *out = kq::t_count_in(k, req, cap);
  return KQ_OK;
}
void kqe_cstat(long long* out) { for (int i = 0; i < 32; i++) { out[i] = kq::g_cs[i]; kq::g_cs[i] = 0; } }
int kqe_engine_create(const kq_config* cfg, void** out) { auto* e = new EmuEngine(); e->cfg = *cfg; *out = e; return KQ_OK; }
void kqe_engine_destroy(void* e) { delete (EmuEngine*)e; }
int kqe_snapshot_patch(void* e, const kq_snapshot* s, uint32_t what) { return ((EmuEngine*)e)->snapshot_patch(s, what); }
int kqe_snapshot_put(void* e, const kq_snapshot* s) { return ((EmuEngine*)e)->snapshot_put(s); }
typedef kq::TasT<kq::EmuBackend> EmuTas;
int kqe_tas_create(void** out) { *out = new EmuTas(); return KQ_OK; }
void kqe_tas_destroy(void* t) { delete (EmuTas*)t; }
int kqe_tas_topology_put(void* t, const kq_tas_topology* tp) { return ((EmuTas*)t)->topology_put(tp); }
int kqe_tas_find(void* t, const kq_tas_requests* r, kq_tas_result* out) { return ((EmuTas*)t)->find(r, out); }
void kqe_tas_use_classes(void* t, int on) { ((EmuTas*)t)->use_classes = on != 0; }
int kqe_tas_usage_apply(void* t, int n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int add) { return ((EmuTas*)t)->usage_apply(n, leaf, count, spr, add); }
int kqe_tas_fits(void* t, int n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int32_t* fits) { return ((EmuTas*)t)->fits(n, leaf, count, spr, fits); }
int kqe_tas_admit(void* t, const kq_tas_requests* r, const kq_tas_result* res, const int32_t* order, int32_t n_order, uint8_t* admitted, int32_t* n_admitted) { return ((EmuTas*)t)->admit(r, res, order, n_order, admitted, n_admitted); }
int kqe_tas_usage_delta(void* t, const kq_tas_requests* r, const kq_tas_result* res, const uint8_t* wl_sel, int64_t* plane) { return ((EmuTas*)t)->usage_delta(r, res, wl_sel, plane); }
int kqe_tas_usage_add(void* t, const int64_t* plane, int32_t sign) { return ((EmuTas*)t)->usage_add(plane, sign); }
int kqe_tas_overflow(void* t, const int64_t* plane, uint8_t* leaf_over, int32_t* n_over) { return ((EmuTas*)t)->overflow(plane, leaf_over, n_over); }
int kqe_tas_find_elastic(void* t, const kq_tas_requests* r, const kq_tas_replacement* x, kq_tas_result* out) { return ((EmuTas*)t)->find_elastic(r, x, out); }
int kqe_tas_find_replacement(void* t, const kq_tas_requests* r, const kq_tas_replacement* x, kq_tas_result* out) { return ((EmuTas*)t)->find_replacement(r, x, out); }
int kqe_tas_exclusion_stats(void* t, const kq_tas_requests* r, const kq_tas_replacement* x, const kq_tas_result* res, int32_t n_sel, const int32_t* podsets, const int32_t* rank, int32_t* td, int32_t* rs) { return ((EmuTas*)t)->exclusion_stats(r, x, res, n_sel, podsets, rank, td, rs); }
int kqe_tas_read_usage(void* t, int64_t* u) { return ((EmuTas*)t)->read_usage(u); }
int64_t kqe_tas_last_bytes(void* t) { return ((EmuTas*)t)->last_bytes; }
const char* kqe_tas_last_error(void* t) { return ((EmuTas*)t)->last_error.c_str(); }
int kqe_snapshot_patch_rows(void* e, const kq_row_patch* p, int32_t* new_index) { return ((EmuEngine*)e)->snapshot_patch_rows(p, new_index); }
int kqe_debug_rows_rebuild(void* e) { return ((EmuEngine*)e)->debug_rows_rebuild(); }
int kqe_debug_read_rows(void* e, int32_t which, void* out, int64_t* bytes) { return ((EmuEngine*)e)->read_rows(which, out, bytes); }
int kqe_cycle_commit(void* e, int32_t* n) { return ((EmuEngine*)e)->cycle_commit(n); }
int kqe_cycle_release(void* e, int age) { return ((EmuEngine*)e)->cycle_release(age); }
int kqe_snapshot_derive(void* e) { return ((EmuEngine*)e)->snapshot_derive(); }
int kqe_read_planes(void* e, int64_t* sq, int64_t* us, uint8_t* fl) { return ((EmuEngine*)e)->read_planes(sq, us, fl); }
int kqe_spec_stats(void* e, int64_t* out8) { return ((EmuEngine*)e)->spec_stats(out8); }
void kqe_spec_variant(void* e, int v) { ((EmuEngine*)e)->be.spec_fixed = v; }  // -1: rotate
void kqe_cs_check(int on) { kq::g_cs_check = on; }
void kqe_disable_scan_search(void* e, int on) { ((EmuEngine*)e)->cs_disable = on != 0; ((EmuEngine*)e)->fs_disable = on != 0; }
void kqe_fs_check(int on) { kq::g_fs_check = on; }
void kqe_force_exact_drs(void* e, int on) { ((EmuEngine*)e)->force_exact_drs = on != 0; }
int kqe_cycle_run_tas(void* e, const kq_heads* h, const kq_cycle_tas* t, kq_decisions* out, kq_cycle_tas_out* tout, int64_t* stats) { return ((EmuEngine*)e)->cycle_run_tas(h, t, out, tout, stats); }
int kqe_cycle_run(void* e, const kq_heads* h, kq_decisions* out) { return ((EmuEngine*)e)->cycle_run(h, out); }
int kqe_cycle_shard_words(void* e, const kq_heads* h, const kq_decisions* out, int32_t world, int64_t* words) { return ((EmuEngine*)e)->cycle_shard_words(h, out, world, words); }
int kqe_cycle_nominate_shard(void* e, const kq_heads* h, const uint8_t* mine, int32_t world, int32_t rank, void* x, kq_decisions* out) { return ((EmuEngine*)e)->cycle_nominate_shard(h, mine, world, rank, x, out); }
int kqe_cycle_process_merged(void* e, int32_t world, int32_t rank, const void* x, kq_decisions* out) { return ((EmuEngine*)e)->cycle_process_merged(world, rank, x, out); }
int kqe_heads_put(void* e, const kq_heads* h, int32_t batch) { return batch < 0 ? KQ_EINVAL : ((EmuEngine*)e)->heads_put(h, batch + 1); }
int kqe_cycle_run_resident(void* e, int32_t batch, kq_decisions* out) { return batch < 0 ? KQ_EINVAL : ((EmuEngine*)e)->cycle_exec(batch + 1, out); }
int kqe_nominate_run_resident(void* e, int32_t batch, kq_decisions* out) { return batch < 0 ? KQ_EINVAL : ((EmuEngine*)e)->cycle_exec(batch + 1, out, true); }
int kqe_pending_put(void* e, const kq_pending* p) { return ((EmuEngine*)e)->pending_put(p); }
int kqe_pending_heads(void* e, int64_t cycle, const uint8_t* act, int32_t* n, int32_t* nps, int32_t* hw) { return ((EmuEngine*)e)->pending_heads(cycle, act, n, nps, hw); }
int kqe_cycle_run_pending(void* e, kq_decisions* out) { return ((EmuEngine*)e)->cycle_run_pending(out); }
int kqe_pending_apply(void* e) { return ((EmuEngine*)e)->pending_apply(); }
int kqe_pending_bounds(void* e, int32_t* a, int32_t* b) { return ((EmuEngine*)e)->pending_bounds(a, b); }
int kqe_pending_step(void* e, int64_t cycle, const uint8_t* act, int32_t tgt_cap, int32_t release_age, int32_t want) { return ((EmuEngine*)e)->pending_step(cycle, act, tgt_cap, release_age, want); }
int kqe_pending_step_wait(void* e, kq_decisions* out, int32_t* n, int32_t* nps, int32_t* hw) { return ((EmuEngine*)e)->pending_step_wait(out, n, nps, hw); }
int kqe_pending_step_reasons(void* e, int32_t cap) { return ((EmuEngine*)e)->pending_step_reasons(cap); }
int kqe_pending_afs_put(void* e, const kq_afs_ledger* l) { return ((EmuEngine*)e)->pending_afs_put(l); }
int kqe_pending_afs_wl_penalty(void* e, int32_t n, const int32_t* wl, const uint64_t* lo, const int64_t* hi, const uint64_t* mask) { return ((EmuEngine*)e)->pending_afs_wl_penalty(n, wl, lo, hi, mask); }
int kqe_pending_afs_sub_penalty(void* e, int32_t n, const int32_t* wl) { return ((EmuEngine*)e)->pending_afs_sub_penalty(n, wl); }
int kqe_pending_afs_set_consumed(void* e, int32_t n, const int32_t* lq, const uint64_t* lo, const int64_t* hi, const double* f64, const int32_t* settle) {
  return ((EmuEngine*)e)->pending_afs_set_consumed(n, lq, lo, hi, f64, settle);
}
int kqe_pending_afs_read(void* e, double* usage, uint64_t* plo, int64_t* phi, uint8_t* pp, uint64_t* clo, int64_t* chi, uint8_t* wrec) {
  return ((EmuEngine*)e)->pending_afs_read(usage, plo, phi, pp, clo, chi, wrec);
}
int kqe_pending_set_lq_usage(void* e, int32_t n, const double* u) { return ((EmuEngine*)e)->pending_set_lq_usage(n, u); }
int kqe_pending_add(void* e, const kq_pending* more, int32_t* first) { return ((EmuEngine*)e)->pending_add(more, first); }
int kqe_pending_update(void* e, int32_t n, const int32_t* wl, const kq_pending* more, int32_t* first) { return ((EmuEngine*)e)->pending_update(n, wl, more, first); }
// the flags of the gathered heads as the cycle will see them (IsPreemptor is decided at gather time)
int kqe_pending_head_flags(void* ep, uint32_t* out, int32_t n) { EmuEngine& e = *(EmuEngine*)ep; for (int i = 0; i < n; i++) out[i] = e.pend.G.flags[i]; return 0; }
int kqe_pending_set_clock(void* e, int64_t now) { return ((EmuEngine*)e)->pending_set_clock(now); }
int kqe_pending_set_requeue_at(void* e, int32_t n, const int32_t* wl, const int64_t* at) { return ((EmuEngine*)e)->pending_set_requeue_at(n, wl, at); }
int kqe_pending_delete(void* e, int32_t n, const int32_t* wl) { return ((EmuEngine*)e)->pending_delete(n, wl); }
int kqe_pending_queue_inadmissible(void* e, int32_t n, const int32_t* cq) { return ((EmuEngine*)e)->pending_queue_inadmissible(n, cq); }
int kqe_pending_read_state(void* e, uint8_t* st, int32_t* counts) { return ((EmuEngine*)e)->pending_read_state(st, counts); }
// Transcribed unit tests of the reference's requeue policy through the DEVICE code: the heads in flight get fabricated decisions
// (status / action / mode / requeue reason per head, tried indices per (podset, resource)) instead of a cycle's.
int kqe_pending_apply_fabricated(void* ep, const uint8_t* status, const uint8_t* action, const uint8_t* mode, const uint8_t* rq, const int32_t* tried) {
  EmuEngine& e = *(EmuEngine*)ep;
  if (!e.pend.valid || e.pend.n_heads < 0) return KQ_EINVAL;
  kq::DOut O{};
  O.status = (uint8_t*)status; O.action = (uint8_t*)action; O.mode = (uint8_t*)mode; O.requeue_reason = (uint8_t*)rq; O.tried_idx = (int32_t*)tried;
  e.pend.O = O; e.pend.H = e.batches[EmuEngine::PEND_SLOT].H; e.pend.ran = true;
  return e.pending_apply();
}
int kqe_cycle_certificate(void* e, int64_t* delta, int64_t* margin, int32_t* flags) { return ((EmuEngine*)e)->cycle_certificate(delta, margin, flags); }
int kqe_snapshot_usage_add(void* e, const int64_t* delta, int32_t sign) { return ((EmuEngine*)e)->snapshot_usage_add(delta, sign); }
// LDS budget of k_process (kq_engine.hip launch_process): static Wave + record buffers
// (the emulation's Wave also carries the TAS view of kq_cycle_run_tas's kernels, which k_process does not have)
void kqe_lds_sizes(int64_t* out) { out[0] = (int64_t)(sizeof(kq::Wave) - sizeof(kq::TAW)); out[1] = (int64_t)(sizeof(kq::PRec) * kq::CH * kq::NBUF); }
int kqe_read_usage(void* e, int64_t* out) { return ((EmuEngine*)e)->read_usage_work(out); }
int kqe_last_bytes(void* e, int64_t* out) { *out = ((EmuEngine*)e)->last_bytes; return KQ_OK; }
int kqe_phase_bytes(void* e, int64_t* out) { out[0] = ((EmuEngine*)e)->last_phase_bytes[0]; out[1] = ((EmuEngine*)e)->last_phase_bytes[1]; return KQ_OK; }
const char* kqe_last_error(void* e) { return ((EmuEngine*)e)->last_error.c_str(); }
}

// ---- include/kq_group.h over emulated engines (tests/test_group.py): the SAME driver as the device build (kq_group_core.hpp: persistent
// rank workers, phase barriers, error containment, the host collective) with the kqe_* engines above and malloc'ed "device" buffers.
// The device collective does not exist here: a group is always created with KQ_GROUP_HOST_COLLECTIVE. `inject`: a rank and a step
// (1 nominate, 2 export copy, 3 import copy, 4 process) that fails once with KQ_EDEVICE — the other ranks must leave the cycle with it.
#include <stdexcept>
#include "../../kueue_amd/csrc/kq_group_core.hpp"
namespace {
struct EmuGroupBackend {
  int inject_rank = -1, inject_step = 0;
  bool take(int rank, int step) { if (rank == inject_rank && step == inject_step) { inject_rank = -1; return true; } return false; }
  bool set_device(int) { return true; }
  int engine_create(const kq_config* c, void** e) { return kqe_engine_create(c, e); }
  void engine_destroy(void* e) { kqe_engine_destroy(e); }
  int rank_init(int, int) { return KQ_OK; }
  void rank_fini(int) {}
  int comm_init(int, const int*, std::string* err) { *err = "the emulation has no device collective"; return KQ_EUNSUPPORTED; }
  void comm_destroy() {}
  void* xalloc(size_t b) { void* p = malloc(b ? b : 1); if (p) memset(p, 0xA5, b); return p; }
  void xfree(void* p) { free(p); }
  void* host_alloc(size_t b) { return calloc(b ? b : 1, 1); }
  void host_free(void* p) { free(p); }
  std::vector<void*> xb;   // exchange buffer of each rank as seen by d2h / h2d (to find the rank for the injection)
  int d2h(void* h, const void* d, size_t b) { for (size_t r = 0; r < xb.size(); r++) if (xb[r] == d && take((int)r, 2)) return KQ_EDEVICE; memcpy(h, d, b); return KQ_OK; }
  int h2d(void* d, const void* h, size_t b) { for (size_t r = 0; r < xb.size(); r++) if (xb[r] == d && take((int)r, 3)) return KQ_EDEVICE; memcpy(d, h, b); return KQ_OK; }
  int allreduce_all(int, const int*, void* const*, size_t, std::string* err) { *err = "the emulation has no device collective"; return KQ_EUNSUPPORTED; }
  int comm_wait(int) { return KQ_OK; }
  int snapshot_put(void* e, const kq_snapshot* s) { return kqe_snapshot_put(e, s); }
  int cycle_run(void* e, const kq_heads* h, kq_decisions* o) { return kqe_cycle_run(e, h, o); }
  int shard_words(void* e, const kq_heads* h, const kq_decisions* o, int world, int64_t* w) { return kqe_cycle_shard_words(e, h, o, world, w); }
  int nominate_shard(void* e, const kq_heads* h, const uint8_t* mine, int world, int rank, void* x, kq_decisions* o) {
    if ((int)xb.size() < world) xb.resize((size_t)world, nullptr);
    xb[(size_t)rank] = x;
    if (take(rank, 1)) return KQ_EDEVICE;
    if (take(rank, 5)) throw std::runtime_error("injected");   // a rank that leaves its job through an exception (Barrier::abort)
    return kqe_cycle_nominate_shard(e, h, mine, world, rank, x, o);
  }
  int process_merged(void* e, int world, int rank, const void* x, kq_decisions* o) { if (take(rank, 4)) return KQ_EDEVICE; if (take(rank, 6)) throw std::bad_alloc(); return kqe_cycle_process_merged(e, world, rank, x, o); }
  int commit(void* e, int32_t* n) { return kqe_cycle_commit(e, n); }
  int release(void* e, int32_t age) { return kqe_cycle_release(e, age); }
  int read_usage(void* e, int64_t* u) { return kqe_read_planes(e, nullptr, u, nullptr); }
  const char* last_error(void* e) { return kqe_last_error(e); }
};
typedef kqg::Group<EmuGroupBackend> EmuGroup;
}  // namespace
extern "C" {
int kqe_group_create(const kq_config* cfg, int32_t n, uint32_t flags, void** out) {
  if (!cfg || !out || n < 1 || n > 64) return KQ_EINVAL;
  auto* g = new EmuGroup();
  g->be.xb.assign((size_t)n, nullptr);   // (sized before the workers exist: nominate_shard only writes its own slot)
  std::vector<int32_t> dev((size_t)n, 0);
  // (bit 30: as given — the duplicate-device refusal of a group without the host collective, tests/test_group.py)
  const int rc = g->create(cfg, n, dev.data(), (flags & (1u << 30)) ? (flags & ~(1u << 30)) : (flags | KQ_GROUP_HOST_COLLECTIVE));
  if (rc != KQ_OK) { g->destroy(); delete g; return rc; }
  *out = g;
  return KQ_OK;
}
void kqe_group_destroy(void* g) { if (g) { ((EmuGroup*)g)->destroy(); delete (EmuGroup*)g; } }
int kqe_group_snapshot_put(void* g, const kq_snapshot* s) { return ((EmuGroup*)g)->snapshot_put(s); }
int kqe_group_cycle_run(void* g, const kq_heads* h, kq_decisions* out) { return ((EmuGroup*)g)->cycle_run(h, out); }
int kqe_group_cycle_commit(void* g, int32_t* n) { return ((EmuGroup*)g)->cycle_commit(n); }
int kqe_group_cycle_release(void* g, int32_t age) { return ((EmuGroup*)g)->cycle_release(age); }
int kqe_group_read_usage(void* g, int32_t rank, int64_t* u) { return ((EmuGroup*)g)->read_usage(rank, u); }
const char* kqe_group_last_error(void* g) { return ((EmuGroup*)g)->last_error.c_str(); }
void kqe_group_inject(void* g, int32_t rank, int32_t step) { ((EmuGroup*)g)->be.inject_rank = rank; ((EmuGroup*)g)->be.inject_step = step; }
}

// kq_host.hpp — host orchestration of the engine behind the C ABI (include/kq_engine.h).
//
// Templated on a Backend that owns "device" memory and launches the three phases of a cycle:
//   nominate  (one wave per head)        scheduler.go:665   nominate
//   order     (rank by pairwise compare) scheduler.go:1110  makeClassicalIterator
//   process   (one wave per root tree)   scheduler.go:392   processEntry, sequential inside a tree
// HipBackend (kq_engine.hip) is the product. EmuBackend (tests/emu) runs the same kernels as plain
// loops with a 1-lane wave so the CPU suite can check the control logic; it is never shipped.
#pragma once
#include <cstdlib>
#include <algorithm>
#include <map>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <unordered_map>

#include "kq_device.hpp"
#include "kq_tas_cycle.hpp"
#include "kq_rows.hpp"

namespace kq {

template <class B> struct EngineT {
  B be;
  kq_config cfg{};
  std::string last_error;
  Prep prep;
  bool have_snapshot = false;
  DSnap S{};
  int64_t* d_usage = nullptr;       // cycle-start usage plane (mutable by derive)
  int64_t* d_sq = nullptr;          // subtree quota (mutable by derive)
  uint8_t* d_qflags = nullptr;
  int32_t* d_big = nullptr;         // K::usage_big
  std::vector<void*> snap_allocs;
  // cycle buffers (grow-only)
  struct Buf { void* p = nullptr; size_t cap = 0; };
  std::vector<Buf*> all_bufs;
  Buf b_usage_work, b_usage_np, b_preempted, b_w, b_cqinfo, b_cls, b_tgt_row, b_tgt_reason, b_order, b_misc, b_prof, b_nom, b_rank, b_cand, b_mark, b_rmb, b_grec, b_cqd, b_defer, b_cert, b_help, b_cqh, b_spkt, b_spc, b_sphdr, b_shard, b_addstage, b_fs[20];
#ifdef KQ_HOST_EMU
  bool spec_stats_on = true;
#else
  bool spec_stats_on = getenv("KQ_SPEC_STATS") != nullptr;  // kq_debug_spec_stats: a handful of global atomics per window, off by default
#endif
  bool force_exact_drs = false;  // tests: take the saturation-safe DRS loops even when the sums would be exact
  bool cs_disable = false;       // tests: classical victim searches always take the candidate-by-candidate walk
  bool fs_disable = false;       // tests: fair-sharing victim searches always take the walk
  bool fs_lrun_on = [] { const char* e = getenv("KQ_FS_LRUN"); return !(e && e[0] == '0'); }();    // A/B switch of the fair iterator's leader-only runs (process_tree_fair)
  int cs_lazy_mode = [] { const char* e = getenv("KQ_CS_LAZY"); return e ? atoi(e) : 1; }();   // prefix rounds of the scan search (kq_cs.hpp): 0 off, 1 in recomputations, 2 always
  // A/B switches of the batched fair search (kq_fs.hpp): bit 0 first strategy, 1 second strategy, 2 fill-back, 3 no restore walk (KQ_FS_BATCH=<bits>)
  int fs_batch_bits = [] { const char* e = getenv("KQ_FS_BATCH"); return e ? atoi(e) : 15; }();
  bool help_disable = false;     // tests: no helper workgroups
  Buf b_cs;
  struct HeadBatch { Buf hb[1]; DHeads H{}; int n = 0; size_t nps = 0; int slot_cap = 1; int64_t cycle = 0; bool valid = false; bool plain = true; int max_nps = KQ_MAXPS; bool partial = true; };
  std::vector<HeadBatch> batches;  // [0] = transient batch of kq_cycle_run, [1+b] = resident batch b
  Buf ob[24];  // output arrays
  uint8_t* hstage = nullptr;  // pinned host staging for the packed decisions
  size_t hstage_cap = 0;
  uint8_t* hup = nullptr;     // pinned host staging for a batch of heads (one H2D per kq_heads_put / kq_cycle_run)
  size_t hup_cap = 0;
  double last_kernel_ms = 0;
  double last_phase_ms[3] = {0, 0, 0};
  int64_t last_bytes = 0;
  int64_t last_phase_bytes[2] = {0, 0};

  int fail(int code, const std::string& msg) { last_error = msg; return code; }

  // ---- pending side on the device (kq_pending.hpp) ---------------------------------------------------------------------
  static constexpr int PEND_SLOT = 4098;  // head batch the gathered heads live in (beyond every caller-visible batch id)
  struct Pending {
    bool valid = false;
    int W = 0, nq = 0, nR = 0, nF = 0, n_tree = 0, slot_cap = 1, max_nps = 1;
    bool plain = true;
    bool partial = false;      // some pending podset may be admitted partially
    bool grouped = false;      // some resident workload has a PodSetGroupName group of several podsets (kq_heads.ps_group)
    std::vector<void*> allocs;
    DPend D{};
    DGather G{};
    int n_heads = -1;        // heads of the cycle in flight (popped, not applied yet); -1 = none
    int n_ps = 0;
    bool ran = false;        // kq_cycle_run_pending succeeded on those heads
    int64_t cycle = 0;
    DOut O{};
    DHeads H{};
    uint8_t* d_active = nullptr;
    int32_t* d_list = nullptr;
    int32_t* d_tree_stamp = nullptr;  // [n_tree] last release that freed quota in the tree
    double* d_lq_usage = nullptr;     // [n_lq] AdmissionFairSharing usage of every LocalQueue
    int n_lq = 0;
    int32_t release_seq = 0;
    int64_t now = 0;          // the queues' clock (kq_pending_set_clock)
    // host mirror of what kq_pending_add needs to merge new workloads into the heap orders and to size the gathered batch
    std::vector<int32_t> h_cq; std::vector<int64_t> h_prio, h_ts; std::vector<uint32_t> h_uid;
    std::vector<int> mps, mrq;       // widest workload of every ClusterQueue (podsets / requests)
    std::vector<int32_t> h_ord;      // the heap orders as uploaded by kq_pending_put
    int32_t* ord_alt = nullptr; int32_t* cq_off_alt = nullptr;   // the second order / offsets buffer kq_pending_add merges into
    size_t gps = 0, grq = 0;         // podset / request rows of the gathered batch as allocated
    size_t nps_total = 0, nreq_total = 0;
  } pend;
  // grow a resident array of the pending set by `add_n` elements (tail from the host, or a fill byte). Arrays are reallocated with
  // head room, so that a stream of small arrivals mostly costs the copy of its own rows.
  std::vector<std::pair<const void*, size_t>> pend_caps;  // capacity in bytes of the arrays pend_regrow allocated
  size_t pend_cap_of(const void* d) const { for (auto& c : pend_caps) if (c.first == d) return c.second; return 0; }
  template <class T> void pend_regrow(T*& d, size_t old_n, size_t add_n, const T* tail, int fill = -2) {
    const size_t need = std::max<size_t>(old_n + add_n, 1) * sizeof(T);
    T* nd = d;
    if (need > pend_cap_of(d)) {
      const size_t cap = need + need / 4 + 65536;
      nd = (T*)be.alloc(cap);
      if (old_n) be.d2d(nd, d, old_n * sizeof(T));
      for (size_t i = 0; i < pend.allocs.size(); i++) if (pend.allocs[i] == (void*)d) { be.free(pend.allocs[i]); pend.allocs.erase(pend.allocs.begin() + i); break; }
      for (size_t i = 0; i < pend_caps.size(); i++) if (pend_caps[i].first == (const void*)d) { pend_caps.erase(pend_caps.begin() + i); break; }
      pend.allocs.push_back(nd);
      pend_caps.emplace_back((const void*)nd, cap);
    }
    if (add_n) { if (tail) be.h2d(nd + old_n, tail, add_n * sizeof(T)); else if (fill != -2) be.memset(nd + old_n, fill, add_n * sizeof(T)); }
    d = nd;
  }
  template <class T> void pend_regrow(const T*& d, size_t old_n, size_t add_n, const T* tail, int fill = -2) {
    T* m = const_cast<T*>(d);
    pend_regrow(m, old_n, add_n, tail, fill);
    d = m;
  }
  void pend_release_ptr(const void* d) {
    for (size_t i = 0; i < pend.allocs.size(); i++) if (pend.allocs[i] == d) { be.free(pend.allocs[i]); pend.allocs.erase(pend.allocs.begin() + i); break; }
    for (size_t i = 0; i < pend_caps.size(); i++) if (pend_caps[i].first == d) { pend_caps.erase(pend_caps.begin() + i); break; }
  }
  // heap order of every ClusterQueue from the host mirror (baseCompareFunc cluster_queue.go:844 without the sticky term)
  void pend_sort(std::vector<int32_t>& ord, std::vector<int32_t>& cq_off) {
    const Pending& P = pend;
    const int W = (int)P.h_cq.size(), nq = prep.nq;
    ord.resize(W); cq_off.assign(nq + 1, 0);
    for (int w = 0; w < W; w++) { ord[w] = w; cq_off[P.h_cq[w] + 1]++; }
    for (int c = 0; c < nq; c++) cq_off[c + 1] += cq_off[c];
    std::sort(ord.begin(), ord.end(), [&](int a, int b) {
      if (P.h_cq[a] != P.h_cq[b]) return P.h_cq[a] < P.h_cq[b];
      if (P.h_prio[a] != P.h_prio[b]) return P.h_prio[a] > P.h_prio[b];
      if (P.h_ts[a] != P.h_ts[b]) return P.h_ts[a] < P.h_ts[b];
      if (P.h_uid[a] != P.h_uid[b]) return P.h_uid[a] < P.h_uid[b];
      return a < b;
    });
  }
  void pend_alloc_gather() {  // the gathered batch holds <= 1 head per ClusterQueue: sized for the widest workload of every ClusterQueue
    if (last_slot == PEND_SLOT) last_cycle_n = -1;  // kq_cycle_commit re-reads the last cycle's K block, whose H.* point into the arrays freed here
    Pending& P = pend;
    const int nq = prep.nq, nR = prep.nR;
    const size_t nfw = (prep.nF + 63) / 64;
    size_t gps = 0, grq = 0;
    for (int c = 0; c < nq; c++) { gps += P.mps[c]; grq += P.mrq[c]; }
    gps += gps / 8 + 16; grq += grq / 8 + 64;  // head room: arrivals seldom widen a ClusterQueue's widest workload
    P.gps = gps; P.grq = grq;
    DGather& G = P.G;
    for (const void* q : {(const void*)G.cq, (const void*)G.priority, (const void*)G.queue_ts, (const void*)G.flags, (const void*)G.ps_off, (const void*)G.ps_count,
                          (const void*)G.ps_min_count, (const void*)G.ps_req_off, (const void*)G.req_res, (const void*)G.req_qty, (const void*)G.ps_flavor_ok,
                          (const void*)G.ps_last_tried, (const void*)G.last_generation, (const void*)G.last_cycle, (const void*)G.last_hash, (const void*)G.hash,
                          (const void*)G.slice_row, (const void*)G.ps_slice_count, (const void*)G.req_slice_flavor, (const void*)G.ps_slice_pods_flavor,
                          (const void*)G.req_slice_qty, (const void*)G.ps_slice_pods_qty, (const void*)G.ps_group})
      if (q) pend_release_ptr(q);
    G.cq = pend_alloc<int32_t>(nq); G.priority = pend_alloc<int64_t>(nq); G.queue_ts = pend_alloc<int64_t>(nq); G.flags = pend_alloc<uint32_t>(nq);
    G.ps_off = pend_alloc<int32_t>(nq + 1);
    G.ps_count = pend_alloc<int32_t>(gps); G.ps_min_count = pend_alloc<int32_t>(gps); G.ps_req_off = pend_alloc<int32_t>(gps + 1);
    G.req_res = pend_alloc<int32_t>(grq); G.req_qty = pend_alloc<int64_t>(grq);
    G.ps_flavor_ok = pend_alloc<uint64_t>(gps * nfw); G.ps_last_tried = pend_alloc<int32_t>(gps * nR);
    G.last_generation = pend_alloc<int64_t>(nq); G.last_cycle = pend_alloc<int64_t>(nq);
    G.last_hash = pend_alloc<uint64_t>(nq); G.hash = pend_alloc<uint64_t>(nq);
    G.slice_row = nullptr; G.ps_slice_count = nullptr; G.req_slice_flavor = nullptr; G.ps_slice_pods_flavor = nullptr; G.req_slice_qty = nullptr; G.ps_slice_pods_qty = nullptr;
    G.ps_group = P.grouped ? pend_alloc<int32_t>(gps) : nullptr;   // PodSetGroupName groups among the resident workloads: the batch carries kq_heads.ps_group
    if (P.D.P.slice_row) {   // the resident set holds workload slices
      G.slice_row = pend_alloc<int32_t>(nq); G.ps_slice_count = pend_alloc<int32_t>(gps); G.ps_slice_pods_flavor = pend_alloc<int32_t>(gps);
      G.ps_slice_pods_qty = pend_alloc<int64_t>(gps); G.req_slice_flavor = pend_alloc<int32_t>(grq); G.req_slice_qty = pend_alloc<int64_t>(grq);
    }
  }
  // the gathered slice columns of a batch of heads (kq_pending_heads / kq_pending_step)
  void pend_wire_slices(DHeads& H) {
    const DGather& G = pend.G;
    H.slice_row = G.slice_row; H.ps_slice_count = G.ps_slice_count; H.req_slice_flavor = G.req_slice_flavor; H.req_slice_qty = G.req_slice_qty;
    H.ps_slice_pods_flavor = G.ps_slice_pods_flavor; H.ps_slice_pods_qty = G.ps_slice_pods_qty;
  }
  template <class T> T* pend_alloc(size_t n, const T* host = nullptr, int fill = -2) {
    T* d = (T*)be.alloc(std::max<size_t>(n, 1) * sizeof(T));
    if (host && n) be.h2d(d, host, n * sizeof(T));
    else if (fill != -2) be.memset(d, fill, std::max<size_t>(n, 1) * sizeof(T));
    pend.allocs.push_back(d);
    return d;
  }
  void pending_free() {
    if (last_slot == PEND_SLOT) last_cycle_n = -1;  // the uncommitted pending cycle's gathered head arrays are freed below
    for (void* p : pend.allocs) be.free(p);
    steps_issued = steps_waited = 0; steps[0].busy = steps[1].busy = false;   // steps in flight die with the set they ran on
    const int64_t now = pend.now;
    pend = Pending{};
    pend.now = now;
    pend_caps.clear();
    if ((int)batches.size() > PEND_SLOT) batches[PEND_SLOT].valid = false;
  }

  template <class T> T* upload(const T* host, size_t n) {
    T* d = (T*)be.alloc(std::max<size_t>(n, 1) * sizeof(T));
    if (n) be.h2d(d, host, n * sizeof(T));
    snap_allocs.push_back(d);
    return d;
  }
  template <class T> T* grow(Buf& b, size_t n) {
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (bytes > b.cap) {
      if (b.p) be.free(b.p);
      // (KQ_EXACT_ALLOC=1, debugging: no head room, so that an overrun of the requested size meets the allocator's own end — ASan's red
      // zone under the emulation, the fence of KQ_EFENCE on the device — instead of the 25 % nobody asked for)
      static const bool exact = getenv("KQ_EXACT_ALLOC") != nullptr;
      size_t cap = exact ? bytes : bytes + bytes / 4;
      b.p = be.alloc(cap);
      b.cap = cap;
    }
    return (T*)b.p;
  }
  void free_snapshot() {
    for (void* p : snap_allocs) be.free(p);
    snap_allocs.clear();
    have_snapshot = false;
  }
  ~EngineT() {
    pending_free();
    free_snapshot();
    if (hstage) be.free_host(hstage);
    if (hup) be.free_host(hup);
    for (Buf* b : {&b_usage_work, &b_usage_np, &b_preempted, &b_w, &b_cqinfo, &b_cls, &b_tgt_row, &b_tgt_reason, &b_order, &b_misc, &b_prof, &b_nom, &b_rank, &b_cand, &b_mark, &b_rmb, &b_grec, &b_cqd, &b_cs, &b_defer, &b_cert, &b_help, &b_cqh, &b_spkt, &b_spc, &b_sphdr, &b_shard, &b_addstage}) if (b->p) be.free(b->p);
    for (auto& b : b_fs) if (b.p) be.free(b.p);
    for (auto& c : ring) for (Buf* b : {&c.cq, &c.use_n, &c.use_fr, &c.use_qty}) if (b->p) be.free(b->p);
    for (auto& hbch : batches) for (auto& b : hbch.hb) if (b.p) be.free(b.p);
    for (auto& b : ob) if (b.p) be.free(b.p);
  }

  // cache.Snapshot -> HBM (snapshot.go:171)
  int snapshot_put(const kq_snapshot* s) {
    free_snapshot();
    for (auto& c : ring) c.live = false;
    commits = 0; last_cycle_n = -1;
    // Resident head batches were validated against, and carry the strides of, the snapshot they were uploaded under
    // (cq < nq, req_res < nR, ps_flavor_ok / ps_last_tried row widths): a new snapshot voids them. kq_heads_put again.
    for (size_t b = 0; b < batches.size(); b++) if ((int)b != PEND_SLOT) batches[b].valid = false;
    prep.want_fs = cfg.fair_sharing != 0;
    // the structures derived from the admitted rows are built on the device from the uploaded row table (kq_rows.hpp) unless fair
    // sharing needs its host-built position-order tables or the sizes leave the sort keys' fields
    prep.skip_rows = rows_device && s->n_adm < (1 << 20) && s->n_cq + s->n_cohort < (1 << 21);
    int rc = build_prep(s, prep);
    if (rc != KQ_OK) return fail(rc, prep.err);
    if (prep.skip_rows && !rows_device_ok()) { prep.skip_rows = false; rc = build_prep(s, prep); if (rc != KQ_OK) return fail(rc, prep.err); }
    // the pending store is indexed by ClusterQueue / resource / flavor: it survives a snapshot refresh with the same dictionary
    if (pend.valid && (pend.nq != prep.nq || pend.nR != prep.nR || pend.nF != prep.nF || pend.n_tree != prep.n_tree)) pending_free();
    const size_t N = prep.N, nfr = prep.nfr, nq = prep.nq;
    S = DSnap{};
    S.nq = prep.nq; S.nc = prep.nc; S.N = prep.N; S.nF = prep.nF; S.nR = prep.nR; S.nfr = prep.nfr;
    S.pods_res = s->pods_resource; S.n_adm = prep.n_adm; S.n_tree = prep.n_tree; S.nfw = (prep.nF + 63) / 64;
    S.resource_order = upload(s->resource_order, prep.nR);
    S.parent = upload(s->parent, N);
    S.nominal = upload(s->nominal, N * nfr);
    S.bl = upload(s->borrow_limit, N * nfr);
    S.ll = upload(s->lend_limit, N * nfr);
    d_sq = upload(s->subtree_quota, N * nfr); S.sq = d_sq;
    d_usage = upload(s->usage, N * nfr); levels_stale = false;
    d_qflags = upload(s->quota_flags, N * nfr); S.qflags = d_qflags;
    { const int32_t zero = 0; d_big = upload(&zero, 1); }
    S.cq_rg_off = upload(s->cq_rg_off, nq + 1);
    S.rg_flavor_off = upload(s->rg_flavor_off, prep.n_rg + 1);
    S.rg_flavor = upload(s->rg_flavor, s->rg_flavor_off[prep.n_rg]);
    S.rg_res_off = upload(s->rg_res_off, prep.n_rg + 1);
    S.rg_res = upload(s->rg_res, s->rg_res_off[prep.n_rg]);
    S.cq_policy = upload(prep.cq_policy_dev.data(), nq);
    S.cq_thr = upload(s->cq_borrow_prio_threshold, nq);
    S.cq_gen = upload(s->cq_generation, nq);
    S.cq_adm_off = upload(s->cq_adm_off, nq + 1);
    S.adm_prio = upload(s->adm_priority, prep.n_adm);
    S.adm_qts = upload(s->adm_queue_ts, prep.n_adm);
    S.adm_flags = upload(s->adm_flags, prep.n_adm);
    S.adm_use_off = upload(s->adm_use_off, prep.n_adm + 1);
    S.adm_use_fr = upload(s->adm_use_fr, s->adm_use_off[prep.n_adm]);
    S.adm_use_qty = upload(s->adm_use_qty, s->adm_use_off[prep.n_adm]);
    S.path = upload(prep.path.data(), prep.path.size());
    S.plen = upload(prep.plen.data(), prep.plen.size());
    S.tree_of = upload(prep.tree_of.data(), prep.tree_of.size());
    S.node_local = upload(prep.node_local.data(), prep.node_local.size());
    S.node_height = upload(prep.node_height.data(), prep.node_height.size());
    S.adm_cq = upload(prep.adm_cq.data(), prep.adm_cq.size());
    h_adm_cq = prep.adm_cq; h_cq_adm_off.assign(s->cq_adm_off, s->cq_adm_off + prep.nq + 1);
    S.cq_local = upload(prep.cq_local.data(), prep.cq_local.size());
    S.tree_node_off = upload(prep.tree_node_off.data(), prep.tree_node_off.size());
    S.tree_nodes = upload(prep.tree_nodes.data(), prep.tree_nodes.size());
    S.tree_cq_off = upload(prep.tree_cq_off.data(), prep.tree_cq_off.size());
    S.tree_cqs = upload(prep.tree_cqs.data(), prep.tree_cqs.size());
    S.tree_row_off = upload(prep.tree_row_off.data(), prep.tree_row_off.size());
    S.tree_rows = upload(prep.tree_rows.data(), prep.tree_rows.size());
    S.lendable = upload(prep.lendable.data(), prep.lendable.size());
    S.rank_pos = upload(prep.rank_pos.data(), prep.rank_pos.size());
    S.frcount = upload(prep.frcount.data(), prep.frcount.size());
    S.frb_off = upload(prep.frb_off.data(), prep.frb_off.size());
    S.frb = upload(prep.frb.data(), prep.frb.size());
    S.cq_row_bytes = upload(prep.cq_row_bytes.data(), prep.cq_row_bytes.size());
    S.tree_rows_asc = upload(prep.tree_rows_asc.data(), prep.tree_rows_asc.size());
    S.top_of = upload(prep.top_of.data(), prep.top_of.size());
    S.fair_weight = upload(s->fair_weight, N);
    S.child_cohort_off = upload(s->child_cohort_off, prep.nc + 1);
    S.child_cohort = upload(s->child_cohort, s->child_cohort_off[prep.nc]);
    S.child_cq_off = upload(s->child_cq_off, prep.nc + 1);
    S.child_cq = upload(s->child_cq, s->child_cq_off[prep.nc]);
    S.depth = upload(prep.depth.data(), prep.depth.size());
    S.adm_rts = upload(s->adm_reserve_ts, prep.n_adm);
    S.adm_uid = upload(s->adm_uid_rank, prep.n_adm);
    S.adm_rec = upload(prep.adm_rec.data(), prep.adm_rec.size());
    S.adm_recx = upload(prep.adm_recx.data(), prep.adm_recx.size());
    for (int l = 0; l < CS_LEVELS; l++) S.frl[l] = upload(prep.frl[l].data(), prep.frl[l].size());
    S.frbr = upload(prep.frbr.data(), prep.frbr.size());
    S.frec = upload(prep.frec.data(), prep.frec.size());
    S.frb_sig = upload(prep.frb_sig.data(), prep.frb_sig.size());
    S.cs_ok = upload(prep.cs_ok.data(), prep.cs_ok.size());
    upload_fs_rows(); upload_fs_quota();
    S.fs_kid = upload(prep.fs_kid.data(), prep.fs_kid.size()); S.fs_koff = upload(prep.fs_koff.data(), prep.fs_koff.size());
    S.fs_knc = upload(prep.fs_knc.data(), prep.fs_knc.size()); S.fs_knh = upload(prep.fs_knh.data(), prep.fs_knh.size());
    S.fs_par = upload(prep.fs_par.data(), prep.fs_par.size());
    S.tree_depth = upload(prep.tree_depth.data(), prep.tree_depth.size());
    S.drank = upload(prep.drank.data(), prep.drank.size()); S.tree_dcnt = upload(prep.tree_dcnt.data(), prep.tree_dcnt.size());
    S.cq_res_rg = upload(prep.cq_res_rg.data(), prep.cq_res_rg.size());
    rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    have_snapshot = true;
    if (prep.skip_rows) { rc = rows_rebuild(prep.n_adm); if (rc != KQ_OK) { have_snapshot = false; return rc; } }
    return KQ_OK;
  }

  // kq_snapshot_patch: the snapshot of the next cycle when only usage and / or the admitted set moved — what
  // clusterQueue.updateWorkloadUsage (pkg/cache/scheduler/clusterqueue.go:594) and updateCohortResourceNode (resource_node.go:190)
  // change between two cache.Snapshot() calls. The quota tree, policies and dictionaries must be the ones of the last
  // kq_snapshot_put (checked where cheap); resident head batches and the pending set stay valid.
  template <class T> void reupload(const T*& field, const T* host, size_t n) {
    for (size_t i = 0; i < snap_allocs.size(); i++) if (snap_allocs[i] == (const void*)field) { be.free(snap_allocs[i]); snap_allocs.erase(snap_allocs.begin() + i); break; }
    field = upload(host, n);
  }
  // static structures of the LDS-resident fair victim search: the part that follows the admitted set / the part that follows the quotas
  void upload_fs_rows() {
    reupload(S.rec_ok, prep.rec_ok.data(), prep.rec_ok.size());
    reupload(S.fs_scan, prep.fs_scan.data(), prep.fs_scan.size()); reupload(S.fs_apply, prep.fs_apply.data(), prep.fs_apply.size());
    reupload(S.fs_posoff, prep.fs_posoff.data(), prep.fs_posoff.size());
  }
  void upload_fs_quota() {
    reupload(S.fs_ok, prep.fs_ok.data(), prep.fs_ok.size());
    reupload(S.fs_c0, prep.fs_c0.data(), prep.fs_c0.size()); reupload(S.fs_c1, prep.fs_c1.data(), prep.fs_c1.size());
    reupload(S.fs_q, prep.fs_q.data(), prep.fs_q.size());
    reupload(S.fs_lend, prep.fs_lend.data(), prep.fs_lend.size()); reupload(S.fs_weight, prep.fs_weight.data(), prep.fs_weight.size());
  }
  int snapshot_patch(const kq_snapshot* s, uint32_t what) {
    if (!have_snapshot) return fail(KQ_EINVAL, "kq_snapshot_patch before kq_snapshot_put");
    if (s->n_cq != prep.nq || s->n_cohort != prep.nc || s->n_flavor != prep.nF || s->n_resource != prep.nR)
      return fail(KQ_EINVAL, "kq_snapshot_patch: dictionary sizes differ from the uploaded snapshot");
    if (memcmp(s->parent, prep.h_parent.data(), (size_t)prep.N * sizeof(int32_t)) != 0) return fail(KQ_EINVAL, "kq_snapshot_patch: the cohort tree changed");
    const size_t Nfr = (size_t)prep.N * prep.nfr;
    for (auto& c : ring) c.live = false;   // commits folded into the old plane are part of the new one by construction
    commits = 0; last_cycle_n = -1;
    if (what & KQ_PATCH_ADMITTED) {
      // the admitted-row structures (candidate rank order, flavor-resource buckets, level orders, row records) are rebuilt on the
      // host and replace the resident ones; quota planes, the tree and the resource groups are not touched
      Prep np;
      np.want_fs = cfg.fair_sharing != 0;
      np.skip_rows = prep.skip_rows;
      int rc = build_prep(s, np);
      if (rc != KQ_OK) return fail(rc, np.err);
      prep = std::move(np);
      S.n_adm = prep.n_adm;
      reupload(S.cq_adm_off, s->cq_adm_off, (size_t)prep.nq + 1);
      reupload(S.adm_prio, s->adm_priority, (size_t)prep.n_adm); reupload(S.adm_qts, s->adm_queue_ts, (size_t)prep.n_adm);
      reupload(S.adm_flags, s->adm_flags, (size_t)prep.n_adm); reupload(S.adm_use_off, s->adm_use_off, (size_t)prep.n_adm + 1);
      reupload(S.adm_use_fr, s->adm_use_fr, (size_t)s->adm_use_off[prep.n_adm]); reupload(S.adm_use_qty, s->adm_use_qty, (size_t)s->adm_use_off[prep.n_adm]);
      reupload(S.adm_rts, s->adm_reserve_ts, (size_t)prep.n_adm); reupload(S.adm_uid, s->adm_uid_rank, (size_t)prep.n_adm);
      reupload(S.adm_cq, prep.adm_cq.data(), prep.adm_cq.size());
      h_adm_cq = prep.adm_cq; h_cq_adm_off.assign(s->cq_adm_off, s->cq_adm_off + prep.nq + 1);
      reupload(S.tree_row_off, prep.tree_row_off.data(), prep.tree_row_off.size()); reupload(S.tree_rows, prep.tree_rows.data(), prep.tree_rows.size());
      reupload(S.tree_rows_asc, prep.tree_rows_asc.data(), prep.tree_rows_asc.size());
      reupload(S.rank_pos, prep.rank_pos.data(), prep.rank_pos.size());
      reupload(S.frb_off, prep.frb_off.data(), prep.frb_off.size()); reupload(S.frb, prep.frb.data(), prep.frb.size());
      reupload(S.cq_row_bytes, prep.cq_row_bytes.data(), prep.cq_row_bytes.size());
      reupload(S.adm_rec, prep.adm_rec.data(), prep.adm_rec.size());
      reupload(S.adm_recx, prep.adm_recx.data(), prep.adm_recx.size());
      for (int l = 0; l < CS_LEVELS; l++) reupload(S.frl[l], prep.frl[l].data(), prep.frl[l].size());
      reupload(S.frbr, prep.frbr.data(), prep.frbr.size()); reupload(S.frec, prep.frec.data(), prep.frec.size());
      reupload(S.frb_sig, prep.frb_sig.data(), prep.frb_sig.size()); reupload(S.cs_ok, prep.cs_ok.data(), prep.cs_ok.size());
      upload_fs_rows(); reupload(S.fs_ok, prep.fs_ok.data(), prep.fs_ok.size());  // the quota-derived tables stay (like S.lendable)
      if (prep.skip_rows) { rc = be.sync(); if (rc != KQ_OK) return fail(rc, be.error()); rc = rows_rebuild(prep.n_adm); if (rc != KQ_OK) return rc; }
      what |= KQ_PATCH_USAGE;  // build_prep re-derived usage_consistent / fs_plain from s->usage: the plane must match
    } else if (what & KQ_PATCH_USAGE) {
      // usage-dependent flags of the prep: "cohort usage == what the children store in it" and the plain range of the amounts
      check_usage(s, prep);
    }
    if (what & KQ_PATCH_USAGE) { be.h2d(d_usage, s->usage, Nfr * sizeof(int64_t)); levels_stale = false; const int32_t zero = 0; be.h2d(d_big, &zero, sizeof(zero)); }
    int rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    return KQ_OK;
  }

  // ---- the admitted-row structures built on the device (kq_rows.hpp) ---------------------------------------------------------------------
  Buf rb[12];
  Buf rk2, rv2;        // second (key, value) pair of the sorts
  Buf rfs;             // resident fs_ok
  Buf rs[24];          // the rebuilt structures live in grow-only buffers (a hipMalloc / hipFree pair per array and call cost more than the sorts)
  Buf rt[2][10];       // the row table of kq_snapshot_patch_rows, double-buffered (the move reads the old table, writes the new one)
  int rt_cur = 0;
  // S.<field> now points into a persistent buffer: release the upload()ed array it pointed to before, if any
  template <class T> void adopt(const T*& field, T* fresh) {
    for (size_t i = 0; i < snap_allocs.size(); i++) if (snap_allocs[i] == (const void*)field) { be.free(snap_allocs[i]); snap_allocs.erase(snap_allocs.begin() + i); break; }
    field = fresh;
  }
  bool rows_device = getenv("KQ_ROWS_HOST") == nullptr;   // (A/B switch for kq_snapshot_put / kq_snapshot_patch: build_prep on the host; kq_snapshot_patch_rows has no host twin and returns KQ_EUNSUPPORTED then)
  std::vector<int32_t> h_adm_cq, h_cq_adm_off;            // host mirrors the row patch needs: ClusterQueue of every row, CSR offsets
  template <class T> void replace_dev(const T*& field, T* fresh) {
    for (size_t i = 0; i < snap_allocs.size(); i++) if (snap_allocs[i] == (const void*)field) { be.free(snap_allocs[i]); snap_allocs.erase(snap_allocs.begin() + i); break; }
    snap_allocs.push_back(fresh);
    field = fresh;
  }
  template <class T> T* dev_new(size_t n) { return (T*)be.alloc(std::max<size_t>(n, 1) * sizeof(T)); }
  bool rows_device_ok() const {
    return prep.n_adm < (1 << 20) && prep.N < (1 << 21) && (int64_t)prep.n_tree * prep.nfr < (1 << 22);
  }
  // Rebuilds every structure derived from the admitted rows from the resident row table (S.cq_adm_off, S.adm_*): the device twin of
  // build_prep's admitted part. n = rows, n_use = usage entries.
  int rows_rebuild(int n) {
    if (!rows_device_ok()) return fail(KQ_EUNSUPPORTED, "device-side row structures: sizes beyond the key layout");
    const int nq = prep.nq, nfr = prep.nfr, n_tree = prep.n_tree, N = prep.N;
    const size_t nb = (size_t)n_tree * nfr;
    DRows R{};
    R.n = n; R.nq = nq; R.nfr = nfr; R.n_tree = n_tree; R.N = N;
    R.cq_adm_off = S.cq_adm_off; R.adm_use_off = S.adm_use_off; R.adm_use_fr = S.adm_use_fr; R.adm_prio = S.adm_prio; R.adm_qts = S.adm_qts;
    R.adm_rts = S.adm_rts; R.adm_use_qty = S.adm_use_qty; R.adm_uid = S.adm_uid; R.adm_flags = S.adm_flags;
    R.tree_of = S.tree_of; R.depth = S.depth; R.parent = S.parent; R.cq_local = S.cq_local; R.node_local = S.node_local;
    // static part of the per-tree flags (build_prep): the tree's shape; the rows can only clear them
    std::vector<uint8_t> cs0(std::max(n_tree, 1), 1), rec0(std::max(n_tree, 1), 1);
    for (int c = 0; c < nq; c++) if (prep.depth[c] > CS_LEVELS) cs0[prep.tree_of[c]] = 0;
    uint8_t* d_cs = grow<uint8_t>(rs[9], n_tree); uint8_t* d_rec = grow<uint8_t>(rs[10], n_tree);
    be.h2d(d_cs, cs0.data(), std::max(n_tree, 1)); be.h2d(d_rec, rec0.data(), std::max(n_tree, 1));
    // fair sharing: the static part of fs_ok (tree shape, index ranges; all zero when fair sharing is off), the position offsets of every
    // ClusterQueue inside its tree (rows per ClusterQueue: O(ClusterQueues) on the host), the bitmap words of the largest tree
    const bool fair = cfg.fair_sharing != 0;
    std::vector<uint8_t> fs0(std::max(n_tree, 1), fair && prep.nfr <= 32767 ? 1 : 0);
    for (int c = 0; c < nq; c++) if (prep.depth[c] + 1 > FS_LV) fs0[prep.tree_of[c]] = 0;
    for (int t = 0; t < n_tree; t++) if (prep.tree_node_off[t + 1] - prep.tree_node_off[t] > 32767) fs0[t] = 0;
    uint8_t* d_fs = grow<uint8_t>(rfs, n_tree);
    be.h2d(d_fs, fs0.data(), std::max(n_tree, 1));
    R.cs_ok = d_cs; R.rec_ok = d_rec; R.fs_ok = d_fs;
    R.path = S.path; R.plen = S.plen; R.nR = prep.nR;
    // bits a sort key field really uses: a radix pass per 8 bits or so, so the tree / bucket / node fields are cut to their ranges
    auto bits_of = [](int64_t maxv) { int b = 1; while (b < 63 && ((int64_t)1 << b) <= maxv) b++; return b; };
    const int tree_bits = bits_of(std::max(n_tree - 1, 1));
    // two pairs of (key, value) buffers: the sorts ping-pong between them
    uint64_t* key2 = nullptr; int32_t* val2 = nullptr;
    auto kv = [&](size_t m) {
      R.key = (uint64_t*)grow<int64_t>(rb[1], m); R.val = grow<int32_t>(rb[2], m);
      key2 = (uint64_t*)grow<int64_t>(rk2, m); val2 = grow<int32_t>(rv2, m);
    };
    kv(n);
    R.ent_cnt = grow<int32_t>(rb[3], (size_t)n + 1); R.ent_off = grow<int32_t>(rb[4], (size_t)n + 1);
    R.tree_cnt = grow<int32_t>(rb[5], (size_t)n_tree + 1); R.bcnt = grow<int32_t>(rb[6], nb + 1); R.scal = grow<int32_t>(rb[7], 4);
    be.memset(R.ent_cnt, 0, ((size_t)n + 1) * 4); be.memset(R.scal, 0, 16);
    // rows per tree from the CSR offsets (host mirror), scanned here: O(ClusterQueues)
    std::vector<int32_t> tro((size_t)n_tree + 1, 0);
    for (int c = 0; c < nq; c++) tro[prep.tree_of[c] + 1] += h_cq_adm_off[c + 1] - h_cq_adm_off[c];
    for (int t = 0; t < n_tree; t++) tro[t + 1] += tro[t];
    R.adm_cq = grow<int32_t>(rs[0], n); R.tree_row_off = grow<int32_t>(rs[1], (size_t)n_tree + 1); R.tree_rows = grow<int32_t>(rs[2], n);
    R.tree_rows_asc = grow<int32_t>(rs[3], n); R.rank_pos = grow<int32_t>(rs[4], n); R.frb_off = grow<int32_t>(rs[5], nb + 1);
    R.cq_row_bytes = grow<int32_t>(rs[6], nq); R.adm_rec = grow<AdmRec>(rs[7], n); R.frb_sig = (uint64_t*)grow<int64_t>(rs[8], nb);
    R.adm_recx = grow<AdmRecX>(rs[20], n);
    be.memset(R.cq_row_bytes, 0, (size_t)std::max(nq, 1) * 4); be.memset(R.frb_sig, 0, std::max<size_t>(nb, 1) * 8);
    be.launch_rows(R, RO_ROW_INIT, n);
    be.sort_pairs(R.key, R.val, key2, val2, n, 32);
    be.launch_rows(R, RO_KEY_RTS, n); be.sort_pairs(R.key, R.val, key2, val2, n, 64);
    be.launch_rows(R, RO_KEY_PRIO, n); be.sort_pairs(R.key, R.val, key2, val2, n, 64);
    if (n_tree > 1) { be.launch_rows(R, RO_KEY_TREE, n); be.sort_pairs(R.key, R.val, key2, val2, n, tree_bits); }
    be.h2d(R.tree_row_off, tro.data(), tro.size() * 4);
    be.launch_rows(R, RO_RANK, n);
    // ascending rows per tree: the rows are in ascending order already, a stable sort by the tree alone groups them
    be.launch_rows(R, RO_KEY_ASC, n); if (n_tree > 1) be.sort_pairs(R.key, R.val, key2, val2, n, tree_bits);
    be.launch_rows(R, RO_ASC, n);
    be.scan_excl(R.ent_cnt, R.ent_off, n + 1);
    int32_t E = 0;
    be.d2h(&E, R.ent_off + n, 4);
    int rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    R.E = E;
    kv(std::max<size_t>((size_t)n, (size_t)CS_LEVELS * E));
    R.frb = grow<int32_t>(rs[11], E); R.frbr = grow<int32_t>(rs[12], E); R.frec = grow<CsRec>(rs[13], E);
    for (int l = 0; l < CS_LEVELS; l++) R.frl[l] = grow<CsEnt>(rs[14 + l], E);
    be.launch_rows(R, RO_ENT_FILL, n);
    be.sort_pairs(R.key, R.val, key2, val2, E, 32 + bits_of(std::max<int64_t>((int64_t)nb - 1, 1)));
    be.launch_rows(R, RO_BOUNDS, E + 1);
    be.launch_rows(R, RO_BUCKET_FILL, E);
    be.launch_rows(R, RO_BUCKET_SIZE, (int)nb);
    // the CS_LEVELS level orders of every bucket in one sort (the passes of three small sorts are launch-bound)
    be.launch_rows(R, RO_LKEY, CS_LEVELS * E); be.sort_pairs(R.key, R.val, key2, val2, CS_LEVELS * E, 46);
    be.launch_rows(R, RO_LFILL, CS_LEVELS * E);
    std::vector<int32_t> posoff;
    if (fair) {
      // position order: one more sort, by (tree, ClusterQueue inside the tree, evicted first, candidate rank)
      auto bits_of2 = [](int64_t maxv) { int b = 1; while (b < 63 && ((int64_t)1 << b) <= maxv) b++; return b; };
      R.cq_bits = bits_of2(std::max(prep.max_tree_cqs - 1, 1));
      R.fs_scan = grow<FsScan>(rs[17], n); R.fs_apply = grow<FsApply>(rs[18], n);
      be.launch_rows(R, RO_KEY_FS, n);
      be.sort_pairs(R.key, R.val, key2, val2, n, 32 + R.cq_bits + (n_tree > 1 ? tree_bits : 0));
      be.launch_rows(R, RO_FS_FILL, n);
      posoff.assign((size_t)nq + n_tree, 0);
      prep.max_tree_mw = 1;
      for (int t = 0; t < n_tree; t++) {
        const int q0 = prep.tree_cq_off[t], nqs = prep.tree_cq_off[t + 1] - q0;
        prep.max_tree_mw = std::max(prep.max_tree_mw, (tro[t + 1] - tro[t] + 63) / 64 + 1);
        int pos = 0;
        for (int i = 0; i < nqs; i++) { const int c = prep.tree_cqs[q0 + i]; posoff[(size_t)q0 + t + i] = pos; pos += h_cq_adm_off[c + 1] - h_cq_adm_off[c]; }
        posoff[(size_t)q0 + t + nqs] = pos;
      }
      int32_t* d_po = grow<int32_t>(rs[19], posoff.size());
      be.h2d(d_po, posoff.data(), posoff.size() * 4);
      adopt(S.fs_scan, R.fs_scan); adopt(S.fs_apply, R.fs_apply); adopt(S.fs_posoff, d_po);
    }
    int32_t scal[4];
    be.d2h(scal, R.scal, 16);
    prep.fs_ok.assign(std::max(n_tree, 1), 0); prep.rec_ok.assign(std::max(n_tree, 1), 0);   // host copies follow (upload_fs_quota re-uploads fs_ok)
    be.d2h(prep.fs_ok.data(), d_fs, std::max(n_tree, 1)); be.d2h(prep.rec_ok.data(), d_rec, std::max(n_tree, 1));
    rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    prep.n_adm = n; S.n_adm = n;
    prep.cs_max_bucket = scal[0];
    prep.max_tree_rows = 0;
    for (int t = 0; t < n_tree; t++) prep.max_tree_rows = std::max(prep.max_tree_rows, tro[t + 1] - tro[t]);
    prep.tree_row_off = tro;
    adopt(S.adm_cq, R.adm_cq); adopt(S.tree_row_off, R.tree_row_off); adopt(S.tree_rows, R.tree_rows);
    adopt(S.tree_rows_asc, R.tree_rows_asc); adopt(S.rank_pos, R.rank_pos); adopt(S.frb_off, R.frb_off); adopt(S.frb, R.frb);
    adopt(S.cq_row_bytes, R.cq_row_bytes); adopt(S.adm_rec, R.adm_rec); adopt(S.adm_recx, R.adm_recx); adopt(S.frbr, R.frbr); adopt(S.frec, R.frec);
    for (int l = 0; l < CS_LEVELS; l++) adopt(S.frl[l], R.frl[l]);
    adopt(S.frb_sig, R.frb_sig); adopt(S.cs_ok, d_cs); adopt(S.rec_ok, d_rec);
    adopt(S.fs_ok, d_fs);
    prep.fs_ok.resize(n_tree); prep.rec_ok.resize(n_tree);
    return KQ_OK;
  }
  // kq_snapshot_patch_rows (include/kq_engine.h): compaction + insertion of rows on the device, then rows_rebuild
  int snapshot_patch_rows(const kq_row_patch* p, int32_t* new_index) {
    if (!have_snapshot) return fail(KQ_EINVAL, "kq_snapshot_patch_rows before kq_snapshot_put");
    if (!p || p->n_remove < 0 || p->n_add < 0 || p->n_evict < 0) return fail(KQ_EINVAL, "bad kq_row_patch");
    if (p->n_evict > 0 && !p->evict_rows) return fail(KQ_EINVAL, "null evict_rows");
    for (int i = 0; i < p->n_evict; i++) if (p->evict_rows[i] < 0 || p->evict_rows[i] >= prep.n_adm) return fail(KQ_EINVAL, "evict_rows out of range");
    if (!rows_device || !rows_device_ok()) return fail(KQ_EUNSUPPORTED, "kq_snapshot_patch_rows: sizes beyond the device path (use kq_snapshot_patch)");
    if (steps_issued != steps_waited) return fail(KQ_EINVAL, "a kq_pending_step is in flight (kq_pending_step_wait first)");
    const int nq = prep.nq, n_old = prep.n_adm, n_rm = p->n_remove, n_add = p->n_add;
    if ((int)h_adm_cq.size() != n_old || (int)h_cq_adm_off.size() != nq + 1) return fail(KQ_EINVAL, "host mirrors of the row table are missing");
    if (n_rm > 0 && !p->remove_rows) return fail(KQ_EINVAL, "null remove_rows");
    if (n_add > 0 && (!p->add_cq || !p->add_priority || !p->add_queue_ts || !p->add_reserve_ts || !p->add_uid_rank || !p->add_flags || !p->add_use_off))
      return fail(KQ_EINVAL, "null array in kq_row_patch");
    std::vector<int32_t> rm(p->remove_rows, p->remove_rows + n_rm);
    std::sort(rm.begin(), rm.end());
    for (int i = 0; i < n_rm; i++) if (rm[i] < 0 || rm[i] >= n_old || (i > 0 && rm[i] == rm[i - 1])) return fail(KQ_EINVAL, "remove_rows: out of range or repeated");
    const int n_ause = n_add > 0 ? p->add_use_off[n_add] : 0;
    if (n_add > 0 && p->add_use_off[0] != 0) return fail(KQ_EINVAL, "add_use_off[0] must be 0");
    for (int i = 0; i < n_add; i++) {
      if (p->add_cq[i] < 0 || p->add_cq[i] >= nq) return fail(KQ_EINVAL, "add_cq out of range");
      if (p->add_use_off[i + 1] < p->add_use_off[i]) return fail(KQ_EINVAL, "add_use_off not monotone");
    }
    if (n_ause > 0 && (!p->add_use_fr || !p->add_use_qty)) return fail(KQ_EINVAL, "null usage entries in kq_row_patch");
    for (int e = 0; e < n_ause; e++) {
      if (p->add_use_fr[e] < 0 || p->add_use_fr[e] >= prep.nfr) return fail(KQ_EINVAL, "add_use_fr out of range");
      if (p->add_use_qty[e] < 0 || p->add_use_qty[e] >= ((int64_t)1 << 50)) return fail(KQ_EUNSUPPORTED, "amount outside the plain range (kq_snapshot_patch rebuilds the flags)");
    }
    // new CSR offsets, the removed rows per ClusterQueue, the target of every added row: O(nq + changes) on the host
    std::vector<int32_t> rm_off((size_t)nq + 1, 0), add_cnt(nq, 0), new_off((size_t)nq + 1, 0), target(std::max(n_add, 1));
    for (int r : rm) rm_off[h_adm_cq[r] + 1]++;
    for (int c = 0; c < nq; c++) rm_off[c + 1] += rm_off[c];
    for (int i = 0; i < n_add; i++) add_cnt[p->add_cq[i]]++;
    for (int c = 0; c < nq; c++) new_off[c + 1] = new_off[c] + (h_cq_adm_off[c + 1] - h_cq_adm_off[c]) - (rm_off[c + 1] - rm_off[c]) + add_cnt[c];
    {
      std::vector<int32_t> next(nq);
      for (int c = 0; c < nq; c++) next[c] = new_off[c + 1] - add_cnt[c];
      for (int i = 0; i < n_add; i++) target[i] = next[p->add_cq[i]]++;
    }
    const int n_new = new_off[nq];
    // the size guards look at the table AFTER the patch, before anything is moved (the old n_adm was tested)
    if (n_new >= (1 << 20)) return fail(KQ_EUNSUPPORTED, "kq_snapshot_patch_rows: the patched table leaves the sort keys' row field (use kq_snapshot_patch)");
    DRows R{};
    R.n_old = n_old; R.n_add = n_add; R.nq = nq;
    R.old_cq_off = S.cq_adm_off; R.o_adm_cq = S.adm_cq;
    Buf* T = rt[rt_cur ^ 1];   // the table the move writes; the resident one (rt[rt_cur], or the put's uploads) is only read
    int32_t* d_new_off = grow<int32_t>(T[0], (size_t)nq + 1);
    be.h2d(d_new_off, new_off.data(), ((size_t)nq + 1) * 4);
    R.new_cq_off = d_new_off;
    auto stage = [&](Buf& b, const void* host, size_t bytes) -> void* { void* d = grow<uint8_t>(b, bytes); if (bytes) be.h2d(d, host, bytes); return d; };
    R.rm_off = (const int32_t*)stage(rb[8], rm_off.data(), ((size_t)nq + 1) * 4);
    R.rm_rows = (const int32_t*)stage(rb[9], rm.data(), (size_t)n_rm * 4);
    // the added rows in one staged region
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
    const size_t o_t = carve((size_t)n_add * 4), o_p = carve((size_t)n_add * 8), o_q = carve((size_t)n_add * 8), o_r = carve((size_t)n_add * 8), o_u = carve((size_t)n_add * 4),
                 o_f = carve(n_add), o_uo = carve(((size_t)n_add + 1) * 4), o_uf = carve((size_t)n_ause * 4), o_uq = carve((size_t)n_ause * 8);
    std::vector<uint8_t> st(std::max<size_t>(off, 16), 0);
    if (n_add > 0) {
      memcpy(&st[o_t], target.data(), (size_t)n_add * 4); memcpy(&st[o_p], p->add_priority, (size_t)n_add * 8); memcpy(&st[o_q], p->add_queue_ts, (size_t)n_add * 8);
      memcpy(&st[o_r], p->add_reserve_ts, (size_t)n_add * 8); memcpy(&st[o_u], p->add_uid_rank, (size_t)n_add * 4); memcpy(&st[o_f], p->add_flags, n_add);
      memcpy(&st[o_uo], p->add_use_off, ((size_t)n_add + 1) * 4);
      if (n_ause > 0) { memcpy(&st[o_uf], p->add_use_fr, (size_t)n_ause * 4); memcpy(&st[o_uq], p->add_use_qty, (size_t)n_ause * 8); }
    }
    const uint8_t* d_st = (const uint8_t*)stage(rb[10], st.data(), st.size());
    R.a_target = (const int32_t*)(d_st + o_t); R.a_prio = (const int64_t*)(d_st + o_p); R.a_qts = (const int64_t*)(d_st + o_q); R.a_rts = (const int64_t*)(d_st + o_r);
    R.a_uid = (const uint32_t*)(d_st + o_u); R.a_flags = d_st + o_f; R.a_use_off = (const int32_t*)(d_st + o_uo); R.a_use_fr = (const int32_t*)(d_st + o_uf);
    R.a_use_qty = (const int64_t*)(d_st + o_uq);
    R.o_use_off = S.adm_use_off; R.o_use_fr = S.adm_use_fr; R.o_use_qty = S.adm_use_qty; R.o_prio = S.adm_prio; R.o_qts = S.adm_qts; R.o_rts = S.adm_rts;
    R.o_uid = S.adm_uid; R.o_flags = S.adm_flags;
    R.n_prio = grow<int64_t>(T[1], n_new); R.n_qts = grow<int64_t>(T[2], n_new); R.n_rts = grow<int64_t>(T[3], n_new); R.n_uid = grow<uint32_t>(T[4], n_new);
    R.n_flags = grow<uint8_t>(T[5], n_new); R.n_use_off = grow<int32_t>(T[6], (size_t)n_new + 1);
    R.n_ucnt = grow<int32_t>(rb[11], (size_t)n_new + 1);
    be.memset(R.n_ucnt, 0, ((size_t)n_new + 1) * 4);
    R.new_of_old = grow<int32_t>(rb[3], std::max(n_old, 1));   // (rb[3] = ent_cnt of the rebuild: free until then)
    // KQ_ROWS_FOLD_USAGE: the usage rows of what leaves and what arrives, packed from the OLD table / the staged rows before anything moves
    const bool fold = (p->flags & KQ_ROWS_FOLD_USAGE) != 0;
    if (p->flags & ~KQ_ROWS_FOLD_USAGE) return fail(KQ_EINVAL, "unknown kq_row_patch.flags");
    int32_t *f_cq = nullptr, *f_un = nullptr, *f_fr = nullptr; int64_t* f_qty = nullptr;
    if (fold && n_rm + n_add > 0) {
      const size_t nf = (size_t)n_rm + n_add;
      f_cq = grow<int32_t>(fb[0], nf); f_un = grow<int32_t>(fb[1], nf); f_fr = grow<int32_t>(fb[2], nf * KQ_MAXU); f_qty = grow<int64_t>(fb[3], nf * KQ_MAXU);
      int32_t* f_err = grow<int32_t>(fb[4], 4);
      be.memset(f_err, 0, 16);
      R.n_rm = n_rm; R.a_cq = (const int32_t*)stage(fb[5], p->add_cq, (size_t)n_add * 4);
      R.f_cq = f_cq; R.f_use_n = f_un; R.f_use_fr = f_fr; R.f_use_qty = f_qty; R.f_err = f_err;
      be.launch_rows(R, RO_FOLD_PACK, (int)nf);
      int32_t err = 0;
      be.d2h(&err, f_err, 4);
      int rc0 = be.sync();
      if (rc0 != KQ_OK) return fail(rc0, be.error());
      if (err) return fail(KQ_EUNSUPPORTED, "KQ_ROWS_FOLD_USAGE: a row of more than KQ_MAXU usage entries");
    }
    be.launch_rows(R, RO_MOVE_ROW, n_old);
    be.launch_rows(R, RO_ADD_ROW, n_add);
    if (p->n_evict > 0) {   // (two marks of one row write the same byte with the same bit: no ordering needed)
      R.ev_rows = (const int32_t*)stage(rb[0], p->evict_rows, (size_t)p->n_evict * 4);
      be.launch_rows(R, RO_EVICT, p->n_evict);
    }
    be.scan_excl(R.n_ucnt, R.n_use_off, n_new + 1);
    int32_t U = 0;
    be.d2h(&U, R.n_use_off + n_new, 4);
    std::vector<int32_t> noo(std::max(n_old, 1));
    be.d2h(noo.data(), R.new_of_old, (size_t)n_old * 4);
    int rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    R.n_use_fr = grow<int32_t>(T[7], U); R.n_use_qty = grow<int64_t>(T[8], U);
    be.launch_rows(R, RO_MOVE_ENT, n_old + n_add);
    rc = be.sync();   // (the staging vectors go out of scope below)
    if (rc != KQ_OK) return fail(rc, be.error());
    // the new row table becomes the resident one
    adopt(S.cq_adm_off, d_new_off); adopt(S.adm_prio, R.n_prio); adopt(S.adm_qts, R.n_qts); adopt(S.adm_rts, R.n_rts);
    adopt(S.adm_uid, R.n_uid); adopt(S.adm_flags, R.n_flags); adopt(S.adm_use_off, R.n_use_off);
    adopt(S.adm_use_fr, R.n_use_fr); adopt(S.adm_use_qty, R.n_use_qty);
    rt_cur ^= 1;
    // host mirrors
    std::vector<int32_t> ncq(std::max(n_new, 1));
    for (int c = 0; c < nq; c++) for (int r = new_off[c]; r < new_off[c + 1]; r++) ncq[r] = c;
    ncq.resize(n_new);
    h_adm_cq.swap(ncq); h_cq_adm_off = new_off;
    if (new_index) for (int r = 0; r < n_old; r++) new_index[r] = noo[r];
    for (auto& c : ring) (void)c;   // (committed usage rows are keyed by ClusterQueue, not by admitted row: they stay valid)
    last_cycle_n = -1;              // the last cycle's argument block names freed arrays
    // resident head batches that name admitted rows (slice_row) name the OLD indices: void them (kq_heads_put again)
    for (size_t b = 0; b < batches.size(); b++) if ((int)b != PEND_SLOT && batches[b].valid && batches[b].H.slice_row) batches[b].valid = false;
    // the resident pending set follows the move: the slices its workloads replace keep their identity, a removed one is simply gone
    // (the head is then an ordinary workload: ReplacedWorkloadSlice finds nothing, workloadslicing.go:371)
    if (pend.valid && pend.D.P.slice_row && pend.W > 0) { R.remap = const_cast<int32_t*>(pend.D.P.slice_row); be.launch_rows(R, RO_REMAP, pend.W); }
    rc = rows_rebuild(n_new);
    // not atomic on failure: the new row table is resident but the structures derived from it are not — no cycle may run on that.
    // The caller re-puts (kq_snapshot_put); every entry point checks have_snapshot.
    if (rc != KQ_OK) { have_snapshot = false; return rc; }
    if (fold) {
      // removeUsage of what left, addUsage of what arrived (resource_node.go:144-165) through the commit's own cell functions; quota that was
      // freed sends the inadmissible workloads of those root cohorts back to their heaps, as a release does
      if (n_rm > 0) {
        DCommit dc{n_rm, f_cq, f_un, f_fr, f_qty, d_usage, d_big};
        apply_commit(dc, false);
        if (pend.valid && pend.nq == prep.nq) be.launch_pend_release(pend.D, S, pend.d_tree_stamp, dc.cq, dc.use_n, n_rm, ++pend.release_seq);
      }
      if (n_add > 0) {
        DCommit dc{n_add, f_cq + n_rm, f_un + n_rm, f_fr + (size_t)n_rm * KQ_MAXU, f_qty + (size_t)n_rm * KQ_MAXU, d_usage, d_big};
        apply_commit(dc, true);
      }
      rc = be.sync();
      if (rc != KQ_OK) return fail(rc, be.error());
    }
    return KQ_OK;
  }
  Buf fb[6];   // KQ_ROWS_FOLD_USAGE: the packed usage rows
  // test hook: the resident admitted-row structures, one by one (which: 0 adm_cq, 1 tree_row_off, 2 tree_rows, 3 tree_rows_asc, 4 rank_pos,
  // 5 frb_off, 6 frb, 7 frbr, 8 cq_row_bytes, 9 adm_rec, 10 frec, 11-13 frl, 14 frb_sig, 15 cs_ok, 16 rec_ok, 17 cq_adm_off, 18 adm_use_off,
  // 19 adm_use_fr, 20 adm_use_qty, 21 adm_prio, 22 adm_qts, 23 adm_rts, 24 adm_uid, 25 adm_flags). *bytes: capacity in, size out.
  int read_rows(int which, void* out, int64_t* bytes) {
    if (!have_snapshot) return fail(KQ_EINVAL, "no snapshot");
    const size_t n = prep.n_adm, nb = (size_t)prep.n_tree * prep.nfr;
    int32_t E = 0, U = 0;
    if (nb > 0) be.d2h(&E, S.frb_off + nb, 4);
    if (n > 0) be.d2h(&U, S.adm_use_off + n, 4);
    int rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    const void* src = nullptr; size_t sz = 0;
    switch (which) {
      case 0: src = S.adm_cq; sz = n * 4; break;
      case 1: src = S.tree_row_off; sz = ((size_t)prep.n_tree + 1) * 4; break;
      case 2: src = S.tree_rows; sz = n * 4; break;
      case 3: src = S.tree_rows_asc; sz = n * 4; break;
      case 4: src = S.rank_pos; sz = n * 4; break;
      case 5: src = S.frb_off; sz = (nb + 1) * 4; break;
      case 6: src = S.frb; sz = (size_t)E * 4; break;
      case 7: src = S.frbr; sz = (size_t)E * 4; break;
      case 8: src = S.cq_row_bytes; sz = (size_t)prep.nq * 4; break;
      case 9: src = S.adm_rec; sz = n * sizeof(AdmRec); break;
      case 10: src = S.frec; sz = (size_t)E * sizeof(CsRec); break;
      case 11: case 12: case 13: src = S.frl[which - 11]; sz = (size_t)E * sizeof(CsEnt); break;
      case 14: src = S.frb_sig; sz = nb * 8; break;
      case 15: src = S.cs_ok; sz = prep.n_tree; break;
      case 16: src = S.rec_ok; sz = prep.n_tree; break;
      case 17: src = S.cq_adm_off; sz = ((size_t)prep.nq + 1) * 4; break;
      case 18: src = S.adm_use_off; sz = (n + 1) * 4; break;
      case 19: src = S.adm_use_fr; sz = (size_t)U * 4; break;
      case 20: src = S.adm_use_qty; sz = (size_t)U * 8; break;
      case 21: src = S.adm_prio; sz = n * 8; break;
      case 22: src = S.adm_qts; sz = n * 8; break;
      case 23: src = S.adm_rts; sz = n * 8; break;
      case 24: src = S.adm_uid; sz = n * 4; break;
      case 25: src = S.adm_flags; sz = n; break;
      case 26: src = S.fs_ok; sz = prep.n_tree; break;
      case 27: src = S.fs_posoff; sz = cfg.fair_sharing ? ((size_t)prep.nq + prep.n_tree) * 4 : 0; break;
      case 28: src = S.fs_scan; sz = cfg.fair_sharing ? n * sizeof(FsScan) : 0; break;
      case 29: src = S.fs_apply; sz = cfg.fair_sharing ? n * sizeof(FsApply) : 0; break;
      case 30: src = S.adm_recx; sz = n * sizeof(AdmRecX); break;
      default: return fail(KQ_EINVAL, "unknown structure");
    }
    if ((int64_t)sz > *bytes) { *bytes = (int64_t)sz; return fail(KQ_ECAPACITY, "buffer too small"); }
    *bytes = (int64_t)sz;
    if (sz) be.d2h(out, src, sz);
    rc = be.sync();
    return rc == KQ_OK ? KQ_OK : fail(rc, be.error());
  }
  int debug_rows_rebuild() {
    if (!have_snapshot) return fail(KQ_EINVAL, "no snapshot");
    return rows_rebuild(prep.n_adm);
  }

  // ---- closed loop: commit the last cycle's admissions into the snapshot, release them later --------------
  struct Committed { Buf cq, use_n, use_fr, use_qty; int n = 0; bool live = false; };
  Committed ring[KQ_COMMIT_RING];
  int64_t commits = 0;
  int last_cycle_n = -1;   // heads of the last executed cycle, -1 = none / already committed / failed
  int last_slot = -1;      // head batch of that cycle (kq_cycle_commit reads its cq array on the device)
  DOut last_O{};
  int max_depth() const { int m = 0; for (int n = 0; n < prep.N; n++) m = std::max(m, (int)prep.depth[n]); return m; }
  // Folding usage rows into the resident snapshot. When the snapshot satisfies "cohort usage = sum of what the children store
  // in it" only the ClusterQueue cells are touched here; the cohort levels are re-derived from them once, right before the
  // next reader (flush_levels) — a commit and a release between two cycles cost one re-derivation, not two.
  bool levels_stale = false;
  void apply_commit(const DCommit& dc, bool add) {
    if (prep.usage_consistent) { be.launch_commit_cells(S, dc, add); levels_stale = true; }
    else be.launch_commit_trees(S, dc, add);
  }
  void flush_levels() {
    if (levels_stale) be.launch_usage_levels(S, d_usage, max_depth());
    levels_stale = false;
  }
  int cycle_commit(int32_t* n_admitted) {
    if (!have_snapshot) return fail(KQ_EINVAL, "kq_cycle_commit before kq_snapshot_put");
    if (last_cycle_n < 0) return fail(KQ_EINVAL, "kq_cycle_commit: no uncommitted cycle");
    const int n = last_cycle_n;
    Committed& c = ring[commits % KQ_COMMIT_RING];
    if (c.live) return fail(KQ_ECAPACITY, "commit ring full: release older commits first");
    c.n = n;
    int32_t count = 0;
    if (n > 0) {
      int32_t* d_cq = grow<int32_t>(c.cq, n);
      int32_t* d_un = grow<int32_t>(c.use_n, n);
      int32_t* d_fr = grow<int32_t>(c.use_fr, (size_t)n * KQ_MAXU);
      int64_t* d_qty = grow<int64_t>(c.use_qty, (size_t)n * KQ_MAXU);
      int32_t* d_count = (int32_t*)grow<int64_t>(b_misc, 2);
      if (n_admitted) be.memset(d_count, 0, 16);  // the count is only read back on request
      be.launch_commit_mask(n, d_un, d_cq, d_fr, d_qty, d_count);  // also keeps the cycle's usage rows for the release
      DCommit dc{n, d_cq, d_un, d_fr, d_qty, d_usage, d_big};
      apply_commit(dc, true);
      if (n_admitted) {  // only a caller that asks for the count pays for a round trip; the stream orders the rest
        be.d2h(&count, d_count, sizeof(count));
        int rc = be.sync();
        if (rc != KQ_OK) return fail(rc, be.error());
      }
    }
    c.live = true;
    commits++;
    last_cycle_n = -1;
    if (n_admitted) *n_admitted = count;
    return KQ_OK;
  }
  int cycle_release(int age) {
    if (age < 1 || age > KQ_COMMIT_RING || age > commits) return fail(KQ_EINVAL, "kq_cycle_release: no such commit");
    Committed& c = ring[(commits - age) % KQ_COMMIT_RING];
    if (!c.live) return fail(KQ_EINVAL, "kq_cycle_release: already released");
    if (c.n > 0) {
      DCommit dc{c.n, (const int32_t*)c.cq.p, (const int32_t*)c.use_n.p, (const int32_t*)c.use_fr.p, (const int64_t*)c.use_qty.p, d_usage, d_big};
      apply_commit(dc, false);
      // the finished workloads freed quota: inadmissible workloads of their root cohorts go back to the heaps
      if (pend.valid && pend.nq == prep.nq) be.launch_pend_release(pend.D, S, pend.d_tree_stamp, dc.cq, dc.use_n, c.n, ++pend.release_seq);
    }
    c.live = false;
    return KQ_OK;
  }

  // kq_snapshot_derive: SubtreeQuota / cohort Usage / flags recomputed on the device from the uploaded Quotas and
  // ClusterQueue usage; the fair-sharing constants that depend on them are rebuilt from the result.
  int snapshot_derive() {
    if (!have_snapshot) return fail(KQ_EINVAL, "kq_snapshot_derive before kq_snapshot_put");
    int max_depth = 0;
    for (int n = 0; n < prep.N; n++) max_depth = std::max(max_depth, (int)prep.depth[n]);
    DDerive d{d_sq, d_usage, d_qflags};
    be.launch_derive(S, d, max_depth); levels_stale = false;  // cohort usage is recomputed from the ClusterQueue cells here
    const size_t cells = (size_t)prep.N * prep.nfr;
    std::vector<int64_t> sq(cells), us(cells);
    std::vector<uint8_t> fl(cells);
    be.d2h(sq.data(), d_sq, cells * 8); be.d2h(us.data(), d_usage, cells * 8); be.d2h(fl.data(), d_qflags, cells);
    int rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    build_fair(prep, sq.data(), us.data(), fl.data());
    S.lendable = upload(prep.lendable.data(), prep.lendable.size());
    S.frcount = upload(prep.frcount.data(), prep.frcount.size());
    upload_fs_quota();
    rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    return KQ_OK;
  }
  int read_planes(int64_t* sq, int64_t* us, uint8_t* fl) {
    if (!have_snapshot) return fail(KQ_EINVAL, "no snapshot");
    const size_t cells = (size_t)prep.N * prep.nfr;
    flush_levels();
    if (sq) be.d2h(sq, d_sq, cells * 8);
    if (us) be.d2h(us, d_usage, cells * 8);
    if (fl) be.d2h(fl, d_qflags, cells);
    return be.sync();
  }

  bool sim_ahead_on = getenv("KQ_SIM_AHEAD_OFF") == nullptr;   // (A/B switch: the full nominate pass runs every SimulatePreemption itself)
  Buf b_sim[4];
  bool lean_only_ok = getenv("KQ_FULL_PASS_ALWAYS") == nullptr;   // (A/B switch: launch the full nominate pass in every cycle)
  bool vh_partial = false;   // validate_heads: some podset of the batch may be admitted partially (minCount < count)
  int validate_heads(const kq_heads* h, int* slot_cap, bool* plain, int* max_nps = nullptr) {
    if (h->n < 0) return fail(KQ_EINVAL, "negative head count");
    int cap = 1;
    int mnps = 1;
    *plain = true;
    vh_partial = false;
    for (int i = 0; i < h->n; i++) {
      if (h->cq[i] < 0 || h->cq[i] >= prep.nq) return fail(KQ_EINVAL, "head cq out of range");
      int nps = h->ps_off[i + 1] - h->ps_off[i];
      if (nps < 0) return fail(KQ_EINVAL, "ps_off not monotone");
      if (nps > KQ_MAXPS) return fail(KQ_EUNSUPPORTED, "more podsets than KQ_MAXPS");
      mnps = std::max(mnps, nps);
      int slots = 0;
      for (int p = h->ps_off[i]; p < h->ps_off[i + 1]; p++) {
        int nreq = h->ps_req_off[p + 1] - h->ps_req_off[p];
        if (nreq < 0) return fail(KQ_EINVAL, "ps_req_off not monotone");
        if (h->ps_min_count && h->ps_min_count[p] >= 0 && h->ps_count[p] > h->ps_min_count[p]) vh_partial = true;
        if (nreq + 1 > KQ_MAXREQ) return fail(KQ_EUNSUPPORTED, "more resources per podset than KQ_MAXREQ");
        for (int e = h->ps_req_off[p]; e < h->ps_req_off[p + 1]; e++)
        {
          if (h->req_res[e] < 0 || h->req_res[e] >= prep.nR) return fail(KQ_EINVAL, "req_res out of range");
          if (h->req_qty[e] < 0 || h->req_qty[e] >= ((int64_t)1 << 40)) *plain = false;
        }
        slots += nreq + 1;
      }
      if (slots > KQ_MAXU) return fail(KQ_EUNSUPPORTED, "more usage entries than KQ_MAXU");
      cap = std::max(cap, slots);
    }
    if (h->slice_row)
      for (int i = 0; i < h->n; i++) {
        const int row = h->slice_row[i];
        if (row < -1 || row >= prep.n_adm) return fail(KQ_EINVAL, "slice_row out of range");
        if (row >= 0 && h_adm_cq[row] != h->cq[i]) return fail(KQ_EINVAL, "slice_row: the replaced slice is not admitted in the head's ClusterQueue");
      }
    // PodSetGroupName groups (kq_heads.ps_group): consecutive podsets of their head (the reference appends a group's podsets together,
    // orderedgroups.InOrder, so an interleaved group would reorder Assignment.PodSets), their summed requests within KQ_MAXREQ, and no
    // group of several podsets on a head that replaces a workload slice
    if (h->ps_group)
      for (int i = 0; i < h->n; i++)
        for (int p = h->ps_off[i]; p < h->ps_off[i + 1]; p++) {
          const int gid = h->ps_group[p];
          if (gid < 0 || (p > h->ps_off[i] && h->ps_group[p - 1] == gid)) continue;   // not in a group / not the first member
          int last = p;
          std::vector<int32_t> res;
          for (int q = p; q < h->ps_off[i + 1]; q++) {
            if (h->ps_group[q] != gid) continue;
            if (q > last + 1) return fail(KQ_EUNSUPPORTED, "the podsets of a PodSetGroupName group must be consecutive");
            last = q;
            for (int e = h->ps_req_off[q]; e < h->ps_req_off[q + 1]; e++) if (std::find(res.begin(), res.end(), h->req_res[e]) == res.end()) res.push_back(h->req_res[e]);
          }
          if (last > p && (int)res.size() + 1 > KQ_MAXREQ) return fail(KQ_EUNSUPPORTED, "more resources in a podset group than KQ_MAXREQ");
          if (last > p && h->slice_row && h->slice_row[i] >= 0) return fail(KQ_EUNSUPPORTED, "a workload slice with a podset group of several podsets");
        }
    *slot_cap = cap;
    if (max_nps) *max_nps = mnps;
    return KQ_OK;
  }
  // the pending side (kq_pending_*) keeps no PodSetGroupName column
  bool heads_grouped(const kq_heads* h) const {
    if (!h->ps_group) return false;
    const size_t nps = h->n > 0 ? (size_t)h->ps_off[h->n] : 0;
    for (size_t p = 0; p < nps; p++) if (h->ps_group[p] >= 0) return true;
    return false;
  }

  // Make one batch of heads resident in HBM. slot 0 is the transient batch of kq_cycle_run.
  int heads_put(const kq_heads* h, int slot) {
    if (!have_snapshot) return fail(KQ_EINVAL, "heads before kq_snapshot_put");
    if (slot < 0 || slot > 4096) return fail(KQ_EINVAL, "bad batch id");
    int slot_cap = 1, max_nps = 1;
    bool plain = true;
    int rc = validate_heads(h, &slot_cap, &plain, &max_nps);
    if (rc != KQ_OK) return rc;
    if ((int)batches.size() <= slot) batches.resize(slot + 1);
    HeadBatch& hbch = batches[slot];
    hbch.max_nps = max_nps;
    if (slot == last_slot) last_cycle_n = -1;  // the uncommitted cycle's head arrays are about to be replaced (or freed)
    const int n = h->n;
    hbch.n = n; hbch.slot_cap = slot_cap; hbch.cycle = h->cycle; hbch.valid = true; hbch.plain = plain; hbch.partial = vh_partial;
    hbch.nps = n ? h->ps_off[n] : 0;
    if (n == 0) return KQ_OK;
    const size_t nps = hbch.nps, nreqs = h->ps_req_off[nps], nR = prep.nR, nfw = (prep.nF + 63) / 64;
    // The sixteen arrays of a batch travel as ONE packed region (16-byte aligned pieces) through a pinned staging buffer: the
    // PCIe-inclusive entry point kq_cycle_run pays one copy per cycle instead of sixteen.
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
    const size_t o_cq = place(n * 4), o_prio = place(n * 8), o_ts = place(n * 8), o_flags = place(n * 4), o_psoff = place((n + 1) * 4);
    const size_t o_pscnt = place(nps * 4), o_psmin = place(nps * 4), o_reqoff = place((nps + 1) * 4), o_rres = place(nreqs * 4), o_rqty = place(nreqs * 8);
    const size_t o_fok = place(nps * nfw * 8), o_ltried = place(nps * nR * 4), o_lgen = place(n * 8), o_lcyc = place(n * 8), o_lhash = place(n * 8), o_hash = place(n * 8);
    const bool sl = h->slice_row != nullptr;
    const size_t o_srow = place(sl ? n * 4 : 0), o_scnt = place(sl ? nps * 4 : 0), o_sfl = place(sl ? nreqs * 4 : 0), o_sq = place(sl ? nreqs * 8 : 0),
                 o_spf = place(sl ? nps * 4 : 0), o_spq = place(sl ? nps * 8 : 0);
    const bool grp = heads_grouped(h);
    const size_t o_grp = place(grp ? nps * 4 : 0);
    const size_t total = off;
    if (hup_cap < total) { if (hup) be.free_host(hup); hup_cap = total + total / 4; hup = (uint8_t*)be.alloc_host(hup_cap); }
    if (!hup) { hup_cap = 0; return fail(KQ_EDEVICE, "pinned staging buffer for the heads could not be allocated"); }
    auto put = [&](size_t o, const void* src, size_t bytes) { if (bytes) memcpy(hup + o, src, bytes); };
    auto fill = [&](size_t o, int byte, size_t bytes) { if (bytes) memset(hup + o, byte, bytes); };
    put(o_cq, h->cq, n * 4); put(o_prio, h->priority, n * 8); put(o_ts, h->queue_ts, n * 8); put(o_flags, h->flags, n * 4);
    put(o_psoff, h->ps_off, (n + 1) * 4); put(o_pscnt, h->ps_count, nps * 4);
    if (h->ps_min_count) put(o_psmin, h->ps_min_count, nps * 4); else fill(o_psmin, 0xff, nps * 4);          // -1: no MinimumCount
    put(o_reqoff, h->ps_req_off, (nps + 1) * 4); put(o_rres, h->req_res, nreqs * 4); put(o_rqty, h->req_qty, nreqs * 8);
    put(o_fok, h->ps_flavor_ok, nps * nfw * 8);
    if (h->ps_last_tried) put(o_ltried, h->ps_last_tried, nps * nR * 4); else fill(o_ltried, 0xff, nps * nR * 4);  // -1: nothing tried
    if (h->last_generation) put(o_lgen, h->last_generation, n * 8); else fill(o_lgen, 0, n * 8);
    if (h->last_cycle) put(o_lcyc, h->last_cycle, n * 8); else fill(o_lcyc, 0, n * 8);
    if (h->last_hash) put(o_lhash, h->last_hash, n * 8); else fill(o_lhash, 0, n * 8);
    if (h->hash) put(o_hash, h->hash, n * 8); else fill(o_hash, 0, n * 8);
    if (sl) {  // workload slices: absent columns read as "no flavor / nothing requested"
      put(o_srow, h->slice_row, n * 4);
      if (h->ps_slice_count) put(o_scnt, h->ps_slice_count, nps * 4); else fill(o_scnt, 0, nps * 4);
      if (h->req_slice_flavor) put(o_sfl, h->req_slice_flavor, nreqs * 4); else fill(o_sfl, 0xff, nreqs * 4);
      if (h->req_slice_qty) put(o_sq, h->req_slice_qty, nreqs * 8); else fill(o_sq, 0, nreqs * 8);
      if (h->ps_slice_pods_flavor) put(o_spf, h->ps_slice_pods_flavor, nps * 4); else fill(o_spf, 0xff, nps * 4);
      if (h->ps_slice_pods_qty) put(o_spq, h->ps_slice_pods_qty, nps * 8); else fill(o_spq, 0, nps * 8);
    }
    if (grp) put(o_grp, h->ps_group, nps * 4);
    uint8_t* d = grow<uint8_t>(hbch.hb[0], total);
    be.h2d(d, hup, total);
    DHeads& H = hbch.H;
    H.n = n;
    H.ps_group = grp ? (const int32_t*)(d + o_grp) : nullptr;
    H.cq = (const int32_t*)(d + o_cq); H.priority = (const int64_t*)(d + o_prio); H.queue_ts = (const int64_t*)(d + o_ts);
    H.flags = (const uint32_t*)(d + o_flags); H.ps_off = (const int32_t*)(d + o_psoff); H.ps_count = (const int32_t*)(d + o_pscnt);
    H.ps_min_count = (const int32_t*)(d + o_psmin); H.ps_req_off = (const int32_t*)(d + o_reqoff); H.req_res = (const int32_t*)(d + o_rres);
    H.req_qty = (const int64_t*)(d + o_rqty); H.ps_flavor_ok = (const uint64_t*)(d + o_fok); H.ps_last_tried = (const int32_t*)(d + o_ltried);
    H.last_generation = (const int64_t*)(d + o_lgen); H.last_cycle = (const int64_t*)(d + o_lcyc);
    H.last_hash = (const uint64_t*)(d + o_lhash); H.hash = (const uint64_t*)(d + o_hash);
    H.slice_row = sl ? (const int32_t*)(d + o_srow) : nullptr; H.ps_slice_count = sl ? (const int32_t*)(d + o_scnt) : nullptr;
    H.req_slice_flavor = sl ? (const int32_t*)(d + o_sfl) : nullptr; H.req_slice_qty = sl ? (const int64_t*)(d + o_sq) : nullptr;
    H.ps_slice_pods_flavor = sl ? (const int32_t*)(d + o_spf) : nullptr; H.ps_slice_pods_qty = sl ? (const int64_t*)(d + o_spq) : nullptr;
    rc = be.sync();  // the staging buffer is reused by the next call
    if (rc != KQ_OK) return fail(rc, be.error());
    return KQ_OK;
  }

  int cycle_shard_words(const kq_heads* h, const kq_decisions* out, int world, int64_t* words) {
    int rc = heads_put(h, 0);
    if (rc != KQ_OK) return rc;
    if (world < 1) return fail(KQ_EINVAL, "world < 1");
    *words = (int64_t)shard_words_for(0, out, world);
    return KQ_OK;
  }
  int cycle_nominate_shard(const kq_heads* h, const uint8_t* mine, int world, int rank, void* xbuf, kq_decisions* out) {
    if (world < 1 || rank < 0 || rank >= world || !xbuf) return fail(KQ_EINVAL, "bad shard arguments");
    int rc = heads_put(h, 0);
    if (rc != KQ_OK) return rc;
    shard_heads_ok = true;
    ShardCall sc; sc.mode = 1; sc.mine = mine; sc.x = (int64_t*)xbuf; sc.world = world; sc.rank = rank;
    return cycle_exec(0, out, false, sc);
  }
  int cycle_process_merged(int world, int rank, const void* xbuf, kq_decisions* out) {
    if (!shard_heads_ok || batches.empty() || !batches[0].valid) return fail(KQ_EINVAL, "kq_cycle_process_merged without kq_cycle_nominate_shard");
    ShardCall sc; sc.mode = 2; sc.x = (int64_t*)const_cast<void*>(xbuf); sc.world = world; sc.rank = rank;
    return cycle_exec(0, out, false, sc);
  }
  bool shard_heads_ok = false;
  int cycle_run(const kq_heads* h, kq_decisions* out) {
    shard_heads_ok = false;
    int rc = heads_put(h, 0);
    if (rc != KQ_OK) return rc;
    return cycle_exec(0, out);
  }

  // ---- Topology-Aware Scheduling inside the cycle (include/kq_cycle_tas.h, kq_tas_cycle.hpp) ------------------------------------------------
  std::vector<Buf> tbuf;
  size_t tnext = 0;
  bool tas_classes_off = getenv("KQ_TAS_CLASSES_OFF") != nullptr;   // (A/B switch and tests: every placement of k_process_tas runs its own phase 1)
  int tas_n_cls = 0;       // (TAS flavor, request class) pairs of the cycle being enqueued
  size_t tas_lds_want = 0; // LDS a class-path placement's working state takes on the largest TAS flavor (kq_tas_device.hpp TLds)
  template <class T> T* tgrow(size_t n) { if (tnext >= tbuf.size()) tbuf.resize(tnext + 16); return grow<T>(tbuf[tnext++], n); }
  template <class T> T* tstage(const T* host, size_t n) { T* d = tgrow<T>(n); if (n) be.h2d(d, host, n * sizeof(T)); return d; }
  int cycle_run_tas(const kq_heads* h, const kq_cycle_tas* t, kq_decisions* out, kq_cycle_tas_out* tout, int64_t* stats) {
    if (!have_snapshot) return fail(KQ_EINVAL, "kq_cycle_run_tas before kq_snapshot_put");
    if (!t || !tout || !tout->ps_tas || !tout->dom_off) return fail(KQ_EINVAL, "null kq_cycle_tas / kq_cycle_tas_out");
    if (t->n_tas < 0) return fail(KQ_EINVAL, "negative n_tas");
    shard_heads_ok = false;
    kq_heads hg = *h;   // the flavor scan's podset groups: the heads' own column, else the TAS side's (one PodSetGroupName, two consumers)
    if (!hg.ps_group) hg.ps_group = t->ps_group;
    int rc = heads_put(&hg, 0);
    if (rc != KQ_OK) return rc;
    const int n = batches[0].n;
    const size_t nps = batches[0].nps;
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    tout->dom_off[0] = 0;
    // no TAS flavor in the snapshot and nothing asks for one: the ordinary cycle. (A podset that asks for TAS without any TAS flavor to
    // land on still goes through WorkloadsTopologyRequests: ErrNoTASFlavorAssigned -> NoFit, tas_flavorassigner.go:60-66.)
    bool any_request = false;
    for (size_t p = 0; p < nps && t->ps_flags; p++) if (t->ps_flags[p] & KQ_PS_TAS_EXPLICIT) any_request = true;
    for (int c = 0; c < prep.nq && t->cq_tas_only; c++) if (t->cq_tas_only[c]) any_request = true;
    if (t->n_tas == 0 && !any_request) {
      rc = cycle_exec(0, out);
      for (size_t p = 0; p < nps; p++) { tout->ps_tas[p] = -1; tout->dom_off[p + 1] = 0; }
      return rc;
    }
    const int nt = t->n_tas, R = nt > 0 ? t->topo[0].n_resources : 1;
    if (R < 1 || R > KQ_TAS_MAXR) return fail(KQ_EUNSUPPORTED, "n_resources out of range");
    if ((nt > 0 && (!t->tas_flavor || !t->topo)) || !t->cq_tas_only || !t->adm_off || !t->ps_flags || !t->ps_kind || !t->ps_level || !t->ps_slice_size ||
        !t->ps_slice_level || !t->ps_group || !t->ps_req) return fail(KQ_EINVAL, "null array in kq_cycle_tas");
    tnext = 0;
    // TASBalancedPlacement (tas_balanced_placement.go): a preferred request's dynamic programme needs a table per wave slot (kq_tas_device.hpp
    // TBal) and the full domain state (no request-class tables, no LDS copy of them): fewer slots, classes off for the cycle
    bool balanced = false;
    for (int i = 0; i < t->n_tas; i++) if (t->topo && (t->topo[i].profile_mixed & KQ_TAS_F_BALANCED_PLACEMENT)) balanced = true;
    const int slots = std::max(1, std::min(n, balanced ? std::min(be.max_slots(), 64) : be.max_slots()));
    be.tas_bal = balanced;   // (the kernels that carry the gate's code)
    // request classes of the cycle's podsets (k_process_tas keeps their phase-1 tables resident, kq_tas_cycle.hpp): same per-pod requests,
    // slice size and slice level on every TAS flavor; podset groups and inner layers stay outside
    constexpr int TC_MAXCLS = 32;
    std::vector<int32_t> ps_class(std::max<size_t>(nps, 1), -1), cls_rep;
    if (!tas_classes_off && !balanced) {
      std::unordered_map<std::string, int> ids;
      std::vector<uint8_t> repl_head(std::max<size_t>(nps, 1), 0);   // podsets of a head that looks for a failed node's replacement: rewritten requests
      if (t->ps_adm_flavor) for (int i = 0; i < n; i++) if (h->flags[i] & KQ_HEAD_HAS_UNHEALTHY_NODES) for (int p = h->ps_off[i]; p < h->ps_off[i + 1]; p++) repl_head[p] = 1;
      for (size_t p = 0; p < nps; p++) {
        if (t->ps_group[p] >= 0 || (t->ps_n_layers && t->ps_n_layers[p] > 1) || repl_head[p]) continue;
        if (t->ps_mask) { bool m = false; for (int i = 0; i < nt; i++) m = m || t->ps_mask[p * nt + i] >= 0; if (m) continue; }   // its phase 1 sees fewer leaves than the class's
        std::string key((const char*)(t->ps_req + p * R), (size_t)R * 8);
        key.append((const char*)&t->ps_slice_size[p], 4);
        key.append((const char*)(t->ps_slice_level + p * nt), (size_t)nt * 4);
        auto it = ids.find(key);
        if (it == ids.end()) {
          if ((int)cls_rep.size() >= TC_MAXCLS) continue;
          it = ids.emplace(key, (int)cls_rep.size()).first;
          cls_rep.push_back((int32_t)p);
        }
        ps_class[p] = it->second;
      }
    }
    const int ncls = (int)cls_rep.size();
    const int xslots = slots + 2 * ncls;   // wave slots, then the working copies of the request classes' tables, then those of their empty-cluster twins
    std::vector<int32_t> tas_of(prep.nF, -1);
    std::vector<TK> tks(nt);
    std::vector<int64_t*> work(nt), np(nt), priv(nt);
    std::vector<int32_t*> cls_tab(nt), cls_tab_e(nt), cflag(nt), par(nt);
    std::vector<long long*> cls_bytes(nt);
    const bool second = t->ps_adm_flavor != nullptr;   // heads that hold an admission (the second pass)
    std::vector<std::vector<int32_t>> h_first(nt), h_cnt(nt), h_par(nt);   // host mirrors of the trees (second pass: requiredReplacementDomain)
    int max_leaves = 1;
    for (int i = 0; i < nt; i++) {
      const kq_tas_topology& tp = t->topo[i];
      if (t->tas_flavor[i] < 0 || t->tas_flavor[i] >= prep.nF) return fail(KQ_EINVAL, "tas_flavor out of range");
      tas_of[t->tas_flavor[i]] = i;
      if (tp.n_levels < 1 || tp.n_levels > KQ_TAS_MAX_LEVELS) return fail(KQ_EUNSUPPORTED, "n_levels out of range");
      if (tp.n_resources != R || tp.pods_resource != t->topo[0].pods_resource) return fail(KQ_EINVAL, "every TAS topology must be built over one resource dictionary");
      TK& tk = tks[i];
      tk = TK{};
      TTopo& T = tk.T;
      if (tp.profile_mixed & ~(KQ_TAS_F_PROFILE_MIXED | KQ_TAS_F_BALANCED_PLACEMENT)) return fail(KQ_EUNSUPPORTED, "TASRespectNodeAffinityPreferred is not implemented: keep the Go path while the gate is on");
      T.L = tp.n_levels; T.R = R; T.pods = tp.pods_resource; T.profile_mixed = tp.profile_mixed & KQ_TAS_F_PROFILE_MIXED;
      T.balanced = (tp.profile_mixed & KQ_TAS_F_BALANCED_PLACEMENT) ? 1 : 0;
      for (int l = 0; l <= T.L; l++) T.level_off[l] = tp.level_off[l];
      T.D = T.level_off[T.L]; T.leaf_base = T.level_off[T.L - 1]; T.n_leaves = T.D - T.leaf_base;
      for (int l = 0; l < T.L; l++) if (T.level_off[l + 1] < T.level_off[l]) return fail(KQ_EINVAL, "level_off not monotone");
      std::vector<int32_t> first(std::max(T.D, 1), -1), cnt(std::max(T.D, 1), 0);
      for (int l = 1; l < T.L; l++) {
        int prev = -1;
        for (int d = T.level_off[l]; d < T.level_off[l + 1]; d++) {
          const int par = tp.parent[d];
          if (par < 0 || par >= T.level_off[l] - T.level_off[l - 1]) return fail(KQ_EINVAL, "parent out of range");
          if (par < prev) return fail(KQ_EINVAL, "domains of a level must be ordered by their parents (lexicographic levelValues)");
          prev = par;
          const int g = T.level_off[l - 1] + par;
          if (first[g] < 0) first[g] = d;
          cnt[g]++;
        }
      }
      max_leaves = std::max(max_leaves, T.n_leaves);
      const size_t cells = (size_t)T.n_leaves * R;
      std::vector<int32_t> parent(std::max(T.D, 1), -1);
      for (int l = 1; l < T.L; l++) for (int d = T.level_off[l]; d < T.level_off[l + 1]; d++) parent[d] = T.level_off[l - 1] + tp.parent[d];
      T.child_first = tstage(first.data(), first.size()); T.child_cnt = tstage(cnt.data(), cnt.size());
      par[i] = tstage(parent.data(), parent.size());
      be.sync();  // (first / cnt / parent are locals)
      if (second) { h_first[i] = first; h_cnt[i] = cnt; h_par[i] = parent; }
      cls_tab[i] = tgrow<int32_t>((size_t)5 * std::max(ncls, 1) * std::max(T.D, 1));
      cls_tab_e[i] = tgrow<int32_t>((size_t)5 * std::max(ncls, 1) * std::max(T.D, 1));
      cls_bytes[i] = (long long*)tgrow<int64_t>(std::max(ncls, 1));
      cflag[i] = tgrow<int32_t>((size_t)std::max(ncls, 1) * std::max(T.D, 1));
      be.memset(cflag[i], 0, (size_t)std::max(ncls, 1) * std::max(T.D, 1) * 4);
      T.free_cap = tstage(tp.free_capacity, cells);
      T.tas_usage = tstage(tp.tas_usage, cells);
      work[i] = tgrow<int64_t>(cells); np[i] = tgrow<int64_t>(cells); priv[i] = tgrow<int64_t>((size_t)slots * cells);
      TScratch& X = tk.X;
      X.max_set = (std::max(T.n_leaves, T.D - T.n_leaves + 1) + 1 + 15) & ~15;
      // (the slots behind the wave slots hold the working copies of the request classes' phase-1 tables)
      const size_t sd = (size_t)xslots * T.D, sm = (size_t)xslots * X.max_set;
      X.pc = tgrow<int32_t>(sd); X.sc = tgrow<int32_t>(sd); X.pcwl = tgrow<int32_t>(sd); X.scwl = tgrow<int32_t>(sd); X.lc = tgrow<int32_t>(sd);
      X.set = tgrow<int32_t>(sm); X.arr = tgrow<int32_t>(sm + xslots); X.cur = tgrow<int32_t>(sm); X.nxt = tgrow<int32_t>(sm);
      X.k0 = (uint64_t*)tgrow<int64_t>(sm); X.k1 = (uint64_t*)tgrow<int64_t>(sm);
      X.assumed = tgrow<int64_t>((size_t)xslots * cells);
      X.log = tgrow<int32_t>(sm); X.meta = tgrow<int32_t>((size_t)xslots * 4);
      if (T.balanced) {
        int wd = 1;
        for (int l = 0; l < T.L; l++) wd = std::max(wd, T.level_off[l + 1] - T.level_off[l]);
        X.bal_w = wd; X.bal_dp = be.bal_dp();
        X.bal_stride = 10ll * T.D + 4ll * wd + 2 * X.bal_dp;
        X.bal = tgrow<int32_t>((size_t)xslots * (size_t)X.bal_stride);
      }
      be.memset(X.meta, 0xff, (size_t)xslots * 4 * sizeof(int32_t));
    }
    const int n_adm = prep.n_adm;
    const int n_ent = n_adm > 0 ? t->adm_off[n_adm] : 0;
    if (n_ent > 0 && (!t->adm_tas || !t->adm_leaf || !t->adm_count || !t->adm_req)) return fail(KQ_EINVAL, "null admitted-row arrays in kq_cycle_tas");
    for (int e = 0; e < n_ent; e++) {
      if (t->adm_tas[e] < 0 || t->adm_tas[e] >= nt) return fail(KQ_EINVAL, "adm_tas out of range");
      if (t->adm_leaf[e] < 0 || t->adm_leaf[e] >= tks[t->adm_tas[e]].T.n_leaves) return fail(KQ_EINVAL, "adm_leaf out of range");
    }
    TCyc c{};
    c.flags = t->flags; c.n_tas = nt; c.R = R; c.slots = slots;
    c.tas_of_flavor = tstage(tas_of.data(), tas_of.size());
    c.tk = tstage(tks.data(), tks.size());
    c.work = tstage(work.data(), work.size()); c.np = tstage(np.data(), np.size()); c.priv = tstage(priv.data(), priv.size());
    c.cq_tas_only = tstage(t->cq_tas_only, (size_t)std::max(prep.nq, 1));
    std::vector<int32_t> zero_off(1, 0);
    c.adm_off = n_adm > 0 ? tstage(t->adm_off, (size_t)n_adm + 1) : tstage(zero_off.data(), 1);
    c.adm_tas = tstage(t->adm_tas, n_ent); c.adm_leaf = tstage(t->adm_leaf, n_ent); c.adm_count = tstage(t->adm_count, n_ent);
    c.adm_req = tstage(t->adm_req, (size_t)n_ent * R);
    c.ps_flags = tstage(t->ps_flags, nps); c.ps_kind = tstage(t->ps_kind, nps);
    c.ps_level = tstage(t->ps_level, nps * nt); c.ps_slice_size = tstage(t->ps_slice_size, nps); c.ps_slice_level = tstage(t->ps_slice_level, nps * nt);
    c.ps_group = tstage(t->ps_group, nps); c.ps_req = tstage(t->ps_req, nps * R);
    bool layered = false;
    if (t->ps_n_layers) for (size_t p = 0; p < nps; p++) {
      if (t->ps_n_layers[p] < 0 || t->ps_n_layers[p] > KQ_TAS_MAX_LEVELS) return fail(KQ_EUNSUPPORTED, "ps_n_layers out of range");
      if (t->ps_n_layers[p] > 1) layered = true;
    }
    if (layered && (!t->ps_layer_level || !t->ps_layer_size)) return fail(KQ_EINVAL, "ps_n_layers without ps_layer_level / ps_layer_size");
    c.ps_n_layers = layered ? tstage(t->ps_n_layers, nps) : nullptr;
    c.ps_layer_level = layered ? tstage(t->ps_layer_level, nps * nt * KQ_TAS_MAX_LEVELS) : nullptr;
    c.ps_layer_size = layered ? tstage(t->ps_layer_size, nps * KQ_TAS_MAX_LEVELS) : nullptr;
    // node feasibility rows (kq_cycle_tas.ps_mask / leaf_mask)
    c.ps_mask = nullptr; c.leaf_mask = nullptr; c.mask_stride = 0;
    if (t->ps_mask) {
      if (!t->leaf_mask || t->n_masks <= 0 || t->mask_stride <= 0) return fail(KQ_EINVAL, "ps_mask without leaf_mask rows");
      for (int i = 0; i < nt; i++) if (tks[(size_t)i].T.n_leaves > t->mask_stride) return fail(KQ_EINVAL, "mask_stride below a topology's leaf count");
      for (size_t p = 0; p < nps * (size_t)nt; p++) if (t->ps_mask[p] >= t->n_masks) return fail(KQ_EINVAL, "ps_mask names a row leaf_mask does not hold");
      c.ps_mask = tstage(t->ps_mask, nps * nt); c.leaf_mask = tstage(t->leaf_mask, (size_t)t->n_masks * t->mask_stride); c.mask_stride = t->mask_stride;
    }
    // ---- the second pass (kq_cycle_tas.ps_adm_flavor / ps_ex_*) ----------------------------------------------------------------
    std::vector<uint8_t> sp_kind;
    std::vector<int32_t> sp_req, sp_tas;
    if (second) {
      const int nR = prep.nR;
      if (t->ps_ex_off && t->ps_ex_off[nps] > 0 && (!t->ps_ex_leaf || !t->ps_ex_count || !t->ps_ex_flags)) return fail(KQ_EINVAL, "ps_ex_off without ps_ex_leaf / ps_ex_count / ps_ex_flags");
      if (t->ps_ex_off && t->ps_ex_off[0] != 0) return fail(KQ_EINVAL, "ps_ex_off[0] must be 0");
      sp_kind.assign(std::max<size_t>(nps, 1), 0); sp_req.assign(std::max<size_t>(nps, 1) * SP_W, 0); sp_tas.assign(std::max<size_t>(nps, 1), -1);
      for (int i = 0; i < n; i++)
        for (int g = h->ps_off[i]; g < h->ps_off[i + 1]; g++) {
          int32_t* sp = sp_req.data() + (size_t)g * SP_W;
          sp[SP_DEL] = -1;
          int tt = -1;
          for (int r = 0; r < nR; r++) {
            const int fl = t->ps_adm_flavor[(size_t)g * nR + r];
            if (fl < -1 || fl >= prep.nF) return fail(KQ_EINVAL, "ps_adm_flavor out of range");
            if (fl >= 0 && !(h->flags[i] & KQ_HEAD_HAS_QUOTA_RESERVATION)) return fail(KQ_EINVAL, "ps_adm_flavor on a head without KQ_HEAD_HAS_QUOTA_RESERVATION");
            if (fl >= 0 && tas_of[fl] >= 0) tt = tas_of[fl];
          }
          sp_tas[g] = tt;
          const int e0 = t->ps_ex_off ? t->ps_ex_off[g] : 0, e1 = t->ps_ex_off ? t->ps_ex_off[g + 1] : 0;
          if (e1 < e0) return fail(KQ_EINVAL, "ps_ex_off not monotone");
          if (e1 == e0) continue;
          if (tt < 0) return fail(KQ_EINVAL, "a podset holds a TopologyAssignment but none of its admitted flavors is a TAS flavor");
          const TTopo& T = tks[tt].T;
          sp_kind[g] |= SP_HAS_EX;
          for (int j = e0; j < e1; j++) {
            if (t->ps_ex_leaf[j] >= T.n_leaves || t->ps_ex_count[j] < 0) return fail(KQ_EINVAL, "existing assignment out of range");
            if (t->ps_ex_flags[j] & KQ_EX_UNHEALTHY) sp_kind[g] |= SP_UNHEALTHY;
          }
          if (!(h->flags[i] & KQ_HEAD_HAS_UNHEALTHY_NODES) || !(sp_kind[g] & SP_UNHEALTHY)) continue;
          // findReplacementAssignment :686 up to the placement: deleteDomain :693, the stale check :694, requiredReplacementDomain :759,
          // the rewrite of the slice request :703-722
          std::vector<std::pair<int, int32_t>> ex;   // the assignment after deleteDomain
          int32_t affected = 0;
          int n_first = 0;
          for (int j = e0; j < e1; j++) if (t->ps_ex_flags[j] & KQ_EX_FIRST) n_first++;
          if (n_first > 1) return fail(KQ_EINVAL, "more than one domain of a podset flagged KQ_EX_FIRST (a node holds one domain of an assignment)");
          for (int j = e0; j < e1; j++) {
            if (t->ps_ex_flags[j] & KQ_EX_FIRST) { affected = t->ps_ex_count[j]; sp[SP_DEL] = t->ps_ex_leaf[j]; }
            else ex.push_back({t->ps_ex_leaf[j], t->ps_ex_count[j]});
          }
          sp[SP_COUNT] = affected;
          sp[SP_SSIZE] = t->ps_slice_size[g]; sp[SP_SLEVEL] = t->ps_slice_level[(size_t)g * nt + tt];
          sp[SP_LO] = 0; sp[SP_HI] = 0;   // (hi <= 0: every leaf)
          int stale = -1;
          for (size_t j = 0; j < ex.size() && stale < 0; j++) if (ex[j].first < 0) stale = (int)j;
          if (stale >= 0) { sp[SP_STATUS] = KQ_TAS_STALE; sp[SP_OPA] = stale; continue; }
          std::vector<std::pair<int, int32_t>> cons;   // PodSetSliceRequiredTopologyConstraints (level, size), outermost first
          const int nl = t->ps_n_layers ? t->ps_n_layers[g] : 0;
          if (nl > 1) for (int j = 0; j < nl; j++) cons.push_back({t->ps_layer_level[((size_t)g * nt + tt) * KQ_TAS_MAX_LEVELS + j], t->ps_layer_size[(size_t)g * KQ_TAS_MAX_LEVELS + j]});
          else if (t->ps_slice_size[g] != 1) cons.push_back({t->ps_slice_level[(size_t)g * nt + tt], t->ps_slice_size[g]});
          const std::vector<int32_t>& P = h_par[tt];
          auto ancestor = [&](int leaf, int lv) { int d = T.leaf_base + leaf; for (int l = T.L - 1; l > lv; l--) d = P[d]; return d; };
          auto incomplete = [&](int32_t missing, int32_t size, int lv) {   // findIncompleteSliceDomain :842 (the first in the canonical domain order)
            if (lv < 0 || lv >= T.L || size <= 0) return -1;
            std::map<int, int32_t> use;
            for (auto& dc : ex) if (dc.first >= 0) use[ancestor(dc.first, lv)] += dc.second;
            for (auto& kv : use) if ((kv.second + missing) % size == 0) return kv.first;
            return -1;
          };
          const int level = t->ps_level[(size_t)g * nt + tt];
          const int32_t size0 = cons.empty() ? 1 : cons[0].second;
          int required = -1;   // requiredReplacementDomain :759
          if (level >= 0 && level < T.L && !ex.empty() && size0 > 0) {
            if (!cons.empty() && affected % size0 != 0) {
              bool found = false;
              if (cons.size() > 1)
                for (size_t j = cons.size(); j-- > 0 && !found;)
                  if (cons[j].second > 0 && affected % cons[j].second != 0) { required = incomplete(affected, cons[j].second, cons[j].first); found = true; }
              if (!found) required = incomplete(affected, size0, cons[0].first);
            } else if (t->ps_kind[g] == KQ_TAS_REQUIRED && ex[0].first >= 0) required = ancestor(ex[0].first, level);
          }
          if (size0 <= 0) { sp[SP_STATUS] = KQ_TAS_BAD_SLICE_SIZE; continue; }   // :699-702
          if (!cons.empty() && required >= 0 && affected % size0 != 0) {   // :703-722
            int32_t eff = 1; int eff_level = -1;
            for (size_t j = cons.size(); j-- > 0;) if (cons[j].second > 0 && affected % cons[j].second == 0) { eff = cons[j].second; eff_level = cons[j].first; break; }
            sp[SP_NLAY] = 1;
            sp[SP_SSIZE] = eff_level >= 0 ? eff : 1; sp[SP_SLEVEL] = eff_level >= 0 ? eff_level : T.L - 1;
          }
          if (required >= 0) {   // the leaves below it (:1902-1905): the domains of a level are ordered by their parents, so a contiguous run
            int lv = 0;
            while (lv < T.L && !(required >= T.level_off[lv] && required < T.level_off[lv + 1])) lv++;
            int lo = required, hi = required + 1;
            for (int l = lv; l < T.L - 1 && hi > lo; l++) {
              int nlo = -1, nhi = -1;
              for (int d = lo; d < hi; d++) if (h_first[tt][d] >= 0) { if (nlo < 0) nlo = h_first[tt][d]; nhi = h_first[tt][d] + h_cnt[tt][d]; }
              if (nlo < 0) { lo = hi = T.leaf_base + 1; break; }   // (a domain without leaves: the empty range [1, 1))
              lo = nlo; hi = nhi;
            }
            sp[SP_LO] = lo - T.leaf_base; sp[SP_HI] = hi - T.leaf_base;
          }
        }
      // (the podsets of a replacement are placed one by one in podset order; the reference walks groupsOrder :591-603)
      for (int i = 0; i < n; i++) {
        if (!(h->flags[i] & KQ_HEAD_HAS_UNHEALTHY_NODES)) continue;
        for (int g = h->ps_off[i]; g < h->ps_off[i + 1]; g++) {
          if (t->ps_group[g] < 0) continue;
          int last = g;
          for (int q = g + 1; q < h->ps_off[i + 1]; q++) if (t->ps_group[q] == t->ps_group[g]) { if (q != last + 1) return fail(KQ_EUNSUPPORTED, "node replacement of a workload whose podset groups interleave"); last = q; }
        }
      }
      c.ps_adm_flavor = tstage(t->ps_adm_flavor, nps * nR);
      c.sp_kind = tstage(sp_kind.data(), sp_kind.size()); c.sp_req = tstage(sp_req.data(), sp_req.size());
      c.sp_del_out = tgrow<int32_t>(std::max<size_t>(nps, 1));
      be.memset(c.sp_del_out, 0, std::max<size_t>(nps, 1) * 4);
      be.sync();   // (sp_kind / sp_req are staged from locals that stay alive, but the copy must not outlive a later resize)
    }
    c.q_i32 = tgrow<int32_t>((size_t)slots * TQ_WORDS); c.q_u8 = tgrow<uint8_t>((size_t)slots * (TC_P + 8)); c.q_spr = tgrow<int64_t>((size_t)slots * TC_P * R);
    // a head's TopologyAssignments hold at most min(pods, leaves) domains per podset
    int d_cap = 1;
    for (int i = 0; i < n; i++) {
      int64_t tot = 0;
      for (int p = h->ps_off[i]; p < h->ps_off[i + 1]; p++) tot += std::min<int64_t>(std::max(h->ps_count[p], 0), max_leaves);
      d_cap = (int)std::max<int64_t>(d_cap, std::min<int64_t>(tot, (int64_t)KQ_MAXPS * max_leaves));
    }
    c.d_cap = d_cap;
    c.d_leaf = tgrow<int32_t>((size_t)slots * 2 * d_cap); c.d_count = tgrow<int32_t>((size_t)slots * 2 * d_cap);
    c.h_tas = tgrow<int32_t>(nps); c.h_pos = tgrow<int32_t>(nps); c.h_n = tgrow<int32_t>(nps);
    be.memset(c.h_tas, 0xff, nps * 4); be.memset(c.h_pos, 0, nps * 4); be.memset(c.h_n, 0, nps * 4);
    c.pool_cap = 2 * std::max(tout->dom_cap, 1) + d_cap;   // a recomputation inside processEntry publishes a second segment
    c.pool_leaf = tgrow<int32_t>(c.pool_cap); c.pool_count = tgrow<int32_t>(c.pool_cap);
    int64_t* tmisc = tgrow<int64_t>(8);
    be.memset(tmisc, 0, 8 * sizeof(int64_t));
    c.stats = (long long*)tmisc; c.pool_used = (int32_t*)(tmisc + 4);
    c.tree_state = tgrow<int32_t>((size_t)std::max(prep.n_tree, 1) * 12);
    be.memset(c.tree_state, 0, (size_t)std::max(prep.n_tree, 1) * 12 * 4);
    // request classes
    c.ncls = ncls;
    c.ps_class = tstage(ps_class.data(), ps_class.size());
    {
      std::vector<int64_t> creq((size_t)std::max(ncls, 1) * R, 0);
      std::vector<int32_t> css(std::max(ncls, 1), 1), csl((size_t)std::max(ncls, 1) * nt, -1);
      for (int q = 0; q < ncls; q++) {
        const size_t p = (size_t)cls_rep[q];
        for (int r = 0; r < R; r++) creq[(size_t)q * R + r] = t->ps_req[p * R + r];
        css[q] = t->ps_slice_size[p];
        for (int i = 0; i < nt; i++) csl[(size_t)q * nt + i] = t->ps_slice_level[p * nt + i];
      }
      c.cls_req = tstage(creq.data(), creq.size()); c.cls_ssize = tstage(css.data(), css.size()); c.cls_slevel = tstage(csl.data(), csl.size());
      c.cls_tab = tstage(cls_tab.data(), cls_tab.size()); c.cls_bytes = tstage(cls_bytes.data(), cls_bytes.size());
      c.cls_tab_e = getenv("KQ_TAS_EMPTY_TABLES_OFF") ? nullptr : tstage(cls_tab_e.data(), cls_tab_e.size());   // (A/B switch)
      c.par = (const int32_t* const*)tstage(par.data(), par.size()); c.cflag = tstage(cflag.data(), cflag.size());
      c.cls_ok = tgrow<uint8_t>((size_t)nt * std::max(ncls, 1));
      be.memset(c.cls_ok, 0, (size_t)nt * std::max(ncls, 1));
      be.sync();  // (locals)
    }
    tas_n_cls = nt * ncls;
    tas_lds_want = 0;
    if (ncls > 0) for (int i = 0; i < nt; i++) tas_lds_want = std::max(tas_lds_want, TX_BYTES + tas_lds_layout(tks[i].T.D, tks[i].X.max_set).total);
    TCyc* d_tc = tstage(&c, 1);
    if (n_ent > 0) be.launch_tas_base(d_tc, n_ent);   // base plane += workload.TASUsage() of every admitted row
    for (int i = 0; i < nt; i++) {
      const size_t bytes = (size_t)tks[i].T.n_leaves * R * sizeof(int64_t);
      be.d2d(work[i], tks[i].T.tas_usage, bytes); be.d2d(np[i], tks[i].T.tas_usage, bytes);
    }
    rc = be.sync();   // (c, tks and the pointer tables are locals)
    if (rc != KQ_OK) return fail(rc, be.error());
    rc = cycle_exec(0, out, false, ShardCall{}, nullptr, d_tc);
    if (rc != KQ_OK) return rc;
    last_cycle_n = -1;   // kq_cycle_commit does not carry leaf usage: a TAS cycle is not committable through it
    std::vector<int32_t> ht(nps * 3);
    int64_t hm[8];
    be.d2h(ht.data(), c.h_tas, nps * 4); be.d2h(ht.data() + nps, c.h_pos, nps * 4); be.d2h(ht.data() + 2 * nps, c.h_n, nps * 4);
    be.d2h(hm, tmisc, sizeof(hm));
    rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    const int used = std::min(((int32_t*)(hm + 4))[0], c.pool_cap);
    std::vector<int32_t> pl(std::max(used, 1)), pc(std::max(used, 1));
    if (used > 0) { be.d2h(pl.data(), c.pool_leaf, (size_t)used * 4); be.d2h(pc.data(), c.pool_count, (size_t)used * 4); }
    if (tout->tas_usage_after) {
      size_t o = 0;
      for (int i = 0; i < nt; i++) { const size_t cells = (size_t)tks[i].T.n_leaves * R; be.d2h(tout->tas_usage_after + o, work[i], cells * sizeof(int64_t)); o += cells; }
    }
    rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    if (stats) { stats[0] = hm[0]; stats[1] = hm[1]; stats[2] = hm[2]; stats[3] = hm[3]; }
    std::vector<int32_t> del_out;
    if (second) { del_out.resize(std::max<size_t>(nps, 1)); be.d2h(del_out.data(), c.sp_del_out, nps * 4); rc = be.sync(); if (rc != KQ_OK) return fail(rc, be.error()); }
    int tot = 0;
    auto put = [&](int leaf, int32_t count) {
      if (tot >= tout->dom_cap) return false;
      if (tout->dom_leaf) tout->dom_leaf[tot] = leaf;
      if (tout->dom_count) tout->dom_count[tot] = count;
      tot++;
      return true;
    };
    for (size_t p = 0; p < nps; p++) {
      int tt = ht[p]; const int pos = ht[nps + p], cnt = tt >= 0 ? ht[2 * nps + p] : 0;
      if (second && (sp_kind[p] & SP_HAS_EX)) {
        // a podset whose admission holds a TopologyAssignment: what the device published is the replacement's net usage — the caller gets
        // mergeTopologyAssignments :2072 of it with the admission's other domains; a podset that was not placed again (no failed node
        // in it, or no replacement found) keeps the admission's assignment as it is
        const int e0 = t->ps_ex_off[p], e1 = t->ps_ex_off[p + 1];
        if (tt >= 0) {
          std::vector<std::pair<int, int32_t>> m;
          for (int j = 0; j < cnt; j++) m.push_back({pl[pos + j], pc[pos + j]});
          if (del_out[p] > 0) m.push_back({sp_req[p * SP_W + SP_DEL], del_out[p]});
          for (int j = e0; j < e1; j++) if (!(t->ps_ex_flags[j] & KQ_EX_FIRST)) m.push_back({t->ps_ex_leaf[j], t->ps_ex_count[j]});
          std::stable_sort(m.begin(), m.end(), [](const std::pair<int, int32_t>& a, const std::pair<int, int32_t>& b) { return a.first < b.first; });
          // (merged on the local list, not on the output arrays: dom_leaf / dom_count may be null for a caller that only wants the offsets)
          size_t u = 0;
          for (size_t j = 0; j < m.size(); j++) { if (u > 0 && m[u - 1].first == m[j].first) m[u - 1].second += m[j].second; else m[u++] = m[j]; }
          m.resize(u);
          for (auto& d : m) if (!put(d.first, d.second)) return fail(KQ_ECAPACITY, "dom_cap too small");
        } else if (del_out[p] >= 0) {   // (-1: a failed result took the assignment away, UpdateForTASResult flavorassigner.go:90)
          tt = sp_tas[p];
          for (int j = e0; j < e1; j++) if (!put(t->ps_ex_leaf[j], t->ps_ex_count[j])) return fail(KQ_ECAPACITY, "dom_cap too small");
        }
        tout->ps_tas[p] = tt;
        tout->dom_off[p + 1] = tot;
        continue;
      }
      tout->ps_tas[p] = tt;
      for (int j = 0; j < cnt; j++) if (!put(pl[pos + j], pc[pos + j])) return fail(KQ_ECAPACITY, "dom_cap too small");
      tout->dom_off[p + 1] = tot;
    }
    return KQ_OK;
  }

  // nominate_only: stop after k_nominate (kq_nominate_run_resident: flavor assignment + targets for every head of the batch,
  // no iterator, no processEntry; nothing to commit afterwards)
  // Sharded nominate / merged process (kq_cycle_nominate_shard, kq_cycle_process_merged): mode 1 = nominate the heads of `mine` and
  // export them into the exchange buffer; mode 2 = import the merged buffer instead of nominating, then the rest of the cycle.
  struct ShardCall { int mode = 0; const uint8_t* mine = nullptr; int64_t* x = nullptr; int world = 1, rank = 0; };
  size_t shard_words_for(int slot, const kq_decisions* out, int world) {
    const HeadBatch& hbch = batches[slot];
    const int rsn_win = out->rsn_cap > 0 ? std::min(std::max(prep.max_rsn_per_podset * hbch.max_nps, 8), 4096) : 0;
    return shard_words(hbch.n, hbch.nps, prep.nR, world, rsn_win, 2 * std::max(out->tgt_cap, 1));
  }
  // where every decision array lives inside the packed output region of a cycle (one D2H per cycle)
  struct PackLayout {
    size_t o_status = 0, o_action = 0, o_nmode = 0, o_mode = 0, o_rq = 0, o_skip = 0, o_borrow = 0, o_order = 0, o_flavor = 0, o_rmode = 0, o_tried = 0,
           o_pscount = 0, o_tpos = 0, o_tn = 0, o_misc = 0, o_rsn = 0, pack_bytes = 0;
    int rsn_win = 0;
  };
  // One asynchronous step of the pending loop in flight (kq_pending_step / kq_pending_step_wait): the packed decisions, the head
  // count and the target pool land in pinned host memory behind an event; the host unpacks them one step later.
  struct StepStage {
    uint8_t* host = nullptr; size_t cap = 0;
    bool busy = false;
    PackLayout lay; int n_bound = 0; size_t nps_bound = 0, nR = 0; int pool_cap = 0; bool with_pool = false, with_heads = false;
    size_t o_counts = 0, o_prow = 0, o_preason = 0, o_headwl = 0, o_rsn = 0;   // o_rsn: the reason windows (kq_pending_step_reasons), staged like the pool
    int tgt_cap = 0; int64_t cycle = 0;
  };
  StepStage steps[2];
  int step_rsn_cap = 0;   // kq_pending_step_reasons: reason records of the steps issued from now on (0: none recorded)
  bool step_unfused = getenv("KQ_STEP_UNFUSED") != nullptr;   // (A/B switch: the step's commit / apply / release as the separate launches of the call-by-call API)
  int64_t steps_issued = 0, steps_waited = 0;

  int cycle_exec(int slot, kq_decisions* out, bool nominate_only = false, ShardCall sc = ShardCall{}, StepStage* st = nullptr, const TCyc* d_tc = nullptr) {
    if (!have_snapshot) return fail(KQ_EINVAL, "kq_cycle_run before kq_snapshot_put");
    if (!st && steps_issued != steps_waited) return fail(KQ_EINVAL, "a kq_pending_step is in flight: its outputs share the cycle's buffers (kq_pending_step_wait first)");
    if (slot < 0 || slot >= (int)batches.size() || !batches[slot].valid) return fail(KQ_EINVAL, "unknown head batch");
    HeadBatch& hbch = batches[slot];
    const int n = hbch.n;
    const int slot_cap = hbch.slot_cap;
    int rc = KQ_OK;
    if (out->tgt_off) out->tgt_off[0] = 0;
    if (n == 0 && !st) {
      last_kernel_ms = 0; last_bytes = 0; last_cycle_n = nominate_only ? -1 : 0;
      flush_levels();
      be.d2d(grow<int64_t>(b_usage_work, (size_t)prep.N * prep.nfr), d_usage, (size_t)prep.N * prep.nfr * sizeof(int64_t));
      return be.sync();
    }
    const size_t nps = hbch.nps, nR = prep.nR;
    const size_t Nfr = (size_t)prep.N * prep.nfr;
    K k{};
    k.S = S;
    k.C.n_fs = std::min(std::max(cfg.n_fs_strategies, 0), 2); k.C.fs[0] = cfg.fs_strategies[0]; k.C.fs[1] = cfg.fs_strategies[1];
    k.C.dbg_variant = 0;
#ifdef KQ_PROF
    { const char* dv = getenv("KQ_DEBUG_VARIANT"); k.C.dbg_variant = dv ? atoi(dv) : 0; }
#endif
    k.C.fs_plain = (prep.fs_plain && hbch.plain && prep.nR <= KQ_MAXR && !force_exact_drs) ? 1 : 0;
    // scan-formulated classical search: usage and admitted quantities must be plain (its prefix sums are ordinary additions)
    k.C.cs_on = (!cfg.fair_sharing && prep.any_preemption && prep.fs_plain && !cs_disable) ? 1 : 0;
    k.C.fs_on = (cfg.fair_sharing && prep.any_preemption && prep.fs_plain && !fs_disable && !d_tc) ? 1 : 0;   // (a TAS cycle's searches carry leaf usage: the walk)
    k.C.fs_batch = fs_batch_bits;
    k.C.cs_lazy = cs_lazy_mode;
    k.C.fs_lrun = fs_lrun_on ? 1 : 0;
    k.C.any_preempt = prep.any_preemption ? 1 : 0;
    k.C.gates = cfg.gates; k.C.fair_sharing = cfg.fair_sharing; k.C.quota_check_strategy = cfg.quota_check_strategy; k.C.cycle = hbch.cycle;
    k.H = hbch.H;
    // outputs
    const int pool_cap = std::max(out->tgt_cap, 1);
    DOut& O = k.O;
    // every array that goes back to the host lives in ONE packed device region -> a single D2H per cycle
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
    PackLayout L;
    L.o_status = carve(n); L.o_action = carve(n); L.o_nmode = carve(n); L.o_mode = carve(n); L.o_rq = carve(n); L.o_skip = carve(n);
    L.o_borrow = carve((size_t)n * 4); L.o_order = carve((size_t)n * 4);
    L.o_flavor = carve(nps * nR * 4); L.o_rmode = carve(nps * nR); L.o_tried = carve(nps * nR * 4); L.o_pscount = carve(nps * 4);
    L.o_tpos = carve((size_t)n * 4); L.o_tn = carve((size_t)n * 4); L.o_misc = carve(4 * sizeof(int64_t));
    const size_t o_status = L.o_status, o_action = L.o_action, o_nmode = L.o_nmode, o_mode = L.o_mode, o_rq = L.o_rq, o_skip = L.o_skip, o_borrow = L.o_borrow,
                 o_order = L.o_order, o_flavor = L.o_flavor, o_rmode = L.o_rmode, o_tried = L.o_tried, o_pscount = L.o_pscount, o_tpos = L.o_tpos, o_tn = L.o_tn,
                 o_misc = L.o_misc;
    // reason records: a window per head, sized for the longest scan any head of the batch can produce
    const int rsn_win = out->rsn_cap > 0 ? std::min(std::max(prep.max_rsn_per_podset * hbch.max_nps, 8), 4096) : 0;
    L.o_rsn = carve(rsn_win ? (size_t)n * 4 : 0);
    L.pack_bytes = off; L.rsn_win = rsn_win;
    const size_t o_rsn = L.o_rsn, pack_bytes = L.pack_bytes;
    uint8_t* pack = grow<uint8_t>(ob[0], pack_bytes);
    O.status = pack + o_status; O.action = pack + o_action; O.nominated_mode = pack + o_nmode; O.mode = pack + o_mode;
    O.requeue_reason = pack + o_rq; O.skip = pack + o_skip;
    O.borrowing = (int32_t*)(pack + o_borrow); O.order = (int32_t*)(pack + o_order);
    O.flavor = (int32_t*)(pack + o_flavor); O.res_mode = pack + o_rmode; O.tried_idx = (int32_t*)(pack + o_tried);
    O.ps_count = (int32_t*)(pack + o_pscount);
    O.use_n = grow<int32_t>(ob[12], n); O.use_fr = grow<int32_t>(ob[13], (size_t)n * KQ_MAXU); O.use_qty = grow<int64_t>(ob[14], (size_t)n * KQ_MAXU);
    O.tgt_pos = (int32_t*)(pack + o_tpos); O.tgt_n = (int32_t*)(pack + o_tn);
    // recomputation on overlap appends a second target segment per head: size the pool for both (sharded: behind every rank's segment)
    // sharded: a rank's nomination gets the budget the whole pool has in the ordinary cycle (2 x tgt_cap: KQ_ECAPACITY beyond it, exactly
    // as there), the merged pool is every rank's segment plus the same room again for recomputations
    const int seg_cap = 2 * pool_cap;
    O.pool_cap = sc.mode == 1 ? seg_cap : (sc.mode == 2 ? seg_cap * (sc.world + 1) : pool_cap * 2);
    O.pool_row = grow<int32_t>(ob[17], O.pool_cap); O.pool_reason = grow<uint8_t>(ob[18], O.pool_cap);
    O.rsn_win = rsn_win; O.rsn_n = rsn_win ? (int32_t*)(pack + o_rsn) : nullptr;
    O.rsn = rsn_win ? grow<RsnRec>(ob[19], (size_t)n * rsn_win) : nullptr;
    int64_t* misc = (int64_t*)(pack + o_misc);
    DPrep pp{};  // the fills and copies every cycle starts with, gathered into one launch (be.launch_prep below)
    auto prep_fill = [&](void* dst, size_t words, uint32_t v) { if (words) pp.op[pp.n++] = DPrepOp{dst, nullptr, (uint32_t)words, v}; };
    auto prep_copy = [&](void* dst, const void* src, size_t words) { if (words) pp.op[pp.n++] = DPrepOp{dst, src, (uint32_t)words, 0}; };
    prep_fill(misc, 8, 0);
    O.pool_count = (int32_t*)misc; O.error = (int32_t*)misc + 1; O.stat_bytes = (long long*)(misc + 1);  // [1]=nominate bytes, [2]=process bytes
    // nominated flavors start empty (a head's rows are rewritten by assign_flavors)
    prep_fill(O.flavor, nps * nR, 0xffffffffu);
    if (rsn_win) prep_fill(O.rsn_n, (size_t)n, 0);
    // scratch: one slot per resident wave
    const int slots_nom = std::min(n, be.max_slots());
    // fair sharing: helper workgroups of k_process_fair (K::help) take the victim searches of a recomputation; each needs a scratch slot
    const bool want_help = cfg.fair_sharing && prep.any_preemption && prep.fs_plain && !fs_disable && !nominate_only && !help_disable;
    const int n_help = want_help ? be.help_blocks(prep.n_tree) : 0;
    const int slots = std::max(slots_nom, prep.n_tree + n_help);
    DScratch& X = k.X;
    X.max_tree_nodes = prep.max_tree_nodes; X.max_tree_cqs = std::max(prep.max_tree_cqs, 1); X.max_tree_rows = std::max(prep.max_tree_rows, 1);
    // fair sharing: the private state of a victim search covers every flavor-resource (DRS reads them all)
    X.slot_cap = cfg.fair_sharing ? std::max<int>(slot_cap, (int)prep.nfr) : slot_cap; X.tgt_cap = std::max(prep.max_tree_rows, 1);
    X.w = grow<int64_t>(b_w, (size_t)slots * X.max_tree_nodes * X.slot_cap);
    X.cqinfo = grow<uint8_t>(b_cqinfo, (size_t)slots * X.max_tree_cqs);
    X.cls = grow<uint8_t>(b_cls, (size_t)slots * X.max_tree_rows);
    X.tgt_row = grow<int32_t>(b_tgt_row, (size_t)slots * X.tgt_cap);
    X.tgt_reason = grow<uint8_t>(b_tgt_reason, (size_t)slots * X.tgt_cap);
    X.nom = grow<int32_t>(b_nom, (size_t)slots * KQ_MAXPS * nR);
    X.cand = grow<int32_t>(b_cand, (size_t)slots * X.max_tree_rows);
    X.mark = (uint64_t*)grow<int64_t>(b_mark, (size_t)slots * ((X.max_tree_rows + 63) / 64));
    X.cs = nullptr; X.cs_bytes = 0;
    if (k.C.cs_on) {  // spill space for the arrays of a search that do not fit the workgroup's LDS (worst case: CS_NS slots)
      X.cs_bytes = (int64_t)((cs_bytes(CS_NS, prep.cs_max_bucket, prep.max_tree_nodes, prep.max_tree_cqs, false) + 255) & ~(size_t)255);
      X.cs = grow<unsigned char>(b_cs, (size_t)slots * (size_t)X.cs_bytes);
    }
    const bool fs_lds = k.C.fs_on != 0;  // kq_fs.hpp may run
    if (fs_lds) {  // the same spill space serves the state of an LDS-formulated fair search that does not (all) fit the LDS
      X.cs_bytes = (int64_t)((fs_bytes(prep.max_tree_nodes, prep.max_tree_cqs, nR, (int)prep.nfr, prep.max_tree_mw, FS_NCMAX) + 255) & ~(size_t)255);
      X.cs = grow<unsigned char>(b_cs, (size_t)slots * (size_t)X.cs_bytes);
    }
    k.cq_rm_bytes = grow<int32_t>(b_rmb, std::max(prep.nq, 1));
    prep_fill(k.cq_rm_bytes, (size_t)std::max(prep.nq, 1), 0);
    if (cfg.fair_sharing) {
      const size_t tq = (size_t)slots * X.max_tree_cqs, tn = (size_t)slots * X.max_tree_nodes;
      X.qcnt = grow<int32_t>(b_fs[0], tq); X.qhead = grow<uint32_t>(b_fs[1], tq); X.cohp = grow<uint8_t>(b_fs[2], tn);
      X.cq_ent = grow<int32_t>(b_fs[3], tq); X.fs_keys = (uint64_t*)grow<int64_t>(b_fs[4], tq * KQ_MAXD * 4);
      X.fs_win = grow<int32_t>(b_fs[7], tn); X.fs_seq = grow<int32_t>(b_fs[8], tq);
      X.fs_key = grow<int32_t>(b_fs[9], n);
      X.fs_stale = grow<uint8_t>(b_fs[10], tq); X.fs_cost = grow<int32_t>(b_fs[11], tq * KQ_MAXD);
      X.fs_sum = (long long*)grow<int64_t>(b_fs[12], slots); X.fs_ctl = grow<int32_t>(b_fs[13], (size_t)slots * 4);
      prep_fill(X.fs_key, (size_t)n, 0xffffffffu);
      // per-node borrowed sums (the segmented reduction DRS is built from), for the cycle-start plane and the work plane
      X.bu_sum = grow<int64_t>(b_fs[14], (size_t)prep.N * nR); X.bu_pos = grow<int32_t>(b_fs[15], prep.N);
      X.bs_sum = grow<int64_t>(b_fs[16], (size_t)prep.N * nR); X.bs_pos = grow<int32_t>(b_fs[17], prep.N);
      X.psum = grow<int64_t>(b_fs[18], tn * nR); X.ppos = grow<int32_t>(b_fs[19], tn);
    }
    k.usage = d_usage; k.usage_big = d_big;
    k.usage_work = grow<int64_t>(b_usage_work, Nfr);
    k.usage_np = grow<int64_t>(b_usage_np, Nfr);
    k.preempted = grow<uint8_t>(b_preempted, ((size_t)std::max(prep.n_adm, 1) + 3) & ~(size_t)3);
    k.prof = (long long*)grow<int64_t>(b_prof, 128);   // [64] segment counters + [64] a sink (KQ_PROF_SKIP_NOMINATE: the nominate kernels count there)
    k.grec = grow<PRec>(b_grec, n);
    k.cq_dirty = grow<uint8_t>(b_cqd, std::max(prep.nq, 1));  // cleared per head by k_records
    k.defer_list = grow<int32_t>(b_defer, (size_t)n + 2 + SIMC_WORDS); k.defer_count = k.defer_list + n; k.nom_ticket = k.defer_list + n + 1;
    k.sim_ctl = k.defer_list + n + 2;   // (zeroed with defer_count)
    k.cq_heads = grow<int32_t>(b_cqh, (size_t)std::max(prep.nq, 1) + std::max(prep.n_tree, 1) + 8);
    k.spec_resume = k.cq_heads + std::max(prep.nq, 1); k.spec_stats = spec_stats_on ? k.spec_resume + std::max(prep.n_tree, 1) : nullptr;
    prep_fill(k.cq_heads, (size_t)std::max(prep.nq, 1) + std::max(prep.n_tree, 1) + 8, 0);  // resume 0: the serial kernel takes the whole tree
    // (not in a TAS cycle: k_process_tas runs no speculative rounds, and k_records — which writes the entry records k_order_scatter's
    // spec_hdr_of reads — is not launched there. Until round 5 the pointers were set all the same, so k_order_scatter read nuse / plen of
    // records nobody had written: nothing with fresh (zero) or poisoned memory, a walk of cbig[u][i] far past the record — a GPU memory
    // access fault when the buffer happened to end a mapped region — with whatever an earlier engine of the process had left there.
    // That was the "unexplained" abort inside kq_cycle_run_tas of rounds 3-5, found with AMD_SERIALIZE_KERNEL=3: profiles/r05p_*.)
    k.spec_kt = (cfg.fair_sharing || d_tc) ? nullptr : grow<int64_t>(b_spkt, (size_t)std::min(std::max(prep.n_tree, 1), (int)SP_SLOTS) * SP_KT_WORDS);
    if (k.spec_kt) {  // per-cell constants of the rounds, written by k_records
      const size_t cells = (size_t)n * FU * FD, slots_ = (size_t)n * FU;
      int64_t* a = grow<int64_t>(b_spc, cells * 2 + slots_ * 2 + (cells + 1) / 2);
      k.spec_hdr = grow<SpecHdr>(b_sphdr, (size_t)n);
      k.spec_K = a; k.spec_T = a + cells; k.spec_push = a + 2 * cells; k.spec_nv = a + 2 * cells + slots_; k.spec_o = (int32_t*)(a + 2 * cells + 2 * slots_);
    }
    prep_fill(k.defer_count, 2 + SIMC_WORDS, 0);
    // simulations ahead of the full nominate pass (K::sim_*): only where a victim search can happen at all, not in a TAS cycle (every head
    // goes through the full code there) and not for the sharded nominate's masked batches (the lean pass still lists per head: fine)
    k.sim_nscan = nullptr; k.sim_task = nullptr; k.sim_res = nullptr; k.sim_key = nullptr; k.sim_cell = nullptr; k.sim_cap = 0;
    if (prep.any_preemption && !d_tc && sim_ahead_on && n > 0) {
      k.sim_cap = n * CELLS * 2;   // (two full scans per head; a scan that finds no room is searched in place by the full pass)
      k.sim_nscan = grow<int32_t>(b_sim[0], (size_t)n * (1 + SIM_KS)); k.sim_key = k.sim_nscan + n;
      k.sim_cell = grow<int32_t>(b_sim[1], (size_t)n * SIM_KS * CELLS);
      k.sim_task = grow<SimTask>(b_sim[2], (size_t)k.sim_cap);
      k.sim_res = grow<HelpRes>(b_sim[3], (size_t)k.sim_cap);
      if (pp.n < 15) prep_fill(k.sim_nscan, (size_t)n, 0); else be.memset(k.sim_nscan, 0, (size_t)n * 4);
    }
    k.help = nullptr; k.help_quit = nullptr; k.help_trees = 0;
    k.tc = d_tc;
    HelpBox* d_help = nullptr;
    if (n_help > 0) {  // one box per tree + the quit counter behind them, zeroed every cycle
      d_help = grow<HelpBox>(b_help, (size_t)prep.n_tree + 1);
      prep_fill(d_help, ((size_t)prep.n_tree + 1) * sizeof(HelpBox) / 4, 0);
    }
    {  // sharding certificate (K::root_margin): one slack per (tree, flavor-resource), one flag per tree
      const size_t cells = (size_t)std::max(prep.n_tree, 1) * prep.nfr;
      k.root_margin = (long long*)grow<int64_t>(b_cert, cells + (std::max(prep.n_tree, 1) + 1) / 2);
      k.cert_flags = (int32_t*)(k.root_margin + cells);
      prep_fill(k.root_margin, cells * 2, 0x7f7f7f7fu);
      prep_fill(k.cert_flags, (size_t)std::max(prep.n_tree, 1), 0);
    }
    int32_t* order_idx = grow<int32_t>(b_order, n);
    k.order_idx = order_idx;
    int32_t* rank = grow<int32_t>(b_rank, n);
    flush_levels();  // commits / releases since the last cycle: cohort usage re-derived before anything reads it
    prep_copy(k.usage_work, d_usage, Nfr * 2);
    prep_copy(k.usage_np, d_usage, Nfr * 2);
    prep_fill(k.preempted, ((size_t)std::max(prep.n_adm, 1) + 3) / 4, 0);
    if (!cfg.fair_sharing) prep_fill(rank, (size_t)n, 0);  // k_order accumulates into it
    // (the plain classical cycle: the fills and copies are launched in front of the nominate pass, with its argument block riding along)
    const bool prep_with_k = B::FUSE_PREP_K && !cfg.fair_sharing && sc.mode == 0 && !d_tc;
    if (!prep_with_k) be.launch_prep(pp);

    be.timer_mark(0);
    if (cfg.fair_sharing) {
      be.launch_fs_sums(k);
      be.d2d(X.bs_sum, X.bu_sum, (size_t)prep.N * nR * sizeof(int64_t));
      be.d2d(X.bs_pos, X.bu_pos, (size_t)prep.N * sizeof(int32_t));
    }
    // LDS of a nominate workgroup: the arrays of a one-slot search (SimulatePreemption, 37 of the 38 searches of a cfg 4 head)
    // (with the quota tables when that stays within half a CU's LDS; what does not fit goes to the spill space array by array)
    size_t nom_lds = 0;
    // (fair victim searches: giving k_nominate dynamic LDS for the search's small state — make_search can place it there — was measured
    // at cfg 4f and lost: 50 KB per workgroup leaves 2 resident waves per CU instead of 16 and the pass went from 6.1 to 9.4 s,
    // profiles/r02i_bench_cfg4f_lds_state.json. Only k_process_fair, which has the LDS anyway, uses the placement.)
    if (k.C.cs_on) {
      nom_lds = cs_bytes(1, prep.cs_max_bucket, prep.max_tree_nodes, prep.max_tree_cqs, true);
      if (nom_lds > 78 * 1024) nom_lds = std::min<size_t>(cs_bytes(1, prep.cs_max_bucket, prep.max_tree_nodes, prep.max_tree_cqs, false), 78 * 1024);
    }
    // fair sharing: the whole state of a victim search (kq_fs.hpp) in LDS — one search per CU, but a pop costs LDS round trips
    // instead of ~30 dependent HBM/L2 accesses. A one-slot search caches the nR columns of its flavor.
    const size_t fs_want = fs_lds ? fs_bytes(prep.max_tree_nodes, prep.max_tree_cqs, nR, (int)prep.nfr, prep.max_tree_mw, 2 * nR) : 0;
    if (fs_lds) nom_lds = std::min<size_t>(fs_want, be.lds_budget());
    if (sc.mode) { k.shard.x = sc.x; k.shard.world = sc.world; k.shard.rank = sc.rank; k.shard.pool_cap = seg_cap; }
    if (sc.mode == 1) {
      k.shard.mine = nullptr;
      if (sc.mine) { uint8_t* dm = grow<uint8_t>(b_shard, (size_t)n); be.h2d(dm, sc.mine, (size_t)n); k.shard.mine = dm; }
      be.memset(sc.x, 0, shard_words(n, nps, (int)nR, sc.world, rsn_win, seg_cap) * sizeof(int64_t));
    }
    if (sc.mode == 2) be.launch_shard_import(k, nps, rsn_win);  // the merged nomination of every rank's heads
    else {
      // the full pass (victim searches, partial admission, replaced slices) only gets heads the lean pass defers; when nothing of the
      // kind exists in the snapshot and the batch, its launch is skipped
      const bool full_pass = prep.any_preemption || hbch.partial || hbch.H.slice_row != nullptr || hbch.H.ps_group != nullptr || !lean_only_ok;   // (a head with a multi-podset group is deferred by the lean pass)
      if (d_tc) be.launch_nominate_tas(k, slots_nom);   // every head through the full nominate code with the TAS hooks (kq_tas_cycle.hpp)
      else { if (prep_with_k) be.launch_prep_k(pp, k); be.launch_nominate(k, slots_nom, nom_lds, full_pass); }
    }
    if (sc.mode == 1) {
      be.launch_shard_export(k, nps, rsn_win);
      last_cycle_n = -1;
      int32_t derr = 0;   // this rank's own device-side error is reported here with its code (the merged error word only says "some rank failed")
      be.d2h(&derr, k.O.error, sizeof(derr));
      rc = be.sync();     // the caller's all-reduce reads the buffer next
      if (rc != KQ_OK) return fail(rc, be.error());
      if (derr != 0) return fail(derr, derr == KQ_ECAPACITY ? "device-side error: target pool too small for this rank's nominations (tgt_cap)" : "device-side error (capacity or unsupported input)");
      return KQ_OK;
    }
    const bool fuse_ro = B::FUSE_RECORDS_ORDER && !nominate_only && !d_tc && !cfg.fair_sharing;   // (one launch for the two: then charged to the order interval)
    if (!nominate_only && !d_tc && !fuse_ro) be.launch_records(k);  // entry records (static part) for k_process; charged to the nominate interval
    be.timer_mark(1);
    if (fuse_ro) be.launch_records_order(k, order_idx, rank);
    else if (!cfg.fair_sharing && !nominate_only) be.launch_order(k, order_idx, rank);
    be.timer_mark(2);
    k.O.stat_bytes = (long long*)(misc + 2);
    if (nominate_only) {}
    else if (d_tc) { if (tas_n_cls > 0) be.launch_tas_cycle_classes(d_tc, tas_n_cls); be.launch_process_tas(k, tas_lds_want); }   // one wave, every tree, entry order (fair sharing: the iterators of the trees interleaved): TAS leaves are shared across root cohorts
    else if (cfg.fair_sharing) {
      if (n_help > 0) { k.help = d_help; k.help_quit = (uint32_t*)(d_help + prep.n_tree); k.help_trees = prep.n_tree; }
      be.launch_process_fair(k, prep.n_tree, (size_t)prep.max_tree_cohorts * prep.nfr * 16, fs_want, rank);
    }
    else be.launch_process(k, prep.n_tree, (size_t)prep.max_tree_cohorts * prep.nfr * 16);
    be.timer_mark(3);
    last_cycle_n = -1;  // set on the success path only: a failed cycle must not be committable

    if (st) {
      // asynchronous step: the packed decisions, the head count the gather left on the device and (where victims can exist) the target
      // pool go to pinned memory behind an event; nothing here waits for the device
      st->lay = L; st->n_bound = n; st->nps_bound = nps; st->nR = nR; st->pool_cap = O.pool_cap; st->with_pool = prep.any_preemption;
      st->tgt_cap = pool_cap; st->cycle = hbch.cycle;
      size_t so = (pack_bytes + 15) & ~(size_t)15;
      st->o_counts = so; so += 16;
      st->o_prow = so; so += st->with_pool ? (((size_t)O.pool_cap * 4 + 15) & ~(size_t)15) : 0;
      st->o_preason = so; so += st->with_pool ? (((size_t)O.pool_cap + 15) & ~(size_t)15) : 0;
      st->o_headwl = so; so += st->with_heads ? (size_t)pend.nq * 4 : 0;
      so = (so + 15) & ~(size_t)15;
      st->o_rsn = so; so += rsn_win ? (size_t)n * rsn_win * sizeof(RsnRec) : 0;
      if (st->cap < so) { if (st->host) be.free_host(st->host); st->cap = so + so / 4; st->host = (uint8_t*)be.alloc_host(st->cap); }
      be.side_fence();                           // the copies run next to the step's tail kernels, behind everything enqueued so far
      be.d2h_side(st->host, pack, pack_bytes);   // (the head / podset counts ride in the pack: pack_counts)
      if (st->with_pool) { be.d2h_side(st->host + st->o_prow, O.pool_row, (size_t)O.pool_cap * 4); be.d2h_side(st->host + st->o_preason, O.pool_reason, (size_t)O.pool_cap); }
      if (st->with_heads) be.d2h_side(st->host + st->o_headwl, pend.D.head_wl, (size_t)pend.nq * 4);
      if (rsn_win) be.d2h_side(st->host + st->o_rsn, O.rsn, (size_t)n * rsn_win * sizeof(RsnRec));   // (a window per head of the bound: the next step rewrites them)
      be.side_done();
      last_O = k.O; last_slot = slot; pend.O = k.O; pend.H = k.H;
      return KQ_OK;
    }
    // decisions back: one D2H of the packed region into host staging, then plain memcpy to the caller's arrays
    if (hstage_cap < pack_bytes) { if (hstage) be.free_host(hstage); hstage_cap = pack_bytes + pack_bytes / 4; hstage = (uint8_t*)be.alloc_host(hstage_cap); }
    be.d2h(hstage, pack, pack_bytes);
    rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    for (int p = 0; p < 3; p++) last_phase_ms[p] = be.timer_ms(p, p + 1);
    last_kernel_ms = be.timer_ms(0, 3);
    rc = cycle_unpack(L, hstage, n, nps, nR, out, k.O, nullptr, nullptr);
    if (rc != KQ_OK) return rc;
    if (!nominate_only) { last_cycle_n = n; last_O = k.O; last_slot = slot; }
    if (slot == PEND_SLOT && !nominate_only) { pend.O = k.O; pend.H = k.H; pend.ran = true; }
    return KQ_OK;
  }
  // The packed region of a finished cycle (host copy `stg`) -> the caller's kq_decisions. n / nps: heads and podsets of the cycle.
  // prow / preason: the target pool when it is already on the host (asynchronous step), else it is fetched here.
  int cycle_unpack(const PackLayout& L, const uint8_t* stg, int n, size_t nps, size_t nR, kq_decisions* out, const DOut& O, const int32_t* prow_h, const uint8_t* preason_h,
                   const RsnRec* win_h = nullptr) {
    int rc = KQ_OK;
    const int rsn_win = L.rsn_win;
    if (out->tgt_off) out->tgt_off[0] = 0;
    if (out->status) memcpy(out->status, stg + L.o_status, n);
    if (out->action) memcpy(out->action, stg + L.o_action, n);
    if (out->nominated_mode) memcpy(out->nominated_mode, stg + L.o_nmode, n);
    if (out->mode) memcpy(out->mode, stg + L.o_mode, n);
    if (out->requeue_reason) memcpy(out->requeue_reason, stg + L.o_rq, n);
    if (out->skip) memcpy(out->skip, stg + L.o_skip, n);
    if (out->borrowing) memcpy(out->borrowing, stg + L.o_borrow, (size_t)n * 4);
    if (out->order) memcpy(out->order, stg + L.o_order, (size_t)n * 4);
    if (out->flavor) memcpy(out->flavor, stg + L.o_flavor, nps * nR * 4);
    if (out->res_mode) memcpy(out->res_mode, stg + L.o_rmode, nps * nR);
    if (out->tried_idx) memcpy(out->tried_idx, stg + L.o_tried, nps * nR * 4);
    if (out->ps_count) memcpy(out->ps_count, stg + L.o_pscount, nps * 4);
    const int32_t* tpos = (const int32_t*)(stg + L.o_tpos);
    const int32_t* tn = (const int32_t*)(stg + L.o_tn);
    int64_t miscs[4];
    memcpy(miscs, stg + L.o_misc, sizeof(miscs));
    int32_t pool_used = ((int32_t*)miscs)[0], dev_err = ((int32_t*)miscs)[1];
    if (rsn_win && dev_err == 0) {  // reason windows -> the caller's CSR (only the used part of every window is copied out)
      const int32_t* rn = (const int32_t*)(stg + L.o_rsn);
      size_t used_heads = 0;
      for (int i = 0; i < n; i++) if (rn[i] != 0) used_heads++;
      std::vector<RsnRec> win_v;
      if (used_heads && !win_h) { win_v.resize((size_t)n * rsn_win); be.d2h(win_v.data(), O.rsn, win_v.size() * sizeof(RsnRec)); rc = be.sync(); if (rc != KQ_OK) return fail(rc, be.error()); }
      const RsnRec* win = win_h ? win_h : win_v.data();
      int tot = 0;
      for (int i = 0; i < n; i++) {
        if (out->rsn_off) out->rsn_off[i] = tot;
        const int cnt = rn[i] < 0 ? -rn[i] : rn[i];
        for (int q = 0; q < cnt + (rn[i] < 0 ? 1 : 0); q++) {
          if (tot >= out->rsn_cap) return fail(KQ_ECAPACITY, "rsn_cap too small");
          RsnRec r{};
          if (q < cnt) r = win[(size_t)i * rsn_win + q]; else { r.code = 255; r.flavor = -1; r.resource = -1; }  // overflow marker
          if (out->rsn_code) out->rsn_code[tot] = r.code;
          if (out->rsn_podset) out->rsn_podset[tot] = r.podset;
          if (out->rsn_flavor) out->rsn_flavor[tot] = r.flavor;
          if (out->rsn_resource) out->rsn_resource[tot] = r.resource;
          if (out->rsn_a) out->rsn_a[tot] = r.a;
          if (out->rsn_b) out->rsn_b[tot] = r.b;
          if (out->rsn_c) out->rsn_c[tot] = r.c;
          tot++;
        }
      }
      if (out->rsn_off) out->rsn_off[n] = tot;
    } else if (out->rsn_off && out->rsn_cap > 0) {
      memset(out->rsn_off, 0, (size_t)(n + 1) * sizeof(int32_t));
    }
    last_phase_bytes[0] = miscs[1]; last_phase_bytes[1] = miscs[2];
    last_bytes = miscs[1] + miscs[2];
    if (dev_err != 0) return fail(dev_err, "device-side error (capacity or unsupported input)");
    // targets CSR: canonical order inside an entry = ascending admitted row
    if (pool_used == 0) {  // no preemption anywhere in this cycle
      if (out->tgt_off) memset(out->tgt_off, 0, (size_t)(n + 1) * sizeof(int32_t));
      return KQ_OK;
    }
    std::vector<int32_t> prow;
    std::vector<uint8_t> preason;
    if (!prow_h) {
      prow.resize(pool_used); preason.resize(pool_used);
      be.d2h(prow.data(), O.pool_row, (size_t)pool_used * sizeof(int32_t)); be.d2h(preason.data(), O.pool_reason, pool_used);
      rc = be.sync();
      if (rc != KQ_OK) return fail(rc, be.error());
      prow_h = prow.data(); preason_h = preason.data();
    }
    int tot = 0;
    for (int i = 0; i < n; i++) {
      if (out->tgt_off) out->tgt_off[i] = tot;
      std::vector<std::pair<int32_t, uint8_t>> ts;
      for (int t = 0; t < tn[i]; t++) ts.push_back({prow_h[tpos[i] + t], preason_h[tpos[i] + t]});
      std::sort(ts.begin(), ts.end());
      for (auto& t : ts) {
        if (tot >= out->tgt_cap) return fail(KQ_ECAPACITY, "tgt_cap too small");
        if (out->tgt_adm) out->tgt_adm[tot] = t.first;
        if (out->tgt_reason) out->tgt_reason[tot] = t.second;
        tot++;
      }
    }
    if (out->tgt_off) out->tgt_off[n] = tot;
    return KQ_OK;
  }

  // ---- pending side: host orchestration ----------------------------------------------------------------------------
  // the slice columns of the resident set for its first W workloads / nps podsets / nreq requests (h: where they come from, null = none replace a slice)
  void pend_slice_columns(int W, size_t nps, size_t nreq, const kq_heads* h) {
    DHeads& S0 = pend.D.P;
    S0.slice_row = pend_alloc<int32_t>(W, h ? h->slice_row : nullptr, 0xff);
    S0.ps_slice_count = pend_alloc<int32_t>(nps, h ? h->ps_slice_count : nullptr, 0);
    S0.req_slice_flavor = pend_alloc<int32_t>(nreq, h ? h->req_slice_flavor : nullptr, 0xff);
    S0.req_slice_qty = pend_alloc<int64_t>(nreq, h ? h->req_slice_qty : nullptr, 0);
    S0.ps_slice_pods_flavor = pend_alloc<int32_t>(nps, h ? h->ps_slice_pods_flavor : nullptr, 0xff);
    S0.ps_slice_pods_qty = pend_alloc<int64_t>(nps, h ? h->ps_slice_pods_qty : nullptr, 0);
  }
  int pending_put(const kq_pending* p) {
    if (!have_snapshot) return fail(KQ_EINVAL, "kq_pending_put before kq_snapshot_put");
    const kq_heads* h = &p->w;
    int slot_cap = 1, max_nps = 1; bool plain = true;
    int rc = validate_heads(h, &slot_cap, &plain, &max_nps);
    if (rc != KQ_OK) return rc;
    pending_free();
    const int W = h->n, nq = prep.nq, nR = prep.nR;
    const size_t nfw = (prep.nF + 63) / 64;
    const size_t nps = W ? h->ps_off[W] : 0, nreq = nps ? h->ps_req_off[nps] : 0;
    Pending& P = pend;
    P.h_cq.assign(h->cq, h->cq + W); P.h_prio.assign(h->priority, h->priority + W); P.h_ts.assign(h->queue_ts, h->queue_ts + W);
    P.h_uid.resize(W);
    for (int w = 0; w < W; w++) P.h_uid[w] = p->uid_rank ? p->uid_rank[w] : (uint32_t)w;
    // heap order of every ClusterQueue (baseCompareFunc cluster_queue.go:844 without the sticky term): priority descending,
    // queue-order timestamp ascending, UID ascending — static while the workloads are pending
    std::vector<int32_t> ord, cq_off;
    pend_sort(ord, cq_off);
    P.h_ord = ord;
    P.mps.assign(nq, 0); P.mrq.assign(nq, 0);
    for (int w = 0; w < W; w++) {
      const int c = h->cq[w], a = h->ps_off[w + 1] - h->ps_off[w], b = h->ps_req_off[h->ps_off[w + 1]] - h->ps_req_off[h->ps_off[w]];
      P.mps[c] = std::max(P.mps[c], a); P.mrq[c] = std::max(P.mrq[c], b);
    }
    P.nps_total = nps; P.nreq_total = nreq;
    P.W = W; P.nq = nq; P.nR = nR; P.nF = prep.nF; P.n_tree = prep.n_tree; P.slot_cap = slot_cap; P.plain = plain; P.max_nps = max_nps; P.partial = vh_partial;
    DPend& D = P.D;
    D.W = W; D.nq = nq; D.nR = nR; D.nfw = (int)nfw;
    DHeads& S0 = D.P;
    S0.n = W;
    S0.cq = pend_alloc(W, h->cq); S0.priority = pend_alloc(W, h->priority); S0.queue_ts = pend_alloc(W, h->queue_ts);
    S0.flags = pend_alloc(W, h->flags); S0.ps_off = pend_alloc(W + 1, h->ps_off);
    S0.ps_count = pend_alloc(nps, h->ps_count);
    S0.ps_min_count = pend_alloc<int32_t>(nps, h->ps_min_count, 0xff);
    S0.ps_group = pend_alloc<int32_t>(nps, h->ps_group, 0xff);   // (-1 everywhere when the caller has no column)
    P.grouped = heads_grouped(h);
    S0.ps_req_off = pend_alloc(nps + 1, h->ps_req_off);
    S0.req_res = pend_alloc(nreq, h->req_res); S0.req_qty = pend_alloc(nreq, h->req_qty);
    S0.ps_flavor_ok = pend_alloc(nps * nfw, h->ps_flavor_ok);
    S0.ps_last_tried = nullptr; S0.last_generation = nullptr; S0.last_cycle = nullptr; S0.last_hash = nullptr;
    S0.hash = pend_alloc<uint64_t>(W, h->hash, 0);
    S0.slice_row = nullptr; S0.ps_slice_count = nullptr; S0.req_slice_flavor = nullptr; S0.req_slice_qty = nullptr; S0.ps_slice_pods_flavor = nullptr; S0.ps_slice_pods_qty = nullptr;
    if (h->slice_row) pend_slice_columns(W, nps, nreq, h);   // workload slices: absent columns read as "no flavor / nothing requested" (as heads_put)
    D.uid = pend_alloc<uint32_t>(W, P.h_uid.data());
    D.cq_off = pend_alloc(nq + 1, cq_off.data()); D.ord = pend_alloc(W, ord.data());
    D.state = pend_alloc<uint8_t>(W, nullptr, 0);  // WL_ACTIVE
    D.bulk = pend_alloc<uint8_t>(W, nullptr, 0);
    D.mflags = pend_alloc(W, h->flags);
    D.last_tried = pend_alloc<int32_t>(nps * nR, h->ps_last_tried, 0xff);
    D.last_gen = pend_alloc<int64_t>(W, h->last_generation, 0); D.last_cycle = pend_alloc<int64_t>(W, h->last_cycle, 0);
    D.last_hash = pend_alloc<uint64_t>(W, h->last_hash, 0);
    D.pw = pend_alloc<int32_t>(nq, nullptr, 0xff); D.pw_sticky = pend_alloc<uint8_t>(nq, nullptr, 0);
    D.pop_cycle = pend_alloc<int64_t>(nq, nullptr, 0); D.qi_cycle = pend_alloc<int64_t>(nq, nullptr, 0xff);
    D.head_wl = pend_alloc<int32_t>(nq, nullptr, 0xff); D.hd = pend_alloc<int32_t>(nq, nullptr, 0); D.hreq = pend_alloc<int32_t>(nq, nullptr, 0);
    D.counts = pend_alloc<int32_t>(4, nullptr, 0);
    D.cq_active = nullptr;
    D.lq = nullptr; D.lq_usage = nullptr; P.n_lq = 0;
    if (p->lq && p->n_lq > 0) {
      for (int w = 0; w < W; w++) if (p->lq[w] < -1 || p->lq[w] >= p->n_lq) { pending_free(); return fail(KQ_EINVAL, "kq_pending: LocalQueue index out of range"); }
      D.lq = pend_alloc(W, p->lq);
      P.d_lq_usage = pend_alloc<double>(p->n_lq, nullptr, 0);
      D.lq_usage = P.d_lq_usage; P.n_lq = p->n_lq;
    }
    D.requeue_at = nullptr; D.now = P.now;
    if (p->requeue_at) D.requeue_at = pend_alloc<int64_t>(W, p->requeue_at);
    P.d_active = pend_alloc<uint8_t>(nq, nullptr, 1);
    P.d_list = pend_alloc<int32_t>(nq, nullptr, 0);
    P.d_tree_stamp = pend_alloc<int32_t>(std::max(prep.n_tree, 1), nullptr, 0);
    P.G = DGather{};
    pend_alloc_gather();
    if (D.requeue_at && W > 0) be.launch_pend_add_fix(D, S, 0, W);  // workloads still backing off start among the inadmissible ones (:414)
    rc = be.sync();
    if (rc != KQ_OK) { pending_free(); return fail(rc, be.error()); }
    P.valid = true; P.n_heads = -1; P.ran = false;
    return KQ_OK;
  }
  int pending_heads(int64_t cycle, const uint8_t* cq_active, int32_t* n_heads, int32_t* n_podsets, int32_t* head_wl) {
    if (!have_snapshot || !pend.valid) return fail(KQ_EINVAL, "kq_pending_heads before kq_pending_put");
    if (pend.n_heads >= 0) return fail(KQ_EINVAL, "kq_pending_heads: the previous heads were not applied (kq_pending_apply)");
    if (steps_issued != steps_waited) return fail(KQ_EINVAL, "kq_pending_heads: an asynchronous step is in flight (kq_pending_step_wait first)");
    Pending& P = pend;
    if (cq_active) { be.h2d(P.d_active, cq_active, P.nq); P.D.cq_active = P.d_active; } else P.D.cq_active = nullptr;
    be.launch_pend_heads(P.D, P.G);
    int32_t counts[4] = {0, 0, 0, 0};
    be.d2h(counts, P.D.counts, sizeof(counts));
    if (head_wl) be.d2h(head_wl, P.D.head_wl, (size_t)P.nq * sizeof(int32_t));
    int rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    P.n_heads = counts[0]; P.n_ps = counts[1]; P.ran = false; P.cycle = cycle;
    if ((int)batches.size() <= PEND_SLOT) batches.resize(PEND_SLOT + 1);
    HeadBatch& hb = batches[PEND_SLOT];
    if (last_slot == PEND_SLOT) last_cycle_n = -1;  // the previous cycle's head arrays were just overwritten
    hb.n = P.n_heads; hb.nps = (size_t)P.n_ps; hb.slot_cap = P.slot_cap; hb.cycle = cycle; hb.valid = true; hb.plain = P.plain; hb.max_nps = P.max_nps; hb.partial = P.partial;
    DHeads& H = hb.H; const DGather& G = P.G;
    H = DHeads{};   // (also drops the device-side count a kq_pending_step left in the slot)
    H.n = P.n_heads; H.cq = G.cq; H.priority = G.priority; H.queue_ts = G.queue_ts; H.flags = G.flags; H.ps_off = G.ps_off;
    H.ps_count = G.ps_count; H.ps_min_count = G.ps_min_count; H.ps_req_off = G.ps_req_off; H.req_res = G.req_res; H.req_qty = G.req_qty;
    H.ps_flavor_ok = G.ps_flavor_ok; H.ps_last_tried = G.ps_last_tried; H.last_generation = G.last_generation; H.last_cycle = G.last_cycle;
    H.last_hash = G.last_hash; H.hash = G.hash; H.ps_group = G.ps_group;
    pend_wire_slices(H);
    if (n_heads) *n_heads = P.n_heads;
    if (n_podsets) *n_podsets = P.n_ps;
    return KQ_OK;
  }
  int cycle_run_pending(kq_decisions* out) {
    if (!pend.valid || pend.n_heads < 0) return fail(KQ_EINVAL, "kq_cycle_run_pending before kq_pending_heads");
    return cycle_exec(PEND_SLOT, out);
  }
  int pending_apply() {
    if (!pend.valid || pend.n_heads < 0) return fail(KQ_EINVAL, "kq_pending_apply: no heads in flight");
    if (pend.n_heads > 0 && !pend.ran) return fail(KQ_EINVAL, "kq_pending_apply: the cycle over these heads did not run");
    if (pend.n_heads > 0) be.launch_pend_apply(pend.D, S, pend.O, pend.H, cfg.gates, pend.cycle, pend.n_heads);
    pend.n_heads = -1; pend.ran = false;
    return KQ_OK;  // stream-ordered with the next kq_pending_heads
  }
  // ---- the pending loop without a host round trip inside a cycle --------------------------------------------------------------
  // kq_pending_step: Heads() -> the cycle -> kq_cycle_commit -> kq_pending_apply (-> kq_cycle_release) enqueued back to back; the head
  // count stays on the device (DHeads::n_dev), arrays and grids are sized by the bound (<= 1 head per ClusterQueue, the widest
  // workload of every ClusterQueue). The decisions are fetched by kq_pending_step_wait, up to two steps later.
  int pending_step(int64_t cycle, const uint8_t* cq_active, int tgt_cap, int release_age, int want_head_wl) {
    if (!have_snapshot || !pend.valid) return fail(KQ_EINVAL, "kq_pending_step before kq_pending_put");
    if (pend.n_heads >= 0) return fail(KQ_EINVAL, "kq_pending_step: heads of kq_pending_heads are in flight (kq_pending_apply first)");
    if (steps_issued - steps_waited >= 2) return fail(KQ_ECAPACITY, "kq_pending_step: two steps in flight, kq_pending_step_wait first");
    if (ring[commits % KQ_COMMIT_RING].live) return fail(KQ_ECAPACITY, "commit ring full: release older commits first");
    if (release_age != 0) {
      if (release_age < 1 || release_age > KQ_COMMIT_RING || release_age > commits + 1) return fail(KQ_EINVAL, "kq_pending_step: no such commit to release");
      if (release_age > 1 && !ring[(commits + 1 - release_age) % KQ_COMMIT_RING].live) return fail(KQ_EINVAL, "kq_pending_step: already released");
    }
    Pending& P = pend;
    StepStage& st = steps[steps_issued & 1];
    be.step_begin((int)(steps_issued & 1));
    // the cohort usage levels (stale since the previous step's commit / release) are re-derived next to Heads(), on a side stream
    if (levels_stale) { be.usage_levels_side(S, d_usage, max_depth()); levels_stale = false; }
    if (cq_active) { be.h2d(P.d_active, cq_active, P.nq); P.D.cq_active = P.d_active; } else P.D.cq_active = nullptr;
    be.launch_pend_heads(P.D, P.G);
    be.usage_join();
    P.n_ps = 0; P.ran = false; P.cycle = cycle;
    if ((int)batches.size() <= PEND_SLOT) batches.resize(PEND_SLOT + 1);
    HeadBatch& hb = batches[PEND_SLOT];
    last_cycle_n = -1;
    hb.n = P.nq; hb.nps = std::max<size_t>(P.gps, 1); hb.slot_cap = P.slot_cap; hb.cycle = cycle; hb.valid = true; hb.plain = P.plain; hb.max_nps = P.max_nps; hb.partial = P.partial;
    DHeads& H = hb.H; const DGather& G = P.G;
    H = DHeads{};
    H.n = P.nq; H.n_dev = P.D.counts;
    H.cq = G.cq; H.priority = G.priority; H.queue_ts = G.queue_ts; H.flags = G.flags; H.ps_off = G.ps_off;
    H.ps_count = G.ps_count; H.ps_min_count = G.ps_min_count; H.ps_req_off = G.ps_req_off; H.req_res = G.req_res; H.req_qty = G.req_qty;
    H.ps_flavor_ok = G.ps_flavor_ok; H.ps_last_tried = G.ps_last_tried; H.last_generation = G.last_generation; H.last_cycle = G.last_cycle;
    H.last_hash = G.last_hash; H.hash = G.hash; H.ps_group = G.ps_group;
    pend_wire_slices(H);
    st.with_heads = want_head_wl != 0;
    kq_decisions caps{};
    caps.tgt_cap = tgt_cap; caps.rsn_cap = step_rsn_cap;
    int rc = cycle_exec(PEND_SLOT, &caps, false, ShardCall{}, &st);
    if (rc == KQ_OK && prep.usage_consistent && !step_unfused) {
      // commit + requeue policy in one launch, the release of the older commit in one more (+ the requeue of the freed trees): the
      // bookkeeping of kq_cycle_commit / kq_cycle_release, the kernels of both fused (k_step_commit_apply, k_step_release)
      const int n = hb.n;           // the bound: rows past the device's count commit nothing
      Committed& c = ring[commits % KQ_COMMIT_RING];
      c.n = n;
      DCommit dc{n, grow<int32_t>(c.cq, n), grow<int32_t>(c.use_n, n), grow<int32_t>(c.use_fr, (size_t)n * KQ_MAXU), grow<int64_t>(c.use_qty, (size_t)n * KQ_MAXU),
                 d_usage, d_big};
      be.launch_step_commit_apply(S, dc, P.D, cfg.gates, cycle);
      levels_stale = true;
      c.live = true; commits++; last_cycle_n = -1;
      if (release_age > 0) {
        Committed& o = ring[(commits - release_age) % KQ_COMMIT_RING];
        if (o.n > 0) {
          DCommit od{o.n, (const int32_t*)o.cq.p, (const int32_t*)o.use_n.p, (const int32_t*)o.use_fr.p, (const int64_t*)o.use_qty.p, d_usage, d_big};
          be.launch_step_release(S, od, P.D, P.d_tree_stamp, ++P.release_seq);
        }
        o.live = false;
      }
    } else if (rc == KQ_OK) {
      last_cycle_n = hb.n;          // the bound: rows past the device's count commit nothing (k_commit_mask)
      rc = cycle_commit(nullptr);   // skipped on the device when the cycle raised its error flag, like the requeue below
      if (rc == KQ_OK) {
        be.launch_pend_apply(P.D, S, pend.O, pend.H, cfg.gates, cycle, hb.n);
        if (release_age > 0) rc = cycle_release(release_age);
      }
    }
    hb.valid = false;               // the batch only exists on the device: kq_cycle_run_pending has nothing to run on
    if (rc != KQ_OK) { be.step_end(); (void)be.sync(); return rc; }
    be.stage_mark();
    be.step_end();
    st.busy = true;
    steps_issued++;
    return KQ_OK;
  }
  int pending_bounds(int32_t* max_heads, int32_t* max_podsets) {
    if (!pend.valid) return fail(KQ_EINVAL, "kq_pending_bounds before kq_pending_put");
    if (max_heads) *max_heads = pend.nq;
    if (max_podsets) *max_podsets = (int32_t)std::max<size_t>(pend.gps, 1);
    return KQ_OK;
  }
  int pending_step_reasons(int rsn_cap) {
    if (rsn_cap < 0) return fail(KQ_EINVAL, "kq_pending_step_reasons: negative rsn_cap");
    step_rsn_cap = rsn_cap;
    return KQ_OK;
  }
  int pending_step_wait(kq_decisions* out, int32_t* n_heads, int32_t* n_podsets, int32_t* head_wl) {
    if (steps_waited == steps_issued) return fail(KQ_EINVAL, "kq_pending_step_wait: no step in flight");
    StepStage& st = steps[steps_waited & 1];
    be.stage_select((int)(steps_waited & 1));
    int rc = be.stage_wait();
    st.busy = false; steps_waited++;
    if (rc != KQ_OK) { be.stage_select(0); return fail(rc, be.error()); }
    for (int p = 0; p < 3; p++) last_phase_ms[p] = be.timer_ms(p, p + 1);
    last_kernel_ms = be.timer_ms(0, 3);
    be.stage_select(0);
    const int32_t* counts = (const int32_t*)(st.host + st.lay.o_misc) + 6;
    const int n = counts[0]; const size_t nps = (size_t)counts[1];
    if (n < 0 || n > st.n_bound || nps > st.nps_bound) return fail(KQ_EDEVICE, "kq_pending_step_wait: head count outside the bound the step was sized for");
    if (n_heads) *n_heads = n;
    if (n_podsets) *n_podsets = (int32_t)nps;
    if (head_wl) { if (!st.with_heads) return fail(KQ_EINVAL, "kq_pending_step_wait: the step was issued without want_head_wl"); memcpy(head_wl, st.host + st.o_headwl, (size_t)pend.nq * 4); }
    if (!out) return KQ_OK;
    if (!st.with_pool) {  // no ClusterQueue can preempt: a target would have nowhere to be fetched from once the next step runs
      int64_t miscs[4]; memcpy(miscs, st.host + st.lay.o_misc, sizeof(miscs));
      if (((int32_t*)miscs)[0] != 0 && ((int32_t*)miscs)[1] == 0) return fail(KQ_EDEVICE, "kq_pending_step_wait: targets in a snapshot without preemption");
    }
    kq_decisions o = *out;
    PackLayout lay = st.lay;
    if (lay.rsn_win == 0) { o.rsn_cap = 0; o.rsn_off = nullptr; }   // the step was issued without reason records (kq_pending_step_reasons)
    else if (o.rsn_cap <= 0) lay.rsn_win = 0;                          // ... or the caller does not want them this time
    return cycle_unpack(lay, st.host, n, nps, st.nR, &o, pend.O, st.with_pool ? (const int32_t*)(st.host + st.o_prow) : nullptr,
                        st.with_pool ? st.host + st.o_preason : nullptr, lay.rsn_win ? (const RsnRec*)(st.host + st.o_rsn) : nullptr);
  }
  // ---- AdmissionFairSharing ledger (kq_pending.hpp DAfs) ----
  int pending_afs_put(const kq_afs_ledger* l) {
    if (!pend.valid) return fail(KQ_EINVAL, "kq_pending_afs_put before kq_pending_put");
    if (pend.n_heads >= 0) return fail(KQ_EINVAL, "kq_pending_afs_put between kq_pending_heads and kq_pending_apply");
    Pending& P = pend;
    if (!l || l->n_lq != P.n_lq || P.n_lq <= 0) return fail(KQ_EINVAL, "kq_pending_afs_put: LocalQueue count differs from kq_pending_put");
    if (l->n_res <= 0 || l->n_res > 64) return fail(KQ_EINVAL, "kq_pending_afs_put: 1..64 ledger resources");
    if (!l->lq_weight || !l->res_weight || !l->consumed_lo || !l->consumed_hi || !l->wl_penalty_lo || !l->wl_penalty_hi || !l->wl_penalty_mask)
      return fail(KQ_EINVAL, "kq_pending_afs_put: missing array");
    if ((l->penalty_lo != nullptr) != (l->penalty_hi != nullptr) || (l->penalty_lo != nullptr) != (l->penalty_present != nullptr))
      return fail(KQ_EINVAL, "kq_pending_afs_put: penalty_lo / penalty_hi / penalty_present go together");
    const size_t cells = (size_t)l->n_lq * l->n_res, wc = (size_t)P.W * l->n_res;
    if (l->n_res < 64) for (int w = 0; w < P.W; w++) if (l->wl_penalty_mask[w] >> l->n_res) return fail(KQ_EINVAL, "kq_pending_afs_put: penalty mask names a resource outside the ledger");
    DAfs& A = P.D.A;
    if (A.n_res > 0) {  // a previous ledger: its arrays go
      const void* old[] = {A.lq_weight, A.res_weight, A.cons_lo, A.cons_hi, A.cons_f64, A.pen_lo, A.pen_hi, A.pen_present, A.wl_lo, A.wl_hi, A.wl_mask, A.wl_rec};
      for (const void* q : old) pend_release_ptr(q);
    }
    A = DAfs{};
    A.n_lq = l->n_lq; A.n_res = l->n_res;
    A.lq_weight = pend_alloc(l->n_lq, l->lq_weight); A.res_weight = pend_alloc(l->n_res, l->res_weight);
    A.cons_lo = pend_alloc(cells, l->consumed_lo); A.cons_hi = pend_alloc(cells, l->consumed_hi);
    A.cons_f64 = pend_alloc<double>(cells, l->consumed_f64, 0);   // NULL: the scale-9 form, evaluated by the device below
    A.pen_lo = pend_alloc<uint64_t>(cells, l->penalty_lo, 0); A.pen_hi = pend_alloc<int64_t>(cells, l->penalty_hi, 0);
    A.pen_present = pend_alloc<uint8_t>(cells, l->penalty_present, 0);
    A.wl_lo = pend_alloc(wc, l->wl_penalty_lo); A.wl_hi = pend_alloc(wc, l->wl_penalty_hi); A.wl_mask = pend_alloc((size_t)P.W, l->wl_penalty_mask);
    A.wl_rec = pend_alloc<uint8_t>((size_t)P.W, nullptr, 0);
    A.usage = P.d_lq_usage;
    be.launch_afs_usage(P.D, l->consumed_f64 == nullptr);
    int rc = be.sync();   // the caller's arrays may go away
    if (rc != KQ_OK) { A.n_res = 0; return fail(rc, be.error()); }
    return KQ_OK;
  }
  int pending_afs_guard(const char* what) {
    if (!pend.valid || pend.D.A.n_res <= 0) return fail(KQ_EINVAL, (std::string(what) + " before kq_pending_afs_put").c_str());
    if (pend.n_heads >= 0) return fail(KQ_EINVAL, (std::string(what) + " between kq_pending_heads and kq_pending_apply").c_str());
    return KQ_OK;
  }
  template <class T> T* afs_tmp(std::vector<void*>& tmp, const T* host, size_t n) {
    T* d = (T*)be.alloc(std::max<size_t>(n, 1) * sizeof(T));
    if (host && n) be.h2d(d, host, n * sizeof(T));
    tmp.push_back(d);
    return d;
  }
  int afs_finish(std::vector<void*>& tmp) {
    int rc = be.sync();
    for (void* q : tmp) be.free(q);
    return rc != KQ_OK ? fail(rc, be.error()) : KQ_OK;
  }
  int pending_afs_wl_penalty(int n, const int32_t* wl, const uint64_t* lo, const int64_t* hi, const uint64_t* mask) {
    int rc = pending_afs_guard("kq_pending_afs_wl_penalty");
    if (rc != KQ_OK) return rc;
    if (n <= 0) return KQ_OK;
    const DAfs& A = pend.D.A;
    if (!wl || !lo || !hi || !mask) return fail(KQ_EINVAL, "kq_pending_afs_wl_penalty: missing array");
    for (int i = 0; i < n; i++) {
      if (wl[i] < 0 || wl[i] >= pend.W) return fail(KQ_EINVAL, "kq_pending_afs_wl_penalty: workload out of range");
      if (A.n_res < 64 && (mask[i] >> A.n_res)) return fail(KQ_EINVAL, "kq_pending_afs_wl_penalty: penalty mask names a resource outside the ledger");
    }
    // a recorded penalty keeps the amount it was pushed with (penaltyRecords holds a copy): refuse to change it under a record
    std::vector<uint8_t> rec((size_t)pend.W);
    be.d2h(rec.data(), A.wl_rec, (size_t)pend.W);
    rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    for (int i = 0; i < n; i++) if (rec[wl[i]]) return fail(KQ_EINVAL, "kq_pending_afs_wl_penalty: the workload has a pending penalty record");
    for (int i = 0; i < n; i++) {
      be.h2d(A.wl_lo + (size_t)wl[i] * A.n_res, lo + (size_t)i * A.n_res, (size_t)A.n_res * 8);
      be.h2d(A.wl_hi + (size_t)wl[i] * A.n_res, hi + (size_t)i * A.n_res, (size_t)A.n_res * 8);
      be.h2d(A.wl_mask + wl[i], mask + i, 8);
    }
    rc = be.sync();
    return rc != KQ_OK ? fail(rc, be.error()) : KQ_OK;
  }
  int pending_afs_sub_penalty(int n, const int32_t* wl) {
    int rc = pending_afs_guard("kq_pending_afs_sub_penalty");
    if (rc != KQ_OK) return rc;
    if (n <= 0) return KQ_OK;
    std::vector<int32_t> seen((size_t)n);
    for (int i = 0; i < n; i++) { if (wl[i] < 0 || wl[i] >= pend.W) return fail(KQ_EINVAL, "kq_pending_afs_sub_penalty: workload out of range"); seen[i] = wl[i]; }
    std::vector<void*> tmp;
    const int32_t* dl = afs_tmp(tmp, wl, (size_t)n);
    be.launch_afs_sub(pend.D, dl, n);   // one thread, list order: two workloads of one LocalQueue must not race on its row
    return afs_finish(tmp);
  }
  int pending_afs_set_consumed(int n, const int32_t* lq, const uint64_t* lo, const int64_t* hi, const double* f64, const int32_t* settle) {
    int rc = pending_afs_guard("kq_pending_afs_set_consumed");
    if (rc != KQ_OK) return rc;
    if (n <= 0) return KQ_OK;
    const DAfs& A = pend.D.A;
    if (!lq || !lo || !hi) return fail(KQ_EINVAL, "kq_pending_afs_set_consumed: missing array");
    std::vector<uint8_t> hit((size_t)A.n_lq, 0);
    for (int i = 0; i < n; i++) {
      if (lq[i] < 0 || lq[i] >= A.n_lq) return fail(KQ_EINVAL, "kq_pending_afs_set_consumed: LocalQueue out of range");
      if (hit[lq[i]]++) return fail(KQ_EINVAL, "kq_pending_afs_set_consumed: a LocalQueue is listed twice");
      if (settle && (settle[i] < -1 || settle[i] >= pend.W)) return fail(KQ_EINVAL, "kq_pending_afs_set_consumed: workload out of range");
    }
    std::vector<void*> tmp;
    const size_t cells = (size_t)n * A.n_res;
    const int32_t* dl = afs_tmp(tmp, lq, (size_t)n);
    const uint64_t* dlo = afs_tmp(tmp, lo, cells); const int64_t* dhi = afs_tmp(tmp, hi, cells);
    const double* df = f64 ? afs_tmp(tmp, f64, cells) : nullptr;
    const int32_t* ds = settle ? afs_tmp(tmp, settle, (size_t)n) : nullptr;
    be.launch_afs_set_consumed(pend.D, dl, dlo, dhi, df, ds, n);
    return afs_finish(tmp);
  }
  int pending_afs_read(double* usage, uint64_t* plo, int64_t* phi, uint8_t* ppres, uint64_t* clo, int64_t* chi, uint8_t* wrec) {
    if (!pend.valid || pend.D.A.n_res <= 0) return fail(KQ_EINVAL, "kq_pending_afs_read before kq_pending_afs_put");
    const DAfs& A = pend.D.A;
    const size_t cells = (size_t)A.n_lq * A.n_res;
    if (usage) be.d2h(usage, A.usage, (size_t)A.n_lq * sizeof(double));
    if (plo) be.d2h(plo, A.pen_lo, cells * 8);
    if (phi) be.d2h(phi, A.pen_hi, cells * 8);
    if (ppres) be.d2h(ppres, A.pen_present, cells);
    if (clo) be.d2h(clo, A.cons_lo, cells * 8);
    if (chi) be.d2h(chi, A.cons_hi, cells * 8);
    if (wrec) be.d2h(wrec, A.wl_rec, (size_t)pend.W);
    int rc = be.sync();
    return rc != KQ_OK ? fail(rc, be.error()) : KQ_OK;
  }
  int pending_set_lq_usage(int n_lq, const double* usage) {
    if (!pend.valid) return fail(KQ_EINVAL, "kq_pending_set_lq_usage before kq_pending_put");
    if (pend.D.A.n_res > 0) return fail(KQ_EINVAL, "kq_pending_set_lq_usage: an AdmissionFairSharing ledger is resident (kq_pending_afs_put) — the usage is the ledger's");
    if (n_lq != pend.n_lq || (n_lq > 0 && !usage)) return fail(KQ_EINVAL, "kq_pending_set_lq_usage: LocalQueue count differs from kq_pending_put");
    if (n_lq == 0) return KQ_OK;
    be.h2d(pend.d_lq_usage, usage, (size_t)n_lq * sizeof(double));
    return be.sync();  // the caller's array may go away
  }
  // PushOrUpdate of new workloads (cluster_queue.go:379): appended, merged into the heap orders, resident state carried over
  // Arrivals travel as ONE packed upload: every column's tail is laid out in a pinned staging buffer, copied to the device once and
  // scattered to the ends of the resident columns by k_prep (copy / fill operations of 4-byte words) — it was ~25 small uploads.
  struct AddStage { uint8_t* host = nullptr; uint8_t* dev = nullptr; size_t off = 0, cap = 0; std::vector<DPrepOp> ops; };
  template <class T> void add_col(AddStage& a, T*& d, size_t old_n, size_t add_n, const T* tail, int fill = -2) {
    static_assert(sizeof(T) % 4 == 0, "staged columns are made of 4-byte words");
    pend_regrow(d, old_n, add_n, (const T*)nullptr, -2);   // room only (the resident part moves device-to-device if it has to)
    if (!add_n) return;
    const size_t bytes = add_n * sizeof(T);
    if (tail) {
      memcpy(a.host + a.off, tail, bytes);
      a.ops.push_back(DPrepOp{d + old_n, a.dev + a.off, (uint32_t)(bytes / 4), 0});
      a.off += (bytes + 15) & ~(size_t)15;
    } else if (fill != -2) a.ops.push_back(DPrepOp{d + old_n, nullptr, (uint32_t)(bytes / 4), fill == 0 ? 0u : 0xffffffffu});
  }
  template <class T> void add_col(AddStage& a, const T*& d, size_t old_n, size_t add_n, const T* tail, int fill = -2) {
    T* m = const_cast<T*>(d);
    add_col(a, m, old_n, add_n, tail, fill);
    d = m;
  }
  int pending_add(const kq_pending* p, int32_t* first_index) {
    if (!have_snapshot || !pend.valid) return fail(KQ_EINVAL, "kq_pending_add before kq_pending_put");
    if (pend.n_heads >= 0) return fail(KQ_EINVAL, "kq_pending_add between kq_pending_heads and kq_pending_apply");
    if (steps_issued != steps_waited) return fail(KQ_EINVAL, "kq_pending_add: an asynchronous step is in flight (kq_pending_step_wait first)");
    const kq_heads* h = &p->w;
    int slot_cap = 1, max_nps = 1; bool plain = true;
    int rc = validate_heads(h, &slot_cap, &plain, &max_nps);
    if (rc != KQ_OK) return rc;
    Pending& P = pend;
    const int n = h->n, W0 = P.W, nq = prep.nq, nR = prep.nR;
    if (first_index) *first_index = W0;
    if (n == 0) return KQ_OK;
    if ((P.n_lq > 0) != (p->lq != nullptr && p->n_lq > 0) || (P.n_lq > 0 && p->n_lq != P.n_lq)) return fail(KQ_EINVAL, "kq_pending_add: LocalQueue indices must match kq_pending_put");
    if (P.n_lq > 0) for (int w = 0; w < n; w++) if (p->lq[w] < -1 || p->lq[w] >= P.n_lq) return fail(KQ_EINVAL, "kq_pending: LocalQueue index out of range");
    const size_t nfw = (prep.nF + 63) / 64;
    const size_t aps = (size_t)h->ps_off[n], arq = aps ? (size_t)h->ps_req_off[aps] : 0;
    const size_t nps0 = P.nps_total, nrq0 = P.nreq_total;
    DPend& D = P.D;
    DHeads& S0 = D.P;
    AddStage a;
    a.cap = (size_t)n * 104 + aps * (48 + 8 * nfw + 4 * (size_t)nR) + arq * 32 + (size_t)(n + nq + 2) * 8 + 96 * 16;
    if (hup_cap < a.cap) { if (hup) be.free_host(hup); hup_cap = a.cap + a.cap / 4; hup = (uint8_t*)be.alloc_host(hup_cap); }
    a.host = hup;
    a.dev = grow<uint8_t>(b_addstage, a.cap);
    add_col(a, S0.cq, W0, n, h->cq); add_col(a, S0.priority, W0, n, h->priority); add_col(a, S0.queue_ts, W0, n, h->queue_ts);
    add_col(a, S0.flags, W0, n, h->flags);
    std::vector<int32_t> t_ps(n), t_rq(aps);
    for (int i = 0; i < n; i++) t_ps[i] = (int32_t)(nps0 + h->ps_off[i + 1]);
    add_col(a, S0.ps_off, (size_t)W0 + 1, n, t_ps.data());
    add_col(a, S0.ps_count, nps0, aps, h->ps_count);
    add_col(a, S0.ps_min_count, nps0, aps, h->ps_min_count, 0xff);
    add_col(a, S0.ps_group, nps0, aps, h->ps_group, 0xff);
    if (!P.grouped && heads_grouped(h)) { P.grouped = true; pend_alloc_gather(); }   // the first arrivals with a PodSetGroupName group
    for (size_t i = 0; i < aps; i++) t_rq[i] = (int32_t)(nrq0 + h->ps_req_off[i + 1]);
    add_col(a, S0.ps_req_off, nps0 + 1, aps, t_rq.data());
    add_col(a, S0.req_res, nrq0, arq, h->req_res); add_col(a, S0.req_qty, nrq0, arq, h->req_qty);
    add_col(a, S0.ps_flavor_ok, nps0 * nfw, aps * nfw, h->ps_flavor_ok);
    add_col(a, S0.hash, W0, n, h->hash, 0);
    if (h->slice_row && !S0.slice_row) { pend_slice_columns(W0, nps0, nrq0, nullptr); pend_alloc_gather(); }   // the first arrivals that replace a slice
    if (S0.slice_row) {
      add_col(a, S0.slice_row, W0, n, h->slice_row, 0xff);
      add_col(a, S0.ps_slice_count, nps0, aps, h->slice_row ? h->ps_slice_count : nullptr, 0);
      add_col(a, S0.req_slice_flavor, nrq0, arq, h->slice_row ? h->req_slice_flavor : nullptr, 0xff);
      add_col(a, S0.req_slice_qty, nrq0, arq, h->slice_row ? h->req_slice_qty : nullptr, 0);
      add_col(a, S0.ps_slice_pods_flavor, nps0, aps, h->slice_row ? h->ps_slice_pods_flavor : nullptr, 0xff);
      add_col(a, S0.ps_slice_pods_qty, nps0, aps, h->slice_row ? h->ps_slice_pods_qty : nullptr, 0);
    }
    std::vector<uint32_t> uid(n);
    for (int w = 0; w < n; w++) uid[w] = p->uid_rank ? p->uid_rank[w] : (uint32_t)(W0 + w);
    add_col(a, D.uid, W0, n, uid.data());
    pend_regrow(D.state, W0, n, (const uint8_t*)nullptr, 0); pend_regrow(D.bulk, W0, n, (const uint8_t*)nullptr, 0);   // byte columns: memset
    add_col(a, D.mflags, W0, n, h->flags);
    add_col(a, D.last_tried, nps0 * nR, aps * nR, h->ps_last_tried, 0xff);
    add_col(a, D.last_gen, W0, n, h->last_generation, 0); add_col(a, D.last_cycle, W0, n, h->last_cycle, 0);
    add_col(a, D.last_hash, W0, n, h->last_hash, 0);
    std::vector<int64_t> t_at;
    if (D.requeue_at || p->requeue_at) {
      if (!D.requeue_at) { t_at.assign((size_t)W0, KQ_REQUEUE_NONE); D.requeue_at = pend_alloc<int64_t>(W0, t_at.data()); rc = be.sync(); if (rc != KQ_OK) return fail(rc, be.error()); }
      std::vector<int64_t> tail((size_t)n, KQ_REQUEUE_NONE);
      if (p->requeue_at) tail.assign(p->requeue_at, p->requeue_at + n);
      add_col(a, D.requeue_at, W0, n, tail.data());   // (copied into the staging buffer right here)
    }
    if (P.n_lq > 0) add_col(a, D.lq, W0, n, p->lq);
    if (D.A.n_res > 0) {  // arrivals carry no entry penalty until kq_pending_afs_wl_penalty
      add_col(a, D.A.wl_lo, (size_t)W0 * D.A.n_res, (size_t)n * D.A.n_res, (const uint64_t*)nullptr, 0);
      add_col(a, D.A.wl_hi, (size_t)W0 * D.A.n_res, (size_t)n * D.A.n_res, (const int64_t*)nullptr, 0);
      add_col(a, D.A.wl_mask, W0, n, (const uint64_t*)nullptr, 0); pend_regrow(D.A.wl_rec, W0, n, (const uint8_t*)nullptr, 0);
    }
    for (int w = 0; w < n; w++) {
      const int c = h->cq[w], x = h->ps_off[w + 1] - h->ps_off[w], b = h->ps_req_off[h->ps_off[w + 1]] - h->ps_req_off[h->ps_off[w]];
      P.mps[c] = std::max(P.mps[c], x); P.mrq[c] = std::max(P.mrq[c], b);
    }
    P.W = W0 + n; D.W = P.W; S0.n = P.W;
    P.nps_total = nps0 + aps; P.nreq_total = nrq0 + arq;
    P.slot_cap = std::max(P.slot_cap, slot_cap); P.plain = P.plain && plain; P.max_nps = std::max(P.max_nps, max_nps); P.partial = P.partial || vh_partial;
    // The heap orders are merged ON THE DEVICE (round 3; it was a host merge over all W workloads + a re-upload of the whole order):
    // the host only sorts the n arrivals among themselves — by ClusterQueue, then baseCompareFunc's static part — and hands over that
    // list with its per-ClusterQueue offsets; a resident workload moves back by the arrivals that sort before it, an arrival lands
    // behind the resident workloads that sort before it (two binary searches, kq_pending.hpp pend_merge_*), into the other of two
    // order buffers.
    std::vector<int32_t> fresh(n), fresh_off((size_t)nq + 1, 0);
    {
      auto before = [&](int x, int y) {   // indices into the arrivals
        if (h->cq[x] != h->cq[y]) return h->cq[x] < h->cq[y];
        if (h->priority[x] != h->priority[y]) return h->priority[x] > h->priority[y];
        if (h->queue_ts[x] != h->queue_ts[y]) return h->queue_ts[x] < h->queue_ts[y];
        if (uid[x] != uid[y]) return uid[x] < uid[y];
        return x < y;
      };
      std::vector<int32_t> idx(n);
      for (int i = 0; i < n; i++) idx[i] = i;
      std::sort(idx.begin(), idx.end(), before);
      for (int i = 0; i < n; i++) { fresh[i] = W0 + idx[i]; fresh_off[h->cq[idx[i]] + 1]++; }
      for (int c = 0; c < nq; c++) fresh_off[c + 1] += fresh_off[c];
    }
    int32_t* d_fresh = nullptr; int32_t* d_fresh_off = nullptr;
    {  // (two more pieces of the same staged upload; they are read by the merge, not scattered)
      memcpy(a.host + a.off, fresh.data(), (size_t)n * 4); d_fresh = (int32_t*)(a.dev + a.off); a.off += ((size_t)n * 4 + 15) & ~(size_t)15;
      memcpy(a.host + a.off, fresh_off.data(), (size_t)(nq + 1) * 4); d_fresh_off = (int32_t*)(a.dev + a.off); a.off += ((size_t)(nq + 1) * 4 + 15) & ~(size_t)15;
    }
    if (a.off > a.cap) return fail(KQ_ENOMEM, "kq_pending_add: staging buffer undersized");
    be.h2d(a.dev, a.host, a.off);
    for (size_t o = 0; o < a.ops.size(); o += 16) {
      DPrep pp{};
      for (size_t q = o; q < a.ops.size() && q < o + 16; q++) pp.op[pp.n++] = a.ops[q];
      be.launch_prep(pp);
    }
    // the other order buffer / offsets buffer (allocated with head room like every resident column)
    int32_t* ord_new = P.ord_alt; int32_t* off_new = P.cq_off_alt;
    pend_regrow(ord_new, 0, (size_t)P.W, (const int32_t*)nullptr, -2);
    if (!off_new) off_new = pend_alloc<int32_t>((size_t)nq + 1, nullptr, 0);
    be.launch_pend_merge(D, D.ord, D.cq_off, ord_new, off_new, d_fresh, d_fresh_off, W0, n);
    P.ord_alt = const_cast<int32_t*>(D.ord); P.cq_off_alt = const_cast<int32_t*>(D.cq_off);
    D.ord = ord_new; D.cq_off = off_new;
    {
      size_t gps = 0, grq = 0;
      for (int c = 0; c < nq; c++) { gps += P.mps[c]; grq += P.mrq[c]; }
      if (gps > P.gps || grq > P.grq) pend_alloc_gather();
    }
    be.launch_pend_add_fix(D, S, W0, n);
    rc = be.sync();   // the staging buffer is reused by the next call
    if (rc != KQ_OK) return fail(rc, be.error());
    return KQ_OK;
  }
  // kq_pending_update (include/kq_engine.h): the replacements are appended and placed like arrivals, then the old records hand over
  int pending_update(int n, const int32_t* wl, const kq_pending* more, int32_t* first_index) {
    if (!have_snapshot || !pend.valid) return fail(KQ_EINVAL, "kq_pending_update before kq_pending_put");
    if (pend.n_heads >= 0) return fail(KQ_EINVAL, "kq_pending_update between kq_pending_heads and kq_pending_apply");
    if (!more || more->w.n != n || n < 0 || (n > 0 && !wl)) return fail(KQ_EINVAL, "kq_pending_update: one replacement per listed workload");
    {
      std::vector<int32_t> seen(wl, wl + n);
      std::sort(seen.begin(), seen.end());
      for (int i = 0; i < n; i++) if (seen[i] < 0 || seen[i] >= pend.W || (i > 0 && seen[i] == seen[i - 1])) return fail(KQ_EINVAL, "kq_pending_update: workload out of range or repeated");
    }
    int32_t first = pend.W;
    int rc = pending_add(more, &first);
    if (first_index) *first_index = first;
    if (rc != KQ_OK || n == 0) return rc;
    // the old indices and (optionally) the "same Generation" bytes behind them, one allocation
    const size_t lb = ((size_t)n * sizeof(int32_t) + 15) & ~(size_t)15;
    unsigned char* d = (unsigned char*)be.alloc(lb + (size_t)n);
    be.h2d(d, wl, (size_t)n * sizeof(int32_t));
    if (more->same_generation) be.h2d(d + lb, more->same_generation, (size_t)n);
    be.launch_pend_update_fix(pend.D, (const int32_t*)d, more->same_generation ? (const uint8_t*)(d + lb) : nullptr, first, n);
    rc = be.sync();
    be.free(d);
    if (rc != KQ_OK) return fail(rc, be.error());
    return KQ_OK;
  }
  int pending_set_clock(int64_t now) { pend.now = now; pend.D.now = now; return KQ_OK; }
  int pending_set_requeue_at(int n, const int32_t* wl, const int64_t* at) {
    if (!have_snapshot || !pend.valid) return fail(KQ_EINVAL, "kq_pending_set_requeue_at before kq_pending_put");
    if (pend.n_heads >= 0) return fail(KQ_EINVAL, "kq_pending_set_requeue_at between kq_pending_heads and kq_pending_apply");
    if (n <= 0) return KQ_OK;
    for (int i = 0; i < n; i++) if (wl[i] < 0 || wl[i] >= pend.W) return fail(KQ_EINVAL, "kq_pending_set_requeue_at: workload out of range");
    DPend& D = pend.D;
    std::vector<int64_t> none;
    if (!D.requeue_at) { none.assign((size_t)pend.W, KQ_REQUEUE_NONE); D.requeue_at = pend_alloc<int64_t>(pend.W, none.data()); }
    int32_t* dl = (int32_t*)be.alloc((size_t)n * sizeof(int32_t));
    int64_t* da = (int64_t*)be.alloc((size_t)n * sizeof(int64_t));
    be.h2d(dl, wl, (size_t)n * sizeof(int32_t)); be.h2d(da, at, (size_t)n * sizeof(int64_t));
    be.launch_pend_requeue_at(D, S, dl, da, n);
    int rc = be.sync();
    be.free(dl); be.free(da);
    if (rc != KQ_OK) return fail(rc, be.error());
    return KQ_OK;
  }
  int pending_delete(int n, const int32_t* wl) {
    if (!pend.valid) return fail(KQ_EINVAL, "kq_pending_delete before kq_pending_put");
    if (pend.n_heads >= 0) return fail(KQ_EINVAL, "kq_pending_delete between kq_pending_heads and kq_pending_apply");
    if (n <= 0) return KQ_OK;
    for (int i = 0; i < n; i++) if (wl[i] < 0 || wl[i] >= pend.W) return fail(KQ_EINVAL, "kq_pending_delete: workload out of range");
    int32_t* d = (int32_t*)be.alloc((size_t)n * sizeof(int32_t));
    be.h2d(d, wl, (size_t)n * sizeof(int32_t));
    be.launch_pend_delete(pend.D, d, n);
    int rc = be.sync();
    be.free(d);
    if (rc != KQ_OK) return fail(rc, be.error());
    return KQ_OK;
  }
  int pending_queue_inadmissible(int n, const int32_t* cq) {
    if (!pend.valid) return fail(KQ_EINVAL, "kq_pending_queue_inadmissible before kq_pending_put");
    if (cq) {
      if (n < 0 || n > pend.nq) return fail(KQ_EINVAL, "bad ClusterQueue list");
      for (int i = 0; i < n; i++) if (cq[i] < 0 || cq[i] >= pend.nq) return fail(KQ_EINVAL, "ClusterQueue out of range");
      if (n == 0) return KQ_OK;
      be.h2d(pend.d_list, cq, (size_t)n * sizeof(int32_t));
      be.launch_pend_qi(pend.D, pend.d_list, n);
      return be.sync();  // the caller's list may go away
    }
    be.launch_pend_qi(pend.D, nullptr, pend.nq);
    return KQ_OK;
  }
  int pending_read_state(uint8_t* state, int32_t* counts) {
    if (!pend.valid) return fail(KQ_EINVAL, "kq_pending_read_state before kq_pending_put");
    std::vector<uint8_t> st(std::max(pend.W, 1));
    be.d2h(st.data(), pend.D.state, pend.W);
    int rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    if (state) memcpy(state, st.data(), pend.W);
    if (counts) { counts[0] = counts[1] = counts[2] = counts[3] = 0; for (int w = 0; w < pend.W; w++) counts[st[w] & 3]++; }
    return KQ_OK;
  }

  // ---- sharded single-root cycles ------------------------------------------------------------------------------------
  // After a cycle: what it added to every usage cell (device buffer of the caller, [N * nfr]) and the certificate of K::root_margin.
  int cycle_certificate(int64_t* delta_dev, int64_t* margin, int32_t* flags) {
    if (!have_snapshot || !b_usage_work.p || !b_cert.p) return fail(KQ_EINVAL, "kq_cycle_certificate before a cycle");
    const size_t cells = (size_t)prep.N * prep.nfr;
    if (delta_dev) be.launch_usage_delta(delta_dev, (const int64_t*)b_usage_work.p, d_usage, cells);
    const size_t mc = (size_t)std::max(prep.n_tree, 1) * prep.nfr;
    if (margin) be.d2h(margin, b_cert.p, mc * sizeof(int64_t));
    if (flags) be.d2h(flags, (const int64_t*)b_cert.p + mc, (size_t)std::max(prep.n_tree, 1) * sizeof(int32_t));
    int rc = be.sync();
    if (rc != KQ_OK) return fail(rc, be.error());
    return KQ_OK;
  }
  // usage[ClusterQueue cells] += sign * delta (device buffer, [n_cq * nfr]); cohort levels follow from them (flush_levels).
  int snapshot_usage_add(const int64_t* delta_dev, int sign) {
    if (!have_snapshot) return fail(KQ_EINVAL, "kq_snapshot_usage_add before kq_snapshot_put");
    if (!prep.usage_consistent) return fail(KQ_EUNSUPPORTED, "the snapshot's cohort usage is not derived from its children: deltas cannot be folded at the ClusterQueue level");
    be.launch_usage_add(d_usage, delta_dev, (size_t)prep.nq * prep.nfr, sign, d_big);
    levels_stale = true;
    last_cycle_n = -1;
    return be.sync();  // the caller's buffer may be reused
  }

  int prof_read(int64_t* out, bool reset) {
    if (!b_prof.p) { for (int i = 0; i < 64; i++) out[i] = 0; return KQ_OK; }
    be.d2h(out, b_prof.p, 64 * sizeof(int64_t));
    int rc = be.sync();
    if (reset) be.memset(b_prof.p, 0, 64 * sizeof(int64_t));
    return rc;
  }
  // diagnostics of the last cycle's speculative rounds (K::spec_stats): windows, rounds, entries decided, trees handed (partly) back to the
  // serial kernel, items, most rounds of one window, abandoned windows, truncated windows
  int spec_stats(int64_t* out) {
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!b_cqh.p || !have_snapshot || !spec_stats_on) return KQ_OK;
    int32_t v[8];
    be.d2h(v, (int32_t*)b_cqh.p + std::max(prep.nq, 1) + std::max(prep.n_tree, 1), sizeof(v));
    int rc = be.sync();
    for (int i = 0; i < 8; i++) out[i] = v[i];
    return rc;
  }
  int read_usage_work(int64_t* out) {  // tests: snapshot usage after the cycle
    be.d2h(out, b_usage_work.p, (size_t)prep.N * prep.nfr * sizeof(int64_t));
    return be.sync();
  }
};

}  // namespace kq

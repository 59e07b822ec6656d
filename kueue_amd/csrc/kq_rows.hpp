// kq_rows.hpp — the candidate structures of the admitted-workload table, built ON THE DEVICE from the resident row table
// (clusterQueue.updateWorkloadUsage, pkg/cache/scheduler/clusterqueue.go:594, adds / removes ONE workload; the engine's
// derived structures must follow without a host rebuild and a re-upload of everything).
//
// What build_prep (kq_prep.hpp) derives from the rows on the host — candidate rank order per tree, the flavor-resource buckets, their
// level orders, row records, bucket fingerprints — is a handful of stable key sorts plus gathers:
//   rank order   rows by (tree, priority asc, reserve time desc, uid rank asc, row asc): LSD passes uid -> ~rts -> prio -> tree
//   buckets      (row, distinct flavor-resource) entries by (tree, flavor-resource, rank position)
//   level order  bucket entries by (bucket, ancestor at depth l + 1 or "none", evicted first, bucket position), l = 0 .. CS_LEVELS - 1,
//                all levels in one sort
// The sort itself is the backend's (rocPRIM radix sort on the GPU, std::stable_sort in the 1-lane emulation); everything else is the
// cell functions below, one thread per row / entry / bucket. The results are byte-identical to build_prep's (tests/test_rows_device.py).
// Fair sharing's position-order tables (kq_fs.hpp: FsScan / FsApply per position) are one more sort by (tree, ClusterQueue, evicted, rank).
#pragma once
#include "kq_device.hpp"

namespace kq {

// temporary storage of the backend's sort / scan primitives: one per engine (its device, its stream), grown on demand
struct RowsScratch { void* p = nullptr; size_t cap = 0; };

enum { RO_ROW_INIT = 0, RO_KEY_RTS, RO_KEY_PRIO, RO_KEY_TREE, RO_RANK, RO_KEY_ASC, RO_ASC, RO_ENT_FILL, RO_BOUNDS, RO_BUCKET_FILL, RO_BUCKET_SIZE,
       RO_LKEY, RO_LFILL, RO_MOVE_ROW, RO_MOVE_ENT, RO_ADD_ROW, RO_KEY_FS, RO_FS_FILL, RO_EVICT, RO_REMAP, RO_FOLD_PACK };

struct DRows {
  // the row table the structures are built from
  int n, E, nq, nfr, n_tree, N;
  const int32_t *cq_adm_off, *adm_use_off, *adm_use_fr;
  const int64_t *adm_prio, *adm_qts, *adm_rts, *adm_use_qty;
  const uint32_t* adm_uid;
  const uint8_t* adm_flags;
  // static tree data
  const int32_t *tree_of, *depth, *parent, *cq_local, *node_local;
  // work
  uint64_t* key;        // [max(n, E)]
  int32_t* val;         // [max(n, E)]
  int32_t* ent_cnt;     // [n + 1] distinct flavor-resources per row (scanned into ent_off)
  int32_t* ent_off;     // [n + 1]
  int32_t* tree_cnt;    // [n_tree + 1]
  int32_t* bcnt;        // [n_tree * nfr + 1]
  int32_t* scal;        // [4] cs_max_bucket, max_tree_rows
  // outputs
  int32_t *adm_cq, *tree_row_off, *tree_rows, *tree_rows_asc, *rank_pos, *frb_off, *frb, *frbr, *cq_row_bytes;
  AdmRec* adm_rec; AdmRecX* adm_recx;
  CsRec* frec;
  CsEnt* frl[CS_LEVELS];
  uint64_t* frb_sig;
  uint8_t *cs_ok, *fs_ok, *rec_ok;
  int level;            // RO_LKEY / RO_LFILL
  // fair sharing (kq_fs.hpp): the candidates in position order — per tree by ClusterQueue (tree-local index), inside a ClusterQueue
  // evicted first, then candidate rank — with the two records per position the LDS search reads
  FsScan* fs_scan; FsApply* fs_apply;
  const int32_t *path, *plen;
  int nR, cq_bits;
  // kq_snapshot_patch_rows: the move of the kept rows into the new table, and the added rows
  const int32_t *old_cq_off, *new_cq_off, *rm_off, *rm_rows;       // per ClusterQueue: old / new row offsets, its removed rows (ascending) as a CSR
  const int32_t *o_use_off, *o_use_fr; const int64_t *o_prio, *o_qts, *o_rts, *o_use_qty; const uint32_t* o_uid; const uint8_t* o_flags; const int32_t* o_adm_cq;
  int32_t *n_use_off, *n_use_fr, *n_ucnt; int64_t *n_prio, *n_qts, *n_rts, *n_use_qty; uint32_t* n_uid; uint8_t* n_flags;
  int32_t* new_of_old;  // [old n] new index of an old row, -1 = removed
  int32_t* remap;       // RO_REMAP: row indices held elsewhere (the resident pending set's slice_row column) follow the move; a removed row becomes -1
  const int32_t* ev_rows;   // old rows that get KQ_ADM_EVICTED
  int n_old, n_add;
  const int32_t *a_target, *a_use_off, *a_use_fr; const int64_t *a_prio, *a_qts, *a_rts, *a_use_qty; const uint32_t* a_uid; const uint8_t* a_flags;
  // KQ_ROWS_FOLD_USAGE: the removed rows [0, n_rm) and the added rows [n_rm, n_rm + n_add) as the usage rows of a commit (DCommit layout)
  int n_rm; const int32_t* a_cq;
  int32_t *f_cq, *f_use_n, *f_use_fr; int64_t* f_use_qty; int32_t* f_err;
};

KQ_DEV uint64_t ro_bias(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }
KQ_DEV bool ro_first_use(const DRows& R, int row, int e) {
  for (int q = R.adm_use_off[row]; q < e; q++) if (R.adm_use_fr[q] == R.adm_use_fr[e]) return false;
  return true;
}
KQ_DEV int ro_cq_of(const DRows& R, int row) {  // the ClusterQueue whose CSR segment holds the row
  int lo = 0, hi = R.nq;
  while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (R.cq_adm_off[m] <= row) lo = m; else hi = m; }
  // (empty segments share offsets: the last ClusterQueue whose offset is <= row and whose end is > row)
  while (lo + 1 < R.nq && R.cq_adm_off[lo + 1] <= row) lo++;
  return lo;
}
KQ_DEV int ro_anc(const DRows& R, int cq, int l) {  // ancestor of the ClusterQueue at depth l + 1 (level order l), -1 = none
  int n = cq;
  const int dd = l + 1;
  if (R.depth[n] < dd) return -1;
  for (int h = R.depth[n]; h > dd; h--) n = R.parent[n];
  return n;
}

// one row: ClusterQueue, tree count, candidate-record bytes of its ClusterQueue, AdmRec, flags, its distinct flavor-resources
KQ_DEV void ro_row_init(const DRows& R, int r) {
  const int c = ro_cq_of(R, r);
  R.adm_cq[r] = c;
  const int t = R.tree_of[c];   // (rows per tree: the host derives them from the CSR offsets — one counter per tree would serialise every row)
  const int k0 = R.adm_use_off[r], k1 = R.adm_use_off[r + 1];
  atomic_add_i32(&R.cq_row_bytes[c], 32 + 12 * (k1 - k0));
  AdmRec a; AdmRecX x;
  a.prio = R.adm_prio[r]; a.qts = R.adm_qts[r]; a.cq = c; a.flags = (R.adm_flags[r] & KQ_ADM_EVICTED) ? 1u : 0u;
  a.rowbytes = 16 * (R.depth[c] + 1) * (k1 - k0); a.fs_pos = 0;
  int distinct = 0;
  for (int e = k0; e < k1; e++) if (ro_first_use(R, r, e)) distinct++;
  const int nf = adm_rec_fold(a, x, k0, k1, [&](int e) { return R.adm_use_fr[e]; }, [&](int e) { return R.adm_use_qty[e]; });
  if (nf < 0) { R.cs_ok[t] = 0; R.rec_ok[t] = 0; R.fs_ok[t] = 0; }
  R.adm_rec[r] = a; R.adm_recx[r] = x;
  R.ent_cnt[r] = distinct;
  R.val[r] = r;
  R.key[r] = (uint64_t)R.adm_uid[r];
}
KQ_DEV void ro_key_rts(const DRows& R, int i) { R.key[i] = ~ro_bias(R.adm_rts[R.val[i]]); }    // reserve time descending
KQ_DEV void ro_key_prio(const DRows& R, int i) { R.key[i] = ro_bias(R.adm_prio[R.val[i]]); }   // priority ascending
KQ_DEV void ro_key_tree(const DRows& R, int i) { R.key[i] = (uint64_t)R.tree_of[R.adm_cq[R.val[i]]]; }
KQ_DEV void ro_rank(const DRows& R, int i) {
  const int row = R.val[i];
  R.tree_rows[i] = row;
  R.rank_pos[row] = i - R.tree_row_off[R.tree_of[R.adm_cq[row]]];
}
KQ_DEV void ro_key_asc(const DRows& R, int r) { R.key[r] = (uint64_t)R.tree_of[R.adm_cq[r]]; R.val[r] = r; }
KQ_DEV void ro_asc(const DRows& R, int i) { R.tree_rows_asc[i] = R.val[i]; }
// the row's distinct flavor-resources as bucket entries: key (tree, flavor-resource, rank position), value = the row
KQ_DEV void ro_ent_fill(const DRows& R, int r) {
  int o = R.ent_off[r];
  const int t = R.tree_of[R.adm_cq[r]];
  for (int e = R.adm_use_off[r]; e < R.adm_use_off[r + 1]; e++) {
    if (!ro_first_use(R, r, e)) continue;
    const int fr = R.adm_use_fr[e];
    R.key[o] = ((uint64_t)((size_t)t * R.nfr + fr) << 32) | (uint32_t)R.rank_pos[r];
    R.val[o] = r;
    o++;
  }
}
// bucket offsets from the sorted entries: entry j opens every bucket between its predecessor's and its own (no counters, no scan)
KQ_DEV void ro_bounds(const DRows& R, int j) {   // j in [0, E]
  const int nb = R.n_tree * R.nfr;
  const int b1 = j < R.E ? (int)(R.key[j] >> 32) : nb;
  const int b0 = j > 0 ? (int)(R.key[j - 1] >> 32) : -1;
  for (int b = b0 + 1; b <= b1; b++) R.frb_off[b] = j;
}
// one bucket entry; the fingerprint terms of a wave's run of equal buckets are summed in the wave first (the entries are sorted by
// bucket: one atomic per run instead of one per entry — 50 000 atomics on 64 addresses took 290 us)
KQ_DEV void ro_bucket_fill(const DRows& R, int j, bool active) {
  int b = -1; int64_t term = 0;
  if (active) {
    const int row = R.val[j];
    b = (int)(R.key[j] >> 32);
    R.frb[j] = (int32_t)(uint32_t)R.key[j];
    R.frbr[j] = row;
    const AdmRec& a = R.adm_rec[row];
    R.frec[j] = CsRec{a.prio, a.qts, row, R.cq_local[a.cq], a.rowbytes, a.flags};
    term = (int64_t)frb_sig_row(row);
  }
  const int lane = lane_id();
  const int64_t P = wprefix_incl_i64(term);
  const int bp = wshfl_i32(b, lane > 0 ? lane - 1 : 0);
  const bool head = lane == 0 || b != bp;
  const uint64_t heads = wballot(head);
  const uint64_t above = lane < WAVE - 1 ? heads >> (lane + 1) : 0ull;
  const int end = above ? lane + ffs64(above) : WAVE - 1;
  const int64_t upto = wshfl_i64(P, end), before = wshfl_i64(P, lane > 0 ? lane - 1 : 0);
  if (head && b >= 0) atomic_add_i64((long long*)&R.frb_sig[b], (long long)(upto - (lane > 0 ? before : 0)));
}
KQ_DEV void ro_bucket_size(const DRows& R, int b) {
  const int M = R.frb_off[b + 1] - R.frb_off[b];
  atomic_add_i64((long long*)&R.frb_sig[b], (long long)frb_sig_size(M));
  atomic_max_i32(&R.scal[0], M);
}
// level order l of every bucket: (bucket, ancestor at depth l + 1 — none last —, evicted first, bucket position)
KQ_DEV int ro_bucket_of(const DRows& R, int j) {  // the bucket holding global entry j
  int lo = 0, hi = R.n_tree * R.nfr;
  while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (R.frb_off[m] <= j) lo = m; else hi = m; }
  while (lo + 1 < R.n_tree * R.nfr && R.frb_off[lo + 1] <= j) lo++;
  return lo;
}
// all CS_LEVELS level orders in ONE sort over CS_LEVELS * E items (x = level * E + entry): key (level, bucket, ancestor, not evicted);
// the sort is stable and the items of a level arrive in bucket-position order, so the position needs no key bits
KQ_DEV void ro_lkey(const DRows& R, int x) {
  const int l = x / R.E, j = x - l * R.E;
  const int row = R.frbr[j];
  const int b = ro_bucket_of(R, j);
  const int anc = ro_anc(R, R.adm_cq[row], l);
  const uint64_t ev = (R.adm_flags[row] & KQ_ADM_EVICTED) ? 0 : 1;
  R.key[x] = ((uint64_t)l << 44) | ((uint64_t)b << 22) | ((uint64_t)(anc < 0 ? R.N : anc) << 1) | ev;
  R.val[x] = j;
}
KQ_DEV void ro_lfill(const DRows& R, int x) {
  const int l = x / R.E, q = x - l * R.E;   // the items of level l occupy [l * E, (l + 1) * E) after the sort
  const int src = R.val[x];
  const int b = (int)((R.key[x] >> 22) & 0x3fffff);
  const int j = src - R.frb_off[b];
  const int row = R.frbr[src];
  const int cq = R.adm_cq[row];
  const int anc = ro_anc(R, cq, l);
  const int fr = b % R.nfr;
  const int64_t qty = adm_rec_qty(R.adm_rec[row], R.adm_recx, row, fr);
  R.frl[l][q] = CsEnt{j | (R.depth[cq] << 24), anc >= 0 ? R.node_local[anc] : -1, row, anc, qty};
}
KQ_DEV uint32_t ro_hkey(const DRows& R, int row) { return ((R.adm_flags[row] & KQ_ADM_EVICTED) ? 0u : 0x80000000u) | (uint32_t)R.rank_pos[row]; }
KQ_DEV void ro_key_fs(const DRows& R, int r) {
  const int c = R.adm_cq[r];
  R.key[r] = ((uint64_t)R.tree_of[c] << (32 + R.cq_bits)) | ((uint64_t)R.cq_local[c] << 32) | ro_hkey(R, r);
  R.val[r] = r;
}
KQ_DEV void ro_fs_fill(const DRows& R, int q) {   // q = global position = tree_row_off[tree] + position inside the tree
  const int r = R.val[q];
  const AdmRec a = R.adm_rec[r];
  const int c = a.cq;
  FsScan sc{}; FsApply ap{};
  sc.prio = a.prio; sc.qts = a.qts; sc.row = r; sc.cql = (int16_t)R.cq_local[c];
  sc.cbytes = (uint16_t)(32 + 12 * (R.adm_use_off[r + 1] - R.adm_use_off[r]));
  ap.cbytes = sc.cbytes; ap.wide = (a.flags & 2u) ? 1 : 0;
  if (ap.wide) sc.cbytes |= FS_SCAN_WIDE;
  for (int e = 0; e < CS_RFR; e++) { sc.fr[e] = ap.fr[e] = (int16_t)a.fr[e]; ap.qty[e] = a.qty[e]; ap.res[e] = (uint8_t)(a.fr[e] >= 0 ? a.fr[e] % R.nR : 255); }
  for (int l = 0; l < FS_LV; l++) ap.lp[l] = l < R.plen[c] ? (int16_t)R.node_local[R.path[(size_t)c * KQ_MAXD + l]] : (int16_t)-1;
  ap.hkey = ro_hkey(R, r); ap.row = r; ap.plen = (uint8_t)(R.plen[c] < 255 ? R.plen[c] : 255);
  R.fs_scan[q] = sc; R.fs_apply[q] = ap;
  R.adm_rec[r].fs_pos = q - R.tree_row_off[R.tree_of[c]];
}

// ---- kq_snapshot_patch_rows: kept rows move to their new index, added rows land behind the kept rows of their ClusterQueue ----------------
KQ_DEV void ro_move_row(const DRows& R, int r) {   // r = old row
  const int c = R.o_adm_cq[r];
  int before = 0; bool gone = false;
  for (int i = R.rm_off[c]; i < R.rm_off[c + 1]; i++) { if (R.rm_rows[i] < r) before++; else if (R.rm_rows[i] == r) gone = true; }
  if (gone) { R.new_of_old[r] = -1; return; }
  const int nr = R.new_cq_off[c] + (r - R.old_cq_off[c]) - before;
  R.new_of_old[r] = nr;
  R.n_prio[nr] = R.o_prio[r]; R.n_qts[nr] = R.o_qts[r]; R.n_rts[nr] = R.o_rts[r]; R.n_uid[nr] = R.o_uid[r]; R.n_flags[nr] = R.o_flags[r];
  R.n_ucnt[nr] = R.o_use_off[r + 1] - R.o_use_off[r];
}
KQ_DEV void ro_add_row(const DRows& R, int i) {    // i = added row: scalars and its usage-entry count
  const int nr = R.a_target[i];
  R.n_prio[nr] = R.a_prio[i]; R.n_qts[nr] = R.a_qts[i]; R.n_rts[nr] = R.a_rts[i]; R.n_uid[nr] = R.a_uid[i]; R.n_flags[nr] = R.a_flags[i];
  R.n_ucnt[nr] = R.a_use_off[i + 1] - R.a_use_off[i];
}
KQ_DEV void ro_evict(const DRows& R, int i) { const int nr = R.new_of_old[R.ev_rows[i]]; if (nr >= 0) R.n_flags[nr] |= KQ_ADM_EVICTED; }
KQ_DEV void ro_remap(const DRows& R, int i) { const int r = R.remap[i]; if (r >= 0) R.remap[i] = r < R.n_old ? R.new_of_old[r] : -1; }
KQ_DEV void ro_move_ent(const DRows& R, int r) {   // r < n_old: an old row's entries; r >= n_old: added row r - n_old (n_use_off scanned)
  if (r < R.n_old) {
    const int nr = R.new_of_old[r];
    if (nr < 0) return;
    const int o = R.n_use_off[nr];
    for (int e = R.o_use_off[r], q = 0; e < R.o_use_off[r + 1]; e++, q++) { R.n_use_fr[o + q] = R.o_use_fr[e]; R.n_use_qty[o + q] = R.o_use_qty[e]; }
  } else {
    const int i = r - R.n_old, nr = R.a_target[i];
    const int o = R.n_use_off[nr];
    for (int e = R.a_use_off[i], q = 0; e < R.a_use_off[i + 1]; e++, q++) { R.n_use_fr[o + q] = R.a_use_fr[e]; R.n_use_qty[o + q] = R.a_use_qty[e]; }
  }
}

KQ_DEV void ro_fold_pack(const DRows& R, int i) {
  int cq, e0, e1; const int32_t* fr; const int64_t* qty;
  if (i < R.n_rm) { const int r = R.rm_rows[i]; cq = R.o_adm_cq[r]; e0 = R.o_use_off[r]; e1 = R.o_use_off[r + 1]; fr = R.o_use_fr; qty = R.o_use_qty; }
  else { const int a = i - R.n_rm; cq = R.a_cq[a]; e0 = R.a_use_off[a]; e1 = R.a_use_off[a + 1]; fr = R.a_use_fr; qty = R.a_use_qty; }
  int n = e1 - e0;
  if (n > KQ_MAXU) { *R.f_err = 1; n = 0; }
  R.f_cq[i] = cq; R.f_use_n[i] = n;
  for (int q = 0; q < n; q++) { R.f_use_fr[(size_t)i * KQ_MAXU + q] = fr[e0 + q]; R.f_use_qty[(size_t)i * KQ_MAXU + q] = qty[e0 + q]; }
}

KQ_DEV void rows_cell(const DRows& R, int op, int i, bool active) {
  if (op == RO_BUCKET_FILL) { ro_bucket_fill(R, i, active); return; }   // (every lane of the wave takes part in its reduction)
  if (!active) return;
  switch (op) {
    case RO_ROW_INIT: ro_row_init(R, i); break;
    case RO_KEY_RTS: ro_key_rts(R, i); break;
    case RO_KEY_PRIO: ro_key_prio(R, i); break;
    case RO_KEY_TREE: ro_key_tree(R, i); break;
    case RO_RANK: ro_rank(R, i); break;
    case RO_KEY_ASC: ro_key_asc(R, i); break;
    case RO_ASC: ro_asc(R, i); break;
    case RO_ENT_FILL: ro_ent_fill(R, i); break;
    case RO_BOUNDS: ro_bounds(R, i); break;
    case RO_BUCKET_SIZE: ro_bucket_size(R, i); break;
    case RO_LKEY: ro_lkey(R, i); break;
    case RO_LFILL: ro_lfill(R, i); break;
    case RO_MOVE_ROW: ro_move_row(R, i); break;
    case RO_MOVE_ENT: ro_move_ent(R, i); break;
    case RO_ADD_ROW: ro_add_row(R, i); break;
    case RO_KEY_FS: ro_key_fs(R, i); break;
    case RO_FS_FILL: ro_fs_fill(R, i); break;
    case RO_EVICT: ro_evict(R, i); break;
    case RO_REMAP: ro_remap(R, i); break;
    case RO_FOLD_PACK: ro_fold_pack(R, i); break;
    default: break;
  }
}

}  // namespace kq

// kq_prep.hpp — host-side preparation of the HBM-resident snapshot image.
//
// Runs once per kq_snapshot_put (pure C++, no device code): validates the flat snapshot and builds
// the static index structures the kernels need so that nothing on the device ever chases a
// pointer or recurses:
//   depth / root / tree partition      hierarchy walk        (pkg/cache/hierarchy/cohort.go)
//   node_height                        getNodeHeight          (classical/hierarchical_preemption.go:209-215)
//   tree_rows (rank order)             CandidatesOrdering's static part: priority asc, quota
//                                      reservation time desc, UID asc (common/ordering.go:66-81)
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/kq_engine.h"

namespace kq {

constexpr int KQ_MAXD = 8;      // max nodes on a CQ->root path (CQ + 7 cohort levels)
constexpr int KQ_MAXREQ = 16;   // max resources requested by one podset (incl. injected "pods")
constexpr int KQ_MAXU = 56;     // max (flavor,resource) entries in one assignment's usage (18 podsets x (2 resources + pods))
constexpr int KQ_MAXPS = 18;    // max podsets per workload = the API limit (apis/kueue/v1beta2/workload_types.go:36 MaxItems=18)
constexpr uint32_t KQ_POL_DEV_RG_OVERLAP = 1u << 31;   // device copy of cq_policy only: two resource groups of the ClusterQueue cover one resource
constexpr int KQ_MAXR = 8;      // max resources for the incremental (sum-based) DRS; more fall back to the exact loops

// ---- static structures of the scan-formulated classical victim search (kq_cs.hpp) ----------------------------------
// One record per admitted row: its usage entries folded per flavor-resource (<= CS_RFR distinct ones), the policy operands and
// the algorithmic cost of one snapshot.RemoveWorkload / AddWorkload of the row.
constexpr int CS_NS = 4;       // flavor-resource slots of one search on the fast path
constexpr int CS_RFR = 4;      // distinct flavor-resources of a row in its record
constexpr int CS_RFX = 4;      // ... and as many again in the extension record of a WIDE row (AdmRec::flags bit 1): a workload whose podsets
                               // landed on two flavors holds up to 8 — without it ONE such admission sent every victim search of its whole
                               // root tree down the candidate-by-candidate walk (round 6: the closed loop admits one in its first cycle)
constexpr int CS_LEVELS = 3;   // candidate ClusterQueues may sit up to this many levels below the LCA with the preemptor's path
struct alignas(16) AdmRec {
  int64_t qty[CS_RFR];
  int32_t fr[CS_RFR];          // -1 = unused
  int64_t prio, qts;
  int32_t cq;
  int32_t rowbytes;            // 16 * (depth + 1) * usage entries of the row (what the oracle charges per Remove/AddWorkload)
  uint32_t flags;              // bit0 evicted, bit1 wide: the row's flavor-resources 5 .. 8 stand in AdmRecX[row]
  int32_t fs_pos;              // the row's position in its tree's fair-sharing position order (FsScan / FsApply below); 0 without fair sharing
};
struct alignas(16) AdmRecX { int64_t qty[CS_RFX]; int32_t fr[CS_RFX]; };   // read only for wide rows
// One entry of a (tree, flavor-resource) bucket in "level order" d (d = 1 .. CS_LEVELS, the depth below the root): grouped by the
// node of depth d on the way from the row's ClusterQueue to the root (`node`, tree-local id; the ClusterQueue itself when it sits
// at depth d; -1 when it is shallower), evicted rows first, then candidate rank — i.e. the order in which the reference's
// candidate loop meets the rows of one subtree. jd = index of the row inside the bucket's rank order | depth of its ClusterQueue << 24.
struct alignas(8) CsEnt { int32_t jd, node, row, gnode; int64_t qty; };  // gnode = global node id of `node` (-1 if none); qty = the
                                                                          // row's quantity of the bucket's own flavor-resource
// One entry of a bucket in candidate rank order: everything the classification of a candidate reads (hierarchical_preemption.go:81-113)
struct alignas(8) CsRec { int64_t prio, qts; int32_t row, cql, rowbytes; uint32_t flags; };  // cql = index of the ClusterQueue in its tree

// ---- static structures of the LDS-resident fair-sharing victim search (kq_fs.hpp) ---------------------------------
// The candidates of a tree in "position order": grouped by ClusterQueue (tree-local index ascending), inside a ClusterQueue in the
// order the TargetClusterQueueOrdering pops them (evicted first, then candidate rank, fairsharing/ordering.go:84-90 +
// common/ordering.go:42-83). A search's candidate classes are bitmaps over these positions. Two records per position: what the
// candidate scan reads, and what a pop / a snapshot.RemoveWorkload of the row reads (tree-local path of its ClusterQueue included).
constexpr int FS_LV = 4;       // path levels (ClusterQueue + 3 cohort levels) of a tree the LDS search handles
constexpr int FS_RFR = CS_RFR + CS_RFX;   // usage entries of a row in the search: the records below carry the first CS_RFR, a WIDE row's other
                                          // entries are read from AdmRecX[row] (FsApply::wide; bit 15 of FsScan::cbytes, FS_SCAN_WIDE)
constexpr uint16_t FS_SCAN_WIDE = 0x8000u;
struct alignas(16) FsScan { int64_t prio, qts; int16_t fr[CS_RFR]; int32_t row; int16_t cql; uint16_t cbytes; };
struct alignas(16) FsApply { int64_t qty[CS_RFR]; int16_t lp[FS_LV]; int16_t fr[CS_RFR]; int32_t row; uint32_t hkey; uint16_t cbytes; uint8_t plen; uint8_t res[CS_RFR]; uint8_t wide; };
struct alignas(16) FsQ { int64_t lq, sqb; };  // localQuota, SubtreeQuota (INT64_MAX where the node has no entry) of one (node, flavor-resource)
static_assert(sizeof(FsScan) == 32 && sizeof(FsApply) == 64 && sizeof(FsQ) == 16, "FsScan / FsApply / FsQ are read as 32 / 64 / 16 byte records");

// fingerprint of a bucket's row set (Prep::frb_sig): equal sets give equal values, different sets almost surely different ones
#if defined(__HIP__) || defined(__HIPCC__)
#define KQ_PREP_HD __host__ __device__ inline
#else
#define KQ_PREP_HD static inline
#endif
// the fold of a row's usage entries into its record(s): -> number of distinct flavor-resources, or -1 when they do not fit
template <class FR, class QTY> KQ_PREP_HD int adm_rec_fold(AdmRec& a, AdmRecX& x, int k0, int k1, FR fr_of, QTY qty_of) {
  for (int e = 0; e < CS_RFR; e++) { a.fr[e] = -1; a.qty[e] = 0; }
  for (int e = 0; e < CS_RFX; e++) { x.fr[e] = -1; x.qty[e] = 0; }
  int nf = 0;
  for (int e = k0; e < k1; e++) {
    const int fr = fr_of(e);
    int k = 0;
    while (k < nf && (k < CS_RFR ? a.fr[k] : x.fr[k - CS_RFR]) != fr) k++;
    if (k == nf) {
      if (nf == CS_RFR + CS_RFX) return -1;
      if (nf < CS_RFR) a.fr[nf] = fr; else x.fr[nf - CS_RFR] = fr;
      nf++;
    }
    // plain sum (entries of a repeated flavor-resource are removed one after another by the reference; on plain amounts that
    // is the removal of their sum). Non-plain rows switch the fast search off through fs_plain_adm.
    int64_t& cell = k < CS_RFR ? a.qty[k] : x.qty[k - CS_RFR];
    cell = (int64_t)((uint64_t)cell + (uint64_t)qty_of(e));
  }
  if (nf > CS_RFR) a.flags |= 2u;
  return nf;
}
// the row's quantity of one flavor-resource from its record(s)
KQ_PREP_HD int64_t adm_rec_qty(const AdmRec& a, const AdmRecX* xs, int row, int fr) {
  int64_t q = 0;
  for (int e = 0; e < CS_RFR; e++) if (a.fr[e] == fr) q = a.qty[e];
  if (a.flags & 2u) { const AdmRecX& x = xs[row]; for (int e = 0; e < CS_RFX; e++) if (x.fr[e] == fr) q = x.qty[e]; }
  return q;
}
KQ_PREP_HD uint64_t frb_sig_row(int row) { uint64_t h = (uint64_t)(uint32_t)(row + 1) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; return h * 0xBF58476D1CE4E5B9ull; }
KQ_PREP_HD uint64_t frb_sig_size(int M) { return (uint64_t)(uint32_t)M * 0xD6E8FEB86659FD93ull; }

struct Prep {
  int nq = 0, nc = 0, N = 0, nF = 0, nR = 0, nfr = 0, n_adm = 0, n_rg = 0;
  std::vector<int32_t> depth, root, tree_of, node_local, node_height;
  std::vector<int32_t> path;         // [nq * KQ_MAXD] cq, parent, ..., root ; -1 padded
  std::vector<int32_t> plen;         // [nq]
  int n_tree = 0;
  std::vector<int32_t> tree_node_off, tree_nodes;  // nodes of a tree (CQs first, then cohorts)
  std::vector<int32_t> tree_cq_off, tree_cqs;      // CQs of a tree
  std::vector<int32_t> cq_local;                   // [nq] index of the CQ inside tree_cqs of its tree
  std::vector<int32_t> tree_row_off, tree_rows;    // admitted rows of a tree, in static candidate rank order
  std::vector<int32_t> adm_cq;                     // [n_adm]
  // fair sharing: calculateLendable(parent(node)) per resource (fair_sharing.go:186-200). It only reads quotas
  // (potentialAvailable ignores usage), so it is a per-snapshot constant. [N * nR]; 0 for root nodes.
  std::vector<int64_t> lendable;
  // victim search: per (tree, flavor-resource) the rank positions of the admitted rows that use it (WorkloadUsesResources,
  // classical/candidate_generator.go:54), ascending = candidate order; and per ClusterQueue the algorithmic bytes of its rows
  std::vector<int32_t> frb_off, frb;
  std::vector<int32_t> cq_row_bytes;
  std::vector<int32_t> tree_rows_asc;              // admitted rows of a tree in ascending row order (same offsets as tree_rows)
  std::vector<int32_t> frcount;                    // [N] flavor-resources with a SubtreeQuota entry (DRS iterates those)
  std::vector<int32_t> h_parent;                   // host copies kept for build_fair after a device derive
  std::vector<int64_t> h_ll, h_bl;
  std::vector<AdmRec> adm_rec;                     // [n_adm]
  std::vector<AdmRecX> adm_recx;                   // [n_adm] (meaningful for wide rows)
  std::vector<CsEnt> frl[CS_LEVELS];               // level orders of every bucket (same offsets as frb)
  std::vector<CsRec> frec;                         // rank order of every bucket (same offsets as frb)
  std::vector<uint64_t> frb_sig;                   // [n_tree * nfr] hash of the bucket's row list (equal sets <=> equal buckets)
  std::vector<uint8_t> cs_ok;                      // [n_tree] the tree's shape fits the fast search
  std::vector<int32_t> frbr;                       // admitted row of every bucket entry (same layout as frb)
  std::vector<int32_t> tree_depth;                 // [n_tree] deepest ClusterQueue (0 = no cohort)
  std::vector<int32_t> drank;                      // [N] cohorts: rank among the cohorts of the same depth in the tree (kq_spec.hpp sort keys); ClusterQueues: 0
  std::vector<int32_t> tree_dcnt;                  // [n_tree * KQ_MAXD] cohorts of the tree at every depth
  bool any_preemption = false;                     // some ClusterQueue may preempt (within its queue or by reclaim)
  int cs_max_bucket = 0;
  bool fs_plain_adm = true;
  bool usage_consistent = true;                    // cohort usage == sum over children of max(0, usage - localQuota) (resource_node.go:217-230)
  bool fs_plain = true;                            // every finite amount is small enough that DRS sums cannot saturate
  std::vector<int32_t> rank_pos;                   // [n_adm] position of the row inside its tree's tree_rows segment
  std::vector<int32_t> top_of;                     // [N] the ancestor-or-self that is a child of the root (-1 for roots)
  int max_tree_nodes = 0, max_tree_cqs = 0, max_tree_rows = 0, max_tree_cohorts = 0;
  std::vector<int8_t> cq_res_rg;                   // [nq * nR] RGByResource: index of the covering group inside the ClusterQueue's groups, -1 = none
  std::vector<uint32_t> cq_policy_dev;             // [nq] kq_snapshot.cq_policy + the engine's own bits (KQ_POL_DEV_*)
  // kq_fs.hpp: per tree position (same offsets as tree_rows)
  std::vector<FsScan> fs_scan;
  std::vector<FsApply> fs_apply;
  std::vector<int32_t> fs_posoff;                  // [nq + n_tree] per tree (offset tree_cq_off[t] + t): nqs + 1 position offsets of its ClusterQueues
  std::vector<uint8_t> rec_ok;                     // [n_tree] every admitted row of the tree has at most CS_RFR distinct flavor-resources: AdmRec describes it completely
  std::vector<uint8_t> fs_ok;                      // [n_tree] the tree fits the LDS search (depth, rows with few flavor-resources, index ranges)
  // kq_fs.hpp: tables in tree-node order (index tree_node_off[t] + tree-local node id), so that the search never needs global ids
  std::vector<int16_t> fs_par;                     // tree-local parent of every node, -1 for the root
  std::vector<int16_t> fs_kid, fs_koff, fs_knc, fs_knh;  // children (tree-local ids; ClusterQueue children first) of every cohort
  std::vector<int16_t> fs_c0, fs_c1;               // algorithmic bytes of one DRS evaluation of the node: always / more when it borrows
  std::vector<FsQ> fs_q;                           // [N * nfr]
  std::vector<int64_t> fs_lend;                    // [N * nR] lendable
  std::vector<double> fs_weight;                   // [N]
  std::vector<double> h_weight;                    // host copy of fair_weight for build_fair after a device derive
  int max_tree_mw = 1;                             // words of a candidate bitmap of the largest tree
  bool want_fs = true;                             // build the kq_fs.hpp structures (the engine clears it when fair sharing is off)
  bool skip_rows = false;                          // leave every structure derived from the admitted rows empty: the engine builds them on the device (kq_rows.hpp)
  int max_rsn_per_podset = 1;  // most reason records one podset's flavor scans can produce: max over ClusterQueues of sum over groups of flavors x (2 x resources + 1)
  std::string err;
};

// fair sharing constants from the derived planes (sq = SubtreeQuota). calculateLendable(parent(node)) per
// resource (fair_sharing.go:186-200) only reads quotas (potentialAvailable ignores usage): a per-snapshot constant.
static inline void build_fair(Prep& p, const int64_t* sq, const int64_t* usage, const uint8_t* flags) {
  const int N = p.N;
  const int64_t U = INT64_MAX;
  auto a_add = [&](int64_t a, int64_t b) -> int64_t {
    if (a == U || b == U) return U;
    if (b > 0 && a > U - b) return U;
    if (b < 0 && a < INT64_MIN - b) return INT64_MIN;
    return a + b;
  };
  auto a_sub = [&](int64_t a, int64_t b) -> int64_t {
    if (a == U && b == U) return 0;
    if (a == U) return U;
    if (b == U) return INT64_MIN;
    if (b < 0 && a > U + b) return U;
    if (b > 0 && a < INT64_MIN + b) return INT64_MIN;
    return a - b;
  };
  const size_t nfr = p.nfr;
  const int32_t* parent = p.h_parent.data();
  // potentialAvailable(node, fr) for every node, parents before children (resource_node.go:129-140)
  std::vector<int32_t> by_depth(N);
  for (int n = 0; n < N; n++) by_depth[n] = n;
  std::sort(by_depth.begin(), by_depth.end(), [&](int a, int b) { return p.depth[a] < p.depth[b]; });
  std::vector<int64_t> pot((size_t)N * nfr, 0);
  for (int n : by_depth)
    for (size_t fr = 0; fr < nfr; fr++) {
      size_t o = (size_t)n * nfr + fr;
      if (parent[n] < 0) { pot[o] = sq[o]; continue; }
      int64_t lq = 0;
      if (p.h_ll[o] != KQ_NIL_LIMIT) lq = std::max<int64_t>(0, a_sub(sq[o], p.h_ll[o]));
      int64_t avail = a_add(lq, pot[(size_t)parent[n] * nfr + fr]);
      if (p.h_bl[o] != KQ_NIL_LIMIT) avail = std::min(a_add(sq[o], p.h_bl[o]), avail);
      pot[o] = avail;
    }
  p.frcount.assign(N, 0);
  p.fs_plain = p.fs_plain_adm;
  const int64_t LIM = (int64_t)1 << 50;
  for (int n = 0; n < N; n++)
    for (size_t fr = 0; fr < nfr; fr++) {
      size_t o = (size_t)n * nfr + fr;
      if (flags[o] & KQ_QF_SUBTREE) p.frcount[n]++;
      auto small = [&](int64_t v) { return v > -LIM && v < LIM; };
      if (!small(usage[o])) p.fs_plain = false;
      if (sq[o] != U && !small(sq[o])) p.fs_plain = false;
    }
  p.lendable.assign((size_t)N * p.nR, 0);
  p.top_of.assign(N, -1);
  for (int n = 0; n < N; n++) {
    int par = parent[n];
    if (par < 0) continue;
    int top = n;
    while (parent[parent[top]] >= 0) top = parent[top];
    p.top_of[n] = top;
    const int root = p.root[n];
    for (size_t fr = 0; fr < nfr; fr++) {
      if (!(flags[(size_t)root * nfr + fr] & KQ_QF_SUBTREE)) continue;  // keys of root.SubtreeQuota
      size_t r = fr % p.nR;
      p.lendable[(size_t)n * p.nR + r] = a_add(p.lendable[(size_t)n * p.nR + r], pot[(size_t)par * nfr + fr]);
    }
  }
  // kq_fs.hpp: the same constants in tree-node order
  if (!p.want_fs) { p.fs_q.clear(); p.fs_lend.clear(); p.fs_weight.clear(); p.fs_c0.clear(); p.fs_c1.clear(); return; }
  p.fs_q.assign((size_t)N * nfr, FsQ{0, U}); p.fs_lend.assign((size_t)N * p.nR, 0);
  p.fs_weight.assign(N, 1.0); p.fs_c0.assign(N, 0); p.fs_c1.assign(N, 0);
  for (int tp = 0; tp < N; tp++) {
    const int n = p.tree_nodes[tp];
    for (size_t fr = 0; fr < nfr; fr++) {
      const size_t o = (size_t)n * nfr + fr, d = (size_t)tp * nfr + fr;
      if (p.h_ll[o] != KQ_NIL_LIMIT) p.fs_q[d].lq = std::max<int64_t>(0, a_sub(sq[o], p.h_ll[o]));
      if (flags[o] & KQ_QF_SUBTREE) p.fs_q[d].sqb = sq[o];
    }
    for (int r = 0; r < p.nR; r++) p.fs_lend[(size_t)tp * p.nR + r] = p.lendable[(size_t)n * p.nR + r];
    if (!p.h_weight.empty()) p.fs_weight[tp] = p.h_weight[n];
    const int64_t c0 = (int64_t)p.frcount[n] * 24, c1 = (int64_t)p.frcount[n] * 40 * (p.depth[n] + 1);
    if (c0 > 32767 || c1 > 32767) { for (auto& f : p.fs_ok) f = 0; } else { p.fs_c0[tp] = (int16_t)c0; p.fs_c1[tp] = (int16_t)c1; }
  }
}

// The two properties of the usage plane the engine's shortcuts rely on, re-checked when only usage changes (kq_snapshot_patch):
// cohort usage == sum over children of max(0, usage - localQuota) (resource_node.go:217-230), and every finite amount small enough
// that the per-node borrowed sums of the DRS cannot saturate.
inline void check_usage(const kq_snapshot* s, Prep& p) {
  const int64_t U = INT64_MAX;
  auto a_add = [&](int64_t x, int64_t y) -> int64_t {
    if (x == U || y == U) return U;
    if (y > 0 && x > U - y) return U;
    if (y < 0 && x < INT64_MIN - y) return INT64_MIN;
    return x + y;
  };
  auto a_sub = [&](int64_t x, int64_t y) -> int64_t {
    if (x == U && y == U) return 0;
    if (x == U) return U;
    if (y == U) return INT64_MIN;
    if (y < 0 && x > U + y) return U;
    if (y > 0 && x < INT64_MIN + y) return INT64_MIN;
    return x - y;
  };
  const int N = p.N, nq = p.nq;
  const size_t nfr = p.nfr;
  p.usage_consistent = true;
  for (int c = nq; c < N && p.usage_consistent; c++) {
    const int kx = c - nq;
    for (size_t fr = 0; fr < nfr; fr++) {
      int64_t sum = 0;
      for (int pass = 0; pass < 2; pass++) {
        const int32_t* off = pass == 0 ? s->child_cohort_off : s->child_cq_off;
        const int32_t* lst = pass == 0 ? s->child_cohort : s->child_cq;
        for (int i = off[kx]; i < off[kx + 1]; i++) {
          const size_t o = (size_t)lst[i] * nfr + fr;
          const int64_t ll = s->lend_limit[o];
          const int64_t lq = ll != KQ_NIL_LIMIT ? std::max<int64_t>(0, a_sub(s->subtree_quota[o], ll)) : 0;
          sum = a_add(sum, std::max<int64_t>(0, a_sub(s->usage[o], lq)));
        }
      }
      if (sum != s->usage[(size_t)c * nfr + fr]) { p.usage_consistent = false; break; }
    }
  }
  const int64_t LIM = (int64_t)1 << 50;
  for (size_t i = 0; i < (size_t)N * nfr && p.fs_plain; i++) if (s->usage[i] < 0 || (s->usage[i] >= LIM && s->usage[i] != U)) p.fs_plain = false;
}

inline int build_prep(const kq_snapshot* s, Prep& p) {
  p.nq = s->n_cq; p.nc = s->n_cohort; p.N = p.nq + p.nc; p.nF = s->n_flavor; p.nR = s->n_resource;
  p.nfr = p.nF * p.nR; p.n_adm = s->n_adm;
  const int N = p.N, nq = p.nq;
  if (p.nq < 0 || p.nc < 0 || p.nF <= 0 || p.nR <= 0) { p.err = "bad dimensions"; return KQ_EINVAL; }
  // sizes are checked before anything is sized by them: a negative or absurd count must come back as an error code, not as a
  // std::length_error / bad_alloc out of a vector (the C ABI also catches those, kq_engine.hip KQ_TRY) or a walk off the caller's arrays
  if (p.n_adm < 0) { p.err = "negative n_adm"; return KQ_EINVAL; }
  if (p.n_adm > KQ_MAX_ADMITTED || (int64_t)p.nq + p.nc > KQ_MAX_NODES || (int64_t)p.nF * p.nR > KQ_MAX_FR || ((int64_t)p.nq + p.nc) * ((int64_t)p.nF * p.nR) > ((int64_t)1 << 31) - 1) {
    p.err = "snapshot larger than the engine's index space (admitted rows / nodes / flavor-resources)"; return KQ_EUNSUPPORTED;
  }
  if (!s->cq_adm_off || s->cq_adm_off[0] != 0 || s->cq_adm_off[nq] != p.n_adm) { p.err = "cq_adm_off does not cover [0, n_adm)"; return KQ_EINVAL; }
  if (p.n_adm > 0 && (!s->adm_use_off || s->adm_use_off[0] != 0)) { p.err = "adm_use_off must start at 0"; return KQ_EINVAL; }
  p.n_rg = s->cq_rg_off[nq];
  p.depth.assign(N, 0); p.root.assign(N, 0);
  for (int n = 0; n < N; n++) {
    int d = 0, a = n;
    while (s->parent[a] >= 0) {
      a = s->parent[a];
      if (a < nq || a >= N) { p.err = "parent must be a cohort node"; return KQ_EINVAL; }
      if (++d > N) { p.err = "cycle in cohort tree"; return KQ_EINVAL; }
    }
    p.depth[n] = d; p.root[n] = a;
  }
  p.path.assign((size_t)nq * KQ_MAXD, -1); p.plen.assign(nq, 0);
  for (int c = 0; c < nq; c++) {
    if (p.depth[c] + 1 > KQ_MAXD) { p.err = "cohort tree deeper than KQ_MAXD"; return KQ_EUNSUPPORTED; }
    int k = 0;
    for (int a = c; a >= 0; a = s->parent[a]) p.path[(size_t)c * KQ_MAXD + k++] = a;
    p.plen[c] = k;
  }
  // trees = connected components by root; order trees by ascending root index
  std::vector<int32_t> roots;
  for (int n = 0; n < N; n++) if (p.root[n] == n) roots.push_back(n);
  p.n_tree = (int)roots.size();
  std::vector<int32_t> tree_of_root(N, -1);
  for (int t = 0; t < p.n_tree; t++) tree_of_root[roots[t]] = t;
  p.tree_of.assign(N, 0);
  for (int n = 0; n < N; n++) p.tree_of[n] = tree_of_root[p.root[n]];
  p.tree_node_off.assign(p.n_tree + 1, 0); p.tree_cq_off.assign(p.n_tree + 1, 0); p.tree_row_off.assign(p.n_tree + 1, 0);
  for (int n = 0; n < N; n++) p.tree_node_off[p.tree_of[n] + 1]++;
  for (int c = 0; c < nq; c++) p.tree_cq_off[p.tree_of[c] + 1]++;
  for (int t = 0; t < p.n_tree; t++) { p.tree_node_off[t + 1] += p.tree_node_off[t]; p.tree_cq_off[t + 1] += p.tree_cq_off[t]; }
  p.max_tree_nodes = p.max_tree_cqs = p.max_tree_rows = p.max_tree_cohorts = 0;
  p.tree_nodes.assign(N, 0); p.tree_cqs.assign(nq, 0); p.node_local.assign(N, 0); p.cq_local.assign(nq, 0);
  {
    std::vector<int32_t> fill(p.tree_node_off.begin(), p.tree_node_off.end() - 1);
    for (int n = 0; n < N; n++) { int t = p.tree_of[n]; p.node_local[n] = fill[t] - p.tree_node_off[t]; p.tree_nodes[fill[t]++] = n; }
    std::vector<int32_t> fq(p.tree_cq_off.begin(), p.tree_cq_off.end() - 1);
    for (int c = 0; c < nq; c++) { int t = p.tree_of[c]; p.cq_local[c] = fq[t] - p.tree_cq_off[t]; p.tree_cqs[fq[t]++] = c; }
  }
  // per tree and depth: the cohorts numbered 0.. (dense sort keys of the speculative process kernel, kq_spec.hpp)
  p.drank.assign(N, 0); p.tree_dcnt.assign((size_t)std::max(p.n_tree, 1) * KQ_MAXD, 0);
  for (int n = nq; n < N; n++) if (p.depth[n] < KQ_MAXD) p.drank[n] = p.tree_dcnt[(size_t)p.tree_of[n] * KQ_MAXD + p.depth[n]]++;
  // node heights, children before parents: process cohorts by decreasing depth
  p.node_height.assign(N, 0);
  {
    std::vector<int32_t> order;
    for (int n = nq; n < N; n++) order.push_back(n);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return p.depth[a] > p.depth[b]; });
    for (int n : order) {
      int k = n - nq;
      int ncc = s->child_cohort_off[k + 1] - s->child_cohort_off[k], ncq = s->child_cq_off[k + 1] - s->child_cq_off[k];
      int h = std::min(ncc + ncq, 1);
      for (int i = s->child_cohort_off[k]; i < s->child_cohort_off[k + 1]; i++) h = std::max(h, p.node_height[s->child_cohort[i]] + 1);
      p.node_height[n] = h;
    }
  }
  // admitted rows
  p.adm_cq.assign(p.n_adm, -1);
  for (int c = 0; c < nq; c++) {
    if (s->cq_adm_off[c] > s->cq_adm_off[c + 1] || s->cq_adm_off[c] < 0 || s->cq_adm_off[c + 1] > p.n_adm) { p.err = "cq_adm_off not monotone"; return KQ_EINVAL; }
    for (int r = s->cq_adm_off[c]; r < s->cq_adm_off[c + 1]; r++) p.adm_cq[r] = c;
  }
  for (int r = 0; r < p.n_adm; r++) {
    if (p.adm_cq[r] < 0) { p.err = "admitted row outside cq_adm_off"; return KQ_EINVAL; }
    if (s->adm_use_off[r] > s->adm_use_off[r + 1]) { p.err = "adm_use_off not monotone"; return KQ_EINVAL; }
    p.tree_row_off[p.tree_of[p.adm_cq[r]] + 1]++;
  }
  for (int t = 0; t < p.n_tree; t++) p.tree_row_off[t + 1] += p.tree_row_off[t];
  p.tree_rows.assign(p.skip_rows ? 0 : p.n_adm, 0);
  if (!p.skip_rows) {
    std::vector<int32_t> fr(p.tree_row_off.begin(), p.tree_row_off.end() - 1);
    for (int r = 0; r < p.n_adm; r++) p.tree_rows[fr[p.tree_of[p.adm_cq[r]]]++] = r;
    for (int t = 0; t < p.n_tree; t++)
      std::sort(p.tree_rows.begin() + p.tree_row_off[t], p.tree_rows.begin() + p.tree_row_off[t + 1], [&](int a, int b) {
        if (s->adm_priority[a] != s->adm_priority[b]) return s->adm_priority[a] < s->adm_priority[b];
        if (s->adm_reserve_ts[a] != s->adm_reserve_ts[b]) return s->adm_reserve_ts[a] > s->adm_reserve_ts[b];
        if (s->adm_uid_rank[a] != s->adm_uid_rank[b]) return s->adm_uid_rank[a] < s->adm_uid_rank[b];
        return a < b;
      });
  }
  p.tree_rows_asc = p.tree_rows;
  for (int t = 0; t < p.n_tree && !p.skip_rows; t++) std::sort(p.tree_rows_asc.begin() + p.tree_row_off[t], p.tree_rows_asc.begin() + p.tree_row_off[t + 1]);
  p.frb_off.assign((size_t)p.n_tree * p.nfr + 1, 0);
  p.cq_row_bytes.assign(nq, 0);
  {
    auto first_use = [&](int row, int e) { for (int q = s->adm_use_off[row]; q < e; q++) if (s->adm_use_fr[q] == s->adm_use_fr[e]) return false; return true; };
    for (int r = 0; r < p.n_adm && !p.skip_rows; r++) {
      p.cq_row_bytes[p.adm_cq[r]] += 32 + 12 * (s->adm_use_off[r + 1] - s->adm_use_off[r]);
      const int t = p.tree_of[p.adm_cq[r]];
      for (int e = s->adm_use_off[r]; e < s->adm_use_off[r + 1]; e++) {
        const int fr = s->adm_use_fr[e];
        if (fr < 0 || fr >= p.nfr) { p.err = "adm_use_fr out of range"; return KQ_EINVAL; }
        if (first_use(r, e)) p.frb_off[(size_t)t * p.nfr + fr + 1]++;
      }
    }
    for (size_t i = 0; i + 1 < p.frb_off.size(); i++) p.frb_off[i + 1] += p.frb_off[i];
    p.frb.assign(p.frb_off.back(), 0);
    std::vector<int32_t> fill(p.frb_off.begin(), p.frb_off.end() - 1);
    for (int t = 0; t < p.n_tree && !p.skip_rows; t++)
      for (int i = p.tree_row_off[t]; i < p.tree_row_off[t + 1]; i++) {  // rank order => every bucket is ascending
        const int r = p.tree_rows[i];
        for (int e = s->adm_use_off[r]; e < s->adm_use_off[r + 1]; e++)
          if (first_use(r, e)) p.frb[fill[(size_t)t * p.nfr + s->adm_use_fr[e]]++] = i - p.tree_row_off[t];
      }
  }
  p.rank_pos.assign(p.skip_rows ? 0 : p.n_adm, 0);
  for (int t = 0; t < p.n_tree && !p.skip_rows; t++)
    for (int i = p.tree_row_off[t]; i < p.tree_row_off[t + 1]; i++) p.rank_pos[p.tree_rows[i]] = i - p.tree_row_off[t];
  // ---- scan-formulated classical search: row records, level orders of the buckets, bucket signatures ----
  {
    p.cs_ok.assign(p.n_tree, 1);
    p.tree_depth.assign(p.n_tree, 0);
    for (int c = 0; c < nq; c++) p.tree_depth[p.tree_of[c]] = std::max(p.tree_depth[p.tree_of[c]], (int32_t)p.depth[c]);
    p.any_preemption = false;
    for (int c = 0; c < nq; c++) if (KQ_POL_WITHIN_CQ(s->cq_policy[c]) != KQ_POLICY_NEVER || KQ_POL_RECLAIM(s->cq_policy[c]) != KQ_POLICY_NEVER) p.any_preemption = true;
    p.frbr.assign(p.frb.size(), 0);
    for (int t = 0; t < p.n_tree; t++)
      for (int i = p.frb_off[(size_t)t * p.nfr]; i < p.frb_off[(size_t)(t + 1) * p.nfr]; i++) p.frbr[i] = p.tree_rows[p.tree_row_off[t] + p.frb[i]];
    p.adm_rec.assign(p.skip_rows ? 0 : p.n_adm, AdmRec{});
    p.adm_recx.assign(p.skip_rows ? 0 : p.n_adm, AdmRecX{});
    p.fs_ok.assign(p.n_tree, 1); p.rec_ok.assign(p.n_tree, 1);
    for (int c = 0; c < nq; c++) if (p.depth[c] > CS_LEVELS) p.cs_ok[p.tree_of[c]] = 0;
    for (int r = 0; r < p.n_adm && !p.skip_rows; r++) {
      AdmRec& a = p.adm_rec[r];
      const int c = p.adm_cq[r];
      a.prio = s->adm_priority[r]; a.qts = s->adm_queue_ts[r]; a.cq = c; a.flags = (s->adm_flags[r] & KQ_ADM_EVICTED) ? 1u : 0u;
      a.rowbytes = 16 * (p.depth[c] + 1) * (s->adm_use_off[r + 1] - s->adm_use_off[r]);
      const int nf = adm_rec_fold(a, p.adm_recx[r], s->adm_use_off[r], s->adm_use_off[r + 1],
                                  [&](int e) { return s->adm_use_fr[e]; }, [&](int e) { return s->adm_use_qty[e]; });
      // more flavor-resources than the two records hold: the scan-formulated searches (classical and fair) are off for the tree
      if (nf < 0) { p.cs_ok[p.tree_of[c]] = 0; p.rec_ok[p.tree_of[c]] = 0; p.fs_ok[p.tree_of[c]] = 0; }
    }
    // ---- kq_fs.hpp: candidates in position order, children lists in tree-node order ----
    for (int c = 0; c < nq; c++) if (p.depth[c] + 1 > FS_LV) p.fs_ok[p.tree_of[c]] = 0;
    if (p.nfr > 32767) for (auto& f : p.fs_ok) f = 0;
    if (!p.want_fs) for (auto& f : p.fs_ok) f = 0;
    p.fs_scan.assign(p.want_fs ? p.n_adm : 0, FsScan{}); p.fs_apply.assign(p.want_fs ? p.n_adm : 0, FsApply{});
    p.fs_posoff.assign((size_t)nq + p.n_tree, 0);
    p.max_tree_mw = 1;
    if (p.skip_rows) { p.fs_scan.clear(); p.fs_apply.clear(); }
    for (int t = 0; t < p.n_tree && p.want_fs && !p.skip_rows; t++) {
      const int q0 = p.tree_cq_off[t], nqs = p.tree_cq_off[t + 1] - q0, r0 = p.tree_row_off[t];
      const int nn = p.tree_node_off[t + 1] - p.tree_node_off[t];
      if (nn > 32767) p.fs_ok[t] = 0;
      p.max_tree_mw = std::max(p.max_tree_mw, (p.tree_row_off[t + 1] - r0 + 63) / 64 + 1);
      int pos = 0;
      std::vector<int32_t> rows;
      for (int i = 0; i < nqs; i++) {
        const int c = p.tree_cqs[q0 + i];
        p.fs_posoff[(size_t)q0 + t + i] = pos;
        rows.assign(0, 0);
        for (int r = s->cq_adm_off[c]; r < s->cq_adm_off[c + 1]; r++) rows.push_back(r);
        auto hkey = [&](int r) { return ((s->adm_flags[r] & KQ_ADM_EVICTED) ? 0u : 0x80000000u) | (uint32_t)p.rank_pos[r]; };
        std::sort(rows.begin(), rows.end(), [&](int a, int b) { return hkey(a) < hkey(b); });
        for (int r : rows) {
          const AdmRec& a = p.adm_rec[r];
          FsScan& sc = p.fs_scan[(size_t)r0 + pos];
          FsApply& ap = p.fs_apply[(size_t)r0 + pos];
          sc.prio = a.prio; sc.qts = a.qts; sc.row = r; sc.cql = (int16_t)i;
          sc.cbytes = (uint16_t)(32 + 12 * (s->adm_use_off[r + 1] - s->adm_use_off[r]));
          ap.cbytes = sc.cbytes; ap.wide = (a.flags & 2u) ? 1 : 0;
          if (ap.wide) sc.cbytes |= FS_SCAN_WIDE;
          for (int e = 0; e < CS_RFR; e++) { sc.fr[e] = ap.fr[e] = (int16_t)a.fr[e]; ap.qty[e] = a.qty[e]; ap.res[e] = (uint8_t)(a.fr[e] >= 0 ? a.fr[e] % p.nR : 255); }
          for (int l = 0; l < FS_LV; l++) ap.lp[l] = l < p.plen[c] ? (int16_t)p.node_local[p.path[(size_t)c * KQ_MAXD + l]] : (int16_t)-1;
          ap.hkey = hkey(r); ap.row = r; ap.plen = (uint8_t)std::min(p.plen[c], 255);
          p.adm_rec[r].fs_pos = pos;
          pos++;
        }
      }
      p.fs_posoff[(size_t)q0 + t + nqs] = pos;
    }
    p.fs_kid.assign(N, -1); p.fs_koff.assign(N, 0); p.fs_knc.assign(N, 0); p.fs_knh.assign(N, 0); p.fs_par.assign(N, -1);
    for (int tp = 0; tp < N; tp++) { const int n = p.tree_nodes[tp]; if (s->parent[n] >= 0) p.fs_par[tp] = (int16_t)p.node_local[s->parent[n]]; }
    for (int t = 0; t < p.n_tree; t++) {
      const int n0 = p.tree_node_off[t], nn = p.tree_node_off[t + 1] - n0;
      int fill = 0;
      for (int li = 0; li < nn; li++) {
        const int n = p.tree_nodes[n0 + li];
        if (n < nq) continue;
        const int kx = n - nq;
        p.fs_koff[n0 + li] = (int16_t)fill;
        p.fs_knc[n0 + li] = (int16_t)(s->child_cq_off[kx + 1] - s->child_cq_off[kx]);
        p.fs_knh[n0 + li] = (int16_t)(s->child_cohort_off[kx + 1] - s->child_cohort_off[kx]);
        for (int i = s->child_cq_off[kx]; i < s->child_cq_off[kx + 1] && fill < nn; i++) p.fs_kid[n0 + fill++] = (int16_t)p.node_local[s->child_cq[i]];
        for (int i = s->child_cohort_off[kx]; i < s->child_cohort_off[kx + 1] && fill < nn; i++) p.fs_kid[n0 + fill++] = (int16_t)p.node_local[s->child_cohort[i]];
      }
    }
    p.frb_sig.assign((size_t)p.n_tree * p.nfr, 0);
    p.cs_max_bucket = 0;
    for (int l = 0; l < CS_LEVELS; l++) p.frl[l].assign(p.frb.size(), CsEnt{0, -1, 0, -1, 0});
    p.frec.assign(p.frb.size(), CsRec{});
    // (M <= 0xfff0 is checked by the search; the depth shares the word with j)
    std::vector<int32_t> anc;  // scratch: ancestor at height l of the row's ClusterQueue
    std::vector<int32_t> idx, keyv, cnt;
    std::vector<int32_t> cq_anc((size_t)CS_LEVELS * std::max(nq, 1), -1);  // ancestor of a ClusterQueue at depth l + 1 (level order l), -1 = none
    for (int l = 0; l < CS_LEVELS; l++)
      for (int c = 0; c < nq; c++) {
        int n = c;
        const int dd = l + 1;
        if (p.depth[n] < dd) n = -1;
        else for (int h = p.depth[n]; h > dd; h--) n = s->parent[n];
        cq_anc[(size_t)l * nq + c] = n;
      }
    for (int t = 0; t < p.n_tree && !p.skip_rows; t++)
      for (int fr = 0; fr < p.nfr; fr++) {
        const size_t b = (size_t)t * p.nfr + fr;
        const int o = p.frb_off[b], M = p.frb_off[b + 1] - o;
        p.cs_max_bucket = std::max(p.cs_max_bucket, M);
        // order-independent (a sum of mixed row ids + the size): the device-side rebuild (kq_rows.hpp) adds the terms with atomics
        uint64_t sig = frb_sig_size(M);
        for (int j = 0; j < M; j++) sig += frb_sig_row(p.tree_rows[p.tree_row_off[t] + p.frb[o + j]]);
        p.frb_sig[b] = sig;
        for (int j = 0; j < M; j++) {
          const int row = p.tree_rows[p.tree_row_off[t] + p.frb[o + j]];
          const AdmRec& a = p.adm_rec[row];
          p.frec[o + j] = CsRec{a.prio, a.qts, row, p.cq_local[a.cq], a.rowbytes, a.flags};
        }
        for (int l = 0; l < CS_LEVELS; l++) {
          anc.resize(M);
          for (int j = 0; j < M; j++) anc[j] = cq_anc[(size_t)l * nq + p.adm_cq[p.tree_rows[p.tree_row_off[t] + p.frb[o + j]]]];
          // order: ancestor node ascending (none last), evicted rows first, then bucket position — a stable counting sort on
          // (ancestor, not evicted): the keys are node ids, the bucket is in rank order already (this was a comparison sort per
          // bucket and level and 60 % of kq_snapshot_put's host time at cfg 3)
          idx.resize(M); keyv.resize(M);
          if (cnt.size() < 2 * ((size_t)N + 1) + 1) cnt.assign(2 * ((size_t)N + 1) + 1, 0);
          for (int j = 0; j < M; j++) {
            const int row = p.tree_rows[p.tree_row_off[t] + p.frb[o + j]];
            const int ea = (s->adm_flags[row] & KQ_ADM_EVICTED) ? 0 : 1;
            keyv[j] = (anc[j] < 0 ? N : anc[j]) * 2 + ea;
            cnt[keyv[j] + 1]++;
          }
          for (size_t q = 1; q < cnt.size(); q++) cnt[q] += cnt[q - 1];
          for (int j = 0; j < M; j++) idx[cnt[keyv[j]]++] = j;
          std::fill(cnt.begin(), cnt.end(), 0);
          for (int q = 0; q < M; q++) {
            const int j = idx[q];
            const int row = p.tree_rows[p.tree_row_off[t] + p.frb[o + j]];
            const int64_t qty = adm_rec_qty(p.adm_rec[row], p.adm_recx.data(), row, fr);
            p.frl[l][o + q] = CsEnt{j | (p.depth[p.adm_cq[row]] << 24), anc[j] >= 0 ? p.node_local[anc[j]] : -1, row, anc[j], qty};
          }
        }
      }
  }
  for (int t = 0; t < p.n_tree; t++) {
    p.max_tree_nodes = std::max(p.max_tree_nodes, p.tree_node_off[t + 1] - p.tree_node_off[t]);
    p.max_tree_cqs = std::max(p.max_tree_cqs, p.tree_cq_off[t + 1] - p.tree_cq_off[t]);
    p.max_tree_rows = std::max(p.max_tree_rows, p.tree_row_off[t + 1] - p.tree_row_off[t]);
    p.max_tree_cohorts = std::max(p.max_tree_cohorts, (p.tree_node_off[t + 1] - p.tree_node_off[t]) - (p.tree_cq_off[t + 1] - p.tree_cq_off[t]));
  }
  p.cq_res_rg.assign((size_t)p.nq * p.nR, -1);
  p.cq_policy_dev.assign(std::max(p.nq, 1), 0);
  for (int c = 0; c < p.nq; c++) {
    if (s->cq_rg_off[c + 1] - s->cq_rg_off[c] > 127) { p.err = "more than 127 resource groups in a ClusterQueue"; return KQ_EUNSUPPORTED; }
    bool overlap = false;   // a resource covered by two groups of the ClusterQueue (the webhook rejects such a spec, the cache keeps it)
    for (int g = s->cq_rg_off[c + 1] - 1; g >= s->cq_rg_off[c]; g--)  // the first group covering a resource wins (util/resourcegroups/resourcegroups.go:62)
      for (int i = s->rg_res_off[g]; i < s->rg_res_off[g + 1]; i++)
        if (s->rg_res[i] >= 0 && s->rg_res[i] < p.nR) {
          int8_t& cell = p.cq_res_rg[(size_t)c * p.nR + s->rg_res[i]];
          if (cell >= 0 && cell != (int8_t)(g - s->cq_rg_off[c])) overlap = true;
          cell = (int8_t)(g - s->cq_rg_off[c]);
        }
    p.cq_policy_dev[c] = (s->cq_policy[c] & 0xfffu) | (overlap ? KQ_POL_DEV_RG_OVERLAP : 0u);
  }
  p.max_rsn_per_podset = 1;
  for (int c = 0; c < p.nq; c++) {
    int tot = 1;
    for (int g = s->cq_rg_off[c]; g < s->cq_rg_off[c + 1]; g++)
      tot += (s->rg_flavor_off[g + 1] - s->rg_flavor_off[g]) * (2 * (s->rg_res_off[g + 1] - s->rg_res_off[g]) + 1);  // (a cell of a head that replaces a workload slice can leave two: flavor mismatch + quota)
    p.max_rsn_per_podset = std::max(p.max_rsn_per_podset, tot);
  }
  // ---- fair sharing constants (depend on SubtreeQuota / usage: recomputed after kq_snapshot_derive) ----
  p.h_parent.assign(s->parent, s->parent + N);
  p.h_ll.assign(s->lend_limit, s->lend_limit + (size_t)N * p.nfr);
  p.h_bl.assign(s->borrow_limit, s->borrow_limit + (size_t)N * p.nfr);
  p.fs_plain_adm = true;
  for (int r = 0; r < p.n_adm; r++)
    for (int e = s->adm_use_off[r]; e < s->adm_use_off[r + 1]; e++)
      if (s->adm_use_qty[e] < 0 || s->adm_use_qty[e] >= ((int64_t)1 << 50)) p.fs_plain_adm = false;
  if (s->fair_weight) p.h_weight.assign(s->fair_weight, s->fair_weight + N); else p.h_weight.clear();
  build_fair(p, s->subtree_quota, s->usage, s->quota_flags);
  // is the uploaded cohort usage what accumulateFromChild would have produced? (lets kq_cycle_commit re-derive it
  // level by level instead of bubbling every admission one after another)
  {
    const int64_t U = INT64_MAX;
    auto a_add = [&](int64_t x, int64_t y) -> int64_t {
      if (x == U || y == U) return U;
      if (y > 0 && x > U - y) return U;
      if (y < 0 && x < INT64_MIN - y) return INT64_MIN;
      return x + y;
    };
    auto a_sub = [&](int64_t x, int64_t y) -> int64_t {
      if (x == U && y == U) return 0;
      if (x == U) return U;
      if (y == U) return INT64_MIN;
      if (y < 0 && x > U + y) return U;
      if (y > 0 && x < INT64_MIN + y) return INT64_MIN;
      return x - y;
    };
    p.usage_consistent = true;
    const size_t nfr = p.nfr;
    for (int c = nq; c < N && p.usage_consistent; c++) {
      const int kx = c - nq;
      for (size_t fr = 0; fr < nfr; fr++) {
        int64_t sum = 0;
        for (int pass = 0; pass < 2; pass++) {
          const int32_t* off = pass == 0 ? s->child_cohort_off : s->child_cq_off;
          const int32_t* lst = pass == 0 ? s->child_cohort : s->child_cq;
          for (int i = off[kx]; i < off[kx + 1]; i++) {
            const size_t o = (size_t)lst[i] * nfr + fr;
            const int64_t ll = s->lend_limit[o];
            const int64_t lq = ll != KQ_NIL_LIMIT ? std::max<int64_t>(0, a_sub(s->subtree_quota[o], ll)) : 0;
            sum = a_add(sum, std::max<int64_t>(0, a_sub(s->usage[o], lq)));
          }
        }
        if (sum != s->usage[(size_t)c * nfr + fr]) { p.usage_consistent = false; break; }
      }
    }
  }
  // index validation
  for (int g = 0; g < p.n_rg; g++) {
    for (int k = s->rg_flavor_off[g]; k < s->rg_flavor_off[g + 1]; k++) if (s->rg_flavor[k] < 0 || s->rg_flavor[k] >= p.nF) { p.err = "rg_flavor out of range"; return KQ_EINVAL; }
    for (int k = s->rg_res_off[g]; k < s->rg_res_off[g + 1]; k++) if (s->rg_res[k] < 0 || s->rg_res[k] >= p.nR) { p.err = "rg_res out of range"; return KQ_EINVAL; }
  }
  for (int k = 0; k < s->adm_use_off[p.n_adm]; k++) if (s->adm_use_fr[k] < 0 || s->adm_use_fr[k] >= p.nfr) { p.err = "adm_use_fr out of range"; return KQ_EINVAL; }
  return KQ_OK;
}

}  // namespace kq

// kq_device.hpp — wave-cooperative device logic of the admission engine (gfx950, wave64).
//
// Execution model: ONE WAVEFRONT PER PENDING HEAD (nominate) / PER ROOT-COHORT TREE (process).
// Inside a wave, control flow is uniform; lanes fan out over
//   * (flavor x resource) cells of a resource group   -> fitsResourceQuota for every cell at once,
//   * admitted workloads in candidate-rank order      -> classification + ballot, in-order drain,
//   * flavor-resource "slots" of one workload's usage -> addUsage / removeUsage / Available per slot.
// Everything a Go pointer/map walk does in the reference is a bounded loop over flat arrays here:
// CQ->root paths are precomputed (kq_prep.hpp), quota planes are dense [node][flavor*nR+resource].
//
// Reference semantics followed (paths under /root/reference/pkg):
//   cache/scheduler/resource_node.go         localQuota :67  LocalAvailable :92  available :106
//                                            potentialAvailable :129  addUsage :144  removeUsage :156
//   scheduler/flavorassigner/flavorassigner.go   assignFlavors :708  findFlavorForPodSets :1065
//                                            fitsResourceQuota :1334  isPreferred :536  shouldTryNextFlavor :1263
//   scheduler/preemption/preemption.go       classicalPreemptions :284  fillBackWorkloads :341  workloadFits :669
//   scheduler/preemption/classical/*.go      candidate classes / order / validity
//   scheduler/preemption/preemption_oracle.go SimulatePreemption :43
//   scheduler/scheduler.go                   getInitialAssignments :880  processEntry :392  fits :771
//
// The same source is compiled by g++ with -DKQ_HOST_EMU into a 1-lane emulation that ONLY the CPU
// test-suite loads (tests/emu): it checks the uniform control logic against the oracle without a
// GPU. It is not a fallback: the product library contains no host implementation of these paths.
#pragma once
#include <cstddef>
#include <cstdint>
#ifdef KQ_HOST_EMU
#include <algorithm>
#include <cstdio>
#include <vector>
#endif

#include "../../include/kq_engine.h"
#include "kq_prep.hpp"

#ifdef KQ_HOST_EMU
#define KQ_DEV static inline
#define KQ_MDEV inline
#define KQ_NOINLINE static
#define KQ_HD static inline
namespace kq {
constexpr int WAVE = 1;
KQ_DEV int lane_id() { return 0; }
KQ_DEV uint64_t wballot(bool p) { return p ? 1ull : 0ull; }
template <class T> KQ_DEV T wbcast(T v, int) { return v; }
KQ_DEV void wsync() {}
KQ_DEV void wsync_lds() {}
KQ_DEV void bsync() {}
KQ_DEV uint64_t wmin_u64(uint64_t v) { return v; }
KQ_DEV int ffs64(uint64_t m) { return __builtin_ctzll(m); }
KQ_DEV int popc64(uint64_t m) { return __builtin_popcountll(m); }
KQ_DEV int atomic_add_i32(int* p, int v) { int o = *p; *p += v; return o; }
KQ_DEV void atomic_max_i32(int* p, int v) { if (v > *p) *p = v; }
KQ_DEV void atomic_min_i32(int* p, int v) { if (v < *p) *p = v; }
KQ_DEV void atomic_add_i64(long long* p, long long v) { *p = (long long)((unsigned long long)*p + (unsigned long long)v); }  // wraps like the device's atomic (bucket fingerprints rely on it)
KQ_DEV void atomic_or_u64(uint64_t* p, uint64_t v) { *p |= v; }
KQ_DEV void atomic_and_u64(uint64_t* p, uint64_t v) { *p &= v; }
KQ_DEV int64_t atomic_cas_i64(int64_t* p, int64_t expect, int64_t v) { int64_t o = *p; if (o == expect) *p = v; return o; }
KQ_DEV int64_t wsum_i64(int64_t v) { return v; }
KQ_DEV int wbcast_u(int v, int) { return v; }
KQ_DEV int64_t wbcast_u(int64_t v, int) { return v; }
// device-wide (agent scope) synchronisation between workgroups: the emulation is one thread
KQ_DEV uint64_t ag_load_u64(const uint64_t* p) { return *p; }
KQ_DEV void ag_store_u64(uint64_t* p, uint64_t v) { *p = v; }
KQ_DEV uint32_t ag_load_u32(const uint32_t* p) { return *p; }
KQ_DEV bool ag_cas_u64(uint64_t* p, uint64_t expect, uint64_t v) { if (*p != expect) return false; *p = v; return true; }
KQ_DEV void ag_add_u32(uint32_t* p, uint32_t v) { *p += v; }
KQ_DEV void ag_release() {}
KQ_DEV void ag_acquire() {}
KQ_DEV void ag_pause() {}
KQ_DEV int wuniform_i32(int v) { return v; }
KQ_DEV int64_t wprefix_incl_i64(int64_t v) { return v; }
KQ_DEV int wprefix_incl_i32(int v) { return v; }
KQ_DEV int64_t wshfl_i64(int64_t v, int) { return v; }
KQ_DEV int wshift_up_i32(int v) { return v; }
KQ_DEV int wshift_down_i32(int v) { return v; }
KQ_DEV int wshfl_i32(int v, int) { return v; }
KQ_DEV int clz64(uint64_t m) { return __builtin_clzll(m); }
static thread_local int g_emu_pipeline = 0;  // tests: emulate the helper waves of k_process prefetching one chunk ahead
}  // namespace kq
#else
#include <hip/hip_runtime.h>
#define KQ_DEV __device__ __forceinline__
#define KQ_MDEV __device__ __forceinline__
#define KQ_NOINLINE __device__ __noinline__
#define KQ_HD __host__ __device__ __forceinline__
namespace kq {
constexpr int WAVE = 64;
KQ_DEV int lane_id() { return (int)(threadIdx.x & 63); }
KQ_DEV uint64_t wballot(bool p) { return __ballot(p); }
KQ_DEV int wbcast(int v, int src) { return __shfl(v, src, 64); }
KQ_DEV int64_t wbcast(int64_t v, int src) { return (int64_t)__shfl((long long)v, src, 64); }
KQ_DEV double wbcast(double v, int src) { return __longlong_as_double(__shfl(__double_as_longlong(v), src, 64)); }
// Wave-level sync: the serial logic of every kernel runs inside ONE wave, whose lanes execute in lockstep, so
// making one lane's LDS / global-scratch writes visible to the others only needs the outstanding memory
// operations drained (workgroup-scope fence = s_waitcnt; the CU's L1 is shared, no invalidate) and the compiler
// kept from moving accesses across. No s_barrier: workgroups may hold helper waves that are not in this code.
KQ_DEV void wsync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
// all waves of the workgroup (phase boundaries of the multi-wave fair-sharing kernel)
KQ_DEV void bsync() { __syncthreads(); }
// Wave reductions on the DPP network (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every lane of a 16-lane row ends up
// with the row's result; the four rows are combined on the scalar unit). ~5x shorter than a __shfl_xor butterfly, whose every
// step is a ds_bpermute round trip. All 64 lanes must be active (uniform control flow), as for the shuffles they replace.
template <int CTRL> KQ_DEV uint64_t dpp_u64(uint64_t v) {
  const int lo = __builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(v >> 32), (int)(v >> 32), CTRL, 0xf, 0xf, false);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
KQ_DEV uint64_t readlane_u64(uint64_t v, int l) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
KQ_DEV uint64_t wmin_u64(uint64_t v) {
  uint64_t t;
  t = dpp_u64<0xB1>(v); v = t < v ? t : v;
  t = dpp_u64<0x4E>(v); v = t < v ? t : v;
  t = dpp_u64<0x141>(v); v = t < v ? t : v;
  t = dpp_u64<0x140>(v); v = t < v ? t : v;
  const uint64_t r0 = readlane_u64(v, 0), r1 = readlane_u64(v, 16), r2 = readlane_u64(v, 32), r3 = readlane_u64(v, 48);
  const uint64_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
  return a < b ? a : b;
}
// LDS-only visibility inside the single wave of a workgroup: the LDS pipeline is in order per wave, so only
// the compiler must be kept from reordering / caching; no wait for outstanding global memory traffic.
KQ_DEV void wsync_lds() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
KQ_DEV int ffs64(uint64_t m) { return __ffsll((unsigned long long)m) - 1; }
KQ_DEV int popc64(uint64_t m) { return __popcll((unsigned long long)m); }
KQ_DEV int atomic_add_i32(int* p, int v) { return atomicAdd(p, v); }
KQ_DEV void atomic_max_i32(int* p, int v) { atomicMax(p, v); }
KQ_DEV void atomic_min_i32(int* p, int v) { atomicMin(p, v); }
KQ_DEV void atomic_add_i64(long long* p, long long v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
KQ_DEV void atomic_or_u64(uint64_t* p, uint64_t v) { atomicOr((unsigned long long*)p, (unsigned long long)v); }
KQ_DEV void atomic_and_u64(uint64_t* p, uint64_t v) { atomicAnd((unsigned long long*)p, (unsigned long long)v); }
KQ_DEV int64_t atomic_cas_i64(int64_t* p, int64_t expect, int64_t v) {
  return (int64_t)atomicCAS((unsigned long long*)p, (unsigned long long)expect, (unsigned long long)v);
}
KQ_DEV int64_t wsum_i64(int64_t x) {
  uint64_t v = (uint64_t)x;
  v += dpp_u64<0xB1>(v);
  v += dpp_u64<0x4E>(v);
  v += dpp_u64<0x141>(v);
  v += dpp_u64<0x140>(v);
  return (int64_t)(readlane_u64(v, 0) + readlane_u64(v, 16) + readlane_u64(v, 32) + readlane_u64(v, 48));
}
// broadcast from a lane that is the same for the whole wave (no LDS crossbar round trip)
KQ_DEV int wbcast_u(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
KQ_DEV int64_t wbcast_u(int64_t v, int src) {
  const int l = __builtin_amdgcn_readfirstlane(src);
  return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l));
}
// device-wide (agent scope) synchronisation between workgroups (helper workgroups of k_process_fair): relaxed loads / stores that
// bypass the non-coherent caches, fences that publish / pick up everything else
KQ_DEV uint64_t ag_load_u64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
KQ_DEV void ag_store_u64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
KQ_DEV uint32_t ag_load_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
KQ_DEV bool ag_cas_u64(uint64_t* p, uint64_t expect, uint64_t v) {
  return __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
KQ_DEV void ag_add_u32(uint32_t* p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
KQ_DEV void ag_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
KQ_DEV void ag_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
KQ_DEV void ag_pause() { __builtin_amdgcn_s_sleep(8); }
KQ_DEV int wuniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
// inclusive prefix sums over the lanes of the wave on the DPP network: Hillis-Steele inside each 16-lane row (row_shr 1, 2, 4, 8,
// lanes without a source add 0), then the row totals travel with row_bcast:15 (rows 1 and 3 take lane 15 of the row before) and
// row_bcast:31 (rows 2 and 3 take lane 31). 6 steps of two v_mov_dpp + one 64-bit add instead of 6 x 2 ds_bpermute round trips:
// the scan-formulated victim search is made of these.
template <int CTRL, int ROWMASK> KQ_DEV uint64_t dpp0_u64(uint64_t v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROWMASK, 0xf, false);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
KQ_DEV int64_t wprefix_incl_i64(int64_t x) {
  uint64_t v = (uint64_t)x;
  v += dpp0_u64<0x111, 0xf>(v);
  v += dpp0_u64<0x112, 0xf>(v);
  v += dpp0_u64<0x114, 0xf>(v);
  v += dpp0_u64<0x118, 0xf>(v);
  v += dpp0_u64<0x142, 0xa>(v);
  v += dpp0_u64<0x143, 0xc>(v);
  return (int64_t)v;
}
KQ_DEV int wprefix_incl_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}
// the value of the lane before / after (lane 0 / lane 63 keep their own): one DPP move across the whole wave
KQ_DEV int wshift_up_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
KQ_DEV int wshift_down_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
KQ_DEV int64_t wshfl_i64(int64_t v, int src) { return (int64_t)__shfl((long long)v, src, 64); }
KQ_DEV int wshfl_i32(int v, int src) { return __shfl(v, src, 64); }
KQ_DEV int clz64(uint64_t m) { return __clzll((long long)m); }
}  // namespace kq
#endif

namespace kq {

constexpr int CELLS = 64;  // cells evaluated per pass (one lane each on gfx950)

// ------------------------------------------------------------------------------------------------
// device-visible images
// ------------------------------------------------------------------------------------------------
struct DSnap {
  int nq, nc, N, nF, nR, nfr, pods_res, n_adm, n_tree, nfw;
  const int32_t* resource_order;
  const int32_t* parent;
  const int64_t *nominal, *bl, *ll, *sq;
  const uint8_t* qflags;
  const int32_t *cq_rg_off, *rg_flavor_off, *rg_flavor, *rg_res_off, *rg_res;
  const uint32_t* cq_policy;
  const int32_t* cq_thr;
  const int64_t* cq_gen;
  const int32_t* cq_adm_off;
  const int64_t *adm_prio, *adm_qts;
  const uint8_t* adm_flags;
  const int32_t *adm_use_off, *adm_use_fr;
  const int64_t* adm_use_qty;
  // prep
  const int32_t *path, *plen, *tree_of, *node_local, *node_height, *adm_cq, *cq_local;
  const int32_t *tree_node_off, *tree_nodes, *tree_cq_off, *tree_cqs, *tree_row_off, *tree_rows;
  // fair sharing
  const int64_t* lendable;   // [N * nR] calculateLendable(parent(node)) per resource (static)
  const int32_t* top_of;     // [N] ancestor-or-self that is a child of the root
  const int32_t* rank_pos;   // [n_adm] position of the row in its tree's rank-ordered tree_rows segment
  const int32_t* frcount;    // [N] flavor-resources with a SubtreeQuota entry
  const int32_t* tree_rows_asc;  // admitted rows of a tree, ascending (offsets = tree_row_off)
  const int32_t *frb_off, *frb;  // [(tree * nfr + fr)] rank positions of the rows using that flavor-resource, ascending
  const int32_t* cq_row_bytes;   // [nq] algorithmic bytes of a ClusterQueue's candidate records
  const double* fair_weight; // [N]
  const int32_t *child_cohort_off, *child_cohort, *child_cq_off, *child_cq, *depth;
  const int64_t* adm_rts;
  const uint32_t* adm_uid;
  // scan-formulated classical victim search (kq_cs.hpp): static per-snapshot structures built by kq_prep.hpp
  const AdmRec* adm_rec;        // [n_adm]
  const AdmRecX* adm_recx;      // [n_adm] flavor-resources 5 .. 8 of a wide row (AdmRec::flags bit 1)
  const CsEnt* frl[CS_LEVELS];  // level orders of the buckets (offsets = frb_off)
  const int32_t* frbr;          // admitted row of every bucket entry
  const CsRec* frec;            // rank order of every bucket (offsets = frb_off)
  const uint64_t* frb_sig;      // [n_tree * nfr]
  const uint8_t* cs_ok;         // [n_tree]
  const int32_t* tree_depth;    // [n_tree]
  const int32_t* drank;         // [N] cohorts: rank among the cohorts of the same depth of the tree (sort keys of kq_spec.hpp)
  const int32_t* tree_dcnt;     // [n_tree * KQ_MAXD] cohorts of the tree per depth
  // LDS-resident fair-sharing victim search (kq_fs.hpp): candidates in position order and the tree's constants in tree-node order
  const FsScan* fs_scan;        // [n_adm] (offsets = tree_row_off)
  const FsApply* fs_apply;      // [n_adm]
  const int32_t* fs_posoff;     // [nq + n_tree] per tree at tree_cq_off[t] + t
  const uint8_t* fs_ok;         // [n_tree]
  const uint8_t* rec_ok;        // [n_tree] AdmRec describes every row of the tree completely
  const int16_t *fs_kid, *fs_koff, *fs_knc, *fs_knh, *fs_c0, *fs_c1, *fs_par;  // [N] at tree_node_off[t] + local id
  const FsQ* fs_q;              // [N * nfr]
  const int64_t* fs_lend;       // [N * nR]
  const double* fs_weight;      // [N]
  const int8_t* cq_res_rg;      // [nq * nR] resource group of the ClusterQueue covering the resource (index inside the ClusterQueue's
                                // groups, -1 = none): resourcegroups.RGByResource as one load instead of a walk of two CSR levels
};

struct DCfg {
  uint32_t gates;
  int fair_sharing;
  int n_fs, fs[2];           // fair-sharing preemption strategies (preemption.go:364-366)
  int fs_plain;              // all amounts small: per-node borrowed sums are exact in plain int64 (no saturation)
  int quota_check_strategy;
  int cs_on;                 // classical victim searches may take the scan formulation (kq_cs.hpp)
  int fs_on;                 // fair-sharing victim searches may take the LDS-resident formulation (kq_fs.hpp)
  int cs_lazy;               // the scan search works on growing prefixes of the candidate time order (kq_cs.hpp; KQ_CS_LAZY=0 turns it off)
  int fs_lrun;               // the fair iterator's leader pops on its own while entries change no usage (KQ_FS_LRUN=0 turns it off)
  int fs_batch;              // ... and evaluate the candidates of a cohort's ClusterQueues as a batch (bits: kq_host.hpp fs_batch_bits; KQ_FS_BATCH=0 turns it off)
  int dbg_variant;           // KQ_PROF builds only: timing experiments (KQ_DEBUG_VARIANT; results are wrong when non-zero)
  int any_preempt;           // some ClusterQueue of the snapshot may preempt (Prep::any_preemption)
  int64_t cycle;
};

struct DHeads {
  int n;                  // head count; with n_dev set: only an upper bound (arrays and grids are sized by it)
  const int32_t* n_dev;   // the count in device memory (the pending side's gather writes it: kq_pending_step) or null
  const int32_t* cq;
  const int64_t *priority, *queue_ts;
  const uint32_t* flags;
  const int32_t *ps_off, *ps_count, *ps_min_count, *ps_req_off, *req_res;
  const int64_t* req_qty;
  const uint64_t* ps_flavor_ok;
  const int32_t* ps_last_tried;
  const int64_t *last_generation, *last_cycle;
  const uint64_t *last_hash, *hash;
  // workload slices (kq_heads.slice_*): all null when no head of the batch replaces a slice
  const int32_t *slice_row, *ps_slice_count, *req_slice_flavor;
  const int64_t* req_slice_qty;
  const int32_t* ps_slice_pods_flavor;
  const int64_t* ps_slice_pods_qty;
  const int32_t* ps_group;   // [n_ps] PodSetGroupName id, -1 = none; null: no podset groups in the batch (kq_heads.ps_group)
};

KQ_DEV int hn(const DHeads& H) { return H.n_dev ? H.n_dev[0] : H.n; }

// One reason why a flavor was not assigned as Fit: the operands of a Status.reasons string (flavorassigner.go:349), see KQ_RSN_*.
struct alignas(16) RsnRec { uint8_t code, podset; int16_t flavor, resource, pad; int64_t a, b, c; };
static_assert(sizeof(RsnRec) == 32, "RsnRec is copied to the host as 32-byte records");

// decisions (same layout as kq_decisions) + per-head internals carried from nominate to process
struct DOut {
  uint8_t *status, *action, *nominated_mode, *mode, *requeue_reason, *skip;
  int32_t *borrowing, *order;
  int32_t* flavor;
  uint8_t* res_mode;
  int32_t* tried_idx;
  int32_t* ps_count;
  // internals
  int32_t* use_n;     // [H]
  int32_t* use_fr;    // [H*KQ_MAXU]
  int64_t* use_qty;   // [H*KQ_MAXU]
  int32_t* tgt_pos;   // [H] offset into the target pool
  int32_t* tgt_n;     // [H]
  int32_t* pool_row;  // [pool_cap]
  uint8_t* pool_reason;
  int32_t pool_cap;
  int32_t* pool_count;  // [1] atomic
  int32_t* error;       // [1] first device-side error (KQ_E*)
  long long* stat_bytes;  // [1] algorithmic bytes, accumulated by lane 0
  // reasons (optional: rsn_win == 0 -> not recorded): a window of rsn_win records per head, rewritten by every assign_flavors of
  // the head, so it ends up describing the assignment the head keeps; rsn_n[h] = records used, negative = window overflowed
  RsnRec* rsn;            // [H * rsn_win]
  int32_t* rsn_n;         // [H]
  int32_t rsn_win;
};

// per-wave-slot scratch in HBM (L2-resident working sets of one victim search)
struct DScratch {
  int64_t* w;        // [slots][max_tree_nodes * slot_cap] private usage of the simulated tree
  uint8_t* cqinfo;   // [slots][max_tree_cqs]  0 = not collected, 1+level = collected under path[level]
  uint8_t* cls;      // [slots][max_tree_rows] candidate class per rank position
  int32_t* tgt_row;  // [slots][tgt_cap]
  uint8_t* tgt_reason;
  int32_t* nom;      // [slots][KQ_MAXPS * nR] NominationMapping of the entry being recomputed (workload.go:262)
  int32_t* cand;     // [slots][max_tree_rows] rank positions of the rows that use a flavor-resource needing preemption
  uint64_t* mark;    // [slots][ceil(max_tree_rows / 64)] bitmap used to merge the buckets of several flavor-resources
  // fair sharing: per-CQ candidate queues of the TargetClusterQueueOrdering (fairsharing/ordering.go:46)
  int32_t* qcnt;     // [slots][max_tree_cqs] remaining candidates of the CQ
  uint32_t* qhead;   // [slots][max_tree_cqs] key of the first remaining candidate (evicted bit | rank position)
  uint8_t* cohp;     // [slots][max_tree_nodes] prunedCohorts
  int64_t* psum;     // [slots][max_tree_nodes * nR] borrowed sums of the search's private state (C.fs_plain)
  int32_t* ppos;     // [slots][max_tree_nodes]
  // fair-sharing iterator (fair_sharing_iterator.go): per tree slot
  int32_t* cq_ent;   // [slots][max_tree_cqs] cqToEntry: the head still to schedule for the CQ, -1 = none
  uint64_t* fs_keys; // [slots][max_tree_cqs * KQ_MAXD][4] drsValues folded into the tournament key (FsKey) of the
                     // CQ's entry at each ancestor level
  int32_t* fs_win;   // [slots][max_tree_nodes] tournament winner per cohort
  int32_t* fs_seq;   // [slots][max_tree_cqs] the tree's pop sequence
  int32_t* fs_key;   // [H] merge key: smallest CQ index at or after the entry in its tree's sequence
  uint8_t* fs_stale; // [slots][max_tree_cqs] lowest path level whose cached DRS is out of date (255 = none)
  int32_t* fs_cost;  // [slots][max_tree_cqs * KQ_MAXD] algorithmic bytes of the cached DRS evaluation
  long long* fs_sum; // [slots] sum of fs_cost over the entries still in the map
  int32_t* fs_ctl;   // [slots][4] leader -> helpers: remaining, last popped CQ, usage changed
  // per-node borrowed amount per resource = sum over flavors of max(0, usage - subtreeQuota), and the number of
  // borrowed flavor-resources: DRS (fair_sharing.go:149-182) is a function of these. bu_* : cycle-start plane
  // (k.usage), bs_* : the work plane as processEntry mutates it.
  int64_t *bu_sum, *bs_sum;  // [N * nR]
  int32_t *bu_pos, *bs_pos;  // [N]
  int32_t max_tree_nodes, max_tree_cqs, max_tree_rows, slot_cap, tgt_cap;
  unsigned char* cs;  // [slots][cs_bytes] arrays of a scan-formulated search when they do not fit the workgroup's LDS
  int64_t cs_bytes;
};

// Helper workgroups of k_process_fair: the victim searches of one flavor scan of a recomputation (PreemptionOracle.SimulatePreemption
// for every flavor-resource cell that needs one, flavorassigner.go:1375-1384) are independent given the snapshot state. The
// leader of a tree posts them as a batch; idle workgroups (and the leader itself) take them one by one.
struct HelpTask { int32_t fr, base_borrow; int64_t val; };
struct SimTask { int32_t head, fr, base_borrow, pad; int64_t val; };
constexpr int SIM_KS = 4;       // scans of one head whose simulations can be listed ahead (K::sim_*); further ones are searched in place
constexpr int SIM_ROUNDS = 2;   // k_nominate_emit rounds behind the lean pass: scans 2 .. SIM_ROUNDS + 1 of a head
struct HelpRes { int32_t pm, borrow; int64_t bytes; };
struct HelpBox {
  uint64_t hdr;      // batch sequence << 32 | tasks (0 tasks = closed)
  uint64_t next;     // batch sequence << 32 | next task to hand out
  uint32_t done;     // tasks of the batch finished
  uint32_t seq;      // leader only
  int32_t entry;     // the head being recomputed
  int32_t pad;
  HelpTask task[CELLS];
  HelpRes res[CELLS];
};

// Sharded nominate, replicated process (include/kq_engine.h kq_cycle_nominate_shard / kq_cycle_process_merged): every rank nominates
// the heads it owns and writes their nomination into an exchange buffer of int64 words that is ZERO for the heads of the others, so
// that one SUM all-reduce over the buffer merges the shards. Layout (n heads, P podsets, C = P * nR cells):
//   [n][SH_HEAD]  nominated mode, borrowing, use_n, tgt_n << 32 | global target position, use_fr[KQ_MAXU], use_qty[KQ_MAXU]
//   [P]           ps_count
//   [C]           (flavor + 1) | res_mode << 24 | (tried_idx + 1) << 32
//   [2 + world]   algorithmic bytes of the nomination, device error, pool entries used per rank
//   [n] + [n][rsn_win][4]   reason records (only with rsn_win > 0)
//   [world][pool_cap]       target pool segments: admitted row | reason << 32
constexpr int SH_HEAD = 4 + 2 * KQ_MAXU;
struct DShard {
  const uint8_t* mine;   // [n] this engine nominates head h (null: every head)
  int64_t* x;            // the exchange buffer (null outside the sharded calls)
  int world, rank, pool_cap;
};
KQ_HD size_t shard_off_ps(int n) { return (size_t)n * SH_HEAD; }
KQ_HD size_t shard_off_cell(int n, size_t P) { return shard_off_ps(n) + P; }
KQ_HD size_t shard_off_misc(int n, size_t P, int nR) { return shard_off_cell(n, P) + P * nR; }
KQ_HD size_t shard_off_rsn(int n, size_t P, int nR, int world) { return shard_off_misc(n, P, nR) + 2 + world; }
KQ_HD size_t shard_off_pool(int n, size_t P, int nR, int world, int rsn_win) { return shard_off_rsn(n, P, nR, world) + (rsn_win ? (size_t)n * (1 + (size_t)rsn_win * 4) : 0); }
KQ_HD size_t shard_words(int n, size_t P, int nR, int world, int rsn_win, int pool_cap) { return shard_off_pool(n, P, nR, world, rsn_win) + (size_t)world * pool_cap; }

struct K {  // everything a kernel needs
  DSnap S;
  DCfg C;
  DHeads H;
  DOut O;
  DScratch X;
  const int64_t* usage;      // cycle-start usage (nominate reads this)
  int64_t* usage_work;       // process: snapshot usage as mutated by the cycle
  int64_t* usage_np;         // process: usage_work minus every workload preempted so far this cycle
  uint8_t* preempted;        // [n_adm] PreemptedWorkloads membership (preempted_workloads.go:26)
  const int32_t* order_idx;  // [H] entry index at iterator position i
  long long* prof;           // [64] optional segment cycle counters (KQ_PROF builds)
  int32_t* cq_rm_bytes;      // [nq] candidate-record bytes of the rows preempted so far this cycle (they left cq.Workloads)
  uint8_t* cq_dirty;         // [nq] a ClusterQueue-level usage cell of this CQ was written in HBM during this cycle's k_process
  struct PRec* grec;         // [H] per-head entry records, static part (rec_fill_static); k_process copies them into LDS
  const int32_t* usage_big;  // [1] set by kq_cycle_commit / release when a ClusterQueue usage cell left the plain range (>= 2^50 or
                             // negative): the sum-based DRS shortcuts (C.fs_plain) are off from then on
  int32_t* defer_list;       // [H] heads the lean nominate pass handed to the full pass
  int32_t* defer_count;      // [1]
  int32_t* nom_ticket;       // [1] next position of defer_list to hand out (k_nominate pulls heads: their costs differ by orders of magnitude)
  // speculative process kernel (kq_spec.hpp)
  int32_t* cq_heads;         // [nq] heads of the batch per ClusterQueue (k_records counts them; > 1: the entries take the serial kernel)
  int32_t* spec_resume;      // [n_tree] iterator position from which the serial kernel (process_tree) takes the tree over; >= H.n: nothing left
  int32_t* spec_stats;       // [8] diagnostics: windows, passes, entries decided by the rounds, entries handed back, ...
  int64_t* spec_kt;          // [workgroups of k_process_spec][SP_KT_WORDS] per-item constants of the window being solved
  // written by k_records for every (head, slot, path level) cell / (head, slot), against the usage at the start of the cycle:
  int64_t *spec_K, *spec_T;  // [H * FU * FD] cohort levels: K = c - usage - val + E0 (term of Available, E0 folded in), T = localQuota - usage
  int32_t* spec_o;           // [H * FU * FD] index of the cell in a usage plane
  int64_t *spec_push, *spec_nv;  // [H * FU] what leaves the ClusterQueue level (max(0, val - E0)); the ClusterQueue's cell after AddUsage(val)
  struct SpecHdr* spec_hdr;  // [H] BY ITERATOR POSITION: what the rounds need to know of the entry at that position (written with the order)
  DShard shard;              // sharded nominate (all zero: the ordinary cycle)
  // Exactness certificate of a cycle run on a SHARD of a root tree (kueue_amd/sharding.py): for every flavor-resource
  // the smallest slack any admitted entry had at the ROOT level of Available (root term of resource_node.go:106-122 minus the
  // request). The root is the only node shards of one tree share; if the usage all other shards add to it stays within this slack,
  // every decision of this shard is the one the unsharded cycle takes. cert_flags: an entry changed usage on a path the slack
  // does not cover (preemption targets, recomputation, non-plain operands, negative reservation, fair sharing).
  long long* root_margin;    // [n_tree * nfr], start = CERT_INF
  int32_t* cert_flags;       // [n_tree]
  const struct TCyc* tc;     // Topology-Aware Scheduling inside the cycle (kq_tas_cycle.hpp; kq_cycle_run_tas), null in the ordinary cycle
  // Simulations ahead of the full nominate pass (round 6, sim_emit / sim_worker below): the SimulatePreemption calls of one flavor scan
  // of a deferred head are independent given the cycle-start snapshot (preemption_oracle.go:43: a pure function of the snapshot, the
  // head and the cell). The lean pass lists those of the head's first scan, k_nominate_sim runs one per wave; k_nominate_emit then walks
  // the deferred heads again with those results in hand and lists the NEXT scan (a later podset, resource group or chunk of flavors:
  // its cells depend on what the scans before it assigned), and so on for SIM_ROUNDS rounds; the full pass reads the results where it
  // would have searched. A cycle in which a handful of heads need victims no longer lasts as long as one head's ~100 searches in a row.
  struct SimTask* sim_task;  // [sim_cap]
  HelpRes* sim_res;          // [sim_cap]
  int32_t* sim_ctl;          // [2 + SIM_ROUNDS] tasks listed, next task to hand out, next deferred head of emit round r (zeroed per cycle, with defer_count)
  int32_t* sim_nscan;        // [H] scans listed for the head (zeroed per cycle); null: the mechanism is off
  int32_t* sim_key;          // [H * SIM_KS] which scan: podset | resName << 8 | first flavor index of the chunk << 16
  int32_t* sim_cell;         // [H * SIM_KS * CELLS] task of cell c of that scan, -1 = none
  int32_t sim_cap;
  HelpBox* help;             // [n_tree] or null: no helper workgroups in this launch
  uint32_t* help_quit;       // [1] trees whose leader has finished
  int help_trees;            // n_tree of the launch (helper workgroups are the blocks after them)
};
KQ_DEV bool fs_plain_now(const K& k) { return k.C.fs_plain && !(k.usage_big && *k.usage_big); }

// ------------------------------------------------------------------------------------------------
// resources.Amount (pkg/resources/amount.go) on raw int64; INT64_MAX == Unlimited.
// Amount.Cmp == plain signed compare because Unlimited is the maximum value.
// ------------------------------------------------------------------------------------------------
constexpr int64_t I64MAX = INT64_MAX;
constexpr int64_t I64MIN = INT64_MIN;
KQ_DEV int64_t sat_add(int64_t a, int64_t b) {
  if (b > 0 && a > I64MAX - b) return I64MAX;
  if (b < 0 && a < I64MIN - b) return I64MIN;
  return a + b;
}
KQ_DEV int64_t sat_sub(int64_t a, int64_t b) {
  if (b < 0 && a > I64MAX + b) return I64MAX;
  if (b > 0 && a < I64MIN + b) return I64MIN;
  return a - b;
}
KQ_DEV int64_t sat_mul(int64_t a, int64_t b) {
  if (a == 0 || b == 0) return 0;
  if ((a == -1 && b == I64MIN) || (b == -1 && a == I64MIN)) return I64MAX;
  int64_t res = (int64_t)((uint64_t)a * (uint64_t)b);
  if (res / b != a) return ((a < 0) == (b < 0)) ? I64MAX : I64MIN;
  return res;
}
KQ_DEV int64_t a_add(int64_t a, int64_t b) { return (a == I64MAX || b == I64MAX) ? I64MAX : sat_add(a, b); }
KQ_DEV int64_t a_addi(int64_t a, int64_t v) { return a == I64MAX ? a : sat_add(a, v); }
KQ_DEV int64_t a_sub(int64_t a, int64_t b) {
  if (a == I64MAX && b == I64MAX) return 0;
  if (a == I64MAX) return I64MAX;
  if (b == I64MAX) return I64MIN;
  return sat_sub(a, b);
}
KQ_DEV int64_t i64max(int64_t a, int64_t b) { return a > b ? a : b; }
KQ_DEV int64_t i64min(int64_t a, int64_t b) { return a < b ? a : b; }

// ---- sharding certificate helpers (K::root_margin) ----
constexpr long long CERT_INF = 0x7f7f7f7f7f7f7f7fll;  // k_prep fills 32-bit words
constexpr int64_t CERT_PLAIN = (int64_t)1 << 56;       // = PLAIN_LIMIT: slack arithmetic is only exact on plain operands
KQ_DEV void cert_min(long long* cell, long long v) {
#ifdef KQ_HOST_EMU
  if (v < *cell) *cell = v;
#else
  atomicMin(cell, v);
#endif
}
// Slack of the root term with resources.Amount arithmetic (Unlimited absorbing, saturating): operands that are not plain. An
// Unlimited term anywhere below or at the root means the root never binds for this entry: no constraint. Negative operands (an
// over-subtracted usage cell) are outside what the argument covers: *bad.
KQ_DEV void cert_margin_exact(long long* cell, const int64_t* lq, const int64_t* un, int64_t sq_root, int plen, int64_t qty, int32_t* bad) {
  int64_t e = 0;
  for (int i = 0; i < plen; i++) if (un[i] < 0 || lq[i] < 0) { if (bad) *bad = 1; return; }
  if (sq_root < 0 || qty < 0) { if (bad) *bad = 1; return; }
  for (int i = 0; i < plen - 1; i++) e = a_add(e, i64max(0, a_sub(lq[i], un[i])));
  const int64_t term = a_add(e, a_sub(sq_root, un[plen - 1]));
  if (term == I64MAX) return;
  cert_min(cell, (long long)a_sub(term, qty));
}
KQ_DEV void cert_unverifiable(const K& k, int tree) { if (k.cert_flags && lane_id() == 0) k.cert_flags[tree] = 1; }

// modes (flavorassigner.go:453-472, :510-533; common/types.go:23)
enum { M_NOFIT = 0, M_PREEMPT = 1, M_DEFERRED = 2, M_FIT = 3 };
enum { PM_NOFIT = 0, PM_NOCAND = 1, PM_PREEMPT = 2, PM_RECLAIM = 3, PM_FIT = 4, PM_NEEDS = 5, PM_SKIP = 6 };
// candidate variants (classical/hierarchical_preemption.go:32-44)
enum { V_NEVER = 0, V_WITHIN_CQ = 1, V_HIER = 2, V_RECLAIM_NO_BORROW = 3, V_RECLAIM_BORROW = 4 };

KQ_DEV bool gate(const K& k, uint32_t g) { return (k.C.gates & g) != 0; }
KQ_DEV size_t ix(const DSnap& S, int n, int fr) { return (size_t)n * S.nfr + fr; }

// resource_node.go:67-72
KQ_DEV int64_t local_quota(const DSnap& S, int n, int fr) {
  int64_t ll = S.ll[ix(S, n, fr)];
  if (ll != KQ_NIL_LIMIT) return i64max(0, a_sub(S.sq[ix(S, n, fr)], ll));
  return 0;
}

// Usage accessors: G = global plane, W = private per-search copy restricted to slots
struct UG {
  const DSnap* S; const int64_t* u; int fr;
  KQ_MDEV int64_t get(int n) const { return u[(size_t)n * S->nfr + fr]; }
};
struct UW {
  const DSnap* S; int64_t* w; int ns; int slot;
  KQ_MDEV int64_t get(int n) const { return w[(size_t)S->node_local[n] * ns + slot]; }
  KQ_MDEV void set(int n, int64_t v) const { w[(size_t)S->node_local[n] * ns + slot] = v; }
};
struct UGW {  // writable global plane
  const DSnap* S; int64_t* u; int fr;
  KQ_MDEV int64_t get(int n) const { return u[(size_t)n * S->nfr + fr]; }
  KQ_MDEV void set(int n, int64_t v) const { u[(size_t)n * S->nfr + fr] = v; }
};

// process kernel: usage plane with the tree's cohort rows served from LDS (plane 0 = work, 1 = np)
struct UP {
  const DSnap* S; int64_t* g; int64_t* lds; int on, ncq, ncoh, plane, fr; uint8_t* dirty;
  KQ_MDEV int64_t* cell(int n) const {
    if (on && n >= S->nq) return lds + ((size_t)plane * ncoh + (S->node_local[n] - ncq)) * S->nfr + fr;
    return g + (size_t)n * S->nfr + fr;
  }
  KQ_MDEV int64_t get(int n) const { return *cell(n); }
  KQ_MDEV void set(int n, int64_t v) const { *cell(n) = v; if (dirty && n < S->nq) dirty[n] = 1; }  // see K::cq_dirty
};

// resource_node.go:106-122, root-first iteration over the precomputed path
template <class U> KQ_DEV int64_t available_of(const DSnap& S, const int32_t* path, int plen, int fr, const U& u) {
  int n = path[plen - 1];
  int64_t a = a_sub(S.sq[ix(S, n, fr)], u.get(n));
  for (int i = plen - 2; i >= 0; i--) {
    n = path[i];
    int64_t lq = local_quota(S, n, fr), uu = u.get(n);
    int64_t blv = S.bl[ix(S, n, fr)];
    if (blv != KQ_NIL_LIMIT) {
      int64_t stored = a_sub(S.sq[ix(S, n, fr)], lq);
      int64_t used = i64max(0, a_sub(uu, lq));
      a = i64min(a_add(a_sub(stored, used), blv), a);
    }
    a = a_add(i64max(0, a_sub(lq, uu)), a);
  }
  return a;
}
// resource_node.go:129-140
KQ_DEV int64_t potential_of(const DSnap& S, const int32_t* path, int plen, int fr) {
  int n = path[plen - 1];
  int64_t a = S.sq[ix(S, n, fr)];
  for (int i = plen - 2; i >= 0; i--) {
    n = path[i];
    a = a_add(local_quota(S, n, fr), a);
    int64_t blv = S.bl[ix(S, n, fr)];
    if (blv != KQ_NIL_LIMIT) a = i64min(a_add(S.sq[ix(S, n, fr)], blv), a);
  }
  return a;
}
// clusterqueue_snapshot.go:155-161 / cohort_snapshot.go:90 ; i == 0 is the CQ itself
template <class U> KQ_DEV bool borrowing_with(const DSnap& S, int n, bool is_cq, int fr, int64_t val, const U& u) {
  int64_t q = is_cq ? S.nominal[ix(S, n, fr)] : S.sq[ix(S, n, fr)];
  return q < a_add(u.get(n), val);
}
// classical/hierarchical_preemption.go:221-234 ; returns height, *may_reclaim
template <class U> KQ_DEV int find_height(const DSnap& S, const int32_t* path, int plen, int fr, int64_t val, const U& u, bool* may_reclaim) {
  int c = path[0];
  bool has_parent = plen > 1;
  if (!borrowing_with(S, c, true, fr, val, u) || !has_parent) { *may_reclaim = has_parent; return 0; }
  int64_t remaining = a_sub(val, i64max(0, a_sub(local_quota(S, c, fr), u.get(c))));
  for (int i = 1; i < plen; i++) {
    int t = path[i];
    if (!borrowing_with(S, t, false, fr, remaining, u)) { *may_reclaim = i < plen - 1; return S.node_height[t]; }
    remaining = a_sub(remaining, i64max(0, a_sub(local_quota(S, t, fr), u.get(t))));
  }
  *may_reclaim = false;
  return S.node_height[path[plen - 1]];
}
// The three root-path walks of one fitsResourceQuota cell (available_of, potential_of, find_height) read the same
// (node, flavor-resource) cells; for short paths they are gathered once — every load issued before the first use — and the
// three quantities are computed from registers with the same operations in the same order as the functions above.
constexpr int GP_MAX = 4;
struct GPath { int64_t sq[GP_MAX], lq[GP_MAX], bl[GP_MAX], us[GP_MAX]; int32_t hgt[GP_MAX]; int64_t nominal0; };
KQ_DEV void gpath_load(const DSnap& S, const int32_t* path, int plen, int fr, const int64_t* usage, GPath& g) {
  int64_t ll[GP_MAX];
  #pragma unroll
  for (int i = 0; i < GP_MAX; i++) {
    const int n = path[i < plen ? i : 0];  // clamped: the loads stay unconditional
    const size_t o = ix(S, n, fr);
    g.sq[i] = S.sq[o]; ll[i] = S.ll[o]; g.bl[i] = S.bl[o]; g.us[i] = usage[o]; g.hgt[i] = S.node_height[n];
  }
  g.nominal0 = S.nominal[ix(S, path[0], fr)];
  #pragma unroll
  for (int i = 0; i < GP_MAX; i++) g.lq[i] = ll[i] != KQ_NIL_LIMIT ? i64max(0, a_sub(g.sq[i], ll[i])) : 0;  // local_quota
}
KQ_DEV int64_t gpath_available(const GPath& g, int plen) {  // = available_of
  int64_t a = 0;
  #pragma unroll
  for (int i = GP_MAX - 1; i >= 0; i--) {
    if (i >= plen) continue;
    if (i == plen - 1) { a = a_sub(g.sq[i], g.us[i]); continue; }
    if (g.bl[i] != KQ_NIL_LIMIT) {
      const int64_t stored = a_sub(g.sq[i], g.lq[i]);
      const int64_t used = i64max(0, a_sub(g.us[i], g.lq[i]));
      a = i64min(a_add(a_sub(stored, used), g.bl[i]), a);
    }
    a = a_add(i64max(0, a_sub(g.lq[i], g.us[i])), a);
  }
  return a;
}
KQ_DEV int64_t gpath_potential(const GPath& g, int plen) {  // = potential_of
  int64_t a = 0;
  #pragma unroll
  for (int i = GP_MAX - 1; i >= 0; i--) {
    if (i >= plen) continue;
    if (i == plen - 1) { a = g.sq[i]; continue; }
    a = a_add(g.lq[i], a);
    if (g.bl[i] != KQ_NIL_LIMIT) a = i64min(a_add(g.sq[i], g.bl[i]), a);
  }
  return a;
}
KQ_DEV int gpath_height(const GPath& g, int plen, int64_t val, bool* may_reclaim) {  // = find_height
  const bool has_parent = plen > 1;
  if (!(g.nominal0 < a_add(g.us[0], val)) || !has_parent) { *may_reclaim = has_parent; return 0; }
  int64_t remaining = a_sub(val, i64max(0, a_sub(g.lq[0], g.us[0])));
  int height = 0;
  bool found = false, mr = false;
  #pragma unroll
  for (int i = 1; i < GP_MAX; i++) {
    if (i >= plen || found) continue;
    if (!(g.sq[i] < a_add(g.us[i], remaining))) { found = true; mr = i < plen - 1; height = g.hgt[i]; continue; }
    remaining = a_sub(remaining, i64max(0, a_sub(g.lq[i], g.us[i])));
  }
  if (!found) {
    mr = false;
    // g.hgt[plen - 1] without a run-time index: a dynamically indexed member sends the WHOLE struct to scratch (7 x 16-byte stores per
    // lane and cell pass: 93 % of k_nominate's HBM write traffic before this, profiles/r02g_cfg3-batch_rocprof_summary.txt)
    #pragma unroll
    for (int i = 0; i < GP_MAX; i++) if (i == plen - 1) height = g.hgt[i];
  }
  *may_reclaim = mr;
  return height;
}
// resource_node.go:144-152 (iterative)
template <class U> KQ_DEV void add_usage(const DSnap& S, const int32_t* path, int plen, int fr, int64_t val, const U& u) {
  for (int i = 0; i < plen; i++) {
    int n = path[i];
    int64_t uu = u.get(n);
    int64_t la = i64max(0, a_sub(local_quota(S, n, fr), uu));
    u.set(n, a_add(uu, val));
    if (i + 1 < plen && val > la) val = a_sub(val, la); else break;
  }
}
// resource_node.go:156-165 (iterative)
template <class U> KQ_DEV void remove_usage(const DSnap& S, const int32_t* path, int plen, int fr, int64_t val, const U& u) {
  for (int i = 0; i < plen; i++) {
    int n = path[i];
    int64_t uu = u.get(n);
    int64_t stored = a_sub(uu, local_quota(S, n, fr));
    u.set(n, a_sub(uu, val));
    if (stored <= 0 || i + 1 >= plen) break;
    val = i64min(val, stored);
  }
}

// ------------------------------------------------------------------------------------------------
// per-wave scratch (LDS on the device)
// ------------------------------------------------------------------------------------------------
#ifdef KQ_TAS_CYCLE
// TAS inside the cycle (kq_tas_cycle.hpp): the wave's view of the TopologyAssignments of the assignment under construction
struct TAW {
  int t, nreq;                    // WorkloadsTopologyRequests of the current flavor assignment: the TAS flavor, the podsets in request order
  uint8_t req_ps[KQ_MAXPS];
  int af_early;                   // assign_flavors stopped behind podset af_early - 1 (atLeastOnePodsAssignmentFailed), 0 = ran through
  uint32_t err_mask, has_mask;    // per podset: psError; holds a TopologyAssignment
  int32_t pos[KQ_MAXPS], n[KQ_MAXPS];  // the podset's domains in the slot's store (half 0)
  int kept_used;
  int plane;                      // which leaf-usage plane "the snapshot" is: 0 cycle start, 1 work, 2 work minus preempted rows
  int srch;                       // a GetTargets walk with TAS requests is in flight: its private plane is live
  struct TLeafJob* mail;          // k_process_tas: phase 1 of a placement is shared with the workgroup's helper waves through this LDS block
  unsigned char* lds;             // k_process_tas: room for a class-path placement's working state (kq_tas_device.hpp TLds), lds_bytes of it
  int lds_bytes;
  int pf_pos;                     // k_process_tas: iterator position whose header helper wave 1 should fetch once this entry's placement is done, -1 = none
  // k_process_tas with the LDS block: the request block of a one-podset class-path find and the wave's two domain stores live behind the
  // placement's working state (kq_tas_cycle.hpp TX_*); the TopologyAssignment the entry published is then still in store half 0
  int q_lds, d_lds, pub_lds;
  const struct TPre* cur_pre;     // k_process_tas: the prefetched header of the entry being processed (null: it was loaded the ordinary way)
  int pool_own, pool_next;        // k_process_tas (pool_own = 1): the wave is the only writer of the published pool, pool_next its next free position; else the atomic counter
  // the last tc_find of this wave was a simulate-empty placement of ONE podset that succeeded and nothing has overwritten its request
  // block / store half since: (podset, TAS flavor, count) of it, -1 = none. updateAssignmentForTAS without targets (scheduler.go:966-975)
  // asks for exactly the placement Assign's Preempt branch (flavorassigner.go:889-897) has just computed — an empty cluster looks the
  // same to both — and takes it from there.
  int em_ps, em_t, em_count;
  int req_valid;   // tc_requests' answer (t, nreq, req_ps) is still the current one
};
#define KQ_TAS_WALK(w) ((w).ta.srch != 0)
#define KQ_TAS_PROCESS(k, w) ((k).tc != nullptr && (w).ta.plane != 0)   // (timing builds) assign_flavors inside k_process_tas's recomputation
#else
#define KQ_TAS_WALK(w) false
#define KQ_TAS_PROCESS(k, w) false
#endif
struct Wave {
#ifdef KQ_TAS_CYCLE
  TAW ta;
#endif
  // current head
  int h, cq, plen, ps_base, nps;
  int64_t prio, ts;
  uint32_t hflags, pol;
  int has_last;
  int32_t path[KQ_MAXD];
  // effective requests of the podset being assigned (flavorassigner.go:742-749), Requests.Iter order
  int nreq;
  int32_t req_res[KQ_MAXREQ];
  int64_t req_qty[KQ_MAXREQ];
  uint8_t req_done[KQ_MAXREQ];   // resource already has a flavor from its resource group (:819)
  int8_t req_rg[KQ_MAXREQ];      // resource group (index inside the ClusterQueue's groups) covering the request, -1 = none
  uint8_t req_src[KQ_MAXREQ];    // position of the request in the head's own list (0xff: the injected `pods`): index of its workload-slice columns
  int slice_row;                  // admitted row of the workload slice the head replaces, -1 = none (ElasticJobsViaWorkloadSlices)
  int32_t req_flavor[KQ_MAXREQ], req_borrow[KQ_MAXREQ], req_tried[KQ_MAXREQ];
  uint8_t req_mode[KQ_MAXREQ];
  // Assignment.Usage.Quota.Assigned (:1017-1041) + per-entry flags
  int nuse;
  int32_t use_fr[KQ_MAXU];
  int64_t use_qty[KQ_MAXU];
  uint8_t use_mode[KQ_MAXU];     // FlavorAssignment.Mode of the (podset,resource) that produced it
  int borrowing, rep_mode;
  // filtered requests of the resource group under scan
  int nf;
  int32_t f_res[KQ_MAXREQ];
  int64_t f_qty[KQ_MAXREQ];
  uint8_t f_slot[KQ_MAXREQ];     // index into req_*
  // cell results of one pass of assign_flavors. They share their LDS with the gathered cells of process_entry_fast (g_* below): that
  // function is a leaf that never runs inside an assign_flavors pass, and the workgroup's LDS is full at cfg 3 (cohort rows of both
  // planes + two record buffers + this struct = 160 KB)
  union { int64_t cell_val[CELLS]; int64_t g_sq[CELLS]; };
  union { int64_t cell_aux[CELLS]; int64_t g_lq[CELLS]; };  // cell_aux: operand of the cell's reason (maximum capacity, or val - available)
  union { struct { int32_t cell_borrow[CELLS]; uint8_t cell_pm[CELLS]; }; int64_t g_bl[CELLS]; };
  int nrsn, rsn_ps0, rsn_g0, rsn_over;  // reason records of the assignment under construction (lane 0)
  int defer_head;                 // lean nominate pass: this head needs the full pass (victim search / partial admission)
  // best flavor so far
  int32_t best_mode[KQ_MAXREQ], best_borrow[KQ_MAXREQ];
  int32_t cur_mode[KQ_MAXREQ], cur_borrow[KQ_MAXREQ];
  // victim search
  int ns;                         // slots
  int32_t s_fr[KQ_MAXU];
  int64_t s_qty[KQ_MAXU];         // workloadUsage quantity (0 if slot is need-only)
  uint8_t s_inu[KQ_MAXU], s_need[KQ_MAXU];
  uint8_t adv_at[KQ_MAXD];        // hasHierarchicalAdvantage when collecting under path[level]
  int ntgt;
  int counts[KQ_MAXPS];           // partial admission counts under test
  // process kernel: the tree's cohort rows of usage_work / usage_np live in LDS for the whole cycle
  int pc_on, pc_ncq, pc_ncoh;     // cache enabled, #CQs of the tree (node_local offset of cohorts), #cohorts
  int64_t* pc_lds;                // [2][pc_ncoh][nfr] : plane 0 = usage_work, plane 1 = usage_np
  int32_t path_coh[KQ_MAXD];      // cohort-local index of path[i] (i >= 1)
  int32_t win_e[256], win_cq[256]; // this tree's entries inside the current window of the order, and their ClusterQueues
  uint8_t win_off[256];           // ... and their positions in the order, relative to the window base
  int nwin2[2], chunk_done, chunk_stop;  // leader -> helper waves of the process workgroup
  // gathered cells of the entry under process: c = u * plen + i
  // g_lq / g_sq / g_bl: see the unions above; flv_*: a pass of assign_flavors, per flavor of the pass (the same sharing rule)
  union { int64_t g_uw[CELLS]; int64_t flv_key[CELLS]; };    // flv_key: pref_key of the flavor's representative mode
  union { int64_t g_un[CELLS]; int64_t flv_pack[CELLS]; };   // flv_pack: representative mode | reasons of its cells << 8 | borrow level << 32
  union { uint8_t g_dirty[CELLS]; uint8_t cell_task[CELLS]; };  // cell_task (assign_flavors inside a recomputation): task index of a cell whose simulation was posted to K::help, 0xff = none
  int64_t bytes;                  // algorithmic bytes (lane 0 meaningful)
  int usage_dirty;                // set when processEntry added usage to the snapshot plane (fair-sharing DRS cache)
  // A NEGATIVE amount was added to a flavor-resource column of the snapshot (quotaResourcesToReserve has no
  // max(0, .) on its borrowing branch, scheduler.go:806): in that column cohort usage no longer equals what the
  // children store in it, removals stop being order-independent, and usage_np cannot be maintained incrementally.
  // Once rows have been preempted too, such columns of usage_np are rebuilt from usage_work before every use,
  // removing rows in the reference's canonical order (ascending row).
  int np_broken;                  // any column broken
  int n_pre;                      // rows preempted in this tree so far this cycle
  uint64_t broken[4];             // column bitmap (nfr <= 256; beyond that every column counts as broken)
  int mono_break;                 // usage went DOWN since the leader last looked: "did not fit when fetched" flags are void
  // scan-formulated classical search (kq_cs.hpp): LDS region for its arrays (null: use DScratch::cs) and the constants of the
  // preemptor's path per (slot, level): subtree quota, local quota, borrowing limit, usage at the start / at the stopping time
  unsigned char* cs_lds; int cs_lds_bytes;
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  long long pacc[16];             // (KQ_PROF) segment cycles of the search in flight
#endif
  int help_on, help_tree, help_nt; // recomputation searches of this tree go through K::help
  int pc_region_bytes;            // process kernels: bytes of the workgroup's dynamic LDS a recomputation's searches may borrow
  int64_t cs_sq[CS_NS][CS_LEVELS + 1], cs_lq[CS_NS][CS_LEVELS + 1], cs_bl[CS_NS][CS_LEVELS + 1], cs_u0[CS_NS][CS_LEVELS + 1], cs_uf[CS_NS][CS_LEVELS + 1];
  int64_t cs_nom[CS_NS];
  int32_t cs_pl[CS_LEVELS + 1];   // tree-local ids of the path nodes
};
KQ_DEV void mark_broken(Wave& w, int fr) {
  if (fr < 256) atomic_or_u64(&w.broken[fr >> 6], 1ull << (fr & 63));
  w.np_broken = 1;
  w.mono_break = 1;
}
KQ_DEV bool col_broken(const Wave& w, int fr) { return fr >= 256 ? w.np_broken != 0 : ((w.broken[fr >> 6] >> (fr & 63)) & 1) != 0; }
// must usage_np be rebuilt before it is read?
KQ_DEV bool np_exact_mode(const Wave& w) { return w.np_broken && w.n_pre > 0; }

// optional in-kernel segment timing (build with -DKQ_PROF): cycles accumulated per segment id
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
#define KQ_T0() long long _t0 = clock64()
#define KQ_TS(k, id) do { long long _t1 = clock64(); if (lane_id() == 0) atomic_add_i64((long long*)(k).prof + (id), _t1 - _t0); _t0 = _t1; } while (0)
#define KQ_A0() long long _a0 = clock64()
#define KQ_AS(k, id) do { long long _a1 = clock64(); if (lane_id() == 0) atomic_add_i64((long long*)(k).prof + (id), _a1 - _a0); } while (0)
// accumulate in the wave's LDS (no global atomic for the next fence to wait on); flushed with KQ_LFLUSH
#define KQ_LS(w, id) do { long long _a1 = clock64(); if (lane_id() == 0) (w).pacc[id] += _a1 - _a0; _a0 = _a1; } while (0)
#define KQ_LZERO(w) do { if (lane_id() == 0) for (int _i = 0; _i < 16; _i++) (w).pacc[_i] = 0; } while (0)
#define KQ_LFLUSH(k, w, base) do { if (lane_id() == 0) for (int _i = 0; _i < 16; _i++) { atomic_add_i64((long long*)(k).prof + (base) + _i, (w).pacc[_i]); (w).pacc[_i] = 0; } } while (0)
#else
#define KQ_A0() do {} while (0)
#define KQ_AS(k, id) do {} while (0)
#define KQ_LS(w, id) do {} while (0)
#define KQ_LFLUSH(k, w, base) do {} while (0)
#define KQ_LZERO(w) do {} while (0)
#define KQ_T0() do {} while (0)
#define KQ_TS(k, id) do {} while (0)
#endif

#ifdef KQ_HOST_EMU
static thread_local long long g_cs[32];   // (thread_local: the emulated group of tests/test_group.py runs one engine per thread)
#define CSTAT(i, v) (g_cs[i] += (v))
static thread_local int g_cs_check = 0, g_cs_force_off = 0;  // tests: run every scan-formulated search a second time as a walk and compare
#else
#define CSTAT(i, v) do {} while (0)
#endif
KQ_DEV void set_error(const K& k, int code) {
  if (lane_id() == 0 && *k.O.error == 0) *k.O.error = code;
}

// ------------------------------------------------------------------------------------------------
// flavor fungibility ordering (flavorassigner.go:536-576) as a sortable key:
// isPreferred(a,b) <=> pref_key(a) > pref_key(b)
// ------------------------------------------------------------------------------------------------
#ifdef KQ_TAS_CYCLE
KQ_DEV void tc_reset(Wave& w);
KQ_NOINLINE void tc_requests(const K& k, Wave& w);
KQ_DEV void tc_assign_tas(const K& k, Wave& w, int slot);
KQ_DEV void tc_search_begin(const K& k, Wave& w, int slot);
KQ_DEV void tc_search_end(Wave& w);
KQ_DEV void tc_search_row(const K& k, Wave& w, int slot, int row, bool add);
KQ_DEV bool tc_search_fits(const K& k, Wave& w, int slot);
KQ_DEV void tc_update_assignment(const K& k, Wave& w, int slot, const int32_t* trow, int nt);
KQ_NOINLINE void tc_publish(const K& k, Wave& w, int slot);
KQ_DEV int tc_adm_flavor(const K& k, int psg, int res);
KQ_DEV void tc_sp_reset(const K& k, int psg);
KQ_DEV bool tc_is_tas_flavor(const K& k, int flavor);
#endif
KQ_DEV int64_t pref_key(int pm, int64_t borrow, uint32_t pol) {
  if (pm == PM_NOFIT) return -1;
  int64_t cls = (pm == PM_NOCAND) ? 0 : 1;
  int64_t nb = (int64_t)0x3fffffff - i64min(borrow, 0x3fffffff);  // lower borrow => larger
  if (KQ_POL_PREFERENCE(pol) == KQ_PREF_PREEMPTION_OVER_BORROWING) return (cls << 40) | (nb << 8) | (int64_t)pm;
  return (cls << 40) | ((int64_t)pm << 32) | nb;
}
// flavorassigner.go:1263-1282
KQ_DEV bool should_try_next(int pm, int64_t borrow, uint32_t pol) {
  if (pm == PM_NOFIT || pm == PM_NOCAND) return true;
  if ((pm == PM_PREEMPT || pm == PM_RECLAIM) && KQ_POL_PREEMPT_TRYNEXT(pol)) return true;
  if (borrow != 0 && KQ_POL_BORROW_TRYNEXT(pol)) return true;
  return false;
}
KQ_DEV int fa_mode(int pm) { return pm == PM_NOFIT ? M_NOFIT : (pm == PM_FIT ? M_FIT : M_PREEMPT); }

// resourcegroups.RGByResource (util/resourcegroups/resourcegroups.go:62)
KQ_DEV int rg_by_resource(const DSnap& S, int cq, int res) {
  const int l = S.cq_res_rg[(size_t)cq * S.nR + res];
  return l < 0 ? -1 : S.cq_rg_off[cq] + l;
}
KQ_DEV int64_t assumed_usage(const Wave& w, int fr) {
  for (int i = 0; i < w.nuse; i++) if (w.use_fr[i] == fr) return w.use_qty[i];
  return 0;
}

// ------------------------------------------------------------------------------------------------
// classical victim search (preemption.go:284-339) on a private, slot-restricted copy of the tree
// ------------------------------------------------------------------------------------------------
struct Search {
  const K* k;
  Wave* w;
  int slot;                 // wave slot -> scratch rows
  const int64_t* usage;     // state the search starts from
  const uint8_t* removed;   // rows deleted from cq.Workloads (NULL = none)
  int64_t* W;               // private usage [local_node][ns]
  uint8_t* cqinfo;
  uint8_t* cls;
  int32_t* cand; uint64_t* mark;
  int32_t* trow;
  uint8_t* treason;
  int tree, row0, nrows;
  int32_t* qcnt; uint32_t* qhead; uint8_t* cohp;  // fair sharing only
  int64_t* psum; int32_t* ppos;                   // fair sharing with C.fs_plain, else NULL
};

KQ_DEV bool row_removed(const Search& s, int row) { return s.removed && s.removed[row]; }

// SubtreeQuota >= Usage on every fr needing preemption (resource_node.go:247-254), private state
KQ_DEV bool within_nominal_w(const Search& s, int n) {
  const DSnap& S = s.k->S; const Wave& w = *s.w;
  for (int u = 0; u < w.ns; u++) {
    if (!w.s_need[u]) continue;
    if (S.sq[ix(S, n, w.s_fr[u])] < s.W[(size_t)S.node_local[n] * w.ns + u]) return false;
  }
  return true;
}
KQ_DEV bool within_nominal_g(const Search& s, int n) {
  const DSnap& S = s.k->S; const Wave& w = *s.w;
  for (int u = 0; u < w.ns; u++) {
    if (!w.s_need[u]) continue;
    if (S.sq[ix(S, n, w.s_fr[u])] < s.usage[ix(S, n, w.s_fr[u])]) return false;
  }
  return true;
}
// level of node on the preemptor path, or -1
KQ_DEV int path_level(const Wave& w, int n) {
  for (int i = 1; i < w.plen; i++) if (w.path[i] == n) return i;
  return -1;
}
// common/preemption_policy.go:27-42
KQ_DEV bool satisfies_policy(const K& k, const Wave& w, int row, int policy) {
  int64_t cp = k.S.adm_prio[row];
  bool lower = w.prio > cp;
  if (policy == KQ_POLICY_LOWER_PRIORITY) return lower;
  if (policy == KQ_POLICY_LOWER_OR_NEWER_EQUAL) return lower || (w.prio == cp && w.ts < k.S.adm_qts[row]);
  return policy == KQ_POLICY_ANY;
}
// candidate_generator.go:54-63 restricted to the need slots
KQ_DEV bool uses_need(const K& k, const Wave& w, int row) {
  for (int e = k.S.adm_use_off[row]; e < k.S.adm_use_off[row + 1]; e++) {
    int fr = k.S.adm_use_fr[e];
    for (int u = 0; u < w.ns; u++) if (w.s_need[u] && w.s_fr[u] == fr) return true;
  }
  return false;
}
// hierarchical_preemption.go:81-113 ; returns (variant) and list id (0 hierarchy, 1 priority, 2 same queue)
KQ_DEV int classify_row(const Search& s, int row, int* list, int* bytes) {
  const K& k = *s.k; const Wave& w = *s.w; const DSnap& S = k.S;
  if (row_removed(s, row)) return V_NEVER;
  int c = S.adm_cq[row];
  bool same = c == w.cq;
  int level = 0;
  if (same) {
    if (KQ_POL_WITHIN_CQ(w.pol) == KQ_POLICY_NEVER) return V_NEVER;
  } else {
    uint8_t info = s.cqinfo[S.cq_local[c]];
    if (info == 0) return V_NEVER;
    level = info - 1;
  }
  (void)bytes;  // candidate-record bytes are charged per collected ClusterQueue (cq_row_bytes); the row uses a needed
                // flavor-resource by construction (it comes from that flavor-resource's bucket)
  int policy = same ? KQ_POL_WITHIN_CQ(w.pol) : KQ_POL_RECLAIM(w.pol);
  if (!satisfies_policy(k, w, row, policy)) return V_NEVER;
  if (same) { *list = 2; return V_WITHIN_CQ; }
  if (w.adv_at[level]) { *list = 0; return V_HIER; }
  *list = 1;
  if (KQ_POL_BORROW_WITHIN(w.pol) == 0) return V_RECLAIM_NO_BORROW;           // IsBorrowingWithinCohortForbidden :71
  int64_t cp = S.adm_prio[row];                                                // isAboveBorrowingThreshold :115
  if (cp >= w.prio) return V_RECLAIM_NO_BORROW;
  if (KQ_POL_HAS_THRESHOLD(w.pol) && cp > (int64_t)S.cq_thr[w.cq]) return V_RECLAIM_NO_BORROW;
  return V_RECLAIM_BORROW;
}
KQ_DEV int variant_reason(int v) {  // hierarchical_preemption.go:46-58
  switch (v) {
    case V_WITHIN_CQ: return KQ_REASON_IN_CLUSTER_QUEUE;
    case V_RECLAIM_BORROW: return KQ_REASON_IN_COHORT_RECLAIM_WHILE_BORROWING;
    default: return KQ_REASON_IN_COHORT_RECLAMATION;
  }
}

// snapshot.RemoveWorkload / AddWorkload (snapshot.go:60-74) on the private state: one lane per slot
KQ_DEV void w_apply_row(const Search& s, int row, bool add) {
  const DSnap& S = s.k->S; const Wave& w = *s.w;
  int c = S.adm_cq[row];
  const int32_t* cpath = S.path + (size_t)c * KQ_MAXD;
  int cplen = S.plen[c];
  for (int u = lane_id(); u < w.ns; u += WAVE) {
    int fr = w.s_fr[u];
    for (int e = S.adm_use_off[row]; e < S.adm_use_off[row + 1]; e++) {
      if (S.adm_use_fr[e] != fr) continue;
      UW uw{&S, s.W, w.ns, u};
      if (add) add_usage(S, cpath, cplen, fr, S.adm_use_qty[e], uw); else remove_usage(S, cpath, cplen, fr, S.adm_use_qty[e], uw);
    }
  }
  wsync();
  if (lane_id() == 0) s.w->bytes += 16 * (int64_t)cplen * (S.adm_use_off[row + 1] - S.adm_use_off[row]);
#ifdef KQ_TAS_CYCLE
  if (s.k->tc) tc_search_row(*s.k, *s.w, s.slot, row, add);
#endif
}
// preemption.go:669-686 on the private state (quota part)
KQ_DEV bool w_fits(const Search& s, bool allow_borrowing) {
  const DSnap& S = s.k->S; Wave& w = *s.w;
  bool bad = false;
  for (int u = lane_id(); u < w.ns; u += WAVE) {
    if (!w.s_inu[u]) continue;
    UW uw{&S, s.W, w.ns, u};
    int fr = w.s_fr[u];
    int64_t v = w.s_qty[u];
    if (!allow_borrowing && borrowing_with(S, w.cq, true, fr, v, uw)) bad = true;
    else if (v > i64max(0, available_of(S, w.path, w.plen, fr, uw))) bad = true;
  }
  if (lane_id() == 0) { int nu = 0; for (int u = 0; u < w.ns; u++) nu += w.s_inu[u] ? 1 : 0; w.bytes += 40 * (int64_t)w.plen * nu; }
#ifdef KQ_TAS_CYCLE
  if (wballot(bad) != 0) return false;
  return s.k->tc ? tc_search_fits(*s.k, w, s.slot) : true;   // preemption.go:676-684: the placement on the state without the victims so far
#else
  return wballot(bad) == 0;
#endif
}
// candidate_generator.go:136-158
KQ_DEV bool candidate_valid(const Search& s, int row, int variant, bool borrow) {
  const DSnap& S = s.k->S; const Wave& w = *s.w;
  int c = S.adm_cq[row];
  if (c == w.cq) return true;
  if (borrow && variant == V_RECLAIM_NO_BORROW) return false;
  if (within_nominal_w(s, c)) return false;
  int lca = w.path[s.cqinfo[S.cq_local[c]] - 1];
  for (int n = S.parent[c]; n >= 0 && n != lca; n = S.parent[n]) if (within_nominal_w(s, n)) return false;
  return true;
}

}  // namespace kq
#include "kq_cs.hpp"
namespace kq {
// Runs the classical search for the slots prepared in w (s_fr/s_qty/s_inu/s_need).
// On return w->ntgt targets are in s.trow/s.treason and the private state has exactly those removed.
KQ_DEV void classical_search(Search& s) {
  const K& k = *s.k; Wave& w = *s.w; const DSnap& S = k.S;
  const int lane = lane_id();
  w.ntgt = 0;
#ifdef KQ_HOST_EMU
  const int64_t bytes_entry = w.bytes;
#endif
  CSTAT(0, 1); if (s.removed) CSTAT(10, 1);
  bool same_on = KQ_POL_WITHIN_CQ(w.pol) != KQ_POLICY_NEVER;
  bool other_on = w.plen > 1 && KQ_POL_RECLAIM(w.pol) != KQ_POLICY_NEVER;
  if (!same_on && !other_on) return;
  s.tree = S.tree_of[w.cq];
  s.row0 = S.tree_row_off[s.tree];
  s.nrows = S.tree_row_off[s.tree + 1] - s.row0;
  if (s.nrows == 0) return;
  const int n0 = S.tree_node_off[s.tree], nn = S.tree_node_off[s.tree + 1] - n0;
  // hasHierarchicalAdvantage per ancestor level (hierarchical_preemption.go:149-175), one lane per slot
  {
    bool adv;
    {
      bool nf = false;
      for (int u = lane; u < w.ns; u += WAVE)
        if (w.s_inu[u] && S.sq[ix(S, w.cq, w.s_fr[u])] < a_add(s.usage[ix(S, w.cq, w.s_fr[u])], w.s_qty[u])) nf = true;
      adv = wballot(nf) == 0;
    }
    // remaining requests per slot live in registers of lane (u % WAVE); ns <= KQ_MAXU so iterate slots serially per lane
    int64_t rem[(KQ_MAXU + WAVE - 1) / WAVE];
    for (int j = 0, u = lane; u < w.ns; u += WAVE, j++) {
      UG ug{&S, s.usage, w.s_fr[u]};
      rem[j] = w.s_inu[u] ? i64max(0, a_sub(w.s_qty[u], i64max(0, a_sub(local_quota(S, w.cq, w.s_fr[u]), ug.get(w.cq))))) : 0;
    }
    for (int l = 1; l < w.plen; l++) {
      if (lane == 0) w.adv_at[l] = adv ? 1 : 0;
      int a = w.path[l];
      bool nf = false;
      for (int j = 0, u = lane; u < w.ns; u += WAVE, j++) {
        if (!w.s_inu[u]) continue;
        int fr = w.s_fr[u];
        int64_t uu = s.usage[ix(S, a, fr)];
        if (S.sq[ix(S, a, fr)] < a_add(uu, rem[j])) nf = true;
        rem[j] = i64max(0, a_sub(rem[j], i64max(0, a_sub(local_quota(S, a, fr), uu))));
      }
      bool fits = wballot(nf) == 0;
      adv = adv || fits;
    }
  }
  // which CQs of the tree are collected, and under which path level (collectCandidatesInSubtree :179-207)
  int cbytes = 0;
  {
    const int q0 = S.tree_cq_off[s.tree], nqs = S.tree_cq_off[s.tree + 1] - q0;
    for (int i = lane; i < nqs; i += WAVE) {
      int c = S.tree_cqs[q0 + i];
      uint8_t info = 0;
      if (other_on && c != w.cq && !within_nominal_g(s, c)) {
        bool ok = true;
        int n = S.parent[c], lvl = -1;
        while (n >= 0) {
          lvl = path_level(w, n);
          if (lvl >= 0) break;
          if (within_nominal_g(s, n)) { ok = false; break; }
          n = S.parent[n];
        }
        if (ok && lvl >= 1) info = (uint8_t)(lvl + 1);
      }
      s.cqinfo[i] = info;
      // every row still in cq.Workloads of a collected ClusterQueue is looked at (candidate record + usage entries)
      if (info != 0 || (c == w.cq && same_on)) cbytes += S.cq_row_bytes[c] - (s.removed ? k.cq_rm_bytes[c] : 0);
    }
  }
  {
    int64_t tot = wsum_i64((int64_t)cbytes);
    if (lane == 0) w.bytes += tot;
  }
  wsync();
#ifdef KQ_HOST_EMU
  if (g_cs_check && !g_cs_force_off && !KQ_TAS_WALK(w)) {
    if (cs_run(s, same_on, other_on)) {
      const int nt1 = w.ntgt; const int64_t b1 = w.bytes;
      std::vector<int> t1(s.trow, s.trow + nt1);
      std::sort(t1.begin(), t1.end());
      std::vector<int64_t> pw;
      for (int l = 0; l < w.plen; l++) for (int u = 0; u < w.ns; u++) pw.push_back(s.W[(size_t)S.node_local[w.path[l]] * w.ns + u]);
      Search s2 = s;
      w.bytes = bytes_entry;  // the nested call charges the candidate records again
      g_cs_force_off = 2;     // nested marker: walk
      classical_search(s2);
      if (w.bytes != b1) { CSTAT(22, 1); fprintf(stderr, "CS BYTES MISMATCH head %d: scan %lld walk %lld\n", w.h, (long long)(b1 - bytes_entry), (long long)(w.bytes - bytes_entry)); }
      g_cs_force_off = 0;
      std::vector<int> t2(s2.trow, s2.trow + w.ntgt);
      std::sort(t2.begin(), t2.end());
      bool same = nt1 == w.ntgt && t1 == t2;
      if (same && nt1 > 0) { int q = 0; for (int l = 0; l < w.plen; l++) for (int u = 0; u < w.ns; u++) if (pw[q++] != s.W[(size_t)S.node_local[w.path[l]] * w.ns + u]) same = false; }
      if (!same) {
        CSTAT(22, 1);
        fprintf(stderr, "CS MISMATCH head %d cq %d ns %d plen %d: scan ntgt %d walk ntgt %d\n", w.h, w.cq, w.ns, w.plen, nt1, w.ntgt);
        fprintf(stderr, "  scan:"); for (int x : t1) fprintf(stderr, " %d", x); fprintf(stderr, "\n  walk:"); for (int x : t2) fprintf(stderr, " %d", x); fprintf(stderr, "\n");
      }
      return;
    }
  }
  if (g_cs_force_off != 0) { /* walk */ } else
#endif
  if (!KQ_TAS_WALK(w) && cs_run(s, same_on, other_on)) return;  // the same search as segmented scans (kq_cs.hpp); false: outside its preconditions
  CSTAT(21, 1);
  // private copy of the tree's usage for the slots
  for (int i = lane; i < nn * w.ns; i += WAVE) {
    int ln = i / w.ns, u = i % w.ns;
    s.W[i] = s.usage[ix(S, S.tree_nodes[n0 + ln], w.s_fr[u])];
  }
  wsync();
  // candidate positions: the rows that use a flavor-resource needing preemption (WorkloadUsesResources), rank order
  const int32_t* positions;
  int M = 0;
  {
    int nneed = 0, one = -1;
    for (int u = 0; u < w.ns; u++) if (w.s_need[u]) { nneed++; one = u; }
    if (nneed == 1) {
      const size_t b = (size_t)s.tree * S.nfr + w.s_fr[one];
      positions = S.frb + S.frb_off[b];
      M = S.frb_off[b + 1] - S.frb_off[b];
    } else {
      const int nwords = (s.nrows + 63) / 64;
      for (int i = lane; i < nwords; i += WAVE) s.mark[i] = 0;
      wsync();
      for (int u = 0; u < w.ns; u++) {
        if (!w.s_need[u]) continue;
        const size_t b = (size_t)s.tree * S.nfr + w.s_fr[u];
        for (int j = S.frb_off[b] + lane; j < S.frb_off[b + 1]; j += WAVE) { const int pos = S.frb[j]; atomic_or_u64(&s.mark[pos >> 6], 1ull << (pos & 63)); }
      }
      wsync();
      for (int base = 0; base < nwords; base += WAVE) {
        const int i = base + lane;
        uint64_t wd = i < nwords ? s.mark[i] : 0;
        const int c = popc64(wd);
        w.cell_borrow[lane] = c;
        wsync_lds();
        int before = 0, total = 0;
        for (int q = 0; q < WAVE; q++) { const int v = w.cell_borrow[q]; if (q < lane) before += v; total += v; }
        int my = M + before;
        while (wd) { const int bit = ffs64(wd); wd &= wd - 1; s.cand[my++] = i * 64 + bit; }
        M += total;
        wsync_lds();
      }
      wsync();
      positions = s.cand;
    }
  }
  // classify the candidate rows once; class byte = 1 + list + 3*evicted + 8*variant (indexed like `positions`)
  int cnt[6] = {0, 0, 0, 0, 0, 0};
  for (int base = 0; base < M; base += WAVE) {
    int j = base + lane;
    uint8_t cb = 0;
    if (j < M) {
      int row = S.tree_rows[s.row0 + positions[j]], list = 0, unused = 0;
      int v = classify_row(s, row, &list, &unused);
      if (v != V_NEVER) {
        int ev = (S.adm_flags[row] & KQ_ADM_EVICTED) ? 0 : 1;  // evicted first
        cb = (uint8_t)(1 + (ev * 3 + list) + 8 * v);
      }
      s.cls[j] = cb;
    }
    for (int p = 0; p < 6; p++) cnt[p] += popc64(wballot(cb != 0 && ((cb - 1) & 7) == p));
  }
  wsync();
  CSTAT(1, M); CSTAT(2, cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4] + cnt[5]); CSTAT(9, w.ns);
  bool no_hier = cnt[0] + cnt[3] == 0, no_other = no_hier && (cnt[1] + cnt[4] == 0);
  if (cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4] + cnt[5] == 0) return;
  bool forbidden = KQ_POL_BORROW_WITHIN(w.pol) == 0;
  bool under_nominal;  // queueUnderNominalInResourcesNeedingPreemption preemption.go:700-707
  {
    bool nb = false;
    for (int u = lane; u < w.ns; u += WAVE)
      if (w.s_need[u] && S.nominal[ix(S, w.cq, w.s_fr[u])] <= s.usage[ix(S, w.cq, w.s_fr[u])]) nb = true;
    under_nominal = wballot(nb) == 0;
  }
  int nattempt; bool attempts[2];
  if (no_other || (forbidden && !under_nominal)) { nattempt = 1; attempts[0] = true; attempts[1] = true; }
  else if (forbidden && no_hier) { nattempt = 2; attempts[0] = false; attempts[1] = true; }
  else { nattempt = 2; attempts[0] = true; attempts[1] = false; }
  for (int at = 0; at < nattempt; at++) {
    bool borrowing = attempts[at];
    int nt = 0;
    CSTAT(3, 1);
    for (int p = 0; p < 6; p++) {
      if (cnt[p] == 0) continue;
      for (int base = 0; base < M; base += WAVE) {
        int i = base + lane;
        uint8_t cb = i < M ? s.cls[i] : 0;
        uint64_t m = wballot(cb != 0 && ((cb - 1) & 7) == p);
        while (m) {
          int b = ffs64(m);
          m &= m - 1;
          int pos = base + b;
          int row = S.tree_rows[s.row0 + positions[pos]];
          int variant = s.cls[pos] >> 3;
          CSTAT(4, 1);
          if (!candidate_valid(s, row, variant, borrowing)) continue;
          CSTAT(5, 1);
          w_apply_row(s, row, false);
          if (nt >= k.X.tgt_cap) { set_error(k, KQ_ECAPACITY); w.ntgt = 0; return; }
          if (lane == 0) { s.trow[nt] = row; s.treason[nt] = (uint8_t)variant_reason(variant); }
          nt++;
          wsync();
          if (w_fits(s, borrowing)) {
            // fillBackWorkloads preemption.go:341-354
            for (int t = nt - 2; t >= 0; t--) {
              int r = s.trow[t];
              w_apply_row(s, r, true);
              if (w_fits(s, borrowing)) {
                if (lane == 0) { s.trow[t] = s.trow[nt - 1]; s.treason[t] = s.treason[nt - 1]; }
                nt--;
                wsync();
              } else {
                w_apply_row(s, r, false);
              }
            }
            w.ntgt = nt;
            CSTAT(6, 1); CSTAT(7, nt);
            // the reference now re-adds the targets (restoreSnapshot :356); the private copy is simply
            // dropped here, but those writes are part of the algorithm's traffic
            if (lane == 0)
              for (int t = 0; t < nt; t++) {
                int r = s.trow[t];
                w.bytes += 16 * (int64_t)S.plen[S.adm_cq[r]] * (S.adm_use_off[r + 1] - S.adm_use_off[r]);
              }
            return;
          }
        }
      }
    }
    // restoreSnapshot :356
    for (int t = 0; t < nt; t++) w_apply_row(s, s.trow[t], true);
  }
  w.ntgt = 0;
}

// ------------------------------------------------------------------------------------------------
// fair sharing: DominantResourceShare (cache/scheduler/fair_sharing.go)
// ------------------------------------------------------------------------------------------------
struct DRSv { double ratio, weight; int borrowing, borrow_on; };
KQ_DEV DRSv drs_negative() { DRSv d; d.ratio = -1.0; d.weight = 1.0; d.borrowing = 0; d.borrow_on = 0; return d; }  // :58
// PreciseWeightedShare :92
KQ_DEV double drs_pws(const DRSv& d) {
  if (d.ratio == 0) return 0.0;
  if (d.weight == 0) return __builtin_inf();
  return d.ratio / d.weight;
}
KQ_DEV bool drs_zwb(const DRSv& d) { return d.weight == 0 && d.ratio != 0; }  // zeroWeightBorrows :145
KQ_DEV int cmp_f64(double a, double b) {  // Go cmp.Compare
  bool an = a != a, bn = b != b;
  if (an && bn) return 0;
  if (an) return -1;
  if (bn) return 1;
  return a < b ? -1 : (a > b ? 1 : 0);
}
KQ_DEV int compare_drs(const DRSv& a, const DRSv& b) {  // :112-123
  if (drs_zwb(a) && drs_zwb(b)) return cmp_f64(a.ratio, b.ratio);
  if (drs_zwb(a)) return 1;
  if (drs_zwb(b)) return -1;
  return cmp_f64(drs_pws(a), drs_pws(b));
}
KQ_DEV bool drs_pos_inf(const DRSv& d) { double v = drs_pws(d); return v == __builtin_inf(); }
// dominantResourceShare :149-182 for `node` on usage plane `pl` (pl.get(node, fr)). calculateLendable(parent) is
// the per-snapshot constant S.lendable. rq_*: the entry's requested flavor-resources (IsBorrowingOn :78), or n = 0.
// Runs per lane: lanes may evaluate different nodes.
template <class PL> KQ_DEV DRSv drs_of(const DSnap& S, int node, const PL& pl, int64_t* bytes,
                                        const int32_t* rq_fr, const int64_t* rq_qty, int rq_n) {
  DRSv d; d.ratio = 0; d.weight = S.fair_weight[node]; d.borrowing = 0; d.borrow_on = 0;
  if (S.parent[node] < 0) return d;
  const int nR = S.nR, nF = S.nfr / S.nR;
  int frcount = 0;
  for (int r = 0; r < nR; r++) {
    int64_t sum = 0; bool any = false;
    for (int f = 0; f < nF; f++) {
      int fr = f * nR + r;
      size_t o = ix(S, node, fr);
      if (!(S.qflags[o] & KQ_QF_SUBTREE)) continue;
      frcount++;
      int64_t b = a_sub(pl.get(node, fr), S.sq[o]);
      if (b > 0) {
        sum = any ? a_add(sum, b) : b; any = true;
        for (int q = 0; q < rq_n; q++) if (rq_fr[q] == fr && rq_qty[q] > 0) d.borrow_on = 1;
      }
    }
    if (any) {
      d.borrowing = 1;
      int64_t lr = S.lendable[(size_t)node * nR + r];
      if (lr > 0) {
        double ratio = (double)sum * 1000.0 / (double)lr;
        if (ratio > d.ratio) d.ratio = ratio;  // ascending resource index == alphabetical, ties keep the first (:176)
      }
    }
  }
  *bytes += (int64_t)frcount * 24 + (d.borrowing ? (int64_t)frcount * 40 * (S.depth[node] + 1) : 0);
  return d;
}

// borrowed sums of one node from a plane (all flavor-resources of resource r), plain int64 (C.fs_plain)
template <class PL> KQ_DEV void node_sums(const DSnap& S, int node, const PL& pl, int r, int64_t* sum, int* pos) {
  const int nR = S.nR, nF = S.nfr / S.nR;
  int64_t s = 0; int p = 0;
  for (int f = 0; f < nF; f++) {
    int fr = f * nR + r;
    size_t o = ix(S, node, fr);
    if (!(S.qflags[o] & KQ_QF_SUBTREE)) continue;
    int64_t b = a_sub(pl.get(node, fr), S.sq[o]);
    if (b > 0) { s += b; p++; }
  }
  *sum = s; *pos = p;
}
KQ_DEV DRSv drs_from_sums(const DSnap& S, int node, const int64_t* sum, int pos, int64_t* bytes) {
  DRSv d; d.ratio = 0; d.weight = S.fair_weight[node]; d.borrowing = pos > 0; d.borrow_on = 0;
  for (int r = 0; r < S.nR; r++) {
    if (sum[r] <= 0) continue;
    int64_t lr = S.lendable[(size_t)node * S.nR + r];
    if (lr > 0) { double ratio = (double)sum[r] * 1000.0 / (double)lr; if (ratio > d.ratio) d.ratio = ratio; }
  }
  const int frcount = S.frcount[node];
  *bytes += (int64_t)frcount * 24 + (d.borrowing ? (int64_t)frcount * 40 * (S.depth[node] + 1) : 0);
  return d;
}
// ------------------------------------------------------------------------------------------------
// fair-sharing victim search (preemption.go:381-631, fairsharing/ordering.go, strategy.go, target.go)
// The private state covers every flavor-resource of every node of the tree: DRS reads all of them.
// ------------------------------------------------------------------------------------------------
struct UF {  // one flavor-resource column of the private plane; keeps the node's borrowed sums current
  const DSnap* S; int64_t* w; int fr; int64_t* psum; int32_t* ppos;
  KQ_MDEV int64_t get(int n) const { return w[(size_t)S->node_local[n] * S->nfr + fr]; }
  KQ_MDEV void set(int n, int64_t v) const {
    const int li = S->node_local[n];
    int64_t* cell = w + (size_t)li * S->nfr + fr;
    if (psum) {
      const size_t o = (size_t)n * S->nfr + fr;
      if (S->qflags[o] & KQ_QF_SUBTREE) {
        const int64_t sqv = S->sq[o];
        const int64_t ob = i64max(0, a_sub(*cell, sqv)), nb = i64max(0, a_sub(v, sqv));
        if (nb != ob) atomic_add_i64((long long*)&psum[(size_t)li * S->nR + fr % S->nR], (long long)(nb - ob));
        const int dp = (nb > 0 ? 1 : 0) - (ob > 0 ? 1 : 0);
        if (dp) atomic_add_i32(&ppos[li], dp);
      }
    }
    *cell = v;
  }
};
struct PF {  // the whole private plane
  const DSnap* S; const int64_t* w;
  KQ_MDEV int64_t get(int n, int fr) const { return w[(size_t)S->node_local[n] * S->nfr + fr]; }
};
struct PG {  // a global plane
  const DSnap* S; const int64_t* u;
  KQ_MDEV int64_t get(int n, int fr) const { return u[(size_t)n * S->nfr + fr]; }
};

// snapshot.RemoveWorkload / AddWorkload or plain Remove/AddUsage of the row's usage: one lane per usage entry
// tas: the row's leaf usage follows on the search's private TAS plane (kq_cycle_run_tas under fair sharing; snapshot.RemoveWorkload /
// AddWorkload carry Usage.TAS). The remove/add pair around ComputeTargetShareAfterRemoval cancels and leaves it alone.
KQ_DEV void f_apply_row(const Search& s, int row, bool add, bool count, bool tas) {
  const DSnap& S = s.k->S;
  KQ_A0();
  int c = S.adm_cq[row];
  const int32_t* cpath = S.path + (size_t)c * KQ_MAXD;
  int cplen = S.plen[c];
  const int e0 = S.adm_use_off[row], e1 = S.adm_use_off[row + 1];
  for (int e = e0 + lane_id(); e < e1; e += WAVE) {
    int fr = S.adm_use_fr[e];
    bool first = true;
    for (int p = e0; p < e; p++) if (S.adm_use_fr[p] == fr) first = false;
    if (!first) continue;  // a repeated flavor-resource is applied by the lane of its first entry, in order
    UF uf{&S, s.W, fr, s.psum, s.ppos};
    for (int p = e; p < e1; p++) {
      if (S.adm_use_fr[p] != fr) continue;
      if (add) add_usage(S, cpath, cplen, fr, S.adm_use_qty[p], uf); else remove_usage(S, cpath, cplen, fr, S.adm_use_qty[p], uf);
    }
  }
  wsync();
  if (count && lane_id() == 0) s.w->bytes += 16 * (int64_t)cplen * (e1 - e0);
#ifdef KQ_TAS_CYCLE
  if (tas && s.k->tc) tc_search_row(*s.k, *s.w, s.slot, row, add);
#else
  (void)tas;
#endif
  KQ_AS(*s.k, 49);
}
// cq.SimulateUsageAddition(workloadUsage) / its revert (preemption.go:557, :690-695)
KQ_DEV void f_apply_preemptor(const Search& s, bool add) {
  const DSnap& S = s.k->S; const Wave& w = *s.w;
  for (int u = lane_id(); u < w.ns; u += WAVE) {
    if (!w.s_inu[u]) continue;
    UF uf{&S, s.W, w.s_fr[u], s.psum, s.ppos};
    if (add) add_usage(S, w.path, w.plen, w.s_fr[u], w.s_qty[u], uf); else remove_usage(S, w.path, w.plen, w.s_fr[u], w.s_qty[u], uf);
  }
  wsync();
}
// workloadFits(ctx, allowBorrowing=true) preemption.go:669-686
KQ_DEV bool f_fits(const Search& s) {
  const DSnap& S = s.k->S; Wave& w = *s.w;
  bool bad = false;
  for (int u = lane_id(); u < w.ns; u += WAVE) {
    if (!w.s_inu[u]) continue;
    UF uf{&S, s.W, w.s_fr[u], s.psum, s.ppos};
    if (w.s_qty[u] > i64max(0, available_of(S, w.path, w.plen, w.s_fr[u], uf))) bad = true;
  }
  if (lane_id() == 0) { int nu = 0; for (int u = 0; u < w.ns; u++) nu += w.s_inu[u] ? 1 : 0; w.bytes += 40 * (int64_t)w.plen * nu; }
#ifdef KQ_TAS_CYCLE
  if (wballot(bad) != 0) return false;
  return s.k->tc ? tc_search_fits(*s.k, w, s.slot) : true;   // preemption.go:676-684: the placement on the leaf usage without the victims so far
#else
  return wballot(bad) == 0;
#endif
}
KQ_DEV bool f_fits_fs(const Search& s) {  // workloadFitsForFairSharing :690-695
  KQ_A0();
  f_apply_preemptor(s, false);
  bool r = f_fits(s);
  f_apply_preemptor(s, true);
  KQ_AS(*s.k, 50);
  return r;
}
KQ_DEV uint32_t f_row_key(const DSnap& S, int row) {
  return ((S.adm_flags[row] & KQ_ADM_EVICTED) ? 0u : 0x80000000u) | (uint32_t)S.rank_pos[row];
}
// CandidatesOrdering (common/ordering.go:42-83) of the queue heads of two CQs, as a sortable key
KQ_DEV uint64_t f_head_key(const Search& s, int c) {
  uint32_t k32 = s.qhead[s.k->S.cq_local[c]];
  return ((uint64_t)(k32 >> 31) << 33) | ((uint64_t)(c == s.w->cq ? 1 : 0) << 32) | (uint64_t)(k32 & 0x7fffffffu);
}
// recount / re-head the queue of tree-local CQ `i` over rows flagged `flag`: one lane per row, wave min
KQ_DEV void f_rescan(const Search& s, int i, uint8_t flag) {
  const DSnap& S = s.k->S;
  const int c = S.tree_cqs[S.tree_cq_off[s.tree] + i];
  const int r0 = S.cq_adm_off[c], nr = S.cq_adm_off[c + 1] - r0;
  uint64_t head = 0xffffffffull; int cnt = 0;
  for (int base = 0; base < nr; base += WAVE) {
    const int j = base + lane_id();
    uint64_t key = 0xffffffffull;
    bool in = false;
    if (j < nr) { const int row = r0 + j; in = s.cls[S.rank_pos[row]] == flag; if (in) key = f_row_key(S, row); }
    cnt += popc64(wballot(in));
    const uint64_t mn = wmin_u64(key);
    if (mn < head) head = mn;
  }
  if (lane_id() == 0) { s.qcnt[i] = cnt; s.qhead[i] = (uint32_t)head; }
  wsync();
}
// PopWorkload (ordering.go:84-90): the popped row's flag becomes `newflag`
KQ_DEV int f_pop(const Search& s, int c, uint8_t flag, uint8_t newflag) {
  const DSnap& S = s.k->S;
  CSTAT(13, 1);
  KQ_A0();
  int i = S.cq_local[c];
  int pos = (int)(s.qhead[i] & 0x7fffffffu);
  int row = S.tree_rows[s.row0 + pos];
  wsync();
  if (lane_id() == 0) s.cls[pos] = newflag;
  wsync();
  f_rescan(s, i, flag);
  KQ_AS(*s.k, 48);
  return row;
}
KQ_DEV DRSv f_drs(const Search& s, int node, int64_t* lb) {
  const DSnap& S = s.k->S;
  if (s.psum) { const int li = S.node_local[node]; return drs_from_sums(S, node, s.psum + (size_t)li * S.nR, s.ppos[li], lb); }
  PF pf{&S, s.W};
  return drs_of(S, node, pf, lb, nullptr, nullptr, 0);
}
KQ_DEV DRSv f_drs_uniform(const Search& s, int node) {
  int64_t lb = 0;
  DRSv d = f_drs(s, node, &lb);
  if (lane_id() == 0) s.w->bytes += lb;
  return d;
}
// nextTarget (ordering.go:144-226), tail recursion unrolled; returns the target CQ or -1
KQ_DEV int f_next_target(const Search& s, int root) {
  const K& k = *s.k; Wave& w = *s.w; const DSnap& S = k.S;
  const int lane = lane_id();
  CSTAT(11, 1);
  int cohort = root;
  int64_t lb = 0;
  int result = -1;
  for (;;) {
    CSTAT(12, 1);
    int best_cq = -1; DRSv best = drs_negative();
    const int kc0 = S.child_cq_off[cohort - S.nq], nkc = S.child_cq_off[cohort - S.nq + 1] - kc0;
    for (int base = 0; base < nkc; base += WAVE) {
      int j = base + lane;
      int c = j < nkc ? S.child_cq[kc0 + j] : -1;
      DRSv d = drs_negative();
      bool elig = false;
      if (c >= 0 && !s.cqinfo[S.cq_local[c]]) {
        d = f_drs(s, c, &lb);
        if ((!d.borrowing && c != w.cq) || s.qcnt[S.cq_local[c]] == 0) s.cqinfo[S.cq_local[c]] = 1;
        else elig = true;
      }
      uint64_t m = wballot(elig);
      while (m) {
        int b = ffs64(m);
        m &= m - 1;
        DRSv db; db.ratio = wbcast(d.ratio, b); db.weight = wbcast(d.weight, b); db.borrowing = 1; db.borrow_on = 0;
        int cb = wbcast(c, b);
        int cmp = compare_drs(db, best);
        if (cmp == 0) { if (f_head_key(s, cb) < f_head_key(s, best_cq)) best_cq = cb; }
        else if (cmp == 1) { best = db; best_cq = cb; }
      }
    }
    int best_co = -1; DRSv bestco = drs_negative();
    const int kh0 = S.child_cohort_off[cohort - S.nq], nkh = S.child_cohort_off[cohort - S.nq + 1] - kh0;
    for (int base = 0; base < nkh; base += WAVE) {
      int j = base + lane;
      int ch = j < nkh ? S.child_cohort[kh0 + j] : -1;
      DRSv d = drs_negative();
      bool elig = false;
      if (ch >= 0 && !s.cohp[S.node_local[ch]]) {
        d = f_drs(s, ch, &lb);
        if (!d.borrowing && path_level(w, ch) < 0) s.cohp[S.node_local[ch]] = 1;
        else elig = true;
      }
      uint64_t m = wballot(elig);
      while (m) {
        int b = ffs64(m);
        m &= m - 1;
        DRSv db; db.ratio = wbcast(d.ratio, b); db.weight = wbcast(d.weight, b); db.borrowing = 1; db.borrow_on = 0;
        int hb = wbcast(ch, b);
        if (compare_drs(db, bestco) >= 0) { bestco = db; best_co = hb; }
      }
    }
    wsync();
    if (best_co < 0 && best_cq < 0) {
      if (lane == 0) s.cohp[S.node_local[cohort]] = 1;
      wsync();
      result = -1;
      break;
    }
    if (compare_drs(bestco, best) >= 0) { cohort = best_co; continue; }
    result = best_cq;
    break;
  }
  int64_t tot = wsum_i64(lb);
  if (lane == 0) w.bytes += tot;
  return result;
}
// TargetClusterQueueOrdering.Iter (ordering.go:92-127) as a "next" call
KQ_DEV int f_ordering_next(const Search& s) {
  const DSnap& S = s.k->S; const Wave& w = *s.w;
  KQ_A0();
  if (w.plen <= 1) {
    int i = S.cq_local[w.cq];
    if (!s.cqinfo[i] && s.qcnt[i] > 0) return w.cq;
    return -1;
  }
  int root = w.path[w.plen - 1];
  while (!s.cohp[S.node_local[root]]) {
    int t = f_next_target(s, root);
    if (t >= 0) { KQ_AS(*s.k, 47); return t; }
  }
  KQ_AS(*s.k, 47);
  return -1;
}
// getAlmostLCAs (least_common_ancestor.go:27-58): nodes just below the LCA on the preemptor's and the target's path
KQ_DEV void f_almost_lcas(const Search& s, int target, int* ap, int* at) {
  const DSnap& S = s.k->S; const Wave& w = *s.w;
  const int32_t* tp = S.path + (size_t)target * KQ_MAXD;
  int tpl = S.plen[target];
  *ap = w.cq; *at = target;
  for (int j = 1; j < tpl; j++) {
    int l = path_level(w, tp[j]);
    if (l >= 1) { *ap = w.path[l - 1]; *at = tp[j - 1]; return; }
  }
  *ap = w.path[w.plen - 1]; *at = tp[tpl - 1];
}
KQ_DEV bool f_push_target(Search& s, int* nt, int row, int reason) {
  if (*nt >= s.k->X.tgt_cap) { set_error(*s.k, KQ_ECAPACITY); return false; }
  if (lane_id() == 0) { s.trow[*nt] = row; s.treason[*nt] = (uint8_t)reason; }
  (*nt)++;
  wsync();
  return true;
}
}  // namespace kq
#include "kq_fs.hpp"
namespace kq {
#ifdef KQ_HOST_EMU
static thread_local int g_fs_check = 0, g_fs_force_off = 0;  // tests: run every LDS-formulated fair search a second time as the walk and compare
#endif
KQ_DEV void fair_search_walk(Search& s);
// fairPreemptions (preemption.go:536-597). On return w->ntgt targets are in s.trow/s.treason and the private
// state has exactly those removed (where callers read it: the preemptor's path).
KQ_DEV void fair_search(Search& s) {
#ifdef KQ_HOST_EMU
  if (g_fs_force_off) { fair_search_walk(s); return; }
  Wave& w = *s.w;
  const int64_t bytes_entry = w.bytes;
  if (!fair_search_lds(s)) { fair_search_walk(s); return; }
  CSTAT(23, 1);
  if (g_fs_check) {
    const DSnap& S = s.k->S;
    const int nt1 = w.ntgt;
    const int64_t b1 = w.bytes;
    std::vector<int32_t> tr(s.trow, s.trow + nt1);
    std::vector<uint8_t> rs(s.treason, s.treason + nt1);
    std::vector<int64_t> pw;
    for (int l = 0; l < w.plen; l++) for (int u = 0; u < w.ns; u++) pw.push_back(s.W[(size_t)S.node_local[w.path[l]] * S.nfr + w.s_fr[u]]);
    w.bytes = bytes_entry;
    fair_search_walk(s);
    bool same = nt1 == w.ntgt && b1 == w.bytes;
    for (int t = 0; same && t < nt1; t++) if (tr[t] != s.trow[t] || rs[t] != s.treason[t]) same = false;
    if (same && nt1 > 0) { int q = 0; for (int l = 0; l < w.plen; l++) for (int u = 0; u < w.ns; u++) { const int64_t v = s.W[(size_t)S.node_local[w.path[l]] * S.nfr + w.s_fr[u]]; if (w.s_inu[u] && pw[q] != v) same = false; q++; } }
    if (!same) { CSTAT(24, 1); fprintf(stderr, "FS MISMATCH head %d: lds %d targets %lld bytes, walk %d targets %lld bytes\n", w.h, nt1, (long long)(b1 - bytes_entry), w.ntgt, (long long)(w.bytes - bytes_entry)); }
  }
#elif defined(KQ_FAIR_WALK_ONLY)
  fair_search_walk(s);
#else
  if (!fair_search_lds(s)) fair_search_walk(s);
#endif
}
KQ_DEV void fair_search_walk(Search& s) {
  const K& k = *s.k; Wave& w = *s.w; const DSnap& S = k.S;
  const int lane = lane_id();
  w.ntgt = 0;
  bool same_on = KQ_POL_WITHIN_CQ(w.pol) != KQ_POLICY_NEVER;
  bool other_on = w.plen > 1 && KQ_POL_RECLAIM(w.pol) != KQ_POLICY_NEVER;
  if (!same_on && !other_on) return;
  s.tree = S.tree_of[w.cq];
  s.row0 = S.tree_row_off[s.tree];
  s.nrows = S.tree_row_off[s.tree + 1] - s.row0;
  if (s.nrows == 0) return;
  const int n0 = S.tree_node_off[s.tree], nn = S.tree_node_off[s.tree + 1] - n0;
  const int q0 = S.tree_cq_off[s.tree], nqs = S.tree_cq_off[s.tree + 1] - q0;
  const int nfr = S.nfr;
  KQ_T0();
  for (int i = lane; i < nn * nfr; i += WAVE) s.W[i] = s.usage[ix(S, S.tree_nodes[n0 + i / nfr], i % nfr)];
  KQ_TS(k, 40);  // fair search: private copy of the plane
  if (s.psum) {
    if (s.usage == k.usage) {  // cycle-start plane: k_fs_sums already reduced it
      for (int i = lane; i < nn * S.nR; i += WAVE) s.psum[i] = k.X.bu_sum[(size_t)S.tree_nodes[n0 + i / S.nR] * S.nR + i % S.nR];
      for (int i = lane; i < nn; i += WAVE) s.ppos[i] = k.X.bu_pos[S.tree_nodes[n0 + i]];
    } else {
      PG pg{&S, s.usage};
      for (int i = lane; i < nn; i += WAVE) {
        int tot = 0;
        for (int r = 0; r < S.nR; r++) { int64_t sm; int ps; node_sums(S, S.tree_nodes[n0 + i], pg, r, &sm, &ps); s.psum[(size_t)i * S.nR + r] = sm; tot += ps; }
        s.ppos[i] = tot;
      }
    }
  }
  for (int i = lane; i < s.nrows; i += WAVE) s.cls[i] = 0;
  for (int i = lane; i < nn; i += WAVE) s.cohp[i] = 0;
  wsync();
  KQ_TS(k, 41);  // fair search: borrowed sums + clears
  // findCandidates (:633-667): one lane per CQ of the tree
  int ncand = 0;
  {
    int cbytes = 0;
    for (int base = 0; base < nqs; base += WAVE) {
      int i = base + lane;
      int cnt = 0; uint32_t head = 0xffffffffu;
      if (i < nqs) {
        int c = S.tree_cqs[q0 + i];
        bool take;
        if (c == w.cq) take = same_on;
        else {
          take = false;
          if (other_on)
            for (int u = 0; u < w.ns; u++)  // cqIsBorrowing :657-667
              if (w.s_need[u] && S.nominal[ix(S, c, w.s_fr[u])] < s.usage[ix(S, c, w.s_fr[u])]) take = true;
        }
        int policy = c == w.cq ? KQ_POL_WITHIN_CQ(w.pol) : KQ_POL_RECLAIM(w.pol);
        if (take)
          for (int row = S.cq_adm_off[c]; row < S.cq_adm_off[c + 1]; row++) {
            if (row_removed(s, row)) continue;
            cbytes += 32 + 12 * (S.adm_use_off[row + 1] - S.adm_use_off[row]);
            if (!satisfies_policy(k, w, row, policy)) continue;
            if (!uses_need(k, w, row)) continue;
            s.cls[S.rank_pos[row]] = 1;
            cnt++;
            uint32_t key = f_row_key(S, row);
            if (key < head) head = key;
          }
        s.qcnt[i] = cnt; s.qhead[i] = head; s.cqinfo[i] = 0;
      }
      ncand += popc64(wballot(cnt > 0));
    }
    int64_t tot = wsum_i64((int64_t)cbytes);
    if (lane == 0) w.bytes += tot;
  }
  wsync();
  CSTAT(14, 1); CSTAT(15, ncand);
  KQ_TS(k, 42);  // fair search: findCandidates
  if (ncand == 0) return;
  f_apply_preemptor(s, true);  // SimulateUsageAddition :557
  int nt = 0;
  bool fits = false;
  const int strategy0 = k.C.n_fs > 0 ? k.C.fs[0] : KQ_FS_LESS_THAN_OR_EQUAL_TO_FINAL_SHARE;
  const bool have_second = k.C.n_fs > 0 ? k.C.n_fs > 1 : true;
  // ---- runFirstFsStrategy :384-470 ----
  {
    bool within_nominal = false;
    if (gate(k, KQ_GATE_FS_PREEMPT_WITHIN_NOMINAL)) {  // queueWithinNominalInResourcesNeedingPreemption :714-721
      within_nominal = true;
      for (int u = 0; u < w.ns; u++)
        if (w.s_need[u]) { UF uf{&S, s.W, w.s_fr[u], s.psum, s.ppos}; if (S.nominal[ix(S, w.cq, w.s_fr[u])] < uf.get(w.cq)) within_nominal = false; }
    }
    for (int cand = f_ordering_next(s); cand >= 0 && !fits; cand = fits ? -1 : f_ordering_next(s)) {
      if (cand == w.cq || within_nominal) {
        int row = f_pop(s, cand, 1, 0);
        f_apply_row(s, row, false, true, true);
        if (!f_push_target(s, &nt, row, cand == w.cq ? KQ_REASON_IN_CLUSTER_QUEUE : KQ_REASON_IN_COHORT_RECLAMATION)) { w.ntgt = 0; return; }
        if (f_fits_fs(s)) fits = true;
        continue;
      }
      int ap, at;
      f_almost_lcas(s, cand, &ap, &at);
      DRSv pn = f_drs_uniform(s, ap), to = f_drs_uniform(s, at);
      const int li = S.cq_local[cand];
      if (drs_pos_inf(pn) && !drs_pos_inf(to)) {  // fsStrategyUnsatisfiable :494-497
        while (s.qcnt[li] > 0) f_pop(s, cand, 1, 2);
        continue;
      }
      while (s.qcnt[li] > 0) {
        int row = f_pop(s, cand, 1, 0);
        // ComputeTargetShareAfterRemoval target.go:67-73
        f_apply_row(s, row, false, false, false);
        DRSv tn = f_drs_uniform(s, at);
        f_apply_row(s, row, true, false, false);
        bool pass = strategy0 == KQ_FS_LESS_THAN_OR_EQUAL_TO_FINAL_SHARE ? compare_drs(pn, tn) <= 0 : compare_drs(pn, to) < 0;  // strategy.go:41,46
        if (pass) {
          f_apply_row(s, row, false, true, true);
          if (!f_push_target(s, &nt, row, KQ_REASON_IN_COHORT_FAIR_SHARING)) { w.ntgt = 0; return; }
          if (f_fits_fs(s)) fits = true;
          break;
        }
        if (lane == 0) s.cls[S.rank_pos[row]] = 2;  // retryCandidates
        wsync();
      }
    }
  }
  KQ_TS(k, 43);  // fair search: first strategy
  // ---- runSecondFsStrategy :501-534 ----
  if (!fits && have_second) {
    for (int i = lane; i < nn; i += WAVE) s.cohp[i] = 0;
    for (int i = lane; i < nqs; i += WAVE) s.cqinfo[i] = 0;
    wsync();
    for (int i = lane; i < nqs; i += WAVE) {  // MakeClusterQueueOrdering(retryCandidates): one lane per CQ
      const int c = S.tree_cqs[q0 + i];
      uint32_t head = 0xffffffffu; int cnt = 0;
      if (s.qcnt[i] >= 0)
        for (int row = S.cq_adm_off[c]; row < S.cq_adm_off[c + 1]; row++)
          if (s.cls[S.rank_pos[row]] == 2) { cnt++; const uint32_t key = f_row_key(S, row); if (key < head) head = key; }
      s.qcnt[i] = cnt; s.qhead[i] = head;
    }
    wsync();
    for (int cand = f_ordering_next(s); cand >= 0 && !fits; cand = fits ? -1 : f_ordering_next(s)) {
      int ap, at;
      f_almost_lcas(s, cand, &ap, &at);
      DRSv pn = f_drs_uniform(s, ap), to = f_drs_uniform(s, at);
      bool passed = compare_drs(pn, to) < 0;
      int row = f_pop(s, cand, 2, 0);
      if (passed) {
        f_apply_row(s, row, false, true, true);
        if (!f_push_target(s, &nt, row, KQ_REASON_IN_COHORT_FAIR_SHARING)) { w.ntgt = 0; return; }
        if (f_fits_fs(s)) fits = true;
      }
      if (lane == 0) s.cqinfo[S.cq_local[cand]] = 1;  // DropQueue
      wsync();
    }
  }
  f_apply_preemptor(s, false);  // revertSimulation
  KQ_TS(k, 44);  // fair search: second strategy
  if (!fits) {
    if (lane == 0)
      for (int t = 0; t < nt; t++) { int r = s.trow[t]; w.bytes += 16 * (int64_t)S.plen[S.adm_cq[r]] * (S.adm_use_off[r + 1] - S.adm_use_off[r]); }
    // restoreSnapshot :356 — the private copy is dropped, but callers read it: put the rows back
    for (int t = 0; t < nt; t++) f_apply_row(s, s.trow[t], true, false, true);
    w.ntgt = 0;
    KQ_TS(k, 45);  // fair search: restore after a failed search
    return;
  }
  CSTAT(16, 1); CSTAT(17, nt);
  // fillBackWorkloads :341-354 with allowBorrowing = true
  for (int t = nt - 2; t >= 0; t--) {
    int r = s.trow[t];
    f_apply_row(s, r, true, true, true);
    if (f_fits(s)) {
      if (lane == 0) { s.trow[t] = s.trow[nt - 1]; s.treason[t] = s.treason[nt - 1]; }
      nt--;
      wsync();
    } else {
      f_apply_row(s, r, false, true, true);
    }
  }
  w.ntgt = nt;
  if (lane == 0)
    for (int t = 0; t < nt; t++) { int r = s.trow[t]; w.bytes += 16 * (int64_t)S.plen[S.adm_cq[r]] * (S.adm_use_off[r + 1] - S.adm_use_off[r]); }
  KQ_TS(k, 46);  // fair search: fillBackWorkloads
}

// LDS bytes the small state of one FAIR victim search wants: the per-node borrowed sums the DRS is computed from (psum, ppos: every
// pop of the TargetClusterQueueOrdering updates them with atomics and reads them back), the per-ClusterQueue candidate-queue heads and
// the pruning maps. The big private usage plane W and the per-row class bytes stay in HBM scratch: they are touched sparsely.
#ifdef KQ_HOST_EMU
static inline
#else
__host__ __device__ inline
#endif
size_t fair_lds_bytes(int nn, int nqs, int nR) {
  size_t b = (size_t)nn * nR * 8;              // psum
  b += ((size_t)nn * 4 + 7) & ~(size_t)7;       // ppos
  b += ((size_t)nqs * 4 + 7) & ~(size_t)7;      // qcnt
  b += ((size_t)nqs * 4 + 7) & ~(size_t)7;      // qhead
  b += ((size_t)nqs + 7) & ~(size_t)7;          // cqinfo
  b += ((size_t)nn + 7) & ~(size_t)7;           // cohp
  return b;
}
KQ_DEV Search make_search(const K& k, Wave& w, int slot, const int64_t* usage, const uint8_t* removed) {
  Search s;
  s.k = &k; s.w = &w; s.slot = slot; s.usage = usage; s.removed = removed;
  s.W = k.X.w + (size_t)slot * k.X.max_tree_nodes * k.X.slot_cap;
  s.cqinfo = k.X.cqinfo + (size_t)slot * k.X.max_tree_cqs;
  s.cls = k.X.cls + (size_t)slot * k.X.max_tree_rows;
  s.cand = k.X.cand + (size_t)slot * k.X.max_tree_rows;
  s.mark = k.X.mark + (size_t)slot * ((k.X.max_tree_rows + 63) / 64);
  s.trow = k.X.tgt_row + (size_t)slot * k.X.tgt_cap;
  s.treason = k.X.tgt_reason + (size_t)slot * k.X.tgt_cap;
  s.tree = 0; s.row0 = 0; s.nrows = 0;
  s.qcnt = k.X.qcnt ? k.X.qcnt + (size_t)slot * k.X.max_tree_cqs : nullptr;
  s.qhead = k.X.qhead ? k.X.qhead + (size_t)slot * k.X.max_tree_cqs : nullptr;
  s.cohp = k.X.cohp ? k.X.cohp + (size_t)slot * k.X.max_tree_nodes : nullptr;
  const bool plain = fs_plain_now(k);
  s.psum = (plain && k.X.psum) ? k.X.psum + (size_t)slot * k.X.max_tree_nodes * k.S.nR : nullptr;
  s.ppos = (plain && k.X.ppos) ? k.X.ppos + (size_t)slot * k.X.max_tree_nodes : nullptr;
  // fair sharing: the search's small, hot state in the workgroup's LDS when the launch provided enough of it (k_nominate's dynamic
  // LDS; in k_process_fair the flushed cohort rows lend theirs for the time of a recomputation)
  if (k.C.fair_sharing && w.cs_lds && k.X.qcnt &&
      (size_t)w.cs_lds_bytes >= fair_lds_bytes(k.X.max_tree_nodes, k.X.max_tree_cqs, k.S.nR)) {
    const int nn = k.X.max_tree_nodes, nqs = k.X.max_tree_cqs;
    unsigned char* p = w.cs_lds;
    auto carve = [&](size_t bytes) { unsigned char* q = p; p += (bytes + 7) & ~(size_t)7; return q; };
    int64_t* lp = (int64_t*)carve((size_t)nn * k.S.nR * 8);
    int32_t* lpp = (int32_t*)carve((size_t)nn * 4);
    if (s.psum) { s.psum = lp; s.ppos = lpp; }
    s.qcnt = (int32_t*)carve((size_t)nqs * 4);
    s.qhead = (uint32_t*)carve((size_t)nqs * 4);
    s.cqinfo = (uint8_t*)carve((size_t)nqs);
    s.cohp = (uint8_t*)carve((size_t)nn);
  }
  return s;
}

// PreemptionOracle.SimulatePreemption (preemption_oracle.go:43-85) for one flavor-resource
KQ_DEV void simulate_preemption(const K& k, Wave& w, int slot, const int64_t* usage, const uint8_t* removed,
                                int fr, int64_t val, int base_borrow, int* pm, int* borrow) {
  if (lane_id() == 0) { w.ns = 1; w.s_fr[0] = fr; w.s_qty[0] = val; w.s_inu[0] = 1; w.s_need[0] = 1; }
  wsync();
  Search s = make_search(k, w, slot, usage, removed);
  if (k.C.fair_sharing) fair_search(s); else classical_search(s);
  wsync();
  if (w.ntgt == 0) { *pm = PM_NOCAND; *borrow = base_borrow; return; }
  bool any_same = false;
  for (int t = 0; t < w.ntgt; t++) if (k.S.adm_cq[s.trow[t]] == w.cq) any_same = true;
  bool mr;
  if (k.C.fair_sharing) { UF uf{&k.S, s.W, fr, nullptr, nullptr}; *borrow = find_height(k.S, w.path, w.plen, fr, val, uf, &mr); }
  else { UW uw{&k.S, s.W, 1, 0}; *borrow = find_height(k.S, w.path, w.plen, fr, val, uw, &mr); }
  *pm = any_same ? PM_PREEMPT : PM_RECLAIM;
}

// ---- helper workgroups (K::help): a batch of SimulatePreemption calls run by whoever is idle ------------------------------------
KQ_DEV void load_head(const K& k, Wave& w, int h);
KQ_DEV uint64_t wuniform_u64(uint64_t v) { return ((uint64_t)(uint32_t)wuniform_i32((int)(v >> 32)) << 32) | (uint32_t)wuniform_i32((int)v); }
// one task of a batch on wave w, which holds the head of the batch (load_head): a pure function of (head, cell, planes)
KQ_DEV void help_run_task(const K& k, Wave& w, int slot, HelpBox* box, int idx) {
  const int64_t b0 = w.bytes;
  const HelpTask t = box->task[idx];
  int pm = 0, borrow = 0;
  simulate_preemption(k, w, slot, k.usage_np, k.preempted, t.fr, t.val, t.base_borrow, &pm, &borrow);
  wsync();
  if (lane_id() == 0) { box->res[idx].pm = pm; box->res[idx].borrow = borrow; box->res[idx].bytes = w.bytes - b0; w.bytes = b0; }
  wsync();
}
// take the next task of batch `seq`, -1 = none left (or the batch is over)
KQ_DEV int help_grab(HelpBox* box, uint32_t seq, int nt) {
  int idx = -1;
  if (lane_id() == 0) {
    uint64_t old = ag_load_u64(&box->next);
    while ((uint32_t)(old >> 32) == seq && (int)(old & 0xffffffffu) < nt) {
      if (ag_cas_u64(&box->next, old, old + 1)) { idx = (int)(old & 0xffffffffu); break; }
      old = ag_load_u64(&box->next);
    }
  }
  return wuniform_i32(idx);
}
// leader: post box->task[0..nt), work on it too, return when every result is in box->res
KQ_DEV void help_exec(const K& k, Wave& w, int slot, HelpBox* box, int nt) {
  const int lane = lane_id();
  uint32_t seq = 0;
  if (lane == 0) {
    seq = ++box->seq;
    box->entry = w.h;
    box->done = 0;
    ag_store_u64(&box->next, (uint64_t)seq << 32);
  }
  seq = (uint32_t)wuniform_i32((int)seq);
  wsync();
  ag_release();
  if (lane == 0) ag_store_u64(&box->hdr, ((uint64_t)seq << 32) | (uint32_t)nt);
  for (;;) {
    const int idx = help_grab(box, seq, nt);
    if (idx < 0) break;
#ifdef KQ_HOST_EMU
    if (idx & 1) {  // what a helper workgroup does: a fresh wave, its own scratch slot, the head reloaded
      Wave hw{};
      hw.cs_lds = nullptr; hw.cs_lds_bytes = 0;
      load_head(k, hw, box->entry);
      help_run_task(k, hw, k.help_trees, box, idx);
      box->done += 1;
      continue;
    }
#endif
    help_run_task(k, w, slot, box, idx);
    ag_release();
    if (lane == 0) ag_add_u32(&box->done, 1);
  }
  int spins = 0;
  for (;;) {
    uint32_t d = 0;
    if (lane == 0) d = ag_load_u32(&box->done);
    if ((uint32_t)wuniform_i32((int)d) >= (uint32_t)nt) break;
    ag_pause();
    if (++spins > (1 << 26)) { set_error(k, KQ_EDEVICE); break; }  // a helper died: fail the cycle instead of hanging the device
  }
  ag_acquire();
  if (lane == 0) ag_store_u64(&box->hdr, (uint64_t)seq << 32);  // closed
  wsync();
}
// helper workgroup: until every tree's leader has finished
KQ_DEV void helper_main(const K& k, Wave& w, int slot) {
  const int lane = lane_id();
  for (;;) {
    uint32_t q = 0;
    if (lane == 0) q = ag_load_u32(k.help_quit);
    if ((uint32_t)wuniform_i32((int)q) >= (uint32_t)k.help_trees) break;
    bool worked = false;
    for (int t = 0; t < k.help_trees && !worked; t++) {
      HelpBox* box = k.help + t;
      uint64_t hdr = 0;
      if (lane == 0) hdr = ag_load_u64(&box->hdr);
      hdr = wuniform_u64(hdr);
      const int nt = (int)(hdr & 0xffffffffu);
      if (nt == 0) continue;
      const int idx = help_grab(box, (uint32_t)(hdr >> 32), nt);
      if (idx < 0) continue;
      ag_acquire();
      int e = 0;
      if (lane == 0) e = box->entry;
      e = wuniform_i32(e);
      load_head(k, w, e);
      if (lane == 0) w.bytes = 0;
      wsync();
      help_run_task(k, w, slot, box, idx);
      ag_release();
      if (lane == 0) ag_add_u32(&box->done, 1);
      worked = true;
    }
    if (!worked) ag_pause();
  }
}

// ------------------------------------------------------------------------------------------------
// FlavorAssigner.assignFlavors (flavorassigner.go:708-908, TAS branches excluded)
// `counts` != NULL : partial admission probe (ScaledTo workload.go:317-340)
// `nominate_map`   : respect NominationMapping = current O.flavor (recompute on overlap)
// ------------------------------------------------------------------------------------------------
KQ_DEV int next_flavor_to_try(const K& k, const Wave& w, int ps_global, int res) {  // workload.go:226-238
  if (!gate(k, KQ_GATE_FLAVOR_FUNGIBILITY)) return 0;
  if (!w.has_last) return 0;
  int idx = k.H.ps_last_tried[(size_t)ps_global * k.S.nR + res];
  return idx < 0 ? 0 : idx + 1;
}
KQ_DEV bool can_preempt_while_borrowing(const K& k, const Wave& w) {  // flavorassigner.go:1386-1389
  return KQ_POL_BORROW_WITHIN(w.pol) != 0 || (k.C.fair_sharing && (KQ_POL_RECLAIM(w.pol) != KQ_POLICY_NEVER || KQ_POL_RECLAIM_UNSET(w.pol)));
}

// Status.appendf (flavorassigner.go:349): lane 0 appends one record to the head's window
KQ_DEV void rsn_push(const K& k, Wave& w, int code, int podset, int flavor, int resource, int64_t a, int64_t b, int64_t c) {
  if (k.O.rsn_win <= 0) return;
  if (w.nrsn >= k.O.rsn_win) { w.rsn_over = 1; return; }
  RsnRec r; r.code = (uint8_t)code; r.podset = (uint8_t)podset; r.flavor = (int16_t)flavor; r.resource = (int16_t)resource; r.pad = 0; r.a = a; r.b = b; r.c = c;
  k.O.rsn[(size_t)w.h * k.O.rsn_win + w.nrsn] = r;
  w.nrsn++;
}

// LEAN (k_nominate's first pass): no victim search is linked in. A cell that needs SimulatePreemption is answered in place when the
// ClusterQueue cannot preempt at all (both searches return "no candidates" before reading anything, preemption.go:284-296 / :536-547);
// otherwise the head is handed to the full pass (w.defer_head) and this call's outputs are dropped.
// workload slices: the flavor / request the replaced slice holds for request slot `a` of podset psg (replaceWorkloadSlice.TotalRequests
// [psID].Flavors / .Requests, flavorassigner.go:1127-1143)
KQ_DEV void slice_of(const K& k, const Wave& w, int psg, int a, int* flavor, int64_t* qty) {
  const DHeads& H = k.H;
  const int src = w.req_src[a];
  if (src == 0xff) { *flavor = H.ps_slice_pods_flavor ? H.ps_slice_pods_flavor[psg] : -1; *qty = H.ps_slice_pods_qty ? H.ps_slice_pods_qty[psg] : 0; return; }
  const int e = H.ps_req_off[psg] + src;
  *flavor = H.req_slice_flavor ? H.req_slice_flavor[e] : -1;
  *qty = H.req_slice_qty ? H.req_slice_qty[e] : 0;
}

// ---- PodSetGroupName groups (flavorassigner.go:782-860) ---------------------------------------------------------------------------
// The podsets of one group are ONE flavor scan over the sum of their requests (requests.Add over the members, :786-790), eligible where
// EVERY member is (checkFlavorForPodSets walks psIDs, :1234), resumed from the FIRST member's bookmark (:1092), pinned to a flavor ANY
// member was nominated on (:1422); each member then keeps the group's flavors for the resources it requests itself — a member that
// requests nothing keeps the group's TAS flavors — (resolvePodSetFlavors :917-945) and the group's Status. Members are consecutive
// podsets of their head (the host checks). A podset outside any group is a group of one and takes the code it always took.
KQ_DEV int group_len(const K& k, const Wave& w, int pi) {
  const int32_t* grp = k.H.ps_group;
  if (!grp) return 1;
  const int gid = grp[w.ps_base + pi];
  if (gid < 0) return 1;
  int gn = 1;
  while (pi + gn < w.nps && grp[w.ps_base + pi + gn] == gid) gn++;
  return gn;
}
// the effective requests of podset pi, one after the other (ScaledTo workload.go:317-340; the injected `pods`, flavorassigner.go:743-749):
// fn(resource, quantity); returns how many there are (Requests.Len())
template <class F> KQ_DEV int member_requests(const K& k, const Wave& w, int pi, const int* counts, bool pods_cov, F&& fn) {
  const DHeads& H = k.H;
  const int psg = w.ps_base + pi;
  const int count = H.ps_count[psg], new_count = counts ? counts[pi] : count;
  const bool scale = counts && count != 0 && count != new_count;
  bool have_pods = false;
  int n = 0;
  for (int e = H.ps_req_off[psg]; e < H.ps_req_off[psg + 1]; e++) {
    const int r = H.req_res[e];
    int64_t q = H.req_qty[e];
    if (scale) q = sat_mul(q / (int64_t)count, (int64_t)new_count);
    if (pods_cov && r == k.S.pods_res) { q = scale ? new_count : count; have_pods = true; }
    fn(r, q); n++;
  }
  if (pods_cov && !have_pods) { fn(k.S.pods_res, (int64_t)(scale ? new_count : count)); n++; }
  return n;
}
// w.req_* = the sum of the members' requests, in arrival order (lane 0; the caller sorts them into Requests.Iter order)
KQ_NOINLINE void group_requests(const K& k, Wave& w, int pi, int gn, const int* counts, bool pods_cov) {
  if (lane_id() == 0) {
    int n = 0;
    for (int m = 0; m < gn; m++)
      member_requests(k, w, pi + m, counts, pods_cov, [&](int r, int64_t q) {
        for (int b = 0; b < n; b++) if (w.req_res[b] == r) { w.req_qty[b] = sat_add(w.req_qty[b], q); return; }   // SliceRequests.Add: SaturatingAdd
        if (n >= KQ_MAXREQ) { *k.O.error = KQ_EUNSUPPORTED; return; }
        w.req_res[n] = r; w.req_qty[n] = q; w.req_src[n] = 0; n++;
      });
    w.nreq = n;
    if (w.slice_row >= 0) *k.O.error = KQ_EUNSUPPORTED;   // (a workload slice with a group of several podsets: the host refuses the batch)
  }
  wsync();
}
// one flavor of the assignment goes to podset psg / resource res: the output row, Assignment.append's usage entry (:1017-1041)
KQ_DEV void group_take(const K& k, Wave& w, int psg, int a, int64_t amount) {
  const DOut& O = k.O;
  const int nR = k.S.nR, res = w.req_res[a];
  const size_t o = (size_t)psg * nR + res;
  O.flavor[o] = w.req_flavor[a]; O.res_mode[o] = w.req_mode[a]; O.tried_idx[o] = w.req_tried[a];
  if (w.req_borrow[a] > w.borrowing) w.borrowing = w.req_borrow[a];
  const int fr = w.req_flavor[a] * nR + res;
  int e = -1;
  for (int i = 0; i < w.nuse; i++) if (w.use_fr[i] == fr) e = i;
  if (e < 0) {
    if (w.nuse >= KQ_MAXU) { *O.error = KQ_EUNSUPPORTED; return; }
    e = w.nuse++; w.use_fr[e] = fr; w.use_qty[e] = 0; w.use_mode[e] = M_FIT;
  }
  w.use_qty[e] = a_addi(w.use_qty[e], amount);
  if (w.req_mode[a] < w.use_mode[e]) w.use_mode[e] = w.req_mode[a];
  w.bytes += 16;
}
// the members of a group behind its scan: flavors, usage, Status and RepresentativeMode of each (flavorassigner.go:836-847). Returns the
// weakest member's mode; *failed = atLeastOnePodsAssignmentFailed.
KQ_NOINLINE int group_finish(const K& k, Wave& w, int pi, int gn, const int* counts, bool pods_cov, bool group_failed, int reasons, bool* failed) {
  const int lane = lane_id();
  int rep = M_FIT;
  bool bad = false;
  if (lane == 0 && k.O.rsn_win > 0) {   // podSetAssignment.Status = groupStatus for every member: the first member's records once more for each of the others
    RsnRec* win = k.O.rsn + (size_t)w.h * k.O.rsn_win;
    const int cnt = w.nrsn - w.rsn_ps0;
    for (int m = 1; m < gn; m++)
      for (int q = 0; q < cnt; q++) {
        if (w.nrsn >= k.O.rsn_win) { w.rsn_over = 1; break; }
        RsnRec r = win[w.rsn_ps0 + q]; r.podset = (uint8_t)(pi + m);
        win[w.nrsn++] = r;
      }
  }
  for (int m = 0; m < gn; m++) {
    const int psg = w.ps_base + pi + m;
    int nfl = 0, mode = M_FIT;
    // (every lane walks the member's few requests; lane 0 writes)
    const int mreq = member_requests(k, w, pi + m, counts, pods_cov, [&](int r, int64_t q) {
      if (group_failed) return;
      int a = -1;
      for (int b = 0; b < w.nreq; b++) if (w.req_res[b] == r) a = b;
      if (a < 0 || !w.req_done[a]) return;
      nfl++; if (w.req_mode[a] < mode) mode = w.req_mode[a];
      if (lane == 0) group_take(k, w, psg, a, q);
    });
#ifdef KQ_TAS_CYCLE
    if (mreq == 0 && !group_failed && k.tc)   // tasFlavorsOnly :996: "a PodSet with no resource requests in a topology group (e.g. an LWS leader) still needs a resolved TAS flavor"
      for (int a = 0; a < w.nreq; a++) {
        if (!w.req_done[a] || !tc_is_tas_flavor(k, w.req_flavor[a])) continue;
        nfl++; if (w.req_mode[a] < mode) mode = w.req_mode[a];
        if (lane == 0) group_take(k, w, psg, a, 0);
      }
#endif
    if (lane == 0) w.bytes += (int64_t)mreq * 8;
    const int pmode = reasons == 0 ? M_FIT : (nfl == 0 ? M_NOFIT : mode);   // PodSetAssignment.RepresentativeMode :386-404
    if (pmode < rep) rep = pmode;
    bad |= mreq > 0 && nfl == 0;
  }
  wsync();
  *failed = bad;
  return rep;
}

// ---- simulations ahead (K::sim_*) ---------------------------------------------------------------------------------------------------
// Lean / emit pass, a head about to be deferred because cell (jj, kk) of the pass needs a real SimulatePreemption: list every cell of
// the pass the scan will simulate — per flavor the cells in front of the first one that is NoFit before any simulation (behind it the
// closure returns early, :1161); flavors skipped by eligibility or the nomination pin have none. Under WhenCanPreempt = TryNextFlavor
// the scan visits all of them (no flavor of this pass is Fit, or the pass would have no live cell); under MayStopSearch it may stop
// earlier and some results go unused. `key` names the scan (sim_find): the passes that follow reach it in the same state.
KQ_DEV int sim_find(const K& k, int h, int key) {
  int n = k.sim_nscan[h];
  if (n > SIM_KS) n = SIM_KS;
  for (int q = 0; q < n; q++) if (k.sim_key[(size_t)h * SIM_KS + q] == key) return q;
  return -1;
}
KQ_NOINLINE void sim_emit(const K& k, Wave& w, int key, int f0, int cs, int nfl, int nf) {
  if (lane_id() == 0 && k.sim_nscan[w.h] < SIM_KS) {
    const int q = k.sim_nscan[w.h];
    int32_t* cell = k.sim_cell + ((size_t)w.h * SIM_KS + q) * CELLS;
    int n = 0;
    for (int pass = 0; pass < 2; pass++) {   // count, then write
      int base = 0;
      if (pass == 1) {
        if (n == 0) break;
        base = atomic_add_i32(&k.sim_ctl[0], n);
        if (base + n > k.sim_cap) break;     // (no room: the full pass searches in place)
      }
      int t = 0;
      for (int jj = 0; jj < nfl; jj++) {
        bool live = w.cell_pm[jj * nf] != PM_SKIP;
        for (int kk = 0; kk < nf; kk++) {
          const int c = jj * nf + kk;
          const uint8_t full = w.cell_pm[c];
          const bool task = live && !(full & 0x40) && (full & 0x3f) == PM_NEEDS;
          if (pass == 1) {
            cell[c] = task ? base + t : -1;
            if (task) k.sim_task[base + t] = SimTask{w.h, k.S.rg_flavor[f0 + cs + jj] * k.S.nR + w.f_res[kk], w.cell_borrow[c], 0, w.cell_val[c]};
          }
          if (task) t++;
          if ((full & 0x40) || (full & 0x3f) == PM_NOFIT) live = false;   // representativeMode is noFit from here on: no further oracle call for this flavor
        }
      }
      n = t;
      if (pass == 1) { k.sim_key[(size_t)w.h * SIM_KS + q] = key; k.sim_nscan[w.h] = q + 1; if (q > 0) CSTAT(8, 1); }
    }
  }
  wsync();
}

template <bool LEAN>
KQ_DEV void assign_flavors(const K& k, Wave& w, int slot, const int64_t* usage, const uint8_t* removed,
                           const int* counts, bool nominate_map) {
  const DSnap& S = k.S; const DHeads& H = k.H; const DOut& O = k.O;
  const int lane = lane_id();
  const int nR = S.nR;
  KQ_T0();
#ifdef KQ_TAS_CYCLE
  if (k.tc) tc_reset(w);
#endif
  if (lane == 0) { w.nuse = 0; w.borrowing = 0; w.rep_mode = M_FIT; w.nrsn = 0; w.rsn_over = 0; }
  wsync_lds();
  int rep = M_FIT;
  bool any_ps = false;
  const bool pods_cov = S.pods_res >= 0 && rg_by_resource(S, w.cq, S.pods_res) >= 0;
  int gn = 1;
  for (int pi = 0; pi < w.nps; pi += gn) {
    const int psg = w.ps_base + pi;
    any_ps = true;
    gn = group_len(k, w, pi);   // the podsets [pi, pi + gn) share one flavor scan (PodSetGroupName); 1 = a podset on its own
    if constexpr (LEAN) {
      // a head with a multi-podset group goes to the full pass: the lean kernel keeps no function call (group_requests / group_finish are
      // not inlined; with them the kernel carried a call frame — 52 B of scratch, 52 more SGPR spills — and every head paid for it:
      // k_nominate_lean 110 -> 123 us at cfg 3, profiles/r06p_cfg3_timeline.txt). The host launches the full pass for batches with groups.
      if (gn > 1) { if (lane == 0) w.defer_head = 1; wsync(); return; }
    }
    // ---- effective requests -------------------------------------------------------------
    int count = H.ps_count[psg];
    int new_count = counts ? counts[pi] : count;
    bool scale = counts && count != 0 && count != new_count;
    // one lane per request: the loads of a podset's requests and of their resource groups are one round trip each (they were a
    // dependent chain on lane 0: ~8 of the ~25 round trips a head cost)
    const int e0 = H.ps_req_off[psg], ne_raw = H.ps_req_off[psg + 1] - e0;
    const int ne = ne_raw < KQ_MAXREQ ? ne_raw : KQ_MAXREQ;
    if (ne_raw > KQ_MAXREQ && lane == 0) *O.error = KQ_EUNSUPPORTED;
    bool is_pods = false;
    bool grouped = false;
    if constexpr (!LEAN) { if (gn > 1) { group_requests(k, w, pi, gn, counts, pods_cov); grouped = true; } }   // the sum of the members' requests (requests.Add :789)
    if (!grouped) {
    for (int a = lane; a < ne; a += WAVE) {
      int64_t q = H.req_qty[e0 + a];
      if (scale) q = sat_mul(q / (int64_t)count, (int64_t)new_count);
      const int r = H.req_res[e0 + a];
      if (pods_cov && r == S.pods_res) { q = scale ? new_count : count; is_pods = true; }
      w.req_res[a] = r; w.req_qty[a] = q; w.req_src[a] = (uint8_t)a;
    }
    const bool have_pods = wballot(is_pods) != 0;
    wsync_lds();
    if (lane == 0) {
      int n = ne;
      if (pods_cov && !have_pods) {
        if (n >= KQ_MAXREQ) *O.error = KQ_EUNSUPPORTED;
        else { w.req_res[n] = S.pods_res; w.req_qty[n] = scale ? new_count : count; w.req_src[n] = 0xff; n++; }
      }
      w.nreq = n;
    }
    wsync_lds();
    }
    {  // Requests.Iter order (slice_requests.go:54-60)
      const int n = w.nreq;
#ifdef KQ_HOST_EMU
      for (int a = 1; a < n; a++) {  // 1-lane emulation: insertion sort by resource_order
        int r = w.req_res[a]; int64_t q = w.req_qty[a]; uint8_t sr = w.req_src[a]; int b = a - 1;
        while (b >= 0 && S.resource_order[w.req_res[b]] > S.resource_order[r]) { w.req_res[b + 1] = w.req_res[b]; w.req_qty[b + 1] = w.req_qty[b]; w.req_src[b + 1] = w.req_src[b]; b--; }
        w.req_res[b + 1] = r; w.req_qty[b + 1] = q; w.req_src[b + 1] = sr;
      }
      for (int a = 0; a < n; a++) { w.req_done[a] = 0; w.req_rg[a] = S.cq_res_rg[(size_t)w.cq * nR + w.req_res[a]]; }
#else
      // device: one lane per request (n <= KQ_MAXREQ < 64): rank by pairwise compare, then a scatter — no serial loop, one round trip
      int r = 0, ord = 0, rank = 0, sr = 0; int64_t q = 0;
      if (lane < n) { r = w.req_res[lane]; q = w.req_qty[lane]; sr = w.req_src[lane]; ord = S.resource_order[r]; }
      for (int b = 0; b < n; b++) {
        const int ob = wshfl_i32(ord, b);
        if (lane < n && (ob < ord || (ob == ord && b < lane))) rank++;
      }
      wsync_lds();
      if (lane < n) { w.req_res[rank] = r; w.req_qty[rank] = q; w.req_src[rank] = (uint8_t)sr; w.req_done[rank] = 0; w.req_rg[rank] = S.cq_res_rg[(size_t)w.cq * nR + r]; }
#endif
      if (lane == 0) w.bytes += (int64_t)n * (gn > 1 ? 8 : 16);  // requests in + requests echoed in the PodSetAssignment (a group: its members echo theirs, group_finish)
    }
    wsync_lds();
    // (the nomination mapping of a recomputation, X.nom, was taken before O.flavor is overwritten: process_entry)
    // clear the podset's output rows
    for (int r = lane; r < gn * nR; r += WAVE) { O.flavor[(size_t)psg * nR + r] = -1; O.res_mode[(size_t)psg * nR + r] = M_NOFIT; O.tried_idx[(size_t)psg * nR + r] = -1; }
    if (lane == 0) {
      O.ps_count[psg] = scale ? new_count : count;
      for (int m = 1; m < gn; m++) { const int c0 = H.ps_count[psg + m], c1 = counts ? counts[pi + m] : c0; O.ps_count[psg + m] = (counts && c0 != 0) ? c1 : c0; }
    }
    wsync();
    if (LEAN || KQ_TAS_PROCESS(k, w)) KQ_TS(k, 57);  // lean: requests of the podset in iterator order, output rows cleared
    bool group_failed = false;
    int ps_reasons = 0, ps_nflavors = 0, ps_mode = M_FIT;
#ifdef KQ_TAS_CYCLE
    if constexpr (!LEAN) if (k.tc) {
      // "Respect preexisting assignments. The PodSet assignments may be already set if this is the second pass of scheduler"
      // (flavorassigner.go:765-779): the admission's flavor, mode Fit, TriedFlavorIdx 0, and no flavor scan for the resource (:819)
      for (int m = 0; m < gn; m++) {   // (a group: "seed the dedup memo with every prior-pass assignment, not just one" :801-804 — maps.Copy in member order)
        tc_sp_reset(k, psg + m);
        for (int a = 0; a < w.nreq; a++) {
          const int fl = tc_adm_flavor(k, psg + m, w.req_res[a]);
          if (fl < 0) continue;
          if (lane == 0) { w.req_done[a] = 1; w.req_flavor[a] = fl; w.req_mode[a] = M_FIT; w.req_borrow[a] = 0; w.req_tried[a] = 0; }
          wsync();
        }
      }
      wsync();
    }
#endif
    if (lane == 0) w.rsn_ps0 = w.nrsn;
    for (int a = 0; a < w.nreq && !group_failed; a++) {
      const int res_name = w.req_res[a];
      const int gl = w.req_rg[a];
      const int g = gl < 0 ? -1 : S.cq_rg_off[w.cq] + gl;
      if (g < 0) {  // flavorassigner.go:809-817
        if (w.req_qty[a] == 0) continue;
        if (gate(k, KQ_GATE_QUOTA_CHECK_STRATEGY) && k.C.quota_check_strategy == KQ_QUOTA_CHECK_IGNORE_UNDECLARED) continue;
        group_failed = true; ps_reasons = 1;  // "resource unavailable in ClusterQueue" :1080
        if (lane == 0) { w.nrsn = w.rsn_ps0; rsn_push(k, w, KQ_RSN_RESOURCE_UNAVAILABLE, pi, -1, res_name, 0, 0, 0); }  // the podset's whole status (:826-829)
        break;
      }
      if (w.req_done[a]) continue;  // :819
      // ---- findFlavorForPodSets :1065-1210 ---------------------------------------------
      if (lane == 0) {
        int nf = 0;
        // filterRequestedResources :1391: the requests group g covers. With overlapping groups (two groups of the ClusterQueue covering one
        // resource — the webhook rejects the spec, the cache keeps it; flavorassigner_test.go:505) that is more than the requests whose
        // RGByResource is g: the scan takes them all and its flavors overwrite what an earlier scan gave them (maps.Copy :831)
        const bool overlap = (w.pol & KQ_POL_DEV_RG_OVERLAP) != 0;
        for (int b = 0; b < w.nreq; b++) {
          bool cov = w.req_rg[b] == gl;
          if (!cov && overlap) for (int i = S.rg_res_off[g]; i < S.rg_res_off[g + 1]; i++) cov = cov || S.rg_res[i] == w.req_res[b];
          if (cov) { w.f_res[nf] = w.req_res[b]; w.f_qty[nf] = w.req_qty[b]; w.f_slot[nf] = (uint8_t)b; nf++; }
        }
        w.nf = nf;
        w.rsn_g0 = w.nrsn;
      }
      wsync_lds();
      const int nf = w.nf;
      const int f0 = S.rg_flavor_off[g], nflv = S.rg_flavor_off[g + 1] - f0;
      const int fpp = CELLS / nf;  // flavors per pass (nf <= KQ_MAXREQ <= CELLS)
      int best = -1; int64_t best_key = -1;   // worstGranularMode
      int best_pm = PM_NOFIT;
      int attempted = -1;
      int reasons = 0;
      bool stop = false;
      int idx0 = next_flavor_to_try(k, w, psg, res_name);
      const int plen = w.plen;
      for (int cs = idx0; cs < nflv && !stop; cs += fpp) {
        const int nfl = (nflv - cs) < fpp ? (nflv - cs) : fpp;
        // ---- all (flavor, resource) cells of this pass at once: fitsResourceQuota :1334-1384
        for (int c = lane; c < nfl * nf; c += WAVE) {
          int j = cs + c / nf, kk = c % nf;
          int f = S.rg_flavor[f0 + j];
          bool ok = (H.ps_flavor_ok[(size_t)psg * S.nfw + (f >> 6)] >> (f & 63)) & 1;  // checkFlavorForPodSets :1212 (host-evaluated)
          for (int m = 1; m < gn; m++) ok = ok && ((H.ps_flavor_ok[(size_t)(psg + m) * S.nfw + (f >> 6)] >> (f & 63)) & 1);   // ... for every podset of the group (:1234)
          uint8_t pm = PM_SKIP; int32_t borrow = ok ? 0 : KQ_RSN_FLAVOR_INELIGIBLE; int64_t val = 0, aux = 0;  // a skipped cell keeps WHY in `borrow`
          if (nominate_map) {  // shouldSkipBasedOnNominationMapping :1422 (checked first, :1096): kept when ANY podset of the group was nominated on the flavor
            bool keep = false;
            for (int m = 0; m < gn; m++) keep = keep || k.X.nom[((size_t)slot * KQ_MAXPS + pi + m) * nR + res_name] == f;
            if (!keep) { ok = false; borrow = KQ_RSN_NOT_IN_NOMINATION; }
          }
          bool mismatch = false;
          if (ok) {
            int fr = f * nR + w.f_res[kk];
            int64_t reqv = w.f_qty[kk];
            if (w.slice_row >= 0) {  // the flavor of the slice it replaces, and only the delta (flavorassigner.go:1125-1145)
              int sfl; int64_t sq;
              slice_of(k, w, psg, w.f_slot[kk], &sfl, &sq);
              if (sfl != f) mismatch = true; else reqv -= sq;   // (a mismatch leaves val unreduced: the `break` comes before the subtraction)
            }
            val = a_addi(assumed_usage(w, fr), reqv);
            int64_t avail, maxcap, nominal;
            bool may_reclaim = false;
            int height = 0;
            if (plen <= GP_MAX) {
              GPath g;
              gpath_load(S, w.path, plen, fr, usage, g);
              avail = i64max(0, gpath_available(g, plen));
              maxcap = gpath_potential(g, plen);
              nominal = g.nominal0;
              if (!(val > maxcap)) height = gpath_height(g, plen, val, &may_reclaim);
            } else {
              UG ug{&S, usage, fr};
              avail = i64max(0, available_of(S, w.path, plen, fr, ug));
              maxcap = potential_of(S, w.path, plen, fr);
              nominal = S.nominal[ix(S, w.cq, fr)];
              if (!(val > maxcap)) height = find_height(S, w.path, plen, fr, val, ug, &may_reclaim);
            }
            if (val > maxcap) { pm = PM_NOFIT; borrow = 0; aux = maxcap; }
            else {
              borrow = height;
              aux = a_sub(val, avail);  // "%s more needed" :1372
              if (val <= avail) pm = PM_FIT;
              else if (nominal >= val || may_reclaim || can_preempt_while_borrowing(k, w)) pm = PM_NEEDS;
              else pm = PM_NOFIT | 0x80;  // noFit with a "insufficient unused quota" reason and borrow kept
            }
          }
          w.cell_pm[c] = pm | (mismatch ? 0x40 : 0); w.cell_borrow[c] = borrow; w.cell_val[c] = val; w.cell_aux[c] = aux;
        }
        wsync_lds();
        if (LEAN || KQ_TAS_PROCESS(k, w)) KQ_TS(k, 58);  // lean: the (flavor, resource) cells of the pass (fitsResourceQuota)
        // ---- dead simulations -------------------------------------------------------------------------------------------------
        // A flavor whose cells are all Fit beats every flavor with a cell that needs SimulatePreemption (isPreferred compares the
        // preemption mode first, :553-558), the scan's Status is nil once the best flavor is Fit (:1199-1207), and a flavor that needs a
        // simulation cannot END the scan under WhenCanPreempt = TryNextFlavor (:1271; NoCandidates never does, :1267) — nor without
        // FlavorFungibility, where only a Fit flavor ends it (:1184-1190). So once a Fit flavor is in hand, or stands anywhere in this
        // pass, what the simulations of the pass would return cannot be observed — not in the assignment, not in the bookmark
        // (attemptedFlavorIdx is where the scan ends: a Fit flavor that does not borrow, or borrows under WhenCanBorrow = MayStopSearch,
        // or the last flavor), not in the reasons: they are not run. In a saturated tree that is every simulation of every head that still
        // finds a flavor with room: at cfg 4 the whole k_nominate pass of a cycle without preemptions, and such heads finish in the lean
        // pass. Not under Preference = PreemptionOverBorrowing (modes are ordered by borrowing level first there: a flavor with a simulated
        // cell can come out as "Fit, borrowing" and win or end the scan) and not under WhenCanPreempt = MayStopSearch (a flavor that can
        // preempt ends the scan). The oracle runs them all, as the reference does, and books their bytes as discarded by the same rule
        // (kq_oracle.cpp findFlavorForPodSets).
        bool dead_all = false;
        if ((!gate(k, KQ_GATE_FLAVOR_FUNGIBILITY) || KQ_POL_PREEMPT_TRYNEXT(w.pol)) && KQ_POL_PREFERENCE(w.pol) != KQ_PREF_PREEMPTION_OVER_BORROWING) {
          dead_all = best_pm == PM_FIT;
          for (int jj = 0; jj < nfl && !dead_all; jj++) {
            if (w.cell_pm[jj * nf] == PM_SKIP) continue;
            bool all_fit = true;
            for (int kk = 0; kk < nf; kk++) all_fit = all_fit && w.cell_pm[jj * nf + kk] == PM_FIT;
            dead_all = all_fit;
          }
        }
        // the simulations of this pass may have been listed by an earlier pass over the head and run by k_nominate_sim (sim_emit)
        const bool sim_elig = k.sim_nscan != nullptr && !counts && !nominate_map && w.slice_row < 0 && gn == 1 && pi < 256 && res_name < 256 && cs < 32768;
        const int sim_key = pi | (res_name << 8) | (cs << 16);
        const int32_t* sim_cells = nullptr;
        int sim_q = -1;
        if (sim_elig) { sim_q = sim_find(k, w.h, sim_key); if (sim_q >= 0) sim_cells = k.sim_cell + ((size_t)w.h * SIM_KS + sim_q) * CELLS; }
        // ---- recomputation inside k_process_fair: every cell of the pass that needs a SimulatePreemption is posted as one batch ----
        bool batched = false;
        HelpBox* hbox = nullptr;
        if constexpr (!LEAN) {
          if (w.help_on && nominate_map && k.help) {
            hbox = k.help + w.help_tree;
            if (lane == 0) {
              int nt = 0;
              for (int jj = 0; jj < nfl; jj++)
                for (int kk = 0; kk < nf; kk++) {
                  const int c = jj * nf + kk;
                  w.cell_task[c] = 0xff;
                  if (w.cell_pm[jj * nf] == PM_SKIP) continue;
                  if ((w.cell_pm[c] & 0x3f) == PM_NEEDS && !dead_all) {
                    hbox->task[nt] = HelpTask{S.rg_flavor[f0 + cs + jj] * nR + w.f_res[kk], w.cell_borrow[c], w.cell_val[c]};
                    w.cell_task[c] = (uint8_t)nt++;
                  }
                }
              w.help_nt = nt;
            }
            wsync();
            if (w.help_nt > 1) { CSTAT(25, 1); CSTAT(26, w.help_nt); help_exec(k, w, slot, hbox, w.help_nt); batched = true; }
          }
        }
        // ---- the scan, flavor-parallel ----------------------------------------------------------------------------------------------
        // When no reason records are kept, the head replaces no slice and no cell of the pass waits for a victim search (none needs one,
        // or every simulation of the pass is dead, or the ClusterQueue cannot preempt at all), a flavor's representative mode is a function
        // of its own cells: one lane per flavor folds them (the loop over kk below, cell by cell in order), and the ordered part of the
        // scan — early exit, isPreferred against the best so far, the bookmark — walks one packed word per flavor instead of every cell.
        // At cfg 3 a head that finds no room walks 16 flavors x 4 resources: 64 cells one after the other were 40 % of k_nominate_lean
        // (tools/prof_lean.py). Written for any lane count: the 1-lane emulation runs the same code.
        bool fast = false;
        if (w.slice_row < 0 && !batched) {
          bool needs_ok = dead_all;
          if constexpr (LEAN) needs_ok = needs_ok || !(KQ_POL_WITHIN_CQ(w.pol) != KQ_POLICY_NEVER || (w.plen > 1 && KQ_POL_RECLAIM(w.pol) != KQ_POLICY_NEVER));
          bool mine = false;
          if (!needs_ok) for (int c = lane; c < nfl * nf; c += WAVE) mine = mine || ((w.cell_pm[c] & 0x3f) == PM_NEEDS && w.cell_pm[(c / nf) * nf] != PM_SKIP);
          fast = needs_ok || wballot(mine) == 0;
        }
        if (fast) {
          for (int jj = lane; jj < nfl; jj += WAVE) {
            int64_t pack = PM_SKIP, key = 0;
            if (w.cell_pm[jj * nf] != PM_SKIP) {
              int rep_pm = PM_FIT; int rep_borrow = 0, rs = 0; int64_t rep_key = pref_key(PM_FIT, 0, w.pol);
              for (int kk = 0; kk < nf; kk++) {
                const int c = jj * nf + kk;
                int pm = w.cell_pm[c] & 0x3f; const int borrow = w.cell_borrow[c];
                if ((w.cell_pm[c] & 0x80) || pm == PM_NOFIT || pm == PM_NEEDS) rs++;   // had_status: a reason either way
                if (rep_pm == PM_NOFIT) continue;            // oracle result unused past a noFit (:1161)
                if (pm == PM_NEEDS) pm = PM_NOCAND;          // a dead simulation / SimulatePreemption with an empty candidate set: borrow kept
                const int64_t key2 = pref_key(pm, borrow, w.pol);
                if (rep_key > key2) { rep_pm = pm; rep_borrow = borrow; rep_key = key2; }
              }
              pack = (int64_t)(((uint64_t)(uint32_t)rep_borrow << 32) | ((uint64_t)rs << 8) | (uint64_t)rep_pm); key = rep_key;
            }
            w.flv_key[jj] = key; w.flv_pack[jj] = pack;
          }
          wsync_lds();
          int best_jj = -1, visited = 0;
          const int nrsn0 = w.nrsn;   // (uniform: lane 0 wrote it in front of a barrier)
          int nrec = 0, jj_end = 0;   // reason records of the flavors the scan reaches, and how far it gets
          for (int jj = 0; jj < nfl && !stop; jj++) {
            const int j = cs + jj;
            attempted = j;
            jj_end = jj + 1;
            const int64_t pack = w.flv_pack[jj];
            const int rep_pm = (int)(pack & 0xff);
            if (rep_pm == PM_SKIP) { reasons++; nrec++; continue; }
            visited++;
            reasons += (int)((pack >> 8) & 0xff);
            nrec += (int)((pack >> 8) & 0xff);
            const int64_t rep_borrow = (int64_t)(int32_t)(uint32_t)((uint64_t)pack >> 32), rep_key = w.flv_key[jj];
            bool take = false;
            if (gate(k, KQ_GATE_FLAVOR_FUNGIBILITY)) {
              if (!should_try_next(rep_pm, rep_borrow, w.pol)) { take = true; stop = true; }
              else if (rep_key > best_key) take = true;
            } else if (rep_pm > best_pm) {
              take = true;
              if (rep_pm == PM_FIT) stop = true;
            }
            if (take) { best = j; best_key = rep_key; best_pm = rep_pm; best_jj = jj; }
          }
          if (best_jj >= 0)
            for (int kk = lane; kk < nf; kk += WAVE) {
              const int c = best_jj * nf + kk;
              int pm = w.cell_pm[c] & 0x3f;
              if (pm == PM_NEEDS) pm = PM_NOCAND;
              w.best_mode[kk] = fa_mode(pm); w.best_borrow[kk] = w.cell_borrow[c];
            }
          if (lane == 0) w.bytes += (int64_t)visited * nf * 40 * plen;  // nf fitsResourceQuota calls per flavor the scan reached, (D+1) x 5 planes x 8 B each
          if (k.O.rsn_win > 0 && nrec > 0) {
            // the reason records of the flavors the scan reached (Status.appendf :349), in scan order: a skipped flavor has one, the others
            // one per cell with a status (:1353-1373). One lane per flavor: its offset is the count of the flavors in front of it.
            RsnRec* win = k.O.rsn + (size_t)w.h * k.O.rsn_win;
            for (int jj = lane; jj < jj_end; jj += WAVE) {
              int pos = nrsn0;
              for (int q = 0; q < jj; q++) { const int64_t pq = w.flv_pack[q]; pos += (int)(pq & 0xff) == PM_SKIP ? 1 : (int)((pq >> 8) & 0xff); }
              const int f = S.rg_flavor[f0 + cs + jj];
              if (w.cell_pm[jj * nf] == PM_SKIP) {
                const int why = w.cell_borrow[jj * nf];
                if (pos < k.O.rsn_win) { RsnRec r; r.code = (uint8_t)why; r.podset = (uint8_t)pi; r.flavor = (int16_t)f; r.resource = (int16_t)(why == KQ_RSN_NOT_IN_NOMINATION ? res_name : -1); r.pad = 0; r.a = 0; r.b = 0; r.c = 0; win[pos] = r; }
                continue;
              }
              for (int kk = 0; kk < nf; kk++) {
                const int c = jj * nf + kk;
                const int pm = w.cell_pm[c] & 0x3f;
                if (!((w.cell_pm[c] & 0x80) || pm == PM_NOFIT || pm == PM_NEEDS)) continue;
                if (pos < k.O.rsn_win) {
                  RsnRec r; r.podset = (uint8_t)pi; r.flavor = (int16_t)f; r.resource = (int16_t)w.f_res[kk]; r.pad = 0;
                  if (pm == PM_NOFIT && !(w.cell_pm[c] & 0x80)) { r.code = (uint8_t)KQ_RSN_EXCEEDS_MAX_CAPACITY; r.a = assumed_usage(w, f * nR + w.f_res[kk]); r.b = w.f_qty[kk]; r.c = w.cell_aux[c]; }
                  else { r.code = (uint8_t)KQ_RSN_INSUFFICIENT_UNUSED; r.a = w.cell_aux[c]; r.b = 0; r.c = 0; }
                  win[pos] = r;
                }
                pos++;
              }
            }
            if (lane == 0) { if (nrsn0 + nrec > k.O.rsn_win) { w.nrsn = nrsn0 > k.O.rsn_win ? nrsn0 : k.O.rsn_win; w.rsn_over = 1; } else w.nrsn = nrsn0 + nrec; }
          }
          wsync_lds();
        }
        // ---- ordered scan of the pass (uniform) ------------------------------------------
        for (int jj = 0; jj < nfl && !stop && !fast; jj++) {
          const int j = cs + jj;
          attempted = j;
          const int f = S.rg_flavor[f0 + j];
          if (w.cell_pm[jj * nf] == PM_SKIP) {
            reasons++;
            if (lane == 0) { const int why = w.cell_borrow[jj * nf]; rsn_push(k, w, why, pi, f, why == KQ_RSN_NOT_IN_NOMINATION ? res_name : -1, 0, 0, 0); }
            continue;
          }
          if (lane == 0) w.bytes += (int64_t)nf * 40 * plen;  // nf fitsResourceQuota calls, (D+1) x 5 planes x 8 B each
          int rep_pm = PM_FIT; int64_t rep_borrow = 0; int64_t rep_key = pref_key(PM_FIT, 0, w.pol);
          for (int kk = 0; kk < nf; kk++) {
            int c = jj * nf + kk;
            int pm = w.cell_pm[c] & 0x3f; int borrow = w.cell_borrow[c];
            bool had_status = (w.cell_pm[c] & 0x80) || pm == PM_NOFIT || pm == PM_NEEDS;
            int64_t reqv = w.f_qty[kk];
            if (w.slice_row >= 0) {
              int sfl; int64_t sq;
              slice_of(k, w, psg, w.f_slot[kk], &sfl, &sq);
              if (w.cell_pm[c] & 0x40) {
                // the flavor is not the replaced slice's: worstGranularMode() and a reason; the closure goes on to fitsResourceQuota
                // (its reasons are appended too) and returns at the noFit check below (flavorassigner.go:1130-1162)
                if (lane == 0) rsn_push(k, w, KQ_RSN_SLICE_FLAVOR_MISMATCH, pi, f, w.f_res[kk], sfl, 0, 0);
                reasons++;
                rep_pm = PM_NOFIT; rep_borrow = 0; rep_key = -1;
              } else reqv -= sq;
            }
            if (had_status && lane == 0) {  // the string fitsResourceQuota formats (:1353-1373)
              const int fr = f * nR + w.f_res[kk];
              if (pm == PM_NOFIT && !(w.cell_pm[c] & 0x80)) rsn_push(k, w, KQ_RSN_EXCEEDS_MAX_CAPACITY, pi, f, w.f_res[kk], assumed_usage(w, fr), reqv, w.cell_aux[c]);
              else rsn_push(k, w, KQ_RSN_INSUFFICIENT_UNUSED, pi, f, w.f_res[kk], w.cell_aux[c], 0, 0);
            }
            if (rep_pm == PM_NOFIT) { if (had_status) reasons++; continue; }  // oracle result unused past a noFit (:1161)
            if (pm == PM_NEEDS) {
              int opm, ob;
              if (dead_all) { opm = PM_NOCAND; ob = borrow; }   // a dead simulation (above): whatever stands here cannot be observed
              else if constexpr (LEAN) {
                const bool can_search = KQ_POL_WITHIN_CQ(w.pol) != KQ_POLICY_NEVER || (w.plen > 1 && KQ_POL_RECLAIM(w.pol) != KQ_POLICY_NEVER);
                if (can_search) {
                  if (sim_cells && sim_cells[c] >= 0) {   // listed by an earlier pass, run by k_nominate_sim: go on to the scans behind this one
                    const HelpRes hr = k.sim_res[sim_cells[c]];
                    opm = hr.pm; ob = hr.borrow;
                    if (lane == 0) w.bytes += hr.bytes;
                  } else {
                    if (sim_elig && !sim_cells) sim_emit(k, w, sim_key, f0, cs, nfl, nf);
                    if (lane == 0) w.defer_head = 1;
                    wsync();
                    return;
                  }
                } else { opm = PM_NOCAND; ob = borrow; }  // simulate_preemption with an empty target set
              } else {
                if (batched && w.cell_task[c] != 0xff) {  // the batch evaluated it; only a consumed result is charged
                  const HelpRes hr = hbox->res[w.cell_task[c]];
                  opm = hr.pm; ob = hr.borrow;
                  if (lane == 0) w.bytes += hr.bytes;
                } else if (sim_cells && sim_cells[c] >= 0) {   // k_nominate_sim ran it (K::sim_*); charged when consumed
                  if (sim_q > 0) CSTAT(19, 1);
                  const HelpRes hr = k.sim_res[sim_cells[c]];
                  opm = hr.pm; ob = hr.borrow;
                  if (lane == 0) w.bytes += hr.bytes;
                } else {
                  simulate_preemption(k, w, slot, usage, removed, f * nR + w.f_res[kk], w.cell_val[c], borrow, &opm, &ob);
                }
              }
              pm = opm; borrow = ob;
            }
            if (had_status) reasons++;
            int64_t key = pref_key(pm, borrow, w.pol);
            if (rep_key > key) { rep_pm = pm; rep_borrow = borrow; rep_key = key; }   // isPreferred(rep, mode) -> rep = mode
            if (rep_pm == PM_NOFIT) continue;
            if (lane == 0) { w.cur_mode[kk] = fa_mode(pm); w.cur_borrow[kk] = borrow; }
          }
          wsync_lds();
          bool take = false;
          if (gate(k, KQ_GATE_FLAVOR_FUNGIBILITY)) {
            if (!should_try_next(rep_pm, rep_borrow, w.pol)) { take = true; stop = true; }
            else if (rep_key > best_key) take = true;
          } else if (rep_pm > best_pm) {
            take = true;
            if (rep_pm == PM_FIT) stop = true;
          }
          if (take) {
            best = j; best_key = rep_key; best_pm = rep_pm;
            if (lane == 0) for (int kk = 0; kk < nf; kk++) { w.best_mode[kk] = w.cur_mode[kk]; w.best_borrow[kk] = w.cur_borrow[kk]; }
            wsync_lds();
          }
        }
      }
      if (LEAN || KQ_TAS_PROCESS(k, w)) KQ_TS(k, 59);  // lean: the serial choice among the flavors
      int tried = -1;
      if (gate(k, KQ_GATE_FLAVOR_FUNGIBILITY)) tried = (attempted == nflv - 1) ? -1 : attempted;
      else tried = 0;
      bool status_nil = best >= 0 && best_pm == PM_FIT;
      if (best < 0) {  // len(flavors)==0 && requests.Len()>0  (:826)
        group_failed = true; ps_reasons = reasons;
        if (lane == 0 && k.O.rsn_win > 0) {  // psAssignment.Status = status (:829): this group's reasons replace the podset's
          RsnRec* win = k.O.rsn + (size_t)w.h * k.O.rsn_win;
          const int cnt = w.nrsn - w.rsn_g0;
          for (int q = 0; q < cnt; q++) win[w.rsn_ps0 + q] = win[w.rsn_g0 + q];
          w.nrsn = w.rsn_ps0 + cnt;
        }
        break;
      }
      if (status_nil && lane == 0) w.nrsn = w.rsn_g0;  // a nil status drops the scan's reasons (:1199-1207)
      // record groupFlavors[res] for every covered requested resource (maps.Copy :831)
      if (lane == 0) {
        int f = S.rg_flavor[f0 + best];
        for (int kk = 0; kk < nf; kk++) {
          int a2 = w.f_slot[kk];
          w.req_done[a2] = 1; w.req_flavor[a2] = f; w.req_mode[a2] = (uint8_t)w.best_mode[kk];
          w.req_borrow[a2] = w.best_borrow[kk]; w.req_tried[a2] = tried;
        }
      }
      wsync_lds();
      if (!status_nil) ps_reasons += reasons;
    }
    // PodSetAssignment.RepresentativeMode :386-404 ; Assignment.append :1017-1041
    int pmode = M_FIT;
    // atLeastOnePodsAssignmentFailed :842-845: a podset that requests something and ends without any flavor — also when every request of
    // it was skipped (:809-817) and no scan ran at all
    ps_nflavors = 0;   // the flavors the podset (group) ends with: seeds of the second pass, the scans' (a later scan may have overwritten an earlier one's)
    for (int a = 0; a < w.nreq; a++) if (w.req_done[a]) { ps_nflavors++; if (w.req_mode[a] < ps_mode) ps_mode = w.req_mode[a]; }
    bool failed = w.nreq > 0 && (group_failed || ps_nflavors == 0);
    bool finished = false;
    if constexpr (!LEAN) { if (gn > 1) { pmode = group_finish(k, w, pi, gn, counts, pods_cov, group_failed, ps_reasons, &failed); finished = true; } }
    if (finished) {}
    else if (group_failed) {
      // groupFlavors = nil: the podset keeps no flavor and contributes no usage
      pmode = ps_reasons == 0 ? M_FIT : M_NOFIT;
    } else {
      pmode = ps_reasons == 0 ? M_FIT : (ps_nflavors == 0 ? M_NOFIT : ps_mode);
      if (lane == 0) {
        for (int a = 0; a < w.nreq; a++) {
          if (!w.req_done[a]) continue;
          size_t o = (size_t)psg * nR + w.req_res[a];
          O.flavor[o] = w.req_flavor[a]; O.res_mode[o] = w.req_mode[a]; O.tried_idx[o] = w.req_tried[a];
          if (w.req_borrow[a] > w.borrowing) w.borrowing = w.req_borrow[a];
          int fr = w.req_flavor[a] * nR + w.req_res[a];
          int e = -1;
          for (int i = 0; i < w.nuse; i++) if (w.use_fr[i] == fr) e = i;
          if (e < 0) {
            if (w.nuse >= KQ_MAXU) { *O.error = KQ_EUNSUPPORTED; continue; }
            e = w.nuse++; w.use_fr[e] = fr; w.use_qty[e] = 0; w.use_mode[e] = M_FIT;
          }
          int64_t amount = w.req_qty[a];
          if (w.slice_row >= 0) { int sfl; int64_t sq; slice_of(k, w, psg, a, &sfl, &sq); amount -= sq; }  // Assignment.append :1028-1035
          w.use_qty[e] = a_addi(w.use_qty[e], amount);
          if (w.req_mode[a] < w.use_mode[e]) w.use_mode[e] = w.req_mode[a];
        }
        for (int a = 0; a < w.nreq; a++) if (w.req_done[a]) w.bytes += 16;  // (flavor, mode, borrow, tried) out
      }
      wsync();
    }
    if (pmode < rep) rep = pmode;
    if (failed) {  // :848-853: later podsets are never assigned
      for (int i = lane + (pi + gn) * nR; i < w.nps * nR; i += WAVE) {
        size_t o = (size_t)w.ps_base * nR + i;
        O.flavor[o] = -1; O.res_mode[o] = M_NOFIT; O.tried_idx[o] = -1;
      }
      for (int q = pi + gn + lane; q < w.nps; q += WAVE) O.ps_count[w.ps_base + q] = H.ps_count[w.ps_base + q];
#ifdef KQ_TAS_CYCLE
      if (k.tc && lane == 0) w.ta.af_early = pi + gn;
#endif
      wsync();
      break;
    }
  }
  if (LEAN || KQ_TAS_PROCESS(k, w)) KQ_TS(k, 60);  // lean: usage list / outputs of the podsets
  if (!any_ps) rep = M_NOFIT;  // RepresentativeMode with no podsets :212-215
  if (lane == 0) {
    w.rep_mode = rep;
    if (O.rsn_win > 0) O.rsn_n[w.h] = w.rsn_over ? -w.nrsn : w.nrsn;
  }
  wsync();
#ifdef KQ_TAS_CYCLE
  if constexpr (!LEAN) if (k.tc && w.rep_mode != M_NOFIT && !w.ta.af_early) tc_assign_tas(k, w, slot);  // flavorassigner.go:857-863
#endif
}

// ------------------------------------------------------------------------------------------------
// Preemptor.GetTargets (preemption.go:132-155): slots = flavor-resources of the assignment;
// need = flavorResourcesNeedPreemption (:586-597); quantities = Assignment.TotalRequestsFor
// (flavorassigner.go:267-296: original requests scaled to the assigned count, zero quantities and
// the injected "pods" pseudo-resource excluded).
// ------------------------------------------------------------------------------------------------
KQ_DEV void prepare_target_slots(const K& k, Wave& w) {
  const DSnap& S = k.S; const DHeads& H = k.H; const DOut& O = k.O;
  if (lane_id() == 0) {
    w.ns = w.nuse;
    for (int e = 0; e < w.nuse; e++) { w.s_fr[e] = w.use_fr[e]; w.s_need[e] = w.use_mode[e] == M_PREEMPT; w.s_qty[e] = 0; w.s_inu[e] = 0; }
    for (int pi = 0; pi < w.nps; pi++) {
      int psg = w.ps_base + pi;
      int count = H.ps_count[psg], newc = O.ps_count[psg];
      if (w.slice_row >= 0) newc = count - (H.ps_slice_count ? H.ps_slice_count[psg] : 0);  // TotalRequestsFor :265-267
      for (int e = H.ps_req_off[psg]; e < H.ps_req_off[psg + 1]; e++) {
        int64_t q = H.req_qty[e];
        if (count != 0 && count != newc) q = sat_mul(q / (int64_t)count, (int64_t)newc);
        if (q == 0) continue;
        int res = H.req_res[e];
        int fl = O.flavor[(size_t)psg * S.nR + res];
        if (fl < 0) continue;
        int fr = fl * S.nR + res;
        for (int u = 0; u < w.ns; u++) if (w.s_fr[u] == fr) { w.s_qty[u] = a_addi(w.s_qty[u], q); w.s_inu[u] = 1; }
      }
    }
  }
  wsync();
}

KQ_DEV Search get_targets(const K& k, Wave& w, int slot, const int64_t* usage, const uint8_t* removed) {
  Search s = make_search(k, w, slot, usage, removed);
  prepare_target_slots(k, w);
#ifdef KQ_TAS_CYCLE
  if (k.tc) tc_search_begin(k, w, slot);   // preemption.go:135-138: tasRequests of the assignment; the walk then carries the leaf usage
#endif
  if (k.C.fair_sharing) fair_search(s); else classical_search(s);
  wsync();
#ifdef KQ_TAS_CYCLE
  if (k.tc) tc_search_end(w);
#endif
  return s;
}

// scheduler.go:840-856
KQ_DEV bool last_assignment_outdated(const K& k, int h, int cq) {
  if (gate(k, KQ_GATE_PRESERVE_SCAN_PROGRESS)) {
    uint64_t lh = k.H.last_hash[h], ch = k.H.hash[h];
    if (!(lh == 0 || ch == 0 || lh == ch)) return true;
    if (k.C.cycle - k.H.last_cycle[h] <= 1) return false;
  }
  return k.S.cq_gen[cq] > k.H.last_generation[h];
}

KQ_DEV void load_head(const K& k, Wave& w, int h) {
  const DSnap& S = k.S; const DHeads& H = k.H;
#if !defined(KQ_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
  // Uniform code, every lane: the head's own chain (head -> ClusterQueue -> policy / path) and, riding with it, the request rows of EVERY
  // podset of the head, touched once by the whole wave. A podset's turn in assign_flavors is a dependent chain of its own (count /
  // request offsets -> requests -> resource order / resource group: ~4 round trips of a lone wave), and the chains of a head's podsets do
  // not depend on each other: here they are one chain for all of them, in flight together with the head's, and the per-podset loads
  // find their lines in the cache (tools/prof_lean.py: "requests in iterator order" was 23 % of k_nominate_lean). The touched values
  // are dropped.
  const int lane = lane_id();
  const int cq = H.cq[h], pb = H.ps_off[h], pn = H.ps_off[h + 1] - pb;
  const int64_t prio = H.priority[h], ts = H.queue_ts[h];
  const uint32_t hfl = H.flags[h];
  const int slice_row = (H.slice_row && gate(k, KQ_GATE_ELASTIC_JOBS)) ? H.slice_row[h] : -1;
  // second round trip
  const uint32_t pol = S.cq_policy[cq];
  const int plen = S.plen[cq];
  const int pth = lane < KQ_MAXD ? S.path[(size_t)cq * KQ_MAXD + lane] : 0;
  const int pnc = pn < 0 ? 0 : (pn < KQ_MAXPS ? pn : KQ_MAXPS);
  int o = 0, sink = 0;
  if (lane <= pnc) { o = H.ps_req_off[pb + lane]; if (lane < pnc) { sink += H.ps_count[pb + lane]; if (H.ps_group) sink += H.ps_group[pb + lane]; } }
  // third and fourth
  const int e0 = __shfl(o, 0, 64), e1 = __shfl(o, pnc, 64);
  if (e0 + lane < e1) {
    const int r = H.req_res[e0 + lane];
    sink += (int)H.req_qty[e0 + lane] + S.resource_order[r] + S.cq_res_rg[(size_t)cq * S.nR + r];
  }
  if (lane < KQ_MAXD) w.path[lane] = pth;
  if (lane == 0) {
    w.h = h; w.cq = cq; w.prio = prio; w.ts = ts; w.hflags = hfl;
    w.pol = pol;
    w.ps_base = pb; w.nps = pn;
    w.plen = plen;
    w.has_last = (hfl & KQ_HEAD_HAS_LAST_ASSIGNMENT) ? 1 : 0;
    w.bytes = 0;
    w.slice_row = slice_row;
    if (pn > KQ_MAXPS) *k.O.error = KQ_EUNSUPPORTED;
  }
  asm volatile("" :: "v"(sink));
  wsync();
#else
  if (lane_id() == 0) {
    w.h = h; w.cq = H.cq[h]; w.prio = H.priority[h]; w.ts = H.queue_ts[h]; w.hflags = H.flags[h];
    w.pol = S.cq_policy[w.cq];
    w.ps_base = H.ps_off[h]; w.nps = H.ps_off[h + 1] - H.ps_off[h];
    w.plen = S.plen[w.cq];
    for (int i = 0; i < KQ_MAXD; i++) w.path[i] = S.path[(size_t)w.cq * KQ_MAXD + i];
    w.has_last = (w.hflags & KQ_HEAD_HAS_LAST_ASSIGNMENT) ? 1 : 0;
    w.bytes = 0;
    w.slice_row = (H.slice_row && gate(k, KQ_GATE_ELASTIC_JOBS)) ? H.slice_row[h] : -1;
    if (w.nps > KQ_MAXPS) *k.O.error = KQ_EUNSUPPORTED;
  }
  wsync();
#endif
}

// Scheduler.getAssignments + getInitialAssignments (scheduler.go:821-924). Leaves the chosen
// assignment in O.flavor/res_mode/tried_idx/ps_count + w.use_*/w.rep_mode/w.borrowing and its targets in the
// returned Search (w.ntgt rows).
KQ_DEV Search get_assignments_inner(const K& k, Wave& w, int slot, const int64_t* usage, const uint8_t* removed, bool nominate_map) {
  const DHeads& H = k.H;
  Search s = make_search(k, w, slot, usage, removed);
  assign_flavors<false>(k, w, slot, usage, removed, nullptr, nominate_map);
  w.ntgt = 0;
  int arm = w.rep_mode;
  if (arm == M_FIT) return s;
  if (arm == M_PREEMPT) {
    s = get_targets(k, w, slot, usage, removed);
    if (w.ntgt > 0) return s;
  }
  // PartialAdmission: PodSetReducer.Search (podset_reducer.go:28-86)
  bool can_partial = false;
  int total_delta = 0;
  if (gate(k, KQ_GATE_PARTIAL_ADMISSION) && w.nps <= KQ_MAXPS)
    for (int pi = 0; pi < w.nps; pi++) {
      int c = H.ps_count[w.ps_base + pi], mc = H.ps_min_count[w.ps_base + pi];
      if (mc >= 0 && c > mc) { can_partial = true; total_delta += c - mc; }
    }
  if (!can_partial || total_delta == 0) { w.ntgt = 0; return s; }
  int lo = 0, hi = total_delta + 1, good = -1, last_probe = -1;
  auto probe = [&](int si) -> bool {
    if (lane_id() == 0)
      for (int pi = 0; pi < w.nps; pi++) {
        int c = H.ps_count[w.ps_base + pi], mc = H.ps_min_count[w.ps_base + pi];
        int d = (mc >= 0 && c > mc) ? c - mc : 0;
        w.counts[pi] = c - (int)((int64_t)d * (int64_t)si / (int64_t)total_delta);
      }
    wsync();
    assign_flavors<false>(k, w, slot, usage, removed, w.counts, nominate_map);
    w.ntgt = 0;
    last_probe = si;
    if (w.rep_mode == M_FIT) return true;
    if (w.rep_mode == M_PREEMPT) { s = get_targets(k, w, slot, usage, removed); return w.ntgt > 0; }
    return false;
  };
  while (lo < hi) {  // sort.Search
    int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
    if (!probe(mid)) lo = mid + 1; else { hi = mid; good = mid; }
  }
  // Re-running an assignment to regenerate its outputs is an artefact of keeping one output slot per
  // head; the reference keeps the value. It is not algorithmic traffic: restore the byte counter.
  const int64_t bytes_before = w.bytes;
  if (good >= 0 && lo == good) {
    if (last_probe != good) probe(good);  // regenerate the outputs of the accepted probe
  } else {
    // no reduced count works: the full assignment, no targets (scheduler.go:923)
    assign_flavors<false>(k, w, slot, usage, removed, nullptr, nominate_map);
    w.ntgt = 0;
#ifdef KQ_TAS_CYCLE
    // the reference returns the SAME fullAssignment object GetTargets (scheduler.go:897) already looked at: a psError that
    // WorkloadsTopologyRequests left on one of its podsets (flavorassigner.go:290, preemption.go:135) is still there when
    // updateAssignmentForTAS asks for its RepresentativeMode. Regenerating the outputs must not lose it.
    if (k.tc && arm == M_PREEMPT) tc_requests(k, w);
#endif
  }
  wsync();
  if (lane_id() == 0) w.bytes = bytes_before;
  wsync();
  return s;
}

// workloadslicing.ReplacedWorkloadSlice (scheduler.go:883-899): the slice the head replaces is a target of every assignment that comes
// with targets at all — Fit: the slice alone; Preempt with victims (also through the PodSetReducer): the slice and the victims; Preempt
// without victims / NoFit: nil. Appended (targets are a set; the host sorts them). If the victim search named the slice itself it is
// listed once, under the search's reason (kq_engine.h KQ_REASON_REPLACED_SLICE).
KQ_DEV Search get_assignments(const K& k, Wave& w, int slot, const int64_t* usage, const uint8_t* removed, bool nominate_map) {
  // ReplacedWorkloadSlice looks the old slice up in queue.Workloads (workloadslicing.go:371): inside an overlap recomputation
  // (SimulateWorkloadRemoval of the other preemptions' victims, scheduler.go:726-727) a slice that was preempted is not there any more
  // and the head is assigned like any other workload
  const int slice_saved = w.slice_row;
  if (slice_saved >= 0 && removed && removed[slice_saved]) { wsync(); if (lane_id() == 0) w.slice_row = -1; wsync(); }
  Search s = get_assignments_inner(k, w, slot, usage, removed, nominate_map);
  if (w.slice_row != slice_saved) {
    wsync(); if (lane_id() == 0) w.slice_row = slice_saved; wsync();
#ifdef KQ_TAS_CYCLE
    if (k.tc) tc_update_assignment(k, w, slot, s.trow, w.ntgt);
#endif
    return s;
  }
  if (w.slice_row >= 0 && (w.rep_mode == M_FIT || w.ntgt > 0)) {
    if (lane_id() == 0) {
      bool dup = false;
      for (int t = 0; t < w.ntgt; t++) if (s.trow[t] == w.slice_row) dup = true;
      if (!dup) {
        if (w.ntgt >= k.X.tgt_cap) *k.O.error = KQ_ECAPACITY;
        else { s.trow[w.ntgt] = w.slice_row; s.treason[w.ntgt] = KQ_REASON_REPLACED_SLICE; w.ntgt++; }
      }
    }
    wsync();
  }
#ifdef KQ_TAS_CYCLE
  if (k.tc) tc_update_assignment(k, w, slot, s.trow, w.ntgt);   // scheduler.go:941-985, behind getInitialAssignments
#endif
  return s;
}

// publish the head's assignment internals for the process kernel
KQ_DEV void publish_assignment(const K& k, Wave& w, const Search& s, int h) {
  const DOut& O = k.O;
  int pos = 0;
  if (lane_id() == 0) {
    O.use_n[h] = w.nuse;
    for (int e = 0; e < w.nuse; e++) { O.use_fr[(size_t)h * KQ_MAXU + e] = w.use_fr[e]; O.use_qty[(size_t)h * KQ_MAXU + e] = w.use_qty[e]; }
    O.borrowing[h] = w.borrowing;
    pos = w.ntgt > 0 ? atomic_add_i32(O.pool_count, w.ntgt) : 0;
    if (pos + w.ntgt > O.pool_cap) { if (*O.error == 0) *O.error = KQ_ECAPACITY; O.tgt_pos[h] = 0; O.tgt_n[h] = 0; }
    else { O.tgt_pos[h] = pos; O.tgt_n[h] = w.ntgt; }
    w.counts[0] = pos;  // broadcast through LDS
  }
#ifdef KQ_TAS_CYCLE
  if (k.tc && w.ta.pool_own) wsync_lds(); else wsync();   // (k_process_tas: the word travels through LDS; the fence below covers the global stores)
#else
  wsync();
#endif
  pos = w.counts[0];
  if (pos + w.ntgt <= O.pool_cap)
    for (int t = lane_id(); t < w.ntgt; t += WAVE) { O.pool_row[pos + t] = s.trow[t]; O.pool_reason[pos + t] = s.treason[t]; }
  wsync();
#ifdef KQ_TAS_CYCLE
  if (k.tc) tc_publish(k, w, s.slot);
#endif
}

// outputs of a nominated head that k_process / the host read (everything but the assignment rows assign_flavors wrote itself)
KQ_DEV void nominate_finish(const K& k, Wave& w, int h) {
  if (lane_id() == 0) {
    k.O.nominated_mode[h] = (uint8_t)w.rep_mode;
    k.O.mode[h] = (uint8_t)w.rep_mode;
    k.O.status[h] = KQ_ST_NOT_NOMINATED; k.O.action[h] = KQ_ACT_NONE; k.O.requeue_reason[h] = KQ_RQ_GENERIC; k.O.skip[h] = KQ_SKIP_NONE;
    k.O.order[h] = -1;
    atomic_add_i64(k.O.stat_bytes, (long long)w.bytes);
  }
}
// First pass of nominate: every head whose assignment needs neither a victim search nor the partial-admission search — the bulk
// of every cycle — is finished here by a kernel that contains nothing else; the others are appended to k.defer_list for the full
// pass (k_nominate). Same results: a deferred head is recomputed from scratch.
KQ_DEV void nominate_head_lean(const K& k, Wave& w, int h) {
  if (k.shard.mine && !k.shard.mine[h]) return;  // sharded nominate: another rank's head
  KQ_T0();
  load_head(k, w, h);
  if (lane_id() == 0) { if (w.has_last && last_assignment_outdated(k, h, w.cq)) w.has_last = 0; w.defer_head = 0; }
  wsync();
  KQ_TS(k, 56);  // lean: load_head
  assign_flavors<true>(k, w, 0, k.usage, nullptr, nullptr, false);
  KQ_TS(k, 61);  // lean: assign_flavors (57..60 are its parts)
  bool defer = w.defer_head != 0 || w.slice_row >= 0;  // (a head that replaces a workload slice always has a target: the full pass publishes it)
  if (!defer && w.rep_mode != M_FIT) {
    // getInitialAssignments (scheduler.go:880-924) past the first Assign: Preempt asks GetTargets, then the partial-admission search
    const bool can_search = KQ_POL_WITHIN_CQ(w.pol) != KQ_POLICY_NEVER || (w.plen > 1 && KQ_POL_RECLAIM(w.pol) != KQ_POLICY_NEVER);
    if (w.rep_mode == M_PREEMPT && can_search) defer = true;
    if (!defer && gate(k, KQ_GATE_PARTIAL_ADMISSION))
      for (int pi = 0; pi < w.nps; pi++) {
        const int c = k.H.ps_count[w.ps_base + pi], mc = k.H.ps_min_count[w.ps_base + pi];
        if (mc >= 0 && c > mc) defer = true;
      }
  }
  if (defer) {
    if (lane_id() == 0) { const int pos = atomic_add_i32(k.defer_count, 1); k.defer_list[pos] = h; }
    wsync();
    return;
  }
  if (lane_id() == 0) {
    k.O.use_n[h] = w.nuse;
    for (int e = 0; e < w.nuse; e++) { k.O.use_fr[(size_t)h * KQ_MAXU + e] = w.use_fr[e]; k.O.use_qty[(size_t)h * KQ_MAXU + e] = w.use_qty[e]; }
    k.O.borrowing[h] = w.borrowing;
    k.O.tgt_pos[h] = 0; k.O.tgt_n[h] = 0;
  }
  nominate_finish(k, w, h);
  wsync();
  KQ_TS(k, 62);  // lean: publish
}

// nominate (scheduler.go:665-705) for one head
KQ_DEV void nominate_head(const K& k, Wave& w, int h, int slot) {
  if (k.shard.mine && !k.shard.mine[h]) return;  // sharded nominate: another rank's head
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  const long long _h0 = clock64();
#endif
  load_head(k, w, h);
  if (w.has_last && last_assignment_outdated(k, h, w.cq)) { if (lane_id() == 0) w.has_last = 0; }
  wsync();
  Search s = get_assignments(k, w, slot, k.usage, nullptr, false);
  publish_assignment(k, w, s, h);
  if (lane_id() == 0) {
    k.O.nominated_mode[h] = (uint8_t)w.rep_mode;
    k.O.mode[h] = (uint8_t)w.rep_mode;
    k.O.status[h] = KQ_ST_NOT_NOMINATED; k.O.action[h] = KQ_ACT_NONE; k.O.requeue_reason[h] = KQ_RQ_GENERIC; k.O.skip[h] = KQ_SKIP_NONE;
    k.O.order[h] = -1;
    atomic_add_i64(k.O.stat_bytes, (long long)w.bytes);
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
    // per-head latency by outcome: the kernel lasts as long as its slowest head
    const long long dt = clock64() - _h0;
    const int cls = w.rep_mode == M_FIT ? 0 : (w.rep_mode == M_PREEMPT ? 1 : 2);
    atomic_add_i64((long long*)k.prof + 21 + cls, dt);
    atomic_add_i64((long long*)k.prof + 24 + cls, 1);
    atomicMax((unsigned long long*)k.prof + 30, (unsigned long long)dt);
#endif
  }
  wsync();
}

// k_nominate_sim, round r (behind the lean pass: r = 0; behind emit round q: r = q + 1): one wave, the tasks listed since the previous
// round pulled by ticket (two searches differ by 10 x); the result of a task is a pure function of the cycle-start snapshot, the head
// and the cell, so any wave may run it and any number of them at once. sim_ctl: [0] tasks listed, [SIMC_TICKET + r] tickets of round
// r, [SIMC_START + r] first task of round r (written by the round before it: nothing is listed while a sim round runs).
constexpr int SIMC_TICKET = 1, SIMC_EMIT = SIMC_TICKET + SIM_ROUNDS + 1, SIMC_START = SIMC_EMIT + SIM_ROUNDS, SIMC_WORDS = SIMC_START + SIM_ROUNDS + 2;
KQ_DEV void sim_worker(const K& k, Wave& w, int slot, int round) {
  const int lane = lane_id();
  int cur = -1;
  int nt = k.sim_ctl[0];
  if (nt > k.sim_cap) nt = k.sim_cap;
  const int start = round == 0 ? 0 : k.sim_ctl[SIMC_START + round];
  wsync();
  if (lane == 0) k.sim_ctl[SIMC_START + round + 1] = nt;   // (every wave of the round writes the same value)
  for (;;) {
    int t = 0;
    if (lane == 0) t = start + atomic_add_i32(&k.sim_ctl[SIMC_TICKET + round], 1);
    t = wuniform_i32(t);
    if (t >= nt) break;
    const SimTask task = k.sim_task[t];
    if (task.head != cur) { load_head(k, w, task.head); cur = task.head; }
    wsync();
    if (lane == 0) w.bytes = 0;
    wsync();
    int pm = 0, borrow = 0;
    simulate_preemption(k, w, slot, k.usage, nullptr, task.fr, task.val, task.base_borrow, &pm, &borrow);
    wsync();
    if (lane == 0) { k.sim_res[t].pm = pm; k.sim_res[t].borrow = borrow; k.sim_res[t].bytes = w.bytes; }
    wsync();
  }
}
// k_nominate_emit: a deferred head walked again by the lean code with the results of the scans listed so far in hand — it gets as far as
// the next scan that needs simulations and lists them (assign_flavors<true> -> sim_emit). Its outputs are dropped: the full pass
// recomputes the head, reading every listed result where it would have searched.
KQ_DEV void nominate_head_emit(const K& k, Wave& w, int h) {
  const int ns = k.sim_nscan[h];
  if (ns == 0 || ns >= SIM_KS) return;   // nothing was listed for it (deferred for another reason, not eligible, no room), or its table is full
  load_head(k, w, h);
  if (lane_id() == 0) { if (w.has_last && last_assignment_outdated(k, h, w.cq)) w.has_last = 0; w.defer_head = 0; }
  wsync();
  assign_flavors<true>(k, w, 0, k.usage, nullptr, nullptr, false);
  wsync();
}

// ------------------------------------------------------------------------------------------------
// processEntry (scheduler.go:392-523) — sequential inside one root-cohort tree
// ------------------------------------------------------------------------------------------------
KQ_DEV UP up_plane(const K& k, const Wave& w, int plane, int fr) {
  return UP{&k.S, plane == 0 ? k.usage_work : k.usage_np, w.pc_lds, w.pc_on, w.pc_ncq, w.pc_ncoh, plane, fr, k.cq_dirty};
}
// LDS <-> HBM for the tree's cohort rows (both planes)
// tid / nthreads: the threads sharing the copy; the caller synchronises them afterwards
KQ_DEV void pc_load_by(const K& k, const Wave& w, int64_t* pcl, int tree, int tid, int nthreads) {
  const DSnap& S = k.S;
  if (!w.pc_on) return;
  const int n0 = S.tree_node_off[tree] + w.pc_ncq;
  const int total = w.pc_ncoh * S.nfr;
  for (int i = tid; i < total; i += nthreads) {
    int node = S.tree_nodes[n0 + i / S.nfr], fr = i % S.nfr;
    pcl[i] = k.usage_work[ix(S, node, fr)];
    pcl[(size_t)total + i] = k.usage_np[ix(S, node, fr)];
  }
}
KQ_DEV void pc_flush_by(const K& k, const Wave& w, const int64_t* pcl, int tree, int tid, int nthreads) {
  const DSnap& S = k.S;
  if (!w.pc_on) return;
  const int n0 = S.tree_node_off[tree] + w.pc_ncq;
  const int total = w.pc_ncoh * S.nfr;
  for (int i = tid; i < total; i += nthreads) {
    int node = S.tree_nodes[n0 + i / S.nfr], fr = i % S.nfr;
    k.usage_work[ix(S, node, fr)] = pcl[i];
    k.usage_np[ix(S, node, fr)] = pcl[(size_t)total + i];
  }
}
KQ_DEV void pc_load(const K& k, Wave& w, int64_t* pcl, int tree) { pc_load_by(k, w, pcl, tree, lane_id(), WAVE); wsync(); }
KQ_DEV void pc_flush(const K& k, Wave& w, int64_t* pcl, int tree) { pc_flush_by(k, w, pcl, tree, lane_id(), WAVE); wsync(); }

// apply / revert the removal of a target row on usage_np, restricted to the entry's own flavor-resources
KQ_DEV void np_apply_row_restricted(const K& k, Wave& w, int row, bool add) {
  const DSnap& S = k.S;
  int c = S.adm_cq[row];
  const int32_t* cpath = S.path + (size_t)c * KQ_MAXD;
  int cplen = S.plen[c];
  for (int u = lane_id(); u < w.nuse; u += WAVE) {
    int fr = w.use_fr[u];
    UP g = up_plane(k, w, 1, fr);
    for (int e = S.adm_use_off[row]; e < S.adm_use_off[row + 1]; e++) {
      if (S.adm_use_fr[e] != fr) continue;
      if (add) add_usage(S, cpath, cplen, fr, S.adm_use_qty[e], g); else remove_usage(S, cpath, cplen, fr, S.adm_use_qty[e], g);
    }
  }
  wsync();
}
// The same for a list of target rows, in list order (add = false) or in reverse (add = true: the revert). restricted: only the
// entry's own flavor-resources (entry_fits), else every flavor-resource of the rows (PreemptedWorkloads.Insert). When the tree's
// row records are complete and the amounts plain, the records of 64 targets are loaded at once (one lane each) and handed to the
// chain lanes through the scalar unit: per target there is then one round of cell / quota loads instead of six dependent ones,
// and no fence — a flavor-resource column is always handled by the same lane, in program order.
KQ_DEV void np_apply_targets(const K& k, Wave& w, const int32_t* trows, int nt, bool add, bool restricted, int tree) {
  const DSnap& S = k.S;
  const int lane = lane_id();
  if (!(fs_plain_now(k) && S.rec_ok && S.rec_ok[tree])) {
    if (restricted) {
      for (int q = 0; q < nt; q++) { const int t = add ? nt - 1 - q : q; if (!k.preempted[trows[t]]) np_apply_row_restricted(k, w, trows[t], add); }
    } else {
      for (int t = 0; t < nt; t++) {
        const int row = trows[t];
        if (lane == 0 && k.preempted[row] == 1) {
          const int c = S.adm_cq[row];
          for (int en = S.adm_use_off[row]; en < S.adm_use_off[row + 1]; en++) {
            UP g = up_plane(k, w, 1, S.adm_use_fr[en]);
            remove_usage(S, S.path + (size_t)c * KQ_MAXD, S.plen[c], S.adm_use_fr[en], S.adm_use_qty[en], g);
          }
        }
        wsync();
      }
    }
    return;
  }
  for (int base = 0; base < nt; base += WAVE) {
    const int cnt = nt - base < WAVE ? nt - base : WAVE;
    int row = -1, skip = 1, plen = 0, pn[4] = {0, 0, 0, 0}, rf[CS_RFR];
    int64_t rq[CS_RFR];
    #pragma unroll
    for (int e = 0; e < CS_RFR; e++) { rf[e] = -1; rq[e] = 0; }
    if (lane < cnt) {
      row = trows[add ? nt - 1 - (base + lane) : base + lane];
      skip = restricted ? (k.preempted[row] != 0) : (k.preempted[row] != 1);  // Insert: rows this entry just marked (1); rows marked 3 were in the set already
      const AdmRec a = S.adm_rec[row];
      plen = (a.flags & 2u) ? KQ_MAXD + 1 : S.plen[a.cq];   // a wide row (more flavor-resources than the gathered record): the row walk below
      #pragma unroll
      for (int e = 0; e < CS_RFR; e++) { rf[e] = a.fr[e]; rq[e] = a.qty[e]; }
      #pragma unroll
      for (int i = 0; i < 4; i++) if (i < plen) pn[i] = S.path[(size_t)a.cq * KQ_MAXD + i];
    }
    for (int jj = 0; jj < cnt; jj++) {
      if (wbcast_u(skip, jj)) continue;
      const int pl = wbcast_u(plen, jj);
      if (pl > 4) {  // deeper than the gathered path, or a wide row: the row walk
        const int rw = wbcast_u(row, jj);
        if (restricted) np_apply_row_restricted(k, w, rw, add);
        else {
          if (lane == 0) {
            const int c = S.adm_cq[rw];
            for (int en = S.adm_use_off[rw]; en < S.adm_use_off[rw + 1]; en++) { UP g = up_plane(k, w, 1, S.adm_use_fr[en]); remove_usage(S, S.path + (size_t)c * KQ_MAXD, S.plen[c], S.adm_use_fr[en], S.adm_use_qty[en], g); }
          }
          wsync();
        }
        continue;
      }
      int n[4], f[CS_RFR];
      int64_t q[CS_RFR];
      #pragma unroll
      for (int i = 0; i < 4; i++) n[i] = wbcast_u(pn[i], jj);
      #pragma unroll
      for (int e = 0; e < CS_RFR; e++) { f[e] = wbcast_u(rf[e], jj); q[e] = (int64_t)(((uint64_t)(uint32_t)wbcast_u((int)(rq[e] >> 32), jj) << 32) | (uint32_t)wbcast_u((int)rq[e], jj)); }
      const int nl = restricted ? w.nuse : CS_RFR;
      for (int u = lane; u < nl; u += WAVE) {
        int fr; int64_t val = 0; bool hit = false;
        if (restricted) {
          fr = w.use_fr[u];
          #pragma unroll
          for (int e = 0; e < CS_RFR; e++) if (f[e] == fr) { val = q[e]; hit = true; }
        } else {
          fr = f[0]; val = q[0];
          if (u == 1) { fr = f[1]; val = q[1]; }
          if (u == 2) { fr = f[2]; val = q[2]; }
          if (u == 3) { fr = f[3]; val = q[3]; }
          hit = fr >= 0;
        }
        if (!hit) continue;
        UP g = up_plane(k, w, 1, fr);
        int64_t v[4], lq[4];
        #pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = 0; lq[i] = 0; if (i < pl) { v[i] = g.get(n[i]); lq[i] = local_quota(S, n[i], fr); } }
        bool go = true;
        #pragma unroll
        for (int i = 0; i < 4; i++) {  // addUsage / removeUsage resource_node.go:144-165
          if (!go || i >= pl) continue;
          const int64_t uu = v[i];
          if (add) {
            const int64_t la = i64max(0, a_sub(lq[i], uu));
            g.set(n[i], a_add(uu, val));
            if (i + 1 < pl && val > la) val = a_sub(val, la); else go = false;
          } else {
            const int64_t stored = a_sub(uu, lq[i]);
            g.set(n[i], a_sub(uu, val));
            if (stored <= 0 || i + 1 >= pl) go = false; else val = i64min(val, stored);
          }
        }
      }
    }
  }
  wsync();
}
// Columns of usage_np := usage_work with every row marked in k.preempted removed, ascending row
// (SimulateWorkloadUsageRemoval snapshot.go:80-100 over the canonical order). `use_only`: the entry's own
// flavor-resources (w.use_fr), else every broken column. Marked rows are found with a ballot scan over the tree's
// rows in ascending order; each selected column is owned by one lane (columns are independent).
KQ_DEV void np_rebuild(const K& k, Wave& w, int tree, bool use_only) {
  const DSnap& S = k.S;
  const int lane = lane_id();
  const int n0 = S.tree_node_off[tree], nn = S.tree_node_off[tree + 1] - n0;
  const int r0 = S.tree_row_off[tree], nrows = S.tree_row_off[tree + 1] - r0;
  const int ncol = use_only ? w.nuse : S.nfr;
  // copy the selected columns
  for (int ci = 0; ci < ncol; ci++) {
    const int fr = use_only ? w.use_fr[ci] : ci;
    if (!use_only && !col_broken(w, fr)) continue;
    UP a = up_plane(k, w, 0, fr), b = up_plane(k, w, 1, fr);
    for (int i = lane; i < nn; i += WAVE) { const int node = S.tree_nodes[n0 + i]; b.set(node, a.get(node)); }
  }
  wsync();
  for (int base = 0; base < nrows; base += WAVE) {
    const int i = base + lane;
    const int row = i < nrows ? S.tree_rows_asc[r0 + i] : -1;
    uint64_t m = wballot(row >= 0 && k.preempted[row] != 0);
    while (m) {
      const int bsrc = ffs64(m);
      m &= m - 1;
      const int prow = wbcast(row, bsrc);
      const int c = S.adm_cq[prow];
      for (int ci = lane; ci < ncol; ci += WAVE) {
        const int fr = use_only ? w.use_fr[ci] : ci;
        if (!use_only && !col_broken(w, fr)) continue;
        UP g = up_plane(k, w, 1, fr);
        for (int e = S.adm_use_off[prow]; e < S.adm_use_off[prow + 1]; e++)
          if (S.adm_use_fr[e] == fr) remove_usage(S, S.path + (size_t)c * KQ_MAXD, S.plen[c], fr, S.adm_use_qty[e], g);
      }
    }
  }
  wsync();
}
// scheduler.fits (scheduler.go:771-777) against usage_np (= snapshot minus PreemptedWorkloads)
KQ_DEV bool entry_fits(const K& k, Wave& w, const int32_t* trows, int nt, bool quota_usage, int tree) {
  const DSnap& S = k.S;
  if (!quota_usage || w.nuse == 0) return true;
  bool exact = false;
  if (w.np_broken && (w.n_pre > 0 || nt > 0))
    for (int u = 0; u < w.nuse; u++) if (col_broken(w, w.use_fr[u])) exact = true;
  if (exact) {
    // exact order of the reference: preempted and new targets leave the snapshot together, ascending row
    if (lane_id() == 0) for (int t = 0; t < nt; t++) if (!k.preempted[trows[t]]) k.preempted[trows[t]] = 2;
    wsync();
    np_rebuild(k, w, tree, true);
    bool bad = false;
    for (int u = lane_id(); u < w.nuse; u += WAVE) {
      UP g = up_plane(k, w, 1, w.use_fr[u]);
      if (i64max(0, available_of(S, w.path, w.plen, w.use_fr[u], g)) < w.use_qty[u]) bad = true;
    }
    const bool ok = wballot(bad) == 0;
    wsync();
    if (lane_id() == 0) for (int t = 0; t < nt; t++) if (k.preempted[trows[t]] == 2) k.preempted[trows[t]] = 0;
    wsync();
    np_rebuild(k, w, tree, true);
    if (lane_id() == 0) w.bytes += (int64_t)w.nuse * 40 * w.plen;
    return ok;
  }
  np_apply_targets(k, w, trows, nt, false, true, tree);
  bool bad = false;
  for (int u = lane_id(); u < w.nuse; u += WAVE) {
    int64_t avail;
    if (!w.pc_on && w.plen <= GP_MAX) {   // the path's cells of every level in one round trip (available_of makes one per level)
      GPath gp;
      gpath_load(S, w.path, w.plen, w.use_fr[u], k.usage_np, gp);
      avail = gpath_available(gp, w.plen);
    } else {
      UP g = up_plane(k, w, 1, w.use_fr[u]);
      avail = available_of(S, w.path, w.plen, w.use_fr[u], g);
    }
    if (i64max(0, avail) < w.use_qty[u]) bad = true;
  }
  bool ok = wballot(bad) == 0;
  wsync();
  np_apply_targets(k, w, trows, nt, true, true, tree);
  if (lane_id() == 0) w.bytes += (int64_t)w.nuse * 40 * w.plen;
  return ok;
}
// cq.AddUsage on both planes (clusterqueue_snapshot.go:107)
KQ_DEV void entry_add_usage(const K& k, Wave& w, const int64_t* qty) {
  const DSnap& S = k.S;
  if (!w.pc_on && w.plen <= GP_MAX) {
    // planes in global memory, a short path: lanes = (usage entry, plane), and a lane fetches the cells of all its levels before it
    // walks them (add_usage makes a round trip per level, and the two planes one after the other)
    for (int it = lane_id(); it < 2 * w.nuse; it += WAVE) {
      const int u = it >> 1, plane = it & 1, fr = w.use_fr[u];
      int64_t* pl = plane == 0 ? k.usage_work : k.usage_np;
      int64_t uu[GP_MAX], lq[GP_MAX];
      #pragma unroll
      for (int i = 0; i < GP_MAX; i++) {
        const int n = w.path[i < w.plen ? i : 0];
        uu[i] = pl[ix(S, n, fr)];
        lq[i] = local_quota(S, n, fr);
      }
      int64_t val = qty[u];
      #pragma unroll
      for (int i = 0; i < GP_MAX; i++) {
        if (i >= w.plen) break;
        const int n = w.path[i];
        const int64_t la = i64max(0, a_sub(lq[i], uu[i]));
        pl[ix(S, n, fr)] = a_add(uu[i], val);
        if (k.cq_dirty && n < S.nq) k.cq_dirty[n] = 1;
        if (i + 1 < w.plen && val > la) val = a_sub(val, la); else break;
      }
    }
    if (lane_id() == 0) { w.bytes += (int64_t)w.nuse * 8 * w.plen; w.usage_dirty = 1; }
    wsync();
    return;
  }
  for (int u = lane_id(); u < w.nuse; u += WAVE) {
    int fr = w.use_fr[u];
    UP a = up_plane(k, w, 0, fr), b = up_plane(k, w, 1, fr);
    add_usage(S, w.path, w.plen, fr, qty[u], a);
    add_usage(S, w.path, w.plen, fr, qty[u], b);
  }
  if (lane_id() == 0) { w.bytes += (int64_t)w.nuse * 8 * w.plen; w.usage_dirty = 1; }
  wsync();
}

// quotaResourcesToReserve (scheduler.go:796-814) for a Preempt-mode entry without targets
KQ_DEV int64_t reserve_amount(int64_t usage, int64_t nominal, int64_t blv, int64_t cur, int borrowing) {
  if (borrowing > 0) return blv == KQ_NIL_LIMIT ? usage : i64min(usage, a_sub(a_add(nominal, blv), cur));
  return i64max(0, i64min(usage, a_sub(nominal, cur)));
}

KQ_DEV void write_entry_result(const K& k, Wave& w, int e, int status, int action, int rq, int skip, int mode) {
  const DOut& O = k.O;
  if (lane_id() == 0) {
    if (status != KQ_ST_NOT_NOMINATED && status != KQ_ST_ASSUMED && rq == KQ_RQ_GENERIC) rq = KQ_RQ_FAILED_AFTER_NOMINATION;  // scheduler.go:1167-1170
    O.status[e] = (uint8_t)status; O.action[e] = (uint8_t)action; O.requeue_reason[e] = (uint8_t)rq; O.skip[e] = (uint8_t)skip;
    O.mode[e] = (uint8_t)mode;
  }
}

// Entries without preemption targets (the bulk of every cycle): every (slot, path level) cell the entry
// can touch is gathered in ONE parallel round trip (cohort levels from LDS, the CQ level from HBM),
// fits / AddUsage run on the gathered copy in LDS, dirty cells are scattered back.
KQ_DEV void process_entry_fast(const K& k, Wave& w, int e) {
  const DSnap& S = k.S;
  const int lane = lane_id();
  KQ_T0();
  const bool quota_usage = !(w.hflags & KQ_HEAD_HAS_QUOTA_RESERVATION);
  const int mode = w.rep_mode;
  const int plen = w.plen, nuse = quota_usage ? w.nuse : 0, ncell = nuse * plen;
  // the reference runs scheduler.fits before looking at the mode (updateAssignmentIfNeeded :713-714)
  if (lane == 0 && nuse > 0) w.bytes += (int64_t)nuse * 40 * plen;
  if (mode == M_NOFIT) { write_entry_result(k, w, e, KQ_ST_NOT_NOMINATED, KQ_ACT_NONE, KQ_RQ_NOFIT, KQ_SKIP_NONE, mode); return; }
  const int total = w.pc_ncoh * S.nfr;
  for (int c = lane; c < ncell; c += WAVE) {
    int u = c / plen, i = c % plen, n = w.path[i], fr = w.use_fr[u];
    size_t o = ix(S, n, fr);
    int64_t sqv = S.sq[o], llv = S.ll[o];
    w.g_sq[c] = sqv; w.g_bl[c] = S.bl[o];
    w.g_lq[c] = llv != KQ_NIL_LIMIT ? i64max(0, a_sub(sqv, llv)) : 0;
    if (i > 0 && w.pc_on) {
      size_t l = (size_t)w.path_coh[i] * S.nfr + fr;
      w.g_uw[c] = w.pc_lds[l]; w.g_un[c] = w.pc_lds[(size_t)total + l];
    } else {
      w.g_uw[c] = k.usage_work[o]; w.g_un[c] = k.usage_np[o];
    }
    w.g_dirty[c] = 0;
  }
  wsync();
  KQ_TS(k, 4);
  // scheduler.fits on usage_np (no targets): Available(cq, fr) >= qty for every flavor-resource
  bool bad = false;
  for (int u = lane; u < nuse; u += WAVE) {
    const int b = u * plen;
    int64_t a = a_sub(w.g_sq[b + plen - 1], w.g_un[b + plen - 1]);
    for (int i = plen - 2; i >= 0; i--) {
      int64_t lq = w.g_lq[b + i], uu = w.g_un[b + i], blv = w.g_bl[b + i];
      if (blv != KQ_NIL_LIMIT) a = i64min(a_add(a_sub(a_sub(w.g_sq[b + i], lq), i64max(0, a_sub(uu, lq))), blv), a);
      a = a_add(i64max(0, a_sub(lq, uu)), a);
    }
    if (i64max(0, a) < w.use_qty[u]) bad = true;
  }
  const bool fits_ok = wballot(bad) == 0;
  KQ_TS(k, 5);
  int status = KQ_ST_NOT_NOMINATED, action = KQ_ACT_NONE, rq = KQ_RQ_GENERIC, skip = KQ_SKIP_NONE;
  bool add = false, reserve = false;
  if (mode == M_PREEMPT) {  // no targets: reserveCapacityForUnreclaimablePreempt :538-543
    rq = KQ_RQ_PREEMPTION_NO_CANDIDATES;
    bool can_always_reclaim = KQ_POL_RECLAIM(w.pol) == KQ_POLICY_ANY;
    reserve = add = !can_always_reclaim || (gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS) && (w.hflags & KQ_HEAD_IS_PREEMPTOR));
  } else if (!fits_ok) {
    status = KQ_ST_SKIPPED; skip = KQ_SKIP_NO_LONGER_FITS;
  } else {
    add = true; status = KQ_ST_ASSUMED; action = KQ_ACT_ADMIT;
  }
  if (add && nuse > 0) {
    for (int u = lane; u < nuse; u += WAVE) {
      const int b = u * plen;
      int64_t val = w.use_qty[u];
      if (reserve) val = reserve_amount(val, S.nominal[ix(S, w.cq, w.use_fr[u])], w.g_bl[b], w.g_uw[b], w.borrowing);
      if (val < 0) { mark_broken(w, w.use_fr[u]); if (k.cert_flags) k.cert_flags[S.tree_of[w.cq]] = 1; }
      if (!reserve && plen > 1 && k.root_margin)  // certificate: slack of the root term of Available (see K::root_margin)
        cert_margin_exact(k.root_margin + (size_t)S.tree_of[w.cq] * S.nfr + w.use_fr[u], &w.g_lq[b], &w.g_un[b], w.g_sq[b + plen - 1], plen, val,
                          k.cert_flags + S.tree_of[w.cq]);
      int64_t v = val;  // resource_node.go:144-152 on both planes
      for (int i = 0; i < plen; i++) {
        int64_t uu = w.g_uw[b + i], la = i64max(0, a_sub(w.g_lq[b + i], uu));
        w.g_uw[b + i] = a_add(uu, v); w.g_dirty[b + i] |= 1;
        if (i + 1 < plen && v > la) v = a_sub(v, la); else break;
      }
      v = val;
      for (int i = 0; i < plen; i++) {
        int64_t uu = w.g_un[b + i], la = i64max(0, a_sub(w.g_lq[b + i], uu));
        w.g_un[b + i] = a_add(uu, v); w.g_dirty[b + i] |= 2;
        if (i + 1 < plen && v > la) v = a_sub(v, la); else break;
      }
    }
    if (lane == 0) { w.bytes += (int64_t)nuse * 8 * plen; w.usage_dirty = 1; }
    wsync();
    for (int c = lane; c < ncell; c += WAVE) {
      uint8_t d = w.g_dirty[c];
      if (!d) continue;
      int u = c / plen, i = c % plen, fr = w.use_fr[u];
      if (i > 0 && w.pc_on) {
        size_t l = (size_t)w.path_coh[i] * S.nfr + fr;
        if (d & 1) w.pc_lds[l] = w.g_uw[c];
        if (d & 2) w.pc_lds[(size_t)total + l] = w.g_un[c];
      } else {
        size_t o = ix(S, w.path[i], fr);
        if (d & 1) k.usage_work[o] = w.g_uw[c];
        if (d & 2) k.usage_np[o] = w.g_un[c];
        if (w.path[i] < S.nq) k.cq_dirty[w.path[i]] = 1;
      }
    }
  }
  KQ_TS(k, 6);
  write_entry_result(k, w, e, status, action, rq, skip, mode);
  KQ_TS(k, 7);
}

// Generic path (preemption targets, overlap recomputation, oversize entries). Deliberately NOT inlined: it
// drags in the whole nominate machinery, and inlining it into the hot loop of k_process bloats the kernel
// past the instruction cache and forces SGPR spills in the serial core.
KQ_NOINLINE void process_entry(const K& k, Wave& w, int e, int pos, int slot, int tree) {
  const DSnap& S = k.S; const DOut& O = k.O;
  const int lane = lane_id();
  KQ_T0();
  load_head(k, w, e);
  KQ_TS(k, 0);
  if (lane == 0) {
    w.nuse = O.use_n[e];
    for (int u = 0; u < w.nuse; u++) { w.use_fr[u] = O.use_fr[(size_t)e * KQ_MAXU + u]; w.use_qty[u] = O.use_qty[(size_t)e * KQ_MAXU + u]; }
    w.borrowing = O.borrowing[e];
    w.rep_mode = O.nominated_mode[e];
    O.order[e] = pos;
    for (int i = 1; i < w.plen; i++) w.path_coh[i] = S.node_local[w.path[i]] - w.pc_ncq;
  }
  wsync();
  KQ_TS(k, 1);
  int nt = O.tgt_n[e];
  if (nt == 0 && w.nuse * w.plen <= CELLS && !np_exact_mode(w)) {
    process_entry_fast(k, w, e);
    KQ_TS(k, 2);
    if (lane == 0) atomic_add_i64(O.stat_bytes, (long long)w.bytes);
    wsync();
    KQ_TS(k, 3);
    return;
  }
  const bool quota_usage = !(w.hflags & KQ_HEAD_HAS_QUOTA_RESERVATION);  // netUsage scheduler.go:785-794
  // the generic path (targets, recomputation, oversize entries, exact np mode) is outside the sharding certificate unless it ends
  // up changing nothing
  if (w.rep_mode != M_NOFIT) cert_unverifiable(k, tree);
  const int32_t* trows = O.pool_row + O.tgt_pos[e];
  // (lanes over the targets: the serial form was one dependent global load per target on every lane)
  auto has_any = [&]() { bool a = false; for (int t = lane; t < nt; t += WAVE) if (k.preempted[trows[t]]) a = true; return wballot(a) != 0; };
  // updateAssignmentIfNeeded :707-769. The reference evaluates fits() first and throws the result away when the targets overlap (the
  // recomputation ends with its own fits(), :747): with up to several hundred nominated targets per entry that discarded evaluation was
  // 95 of the 620 ms of k_process at cfg 4c (profiles/r04a_prof_cfg4c.txt) — only its algorithmic bytes are charged then.
  const bool recompute = has_any() && gate(k, KQ_GATE_RECOMPUTE_ON_OVERLAP);
  bool fits_ok = true;
  if (!recompute) fits_ok = entry_fits(k, w, trows, nt, quota_usage, tree);
  else if (quota_usage && w.nuse > 0 && lane == 0) w.bytes += (int64_t)w.nuse * 40 * w.plen;
  KQ_TS(k, 34);  // generic path: first fits
  int mode = w.rep_mode;
  if (recompute) {
    // SimulateWorkloadRemoval(victimsOfOtherPreemptions) == evaluate on usage_np with those rows deleted.
    // The generic nominate code reads HBM planes: publish the LDS-resident cohort rows first.
    KQ_TS(k, 4);  // (KQ_PROF) everything of this entry before the recomputation
    if (np_exact_mode(w)) np_rebuild(k, w, tree, false);
    pc_flush(k, w, w.pc_lds, tree);
    KQ_TS(k, 5);  // flush of the LDS-resident cohort rows
    if (lane == 0) w.has_last = 0;
    // e.NominationMapping = e.readResourceToFlavorMapping() (scheduler.go:734): fixed for the whole recomputation
    for (int i = lane; i < w.nps * S.nR; i += WAVE) k.X.nom[(size_t)slot * KQ_MAXPS * S.nR + i] = O.flavor[(size_t)w.ps_base * S.nR + i];
    wsync();
    // the flushed rows are not read until the recomputation is over: their LDS serves the victim searches meanwhile (kq_cs.hpp)
    const bool lend = w.pc_region_bytes > 0;  // classical: the scan search's arrays (kq_cs.hpp); fair sharing: the search's state (kq_fs.hpp)
    if (lend && lane == 0) { w.cs_lds = (unsigned char*)w.pc_lds; w.cs_lds_bytes = w.pc_region_bytes; }
    wsync();
    Search s = get_assignments(k, w, slot, k.usage_np, k.preempted, true);
    publish_assignment(k, w, s, e);
    KQ_TS(k, 6);  // the recomputation itself
    if (lend) { if (lane == 0) { w.cs_lds = nullptr; w.cs_lds_bytes = 0; } wsync(); pc_load(k, w, w.pc_lds, tree); }
    KQ_TS(k, 7);  // reload of the rows
    trows = O.pool_row + O.tgt_pos[e];
    nt = O.tgt_n[e];
    mode = w.rep_mode;
    if (mode == M_FIT) {  // SetRepresentativeMode(DeferredFit) flavorassigner.go:109-114
      mode = M_DEFERRED;
      for (int i = lane; i < w.nps * S.nR; i += WAVE) {
        size_t o = (size_t)w.ps_base * S.nR + i;
        if (O.flavor[o] >= 0) O.res_mode[o] = M_DEFERRED;
      }
    }
    wsync();
    fits_ok = entry_fits(k, w, trows, nt, quota_usage, tree);
    KQ_TS(k, 32);  // fits after the recomputation
  }
  int status = KQ_ST_NOT_NOMINATED, action = KQ_ACT_NONE, rq = KQ_RQ_GENERIC, skip = KQ_SKIP_NONE;
  bool done = false;
  if (mode == M_NOFIT) { rq = KQ_RQ_NOFIT; done = true; }
  if (!done && mode == M_PREEMPT && nt == 0) {
    rq = KQ_RQ_PREEMPTION_NO_CANDIDATES;
    // reserveCapacityForUnreclaimablePreempt :538-543 ; quotaResourcesToReserve :796-814
    bool can_always_reclaim = KQ_POL_RECLAIM(w.pol) == KQ_POLICY_ANY;
    if ((!can_always_reclaim || (gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS) && (w.hflags & KQ_HEAD_IS_PREEMPTOR))) && quota_usage) {
      if (lane == 0)
        for (int u = 0; u < w.nuse; u++) {
          int fr = w.use_fr[u];
          w.s_qty[u] = reserve_amount(w.use_qty[u], S.nominal[ix(S, w.cq, fr)], S.bl[ix(S, w.cq, fr)], k.usage_work[ix(S, w.cq, fr)], w.borrowing);
          if (w.s_qty[u] < 0) mark_broken(w, fr);
        }
      wsync();
      entry_add_usage(k, w, w.s_qty);
    }
    done = true;
  }
  if (!done && mode == M_DEFERRED) {
    rq = KQ_RQ_PENDING_PREEMPTION;
    if (quota_usage) entry_add_usage(k, w, w.use_qty);
    done = true;
  }
  if (!done && has_any()) { status = KQ_ST_SKIPPED; skip = KQ_SKIP_OVERLAP; done = true; }
  if (!done && !fits_ok) { status = KQ_ST_SKIPPED; skip = KQ_SKIP_NO_LONGER_FITS; done = true; }
  if (!done) {
    // preemptedWorkloads.Insert(targets): rows leave usage_np for the rest of the cycle
    {  // rows already in the set are marked 3 for the time of the removal pass, new ones 1
      int fresh = 0;
      for (int base = 0; base < nt; base += WAVE) {
        const int t = base + lane;
        bool isnew = false;
        if (t < nt) {
          const int row = trows[t];
          if (!k.preempted[row]) {
            isnew = true;
            k.preempted[row] = 1;
            atomic_add_i32(&k.cq_rm_bytes[S.adm_cq[row]], 32 + 12 * (S.adm_use_off[row + 1] - S.adm_use_off[row]));
          } else if (k.preempted[row] == 1) k.preempted[row] = 3;
        }
        fresh += popc64(wballot(isnew));
      }
      if (lane == 0) w.n_pre += fresh;
      wsync();
      np_apply_targets(k, w, trows, nt, false, false, tree);
      for (int t = lane; t < nt; t += WAVE) if (k.preempted[trows[t]] == 3) k.preempted[trows[t]] = 1;
      wsync();
    }
    if (quota_usage) entry_add_usage(k, w, w.use_qty);
    if (mode == M_PREEMPT) { action = KQ_ACT_PREEMPT; rq = KQ_RQ_PENDING_PREEMPTION; }
    else { status = KQ_ST_ASSUMED; action = KQ_ACT_ADMIT; }
  }
  write_entry_result(k, w, e, status, action, rq, skip, mode);
  if (lane == 0) atomic_add_i64(O.stat_bytes, (long long)w.bytes);
  wsync();
  KQ_TS(k, 33);  // generic path: everything after the (optional) recomputation
}

// ---- chunked fast path --------------------------------------------------------------------------
// A lone wave is latency-bound: every dependent HBM/L2 access costs ~1-2k cycles and nothing hides it.
// So entries are handled CH at a time: (1) all lanes prefetch the CH entries' records and the static
// quota constants of every (slot, path level) cell they can touch into LDS (latency paid once per
// chunk, in parallel), (2) a serial core walks the chunk touching only LDS (cohort rows are resident
// there for the whole kernel), (3) results are written back in parallel.
constexpr int FU = 8;    // max flavor-resources of an entry on the fast path
constexpr int FD = 4;    // max path length (CQ + 3 cohort levels) on the fast path
constexpr int CH = 16;   // entries per chunk; two chunk buffers live in LDS (one being processed, one being prefetched)
constexpr int NBUF = 2;
constexpr int64_t PLAIN_LIMIT = (int64_t)1 << 56;  // see "serial core" below
constexpr int64_t SP_SMALL = (int64_t)1 << 44;      // kq_spec.hpp: requests below it => prefix sums over <= 2^12 items stay below 2^56
constexpr int64_t SP_C_NOLIMIT = (int64_t)1 << 60, SP_T_INF = (int64_t)1 << 59;  // kq_spec.hpp: "this level never binds" / "everything stays local"
constexpr int64_t QC_NOLIMIT = (int64_t)1 << 61;   // "no borrowing limit at this level" in PRec::ccv (sums of plain values stay below it)
struct alignas(16) PRec {
  // ---- static part: a function of the head's nomination and the quota constants only. Written once per cycle for every head
  // (rec_fill_static, after k_nominate), copied into LDS by the prefetch of k_process.
  int32_t e, cq, plen, nuse, borrowing, mode, slow_static, pad0;
  uint32_t pol, flags;
  int32_t pad1[2];
  int32_t fr[FU];
  int64_t qty[FU], nominal[FU];
  int64_t lq[FU][FD], sqv[FU][FD];
  // quad core: Available(cq) = min over the path levels j of  sum_{k<j} max(0, lq_k - usage_k) + ccv_j - usage_j  with
  // ccv = subtree quota (+ borrowing limit below the root; QC_NOLIMIT without one)  — see core_run_quad
  int64_t ccv[FU][FD];
  // cohort levels (i >= 1): index of the (node, flavor-resource) cell inside a plane of the tree's resident rows
  int32_t uoff[FU][FD];
  uint8_t cbig[FU][FD];    // one of the cell's constants is not a plain quantity
  // the ClusterQueue-level usage cells, both planes, AS THEY WERE AT THE START OF THE CYCLE (the serial core updates the LDS
  // copy). Valid for an entry as long as K::cq_dirty of its ClusterQueue is clear; otherwise the entry takes the generic path.
  int64_t uw0[FU], un0[FU];
  // ---- dynamic part: written by the prefetch (pos .. pre) and by the serial core (the rest)
  int32_t pos, slow;
  union { uint8_t b[FU]; uint64_t all; } pre;  // per slot: "already did not fit when the record was fetched" (see chunk_prefetch)
  uint8_t status, action, rq, skip, omode, dirty, added, pad;  // dirty: CQ-level usage cells not yet in HBM; added: AddUsage ran
};
constexpr size_t PREC_STATIC = offsetof(PRec, pos);
static_assert(offsetof(PRec, status) % 4 == 0 && offsetof(PRec, action) == offsetof(PRec, status) + 1 && offsetof(PRec, rq) == offsetof(PRec, status) + 2 &&
              offsetof(PRec, skip) == offsetof(PRec, status) + 3, "core_fit_quad writes status/action/rq/skip as one word");
static_assert(PREC_STATIC % 16 == 0 && sizeof(PRec) % 16 == 0, "the static part is copied 16 bytes at a time");
static_assert(offsetof(PRec, un0) - offsetof(PRec, uw0) == FU * sizeof(int64_t), "quad core: un0[u] is FU cells after uw0[u]");

// Static part of head e's record, one call per (slot, level) cell c = u * FD + i; the call with c == 0 also writes the header.
// Runs for every head between k_nominate and k_process (k_records), so that the prefetch inside k_process is two dependent
// memory round trips (entry -> record, record -> ClusterQueue usage cells) instead of walking head -> ClusterQueue -> path ->
// quota cells for every chunk while the serial core waits.
// kq_pending_step: the head / podset counts the gather left on the device travel to the host inside the packed decisions (the free
// word behind the byte counters), so the step needs no copy of its own for them. One call per cycle (k_records).
KQ_DEV void pack_counts(const K& k) {
  if (!k.H.n_dev) return;
  int32_t* m = k.O.pool_count + 6;   // misc[3] of the pack: [pool_count | error][nominate bytes][process bytes][n_heads | n_podsets]
  m[0] = k.H.n_dev[0]; m[1] = k.H.n_dev[1];
}
KQ_DEV void rec_fill_static(const K& k, int e, int c) {
  const DSnap& S = k.S; const DOut& O = k.O; const DHeads& H = k.H;
  PRec& r = k.grec[e];
  const int u = c / FD, i = c % FD;
  const int cq = H.cq[e];
  const uint32_t flags = H.flags[e];
  const int plen = S.plen[cq];
  const int nuse = (flags & KQ_HEAD_HAS_QUOTA_RESERVATION) ? 0 : O.use_n[e];  // netUsage scheduler.go:785-794
  if (c == 0) {
    r.e = e; r.cq = cq; r.flags = flags; r.pol = S.cq_policy[cq];
    r.plen = plen; r.borrowing = O.borrowing[e]; r.mode = O.nominated_mode[e]; r.nuse = nuse;
    r.slow_static = (O.tgt_n[e] != 0 || nuse > FU || plen > FD) ? 1 : 0;
    k.cq_dirty[cq] = 0;
    if (k.cq_heads) atomic_add_i32(&k.cq_heads[cq], 1);
  }
  if (u >= nuse || nuse > FU || i >= plen || plen > FD) return;
  const int fr = O.use_fr[(size_t)e * KQ_MAXU + u];
  const int n = S.path[(size_t)cq * KQ_MAXD + i];
  const size_t o = ix(S, n, fr);
  const int64_t sqv = S.sq[o], llv = S.ll[o];
  const int64_t blv = S.bl[o], lqv = llv != KQ_NIL_LIMIT ? i64max(0, a_sub(sqv, llv)) : 0;
  r.sqv[u][i] = sqv; r.lq[u][i] = lqv;
  r.ccv[u][i] = i == plen - 1 ? sqv : (blv != KQ_NIL_LIMIT ? (int64_t)((uint64_t)sqv + (uint64_t)blv) : QC_NOLIMIT);
  uint64_t big = (uint64_t)sqv | (uint64_t)lqv | (blv != KQ_NIL_LIMIT ? (uint64_t)blv : 0ull);
  // what the speculative rounds (kq_spec.hpp) cannot take: a constant that is neither plain nor exactly Unlimited (saturating
  // arithmetic would matter), a usage cell outside the plain range, a request too large to sum thousands of
  auto odd = [](int64_t v) { return (uint64_t)v >= (uint64_t)PLAIN_LIMIT && v != I64MAX; };
  const int64_t ucell = k.usage[o];
  bool unsup = odd(sqv) || odd(lqv) || (blv != KQ_NIL_LIMIT && odd(blv)) || (uint64_t)ucell >= (uint64_t)PLAIN_LIMIT;
  if (i == 0) {
    const int64_t qty = O.use_qty[(size_t)e * KQ_MAXU + u], nominal = S.nominal[o];
    r.fr[u] = fr; r.qty[u] = qty; r.nominal[u] = nominal;
    big |= (uint64_t)qty | (uint64_t)nominal;
    unsup = unsup || (uint64_t)qty >= (uint64_t)SP_SMALL;
    r.uoff[u][0] = 0;
    r.uw0[u] = r.un0[u] = ucell;  // both work planes start the cycle as copies of the snapshot's usage
  } else {
    const int tree = S.tree_of[cq];
    const int ncq = S.tree_cq_off[tree + 1] - S.tree_cq_off[tree];
    r.uoff[u][i] = (S.node_local[n] - ncq) * S.nfr + fr;
  }
  // ---- constants of the speculative rounds (kq_spec.hpp), against the usage at the start of the cycle. Unlimited constants
  // (resources.Amount: absorbing): with finite usage, a level whose SubtreeQuota, localQuota or borrowing limit is Unlimited never binds
  // (its term of Available is Unlimited), and an Unlimited localQuota keeps everything local (LocalAvailable Unlimited, nothing is
  // passed up, resource_node.go:92-152) — both are the closed form with a large constant.
  int spf = 0;
  if (k.spec_K) {
    const int mode = O.nominated_mode[e];
    const uint32_t pol = S.cq_policy[cq];
    const size_t o0 = ix(S, cq, fr);
    const int64_t u0 = k.usage[o0], qty = O.use_qty[(size_t)e * KQ_MAXU + u];
    int64_t val = qty;
    if (mode == M_PREEMPT) {  // quotaResourcesToReserve scheduler.go:796-814 (only used when the entry reserves, see sp_chunk_classify)
      const int64_t nominal = S.nominal[o0], bl0 = S.bl[o0];
      const bool bl_inf = bl0 == KQ_NIL_LIMIT || bl0 == I64MAX;
      if (O.borrowing[e] > 0) {
        if (bl_inf || nominal == I64MAX) val = qty;
        else if ((uint64_t)nominal >= (uint64_t)PLAIN_LIMIT || (uint64_t)bl0 >= (uint64_t)PLAIN_LIMIT) unsup = true;
        else val = i64min(qty, (nominal + bl0) - u0);
      } else if (nominal == I64MAX) val = i64max(0, qty);
      else if ((uint64_t)nominal >= (uint64_t)PLAIN_LIMIT) unsup = true;
      else val = i64max(0, i64min(qty, nominal - u0));
      (void)pol;
    }
    const int64_t sq0 = S.sq[o0], ll0 = S.ll[o0];
    const int64_t lq0 = ll0 != KQ_NIL_LIMIT ? i64max(0, a_sub(sq0, ll0)) : 0;
    const int64_t E0 = lq0 == I64MAX ? SP_T_INF : i64max(0, lq0 - u0);
    auto limit = [&](int64_t sq_, int64_t lq_, int64_t bl_, bool root) {
      return (sq_ == I64MAX || lq_ == I64MAX || (!root && (bl_ == KQ_NIL_LIMIT || bl_ == I64MAX))) ? SP_C_NOLIMIT : (root ? sq_ : sq_ + bl_);
    };
    if (!unsup) {
      if (i == 0) {
        k.spec_push[(size_t)e * FU + u] = i64max(0, val - E0);
        k.spec_o[((size_t)e * FU + u) * FD] = (int32_t)o;
        k.spec_nv[(size_t)e * FU + u] = u0 + val;
        const int64_t c0 = limit(sqv, lqv, blv, plen == 1);
        if (mode == M_FIT && qty > 0 && c0 - u0 < qty) spf |= 16;   // level-0 term of Available fails (static: one head per ClusterQueue)
        if (val < 0) spf |= 32;                                       // negative reservation (scheduler.go:806)
      } else {
        // (a request of 0 always fits — max(0, Available) < 0 is false — whatever the cell holds: never tested)
        const int64_t cc = (mode == M_FIT && qty <= 0) ? SP_C_NOLIMIT : limit(sqv, lqv, blv, i == plen - 1);
        const int64_t Tv = lqv == I64MAX ? SP_T_INF : lqv - ucell;
        const size_t ci = ((size_t)e * FU + u) * FD + i;
        k.spec_K[ci] = cc - ucell - val + E0; k.spec_T[ci] = Tv; k.spec_o[ci] = (int32_t)o;
        // the prefix P of a cell is >= 0: a level's term can only bind with a finite limit (and only Fit entries are tested); local
        // quota is only left while T > 0. A depth where neither holds for any item needs no scan.
        if (cc != SP_C_NOLIMIT && mode == M_FIT) spf |= 4;
        if (Tv > 0) spf |= 8;
      }
    }
  }
  // bit 0: not a plain quantity (serial core -> exact Amount arithmetic); bit 1: outside what kq_spec.hpp handles; bits 2-5: see above
  r.cbig[u][i] = (big >= (uint64_t)PLAIN_LIMIT ? 1 : 0) | (unsup ? 2 : 0) | spf;
}

// Fills the LDS records of one chunk from the static records, in three independent parts. No synchronisation inside: every
// thread starts from the entry list and the global records, never from what another thread of this call wrote, so any subset
// of the workgroup's threads can run it while wave 0 is busy with the previous chunk — and the three parts can run on
// different waves at the same time (chunk_prefetch): what matters is the LATENCY of one call (the serial core waits for it at
// the end of every chunk), i.e. the number of dependent memory round trips, not the amount of work.
typedef uint32_t V16 __attribute__((vector_size(16)));  // a native 16-byte vector: arrays of it stay in registers
// (1) static parts: 16 bytes per thread and step; all loads of a thread are issued before its first store
KQ_DEV void prefetch_copy(const K& k, PRec* rec, const int32_t* ent, int nch, int t, int nt) {
  constexpr int W16 = (int)(PREC_STATIC / 16), STAGE = 20;  // 16 records over one wave: 16 * 79 / 64 = 19.75 words per lane
  const PRec* grec = k.grec;
  const int total = nch * W16;
  for (int base = t; base < total; base += nt * STAGE) {
    V16 tmp[STAGE];
    #pragma unroll
    for (int m = 0; m < STAGE; m++) {  // unconditional (index clamped): the loads stay independent and `tmp` stays in registers
      const int idx = base + m * nt < total ? base + m * nt : total - 1;
      tmp[m] = ((const V16*)&grec[ent[idx / W16]])[idx % W16];
    }
    #pragma unroll
    for (int m = 0; m < STAGE; m++) {
      const int idx = base + m * nt;
      if (idx < total) ((V16*)&rec[idx / W16])[idx % W16] = tmp[m];
    }
  }
}
// the entries of a chunk: indices into the heads, their ClusterQueues, their positions in the order (base + offset)
struct EntList { const int32_t* e; const int32_t* cq; const uint8_t* off; int posbase; };
// (2) per (record, slot): the slot's early "no longer fits" (scheduler.go:1167); the lane of slot 0 also decides whether the
// entry may use the serial core at all (position, slow flag).
// Screening: between two generic-path entries the serial core only ever ADDS usage, and Available (resource_node.go:106-122)
// does not increase when usage grows, so a slot that does not fit against the usage visible now (cohort rows as they are in
// LDS at this moment, possibly while wave 0 is still adding to them; the ClusterQueue's own cells as recorded at the start of
// the cycle, only used while K::cq_dirty says nothing wrote them) cannot fit when its turn comes. The serial core then skips
// the arithmetic for that entry. The core ends the chunk (and drops the records fetched ahead) whenever it subtracts: a
// negative reservation (:806). Same closed form as core_run_quad.
// One global round trip (the record's fields and the dirty flag), then LDS reads.
// prevcq / nprev: the ClusterQueues of the chunk that is still being processed — its CQ-level cells are neither in HBM nor
// flagged dirty yet, so an entry of the same ClusterQueue takes the generic path; same for an earlier entry of this chunk.
KQ_DEV void prefetch_cells(const K& k, const Wave& w, PRec* rec, const EntList& L, int nch, const int32_t* prevcq, int nprev, int t, int nt) {
  const DSnap& S = k.S;
  const PRec* grec = k.grec;
  const int64_t* np_rows = w.pc_lds + (size_t)w.pc_ncoh * S.nfr;
  const bool rows = w.pc_on != 0;
  for (int idx = t; idx < nch * FU; idx += nt) {
    const int j = idx / FU, u = idx % FU;
    PRec& r = rec[j];
    const PRec& g = grec[L.e[j]];
    const int cq = L.cq[j];
    // everything from the record and the flag first: independent loads
    const int nuse = g.nuse, plen = g.plen, mode = g.mode, slow_static = g.slow_static;
    const int dirty = k.cq_dirty[cq];
    const int64_t qty = g.qty[u], un0 = g.un0[u];
    int64_t ccv[FD], lq[FD]; int uoff[FD]; uint8_t cbig[FD];
    #pragma unroll
    for (int i = 0; i < FD; i++) { ccv[i] = g.ccv[u][i]; lq[i] = g.lq[u][i]; uoff[i] = g.uoff[u][i]; cbig[i] = g.cbig[u][i]; }
    const bool on = u < nuse && nuse <= FU && plen <= FD;
    const bool screen = on && (plen == 1 || rows) && mode != M_PREEMPT && mode != M_NOFIT;
    int64_t un[FD];
    un[0] = un0;
    #pragma unroll
    for (int i = 1; i < FD; i++) un[i] = (screen && i < plen) ? np_rows[uoff[i]] : 0;
    if (u == 0) {
      int slow = (slow_static || dirty || (plen > 1 && !rows)) ? 1 : 0;
      for (int q = 0; q < j; q++) if (L.cq[q] == cq) slow = 1;
      for (int q = 0; q < nprev; q++) if (prevcq[q] == cq) slow = 1;
      r.pos = L.posbase + L.off[j]; r.slow = slow; r.dirty = 0; r.added = 0;
    }
    bool bad = false;
    if (screen) {
      int64_t E = 0, a = QC_NOLIMIT;
      bool plain = (uint64_t)qty < (uint64_t)PLAIN_LIMIT;
      #pragma unroll
      for (int i = 0; i < FD; i++) {
        if (i >= plen) continue;
        plain = plain && (cbig[i] & 1) == 0 && (uint64_t)un[i] < (uint64_t)PLAIN_LIMIT;
        const int64_t x = (int64_t)((uint64_t)E + (uint64_t)ccv[i] - (uint64_t)un[i]);  // garbage (never UB) when not plain
        a = x < a ? x : a;
        E = (int64_t)((uint64_t)E + (uint64_t)i64max(0, (int64_t)((uint64_t)lq[i] - (uint64_t)un[i])));
      }
      bad = plain && i64max(0, a) < qty;  // not plain: the sums could have wrapped, no claim
    }
    r.pre.b[u] = bad ? 1 : 0;
  }
}
KQ_DEV void chunk_prefetch(const K& k, const Wave& w, PRec* rec, const EntList& L, int nch, const int32_t* prevcq, int nprev, int tid, int nthreads) {
  if (nthreads >= 3 * WAVE && nthreads % WAVE == 0) {
    // two waves on the per-slot tasks (one task per lane), the rest on the copy: one global round trip each
    constexpr int NCELL = 2 * WAVE;
    if (tid < NCELL) prefetch_cells(k, w, rec, L, nch, prevcq, nprev, tid, NCELL);
    else prefetch_copy(k, rec, L.e, nch, tid - NCELL, nthreads - NCELL);
    return;
  }
  prefetch_copy(k, rec, L.e, nch, tid, nthreads);
  prefetch_cells(k, w, rec, L, nch, prevcq, nprev, tid, nthreads);
}
KQ_DEV void chunk_prefetch_wave(const K& k, Wave& w, PRec* rec, const EntList& L, int nch) {
  chunk_prefetch(k, w, rec, L, nch, nullptr, 0, lane_id(), WAVE);
  wsync();
}

// CQ-level cells of the chunk's first n entries that added usage go back to HBM, one lane per (entry, slot)
// Pointers the per-chunk write-out needs, read from the argument block once per tree: read through `k` where they are used
// they cost a global round trip per chunk (the compiler must assume the stores in between may have changed the block).
struct ProcPtrs { uint8_t *status, *action, *rq, *skip, *mode; int32_t* order; int64_t *uw, *un; uint8_t* dirty; int nfr; };
KQ_DEV ProcPtrs proc_ptrs(const K& k) {
  return ProcPtrs{k.O.status, k.O.action, k.O.requeue_reason, k.O.skip, k.O.mode, k.O.order, k.usage_work, k.usage_np, k.cq_dirty, k.S.nfr};
}
KQ_DEV void chunk_scatter(const ProcPtrs& P, PRec* rec, int n) {
  for (int idx = lane_id(); idx < n * FU; idx += WAVE) {
    PRec& r = rec[idx / FU];
    const int u = idx % FU;
    if (r.slow || !r.dirty || u >= r.nuse) continue;
    const size_t o = (size_t)r.cq * P.nfr + r.fr[u];
    P.uw[o] = r.uw0[u]; P.un[o] = r.un0[u];
    P.dirty[r.cq] = 1;
  }
  wsync();
  // written back exactly once: a later generic-path entry may update the same ClusterQueue's cells in HBM
  for (int q = lane_id(); q < n; q += WAVE) rec[q].dirty = 0;
  wsync();
}
KQ_DEV void chunk_scatter(const K& k, PRec* rec, int n) { chunk_scatter(proc_ptrs(k), rec, n); }

// ---- serial core ------------------------------------------------------------------------------------
// One lane per flavor-resource slot, everything in registers, straight-line code specialised on the path
// length. It only handles the PLAIN case: every operand (quota constants, both usage planes, the request) is
// a bounded non-negative quantity < 2^56, so the saturation / Unlimited branches of resources.Amount
// (amount.go:114-145) cannot trigger and plain 64-bit add/sub/min/max are bit-identical to them. Anything
// else (Unlimited quota cells, over-large values) returns false and takes the generic exact path.

// Scalars the serial core needs from the kernel argument block. K lives in global memory and the compiler cannot prove the
// core's stores do not alias it, so reading them through `k` costs a global load (+ wait) per entry; they are read once
// per tree instead.
struct CoreCtx { int nfr, total, dbg; bool prio_preemptors; const int64_t* bl_tab; const int32_t* path_tab; long long* margin; int32_t* cert_flag; };
KQ_DEV CoreCtx core_ctx(const K& k, const Wave& w, int tree) {
  CoreCtx c; c.nfr = k.S.nfr; c.total = w.pc_ncoh * k.S.nfr; c.prio_preemptors = gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS);
  c.bl_tab = k.S.bl; c.path_tab = k.S.path;
  c.margin = k.root_margin ? k.root_margin + (size_t)tree * k.S.nfr : nullptr; c.cert_flag = k.cert_flags ? k.cert_flags + tree : nullptr;
#ifdef KQ_PROF
  c.dbg = k.C.dbg_variant;  // timing experiments (tools/prof_process.py): results are wrong when non-zero
#else
  c.dbg = 0;
#endif
  return c;
}
// algorithmic bytes of a record the serial core handled: scheduler.fits reads nuse * 40 * plen, AddUsage writes nuse * 8 * plen
KQ_DEV int64_t rec_algo_bytes(const PRec& r) {
  return r.nuse > 0 ? (int64_t)r.nuse * r.plen * (40 + (r.added ? 8 : 0)) : 0;
}

// What the serial core reads from a record before it touches a usage cell. None of it is written while a chunk is walked, so
// the leader loads it one entry ahead (the loads are in flight while the previous entry's arithmetic runs).
struct QPre {
  int slow, plen, nuse, mode, borrowing;
  uint32_t pol, flags;
  uint64_t pre;
#ifndef KQ_HOST_EMU
  int64_t lq, ccv, qty;  // this lane's (slot, level) cell: lane = plane * 32 + slot * 4 + level
  int uoff, cbig;
#endif
};
KQ_DEV QPre rec_preload(const PRec& r) {
  QPre q;
  q.slow = r.slow; q.plen = r.plen; q.nuse = r.nuse; q.mode = r.mode; q.borrowing = r.borrowing; q.pol = r.pol; q.flags = r.flags;
  q.pre = r.pre.all;
#ifndef KQ_HOST_EMU
  const int lane = lane_id(), u = (lane >> 2) & 7, i = lane & 3;
  q.lq = r.lq[u][i]; q.ccv = r.ccv[u][i]; q.qty = r.qty[u];
  q.uoff = r.uoff[u][i]; q.cbig = r.cbig[u][i] & 1;
#endif
  return q;
}

template <int PLEN> struct SlotState { int64_t un[PLEN], uw[PLEN], lq[PLEN], sq[PLEN], bl[PLEN], qty, nominal; int cidx[PLEN]; };

// `pcl` is the LDS base of the cohort rows, passed down explicitly (NOT re-read from Wave::pc_lds) so that after
// inlining the compiler still knows the address space and emits ds_* instead of flat_* in the serial core.
template <int PLEN> KQ_DEV void slot_load(const CoreCtx& cc, const int64_t* pcl, const PRec& r, int u, SlotState<PLEN>& x) {
  const int nfr = cc.nfr, total = cc.total, fr = r.fr[u];
  x.qty = r.qty[u]; x.nominal = r.nominal[u];
  #pragma unroll
  for (int i = 0; i < PLEN; i++) {
    x.lq[i] = r.lq[u][i]; x.sq[i] = r.sqv[u][i];
    x.bl[i] = cc.bl_tab[(size_t)cc.path_tab[(size_t)r.cq * KQ_MAXD + i] * nfr + fr];  // not kept in the record (the quad core reads ccv)
    x.cidx[i] = i == 0 ? 0 : r.uoff[u][i];
  }
  x.uw[0] = r.uw0[u]; x.un[0] = r.un0[u];
  #pragma unroll
  for (int i = 1; i < PLEN; i++) { x.uw[i] = pcl[x.cidx[i]]; x.un[i] = pcl[total + x.cidx[i]]; }
}

template <bool PLAIN> KQ_DEV int64_t q_add(int64_t a, int64_t b) { if (PLAIN) return a + b; return a_add(a, b); }
template <bool PLAIN> KQ_DEV int64_t q_sub(int64_t a, int64_t b) { if (PLAIN) return a - b; return a_sub(a, b); }

// PLAIN = true : returns false (and does nothing) when some operand is not plain.
// PLAIN = false: exact resources.Amount arithmetic, always succeeds.
template <int PLEN, bool PLAIN> KQ_DEV bool core_run(Wave& w, int64_t* pcl, PRec& r, const CoreCtx& cc) {
  const int lane = lane_id();
  const int nuse = r.nuse, mode = r.mode;
  const int total = cc.total;
  SlotState<PLEN> x;
  bool notplain = false, bad = false;
  // one pass on the device (nuse <= FU < 64 lanes); the 1-lane emulation walks the slots one by one
  for (int u = lane; u < nuse; u += WAVE) {
    slot_load<PLEN>(cc, pcl, r, u, x);
    uint64_t big = (uint64_t)x.qty | (uint64_t)x.nominal;
    #pragma unroll
    for (int i = 0; i < PLEN; i++) big |= (uint64_t)x.un[i] | (uint64_t)x.uw[i] | (uint64_t)x.lq[i] | (uint64_t)x.sq[i] | (x.bl[i] == KQ_NIL_LIMIT ? 0ull : (uint64_t)x.bl[i]);
    if (PLAIN && big >= (uint64_t)PLAIN_LIMIT) { notplain = true; continue; }  // negatives and Unlimited land here too
    // scheduler.fits: Available(cq, fr) on usage_np, root first (resource_node.go:106-122)
    int64_t a = q_sub<PLAIN>(x.sq[PLEN - 1], x.un[PLEN - 1]);
    #pragma unroll
    for (int i = PLEN - 2; i >= 0; i--) {
      const int64_t wm = q_add<PLAIN>(q_sub<PLAIN>(q_sub<PLAIN>(x.sq[i], x.lq[i]), i64max(0, q_sub<PLAIN>(x.un[i], x.lq[i]))), x.bl[i]);
      a = (x.bl[i] != KQ_NIL_LIMIT && wm < a) ? wm : a;
      a = q_add<PLAIN>(i64max(0, q_sub<PLAIN>(x.lq[i], x.un[i])), a);
    }
    if (i64max(0, a) < x.qty) bad = true;
  }
  if (PLAIN && wballot(notplain) != 0) return false;
  const bool fits_ok = wballot(bad) == 0;
  int status = KQ_ST_NOT_NOMINATED, action = KQ_ACT_NONE, rq = KQ_RQ_GENERIC, skip = KQ_SKIP_NONE;
  bool add = false, reserve = false;
  if (mode == M_PREEMPT) {  // no targets: reserveCapacityForUnreclaimablePreempt scheduler.go:538-543
    rq = KQ_RQ_PREEMPTION_NO_CANDIDATES;
    const bool can_always_reclaim = KQ_POL_RECLAIM(r.pol) == KQ_POLICY_ANY;
    reserve = add = !can_always_reclaim || (cc.prio_preemptors && (r.flags & KQ_HEAD_IS_PREEMPTOR));
  } else if (!fits_ok) {
    status = KQ_ST_SKIPPED; skip = KQ_SKIP_NO_LONGER_FITS; rq = KQ_RQ_FAILED_AFTER_NOMINATION;  // scheduler.go:1167-1170
  } else {
    add = true; status = KQ_ST_ASSUMED; action = KQ_ACT_ADMIT;
  }
  if (add) {
    for (int u = lane; u < nuse; u += WAVE) {
      if (WAVE < FU) slot_load<PLEN>(cc, pcl, r, u, x);  // device: registers of the first pass are still live
      int64_t val = x.qty;
      if (!reserve && PLEN > 1 && cc.margin) cert_margin_exact(cc.margin + r.fr[u], x.lq, x.un, x.sq[PLEN - 1], PLEN, val, cc.cert_flag);  // K::root_margin
      if (reserve) {  // quotaResourcesToReserve scheduler.go:796-814
        if (r.borrowing > 0) val = x.bl[0] == KQ_NIL_LIMIT ? x.qty : i64min(x.qty, q_sub<PLAIN>(q_add<PLAIN>(x.nominal, x.bl[0]), x.uw[0]));
        else val = i64max(0, i64min(x.qty, q_sub<PLAIN>(x.nominal, x.uw[0])));
        if (val < 0) { mark_broken(w, r.fr[u]); if (cc.cert_flag) *cc.cert_flag = 1; }
      }
      // addUsage resource_node.go:144-152 on usage_work, then on usage_np
      int64_t v = val;
      bool go = true;
      #pragma unroll
      for (int i = 0; i < PLEN; i++) {
        const int64_t la = i64max(0, q_sub<PLAIN>(x.lq[i], x.uw[i]));
        if (go) { if (i == 0) r.uw0[u] = q_add<PLAIN>(x.uw[0], v); else pcl[x.cidx[i]] = q_add<PLAIN>(x.uw[i], v); }
        go = go && (i + 1 < PLEN) && v > la;
        v = q_sub<PLAIN>(v, la);
      }
      v = val; go = true;
      #pragma unroll
      for (int i = 0; i < PLEN; i++) {
        const int64_t la = i64max(0, q_sub<PLAIN>(x.lq[i], x.un[i]));
        if (go) { if (i == 0) r.un0[u] = q_add<PLAIN>(x.un[0], v); else pcl[total + x.cidx[i]] = q_add<PLAIN>(x.un[i], v); }
        go = go && (i + 1 < PLEN) && v > la;
        v = q_sub<PLAIN>(v, la);
      }
    }
    if (lane == 0) { r.dirty = 1; r.added = 1; }
    wsync_lds();  // cohort rows in LDS must be visible to the next entry's lanes
  }
  if (lane == 0) { r.status = (uint8_t)status; r.action = (uint8_t)action; r.rq = (uint8_t)rq; r.skip = (uint8_t)skip; r.omode = (uint8_t)mode; }
  return true;
}

#ifndef KQ_HOST_EMU
// ---- quad core (device only) ------------------------------------------------------------------------------
// One lane per (usage plane, flavor-resource slot, path level): lane = plane * 32 + slot * 4 + level, so the levels of a slot
// are one DPP quad and both planes run the same instructions. Plain operands only (anything else -> false, exact core), which
// lets the two recurrences of the reference be rewritten without a level-by-level chain. With u_k the usage of path level k
// (0 = ClusterQueue), t_k = max(0, localQuota_k - u_k) and E_j = t_0 + .. + t_{j-1}:
//  * available (resource_node.go:106-122):  a_root = sq - u,  a_k = t_k + min(a_{k+1}, (sq_k - lq_k) - max(0, u_k - lq_k) + bl_k).
//    t_k - max(0, u_k - lq_k) = lq_k - u_k, so the second operand of the min is sq_k + bl_k - u_k - ... unrolled:
//    Available(cq) = min_j ( E_j + ccv_j - u_j ),  ccv_j = sq_j + bl_j below the root (no limit: QC_NOLIMIT), sq_j at the root.
//  * addUsage (:144-152): level j receives val - E_j, and only if val > E_j (level 0 always).
// E is three independent quad_perm reads of t; the min is a two-step butterfly.
template <int CTRL> KQ_DEV int64_t dpp64(int64_t v) {
  const int lo = __builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp((int)(v >> 32), CTRL, 0xf, 0xf, false);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
static_assert(FD == 4 && FU == 8, "the quad core maps (plane, slot, level) onto the 64 lanes: lane = plane * 32 + slot * 4 + level");
KQ_DEV bool core_run_quad(Wave& w, int64_t* pcl, PRec& r, const CoreCtx& cc, const QPre& q) {
  const int lane = lane_id();
  const int plane = lane >> 5, u = (lane >> 2) & 7, i = lane & 3;  // plane 0: usage_work, plane 1: usage_np
  const int plen = q.plen, nuse = q.nuse, mode = q.mode;
  const int64_t qty = q.qty;
  const bool act = u < nuse && i < plen;
  // the only dependent LDS round trip: this lane's usage cell in its plane (un0 follows uw0 in the record; plane 1 of the rows is `total` further)
  const int ua = i == 0 ? (int)(&r.uw0[u] - pcl) + (plane ? FU : 0) : q.uoff + (plane ? cc.total : 0);
  const int64_t cur = act ? pcl[ua] : 0;
  if (wballot(act && (q.cbig != 0 || (uint64_t)cur >= (uint64_t)PLAIN_LIMIT)) != 0) return false;  // negatives and Unlimited land here too
  const int64_t t = i64max(0, q.lq - cur);
  const int64_t t1 = dpp64<0x90>(t), t2 = dpp64<0x40>(t), t3 = dpp64<0x00>(t);  // t of level i-1, i-2, i-3 (where they exist)
  const int64_t E = (i >= 1 ? t1 : 0) + (i >= 2 ? t2 : 0) + (i >= 3 ? t3 : 0);
  // scheduler.fits: Available(cq, fr), meaningful in plane 1
  int64_t a = act ? E + q.ccv - cur : QC_NOLIMIT;
  { const int64_t o = dpp64<0xB1>(a); a = o < a ? o : a; }
  { const int64_t o = dpp64<0x4E>(a); a = o < a ? o : a; }
  const bool bad = act && plane == 1 && i == 0 && i64max(0, a) < qty;
  const bool fits_ok = wballot(bad) == 0;
  int status = KQ_ST_NOT_NOMINATED, action = KQ_ACT_NONE, rq = KQ_RQ_GENERIC, skip = KQ_SKIP_NONE;
  bool add = false, reserve = false;
  if (mode == M_PREEMPT) {  // no targets: reserveCapacityForUnreclaimablePreempt scheduler.go:538-543
    rq = KQ_RQ_PREEMPTION_NO_CANDIDATES;
    const bool can_always_reclaim = KQ_POL_RECLAIM(q.pol) == KQ_POLICY_ANY;
    reserve = add = !can_always_reclaim || (cc.prio_preemptors && (q.flags & KQ_HEAD_IS_PREEMPTOR));
  } else if (!fits_ok) {
    status = KQ_ST_SKIPPED; skip = KQ_SKIP_NO_LONGER_FITS; rq = KQ_RQ_FAILED_AFTER_NOMINATION;  // scheduler.go:1167-1170
  } else {
    add = true; status = KQ_ST_ASSUMED; action = KQ_ACT_ADMIT;
  }
  if (add && cc.dbg != 2) {
    int64_t val = qty;
    if (reserve && act) {  // quotaResourcesToReserve scheduler.go:796-814: every lane of the slot derives the same value
      const int64_t uw0 = r.uw0[u], nominal = r.nominal[u];
      const int64_t bl0 = cc.bl_tab[(size_t)r.cq * cc.nfr + r.fr[u]];
      if (q.borrowing > 0) val = bl0 == KQ_NIL_LIMIT ? qty : i64min(qty, (nominal + bl0) - uw0);
      else val = i64max(0, i64min(qty, nominal - uw0));
      if (i == 0 && plane == 0 && val < 0) mark_broken(w, r.fr[u]);
    }
    if (!reserve && cc.margin && act && plane == 1 && plen > 1 && i == plen - 1) cert_min(cc.margin + r.fr[u], (E + q.ccv - cur) - qty);  // K::root_margin
    if (reserve && val < 0 && cc.cert_flag && act && i == 0 && plane == 0) *cc.cert_flag = 1;
    // addUsage resource_node.go:144-152, both planes at once
    if (act && (i == 0 || val > E)) pcl[ua] = cur + (val - E);
    if (lane == 0) { r.dirty = 1; r.added = 1; }
    wsync_lds();  // cohort rows in LDS must be visible to the next entry's lanes
  }
  if (lane == 0) { r.status = (uint8_t)status; r.action = (uint8_t)action; r.rq = (uint8_t)rq; r.skip = (uint8_t)skip; r.omode = (uint8_t)mode; }
  return true;
}
// The serial loop over the Fit-class records of a chunk, as a function of its own: inside process_tree the loop shares
// registers with everything that kernel inlines, and the shuffling between entries (accumulator-register and lane spills) cost
// more than the arithmetic. Here the live state is a work mask, one record's lane constants and the next one's.
// Visits the set bits of `work` in ascending order while they are not in `special`; returns the mask that is left — its
// lowest bit, if any, is a record this loop cannot handle (generic path, reservation, or operands that are not plain).
#define KQ_LDS __attribute__((address_space(3)))
KQ_DEV uint64_t uniform_u64(uint64_t v) {  // the same in every lane: keep it in scalar registers
  return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32;
}
__device__ __noinline__ uint64_t fit_run(int64_t* pcl_generic, PRec* rc_generic, uint64_t work_v, uint64_t special_v, int total_v, long long* margin) {
  uint64_t work = uniform_u64(work_v);
  const uint64_t special = uniform_u64(special_v);
  const int total = __builtin_amdgcn_readfirstlane(total_v);
  KQ_LDS int64_t* const pcl = (KQ_LDS int64_t*)pcl_generic;
  KQ_LDS PRec* const rc = (KQ_LDS PRec*)rc_generic;
  const int lane = lane_id();
  const int plane = lane >> 5, u = (lane >> 2) & 7, i = lane & 3;  // plane 0: usage_work, plane 1: usage_np
  struct LC { int plen, nuse, mode, uoff, cbig, fr; int64_t lq, ccv, qty; };
  auto load = [&](int j) {
    KQ_LDS PRec& r = rc[j];
    LC x;
    x.plen = r.plen; x.nuse = r.nuse; x.mode = r.mode; x.uoff = r.uoff[u][i]; x.cbig = r.cbig[u][i] & 1; x.fr = r.fr[u];
    x.lq = r.lq[u][i]; x.ccv = r.ccv[u][i]; x.qty = r.qty[u];
    return x;
  };
  if (!work || ((special >> ffs64(work)) & 1)) return work;
  LC q = load(ffs64(work));
  for (;;) {
    const int cj = ffs64(work);
    const uint64_t rest = work & (work - 1);
    const bool more = rest != 0 && !((special >> ffs64(rest)) & 1);
    const LC qn = load(more ? ffs64(rest) : cj);  // in flight during this record's arithmetic
    KQ_LDS PRec& r = rc[cj];
    const bool act = u < q.nuse && i < q.plen;
    // this lane's usage cell in its plane: un0 follows uw0 in the record; plane 1 of the resident rows is `total` further
    KQ_LDS int64_t* const cell = i == 0 ? &r.uw0[u] + (plane ? FU : 0) : pcl + q.uoff + (plane ? total : 0);
    const int64_t cur = act ? *cell : 0;
    if (__builtin_expect(wballot(act && (q.cbig != 0 || (uint64_t)cur >= (uint64_t)PLAIN_LIMIT)) != 0, 0)) return work;
    const int64_t t = i64max(0, q.lq - cur);
    const int64_t t1 = dpp64<0x90>(t), t2 = dpp64<0x40>(t), t3 = dpp64<0x00>(t);
    const int64_t E = (i >= 1 ? t1 : 0) + (i >= 2 ? t2 : 0) + (i >= 3 ? t3 : 0);
    int64_t a = act ? E + q.ccv - cur : QC_NOLIMIT;
    { const int64_t o = dpp64<0xB1>(a); a = o < a ? o : a; }
    { const int64_t o = dpp64<0x4E>(a); a = o < a ? o : a; }
    const bool fits_ok = wballot(act && plane == 1 && i == 0 && i64max(0, a) < q.qty) == 0;
    // sharding certificate: the slack of the root term (K::root_margin); a fire-and-forget atomic, nothing waits for it
    if (margin && fits_ok && act && plane == 1 && q.plen > 1 && i == q.plen - 1) atomicMin(margin + q.fr, (long long)((E + q.ccv - cur) - q.qty));
    // addUsage on both planes when it fits (scheduler.go:1167-1175), predicated
    if (fits_ok && act && (i == 0 || q.qty > E)) *cell = cur + (q.qty - E);
    if (lane == 0) {
      const uint32_t res = fits_ok ? ((uint32_t)KQ_ST_ASSUMED | (uint32_t)KQ_ACT_ADMIT << 8 | (uint32_t)KQ_RQ_GENERIC << 16 | (uint32_t)KQ_SKIP_NONE << 24)
                                   : ((uint32_t)KQ_ST_SKIPPED | (uint32_t)KQ_ACT_NONE << 8 | (uint32_t)KQ_RQ_FAILED_AFTER_NOMINATION << 16 | (uint32_t)KQ_SKIP_NO_LONGER_FITS << 24);
      *(KQ_LDS uint32_t*)&r.status = res;
      r.omode = (uint8_t)q.mode; r.dirty = fits_ok ? 1 : 0; r.added = fits_ok ? 1 : 0;
    }
    wsync_lds();  // cohort rows in LDS must be visible to the next record's lanes
    work = rest;
    if (!more) return work;
    q = qn;
  }
}
#endif

// serial core dispatch for one fast entry; false => the entry needs the generic exact path
KQ_DEV bool chunk_entry_fast(Wave& w, int64_t* pcl, PRec& r, const CoreCtx& cc, const QPre& q) {
  const int plen = q.plen, nuse = q.nuse, mode = q.mode;
  if (mode == M_NOFIT || nuse == 0) {
    // scheduler.fits still runs before the mode is looked at (updateAssignmentIfNeeded :713-714): rec_bytes counts it
    int status = KQ_ST_NOT_NOMINATED, action = KQ_ACT_NONE, rq = KQ_RQ_GENERIC;
    if (mode == M_NOFIT) rq = KQ_RQ_NOFIT;
    else if (mode == M_PREEMPT) rq = KQ_RQ_PREEMPTION_NO_CANDIDATES;
    else { status = KQ_ST_ASSUMED; action = KQ_ACT_ADMIT; }  // no quota usage: fits trivially
    if (lane_id() == 0) { r.status = (uint8_t)status; r.action = (uint8_t)action; r.rq = (uint8_t)rq; r.skip = KQ_SKIP_NONE; r.omode = (uint8_t)mode; }
    return true;
  }
  if ((mode != M_PREEMPT && q.pre != 0) || cc.dbg == 1) {  // did not fit when fetched => does not fit now (scheduler.go:1167-1170)
    if (lane_id() == 0) { r.status = KQ_ST_SKIPPED; r.action = KQ_ACT_NONE; r.rq = KQ_RQ_FAILED_AFTER_NOMINATION; r.skip = KQ_SKIP_NO_LONGER_FITS; r.omode = (uint8_t)mode; }
    return true;
  }
#ifndef KQ_HOST_EMU
  if (core_run_quad(w, pcl, r, cc, q)) return true;
  switch (plen) {  // not plain: exact Amount arithmetic, one lane per slot
    case 1: core_run<1, false>(w, pcl, r, cc); break;
    case 2: core_run<2, false>(w, pcl, r, cc); break;
    case 3: core_run<3, false>(w, pcl, r, cc); break;
    default: core_run<4, false>(w, pcl, r, cc); break;
  }
  return true;
#endif
  switch (plen) {
    case 1: if (!core_run<1, true>(w, pcl, r, cc)) core_run<1, false>(w, pcl, r, cc); break;
    case 2: if (!core_run<2, true>(w, pcl, r, cc)) core_run<2, false>(w, pcl, r, cc); break;
    case 3: if (!core_run<3, true>(w, pcl, r, cc)) core_run<3, false>(w, pcl, r, cc); break;
    default: if (!core_run<4, true>(w, pcl, r, cc)) core_run<4, false>(w, pcl, r, cc); break;
  }
  return true;
}

// one wave per root-cohort tree: drain the tree's entries in iterator order.
// lds layout: [2 planes of the tree's cohort rows][CH records]; too small for the rows => they stay in HBM.
// tid / nthreads: every wave of the workgroup helps with the parallel parts (record prefetch); wave 0 ("leader") runs the
// serial core and the generic path. Workgroup-uniform control flow comes from LDS scalars read after a bsync().
KQ_DEV void process_tree(const K& k, Wave& w, int tree, int slot, int64_t* lds, size_t lds_bytes, int tid, int nthreads) {
  const DSnap& S = k.S; const DOut& O = k.O;
  const int n = hn(k.H);
  const int lane = lane_id();
  const bool leader = tid < WAVE;
  const size_t rec_bytes = sizeof(PRec) * CH * NBUF;
  if (tid == 0) {
    w.pc_ncq = S.tree_cq_off[tree + 1] - S.tree_cq_off[tree];
    w.pc_ncoh = (S.tree_node_off[tree + 1] - S.tree_node_off[tree]) - w.pc_ncq;
    w.pc_lds = lds;
    w.pc_on = (w.pc_ncoh > 0 && lds_bytes >= rec_bytes && (size_t)w.pc_ncoh * S.nfr * 16 <= lds_bytes - rec_bytes) ? 1 : 0;
    w.pc_region_bytes = w.pc_on ? (int)((size_t)w.pc_ncoh * S.nfr * 16) : 0;
    w.np_broken = 0; w.n_pre = 0; w.broken[0] = w.broken[1] = w.broken[2] = w.broken[3] = 0;
    w.nwin2[0] = w.nwin2[1] = 0; w.chunk_done = 0; w.chunk_stop = 0; w.mono_break = 0;
  }
  bsync();
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  const long long _tree0 = clock64(), _wall0 = wall_clock64();
#endif
  const CoreCtx cc = core_ctx(k, w, tree);
  const ProcPtrs P = proc_ptrs(k);
  const bool chunked = lds_bytes >= rec_bytes;
  PRec* rec = (PRec*)((unsigned char*)lds + (lds_bytes - (chunked ? rec_bytes : 0)));
  bool loaded = false;
  int64_t bytes = 0;  // per-lane partial sum (the result pass below adds what its records cost)
  constexpr int WIN = 256;  // entries of the global order examined per window (independent of the wave width)
  // entries in front of `resume` were decided by the speculative rounds (kq_spec.hpp); their usage is in the planes already
  const int resume = k.spec_resume ? k.spec_resume[tree] : 0;
  for (int base = resume < n ? (resume / WIN) * WIN : n, par = 0; base < n; base += WIN, par ^= 1) {
    // leader: compact this window's entries of the tree (in order) into LDS lists
    if (leader) {
      int nwin = 0;
      for (int off = 0; off < WIN; off += WAVE) {
        const int i = base + off + lane;
        bool mine = false;
        int e = 0, ecq = 0;
        if (i < n && i >= resume) { e = k.order_idx[i]; ecq = k.H.cq[e]; mine = S.tree_of[ecq] == tree; }
        const uint64_t m = wballot(mine);
        const int my = nwin + popc64(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        if (mine) { w.win_e[my] = e; w.win_cq[my] = ecq; w.win_off[my] = (uint8_t)(off + lane); }
        nwin += popc64(m);
      }
      if (lane == 0) w.nwin2[par] = nwin;
      wsync();
    }
    bsync();
    const int nwin = w.nwin2[par];
    if (nwin == 0) continue;
    if (!loaded) {  // the whole workgroup copies the rows in; the record prefetch below already reads them
      KQ_T0(); pc_load_by(k, w, lds, tree, tid, nthreads); bsync(); if (leader) KQ_TS(k, 8);
      loaded = true;
    }
    if (!chunked) {
      if (leader) for (int q = 0; q < nwin; q++) process_entry(k, w, w.win_e[q], base + w.win_off[q], slot, tree);
      continue;
    }
    // Chunks of CH entries, double buffered: while wave 0 walks chunk c (serial core), the other waves fetch the records of
    // chunk c + 1. A chunk that stops early (generic path taken: it may touch HBM cells other records were fetched from)
    // discards the prefetched buffer and restarts behind the entry it stopped at.
#ifdef KQ_HOST_EMU
    // single-threaded emulation of the pipeline: the "helper" prefetch of chunk c + 1 runs BEFORE chunk c is processed,
    // i.e. the earliest it could ever run on the device
    const bool helpers = nthreads > WAVE || g_emu_pipeline;
#else
    const bool helpers = nthreads > WAVE;
#endif
    int done = 0, cur = 0;
    {
      KQ_T0();
      const int n0 = nwin < CH ? nwin : CH;
      chunk_prefetch(k, w, rec, EntList{w.win_e, w.win_cq, w.win_off, base}, n0, nullptr, 0, tid, nthreads);
      bsync();
      if (leader) KQ_TS(k, 10);
    }
    while (done < nwin) {
      const int nch = (nwin - done) < CH ? (nwin - done) : CH;
      const int next0 = done + nch;
      const int nnext = (nwin - next0) < CH ? (nwin - next0) : CH;
      PRec* rc = rec + cur * CH;
      PRec* rn = rec + (cur ^ 1) * CH;
#ifdef KQ_HOST_EMU
      if (g_emu_pipeline && nnext > 0) chunk_prefetch(k, w, rn, EntList{w.win_e + next0, w.win_cq + next0, w.win_off + next0, base}, nnext, w.win_cq + done, nch, 0, 1);
#endif
      if (leader) {
        KQ_T0();
        int j = 0;
        bool stopped = false;
        // Classify the chunk's records first (one lane per record). Records that need no arithmetic — nothing to reserve, or
        // screened at fetch time — get their result here, in parallel; the serial loop below only visits the others.
        const bool exact_now = np_exact_mode(w);  // can only switch on inside the chunk through mark_broken => mono_break => stop
        uint64_t work = 0, slowm = 0, resm = 0;
        for (int b = 0; b < nch; b += WAVE) {
          const int jj = b + lane;
          int cls = 0;  // 0: done here, 1: generic path, 2: serial core (Fit), 3: serial core (reservation for a preemptor)
          if (jj < nch) {
            PRec& r = rc[jj];
            const int mode = r.mode, nuse = r.nuse;
            if (r.slow || exact_now) cls = 1;
            else if (mode == M_NOFIT || nuse == 0) {
              // scheduler.fits still runs before the mode is looked at (updateAssignmentIfNeeded :713-714): rec_algo_bytes counts it
              int status = KQ_ST_NOT_NOMINATED, action = KQ_ACT_NONE, rq = KQ_RQ_GENERIC;
              if (mode == M_NOFIT) rq = KQ_RQ_NOFIT;
              else if (mode == M_PREEMPT) rq = KQ_RQ_PREEMPTION_NO_CANDIDATES;
              else { status = KQ_ST_ASSUMED; action = KQ_ACT_ADMIT; }  // no quota usage: fits trivially
              r.status = (uint8_t)status; r.action = (uint8_t)action; r.rq = (uint8_t)rq; r.skip = KQ_SKIP_NONE; r.omode = (uint8_t)mode;
            } else if ((mode != M_PREEMPT && r.pre.all != 0) || cc.dbg == 1) {  // did not fit when fetched => does not fit now (:1167-1170)
              r.status = KQ_ST_SKIPPED; r.action = KQ_ACT_NONE; r.rq = KQ_RQ_FAILED_AFTER_NOMINATION; r.skip = KQ_SKIP_NO_LONGER_FITS; r.omode = (uint8_t)mode;
            } else cls = mode == M_PREEMPT ? 3 : 2;
          }
          work |= wballot(cls != 0) << b; slowm |= wballot(cls == 1) << b; resm |= wballot(cls == 3) << b;
        }
        wsync_lds();
        j = nch;
        while (work) {
#ifndef KQ_HOST_EMU
          work = fit_run(lds, rc, work, slowm | resm, cc.total, cc.margin);  // the Fit-class records up to the next special one
          if (!work) break;
#endif
          const int cj = ffs64(work);
          work &= work - 1;
          PRec& r = rc[cj];
          bool generic = ((slowm >> cj) & 1) != 0;
          // reservation for a preemptor, or a Fit-class record fit_run handed back (operands not plain): the general core
          if (!generic) generic = !chunk_entry_fast(w, lds, r, cc, rec_preload(r));  // (the exact core always succeeds today)
          if (generic) {
            // generic path (targets / recompute / oversize / usage_np no longer incremental). It reads and writes HBM for
            // CQ-level cells, so prefetched CQ-level values may be stale afterwards: stop the chunk behind it.
            chunk_scatter(P, rc, cj);  // earlier fast entries' CQ-level cells must be in HBM first
            wsync();
            process_entry(k, w, r.e, r.pos, slot, tree);
            if (lane == 0) r.slow = 2;
            wsync();
            j = cj + 1;
            stopped = true;
            break;
          }
          if (((resm >> cj) & 1) && w.mono_break) {
            // a negative reservation (scheduler.go:806) lowered usage: the records fetched so far were screened against
            // more usage than there is now; fetch the rest again
            j = cj + 1;
            stopped = true;
            break;
          }
        }
        if (stopped) { wsync(); if (lane == 0) w.mono_break = 0; }
        KQ_TS(k, 11);
        wsync();
        for (int q = lane; q < j; q += WAVE) {
          const PRec& r = rc[q];
          if (r.slow) continue;  // the generic path wrote its own result
          P.status[r.e] = r.status; P.action[r.e] = r.action; P.rq[r.e] = r.rq; P.skip[r.e] = r.skip; P.mode[r.e] = r.omode;
          P.order[r.e] = r.pos;
          bytes += rec_algo_bytes(r);
        }
        chunk_scatter(P, rc, j);
        if (lane == 0) { w.chunk_done = j; w.chunk_stop = stopped ? 1 : 0; }
        wsync();
        KQ_TS(k, 12);
      } else if (nnext > 0) {
        chunk_prefetch(k, w, rn, EntList{w.win_e + next0, w.win_cq + next0, w.win_off + next0, base}, nnext, w.win_cq + done, nch, tid - WAVE, nthreads - WAVE);
      }
      { KQ_T0(); bsync(); if (leader) KQ_TS(k, 13); }
      const int j = w.chunk_done;
      const bool clean = j == nch && !w.chunk_stop;
      if (clean && helpers) {
        done = next0; cur ^= 1;  // the next chunk's records are ready
      } else {
        done += j;
        if (done < nwin) {
          KQ_T0();
          const int nre = (nwin - done) < CH ? (nwin - done) : CH;
          chunk_prefetch(k, w, rc, EntList{w.win_e + done, w.win_cq + done, w.win_off + done, base}, nre, nullptr, 0, tid, nthreads);
          bsync();
          if (leader) { KQ_TS(k, 27); KQ_TS(k, 28); }
        }
      }
    }
  }
  if (loaded) { KQ_T0(); pc_flush_by(k, w, lds, tree, tid, nthreads); if (leader) KQ_TS(k, 9); }  // after the loop's last bsync
  if (!leader) return;
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
  if (lane == 0) { atomic_add_i64((long long*)k.prof + 14, clock64() - _tree0); atomic_add_i64((long long*)k.prof + 29, wall_clock64() - _wall0); }
#endif
  bytes = wsum_i64(bytes);
  if (lane == 0 && bytes) atomic_add_i64(O.stat_bytes, (long long)bytes);
}

}  // namespace kq
#include "kq_spec.hpp"
namespace kq {

// ------------------------------------------------------------------------------------------------
// fair-sharing iterator (fair_sharing_iterator.go) — per root-cohort tree: pop = computeDRS + runTournament,
// then processEntry on the winner; the next pop sees the usage that entry added (scheduler.go:358-359).
// ------------------------------------------------------------------------------------------------
// usage_work with one entry's assignment usage added at its CQ (SimulateUsageAddition :251), evaluated
// functionally so that lanes can look at different entries at once: the amount reaching path[level] is
// what AddUsage would have bubbled there (resource_node.go:144-152).
struct PE {
  const K* k; const Wave* w; const int32_t* path; int level; const int32_t* ufr; const int64_t* uqty; int nu;
  KQ_MDEV int64_t get(int node, int fr) const {
    const DSnap& S = k->S;
    int64_t base = up_plane(*k, *w, 0, fr).get(node);
    for (int q = 0; q < nu; q++) {
      if (ufr[q] != fr) continue;
      int64_t delta = uqty[q];
      bool reach = true;
      for (int l = 0; l < level; l++) {
        int n = path[l];
        int64_t la = i64max(0, a_sub(local_quota(S, n, fr), up_plane(*k, *w, 0, fr).get(n)));
        if (delta > la) delta = a_sub(delta, la); else { reach = false; break; }
      }
      if (reach) base = a_add(base, delta);
    }
    return base;
  }
};
struct PW { const K* k; const Wave* w; KQ_MDEV int64_t get(int node, int fr) const { return up_plane(*k, *w, 0, fr).get(node); } };
// DRS of path[level] with the entry's usage added, from the node's borrowed sums: only the entry's own
// flavor-resources can differ from the cached sums. Exact when C.fs_plain (no saturating step can trigger).
KQ_DEV DRSv drs_entry_level(const K& k, const Wave& w, const int32_t* path, int level, const int32_t* ufr, const int64_t* uqty, int nu,
                            bool want_bon, int64_t* bytes) {
  const DSnap& S = k.S;
  const int n = path[level], nR = S.nR;
  int64_t sum[KQ_MAXR];
  #pragma unroll
  for (int r = 0; r < KQ_MAXR; r++) sum[r] = r < nR ? k.X.bs_sum[(size_t)n * nR + r] : 0;
  int pos = k.X.bs_pos[n];
  int bon = 0;
  for (int q = 0; q < nu; q++) {
    const int fr = ufr[q];
    const size_t o = ix(S, n, fr);
    if (!(S.qflags[o] & KQ_QF_SUBTREE)) continue;
    int64_t delta = uqty[q];
    for (int l = 0; l < level; l++) {  // what AddUsage bubbles up to this level (resource_node.go:144-152)
      int m = path[l];
      int64_t la = i64max(0, a_sub(local_quota(S, m, fr), up_plane(k, w, 0, fr).get(m)));
      if (delta > la) delta = a_sub(delta, la); else { delta = 0; break; }
    }
    const int64_t u = up_plane(k, w, 0, fr).get(n), sqv = S.sq[o];
    const int64_t ob = i64max(0, a_sub(u, sqv)), nb = i64max(0, a_sub(a_add(u, delta), sqv));
    const int r = fr % nR;
    #pragma unroll
    for (int rr = 0; rr < KQ_MAXR; rr++) if (rr == r) sum[rr] += nb - ob;
    pos += (nb > 0 ? 1 : 0) - (ob > 0 ? 1 : 0);
    if (want_bon && nb > 0 && uqty[q] > 0) bon = 1;
  }
  DRSv d = drs_from_sums(S, n, sum, pos, bytes);
  d.borrow_on = bon;
  return d;
}
// drs_entry_level for a path of at most four levels whose nodes are known (nd: global ids; row: the cohort's row in the LDS-resident
// planes, -1 = read HBM). The walk above costs one dependent round trip per (usage entry, level) — ~40 k cycles for a ClusterQueue that
// shares its cohort with the popped entry, and the leader waits for the slowest thread of the pass. Here every operand of four usage
// entries is requested before the first one is used: one round trip per four entries. Same arithmetic, same order.
KQ_DEV DRSv drs_entry_level_g(const K& k, const Wave& w, const int* nd, const int* row, int level, const int32_t* ufr, const int64_t* uqty, int nu,
                              bool want_bon, int64_t* bytes, const int* fr0 = nullptr, const int64_t* qty0 = nullptr) {
  const DSnap& S = k.S;
  const int nR = S.nR, n = fs_sel4(nd, level), rown = fs_sel4(row, level);
  auto usage_of = [&](int node, int rw, int fr) -> int64_t {
    return rw >= 0 ? w.pc_lds[(size_t)rw * S.nfr + fr] : k.usage_work[(size_t)node * S.nfr + fr];   // plane 0 of the resident rows = usage_work
  };
  int64_t sum[KQ_MAXR], lend[KQ_MAXR];
  #pragma unroll
  for (int r = 0; r < KQ_MAXR; r++) { sum[r] = r < nR ? k.X.bs_sum[(size_t)n * nR + r] : 0; lend[r] = r < nR ? S.lendable[(size_t)n * nR + r] : 0; }
  int pos = k.X.bs_pos[n];
  // (what drs_from_sums reads of the node: requested with the cells, not behind them)
  const double weight = S.fair_weight[n];
  const int frcount = S.frcount[n], depth = S.depth[n];
  int bon = 0;
  for (int q0 = 0; q0 < nu; q0 += 4) {
    int fr[4]; int64_t qty[4], sqn[4], un[4], lq[4][3], ul[4][3]; uint8_t qf[4]; bool in[4];
    #pragma unroll
    for (int j = 0; j < 4; j++) {
      in[j] = q0 + j < nu;
      if (q0 == 0 && fr0) { fr[j] = in[j] ? fr0[j] : fr0[0]; qty[j] = in[j] ? qty0[j] : qty0[0]; }   // the caller fetched the first four with its other operands (entries past nu hold anything)
      else { fr[j] = ufr[in[j] ? q0 + j : q0]; qty[j] = uqty[in[j] ? q0 + j : q0]; }
    }
    #pragma unroll
    for (int j = 0; j < 4; j++) {
      const size_t o = ix(S, n, fr[j]);
      qf[j] = S.qflags[o]; sqn[j] = S.sq[o]; un[j] = usage_of(n, rown, fr[j]);
      #pragma unroll
      for (int l = 0; l < 3; l++) {
        lq[j][l] = 0; ul[j][l] = 0;
        if (l < level) {
          const size_t om = ix(S, nd[l], fr[j]);
          const int64_t ll = S.ll[om], sqm = S.sq[om];
          lq[j][l] = ll != KQ_NIL_LIMIT ? i64max(0, a_sub(sqm, ll)) : 0;   // local_quota
          ul[j][l] = usage_of(nd[l], row[l], fr[j]);
        }
      }
    }
    #pragma unroll
    for (int j = 0; j < 4; j++) {
      if (!in[j] || !(qf[j] & KQ_QF_SUBTREE)) continue;
      int64_t delta = qty[j];
      #pragma unroll
      for (int l = 0; l < 3; l++) {  // what AddUsage bubbles up to this level (resource_node.go:144-152)
        if (l >= level || delta == 0) continue;
        const int64_t la = i64max(0, a_sub(lq[j][l], ul[j][l]));
        if (delta > la) delta = a_sub(delta, la); else delta = 0;
      }
      const int64_t ob = i64max(0, a_sub(un[j], sqn[j])), nb = i64max(0, a_sub(a_add(un[j], delta), sqn[j]));
      const int r = fr[j] % nR;
      #pragma unroll
      for (int rr = 0; rr < KQ_MAXR; rr++) if (rr == r) sum[rr] += nb - ob;
      pos += (nb > 0 ? 1 : 0) - (ob > 0 ? 1 : 0);
      if (want_bon && nb > 0 && qty[j] > 0) bon = 1;
    }
  }
  // drs_from_sums on the operands gathered above
  DRSv d; d.ratio = 0; d.weight = weight; d.borrowing = pos > 0; d.borrow_on = bon;
  #pragma unroll
  for (int r = 0; r < KQ_MAXR; r++) {
    if (r >= nR || sum[r] <= 0) continue;
    if (lend[r] > 0) { const double ratio = (double)sum[r] * 1000.0 / (double)lend[r]; if (ratio > d.ratio) d.ratio = ratio; }
  }
  *bytes += (int64_t)frcount * 24 + (d.borrowing ? (int64_t)frcount * 40 * (depth + 1) : 0);
  return d;
}
// entryComparer.less (fair_sharing_iterator.go:176-221) as a lexicographic key: a wins over b inside
// parentCohort <=> key(a) < key(b); ties keep the earlier tournament candidate.
//   k1: [PrioritizePreemptors] non-preemptor, [FairSharingPrioritizeNonBorrowing] borrowing on a requested
//       flavor-resource, zero-weight-but-borrowing (CompareDRS :112-123 puts those above everything else)
//   k2: the share CompareDRS compares (the raw ratio between two zero-weight borrowers, else PreciseWeightedShare);
//       both are >= 0, so the IEEE bit pattern orders them
//   k3: [PrioritySortingWithinCohort] priority descending ; k4: queue-order timestamp ascending
struct FsKey { uint64_t k1, k2, k3, k4; };
KQ_DEV bool fskey_less(const FsKey& a, const FsKey& b) {
  if (a.k1 != b.k1) return a.k1 < b.k1;
  if (a.k2 != b.k2) return a.k2 < b.k2;
  if (a.k3 != b.k3) return a.k3 < b.k3;
  return a.k4 < b.k4;
}
KQ_DEV uint64_t f64_bits(double v) {
  union { double d; uint64_t u; } x;
  x.d = v;
  return x.u;
}
KQ_DEV FsKey fs_make_key(const K& k, int en, const DRSv& d) {
  const DHeads& H = k.H;
  const bool zwb = drs_zwb(d);
  FsKey key;
  key.k1 = (gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS) && !(H.flags[en] & KQ_HEAD_IS_PREEMPTOR) ? 4u : 0u) |
           (gate(k, KQ_GATE_FS_PRIORITIZE_NON_BORROWING) && d.borrow_on ? 2u : 0u) | (zwb ? 1u : 0u);
  key.k2 = f64_bits(zwb ? d.ratio : drs_pws(d));
  key.k3 = gate(k, KQ_GATE_PRIORITY_SORTING_IN_COHORT) ? ~((uint64_t)H.priority[en] ^ 0x8000000000000000ull) : 0;
  key.k4 = (uint64_t)H.queue_ts[en] ^ 0x8000000000000000ull;
  return key;
}
// the cached key of entry `en` for the tournament inside `cohort`
KQ_DEV FsKey fs_key(const K& k, int slot, int en, int cohort) {
  const DSnap& S = k.S;
  const int c = k.H.cq[en];
  const uint64_t* p = k.X.fs_keys + (((size_t)slot * k.X.max_tree_cqs + S.cq_local[c]) * KQ_MAXD + (S.depth[c] - S.depth[cohort] - 1)) * 4;
  FsKey key; key.k1 = p[0]; key.k2 = p[1]; key.k3 = p[2]; key.k4 = p[3];
  return key;
}
// runTournament for ONE cohort (fair_sharing_iterator.go:125-163): candidates = the winners of the child
// cohorts, then the entries of the child ClusterQueues; one lane per candidate, wave arg-min on the key.
KQ_DEV int tournament_cohort(const K& k, int slot, int x, const int32_t* win, const int32_t* cq_ent) {
  const DSnap& S = k.S;
  const int lane = lane_id();
  const int h0 = S.child_cohort_off[x - S.nq], nh = S.child_cohort_off[x - S.nq + 1] - h0;
  const int c0 = S.child_cq_off[x - S.nq], ncq = S.child_cq_off[x - S.nq + 1] - c0;
  int best = -1;
  FsKey bk; bk.k1 = bk.k2 = bk.k3 = bk.k4 = 0;
  for (int base = 0; base < nh + ncq; base += WAVE) {
    const int j = base + lane;
    int cnd = -1;
    if (j < nh) cnd = win[S.node_local[S.child_cohort[h0 + j]]];
    else if (j < nh + ncq) cnd = cq_ent[S.cq_local[S.child_cq[c0 + j - nh]]];
    bool in = cnd >= 0;
    FsKey key; key.k1 = key.k2 = key.k3 = key.k4 = 0;
    if (in) key = fs_key(k, slot, cnd, x);
    if (wballot(in) == 0) continue;
    uint64_t mn;
    mn = wmin_u64(in ? key.k1 : ~0ull); in = in && key.k1 == mn;
    mn = wmin_u64(in ? key.k2 : ~0ull); in = in && key.k2 == mn;
    mn = wmin_u64(in ? key.k3 : ~0ull); in = in && key.k3 == mn;
    mn = wmin_u64(in ? key.k4 : ~0ull); in = in && key.k4 == mn;
    const int b = ffs64(wballot(in));  // ties keep the first candidate
    FsKey wk;
    wk.k1 = (uint64_t)wbcast_u((int64_t)key.k1, b); wk.k2 = (uint64_t)wbcast_u((int64_t)key.k2, b);
    wk.k3 = (uint64_t)wbcast_u((int64_t)key.k3, b); wk.k4 = (uint64_t)wbcast_u((int64_t)key.k4, b);
    const int wc = wbcast_u(cnd, b);
    if (best < 0 || fskey_less(wk, bk)) { best = wc; bk = wk; }  // an earlier chunk's winner keeps ties
  }
  return best;
}
// The iterator's per-tree state in LDS. A pop of the fair-sharing iterator is three tournaments on the popped entry's
// path (fair_sharing_iterator.go:125-163); with the state in global memory each of them was a chain of seven dependent loads (children
// offsets -> child id -> tree-local id -> winner -> its ClusterQueue -> that one's local id and depth -> the key): ~20 round trips per
// pop, most of the 24 us a pop took at cfg 3f. Here the children lists (tree-local ids, kq_prep.hpp fs_kid / fs_koff / fs_knc / fs_knh),
// the parent table, the depths, cqToEntry and the winners (entry + its ClusterQueue) live in the workgroup's LDS: a tournament is LDS
// reads and ONE round of independent key loads.
struct FIter { int32_t *ent, *win; int16_t *wcq, *kid, *koff, *knc, *knh, *par; int8_t* dep; uint8_t* stl; bool on; };   // stl: lowest path level whose cached DRS is out of date (255 = none)
KQ_HD size_t fiter_bytes(int nn, int nqs) {
  auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const int ncoh = nn - nqs > 0 ? nn - nqs : 1;
  return al((size_t)nqs * 4) + al((size_t)nn * 4) + al((size_t)nn * 2) * 3 + al((size_t)ncoh * 2) * 3 + al((size_t)nn) + al((size_t)nqs) + 64;
}
struct FWin { int e, wc; };
KQ_DEV FWin tournament_cohort_lds(const FIter& it, const uint64_t* fkeys, int xl, int nqs) {
  const int lane = lane_id();
  const int xc = xl - nqs;
  const int k0 = it.koff[xc], nkc = it.knc[xc], nkh = it.knh[xc];
  const int dx = it.dep[xl];
  FWin best{-1, -1};
  FsKey bk; bk.k1 = bk.k2 = bk.k3 = bk.k4 = 0;
  for (int base = 0; base < nkh + nkc; base += WAVE) {
    const int j = base + lane;
    int cnd = -1, wc = 0;
    if (j < nkh) { const int ch = it.kid[k0 + nkc + j]; cnd = it.win[ch]; wc = it.wcq[ch]; }          // winners of the child cohorts first (:129-137)
    else if (j < nkh + nkc) { wc = it.kid[k0 + j - nkh]; cnd = it.ent[wc]; }                             // then the child ClusterQueues' entries (:138-144)
    bool in = cnd >= 0;
    FsKey key; key.k1 = key.k2 = key.k3 = key.k4 = 0;
    if (in) { const uint64_t* p = fkeys + ((size_t)wc * KQ_MAXD + (it.dep[wc] - dx - 1)) * 4; key.k1 = p[0]; key.k2 = p[1]; key.k3 = p[2]; key.k4 = p[3]; }
    if (wballot(in) == 0) continue;
    uint64_t mn;
    mn = wmin_u64(in ? key.k1 : ~0ull); in = in && key.k1 == mn;
    mn = wmin_u64(in ? key.k2 : ~0ull); in = in && key.k2 == mn;
    mn = wmin_u64(in ? key.k3 : ~0ull); in = in && key.k3 == mn;
    mn = wmin_u64(in ? key.k4 : ~0ull); in = in && key.k4 == mn;
    const int b = ffs64(wballot(in));  // ties keep the first candidate
    FsKey wk;
    wk.k1 = (uint64_t)wbcast_u((int64_t)key.k1, b); wk.k2 = (uint64_t)wbcast_u((int64_t)key.k2, b);
    wk.k3 = (uint64_t)wbcast_u((int64_t)key.k3, b); wk.k4 = (uint64_t)wbcast_u((int64_t)key.k4, b);
    const int we = wbcast_u(cnd, b), wwc = wbcast_u(wc, b);
    if (best.e < 0 || fskey_less(wk, bk)) { best.e = we; best.wc = wwc; bk = wk; }  // an earlier chunk's winner keeps ties
  }
  return best;
}
// One workgroup per root-cohort tree. Wave 0 ("leader") runs the serial part — tournament on the path of the
// last popped entry, processEntry on the winner — and every wave of the workgroup takes part in computeDRS.
// computeDRS is incremental: DRS(path[l] of entry i, with i admitted) only reads usage rows of path[0..l], so it
// is recomputed only for levels at or above the lowest ancestor shared with the entry processEntry just changed;
// the algorithmic-byte counter is still charged for every (entry, level) the reference evaluates on each pop.
KQ_DEV void process_tree_fair(const K& k, Wave& w, int tree, int slot, int64_t* lds, size_t lds_bytes, int tid, int nthreads,
                              unsigned char* iter_mem = nullptr, size_t iter_bytes = 0) {
  const DSnap& S = k.S; const DOut& O = k.O; const DHeads& H = k.H;
  const int lane = lane_id();
  const bool leader = tid < WAVE;
  const int q0 = S.tree_cq_off[tree], nqs = S.tree_cq_off[tree + 1] - q0;
  const int n0 = S.tree_node_off[tree], nn = S.tree_node_off[tree + 1] - n0;
  int32_t* cq_ent = k.X.cq_ent + (size_t)slot * k.X.max_tree_cqs;
  int32_t* win = k.X.fs_win + (size_t)slot * k.X.max_tree_nodes;
  FIter it{};
  it.on = iter_mem != nullptr && nn > nqs && nn <= 32767 && iter_bytes >= fiter_bytes(nn, nqs);
  if (it.on) {
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    unsigned char* q = iter_mem;
    const int ncoh = nn - nqs;
    it.ent = (int32_t*)q; q += al((size_t)nqs * 4); it.win = (int32_t*)q; q += al((size_t)nn * 4);
    it.wcq = (int16_t*)q; q += al((size_t)nn * 2); it.kid = (int16_t*)q; q += al((size_t)nn * 2); it.par = (int16_t*)q; q += al((size_t)nn * 2);
    it.koff = (int16_t*)q; q += al((size_t)ncoh * 2); it.knc = (int16_t*)q; q += al((size_t)ncoh * 2); it.knh = (int16_t*)q; q += al((size_t)ncoh * 2);
    it.dep = (int8_t*)q; q += al((size_t)nn); it.stl = (uint8_t*)q;
    for (int i = tid; i < nn; i += nthreads) {
      it.kid[i] = S.fs_kid[n0 + i]; it.par[i] = S.fs_par[n0 + i]; it.dep[i] = (int8_t)S.depth[S.tree_nodes[n0 + i]]; it.wcq[i] = 0;
      if (i >= nqs) { it.koff[i - nqs] = S.fs_koff[n0 + i]; it.knc[i - nqs] = S.fs_knc[n0 + i]; it.knh[i - nqs] = S.fs_knh[n0 + i]; }
    }
    cq_ent = it.ent; win = it.win;
  }
  int32_t* seq = k.X.fs_seq + (size_t)slot * k.X.max_tree_cqs;
  uint64_t* fkeys = k.X.fs_keys + (size_t)slot * k.X.max_tree_cqs * KQ_MAXD * 4;
  // lds layout: [2 planes of the tree's cohort rows, if they fit][one prefetched entry record]
  const bool have_rec = lds_bytes >= sizeof(PRec);
  PRec* rec = (PRec*)((unsigned char*)lds + (lds_bytes - (have_rec ? sizeof(PRec) : 0)));
  int64_t fast_bytes = 0;
  uint8_t* stale = it.on ? it.stl : k.X.fs_stale + (size_t)slot * k.X.max_tree_cqs;
  int32_t* cost = k.X.fs_cost + (size_t)slot * k.X.max_tree_cqs * KQ_MAXD;
  long long* sum = k.X.fs_sum + slot;
  int32_t* ctl = k.X.fs_ctl + (size_t)slot * 4;
  if (tid == 0) {
    w.pc_ncq = nqs;
    w.pc_ncoh = nn - nqs;
    w.pc_lds = lds;
    w.pc_on = (w.pc_ncoh > 0 && have_rec && (size_t)w.pc_ncoh * S.nfr * 16 <= lds_bytes - sizeof(PRec)) ? 1 : 0;
    w.pc_region_bytes = have_rec ? (int)(lds_bytes - sizeof(PRec)) : 0;  // the rows (if resident) are flushed before a recomputation borrows the region
    w.help_on = k.help ? 1 : 0; w.help_tree = tree;
    *sum = 0; ctl[0] = 0; ctl[1] = -1; ctl[2] = 0; ctl[3] = 0;
    w.np_broken = 0; w.n_pre = 0; w.broken[0] = w.broken[1] = w.broken[2] = w.broken[3] = 0; w.n_pre = 0; w.broken[0] = w.broken[1] = w.broken[2] = w.broken[3] = 0;
  }
  for (int i = tid; i < nqs; i += nthreads) { cq_ent[i] = -1; stale[i] = 0; }
  for (int i = tid; i < nqs * KQ_MAXD; i += nthreads) cost[i] = 0;
  for (int i = tid; i < nn; i += nthreads) win[i] = -1;
  bsync();
  const CoreCtx cc = core_ctx(k, w, tree);
  cert_unverifiable(k, tree);  // the DRS tournament compares ClusterQueues of every subtree: a shard of the tree cannot run it alone
  // cqToEntry: the last head of a CQ wins (:58-60)
  for (int h = tid; h < hn(H); h += nthreads) {
    int c = H.cq[h];
    if (S.tree_of[c] == tree) atomic_max_i32(&cq_ent[S.cq_local[c]], h);
  }
  bsync();
  if (leader) {
    int remaining = 0;
    for (int base = 0; base < nqs; base += WAVE) { int i = base + lane; remaining += popc64(wballot(i < nqs && cq_ent[i] >= 0)); }
    if (lane == 0) ctl[0] = remaining;
    if (remaining > 0) pc_load(k, w, lds, tree);
    wsync();
  }
  bsync();
  if (ctl[0] == 0) return;
  if (nn == 1) {  // ClusterQueue without Cohort: its workload is simply returned (:71-78)
    if (leader) {
      int e = cq_ent[0];
      process_entry(k, w, e, 0, slot, tree);
      if (lane == 0) k.X.fs_key[e] = H.cq[e];
      wsync();
      pc_flush(k, w, lds, tree);
    }
    return;
  }
  const int root = S.path[(size_t)S.tree_cqs[q0] * KQ_MAXD + S.plen[S.tree_cqs[q0]] - 1];
  const int root_local = S.node_local[root];
  int last_ei = 0;
  const bool want_bon = gate(k, KQ_GATE_FS_PRIORITIZE_NON_BORROWING);
  const bool fs_plain = fs_plain_now(k);
  int lpos = 0;
  bool first = true;
  for (;;) {
    // ---- computeDRS (:227-263), all waves: one thread per ClusterQueue that still has an entry ----
    KQ_T0();
    {
      const int xc = ctl[1];
      const bool changed = ctl[2] != 0;
      int64_t delta = 0;
      const int xi = ctl[3];   // tree-local index of the ClusterQueue popped last
      for (int i = tid; i < nqs; i += nthreads) {
        const int en = cq_ent[i];
        if (en < 0) continue;
        if (fs_plain && it.on && it.dep[i] <= 3 && (!changed || it.dep[xi] <= 3)) {
          // The paths come from the iterator's LDS state (parents, depths): which levels are out of date is known without a global
          // access, and a ClusterQueue that only shares the root with the popped one — nine out of ten — is done here.
          const int plen = it.dep[i] + 1;
          int lp[4];
          lp[0] = i;
          #pragma unroll
          for (int l = 1; l < 4; l++) lp[l] = l < plen ? (int)it.par[lp[l - 1]] : -1;
          int from = stale[i];
          if (changed) {
            const int xl = it.dep[xi] + 1;
            int xp[4];
            xp[0] = xi;
            #pragma unroll
            for (int l = 1; l < 4; l++) xp[l] = l < xl ? (int)it.par[xp[l - 1]] : -1;
            int t = 0;
            while (t < plen - 1 && t < xl - 1 && fs_sel4(lp, plen - 2 - t) == fs_sel4(xp, xl - 2 - t)) t++;
            const int lvl = plen - 1 - t;
            if (lvl < from) from = lvl;
          }
          if (from + 1 >= plen) { if (from != 255) stale[i] = 255; continue; }
          int nd[4], row[4];
          #pragma unroll
          for (int l = 0; l < 4; l++) {
            nd[l] = S.tree_nodes[n0 + (l < plen ? lp[l] : i)];
            row[l] = (l >= 1 && l < plen && w.pc_on) ? lp[l] - nqs : -1;
          }
          const int32_t* ufr = O.use_fr + (size_t)en * KQ_MAXU; const int64_t* uqty = O.use_qty + (size_t)en * KQ_MAXU;
          const int nu = (H.flags[en] & KQ_HEAD_HAS_QUOTA_RESERVATION) ? 0 : O.use_n[en];  // netUsage scheduler.go:785-794
          for (int l = from; l + 1 < plen; l++) {
            int64_t lb = 0;
            const DRSv d = drs_entry_level_g(k, w, nd, row, l, ufr, uqty, nu, want_bon, &lb);
            const size_t o = (size_t)i * KQ_MAXD + l;
            const FsKey key = fs_make_key(k, en, d);
            fkeys[o * 4 + 0] = key.k1; fkeys[o * 4 + 1] = key.k2; fkeys[o * 4 + 2] = key.k3; fkeys[o * 4 + 3] = key.k4;
            delta += lb - cost[o];
            cost[o] = (int32_t)lb;
          }
          stale[i] = 255;
          continue;
        }
        const int c = S.tree_cqs[q0 + i];
        const int32_t* path = S.path + (size_t)c * KQ_MAXD;
        const int plen = S.plen[c];
        int from = stale[i];
        if (changed) {  // lowest level of this path that is also on the popped entry's path
          const int32_t* xp = S.path + (size_t)xc * KQ_MAXD;
          const int xl = S.plen[xc];
          int t = 0;
          while (t < plen - 1 && t < xl - 1 && path[plen - 1 - t - 1] == xp[xl - 1 - t - 1]) t++;
          const int lvl = plen - 1 - t;
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
      if (lane == 0 && tot) atomic_add_i64(sum, (long long)tot);
    }
    if (leader) KQ_TS(k, 16);
    bsync();
    if (leader) KQ_TS(k, 17);
    // ---- leader: tournament, pop, processEntry ----
    // An entry that changes no usage (skipped, NoFit, no longer fits: nine out of ten pops at cfg 3f) leaves every DRS value as it is:
    // the next pop needs no computeDRS pass and no barrier pair, only the tournaments on the popped entry's path — the leader keeps
    // popping on its own until an entry does change usage (the other waves wait at the barrier below meanwhile).
    if (leader) for (bool lrun = true; lrun;) {
      if (it.on) {
        if (first) {
          for (int d = KQ_MAXD - 1; d >= 0; d--)
            for (int i = nqs; i < nn; i++) {
              if (it.dep[i] != d) continue;
              const FWin b = tournament_cohort_lds(it, fkeys, i, nqs);
              if (lane == 0) { it.win[i] = b.e; it.wcq[i] = (int16_t)(b.wc < 0 ? 0 : b.wc); }
              wsync_lds();
            }
        } else {
          for (int xl = it.par[last_ei]; xl >= 0; xl = it.par[xl]) {  // only the cohorts that nominated the popped entry can change
            const FWin b = tournament_cohort_lds(it, fkeys, xl, nqs);
            if (lane == 0) { it.win[xl] = b.e; it.wcq[xl] = (int16_t)(b.wc < 0 ? 0 : b.wc); }
            wsync_lds();
          }
        }
      } else if (first) {
        for (int d = KQ_MAXD - 1; d >= 0; d--)
          for (int i = nqs; i < nn; i++) {
            const int x = S.tree_nodes[n0 + i];
            if (S.depth[x] != d) continue;
            const int b = tournament_cohort(k, slot, x, win, cq_ent);
            if (lane == 0) win[i] = b;
            wsync();
          }
      } else {
        const int xc = ctl[1];
        for (int l = 1; l < S.plen[xc]; l++) {  // only the cohorts that nominated the popped entry can change
          const int x = S.path[(size_t)xc * KQ_MAXD + l];
          const int b = tournament_cohort(k, slot, x, win, cq_ent);
          if (lane == 0) win[S.node_local[x]] = b;
          wsync();
        }
      }
      KQ_TS(k, 18);
      const int e = win[root_local];
      const int ei = it.on ? (int)it.wcq[root_local] : S.cq_local[H.cq[e]];
      const int ec = it.on ? S.tree_cqs[q0 + ei] : H.cq[e];
      last_ei = ei;
      if (lane == 0) {
        atomic_add_i64(O.stat_bytes, *sum);  // the reference evaluates every remaining (entry, level) on every pop
        long long mine = 0;
        for (int l = 0; l + 1 < S.plen[ec]; l++) mine += cost[(size_t)ei * KQ_MAXD + l];
        *sum -= mine;
        cq_ent[ei] = -1; seq[lpos] = e;
        w.usage_dirty = 0;
      }
      wsync();
      KQ_TS(k, 19);
      // processEntry: the straight-line core on a prefetched record when the entry has no preemption targets
      bool done = false;
      if (have_rec && !np_exact_mode(w)) {
        if (lane == 0) { w.win_e[0] = e; w.win_cq[0] = ec; w.win_off[0] = 0; }
        wsync();
        chunk_prefetch_wave(k, w, rec, EntList{w.win_e, w.win_cq, w.win_off, lpos}, 1);
        if (!rec->slow) {
          chunk_entry_fast(w, lds, *rec, cc, rec_preload(*rec));
          wsync();
          fast_bytes += rec_algo_bytes(*rec);
          if (lane == 0) {
            O.status[e] = rec->status; O.action[e] = rec->action; O.requeue_reason[e] = rec->rq; O.skip[e] = rec->skip; O.mode[e] = rec->omode;
            O.order[e] = lpos;
            w.usage_dirty = rec->dirty;
          }
          wsync();
          chunk_scatter(k, rec, 1);
          done = true;
        }
      }
      if (!done) process_entry(k, w, e, lpos, slot, tree);
      KQ_TS(k, 20);
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
      lpos++;
#if defined(KQ_PROF) && !defined(KQ_HOST_EMU)
      if (lane == 0 && w.usage_dirty) atomic_add_i64((long long*)k.prof + 15, 1);   // pops that changed usage (each one costs a computeDRS pass)
#endif
      if (lane == 0) { ctl[0] -= 1; ctl[1] = ec; ctl[2] = w.usage_dirty; ctl[3] = ei; }
      wsync();
      first = false;
      lrun = k.C.fs_lrun && ctl[0] > 0 && !w.usage_dirty;
    }
    bsync();
    if (ctl[0] == 0) break;
    first = false;
  }
  if (!leader) return;
  if (lane == 0 && fast_bytes) atomic_add_i64(O.stat_bytes, (long long)fast_bytes);
  // merge key of the canonical getCq (lowest CQ index still in the map): the tree pops until
  // that CQ's own entry has been returned, so an entry is emitted in the turn of the smallest CQ at or after it.
  if (lane == 0) {
    int run = 0x7fffffff;
    for (int j = lpos - 1; j >= 0; j--) { int e = seq[j]; int c = H.cq[e]; if (c < run) run = c; k.X.fs_key[e] = run; }
  }
  wsync();
  pc_flush(k, w, lds, tree);
}
// cycle-start borrowed sums (k_fs_sums): the count of borrowed cells of (node, r) is parked in bs_pos scratch
// layout [N*nR] is not available, so bu_pos is produced by a second pass over the node's resources
KQ_DEV void fs_sums_cell(const K& k, int node, int r) {
  PG pg{&k.S, k.usage};
  int64_t sm; int ps;
  node_sums(k.S, node, pg, r, &sm, &ps);
  k.X.bu_sum[(size_t)node * k.S.nR + r] = sm;
}
KQ_DEV void fs_pos_node(const K& k, int node) {
  PG pg{&k.S, k.usage};
  int tot = 0;
  for (int r = 0; r < k.S.nR; r++) { int64_t sm; int ps; node_sums(k.S, node, pg, r, &sm, &ps); tot += ps; }
  k.X.bu_pos[node] = tot;
}
// global iteration position of a fair-sharing entry: rank of (merge key, position in the tree's sequence)
KQ_DEV int fair_rank(const K& k, int e, int f_begin, int f_end) {
  const int ke = k.X.fs_key[e], pe = k.O.order[e];
  int r = 0;
  for (int f = f_begin; f < f_end; f++) {
    int kf = k.X.fs_key[f];
    if (kf < 0) continue;
    if (kf < ke || (kf == ke && k.O.order[f] < pe)) r++;
  }
  return r;
}

// ------------------------------------------------------------------------------------------------
// snapshot derivation (resource_node.go:167-230): SubtreeQuota of every node and Usage of every Cohort from
// the Quotas and the ClusterQueue usage. One thread per (node, flavor-resource); cohorts level by level, deepest
// first, children in hierarchy order (child cohorts, then ClusterQueues) as accumulateFromChild is called.
// ------------------------------------------------------------------------------------------------
struct DDerive { int64_t* sq; int64_t* usage; uint8_t* flags; };
KQ_DEV void derive_cq_cell(const DSnap& S, const DDerive& d, int cq, int fr) {  // updateClusterQueueResourceNode :167-173
  const size_t o = ix(S, cq, fr);
  uint8_t f = d.flags[o] & (uint8_t)~KQ_QF_SUBTREE;
  int64_t q = 0;
  if (f & KQ_QF_QUOTA) { q = S.nominal[o]; f |= KQ_QF_SUBTREE; }
  d.sq[o] = q; d.flags[o] = f;
}
KQ_DEV void derive_cohort_cell(const DSnap& S, const DDerive& d, int cohort, int fr) {  // updateCohortResourceNode :183-230
  const size_t o = ix(S, cohort, fr);
  uint8_t f = d.flags[o] & (uint8_t)~KQ_QF_SUBTREE;
  int64_t q = 0, u = 0;
  if (f & KQ_QF_QUOTA) { q = S.nominal[o]; f |= KQ_QF_SUBTREE; }
  const int kx = cohort - S.nq;
  for (int pass = 0; pass < 2; pass++) {
    const int32_t* off = pass == 0 ? S.child_cohort_off : S.child_cq_off;
    const int32_t* lst = pass == 0 ? S.child_cohort : S.child_cq;
    for (int i = off[kx]; i < off[kx + 1]; i++) {
      const size_t co = ix(S, lst[i], fr);
      const int64_t csq = d.sq[co], ll = S.ll[co];
      const int64_t lq = ll != KQ_NIL_LIMIT ? i64max(0, a_sub(csq, ll)) : 0;
      if (d.flags[co] & KQ_QF_SUBTREE) { q = a_add(q, a_sub(csq, lq)); f |= KQ_QF_SUBTREE; }
      u = a_add(u, i64max(0, a_sub(d.usage[co], lq)));
    }
  }
  d.sq[o] = q; d.usage[o] = u; d.flags[o] = f;
}

// ------------------------------------------------------------------------------------------------
// closed-loop support: fold admitted usage into the resident snapshot / take it out again
// (cache side of assumeWorkload: clusterqueue.go:594 updateWorkloadUsage -> resource_node.go:144-165)
// ------------------------------------------------------------------------------------------------
// Start-of-cycle housekeeping as ONE launch: fills and device-to-device copies of 4-byte words (k_prep).
struct DPrepOp { void* dst; const void* src; uint32_t words; uint32_t fill; };  // src == nullptr: fill
struct DPrep { int n; DPrepOp op[16]; };
KQ_DEV void prep_word(const DPrep& p, int o, uint32_t i) {
  const DPrepOp& x = p.op[o];
  ((uint32_t*)x.dst)[i] = x.src ? ((const uint32_t*)x.src)[i] : x.fill;
}
struct DCommit {
  int n;                  // heads of the committed cycle
  const int32_t* cq;      // [n]
  const int32_t* use_n;   // [n] 0 for heads that were not admitted (or hold a quota reservation already)
  const int32_t* use_fr;  // [n * KQ_MAXU]
  const int64_t* use_qty;
  int64_t* usage;         // the snapshot's usage plane
  int32_t* big;           // [1] K::usage_big
};
// one wave per root-cohort tree; entries of a tree touch the same cohort cells, so they are applied one after another,
// lanes = the entry's flavor-resources (independent columns)
KQ_DEV void commit_tree(const DSnap& S, const DCommit& c, int tree, bool add) {
  for (int h = 0; h < c.n; h++) {
    const int nu = c.use_n[h];
    if (nu == 0) continue;
    const int cq = c.cq[h];
    if (S.tree_of[cq] != tree) continue;
    for (int u = lane_id(); u < nu; u += WAVE) {
      const int fr = c.use_fr[(size_t)h * KQ_MAXU + u];
      UGW g{&S, c.usage, fr};
      if (add) add_usage(S, S.path + (size_t)cq * KQ_MAXD, S.plen[cq], fr, c.use_qty[(size_t)h * KQ_MAXU + u], g);
      else remove_usage(S, S.path + (size_t)cq * KQ_MAXD, S.plen[cq], fr, c.use_qty[(size_t)h * KQ_MAXU + u], g);
      if (c.big && (uint64_t)g.get(cq) >= ((uint64_t)1 << 50)) *c.big = 1;
    }
    wsync();
  }
}
// Fast commit when the snapshot's cohort usage is consistent with its children (kq_prep usage_consistent): the
// admissions are added at the ClusterQueue level in parallel and cohort usage is re-derived level by level —
// addUsage / removeUsage preserve "cohort usage = sum of what the children store in it" (resource_node.go:217-230),
// so the result is the one sequential bubbling produces.
KQ_DEV void commit_cq_cell(const DCommit& c, const DSnap& S, int h, int u, bool add) {
  if (u >= c.use_n[h]) return;
  const int fr = c.use_fr[(size_t)h * KQ_MAXU + u];
  const int64_t q = c.use_qty[(size_t)h * KQ_MAXU + u];
  // resources.Amount.Add / Sub (amount.go:114-145: saturating, Unlimited absorbing) under a compare-and-swap: admissions of one
  // commit only add, releases only subtract, so the result does not depend on the order the heads arrive in
  int64_t* cell = &c.usage[(size_t)c.cq[h] * S.nfr + fr];
  int64_t old = *(volatile int64_t*)cell, nw;
  for (;;) {
    nw = add ? a_add(old, q) : a_sub(old, q);
    const int64_t seen = atomic_cas_i64(cell, old, nw);
    if (seen == old) break;
    old = seen;
  }
  if (c.big && (uint64_t)nw >= ((uint64_t)1 << 50)) *c.big = 1;  // negatives land here too
}
// kq_pending_step: the commit of the cycle in ONE pass over the (head, slot) cells — what commit_keep_cell + commit_cq_cell do in two
// launches: keep the cycle's usage rows for the later release and fold the admitted ones into the ClusterQueue cells. Same
// arithmetic (commit_cq_cell on the kept row), so the plane is identical; only valid while the cohort levels are re-derived from
// the ClusterQueue cells (usage_consistent).
KQ_DEV void commit_fused_cell(const K& k, const DSnap& S, const DCommit& c, int i) {
  int32_t* use_n_out = const_cast<int32_t*>(c.use_n); int32_t* cq_out = const_cast<int32_t*>(c.cq);
  int32_t* fr_out = const_cast<int32_t*>(c.use_fr); int64_t* qty_out = const_cast<int64_t*>(c.use_qty);
  const int h = i / KQ_MAXU, u = i % KQ_MAXU;
  if (h >= hn(k.H)) { if (u == 0) use_n_out[h] = 0; return; }
  const bool take = k.O.action[h] == KQ_ACT_ADMIT && !(k.H.flags[h] & KQ_HEAD_HAS_QUOTA_RESERVATION) && !(k.O.error && k.O.error[0] != 0);
  const int nu = take ? k.O.use_n[h] : 0;
  const int fr = k.O.use_fr[i]; const int64_t q = k.O.use_qty[i];
  fr_out[i] = fr; qty_out[i] = q;
  const int cq = k.H.cq[h];
  if (u == 0) { use_n_out[h] = nu; cq_out[h] = cq; }
  if (u >= nu) return;
  int64_t* cell = &c.usage[(size_t)cq * S.nfr + fr];
  int64_t old = *(volatile int64_t*)cell, nw;
  for (;;) {
    nw = a_add(old, q);
    const int64_t seen = atomic_cas_i64(cell, old, nw);
    if (seen == old) break;
    old = seen;
  }
  if (c.big && (uint64_t)nw >= ((uint64_t)1 << 50)) *c.big = 1;
}
KQ_DEV void derive_usage_cell(const DSnap& S, int64_t* usage, int cohort, int fr) {
  int64_t u = 0;
  const int kx = cohort - S.nq;
  for (int pass = 0; pass < 2; pass++) {
    const int32_t* off = pass == 0 ? S.child_cohort_off : S.child_cq_off;
    const int32_t* lst = pass == 0 ? S.child_cohort : S.child_cq;
    // children in chunks of 16: the ids first, then the three cells of every child — the loads of a chunk are in flight together (one child
    // after the other was two dependent round trips per child: 10 children x 3 levels = most of k_usage_cols' 22 us at cfg 3); the sum
    // keeps the children's order
    constexpr int CH = 16;
    for (int i0 = off[kx], i1 = off[kx + 1]; i0 < i1; i0 += CH) {
      int ch[CH]; int64_t uv[CH], llv[CH], sqv[CH];
      #pragma unroll
      for (int q = 0; q < CH; q++) ch[q] = i0 + q < i1 ? lst[i0 + q] : -1;
      #pragma unroll
      for (int q = 0; q < CH; q++) {
        uv[q] = 0; llv[q] = KQ_NIL_LIMIT; sqv[q] = 0;
        if (ch[q] >= 0) { const size_t co = ix(S, ch[q], fr); uv[q] = usage[co]; llv[q] = S.ll[co]; sqv[q] = S.sq[co]; }
      }
      #pragma unroll
      for (int q = 0; q < CH; q++) {
        if (ch[q] < 0) continue;
        const int64_t lq = llv[q] != KQ_NIL_LIMIT ? i64max(0, a_sub(sqv[q], llv[q])) : 0;   // local_quota
        u = a_add(u, i64max(0, a_sub(uv[q], lq)));
      }
    }
  }
  usage[ix(S, cohort, fr)] = u;
}
// sharded single-root cycles (kueue_amd/sharding.py): what the cycle added to every usage cell, and folding an all-reduced
// ClusterQueue-level delta into the resident snapshot
KQ_DEV void usage_delta_cell(int64_t* out, const int64_t* work, const int64_t* start, size_t i) { out[i] = work[i] - start[i]; }
KQ_DEV void usage_add_cell(int64_t* usage, const int64_t* delta, size_t i, int sign, int32_t* big) {
  const int64_t d = delta[i];
  if (d == 0) return;
  const int64_t v = sign > 0 ? a_add(usage[i], d) : a_sub(usage[i], d);
  usage[i] = v;
  if (big && (uint64_t)v >= ((uint64_t)1 << 50)) *big = 1;
}
// which heads of the last cycle count: action == admit and no quota reservation held (netUsage scheduler.go:785-794)
KQ_DEV void commit_mask_head(const K& k, int h, int32_t* use_n_out, int32_t* cq_out, int32_t* count) {
  // (a cycle that ended with a device-side error commits nothing: the asynchronous step enqueues the commit before the host has seen the flag)
  const bool take = k.O.action[h] == KQ_ACT_ADMIT && !(k.H.flags[h] & KQ_HEAD_HAS_QUOTA_RESERVATION) && !(k.O.error && k.O.error[0] != 0);
  use_n_out[h] = take ? k.O.use_n[h] : 0;
  cq_out[h] = k.H.cq[h];
  if (take) atomic_add_i32(count, 1);
}
// one call per (head, slot): keeps the cycle's usage rows for a later release (the cycle's own buffers are reused)
KQ_DEV void commit_keep_cell(const K& k, int i, int32_t* use_n_out, int32_t* cq_out, int32_t* fr_out, int64_t* qty_out, int32_t* count) {
  if (i >= hn(k.H) * KQ_MAXU) {  // rows between the head count and the bound the arrays are sized for (kq_pending_step) commit nothing
    if (i % KQ_MAXU == 0) use_n_out[i / KQ_MAXU] = 0;
    return;
  }
  fr_out[i] = k.O.use_fr[i]; qty_out[i] = k.O.use_qty[i];
  if (i % KQ_MAXU == 0) commit_mask_head(k, i / KQ_MAXU, use_n_out, cq_out, count);
}

// classical entry order (scheduler.go:1110-1163): a precedes b
// ---- sharded nominate: export of the own heads' nomination / import of the merged one (one call per head; nps_total = podsets of the batch)
KQ_DEV void shard_export_head(const K& k, int h, size_t nps_total) {
  const DShard& sh = k.shard; const DOut& O = k.O; const DHeads& H = k.H;
  if (sh.mine && !sh.mine[h]) return;
  const int n = H.n, nR = k.S.nR;
  int64_t* x = sh.x + (size_t)h * SH_HEAD;
  const int tn = O.tgt_n[h];
  x[0] = O.nominated_mode[h]; x[1] = O.borrowing[h]; x[2] = O.use_n[h];
  x[3] = (int64_t)tn << 32 | (int64_t)(uint32_t)(tn ? sh.rank * sh.pool_cap + O.tgt_pos[h] : 0);
  const int un = O.use_n[h] < KQ_MAXU ? O.use_n[h] : KQ_MAXU;
  for (int u = 0; u < un; u++) { x[4 + u] = O.use_fr[(size_t)h * KQ_MAXU + u]; x[4 + KQ_MAXU + u] = O.use_qty[(size_t)h * KQ_MAXU + u]; }
  int64_t* xp = sh.x + shard_off_ps(n);
  int64_t* xc = sh.x + shard_off_cell(n, nps_total);
  for (int p = H.ps_off[h]; p < H.ps_off[h + 1]; p++) {
    xp[p] = O.ps_count[p];
    for (int r = 0; r < nR; r++) {
      const size_t c = (size_t)p * nR + r;
      xc[c] = (int64_t)(uint32_t)(O.flavor[c] + 1) | (int64_t)O.res_mode[c] << 24 | (int64_t)(uint32_t)(O.tried_idx[c] + 1) << 32;
    }
  }
  if (O.rsn_win) {
    int64_t* xr = sh.x + shard_off_rsn(n, nps_total, nR, sh.world);
    xr[h] = O.rsn_n[h];
    const int cnt = O.rsn_n[h] < 0 ? -O.rsn_n[h] : O.rsn_n[h];
    const int64_t* src = (const int64_t*)(O.rsn + (size_t)h * O.rsn_win);
    int64_t* dst = xr + n + (size_t)h * O.rsn_win * 4;
    for (int q = 0; q < cnt * 4; q++) dst[q] = src[q];
  }
}
KQ_DEV void shard_export_misc(const K& k, size_t nps_total, int rsn_win) {  // one thread: counters; the pool is copied by shard_export_pool
  const DShard& sh = k.shard;
  int64_t* xm = sh.x + shard_off_misc(k.H.n, nps_total, k.S.nR);
  xm[0] = *k.O.stat_bytes; xm[1] = *k.O.error; xm[2 + sh.rank] = *k.O.pool_count;
  (void)rsn_win;
}
KQ_DEV void shard_export_pool(const K& k, size_t nps_total, int rsn_win, int t) {
  const DShard& sh = k.shard;
  if (t >= *k.O.pool_count || t >= sh.pool_cap) return;
  sh.x[shard_off_pool(k.H.n, nps_total, k.S.nR, sh.world, rsn_win) + (size_t)sh.rank * sh.pool_cap + t] = (int64_t)(uint32_t)k.O.pool_row[t] | (int64_t)k.O.pool_reason[t] << 32;
}
KQ_DEV void shard_import_head(const K& k, int h, size_t nps_total) {
  const DShard& sh = k.shard; const DOut& O = k.O; const DHeads& H = k.H;
  const int n = H.n, nR = k.S.nR;
  const int64_t* x = sh.x + (size_t)h * SH_HEAD;
  O.nominated_mode[h] = (uint8_t)x[0]; O.borrowing[h] = (int32_t)x[1]; O.use_n[h] = (int32_t)x[2];
  // (what nominate_head leaves besides the nomination itself)
  O.mode[h] = (uint8_t)x[0]; O.status[h] = KQ_ST_NOT_NOMINATED; O.action[h] = KQ_ACT_NONE; O.requeue_reason[h] = KQ_RQ_GENERIC; O.skip[h] = KQ_SKIP_NONE; O.order[h] = -1;
  O.tgt_n[h] = (int32_t)(x[3] >> 32); O.tgt_pos[h] = (int32_t)(x[3] & 0xffffffff);
  const int un = x[2] < KQ_MAXU ? (int)x[2] : KQ_MAXU;
  for (int u = 0; u < un; u++) { O.use_fr[(size_t)h * KQ_MAXU + u] = (int32_t)x[4 + u]; O.use_qty[(size_t)h * KQ_MAXU + u] = x[4 + KQ_MAXU + u]; }
  const int64_t* xp = sh.x + shard_off_ps(n);
  const int64_t* xc = sh.x + shard_off_cell(n, nps_total);
  for (int p = H.ps_off[h]; p < H.ps_off[h + 1]; p++) {
    O.ps_count[p] = (int32_t)xp[p];
    for (int r = 0; r < nR; r++) {
      const size_t c = (size_t)p * nR + r;
      const int64_t w = xc[c];
      O.flavor[c] = (int32_t)(w & 0xffffff) - 1; O.res_mode[c] = (uint8_t)((w >> 24) & 0xff); O.tried_idx[c] = (int32_t)(uint32_t)(w >> 32) - 1;
    }
  }
  if (O.rsn_win) {
    const int64_t* xr = sh.x + shard_off_rsn(n, nps_total, nR, sh.world);
    O.rsn_n[h] = (int32_t)xr[h];
    const int cnt = O.rsn_n[h] < 0 ? -O.rsn_n[h] : O.rsn_n[h];
    int64_t* dst = (int64_t*)(O.rsn + (size_t)h * O.rsn_win);
    const int64_t* src = xr + n + (size_t)h * O.rsn_win * 4;
    for (int q = 0; q < cnt * 4; q++) dst[q] = src[q];
  }
}
KQ_DEV void shard_import_misc(const K& k, size_t nps_total) {
  const DShard& sh = k.shard;
  const int64_t* xm = sh.x + shard_off_misc(k.H.n, nps_total, k.S.nR);
  *k.O.stat_bytes = xm[0];
  if (xm[1] != 0) *k.O.error = KQ_EUNSUPPORTED;   // (the sum of the ranks' error words: any rank's device-side error fails the cycle)
  *k.O.pool_count = sh.world * sh.pool_cap;        // a recomputation appends behind the imported segments
}
KQ_DEV void shard_import_pool(const K& k, size_t nps_total, int rsn_win, int t) {
  const DShard& sh = k.shard;
  if (t >= sh.world * sh.pool_cap) return;
  const int64_t w = sh.x[shard_off_pool(k.H.n, nps_total, k.S.nR, sh.world, rsn_win) + t];
  k.O.pool_row[t] = (int32_t)(w & 0xffffffff); k.O.pool_reason[t] = (uint8_t)(w >> 32);
}

KQ_DEV bool entry_before(const K& k, int a, int b) {
  bool aq = k.H.flags[a] & KQ_HEAD_HAS_QUOTA_RESERVATION, bq = k.H.flags[b] & KQ_HEAD_HAS_QUOTA_RESERVATION;
  if (aq != bq) return aq;
  if (gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS)) {
    bool ap = k.H.flags[a] & KQ_HEAD_IS_PREEMPTOR, bp = k.H.flags[b] & KQ_HEAD_IS_PREEMPTOR;
    if (ap != bp) return ap;
  }
  int ab = k.O.borrowing[a], bb = k.O.borrowing[b];
  if (ab != bb) return ab < bb;
  if (gate(k, KQ_GATE_PRIORITY_SORTING_IN_COHORT)) {
    int64_t pa = k.H.priority[a], pb = k.H.priority[b];
    if (pa != pb) return pa > pb;
  }
  int64_t ta = k.H.queue_ts[a], tb = k.H.queue_ts[b];
  if (ta != tb) return ta < tb;
  return a < b;  // canonical stable order
}
}  // namespace kq
#include "kq_pending.hpp"

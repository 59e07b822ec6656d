// kq_engine.hip — gfx950 kernels + the C-ABI entry points of include/kq_engine.h.
//
// Launch geometry (MI355X: 256 CUs x 4 SIMD-32, wave64, 8 XCDs with private L2):
//   k_nominate : 1 wave (64 threads) per workgroup, grid-stride over heads; up to 4096 resident waves.
//                A head's CQ->root path rows (5 planes x (D+1) x FR) are the hot data; consecutive heads
//                are consecutive ClusterQueues (canonical name order) i.e. siblings in the cohort tree, so
//                workgroup w -> XCD w%8 keeps a cohort's ancestor rows hot in 8 L2s at once (read-only
//                sharing across XCDs is safe: the planes are not written during nominate).
//   k_prep     : the fills and device-to-device copies a cycle starts with, one launch (blockIdx.y = operation).
//   k_records  : 1 thread per (head, flavor-resource slot, path level): the static part of the entry records k_process
//                copies into LDS.
//   k_order    : rank of every entry by pairwise compare of three-word keys (H <= a few thousand per cycle at reference
//                semantics: <= 1 head per ClusterQueue); 256 entries x 64-key tiles per workgroup.
//   k_process  : 1 workgroup of 4 waves per root-cohort tree; entries of one tree are sequentially dependent
//                (scheduler.go:486 cq.AddUsage changes what later entries see): wave 0 walks them, waves 1-3 fetch and
//                screen the records of the next chunk.
// No MFMA anywhere: saturating int64 compares/adds over gathered quota rows -> HBM/L2-bound.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "kq_host.hpp"
#include "kq_tas_host.hpp"

using namespace kq;

// Kernels take the argument block K by pointer (it lives in HBM and is read through the scalar cache):
// passing ~700 B by value makes the compiler spill it to scratch as soon as any non-inlined device
// function takes a reference to it, which put scratch loads into the serial core of k_process.
// First pass over every head: flavor assignment without any victim-search / partial-admission machinery linked in (the common
// case by far); heads that need more are appended to K::defer_list.
__global__ __launch_bounds__(64) void k_nominate_lean(const K* __restrict__ kp, int slots) {
  const K& k = *kp;
  __shared__ Wave w;
  for (int h = blockIdx.x, n = hn(k.H); h < n; h += slots) nominate_head_lean(k, w, h);
}
// Full pass (victim searches, GetTargets, partial admission) over the heads the first pass deferred.
__global__ __launch_bounds__(64) void k_nominate(const K* __restrict__ kp, int slots, unsigned lds_bytes) {
  const K& k = *kp;
  __shared__ Wave w;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  if (threadIdx.x == 0) { w.cs_lds = lds_bytes ? dyn_lds : nullptr; w.cs_lds_bytes = (int)lds_bytes; w.help_on = 0; }
  __syncthreads();
  const int slot = blockIdx.x;
  const int nd = *k.defer_count;
  (void)slots;
  // heads are pulled, not dealt: a head with victim searches costs 10^3 x one without, and the searches of two heads differ by 10 x
  __shared__ int next;
  for (;;) {
    if (threadIdx.x == 0) next = atomicAdd(k.nom_ticket, 1);
    __syncthreads();
    const int i = next;
    __syncthreads();
    if (i >= nd) break;
    nominate_head(k, w, k.defer_list[i], slot);
  }
}

// Between the two passes: the SimulatePreemption calls the deferred heads' flavor scans will make, one per wave (kq_device.hpp sim_worker),
// scan by scan: k_nominate_emit walks the deferred heads again with the results so far and lists each head's next scan.
__global__ __launch_bounds__(64) void k_nominate_emit(const K* __restrict__ kp, int round) {
  const K& k = *kp;
  __shared__ Wave w;
  __shared__ int next;
  const int nd = *k.defer_count;
  for (;;) {
    if (threadIdx.x == 0) next = atomicAdd(&k.sim_ctl[SIMC_EMIT + round], 1);
    __syncthreads();
    const int i = next;
    __syncthreads();
    if (i >= nd) break;
    nominate_head_emit(k, w, k.defer_list[i]);
  }
}
__global__ __launch_bounds__(64) void k_nominate_sim(const K* __restrict__ kp, unsigned lds_bytes, int round) {
  const K& k = *kp;
  __shared__ Wave w;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  if (threadIdx.x == 0) { w.cs_lds = lds_bytes ? dyn_lds : nullptr; w.cs_lds_bytes = (int)lds_bytes; w.help_on = 0; }
  __syncthreads();
  sim_worker(k, w, blockIdx.x, round);
}

// Entry order (scheduler.go:1110-1163): rank(i) = number of entries that precede i. 2-D grid: block (bi, bj)
// compares 256 entries i against a 64-key tile j staged in LDS and adds its partial count to rank[i];
// k_order_scatter then writes order_idx[rank[i]] = i. H^2/16384 blocks keep more CUs busy than H/256.
// The predicate of kq::entry_before as a lexicographic compare of three unsigned words (then the index), so that the inner
// loop has no divergent branches: a = [no quota reservation | not a preemptor (gate) | borrowing level], b = priority
// descending (gate), c = queue timestamp ascending.
struct OrderKey { uint64_t a, b, c; };
constexpr int ORDER_TILE = 64;
__device__ __forceinline__ OrderKey order_key(const K& k, int h, bool pre, bool psort) {
  const uint32_t fl = k.H.flags[h];
  OrderKey o;
  o.a = ((fl & KQ_HEAD_HAS_QUOTA_RESERVATION) ? 0ull : 1ull << 63) | ((pre && !(fl & KQ_HEAD_IS_PREEMPTOR)) ? 1ull << 62 : 0ull) |
        (uint64_t)((uint32_t)k.O.borrowing[h] ^ 0x80000000u);
  o.b = psort ? ~((uint64_t)k.H.priority[h] ^ 0x8000000000000000ull) : 0ull;
  o.c = (uint64_t)k.H.queue_ts[h] ^ 0x8000000000000000ull;
  return o;
}
__device__ __forceinline__ void order_block(const K& k, int32_t* rank, int bx, int by, OrderKey* tile) {
  const int n = hn(k.H);
  const int i = bx * 256 + threadIdx.x;
  const int base = by * ORDER_TILE;
  const bool pre = gate(k, KQ_GATE_PRIORITIZE_PREEMPTORS), psort = gate(k, KQ_GATE_PRIORITY_SORTING_IN_COHORT);
  if ((int)threadIdx.x < ORDER_TILE && base + (int)threadIdx.x < n) tile[threadIdx.x] = order_key(k, base + threadIdx.x, pre, psort);
  __syncthreads();
  if (i >= n) return;
  const OrderKey me = order_key(k, i, pre, psort);
  const int m = (n - base) < ORDER_TILE ? (n - base) : ORDER_TILE;
  int cnt = 0;
  for (int t = 0; t < m; t++) {
    const OrderKey o = tile[t];
    const int jj = base + t;
    // does entry jj precede entry i ?
    const bool before = o.a < me.a || (o.a == me.a && (o.b < me.b || (o.b == me.b && (o.c < me.c || (o.c == me.c && jj < i)))));
    cnt += before ? 1 : 0;
  }
  if (cnt) atomicAdd(&rank[i], cnt);
}
__global__ __launch_bounds__(256) void k_order(const K* __restrict__ kp, int32_t* rank) {
  __shared__ OrderKey tile[ORDER_TILE];
  order_block(*kp, rank, blockIdx.x, blockIdx.y, tile);
}
__global__ __launch_bounds__(256) void k_order_scatter(const K* __restrict__ kp, int n, const int32_t* rank, int32_t* order_idx, int patch_stat) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  // the process step charges its bytes to the next counter: the argument block is patched in place instead of being uploaded a second
  // time (classical path; nothing in this kernel reads the field)
  if (i == 0 && patch_stat) const_cast<K*>(kp)->O.stat_bytes = kp->O.stat_bytes + 1;
  if (i >= n || i >= hn(kp->H)) return;
  const int r = rank[i];
  order_idx[r] = i;
  if (kp->spec_hdr) kp->spec_hdr[r] = spec_hdr_of(*kp, i);  // the rounds' view of the entry, by iterator position (kq_spec.hpp)
}

// static part of every head's entry record (kq::rec_fill_static): one thread per (head, flavor-resource slot, path level)
__global__ __launch_bounds__(256) void k_records(const K* __restrict__ kp) {
  const K& k = *kp;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) pack_counts(k);
  if (idx < hn(k.H) * FU * FD) rec_fill_static(k, idx / (FU * FD), idx % (FU * FD));
}

// the entry order and the entry records in one launch: both only read what the nominate pass wrote, and neither reads the other's output
// (k_order_scatter, which reads both, follows) — the first n_order workgroups rank, the others fill records
__global__ __launch_bounds__(256) void k_order_records(const K* __restrict__ kp, int32_t* rank, int nbx, int n_order) {
  __shared__ OrderKey tile[ORDER_TILE];
  const K& k = *kp;
  const int b = blockIdx.x;
  if (b < n_order) { order_block(k, rank, b % nbx, b / nbx, tile); return; }
  const int idx = (b - n_order) * 256 + threadIdx.x;
  if (idx == 0) pack_counts(k);
  if (idx < hn(k.H) * FU * FD) rec_fill_static(k, idx / (FU * FD), idx % (FU * FD));
}

// k_process_spec (speculative parallel rounds over the plain entries of a tree, kq_spec.hpp) is compiled in its own translation unit
// (kq_spec_kernel.hip): it runs in front of k_process, which takes over at K::spec_resume[tree] (nothing left in the common case).
namespace kq { hipError_t launch_process_spec(const K* d, int n_tree, hipStream_t stream); }
// device-side rebuild of the admitted-row structures: kq_rows_kernel.hip (cell kernel + rocPRIM sort / scan)
namespace kq {
hipError_t rows_launch(const DRows& R, int op, int n, hipStream_t stream);
hipError_t rows_sort_pairs(RowsScratch& tmp, uint64_t*& key, int32_t*& val, uint64_t*& key2, int32_t*& val2, int n, int bits, hipStream_t stream);
hipError_t rows_scan_excl(RowsScratch& tmp, const int32_t* in, int32_t* out, int n, hipStream_t stream);
}
// kernels of kq_cycle_run_tas: kq_tas_cycle_kernel.hip
namespace kq {
hipError_t launch_tas_base_k(const TCyc* c, int n, hipStream_t stream);
hipError_t launch_nominate_tas_k(const K* d, int slots, hipStream_t stream);
hipError_t launch_tas_cycle_classes_k(const TCyc* c, int n, hipStream_t stream);
hipError_t launch_process_tas_k(const K* d, size_t want, size_t* attr, hipStream_t stream);
// the same kernels with tas_balanced_placement.go inside the placement (features.TASBalancedPlacement): kq_tas_cycle_kernel_bal.hip, kq_tas_bal_kernel.hip
hipError_t launch_nominate_tas_k_bal(const K* d, int slots, hipStream_t stream);
hipError_t launch_process_tas_k_bal(const K* d, size_t want, size_t* attr, hipStream_t stream);
hipError_t launch_tas_find_bal_k(const TK* d, int slots, hipStream_t stream);
}

constexpr int PROCESS_THREADS = 256;   // wave 0 runs the serial core; all 4 waves prefetch the entry records of a chunk
__global__ __launch_bounds__(PROCESS_THREADS) void k_process(const K* __restrict__ kp, unsigned lds_bytes) {
  const K& k = *kp;
  __shared__ Wave w;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  if (threadIdx.x == 0) { w.cs_lds = nullptr; w.cs_lds_bytes = 0; w.help_on = 0; }  // searches inside k_process use the HBM spill space
  process_tree(k, w, blockIdx.x, blockIdx.x, (int64_t*)dyn_lds, lds_bytes, (int)threadIdx.x, PROCESS_THREADS);
}

// Fair sharing: the iterator pops interleave with processEntry (scheduler.go:358), so ordering and processing
// are one kernel: one wave per root-cohort tree; the tree's cohort usage rows stay in LDS.
#ifndef KQ_FAIR_THREADS
#define KQ_FAIR_THREADS 512
#endif
constexpr int FAIR_THREADS = KQ_FAIR_THREADS;  // wave 0 leads, all waves recompute DRS values between pops (A/B builds: -DKQ_FAIR_THREADS=256 gives the leader 512 registers)
__global__ __launch_bounds__(FAIR_THREADS) void k_process_fair(const K* __restrict__ kp, unsigned lds_bytes, unsigned iter_bytes) {
  const K& k = *kp;
  __shared__ Wave w;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  if (k.help && (int)blockIdx.x >= k.help_trees) {  // helper workgroup: one wave takes victim searches the leaders post (K::help)
    if (threadIdx.x >= 64) return;
    if (threadIdx.x == 0) { w.cs_lds = dyn_lds; w.cs_lds_bytes = (int)lds_bytes; w.help_on = 0; w.bytes = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    helper_main(k, w, blockIdx.x);
    return;
  }
  if (threadIdx.x == 0) { w.cs_lds = nullptr; w.cs_lds_bytes = 0; w.help_on = 0; }
  // [region | one record][the iterator's state (FIter), when it fits]
  process_tree_fair(k, w, blockIdx.x, blockIdx.x, (int64_t*)dyn_lds, lds_bytes - iter_bytes, (int)threadIdx.x, FAIR_THREADS,
                    iter_bytes ? dyn_lds + (lds_bytes - iter_bytes) : nullptr, iter_bytes);
  if (k.help && threadIdx.x == 0) { ag_release(); ag_add_u32(k.help_quit, 1); }  // this tree needs no more help
}
// global iteration positions from the per-tree sequences (kq::fair_rank): 2-D grid like k_order
__global__ __launch_bounds__(256) void k_fair_rank(const K* __restrict__ kp, int32_t* rank) {
  const K& k = *kp;
  const int n = hn(k.H);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || k.X.fs_key[i] < 0) return;
  const int base = blockIdx.y * 256;
  const int cnt = fair_rank(k, i, base, (base + 256) < n ? (base + 256) : n);
  if (cnt) atomicAdd(&rank[i], cnt);
}
__global__ __launch_bounds__(256) void k_fair_rank_apply(const K* __restrict__ kp, const int32_t* rank) {
  const K& k = *kp;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < hn(k.H) && k.X.fs_key[i] >= 0) k.O.order[i] = rank[i];
}

__global__ __launch_bounds__(256) void k_commit_mask(const K* __restrict__ kp, int32_t* use_n_out, int32_t* cq_out, int32_t* fr_out, int64_t* qty_out, int32_t* count) {
  const K& k = *kp;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < k.H.n * KQ_MAXU) commit_keep_cell(k, i, use_n_out, cq_out, fr_out, qty_out, count);
}
// start-of-cycle fills and copies in one launch (kq::DPrep): blockIdx.y = operation
__global__ __launch_bounds__(256) void k_prep(DPrep p) {
  const int o = blockIdx.y;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < p.op[o].words; i += gridDim.x * 256) prep_word(p, o, i);
}
// k_prep with the cycle's argument block riding along: the block arrives as a kernel argument and the first workgroup stores it into
// the device copy the cycle's kernels read — instead of an H2D copy between k_prep and the nominate pass (a blit kernel of its own:
// 3.6 us + 6 us of gaps per cycle at cfg 3, profiles/r06p_cfg3_timeline.txt)
static_assert(sizeof(DPrep) + sizeof(K) + 16 <= 4096, "k_prep_k: the kernel argument segment holds 4 KB");
static_assert(sizeof(K) % 4 == 0, "k_prep_k copies words");
__global__ __launch_bounds__(256) void k_prep_k(DPrep p, K kb, K* dst) {
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    const uint32_t* s = (const uint32_t*)&kb;
    uint32_t* d = (uint32_t*)dst;
    for (unsigned i = threadIdx.x; i < sizeof(K) / 4; i += 256) d[i] = s[i];
  }
  const int o = blockIdx.y;
  if (o < p.n) for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < p.op[o].words; i += gridDim.x * 256) prep_word(p, o, i);
}
// cohort usage re-derived from the ClusterQueue cells, every level in one launch (one workgroup; small populations)
__global__ __launch_bounds__(1024) void k_usage_levels(DSnap S, int64_t* usage, int max_depth) {
  const int cells = S.nc * S.nfr;
  for (int dep = max_depth; dep >= 0; dep--) {
    for (int i = threadIdx.x; i < cells; i += 1024) {
      const int cohort = S.nq + i / S.nfr;
      if (S.depth[cohort] == dep) derive_usage_cell(S, usage, cohort, i % S.nfr);
    }
    __threadfence_block();
    __syncthreads();
  }
}
__global__ __launch_bounds__(64) void k_commit(DSnap S, DCommit c, int add) { commit_tree(S, c, blockIdx.x, add != 0); }
__global__ __launch_bounds__(256) void k_commit_cq(DSnap S, DCommit c, int add) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < c.n * KQ_MAXU) commit_cq_cell(c, S, i / KQ_MAXU, i % KQ_MAXU, add != 0);
}
// the same for larger trees in ONE launch: the levels of a flavor-resource column only depend on that column, so a workgroup takes a
// few columns of every cohort through all levels (barrier between levels) — 1 launch instead of max_depth + 1 (4 x ~6 us at cfg 3)
// A cell is 8 bytes and a node's cells lie side by side (ix = node * nfr + fr): a workgroup that takes 2 columns uses 16 of the 64 bytes of
// every line it touches, and at cfg 3 (111 cohorts, 64 columns) the kernel fetched 6.9 MB to read 1.6 MB (profiles/r07g_cfg3_rocprof_summary.txt).
// Workgroups of up to 1024 threads take whole lines — 8 columns, the lanes of a cell group side by side in fr — still one cell per thread
// and level.
__global__ __launch_bounds__(1024) void k_usage_cols(DSnap S, int64_t* usage, int max_depth, int cols) {
  const int col0 = blockIdx.x * cols;
  const int ncol = (col0 + cols <= S.nfr) ? cols : S.nfr - col0;
  for (int dep = max_depth; dep >= 0; dep--) {
    for (int i = threadIdx.x; i < S.nc * ncol; i += blockDim.x) {
      const int cohort = S.nq + i / ncol;
      if (S.depth[cohort] == dep) derive_usage_cell(S, usage, cohort, col0 + i % ncol);
    }
    __threadfence_block();
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_usage_level(DSnap S, int64_t* usage, int depth) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S.nc * S.nfr) return;
  const int cohort = S.nq + i / S.nfr;
  if (S.depth[cohort] == depth) derive_usage_cell(S, usage, cohort, i % S.nfr);
}

// per-node borrowed sums of the cycle-start plane: one thread per (node, resource)
__global__ __launch_bounds__(256) void k_fs_sums(const K* __restrict__ kp) {
  const K& k = *kp;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k.S.N * k.S.nR) return;
  fs_sums_cell(k, i / k.S.nR, i % k.S.nR);
}
__global__ __launch_bounds__(256) void k_fs_pos(const K* __restrict__ kp) {
  const K& k = *kp;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < k.S.N) fs_pos_node(k, n);
}

// kq_snapshot_derive: ClusterQueue cells, then one launch per cohort depth (deepest first)
__global__ __launch_bounds__(256) void k_derive_cq(DSnap S, DDerive d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < S.nq * S.nfr) derive_cq_cell(S, d, i / S.nfr, i % S.nfr);
}
__global__ __launch_bounds__(256) void k_derive_level(DSnap S, DDerive d, int depth) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S.nc * S.nfr) return;
  const int cohort = S.nq + i / S.nfr;
  if (S.depth[cohort] == depth) derive_cohort_cell(S, d, cohort, i % S.nfr);
}

// TAS: one wavefront per workload (FindTopologyAssignmentsForFlavor), grid-stride over the batch
// (two waves per SIMD: the wide-resource tiers of phase 1 — up to 32 resources per leaf in registers — spill rather than halve the occupancy
// of the usual four-resource case: 182 registers before the tiers existed)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_tas_find(const TK* __restrict__ kp, int slots) {
  const TK& k = *kp;
  // workloads are sorted by request class; a slot walks a contiguous piece so that it rarely changes class
  const int per = (k.Q.n_wl + slots - 1) / slots;
  const int lo = blockIdx.x * per, hi = (lo + per) < k.Q.n_wl ? (lo + per) : k.Q.n_wl;
  for (int i = lo; i < hi; i++) t_workload_t<false>(k, blockIdx.x, k.C.order[i]);   // (the batch kernel's state is in global memory)
}
__global__ __launch_bounds__(64) void k_tas_classes(const TK* __restrict__ kp) { t_class(*kp, blockIdx.x); }
__global__ __launch_bounds__(64) void k_tas_usage(TTopo T, int n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int add) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) t_usage_cell(T, i, leaf, count, spr, add);
}
__global__ __launch_bounds__(64) void k_tas_fits(TTopo T, int n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int32_t* flag) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) t_fits_cell(T, i, leaf, count, spr, flag);
}

// batch admission in entry order: one wavefront (the entries of a TAS flavor are sequentially dependent through the leaf usage)
__global__ __launch_bounds__(64) void k_tas_admit(TTopo T, TAdmit A) { t_admit_seq(T, A); }
__global__ __launch_bounds__(256) void k_tas_delta(TTopo T, int n, const uint8_t* sel, const int32_t* dom_off, const int32_t* dom_leaf, const int32_t* dom_count,
                                                   const int64_t* spr, int64_t* plane) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < n) t_delta_cell(T, p, sel, dom_off, dom_leaf, dom_count, spr, plane);
}
__global__ __launch_bounds__(256) void k_tas_plane_add(TTopo T, const int64_t* plane, int sign) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)T.n_leaves * T.R) t_plane_add_cell(T, i, plane, sign);
}
__global__ __launch_bounds__(256) void k_tas_overflow(TTopo T, const int64_t* plane, uint8_t* over, int32_t* n_over) {
  const int leaf = blockIdx.x * 256 + threadIdx.x;
  if (leaf < T.n_leaves) t_overflow_cell(T, leaf, plane, over, n_over);
}
// tasExclusionStats (kq_tas_exclusion_stats): one thread per (selected podset, leaf); blockIdx.y = the podset
__global__ __launch_bounds__(256) void k_tas_excl(TTopo T, TExcl E) {
  t_excl_cell(T, E, blockIdx.y, blockIdx.x * 256 + threadIdx.x);   // (all lanes: the cell ballots)
}

// pending side on the device (kq_pending.hpp): Heads() = pop per ClusterQueue + compaction + gather; requeue from the decisions
__global__ __launch_bounds__(64) void k_pend_pop(DPend D) { pend_pop(D, blockIdx.x); }
constexpr int PEND_SCAN_THREADS = 1024;
__global__ __launch_bounds__(PEND_SCAN_THREADS) void k_pend_scan(DPend D, DGather G) {
  __shared__ int32_t scan3[3 * PEND_SCAN_THREADS];
  pend_scan(D, G, (int)threadIdx.x, PEND_SCAN_THREADS, scan3);
}
__global__ __launch_bounds__(64) void k_pend_gather(DPend D, DGather G) { if ((int)blockIdx.x < D.counts[0]) pend_gather_head(D, G, blockIdx.x); }
__global__ __launch_bounds__(64) void k_pend_apply(DPend D, DSnap S, DOut O, DHeads H, uint32_t gates, int64_t cycle) {
  pend_apply_head(D, S, O, H, gates, cycle, blockIdx.x);
}
// kq_pending_add: thread i < W0 places resident heap position i, thread W0 + r places arrival r, thread c <= nq writes the new offset of
// ClusterQueue c — all into the second order / offsets buffers
__global__ __launch_bounds__(256) void k_pend_merge(DPend D, const int32_t* ord_old, const int32_t* off_old, int32_t* ord_new, int32_t* off_new,
                                                    const int32_t* fresh, const int32_t* fresh_off, int W0, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i <= D.nq) off_new[i] = off_old[i] + fresh_off[i];
  if (i < W0) pend_merge_old(D, ord_old, ord_new, fresh, fresh_off, i);
  else if (i < W0 + n) pend_merge_new(D, ord_old, off_old, ord_new, fresh, i - W0);
}
__global__ __launch_bounds__(64) void k_pend_add_fix(DPend D, DSnap S, int first) { pend_add_fix(D, S, first + (int)blockIdx.x); }
__global__ __launch_bounds__(64) void k_pend_requeue_at(DPend D, DSnap S, const int32_t* list, const int64_t* at) { pend_requeue_at(D, S, list, at, blockIdx.x); }
__global__ __launch_bounds__(256) void k_pend_update_fix(DPend D, const int32_t* list, const uint8_t* same_gen, int first, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) pend_update_fix(D, list, same_gen, first, i); }
__global__ __launch_bounds__(256) void k_pend_delete(DPend D, const int32_t* list, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) pend_delete(D, list, i); }
// kq_pending_step, after the cycle: blocks [0, nb) fold the admissions into the snapshot and keep the rows for the release
// (commit_fused_cell), blocks [nb, nb + n) run the requeue policy of one head each (wave 0 of the block) — one launch instead of three
__global__ __launch_bounds__(256) void k_step_commit_apply(const K* __restrict__ kp, DSnap S, DCommit c, DPend D, uint32_t gates, int64_t cycle, int nb) {
  if ((int)blockIdx.x < nb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < c.n * KQ_MAXU) commit_fused_cell(*kp, S, c, i);
    return;
  }
  if (threadIdx.x >= 64) return;
  pend_apply_head(D, S, kp->O, kp->H, gates, cycle, (int)blockIdx.x - nb);
}
// ... and the release of an older commit: removeUsage of its rows + the stamp of the trees whose quota was freed, one launch
__global__ __launch_bounds__(256) void k_step_release(DSnap S, DCommit c, int32_t* tree_stamp, int32_t stamp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c.n * KQ_MAXU) return;
  commit_cq_cell(c, S, i / KQ_MAXU, i % KQ_MAXU, false);
  if (i % KQ_MAXU == 0) pend_release_mark(S, tree_stamp, c.cq, c.use_n, c.n, i / KQ_MAXU, stamp);
}
__global__ __launch_bounds__(256) void k_afs_usage(DPend D, int init_f64) { const int l = blockIdx.x * 256 + threadIdx.x; if (l < D.A.n_lq) afs_init_lq(D.A, l, init_f64 != 0); }
__global__ __launch_bounds__(64) void k_afs_sub(DPend D, const int32_t* list, int n) { if (threadIdx.x == 0) afs_sub_list(D, list, n); }
__global__ __launch_bounds__(256) void k_afs_set_consumed(DPend D, const int32_t* lq, const uint64_t* lo, const int64_t* hi, const double* f64, const int32_t* settle, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) afs_set_consumed(D.A, D.lq, lq, lo, hi, f64, settle, i);
}
__global__ __launch_bounds__(64) void k_pend_qi(DPend D, const int32_t* list) { pend_queue_inadmissible(D, list ? list[blockIdx.x] : (int)blockIdx.x); }

__global__ __launch_bounds__(256) void k_pend_release_mark(DSnap S, int32_t* tree_stamp, const int32_t* cq, const int32_t* use_n, int n, int32_t stamp) {
  pend_release_mark(S, tree_stamp, cq, use_n, n, blockIdx.x * 256 + threadIdx.x, stamp);
}
__global__ __launch_bounds__(64) void k_pend_release_requeue(DPend D, DSnap S, const int32_t* tree_stamp, int32_t stamp) {
  pend_release_requeue(D, S, tree_stamp, blockIdx.x, stamp);
}

// sharded nominate: export of this rank's nomination into the exchange buffer / import of the merged one (kq_device.hpp DShard)
__global__ __launch_bounds__(256) void k_shard_export(const K* __restrict__ kp, size_t nps_total, int rsn_win) {
  const K& k = *kp;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < k.H.n) shard_export_head(k, i, nps_total);
  if (i == 0) shard_export_misc(k, nps_total, rsn_win);
  shard_export_pool(k, nps_total, rsn_win, i);
}
__global__ __launch_bounds__(256) void k_shard_import(const K* __restrict__ kp, size_t nps_total, int rsn_win) {
  const K& k = *kp;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < k.H.n) shard_import_head(k, i, nps_total);
  if (i == 0) shard_import_misc(k, nps_total);
  shard_import_pool(k, nps_total, rsn_win, i);
}
__global__ __launch_bounds__(256) void k_usage_delta(int64_t* out, const int64_t* work, const int64_t* start, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) usage_delta_cell(out, work, start, i);
}
__global__ __launch_bounds__(256) void k_usage_add(int64_t* usage, const int64_t* delta, size_t n, int sign, int32_t* big) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) usage_add_cell(usage, delta, i, sign, big);
}

namespace kq {
struct HipBackend {
  hipStream_t stream = nullptr;
  hipEvent_t ev[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};  // phase timers, one set per step in flight
  hipEvent_t sev[2] = {nullptr, nullptr};   // "everything of this asynchronous step is done" (kq_pending_step_wait)
  int stage = 0;                            // which of the two sets the calls below use
  // Asynchronous steps overlap what does not depend on each other on two side streams: the argument-block upload and the decisions'
  // D2H on `cstream`, the re-derivation of the cohort usage levels on `ustream` (next to Heads()); events order them with `stream`.
  hipStream_t cstream = nullptr, ustream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_usage = nullptr, ev_kup[2] = {nullptr, nullptr}, ev_cycle[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
  bool copied_pending[2] = {false, false};  // ev_copied[i] was recorded and nobody has ordered `stream` behind it yet
  bool in_step = false;                     // between step_begin / step_end: side streams in use
  // Measured at cfg 3 (profiles/r03h_*): 0.428 ms per cycle with the side streams against 0.416 ms with everything on the one stream — the
  // event waits cost 8-13 us of gap each and the overlapped kernels slow each other down; so this is an experiment, off unless
  // KQ_STEP_SIDE_STREAMS is set (the GPU suite runs it once).
  bool side_off = getenv("KQ_STEP_SIDE_STREAMS") == nullptr;
  int device = 0;
  int n_cu = 256;
  hipError_t err = hipSuccess;
  std::string msg;
  RowsScratch rows_tmp;                     // rocPRIM temporary storage of this engine's row sorts / scans (kq_rows_kernel.hip)

  void chk(hipError_t e, const char* what) {
    if (e != hipSuccess && err == hipSuccess) { err = e; msg = std::string(what) + ": " + hipGetErrorString(e); }
  }
  int init(int dev) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) { msg = "no HIP device available"; return KQ_ENODEVICE; }
    if (dev < 0 || dev >= count) { msg = "device ordinal out of range"; return KQ_EINVAL; }
    device = dev;
    chk(hipSetDevice(dev), "hipSetDevice");
    hipDeviceProp_t prop;
    chk(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    if (err == hipSuccess) n_cu = prop.multiProcessorCount;
    chk(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
    for (auto& set : ev) for (auto& e2 : set) chk(hipEventCreate(&e2), "hipEventCreate");
    for (auto& e2 : sev) chk(hipEventCreateWithFlags(&e2, hipEventDisableTiming), "hipEventCreate");
    chk(hipStreamCreateWithFlags(&cstream, hipStreamNonBlocking), "hipStreamCreate");
    chk(hipStreamCreateWithFlags(&ustream, hipStreamNonBlocking), "hipStreamCreate");
    for (hipEvent_t* e2 : {&ev_begin, &ev_usage, &ev_kup[0], &ev_kup[1], &ev_cycle[0], &ev_cycle[1], &ev_copied[0], &ev_copied[1]})
      chk(hipEventCreateWithFlags(e2, hipEventDisableTiming), "hipEventCreate");
    chk(hipHostMalloc((void**)&hk, 4 * sizeof(K), hipHostMallocDefault), "hipHostMalloc K");
    return err == hipSuccess ? KQ_OK : KQ_EDEVICE;
  }
  void destroy() {
    for (auto& set : ev) for (auto& e2 : set) if (e2) (void)hipEventDestroy(e2);
    for (auto& e2 : sev) if (e2) (void)hipEventDestroy(e2);
    for (hipEvent_t e2 : {ev_begin, ev_usage, ev_kup[0], ev_kup[1], ev_cycle[0], ev_cycle[1], ev_copied[0], ev_copied[1]}) if (e2) (void)hipEventDestroy(e2);
    if (cstream) (void)hipStreamDestroy(cstream);
    if (ustream) (void)hipStreamDestroy(ustream);
    for (auto& d : dk2) if (d) (void)hipFree(d);
    if (hk) (void)hipHostFree(hk);
    for (auto& d : dk) if (d) (void)hipFree(d);
    if (dtk) (void)hipFree(dtk);
    if (rows_tmp.p) (void)hipFree(rows_tmp.p);
    if (stream) (void)hipStreamDestroy(stream);
  }
  // KQ_GUARD=1 (tools/fuzz_put_guard.py): every device buffer of the engine sits between two 256-byte guard zones filled with 0xC7 — the
  // tail zone starts at the first byte behind the requested size — and kq_debug_check_guards reads them all back. An out-of-bounds
  // write of a kernel or a copy (the suspect behind the memory-access fault round 3 / 4 saw once inside kq_snapshot_put) lands in a
  // guard instead of in a neighbouring allocation, on every box, whatever the allocator's layout.
  static constexpr size_t GW = 256;
  struct GuardRec { char* base; size_t n; };
  std::vector<GuardRec> guards;
  bool guard_on = getenv("KQ_GUARD") != nullptr;
  // KQ_EFENCE=1 (debugging only): every buffer gets a region of its own, a multiple of 2 MiB (the granularity the memory-access faults
  // of r05l-r05n were reported at), and sits at the END of it, 16-byte aligned — a kernel that reads or writes more than the alignment
  // slack past a buffer faults at once, on every box, instead of when the allocator happens to put the buffer last in a mapped chunk.
  // The body is poisoned like KQ_GUARD's.
  // What a fresh buffer holds under KQ_GUARD / KQ_EFENCE: 0xA5 bytes by default; KQ_POISON=small fills it with 32-bit words in [0, 300) —
  // what the pages of a long-lived process typically hold (counts, indices of an earlier engine) and what made k_order_scatter walk off
  // its records in round 5: garbage that LOOKS valid. KQ_POISON=<hex byte> for any other constant.
  void poison(void* p, size_t n) {
    if (!n) return;
    const char* mode = getenv("KQ_POISON");
    if (mode && !strcmp(mode, "small")) {
      static std::vector<uint32_t> pat;
      if (pat.empty()) { pat.resize(1 << 18); uint32_t x = 2463534242u; for (auto& v : pat) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = x % 300u; } }
      for (size_t o = 0; o < n; o += pat.size() * 4) chk(hipMemcpy((char*)p + o, pat.data(), std::min(n - o, pat.size() * 4), hipMemcpyHostToDevice), "poison fill");
      return;
    }
    chk(hipMemset(p, mode ? (int)strtol(mode, nullptr, 16) : 0xA5, n), "poison fill");
  }
  bool efence_on = getenv("KQ_EFENCE") != nullptr;
  struct FenceRec { char* base; char* user; };
  std::vector<FenceRec> fences;
  void* alloc(size_t n) {
    if (efence_on) {
      const size_t CH = (size_t)2 << 20, body = (n + 15) & ~(size_t)15, total = ((body + CH - 1) / CH) * CH + (body == 0 ? CH : 0);
      char* p = nullptr;
      chk(hipMalloc((void**)&p, total), "hipMalloc");
      if (!p) return nullptr;
      char* u = p + total - body;
      poison(u, n);
      fences.push_back(FenceRec{p, u});
      return u;
    }
    if (!guard_on) { void* p = nullptr; chk(hipMalloc(&p, n), "hipMalloc"); return p; }
    char* p = nullptr;
    const size_t body = (n + 255) & ~(size_t)255;
    chk(hipMalloc((void**)&p, body + 2 * GW), "hipMalloc");
    if (!p) return nullptr;
    chk(hipMemset(p, 0xC7, GW), "guard fill");
    chk(hipMemset(p + GW + n, 0xC7, body - n + GW), "guard fill");
    // ... and the buffer itself is POISONED (0xA5, as the emulation's allocator does): hipMalloc hands out whatever the previous owner of
    // the pages left — zeroes in a fresh process, garbage in a long-lived one — so code that only works on zero-initialised memory
    // passes every short test and faults in a controller that has been up for a day (or in the 400th test of a pytest worker)
    poison(p + GW, n);
    guards.push_back(GuardRec{p, n});
    return p + GW;
  }
  void free(void* p) {
    if (efence_on && p) {
      for (size_t i = 0; i < fences.size(); i++)
        if (fences[i].user == (char*)p) { (void)hipFree(fences[i].base); fences[i] = fences.back(); fences.pop_back(); return; }
      (void)hipFree(p);
      return;
    }
    if (!guard_on || !p) { (void)hipFree(p); return; }
    for (size_t i = 0; i < guards.size(); i++)
      if (guards[i].base + GW == (char*)p) { (void)hipFree(guards[i].base); guards[i] = guards.back(); guards.pop_back(); return; }
    (void)hipFree(p);
  }
  // out3: buffers checked, buffers with a damaged guard, guard bytes read. The text names the first damaged buffers.
  int check_guards(int64_t* out3, std::string* text) {
    out3[0] = out3[1] = out3[2] = 0;
    if (!guard_on) { *text = "KQ_GUARD is not set"; return KQ_EUNSUPPORTED; }
    if (hipDeviceSynchronize() != hipSuccess) { *text = "device error before the guard check"; return KQ_EDEVICE; }
    std::vector<unsigned char> h;
    for (const GuardRec& g : guards) {
      const size_t body = (g.n + 255) & ~(size_t)255, tail = body - g.n + GW;
      h.resize(GW + tail);
      if (hipMemcpy(h.data(), g.base, GW, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h.data() + GW, g.base + GW + g.n, tail, hipMemcpyDeviceToHost) != hipSuccess) { *text = "guard read failed"; return KQ_EDEVICE; }
      out3[0]++; out3[2] += (int64_t)h.size();
      long first = -1; size_t bad = 0;
      for (size_t i = 0; i < h.size(); i++) if (h[i] != 0xC7) { bad++; if (first < 0) first = (long)i; }
      if (bad) {
        out3[1]++;
        if (text->size() < 600) *text += "buffer of " + std::to_string(g.n) + " B: " + std::to_string(bad) + " guard bytes overwritten, first at " +
                                         (first < (long)GW ? std::to_string(first - (long)GW) : "+" + std::to_string(first - (long)GW)) + " from its " + (first < (long)GW ? "start" : "end") + "; ";
      }
    }
    return KQ_OK;
  }
  void* alloc_host(size_t n) { void* p = nullptr; chk(hipHostMalloc(&p, n, hipHostMallocDefault), "hipHostMalloc"); return p; }
  void free_host(void* p) { (void)hipHostFree(p); }
  void h2d(void* d, const void* h, size_t n) { chk(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, stream), "h2d"); }
  void d2h(void* h, const void* d, size_t n) { chk(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, stream), "d2h"); }
  void d2d(void* d, const void* s, size_t n) { chk(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, stream), "d2d"); }
  void memset(void* d, int v, size_t n) { chk(hipMemsetAsync(d, v, n, stream), "memset"); }
  int sync() {
    chk(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (copied_pending[0] || copied_pending[1]) { chk(hipStreamSynchronize(cstream), "hipStreamSynchronize"); copied_pending[0] = copied_pending[1] = false; }
    if (err != hipSuccess) { hipError_t e = err; (void)e; err = hipSuccess; (void)hipGetLastError(); return KQ_EDEVICE; }
    return KQ_OK;
  }
  const char* error() { return msg.c_str(); }
  int max_slots() { return n_cu * 16; }  // 16 one-wave workgroups per CU (4 per SIMD)
  long long bal_dp() { return (long long)1 << 20; }   // states of a balanced placement's dynamic programme per wave slot (8 MB)
  size_t lds_budget() { return 160 * 1024 - sizeof(Wave) - 256; }  // dynamic LDS a workgroup can get next to its static Wave
  // HIP events on the engine's own stream bracket each kernel (live per-kernel duration)
  void timer_mark(int i) { chk(hipEventRecord(ev[stage][i], stream), "hipEventRecord"); }
  double timer_ms(int a, int b) { float ms = 0; chk(hipEventElapsedTime(&ms, ev[stage][a], ev[stage][b]), "hipEventElapsedTime"); return ms; }
  // asynchronous steps: two in flight at most, each with its own timers, K staging and completion event
  void stage_select(int i) { stage = i & 1; }
  void stage_mark() { chk(hipEventRecord(sev[stage], stream), "hipEventRecord"); }
  int stage_wait() {
    chk(hipEventSynchronize(sev[stage]), "hipEventSynchronize");
    if (copied_pending[stage]) { chk(hipEventSynchronize(ev_copied[stage]), "hipEventSynchronize"); copied_pending[stage] = false; }   // the decisions' copy ran on the side stream
    if (err != hipSuccess) { err = hipSuccess; (void)hipGetLastError(); return KQ_EDEVICE; }
    return KQ_OK;
  }
  // An asynchronous step starts: whatever the previous steps' side copies still read (the packed outputs, the popped heads) must not
  // be overwritten before they are through, and the side streams start behind everything already enqueued on `stream`.
  void step_begin(int i) {
    stage_select(i);
    if (side_off) return;
    in_step = true;
    for (int q = 0; q < 2; q++) if (copied_pending[q]) chk(hipStreamWaitEvent(stream, ev_copied[q], 0), "hipStreamWaitEvent");
    chk(hipEventRecord(ev_begin, stream), "hipEventRecord");
  }
  void step_end() { in_step = false; stage_select(0); }
  // cohort usage levels on the side stream (they only depend on what was enqueued before the step), joined before the cycle reads them
  void usage_levels_side(const DSnap& S, int64_t* usage, int max_depth) {
    if (!in_step) { launch_usage_levels(S, usage, max_depth); return; }
    chk(hipStreamWaitEvent(ustream, ev_begin, 0), "hipStreamWaitEvent");
    hipStream_t keep = stream; stream = ustream;
    launch_usage_levels(S, usage, max_depth);
    stream = keep;
    chk(hipEventRecord(ev_usage, ustream), "hipEventRecord");
    usage_side = true;
  }
  bool usage_side = false;
  void usage_join() { if (usage_side) chk(hipStreamWaitEvent(stream, ev_usage, 0), "hipStreamWaitEvent"); usage_side = false; }
  // the step's outputs to pinned host memory on the side stream: the tail kernels of the step do not wait for the copies
  void side_fence() {
    if (!in_step) return;
    chk(hipEventRecord(ev_cycle[stage], stream), "hipEventRecord");
    chk(hipStreamWaitEvent(cstream, ev_cycle[stage], 0), "hipStreamWaitEvent");
  }
  void d2h_side(void* h, const void* d, size_t n) { chk(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, in_step ? cstream : stream), "d2h"); }
  void side_done() { if (in_step) { chk(hipEventRecord(ev_copied[stage], cstream), "hipEventRecord"); copied_pending[stage] = true; } }
  K* dk[2] = {nullptr, nullptr};   // device copies of the argument block (nominate/order, process)
  K* dk2[4] = {nullptr, nullptr, nullptr, nullptr};   // the same per step in flight [stage][which]: uploaded on the side stream ahead of time
  const K* dcur0 = nullptr;        // the block of the nominate / order step of the cycle being enqueued
  K* hk = nullptr;                 // pinned staging [stage][which]: a pageable source would make hipMemcpyAsync wait for the stream
  const K* put_k(const K& k, int which) {
    K* h = hk + stage * 2 + which;
    memcpy((void*)h, (const void*)&k, sizeof(K));
    K* d;
    if (in_step) {
      K*& dd = dk2[stage * 2 + which];
      if (!dd) chk(hipMalloc((void**)&dd, sizeof(K)), "hipMalloc K");
      d = dd;
      chk(hipMemcpyAsync(d, h, sizeof(K), hipMemcpyHostToDevice, cstream), "memcpy K");
      chk(hipEventRecord(ev_kup[stage], cstream), "hipEventRecord");
      chk(hipStreamWaitEvent(stream, ev_kup[stage], 0), "hipStreamWaitEvent");
    } else {
      if (!dk[which]) chk(hipMalloc((void**)&dk[which], sizeof(K)), "hipMalloc K");
      d = dk[which];
      chk(hipMemcpyAsync(d, h, sizeof(K), hipMemcpyHostToDevice, stream), "memcpy K");
    }
    if (which == 0) dcur0 = d;
    return d;
  }
  TK* dtk = nullptr;
  TK htk;
  void put_tk(const TK& k) {
    if (!dtk) chk(hipMalloc((void**)&dtk, sizeof(TK)), "hipMalloc TK");
    htk = k;
    chk(hipMemcpyAsync(dtk, &htk, sizeof(TK), hipMemcpyHostToDevice, stream), "memcpy TK");
  }
  void launch_tas_classes(const TK& k) {
    put_tk(k);
    hipLaunchKernelGGL(k_tas_classes, dim3(k.C.n), dim3(64), 0, stream, (const TK*)dtk);
    chk(hipGetLastError(), "k_tas_classes");
  }
  void launch_tas_find(const TK& k, int slots) {
    put_tk(k);
    if (k.T.balanced) { chk(launch_tas_find_bal_k((const TK*)dtk, slots, stream), "k_tas_find_bal"); return; }   // (the gate's own kernel)
    hipLaunchKernelGGL(k_tas_find, dim3(slots), dim3(64), 0, stream, (const TK*)dtk, slots);
    chk(hipGetLastError(), "k_tas_find");
  }
  void launch_tas_usage(const TTopo& T, int n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int add) {
    hipLaunchKernelGGL(k_tas_usage, dim3((n + 63) / 64), dim3(64), 0, stream, T, n, leaf, count, spr, add);
    chk(hipGetLastError(), "k_tas_usage");
  }
  void launch_tas_fits(const TTopo& T, int n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int32_t* flag) {
    hipLaunchKernelGGL(k_tas_fits, dim3((n + 63) / 64), dim3(64), 0, stream, T, n, leaf, count, spr, flag);
    chk(hipGetLastError(), "k_tas_fits");
  }
  void launch_tas_admit(const TTopo& T, const TAdmit& A) {
    hipLaunchKernelGGL(k_tas_admit, dim3(1), dim3(64), 0, stream, T, A);
    chk(hipGetLastError(), "k_tas_admit");
  }
  void launch_tas_delta(const TTopo& T, int n, const uint8_t* sel, const int32_t* dom_off, const int32_t* dom_leaf, const int32_t* dom_count, const int64_t* spr, int64_t* plane) {
    hipLaunchKernelGGL(k_tas_delta, dim3((n + 255) / 256), dim3(256), 0, stream, T, n, sel, dom_off, dom_leaf, dom_count, spr, plane);
    chk(hipGetLastError(), "k_tas_delta");
  }
  void launch_tas_plane_add(const TTopo& T, const int64_t* plane, int sign) {
    const size_t n = (size_t)T.n_leaves * T.R;
    hipLaunchKernelGGL(k_tas_plane_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, T, plane, sign);
    chk(hipGetLastError(), "k_tas_plane_add");
  }
  void launch_tas_overflow(const TTopo& T, const int64_t* plane, uint8_t* over, int32_t* n_over) {
    hipLaunchKernelGGL(k_tas_overflow, dim3((T.n_leaves + 255) / 256), dim3(256), 0, stream, T, plane, over, n_over);
    chk(hipGetLastError(), "k_tas_overflow");
  }
  void launch_tas_excl(const TTopo& T, const TExcl& E) {
    hipLaunchKernelGGL(k_tas_excl, dim3((T.n_leaves + 255) / 256, E.n_sel), dim3(256), 0, stream, T, E);
    chk(hipGetLastError(), "k_tas_excl");
  }
  void launch_commit_mask(int n, int32_t* use_n_out, int32_t* cq_out, int32_t* fr_out, int64_t* qty_out, int32_t* count) {  // uses the K block of the last cycle
    hipLaunchKernelGGL(k_commit_mask, dim3((n * KQ_MAXU + 255) / 256), dim3(256), 0, stream, dproc, use_n_out, cq_out, fr_out, qty_out, count);
    chk(hipGetLastError(), "k_commit_mask");
  }
  // ClusterQueue-level cells only; the cohort levels follow from them (launch_usage_levels, deferred by the host)
  void launch_commit_cells(const DSnap& S, const DCommit& c, bool add) {
    hipLaunchKernelGGL(k_commit_cq, dim3((c.n * KQ_MAXU + 255) / 256), dim3(256), 0, stream, S, c, add ? 1 : 0);
    chk(hipGetLastError(), "k_commit_cq");
  }
  void launch_commit_trees(const DSnap& S, const DCommit& c, bool add) {
    if (S.n_tree > 0) hipLaunchKernelGGL(k_commit, dim3(S.n_tree), dim3(64), 0, stream, S, c, add ? 1 : 0);
    chk(hipGetLastError(), "k_commit");
  }
  void launch_usage_levels(const DSnap& S, int64_t* usage, int max_depth) {
    const int cells = S.nc * S.nfr;
    if (cells == 0) return;
    // one workgroup for all levels only while every thread has at most one cell per level; beyond that a launch per level
    // (many workgroups) is faster than the serial passes of one (measured: 75 us vs 4 x 6.5 us at 7104 cells)
    if (cells <= 1024) hipLaunchKernelGGL(k_usage_levels, dim3(1), dim3(1024), 0, stream, S, usage, max_depth);
    else if (S.nc <= 4096) {  // one launch: a workgroup per group of columns (as many as keep one cell per thread and level)
      int nt = 256, cols = std::max(1, std::min(S.nfr, nt / std::max(S.nc, 1)));
      if (cols < 8 && S.nfr >= 8 && S.nc * 8 <= 1024 * 2) { nt = std::min(1024, ((S.nc * 8 + 63) / 64) * 64); cols = 8; }   // whole 64-byte lines; at most two cells per thread and level
      else if (cols > 8) cols &= ~7;
      hipLaunchKernelGGL(k_usage_cols, dim3((S.nfr + cols - 1) / cols), dim3(nt), 0, stream, S, usage, max_depth, cols);
    }
    else for (int dep = max_depth; dep >= 0; dep--)
      hipLaunchKernelGGL(k_usage_level, dim3((cells + 255) / 256), dim3(256), 0, stream, S, usage, dep);
    chk(hipGetLastError(), "k_usage_level");
  }
  void launch_prep(const DPrep& p) {
    uint32_t mx = 0;
    for (int o = 0; o < p.n; o++) mx = std::max(mx, p.op[o].words);
    if (p.n == 0 || mx == 0) return;
    const unsigned gx = std::min<unsigned>((mx + 1023) / 1024, 256);
    hipLaunchKernelGGL(k_prep, dim3(gx, p.n), dim3(256), 0, stream, p);
    chk(hipGetLastError(), "k_prep");
  }
  // launch_prep + the upload of the nominate step's argument block in one launch (launch_nominate takes the block from here)
  static constexpr bool FUSE_PREP_K = true;
  bool k0_ready = false;
  void launch_prep_k(const DPrep& p, const K& k) {
    if (in_step) { launch_prep(p); return; }   // (side streams: the block travels on the copy stream)
    uint32_t mx = 0;
    for (int o = 0; o < p.n; o++) mx = std::max(mx, p.op[o].words);
    const unsigned gx = std::max(1u, std::min<unsigned>((mx + 1023) / 1024, 256));
    if (!dk[0]) chk(hipMalloc((void**)&dk[0], sizeof(K)), "hipMalloc K");
    hipLaunchKernelGGL(k_prep_k, dim3(gx, std::max(p.n, 1)), dim3(256), 0, stream, p, k, dk[0]);
    chk(hipGetLastError(), "k_prep_k");
    dcur0 = dk[0]; k0_ready = true;
  }
  void launch_derive(const DSnap& S, const DDerive& d, int max_depth) {
    if (S.nq * S.nfr > 0) hipLaunchKernelGGL(k_derive_cq, dim3((S.nq * S.nfr + 255) / 256), dim3(256), 0, stream, S, d);
    if (S.nc * S.nfr > 0)
      for (int dep = max_depth; dep >= 0; dep--)
        hipLaunchKernelGGL(k_derive_level, dim3((S.nc * S.nfr + 255) / 256), dim3(256), 0, stream, S, d, dep);
    chk(hipGetLastError(), "k_derive");
  }
  void launch_fs_sums(const K& k) {
    const K* d = put_k(k, 0);
    hipLaunchKernelGGL(k_fs_sums, dim3((k.S.N * k.S.nR + 255) / 256), dim3(256), 0, stream, d);
    hipLaunchKernelGGL(k_fs_pos, dim3((k.S.N + 255) / 256), dim3(256), 0, stream, d);
    chk(hipGetLastError(), "k_fs_sums");
  }
  void launch_pend_heads(const DPend& D, const DGather& G) {
    if (D.nq == 0) return;
    hipLaunchKernelGGL(k_pend_pop, dim3(D.nq), dim3(64), 0, stream, D);
    hipLaunchKernelGGL(k_pend_scan, dim3(1), dim3(PEND_SCAN_THREADS), 0, stream, D, G);
    hipLaunchKernelGGL(k_pend_gather, dim3(D.nq), dim3(64), 0, stream, D, G);
    chk(hipGetLastError(), "k_pend_heads");
  }
  void launch_step_commit_apply(const DSnap& S, const DCommit& c, const DPend& D, uint32_t gates, int64_t cycle) {
    const int nb = (c.n * KQ_MAXU + 255) / 256;
    hipLaunchKernelGGL(k_step_commit_apply, dim3(nb + c.n), dim3(256), 0, stream, dproc, S, c, D, gates, cycle, nb);
    chk(hipGetLastError(), "k_step_commit_apply");
  }
  void launch_step_release(const DSnap& S, const DCommit& c, const DPend& D, int32_t* tree_stamp, int32_t stamp) {
    hipLaunchKernelGGL(k_step_release, dim3((c.n * KQ_MAXU + 255) / 256), dim3(256), 0, stream, S, c, tree_stamp, stamp);
    hipLaunchKernelGGL(k_pend_release_requeue, dim3(D.nq), dim3(64), 0, stream, D, S, (const int32_t*)tree_stamp, stamp);
    chk(hipGetLastError(), "k_step_release");
  }
  void launch_afs_usage(const DPend& D, bool init_f64) {
    if (D.A.n_lq > 0) hipLaunchKernelGGL(k_afs_usage, dim3((D.A.n_lq + 255) / 256), dim3(256), 0, stream, D, init_f64 ? 1 : 0);
    chk(hipGetLastError(), "k_afs_usage");
  }
  void launch_afs_sub(const DPend& D, const int32_t* list, int n) {
    hipLaunchKernelGGL(k_afs_sub, dim3(1), dim3(64), 0, stream, D, list, n);
    chk(hipGetLastError(), "k_afs_sub");
  }
  void launch_afs_set_consumed(const DPend& D, const int32_t* lq, const uint64_t* lo, const int64_t* hi, const double* f64, const int32_t* settle, int n) {
    hipLaunchKernelGGL(k_afs_set_consumed, dim3((n + 255) / 256), dim3(256), 0, stream, D, lq, lo, hi, f64, settle, n);
    chk(hipGetLastError(), "k_afs_set_consumed");
  }
  void launch_pend_apply(const DPend& D, const DSnap& S, const DOut& O, const DHeads& H, uint32_t gates, int64_t cycle, int n) {
    hipLaunchKernelGGL(k_pend_apply, dim3(n), dim3(64), 0, stream, D, S, O, H, gates, cycle);
    chk(hipGetLastError(), "k_pend_apply");
  }
  void launch_pend_merge(const DPend& D, const int32_t* ord_old, const int32_t* off_old, int32_t* ord_new, int32_t* off_new,
                         const int32_t* fresh, const int32_t* fresh_off, int W0, int n) {
    const int items = std::max(W0 + n, D.nq + 1);
    hipLaunchKernelGGL(k_pend_merge, dim3((items + 255) / 256), dim3(256), 0, stream, D, ord_old, off_old, ord_new, off_new, fresh, fresh_off, W0, n);
    chk(hipGetLastError(), "k_pend_merge");
  }
  void launch_pend_add_fix(const DPend& D, const DSnap& S, int first, int n) {
    if (n > 0) hipLaunchKernelGGL(k_pend_add_fix, dim3(n), dim3(64), 0, stream, D, S, first);
    chk(hipGetLastError(), "k_pend_add_fix");
  }
  void launch_pend_requeue_at(const DPend& D, const DSnap& S, const int32_t* list, const int64_t* at, int n) {
    if (n > 0) hipLaunchKernelGGL(k_pend_requeue_at, dim3(n), dim3(64), 0, stream, D, S, list, at);
    chk(hipGetLastError(), "k_pend_requeue_at");
  }
  void launch_pend_update_fix(const DPend& D, const int32_t* list, const uint8_t* same_gen, int first, int n) {
    if (n > 0) hipLaunchKernelGGL(k_pend_update_fix, dim3((n + 255) / 256), dim3(256), 0, stream, D, list, same_gen, first, n);
    chk(hipGetLastError(), "k_pend_update_fix");
  }
  void launch_pend_delete(const DPend& D, const int32_t* list, int n) {
    if (n > 0) hipLaunchKernelGGL(k_pend_delete, dim3((n + 255) / 256), dim3(256), 0, stream, D, list, n);
    chk(hipGetLastError(), "k_pend_delete");
  }
  void launch_pend_qi(const DPend& D, const int32_t* list, int n) {
    if (n > 0) hipLaunchKernelGGL(k_pend_qi, dim3(n), dim3(64), 0, stream, D, list);
    chk(hipGetLastError(), "k_pend_qi");
  }
  void launch_pend_release(const DPend& D, const DSnap& S, int32_t* tree_stamp, const int32_t* cq, const int32_t* use_n, int n, int32_t stamp) {
    if (n <= 0 || D.nq == 0) return;
    hipLaunchKernelGGL(k_pend_release_mark, dim3((n + 255) / 256), dim3(256), 0, stream, S, tree_stamp, cq, use_n, n, stamp);
    hipLaunchKernelGGL(k_pend_release_requeue, dim3(D.nq), dim3(64), 0, stream, D, S, (const int32_t*)tree_stamp, stamp);
    chk(hipGetLastError(), "k_pend_release");
  }
  void launch_usage_delta(int64_t* out, const int64_t* work, const int64_t* start, size_t n) {
    if (n) hipLaunchKernelGGL(k_usage_delta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, work, start, n);
    chk(hipGetLastError(), "k_usage_delta");
  }
  void launch_usage_add(int64_t* usage, const int64_t* delta, size_t n, int sign, int32_t* big) {
    if (n) hipLaunchKernelGGL(k_usage_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, usage, delta, n, sign, big);
    chk(hipGetLastError(), "k_usage_add");
  }
  void launch_shard_export(const K& k, size_t nps_total, int rsn_win) {
    const int items = std::max(k.H.n, k.shard.pool_cap);
    hipLaunchKernelGGL(k_shard_export, dim3((items + 255) / 256), dim3(256), 0, stream, dcur0, nps_total, rsn_win);
    chk(hipGetLastError(), "k_shard_export");
  }
  void launch_shard_import(const K& k, size_t nps_total, int rsn_win) {
    const K* d = put_k(k, 0);
    const int items = std::max(k.H.n, k.shard.world * k.shard.pool_cap);
    hipLaunchKernelGGL(k_shard_import, dim3((items + 255) / 256), dim3(256), 0, stream, d, nps_total, rsn_win);
    chk(hipGetLastError(), "k_shard_import");
  }
  size_t lds_attr_nom = 0;
  void launch_nominate(const K& k, int slots, size_t lds, bool full_pass) {
    stat_patched = false;
    if (lds > 48 * 1024 && lds != lds_attr_nom) {
      chk(hipFuncSetAttribute((const void*)k_nominate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
      lds_attr_nom = lds;
    }
    // (timing builds: KQ_PROF_SKIP_NOMINATE=1 sends the nominate kernels' segment counters to a sink, so that the shared search code's
    //  timers show the process kernel alone)
    static const bool prof_skip_nom = getenv("KQ_PROF_SKIP_NOMINATE") != nullptr;
    K kk = k;
    if (prof_skip_nom && kk.prof) kk.prof += 64;
    const K* d = (k0_ready && !prof_skip_nom) ? dcur0 : put_k(kk, 0);   // (k0_ready: launch_prep_k stored this block)
    k0_ready = false;
    hipLaunchKernelGGL(k_nominate_lean, dim3(slots), dim3(64), 0, stream, d, slots);
    if (full_pass && k.sim_nscan) {
      if (lds > 48 * 1024 && lds != lds_attr_sim) {
        chk(hipFuncSetAttribute((const void*)k_nominate_sim, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
        lds_attr_sim = lds;
      }
      hipLaunchKernelGGL(k_nominate_sim, dim3(slots), dim3(64), lds, stream, d, (unsigned)lds, 0);
      for (int r = 0; r < SIM_ROUNDS; r++) {
        hipLaunchKernelGGL(k_nominate_emit, dim3(slots), dim3(64), 0, stream, d, r);
        hipLaunchKernelGGL(k_nominate_sim, dim3(slots), dim3(64), lds, stream, d, (unsigned)lds, r + 1);
      }
    }
    if (full_pass) hipLaunchKernelGGL(k_nominate, dim3(slots), dim3(64), lds, stream, d, slots, (unsigned)lds);
    chk(hipGetLastError(), "k_nominate");
  }
  void launch_records(const K& k) {
    if (k.H.n == 0) return;
    hipLaunchKernelGGL(k_records, dim3((k.H.n * FU * FD + 255) / 256), dim3(256), 0, stream, dcur0);
    chk(hipGetLastError(), "k_records");
  }
  void launch_order(const K& k, int32_t* order_idx, int32_t* rank) {
    const int nb = (k.H.n + 255) / 256;
    hipLaunchKernelGGL(k_order, dim3(nb, (k.H.n + ORDER_TILE - 1) / ORDER_TILE), dim3(256), 0, stream, dcur0, rank);
    hipLaunchKernelGGL(k_order_scatter, dim3(nb), dim3(256), 0, stream, dcur0, k.H.n, (const int32_t*)rank, order_idx, 1);
    stat_patched = true;
    chk(hipGetLastError(), "k_order");
  }
  // launch_records + launch_order with the two independent kernels in one launch
  static constexpr bool FUSE_RECORDS_ORDER = true;
  void launch_records_order(const K& k, int32_t* order_idx, int32_t* rank) {
    if (k.H.n == 0) return;
    const int nb = (k.H.n + 255) / 256, ny = (k.H.n + ORDER_TILE - 1) / ORDER_TILE, nrec = (k.H.n * FU * FD + 255) / 256;
    hipLaunchKernelGGL(k_order_records, dim3(nb * ny + nrec), dim3(256), 0, stream, dcur0, rank, nb, nb * ny);
    hipLaunchKernelGGL(k_order_scatter, dim3(nb), dim3(256), 0, stream, dcur0, k.H.n, (const int32_t*)rank, order_idx, 1);
    stat_patched = true;
    chk(hipGetLastError(), "k_order_records");
  }
  // dynamic LDS = [cohort rows (2 planes) of the largest tree, if they fit][CH prefetched entry records]
  void launch_process(const K& k, int n_tree, size_t cohort_rows_bytes) {
    // 160 KB per CU: the kernel's static LDS (Wave) + [cohort rows of both planes, if they fit] + two record buffers
    const size_t rec = sizeof(PRec) * CH * NBUF, budget = 160 * 1024 - sizeof(Wave) - 256;
    size_t lds = rec + (cohort_rows_bytes + rec <= budget ? cohort_rows_bytes : 0);
    if (lds > 48 * 1024 && lds != lds_attr) {
      chk(hipFuncSetAttribute((const void*)k_process, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
      lds_attr = lds;
    }
    // after launch_order the block of the nominate step is already what the process step needs (k_order_scatter moved its byte counter)
    const K* d = (stat_patched && !getenv("KQ_PROF_SKIP_NOMINATE")) ? dcur0 : put_k(k, 1);   // (timing builds: the nominate step's block counts into the sink)
    dproc = d; stat_patched = false;
    if (!spec_off && k.spec_kt) chk(launch_process_spec(d, n_tree, stream), "k_process_spec");
    hipLaunchKernelGGL(k_process, dim3(n_tree), dim3(PROCESS_THREADS), lds, stream, d, (unsigned)lds);
    chk(hipGetLastError(), "k_process");
  }
  // admitted-row structures on the device (kq_rows_kernel.hip)
  void launch_rows(const DRows& R, int op, int n) { chk(rows_launch(R, op, n, stream), "k_rows"); }
  void sort_pairs(uint64_t*& key, int32_t*& val, uint64_t*& key2, int32_t*& val2, int n, int bits) { chk(rows_sort_pairs(rows_tmp, key, val, key2, val2, n, bits, stream), "rows_sort_pairs"); }
  void scan_excl(const int32_t* in, int32_t* out, int n) { chk(rows_scan_excl(rows_tmp, in, out, n, stream), "rows_scan_excl"); }
  // kq_cycle_run_tas (kq_tas_cycle_kernel.hip)
  void launch_tas_base(const TCyc* c, int n) { chk(launch_tas_base_k(c, n, stream), "k_tas_base"); }
  void launch_tas_cycle_classes(const TCyc* c, int n) { chk(launch_tas_cycle_classes_k(c, n, stream), "k_tas_cycle_classes"); }
  void launch_nominate_tas(const K& k, int slots) {
    stat_patched = false;
    const K* d = put_k(k, 0);
    chk(tas_bal ? launch_nominate_tas_k_bal(d, slots, stream) : launch_nominate_tas_k(d, slots, stream), "k_nominate_tas");
  }
  bool tas_bal = false;   // this cycle's TAS flavors carry KQ_TAS_F_BALANCED_PLACEMENT: the _bal kernels (set by cycle_run_tas)
  void launch_process_tas(const K& k, size_t lds_want) {
    const K* d = stat_patched ? dcur0 : put_k(k, 1);
    dproc = d; stat_patched = false;
    chk(tas_bal ? launch_process_tas_k_bal(d, lds_want, &lds_attr_tas_bal, stream) : launch_process_tas_k(d, lds_want, &lds_attr_tas, stream), "k_process_tas");
  }
  size_t lds_attr_tas = 0, lds_attr_tas_bal = 0;
  size_t lds_attr = 0, lds_attr_fair = 0, lds_attr_sim = 0;
  const K* dproc = nullptr;    // argument block of the last process launch (kq_cycle_commit reads the cycle's outputs through it)
  bool stat_patched = false;
  bool spec_off = getenv("KQ_SPEC_OFF") != nullptr;  // KQ_SPEC_OFF: every tree goes to the serial kernel (A/B timing)
  // helper workgroups of k_process_fair (K::help): KQ_HELP_BLOCKS of them, none by default. A recomputation under its nomination
  // mapping only simulates the nominated flavor (2-3 searches per batch at cfg 4f): measured 14.8 s against 15.3 s per cycle with 16
  // helpers — kept as an experiment (GPU tests run it), not worth being on. They only exist while every tree's leader workgroup is
  // resident as well (free CUs), and they leave when the last leader is done.
  int help_blocks(int n_tree) {
    const char* e = getenv("KQ_HELP_BLOCKS");
    const int want = e ? atoi(e) : 0;
    const int free_cu = n_cu - n_tree;
    return want > 0 && free_cu >= 8 ? std::min(free_cu, want) : 0;
  }
  void launch_process_fair(const K& k, int n_tree, size_t cohort_rows_bytes, size_t search_bytes, int32_t* rank) {
    // [cohort rows of both planes, if they fit | the state of a recomputation's victim search (kq_fs.hpp), which borrows the region][one record]
    const size_t budget = 160 * 1024 - sizeof(Wave) - 256;
    size_t region = cohort_rows_bytes + sizeof(PRec) <= budget ? cohort_rows_bytes : 0;
    region = std::max(region, std::min(search_bytes, budget - sizeof(PRec)));
    size_t lds = sizeof(PRec) + region;
    lds = (lds + 15) & ~(size_t)15;
    // the iterator's per-tree state next to it (kq_device.hpp FIter): only if the whole thing still fits the CU
    size_t iter = (fiter_bytes(k.X.max_tree_nodes, k.X.max_tree_cqs) + 15) & ~(size_t)15;
    if (getenv("KQ_FS_ITER_LDS") && getenv("KQ_FS_ITER_LDS")[0] == '0') iter = 0;
    if (lds + iter > budget) iter = 0;
    lds += iter;
    if (lds > 48 * 1024 && lds != lds_attr_fair) {
      chk(hipFuncSetAttribute((const void*)k_process_fair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
      lds_attr_fair = lds;
    }
    const K* d = put_k(k, 1);
    dproc = d;
    const int nblk = n_tree + (k.help ? help_blocks(n_tree) : 0);
    hipLaunchKernelGGL(k_process_fair, dim3(nblk), dim3(FAIR_THREADS), lds, stream, d, (unsigned)lds, (unsigned)iter);
    const int nb = (k.H.n + 255) / 256;
    chk(hipMemsetAsync(rank, 0, (size_t)k.H.n * sizeof(int32_t), stream), "memset rank");
    hipLaunchKernelGGL(k_fair_rank, dim3(nb, nb), dim3(256), 0, stream, d, rank);
    hipLaunchKernelGGL(k_fair_rank_apply, dim3(nb), dim3(256), 0, stream, d, (const int32_t*)rank);
    chk(hipGetLastError(), "k_process_fair");
  }
};
}  // namespace kq

struct kq_engine {
  EngineT<HipBackend> e;
};
struct kq_tas { TasT<HipBackend> e; };

// No exception crosses the C ABI (include/kq_engine.h: "return 0 or a negative KQ_E*"): the host side is std::vector code, and a
// std::bad_alloc / std::length_error escaping an extern "C" function is std::terminate in the controller process. Every entry point
// that reaches engine code runs under KQ_TRY; the message is kept for kq_last_error / kq_tas_last_error.
namespace {
template <class T> int kq_fail(T* obj, int code, const char* what) noexcept {
  try { if (obj) obj->e.last_error = std::string("exception at the C ABI: ") + what; } catch (...) {}
  return code;
}
}  // namespace
#define KQ_TRY(obj, ...)                                                                                      \
  try { __VA_ARGS__; }                                                                                        \
  catch (const std::bad_alloc&) { return kq_fail(obj, KQ_ENOMEM, "out of host memory"); }                     \
  catch (const std::length_error& x) { return kq_fail(obj, KQ_ENOMEM, x.what()); }                            \
  catch (const std::exception& x) { return kq_fail(obj, KQ_EINVAL, x.what()); }                               \
  catch (...) { return kq_fail(obj, KQ_EINVAL, "unknown exception"); }

extern "C" {

int kq_abi_version(void) { return KQ_ABI_VERSION; }

const char* kq_strerror(int code) {
  switch (code) {
    case KQ_OK: return "ok";
    case KQ_EINVAL: return "invalid argument";
    case KQ_ENOMEM: return "out of memory";
    case KQ_EDEVICE: return "HIP error";
    case KQ_EUNSUPPORTED: return "input not supported by the device path";
    case KQ_ECAPACITY: return "output buffer too small";
    case KQ_ENODEVICE: return "no HIP device";
    default: return "unknown error";
  }
}

int kq_engine_create(const kq_config* cfg, kq_engine** out) {
  if (!cfg || !out) return KQ_EINVAL;
  if (cfg->abi_version != KQ_ABI_VERSION) return KQ_EINVAL;
  kq_engine* en = new (std::nothrow) kq_engine();
  if (!en) return KQ_ENOMEM;
  en->e.cfg = *cfg;
  int rc = KQ_EINVAL;
  try { rc = en->e.be.init(cfg->device); } catch (const std::bad_alloc&) { rc = KQ_ENOMEM; } catch (...) { rc = KQ_EINVAL; }
  if (rc != KQ_OK) { fprintf(stderr, "kq_engine_create: %s\n", en->e.be.msg.c_str()); delete en; return rc; }
  *out = en;
  return KQ_OK;
}

void kq_engine_destroy(kq_engine* en) {
  if (!en) return;
  try {
    (void)hipSetDevice(en->e.be.device);
    en->e.pending_free();
    en->e.free_snapshot();
    HipBackend be = en->e.be;
    delete en;
    be.destroy();
  } catch (...) {}
}

int kq_snapshot_put(kq_engine* en, const kq_snapshot* s) {
  if (!en || !s) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.snapshot_put(s));
}

int kq_snapshot_patch(kq_engine* en, const kq_snapshot* s, uint32_t what) {
  if (!en || !s) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.snapshot_patch(s, what));
}

int kq_cycle_run(kq_engine* en, const kq_heads* h, kq_decisions* out) {
  if (!en || !h || !out) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_run(h, out));
}

int kq_cycle_shard_words(kq_engine* en, const kq_heads* h, const kq_decisions* out, int32_t world, int64_t* words) {
  if (!en || !h || !out || !words) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_shard_words(h, out, world, words));
}
int kq_cycle_nominate_shard(kq_engine* en, const kq_heads* h, const uint8_t* mine, int32_t world, int32_t rank, void* xbuf_dev, kq_decisions* out) {
  if (!en || !h || !out) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_nominate_shard(h, mine, world, rank, xbuf_dev, out));
}
int kq_cycle_process_merged(kq_engine* en, int32_t world, int32_t rank, const void* xbuf_dev, kq_decisions* out) {
  if (!en || !out || !xbuf_dev) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_process_merged(world, rank, xbuf_dev, out));
}

int kq_heads_put(kq_engine* en, const kq_heads* h, int32_t batch) {
  if (!en || !h || batch < 0) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.heads_put(h, batch + 1));
}

int kq_cycle_run_resident(kq_engine* en, int32_t batch, kq_decisions* out) {
  if (!en || !out || batch < 0) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_exec(batch + 1, out));
}

int kq_nominate_run_resident(kq_engine* en, int32_t batch, kq_decisions* out) {
  if (!en || !out || batch < 0) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_exec(batch + 1, out, true));
}

int kq_pending_put(kq_engine* en, const kq_pending* p) {
  if (!en || !p) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_put(p));
}
int kq_pending_heads(kq_engine* en, int64_t cycle, const uint8_t* cq_active, int32_t* n_heads, int32_t* n_podsets, int32_t* head_wl) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_heads(cycle, cq_active, n_heads, n_podsets, head_wl));
}
int kq_cycle_run_pending(kq_engine* en, kq_decisions* out) {
  if (!en || !out) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_run_pending(out));
}
int kq_pending_apply(kq_engine* en) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_apply());
}
int kq_pending_bounds(kq_engine* en, int32_t* max_heads, int32_t* max_podsets) {
  if (!en) return KQ_EINVAL;
  KQ_TRY(en, return en->e.pending_bounds(max_heads, max_podsets));
}
int kq_pending_step(kq_engine* en, int64_t cycle, const uint8_t* cq_active, int32_t tgt_cap, int32_t release_age, int32_t want_head_wl) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_step(cycle, cq_active, tgt_cap, release_age, want_head_wl));
}
int kq_pending_step_wait(kq_engine* en, kq_decisions* out, int32_t* n_heads, int32_t* n_podsets, int32_t* head_wl) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_step_wait(out, n_heads, n_podsets, head_wl));
}
int kq_pending_step_reasons(kq_engine* en, int32_t rsn_cap) { if (!en) return KQ_EINVAL; KQ_TRY(en, return en->e.pending_step_reasons(rsn_cap)); }
int kq_pending_afs_put(kq_engine* en, const kq_afs_ledger* l) {
  if (!en || !l) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_afs_put(l));
}
int kq_pending_afs_wl_penalty(kq_engine* en, int32_t n, const int32_t* wl, const uint64_t* lo, const int64_t* hi, const uint64_t* mask) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_afs_wl_penalty(n, wl, lo, hi, mask));
}
int kq_pending_afs_sub_penalty(kq_engine* en, int32_t n, const int32_t* wl) {
  if (!en || (n > 0 && !wl)) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_afs_sub_penalty(n, wl));
}
int kq_pending_afs_set_consumed(kq_engine* en, int32_t n, const int32_t* lq, const uint64_t* lo, const int64_t* hi, const double* f64, const int32_t* settle_wl) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_afs_set_consumed(n, lq, lo, hi, f64, settle_wl));
}
int kq_pending_afs_read(kq_engine* en, double* usage, uint64_t* plo, int64_t* phi, uint8_t* ppres, uint64_t* clo, int64_t* chi, uint8_t* wrec) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_afs_read(usage, plo, phi, ppres, clo, chi, wrec));
}
int kq_pending_set_lq_usage(kq_engine* en, int32_t n_lq, const double* usage) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_set_lq_usage(n_lq, usage));
}
int kq_pending_add(kq_engine* en, const kq_pending* more, int32_t* first_index) {
  if (!en || !more) return KQ_EINVAL;
  KQ_TRY(en, return en->e.pending_add(more, first_index));
}
int kq_pending_update(kq_engine* en, int32_t n, const int32_t* wl, const kq_pending* more, int32_t* first_index) {
  if (!en || !more || (n > 0 && !wl)) return KQ_EINVAL;
  KQ_TRY(en, return en->e.pending_update(n, wl, more, first_index));
}
int kq_pending_set_clock(kq_engine* en, int64_t now_ns) { if (!en) return KQ_EINVAL; KQ_TRY(en, return en->e.pending_set_clock(now_ns)); }
int kq_pending_set_requeue_at(kq_engine* en, int32_t n, const int32_t* wl, const int64_t* at) {
  if (!en || (n > 0 && (!wl || !at))) return KQ_EINVAL;
  KQ_TRY(en, return en->e.pending_set_requeue_at(n, wl, at));
}
int kq_pending_delete(kq_engine* en, int32_t n, const int32_t* wl) {
  if (!en || (n > 0 && !wl)) return KQ_EINVAL;
  KQ_TRY(en, return en->e.pending_delete(n, wl));
}
int kq_pending_queue_inadmissible(kq_engine* en, int32_t n, const int32_t* cq) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_queue_inadmissible(n, cq));
}
int kq_pending_read_state(kq_engine* en, uint8_t* state, int32_t* counts) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.pending_read_state(state, counts));
}

int kq_cycle_certificate(kq_engine* en, int64_t* usage_delta_dev, int64_t* root_margin, int32_t* flags) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_certificate(usage_delta_dev, root_margin, flags));
}
int kq_snapshot_patch_rows(kq_engine* en, const kq_row_patch* p, int32_t* new_index) {
  if (!en || !p) return KQ_EINVAL;
  KQ_TRY(en, return en->e.snapshot_patch_rows(p, new_index));
}
int kq_debug_check_guards(kq_engine* en, int64_t* out3) {
  if (!en || !out3) return KQ_EINVAL;
  KQ_TRY(en, { std::string t; const int rc = en->e.be.check_guards(out3, &t); if (rc != KQ_OK || out3[1]) en->e.last_error = t; return rc; });
}
int kq_debug_rows_rebuild(kq_engine* en) { if (!en) return KQ_EINVAL; KQ_TRY(en, return en->e.debug_rows_rebuild()); }
int kq_debug_read_rows(kq_engine* en, int32_t which, void* out, int64_t* bytes) { if (!en || !bytes) return KQ_EINVAL; KQ_TRY(en, return en->e.read_rows(which, out, bytes)); }
int kq_cycle_run_tas(kq_engine* en, const kq_heads* h, const kq_cycle_tas* t, kq_decisions* out, kq_cycle_tas_out* tout, int64_t* stats) {
  if (!en || !h || !out) return KQ_EINVAL;
  KQ_TRY(en, return en->e.cycle_run_tas(h, t, out, tout, stats));
}
int kq_snapshot_usage_add(kq_engine* en, const int64_t* delta_dev, int32_t sign) {
  if (!en || !delta_dev) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.snapshot_usage_add(delta_dev, sign));
}

int kq_last_cycle_phases(kq_engine* en, double* phase_ms, int64_t* phase_bytes) {
  if (!en) return KQ_EINVAL;
  if (phase_ms) for (int p = 0; p < 3; p++) phase_ms[p] = en->e.last_phase_ms[p];
  if (phase_bytes) { phase_bytes[0] = en->e.last_phase_bytes[0]; phase_bytes[1] = en->e.last_phase_bytes[1]; }
  return KQ_OK;
}

int kq_last_cycle_stats(kq_engine* en, double* kernel_ms, int64_t* algorithmic_bytes) {
  if (!en) return KQ_EINVAL;
  if (kernel_ms) *kernel_ms = en->e.last_kernel_ms;
  if (algorithmic_bytes) *algorithmic_bytes = en->e.last_bytes;
  return KQ_OK;
}

int kq_cycle_commit(kq_engine* en, int32_t* n_admitted) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_commit(n_admitted));
}
int kq_cycle_release(kq_engine* en, int32_t age) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.cycle_release(age));
}
int kq_snapshot_derive(kq_engine* en) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.snapshot_derive());
}

int kq_snapshot_read_planes(kq_engine* en, int64_t* subtree_quota, int64_t* usage, uint8_t* quota_flags) {
  if (!en || !en->e.have_snapshot) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.read_planes(subtree_quota, usage, quota_flags));  // (re-derives the cohort levels first if commits are pending)
}

const char* kq_last_error(kq_engine* en) { return en ? en->e.last_error.c_str() : "null engine"; }

// ---- include/kq_tas.h ------------------------------------------------------------------------------
int kq_tas_create(int32_t device, kq_tas** out) {
  if (!out) return KQ_EINVAL;
  kq_tas* t = new (std::nothrow) kq_tas();
  if (!t) return KQ_ENOMEM;
  int rc = KQ_EINVAL;
  try { rc = t->e.be.init(device); } catch (const std::bad_alloc&) { rc = KQ_ENOMEM; } catch (...) { rc = KQ_EINVAL; }
  if (rc != KQ_OK) { fprintf(stderr, "kq_tas_create: %s\n", t->e.be.error()); delete t; return rc; }
  *out = t;
  return KQ_OK;
}
void kq_tas_destroy(kq_tas* t) {
  if (!t) return;
  try {
    (void)hipSetDevice(t->e.be.device);
    t->e.free_topo();
    HipBackend be = t->e.be;  // the backend handles outlive the engine object: its destructor still frees device buffers
    delete t;
    be.destroy();
  } catch (...) {}
}
int kq_tas_topology_put(kq_tas* t, const kq_tas_topology* tp) { if (!t || !tp) return KQ_EINVAL; (void)hipSetDevice(t->e.be.device); KQ_TRY(t, return t->e.topology_put(tp)); }
int kq_tas_find(kq_tas* t, const kq_tas_requests* r, kq_tas_result* out) { if (!t || !r || !out) return KQ_EINVAL; (void)hipSetDevice(t->e.be.device); KQ_TRY(t, return t->e.find(r, out)); }
int kq_tas_usage_apply(kq_tas* t, int32_t n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int32_t add) {
  if (!t) return KQ_EINVAL;
  (void)hipSetDevice(t->e.be.device);
  KQ_TRY(t, return t->e.usage_apply(n, leaf, count, spr, add));
}
int kq_tas_fits(kq_tas* t, int32_t n, const int32_t* leaf, const int32_t* count, const int64_t* spr, int32_t* fits) {
  if (!t || !fits) return KQ_EINVAL;
  (void)hipSetDevice(t->e.be.device);
  KQ_TRY(t, return t->e.fits(n, leaf, count, spr, fits));
}
int kq_tas_admit(kq_tas* t, const kq_tas_requests* r, const kq_tas_result* res, const int32_t* order, int32_t n_order, uint8_t* admitted, int32_t* n_admitted) {
  if (!t || !r || !res) return KQ_EINVAL;
  (void)hipSetDevice(t->e.be.device);
  KQ_TRY(t, return t->e.admit(r, res, order, n_order, admitted, n_admitted));
}
int kq_tas_usage_delta(kq_tas* t, const kq_tas_requests* r, const kq_tas_result* res, const uint8_t* wl_sel, int64_t* plane_dev) {
  if (!t || !r || !res) return KQ_EINVAL;
  (void)hipSetDevice(t->e.be.device);
  KQ_TRY(t, return t->e.usage_delta(r, res, wl_sel, plane_dev));
}
int kq_tas_usage_add(kq_tas* t, const int64_t* plane_dev, int32_t sign) { if (!t) return KQ_EINVAL; (void)hipSetDevice(t->e.be.device); KQ_TRY(t, return t->e.usage_add(plane_dev, sign)); }
int kq_tas_overflow(kq_tas* t, const int64_t* plane_dev, uint8_t* leaf_over, int32_t* n_over) {
  if (!t) return KQ_EINVAL;
  (void)hipSetDevice(t->e.be.device);
  KQ_TRY(t, return t->e.overflow(plane_dev, leaf_over, n_over));
}
int kq_tas_find_elastic(kq_tas* t, const kq_tas_requests* r, const kq_tas_replacement* prev, kq_tas_result* out) {
  if (!t || !r || !prev || !out) return KQ_EINVAL;
  (void)hipSetDevice(t->e.be.device);
  KQ_TRY(t, return t->e.find_elastic(r, prev, out));
}
int kq_tas_find_replacement(kq_tas* t, const kq_tas_requests* r, const kq_tas_replacement* x, kq_tas_result* out) {
  if (!t || !r || !x || !out) return KQ_EINVAL;
  (void)hipSetDevice(t->e.be.device);
  KQ_TRY(t, return t->e.find_replacement(r, x, out));
}
int kq_tas_exclusion_stats(kq_tas* t, const kq_tas_requests* r, const kq_tas_replacement* x, const kq_tas_result* res, int32_t n_sel, const int32_t* podsets,
                           const int32_t* resource_rank, int32_t* topology_domain, int32_t* resources) {
  if (!t || !r || !res) return KQ_EINVAL;
  (void)hipSetDevice(t->e.be.device);
  KQ_TRY(t, return t->e.exclusion_stats(r, x, res, n_sel, podsets, resource_rank, topology_domain, resources));
}
int kq_tas_read_usage(kq_tas* t, int64_t* u) { if (!t || !u) return KQ_EINVAL; (void)hipSetDevice(t->e.be.device); KQ_TRY(t, return t->e.read_usage(u)); }
int kq_tas_last_stats(kq_tas* t, double* ms, int64_t* bytes) { if (!t) return KQ_EINVAL; if (ms) *ms = t->e.last_ms; if (bytes) *bytes = t->e.last_bytes; return KQ_OK; }
const char* kq_tas_last_error(kq_tas* t) { return t ? t->e.last_error.c_str() : "null engine"; }

// profiling hook (KQ_PROF builds): 64 segment cycle counters accumulated since the last reset
// tests: take the saturation-safe DRS loops even when the incremental sums would be exact
int kq_debug_disable_scan_search(kq_engine* en, int on) { if (!en) return KQ_EINVAL; en->e.cs_disable = on != 0; en->e.fs_disable = on != 0; return KQ_OK; }
int kq_debug_force_exact_drs(kq_engine* en, int on) { if (!en) return KQ_EINVAL; en->e.force_exact_drs = on != 0; return KQ_OK; }
int kq_debug_spec_stats(kq_engine* en, int64_t* out8) { if (!en) return KQ_EINVAL; KQ_TRY(en, return en->e.spec_stats(out8)); }
int kq_debug_prof(kq_engine* en, int64_t* out, int reset) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.prof_read(out, reset != 0));
}

// test hook (not part of the drop-in boundary): snapshot usage as left by the last cycle
int kq_debug_read_usage_work(kq_engine* en, int64_t* out) {
  if (!en) return KQ_EINVAL;
  (void)hipSetDevice(en->e.be.device);
  KQ_TRY(en, return en->e.read_usage_work(out));
}

}  // extern "C"

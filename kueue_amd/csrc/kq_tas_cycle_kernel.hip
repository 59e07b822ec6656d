// kq_tas_cycle_kernel.hip — the kernels of kq_cycle_run_tas (include/kq_cycle_tas.h): Topology-Aware Scheduling INSIDE the scheduling
// cycle. The cycle's device code (kq_device.hpp) is compiled here a second time with the TAS hooks switched on (KQ_TAS_CYCLE,
// kq_tas_cycle.hpp), so that the kernels of the ordinary cycle (kq_engine.hip) carry none of it.
//   k_tas_base      1 thread per TopologyDomainRequests entry of the admitted rows: base leaf usage = tas_usage + admitted usage
//   k_nominate_tas  1 wave per head, grid-stride (as k_nominate): flavor assignment, the placement (FindTopologyAssignmentsForFlavor,
//                   t_workload in-wave: lanes = leaves / domains), GetTargets with the TAS-aware workloadFits, partial admission
//   k_process_tas   1 wave for the whole cycle: entries of every root cohort in iterator order (a TAS flavor's leaves are shared by
//                   ClusterQueues of different root cohorts, snapshot.go:260), recomputation on overlap / on lost TAS capacity
#define KQ_TAS_CYCLE 1
// kq_tas_cycle_kernel_bal.hip compiles this file a second time with KQ_TAS_BAL (tas_balanced_placement.go inside the placement): the two
// kernels that carry the placement get a second name, the others exist once
#ifdef KQ_TAS_BAL
#define KQ_TC_NAME(x) x##_bal
#else
#define KQ_TC_NAME(x) x
#endif
#define KQ_FAIR_WALK_ONLY 1   // a fair-sharing TAS cycle's victim searches carry leaf usage: the candidate-by-candidate walk (kq_fs.hpp's LDS formulation stays out of these kernels)
#include <hip/hip_runtime.h>

#include "kq_device.hpp"
#include "kq_tas_cycle.hpp"

using namespace kq;

#ifndef KQ_TAS_BAL
__global__ __launch_bounds__(256) void k_tas_base(const TCyc* __restrict__ c, int n) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n) tc_base_cell(*c, e);
}
// phase 1 of every (TAS flavor, request class) over the work plane, one wave each, right before k_process_tas starts its walk
__global__ __launch_bounds__(64) void k_tas_cycle_classes(const TCyc* __restrict__ c) {
  tc_class_init(*c, (int)blockIdx.x / c->ncls, (int)blockIdx.x % c->ncls);
}
#endif
__global__ __launch_bounds__(64) void KQ_TC_NAME(k_nominate_tas)(const K* __restrict__ kp, int slots) {
  const K& k = *kp;
  __shared__ Wave w;
  if (threadIdx.x == 0) { w.cs_lds = nullptr; w.cs_lds_bytes = 0; w.help_on = 0; w.ta.plane = 0; w.ta.srch = 0; w.ta.mail = nullptr; w.ta.lds = nullptr; w.ta.lds_bytes = 0; w.ta.pf_pos = -1; w.ta.q_lds = 0; w.ta.d_lds = 0; w.ta.pub_lds = 0; w.ta.pool_own = 0; w.ta.pool_next = 0; w.ta.cur_pre = nullptr; w.ta.req_valid = 0; w.ta.em_ps = -1; }
  __syncthreads();
  const int slot = blockIdx.x;
  for (int h = slot, n = hn(k.H); h < n; h += slots) nominate_head(k, w, h, slot);
}
// wave 0 walks the entries; waves 1..3 wait for the placements' phase-1 jobs (TLeafJob in LDS) and fill their stripes of the leaves.
// 256 threads = one wave per SIMD: the leader keeps its 512 registers.
constexpr int PROCESS_TAS_THREADS = 256;
// lds_bytes of dynamic LDS hold the working state of a class-path placement (kq_tas_device.hpp TLds); 0 = the slot's global rows
__global__ __launch_bounds__(PROCESS_TAS_THREADS) void KQ_TC_NAME(k_process_tas)(const K* __restrict__ kp, unsigned lds_bytes, int coop_min) {
  __shared__ Wave w;
  __shared__ TLeafJob job;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  if (threadIdx.x == 0) { job.cmd = 0; job.nw = PROCESS_TAS_THREADS / 64; job.bytes = 0; job.coop_min = coop_min; job.early_pending = 0; job.early_cls = -1; }
  __syncthreads();
  if (threadIdx.x < 64) {
    process_all_tas(*kp, w, 0, &job, dyn_lds, (int)lds_bytes);
    t_post_begin(job);   // (a split-phase copy nobody asked for any more)
    if (threadIdx.x == 0) job.cmd = 2;
    __syncthreads();
  } else {
    t_leaf_helper(job, (int)(threadIdx.x >> 6), PROCESS_TAS_THREADS / 64);
  }
}

namespace kq {
#ifndef KQ_TAS_BAL
hipError_t launch_tas_base_k(const TCyc* c, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_tas_base, dim3((n + 255) / 256), dim3(256), 0, stream, c, n);
  return hipGetLastError();
}
hipError_t launch_tas_cycle_classes_k(const TCyc* c, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_tas_cycle_classes, dim3(n), dim3(64), 0, stream, c);
  return hipGetLastError();
}
#endif
hipError_t KQ_TC_NAME(launch_nominate_tas_k)(const K* d, int slots, hipStream_t stream) {
  if (slots <= 0) return hipSuccess;
  hipLaunchKernelGGL(KQ_TC_NAME(k_nominate_tas), dim3(slots), dim3(64), 0, stream, d, slots);
  return hipGetLastError();
}
// want = bytes of LDS a placement's working state needs (0: none); granted when it fits next to the kernel's static LDS
hipError_t KQ_TC_NAME(launch_process_tas_k)(const K* d, size_t want, size_t* attr_p, hipStream_t stream) {
  size_t& attr = *attr_p;   // (per engine: the attribute is per device)
  hipFuncAttributes fa{};
  hipError_t e = hipFuncGetAttributes(&fa, (const void*)KQ_TC_NAME(k_process_tas));
  if (e != hipSuccess) return e;
  // what the device grants a workgroup (160 KB on gfx950): asked once, not assumed — on a part or a driver that grants less, or when the
  // attribute is refused, the placement's state stays in global memory (lds = 0, the KQ_TAS_LDS_OFF path) instead of failing the cycle
  static const size_t dev_lds = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) return (size_t)64 * 1024;
    return (size_t)v;
  }();
  const size_t room = dev_lds > fa.sharedSizeBytes + 256 ? dev_lds - fa.sharedSizeBytes - 256 : 0;
  size_t lds = (want > 0 && want <= room && !getenv("KQ_TAS_LDS_OFF")) ? want : 0;
  if (lds > attr) {
    e = hipFuncSetAttribute((const void*)KQ_TC_NAME(k_process_tas), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); lds = 0; }
    else attr = lds;
  }
  const char* cm = getenv("KQ_TAS_COOP_MIN");   // (tests: short slices shared as well)
  hipLaunchKernelGGL(KQ_TC_NAME(k_process_tas), dim3(1), dim3(PROCESS_TAS_THREADS), lds, stream, d, (unsigned)lds, cm ? atoi(cm) : 1024);
  return hipGetLastError();
}
}  // namespace kq

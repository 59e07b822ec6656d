// kq_tas_bal_kernel.hip — k_tas_find_bal: kq_tas_find's kernel with tas_balanced_placement.go inside the placement (kq_tas_device.hpp
// t_balanced_lane0), launched instead of k_tas_find when the topology carries KQ_TAS_F_BALANCED_PLACEMENT (features.TASBalancedPlacement,
// default off): the kernel everybody runs keeps its registers and stays without scratch.
#define KQ_TAS_BAL 1
#include <hip/hip_runtime.h>

#include "kq_tas_device.hpp"
#include "kq_tas_cycle.hpp"   // (as kq_engine.hip sees the placement: the definitions behind its forward declarations)

using namespace kq;

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_tas_find_bal(const TK* __restrict__ kp, int slots) {
  const TK& k = *kp;
  const int per = (k.Q.n_wl + slots - 1) / slots;
  const int lo = blockIdx.x * per, hi = (lo + per) < k.Q.n_wl ? (lo + per) : k.Q.n_wl;
  for (int i = lo; i < hi; i++) t_workload_t<false>(k, blockIdx.x, k.C.order[i]);
}
namespace kq {
hipError_t launch_tas_find_bal_k(const TK* d, int slots, hipStream_t stream) {
  hipLaunchKernelGGL(k_tas_find_bal, dim3(slots), dim3(64), 0, stream, d, slots);
  return hipGetLastError();
}
}  // namespace kq

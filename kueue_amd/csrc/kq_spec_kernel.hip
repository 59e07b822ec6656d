// kq_spec_kernel.hip — k_process_spec: processEntry for the plain entries of every root-cohort tree as speculative parallel rounds
// (kq_spec.hpp). Its own translation unit: the kernel is register- and LDS-heavy and changes often; the rest of the engine
// (kq_engine.hip) takes minutes to compile.
#include <hip/hip_runtime.h>

#include "kq_device.hpp"

using namespace kq;

// One 512-thread workgroup per root-cohort tree (at most SP_SLOTS workgroups, each looping over trees). LDS = SpecLds (dynamic).
__global__ __launch_bounds__(SP_NT) void k_process_spec(const K* __restrict__ kp, int slots) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  const K& k = *kp;
  for (int tree = blockIdx.x; tree < k.S.n_tree; tree += slots)
    spec_tree(k, tree, *(SpecLds*)dyn_lds, k.spec_kt + (size_t)blockIdx.x * SP_KT_WORDS, (int)threadIdx.x);
}

namespace kq {
hipError_t launch_process_spec(const K* d, int n_tree, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)k_process_spec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SpecLds));
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int slots = n_tree < SP_SLOTS ? n_tree : SP_SLOTS;
  if (slots <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_process_spec, dim3(slots), dim3(SP_NT), sizeof(SpecLds), stream, d, slots);
  return hipGetLastError();
}
}  // namespace kq

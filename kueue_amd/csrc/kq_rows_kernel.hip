// kq_rows_kernel.hip — the device-side rebuild of the admitted-row candidate structures (kq_rows.hpp): the cell kernel, and the two
// library primitives it needs — a stable radix sort of (64-bit key, int32 value) pairs and an exclusive prefix sum — from rocPRIM.
// Its own translation unit: the rocPRIM headers stay out of the engine's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kq_rows.hpp"

using namespace kq;

__global__ __launch_bounds__(256) void k_rows(DRows R, int op, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  rows_cell(R, op, i, i < n);
}

namespace kq {
// rocPRIM's temporary storage belongs to the engine that sorts (RowsScratch in its HipBackend): its device, its stream. (Round 3 kept
// one process-global buffer: two engines on different streams raced on it, an engine on device 1 used device 0's allocation.)
static hipError_t need(RowsScratch& s, size_t bytes, hipStream_t stream) {
  if (bytes <= s.cap) return hipSuccess;
  if (s.p) {
    // the buffer may still be read by a sort enqueued earlier on this stream
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    (void)hipFree(s.p);
    s.p = nullptr; s.cap = 0;
  }
  const size_t cap = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(&s.p, cap);
  if (e == hipSuccess) s.cap = cap;
  return e;
}

// KQ_ROWS_TRACE=1: every step of the rebuild is announced on stderr and waited for (a faulting step is then the last one named)
static bool rows_trace() { static const bool on = getenv("KQ_ROWS_TRACE") != nullptr; return on; }
static hipError_t rows_traced(const char* what, int a, int n, hipStream_t stream, hipError_t e) {
  if (!rows_trace()) return e;
  fprintf(stderr, "[kq_rows] %s %d n=%d -> %s", what, a, n, hipGetErrorString(e)); fflush(stderr);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  fprintf(stderr, " / %s\n", hipGetErrorString(e)); fflush(stderr);
  return e;
}
hipError_t rows_launch(const DRows& R, int op, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_rows, dim3((n + 255) / 256), dim3(256), 0, stream, R, op, n);
  return rows_traced("op", op, n, stream, hipGetLastError());
}
// stable sort of (key, val) by the low `bits` bits of the key between two pairs of buffers (rocPRIM's double-buffer form: no copy
// back); on return key / val name the pair that holds the result, key2 / val2 the other one
hipError_t rows_sort_pairs(RowsScratch& tmp, uint64_t*& key, int32_t*& val, uint64_t*& key2, int32_t*& val2, int n, int bits, hipStream_t stream) {
  if (n <= 1 || bits <= 0) return hipSuccess;
  rocprim::double_buffer<uint64_t> dk(key, key2);
  rocprim::double_buffer<int32_t> dv(val, val2);
  size_t bytes = 0;
  hipError_t e;
  if ((e = rocprim::radix_sort_pairs(nullptr, bytes, dk, dv, (unsigned)n, 0u, (unsigned)bits, stream)) != hipSuccess) return e;
  if ((e = need(tmp, bytes, stream)) != hipSuccess) return e;
  if ((e = rocprim::radix_sort_pairs(tmp.p, bytes, dk, dv, (unsigned)n, 0u, (unsigned)bits, stream)) != hipSuccess) return e;
  if (dk.current() != key) { uint64_t* t = key; key = key2; key2 = t; }
  if (dv.current() != val) { int32_t* t = val; val = val2; val2 = t; }
  return rows_traced("sort bits", bits, n, stream, hipSuccess);
}
hipError_t rows_scan_excl(RowsScratch& tmp, const int32_t* in, int32_t* out, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  size_t bytes = 0;
  hipError_t e;
  if ((e = rocprim::exclusive_scan(nullptr, bytes, in, out, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), stream)) != hipSuccess) return e;
  if ((e = need(tmp, bytes, stream)) != hipSuccess) return e;
  return rows_traced("scan", 0, n, stream, rocprim::exclusive_scan(tmp.p, bytes, in, out, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), stream));
}
}  // namespace kq

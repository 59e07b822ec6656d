// Shader clock seen by a short latency-bound kernel: clock64() (s_memtime, shader cycles) against wall_clock64() (100 MHz) —
// does the engine clock stay low when the GPU runs one small kernel after the other?   hipcc --offload-arch=gfx950 -O2 -o clk_probe clk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void k_spin(long long cycles, long long* out) {
  const long long c0 = clock64(), w0 = wall_clock64();
  while (clock64() - c0 < cycles) {}
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
}
__global__ void k_burn(float* x, int n) {   // keeps every CU busy for a while
  float a = x[threadIdx.x];
  for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f;
  x[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main(int argc, char** argv) {
  long long* d; hipMalloc(&d, 4096 * 16);
  float* x; hipMalloc(&x, 4096 * 256 * 4);
  long long h[2];
  auto probe = [&](const char* what, int blocks, long long cyc) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, 0, cyc, d); hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s blocks %5d: %lld clock64 ticks in %lld wall ticks (100 MHz) -> %.0f MHz ; event %.1f us\n", what, blocks, h[0], h[1], (double)h[0] / h[1] * 100.0, ms * 1000);
  };
  probe("cold, one short kernel", 1000, 100000);
  for (int i = 0; i < 5; i++) probe("short kernels back to back", 1000, 100000);
  usleep(200000);
  probe("after 200 ms idle", 1000, 100000);
  for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_burn, dim3(4096), dim3(256), 0, 0, x, 200000);
  hipDeviceSynchronize();
  probe("right after a heavy burn", 1000, 100000);
  for (int i = 0; i < 3; i++) probe("short kernels after the burn", 1000, 100000);
  probe("one long spin (1e7 ticks)", 1000, 10000000);
  return 0;
}

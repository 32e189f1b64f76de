// Throughput of fp32 global atomics by memory scope on MI355X (8 XCDs, private L2s).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int SCOPE>
__global__ void k(float* t, unsigned mask, int per, unsigned stride) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  unsigned xcd = 0;
  if (stride) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd)); xcd &= 7; }
  float* base = t + (size_t)xcd * stride;
  for (int i = 0; i < per; ++i) {
    s = s * 1664525u + 1013904223u;
    __hip_atomic_fetch_add(base + ((s >> 8) & mask), 1.0f, __ATOMIC_RELAXED, SCOPE);
  }
}
int main() {
  const unsigned entries = 1u << 20;   // 4 MB table (per copy)
  float* t; hipMalloc(&t, (size_t)entries * 8 * 4); hipMemset(t, 0, (size_t)entries * 8 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 4096, per = 256;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k<__HIP_MEMORY_SCOPE_AGENT><<<blocks, 256>>>(t, entries - 1, per, 0);
      if (mode == 1) k<__HIP_MEMORY_SCOPE_WORKGROUP><<<blocks, 256>>>(t, entries - 1, per, 0);
      if (mode == 2) k<__HIP_MEMORY_SCOPE_WORKGROUP><<<blocks, 256>>>(t, entries - 1, per, entries);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("mode %d (%s): %.3f ms, %.1f G atomics/s\n", mode,
                           mode == 0 ? "agent scope, shared table" : mode == 1 ? "workgroup scope, shared table (NOT coherent across XCDs)" : "workgroup scope, per-XCD table",
                           ms, (double)blocks * 256 * per / ms / 1e6);
    }
  }
  // correctness of mode 2: the 8 copies together must hold exactly blocks*256*per*3 increments
  float* h = (float*)malloc((size_t)entries * 8 * 4); hipMemset(t, 0, (size_t)entries * 8 * 4);
  k<__HIP_MEMORY_SCOPE_WORKGROUP><<<blocks, 256>>>(t, entries - 1, per, entries); hipDeviceSynchronize();
  hipMemcpy(h, t, (size_t)entries * 8 * 4, hipMemcpyDeviceToHost);
  double sum = 0; for (size_t i = 0; i < (size_t)entries * 8; ++i) sum += h[i];
  printf("per-XCD tables: total %.0f, expected %.0f\n", sum, (double)blocks * 256 * per);
  hipMemset(t, 0, (size_t)entries * 8 * 4);
  k<__HIP_MEMORY_SCOPE_WORKGROUP><<<blocks, 256>>>(t, entries - 1, per, 0); hipDeviceSynchronize();
  hipMemcpy(h, t, (size_t)entries * 4, hipMemcpyDeviceToHost);
  sum = 0; for (size_t i = 0; i < entries; ++i) sum += h[i];
  printf("workgroup scope on ONE shared table: total %.0f, expected %.0f (lost updates across XCDs)\n", sum, (double)blocks * 256 * per);
  return 0;
}

// scratch: do fp32 atomics of neighbouring lanes to the SAME 8-byte pair / same 64-byte sector cost one L2 transaction?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int GROUP>   // GROUP consecutive lanes hit GROUP consecutive floats (one random base per group)
__global__ void k(float* t, unsigned mask, int per) {
  const unsigned lane = threadIdx.x & 63, g = lane / GROUP, o = lane % GROUP;
  unsigned s = (blockIdx.x * 256 + (threadIdx.x - o)) * 2654435761u;      // same seed for the lanes of a group
  for (int i = 0; i < per; ++i) {
    s = s * 1664525u + 1013904223u;
    const unsigned base = ((s >> 8) & mask) & ~(unsigned)(GROUP - 1);
    __hip_atomic_fetch_add(t + base + o, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void kd(double* t, unsigned mask, int per) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  for (int i = 0; i < per; ++i) { s = s * 1664525u + 1013904223u; __hip_atomic_fetch_add(t + ((s >> 8) & mask), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
int main() {
  const unsigned entries = 1u << 20;
  float* t; hipMalloc(&t, (size_t)entries * 8); hipMemset(t, 0, (size_t)entries * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 4096, per = 256;
  for (int mode = 0; mode < 6; ++mode) for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (mode == 0) k<1><<<blocks, 256>>>(t, entries - 1, per);
    if (mode == 1) k<2><<<blocks, 256>>>(t, entries - 1, per);
    if (mode == 2) k<4><<<blocks, 256>>>(t, entries - 1, per);
    if (mode == 3) k<16><<<blocks, 256>>>(t, entries - 1, per);
    if (mode == 4) k<32><<<blocks, 256>>>(t, entries - 1, per);
    if (mode == 5) kd<<<blocks, 256>>>((double*)t, entries - 1, per);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("mode %d (%s): %.3f ms, %.1f G atomics/s\n", mode, mode == 5 ? "f64 random" : mode == 0 ? "f32 random" : "f32 groups of 2/4/16/32 adjacent", ms, (double)blocks * 256 * per / ms / 1e6);
  }
  return 0;
}

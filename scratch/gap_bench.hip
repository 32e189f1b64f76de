// scratch: idle time between back-to-back launches on one stream: kernels that dirty a lot of memory (plain / nontemporal /
// write-through stores) vs kernels that only read
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>   // 0 plain stores, 1 nontemporal, 2 sc0 sc1 write-through, 3 read only
__global__ __launch_bounds__(512) void k(uint4* buf, size_t per_wg, float* out) {
  uint4* p = buf + (size_t)blockIdx.x * per_wg;
  uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i < per_wg; i += 512) {
    if (MODE == 0) p[i] = v;
    if (MODE == 1) { typedef unsigned __attribute__((ext_vector_type(4))) u4; const u4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (u4*)(p + i)); }
    if (MODE == 2) { typedef unsigned __attribute__((ext_vector_type(4))) u4; const u4 w = {v.x, v.y, v.z, v.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p + i), "v"(w) : "memory"); }
    if (MODE == 3) { uint4 r = p[i]; acc += r.x + r.w; }
  }
  if (MODE == 3 && acc == 12345) out[blockIdx.x] = acc;
}
int main() {
  const size_t per_wg = 65536;   // 1 MiB per workgroup, 256 MiB per launch
  uint4* buf; hipMalloc(&buf, 256 * per_wg * 16); hipMemset(buf, 0, 256 * per_wg * 16);
  float* out; hipMalloc(&out, 4096);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 6; ++i) k<0><<<256, 512>>>(buf, per_wg, out);
    for (int i = 0; i < 6; ++i) k<1><<<256, 512>>>(buf, per_wg, out);
    for (int i = 0; i < 6; ++i) k<2><<<256, 512>>>(buf, per_wg, out);
    for (int i = 0; i < 6; ++i) k<3><<<256, 512>>>(buf, per_wg, out);
  }
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}

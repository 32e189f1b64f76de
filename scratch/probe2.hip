#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned a = 1000 + l, b = 2000 + l;
  u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l * 2] = r[0]; out[l * 2 + 1] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 512);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 4) printf("L%02d: v=%u s=%u | L%02d: v=%u s=%u\n", l, h[l*2], h[l*2+1], l+1, h[l*2+2], h[l*2+3]);
  return 0;
}

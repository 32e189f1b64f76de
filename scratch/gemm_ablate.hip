// Ablation of k_gemm_nt_bf16 (128x128x64, 4 waves): where does the time go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define TB (128 * 64 * 2)
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) { const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3; return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k; }
__device__ __forceinline__ uint16_t f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <int MODE>  // 0 full, 1 no store, 2 no glds in loop, 3 no mfma
__global__ __launch_bounds__(256, 2) void k(int M, int N, int K, const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bt, const float* __restrict__ bias, uint16_t* __restrict__ C) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / 128, ntm = M / 128;
  const int t = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (t / ntn) * 128, n0 = (t % ntn) * 128;
  const int wm = wv >> 1, wn = wv & 1, nk = K / 64;
  auto stage = [&](int kt, int buf) {
    unsigned char* la = lds + buf * 2 * TB; unsigned char* lb = la + TB;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int p = it * 256 + tid, row = p >> 3, pos = p & 7, c = pos ^ (row & 7);
      glds16(A + (size_t)(m0 + row) * K + kt * 64 + c * 8, la + (it * 256 + wv * 64) * 16);
      glds16(Bt + (size_t)(n0 + row) * K + kt * 64 + c * 8, lb + (it * 256 + wv * 64) * 16);
    }
  };
  f32x4_t acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
  stage(0, 0);
  const int r16 = lane & 15, kb = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    __syncthreads();
    if (MODE != 2 && kt + 1 < nk) stage(kt + 1, buf ^ 1);
    const unsigned char* la = lds + buf * 2 * TB; const unsigned char* lb = la + TB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t xa[4], wb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int row = wm * 64 + i * 16 + r16; xa[i] = *(const bf16x8_t*)(la + row * 128 + (((kk * 4 + kb) ^ (row & 7)) << 4)); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int row = wn * 64 + j * 16 + r16; wb[j] = *(const bf16x8_t*)(lb + row * 128 + (((kk * 4 + kb) ^ (row & 7)) << 4)); }
      if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(xa[i])); asm volatile("" ::"v"(wb[i])); }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + i * 16 + r16, n = n0 + wn * 64 + j * 16 + kb * 4;
      const float4 b = *(const float4*)(bias + n);
      float x0 = fmaxf(acc[i][j][0] + b.x, 0.f), x1 = fmaxf(acc[i][j][1] + b.y, 0.f), x2 = fmaxf(acc[i][j][2] + b.z, 0.f), x3 = fmaxf(acc[i][j][3] + b.w, 0.f);
      uint2 pk; pk.x = f2bf(x0) | ((uint32_t)f2bf(x1) << 16); pk.y = f2bf(x2) | ((uint32_t)f2bf(x3) << 16);
      if (MODE == 1) { asm volatile("" ::"v"(pk.x), "v"(pk.y)); } else *(uint2*)(C + (size_t)m * N + n) = pk;
    }
}
template <int MODE> float run(int M, int N, int K, uint16_t* A, uint16_t* B, float* bias, uint16_t* C) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k<MODE><<<(M / 128) * (N / 128), 256>>>(M, N, K, A, B, bias, C);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) k<MODE><<<(M / 128) * (N / 128), 256>>>(M, N, K, A, B, bias, C);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main() {
  const int M = 131072, N = 1024, K = 1024;
  std::vector<uint16_t> h((size_t)M * K);
  for (auto& x : h) x = (uint16_t)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15));  // random bf16 in +-[0.0078,0.0156)... sign random
  uint16_t *A, *B, *C; float* bias;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
  hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, N * 4);
  const double fl = 2.0 * M * N * K;
  float t;
  t = run<0>(M, N, K, A, B, bias, C); printf("full      %.3f ms %.0f TF\n", t, fl / t / 1e9);
  t = run<1>(M, N, K, A, B, bias, C); printf("no-store  %.3f ms %.0f TF\n", t, fl / t / 1e9);
  t = run<2>(M, N, K, A, B, bias, C); printf("no-glds   %.3f ms %.0f TF\n", t, fl / t / 1e9);
  t = run<3>(M, N, K, A, B, bias, C); printf("no-mfma   %.3f ms %.0f TF-equiv\n", t, fl / t / 1e9);
  return 0;
}

// Sustained MFMA rate of the chip: every SIMD issues independent v_mfma_f32_16x16x32_bf16 back to back from registers
// (no memory traffic at all), for a few kernel lengths; the shader clock is read back as s_memtime / s_memrealtime (100 MHz).
// hipcc --offload-arch=gfx950 -O3 scratch/mfma_peak.hip -o scratch/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// variants: MODE 0 = 16x16x32 bf16, 1 = 32x32x16 bf16 (NACC/4 accumulators of 16 floats)
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma32(int iters, float seed, float* out, unsigned long long* clk) {
  f32x16_t acc[NACC / 4];
#pragma unroll
  for (int i = 0; i < NACC / 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (float)((threadIdx.x * 7 + i * 13) % 31 - 15)); b[i] = (__bf16)(seed * (float)((threadIdx.x * 3 + i * 5) % 29 - 14)); }
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC / 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC / 4; ++i) s += acc[i][0] + acc[i][5];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(int iters, float seed, float* out, unsigned long long* clk) {
  extern __shared__ char lds[];
  f32x4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (float)((threadIdx.x * 7 + i * 13) % 31 - 15)); b[i] = (__bf16)(seed * (float)((threadIdx.x * 3 + i * 5) % 29 - 14)); }
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  if (iters < 0) lds[threadIdx.x] = 0;
}

template <int NACC>
void run(const char* name, int wg_per_cu, int iters, size_t lds, float seed = 1.f, int threads = 256, int mode = 0, int ncu = 256) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = ncu * wg_per_cu;
  hipFuncSetAttribute((const void*)k_mfma<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_mfma32<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  float best = 1e30f; unsigned long long h[2] = {0, 1};
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (mode == 0) k_mfma<NACC><<<grid, threads, lds>>>(iters, seed, out, clk);
    else k_mfma32<NACC><<<grid, threads, lds>>>(iters, seed, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
  }
  const double nmfma = (double)iters * (mode ? NACC / 4 : NACC), waves = (double)grid * threads / 64;
  const double flop = waves * nmfma * 16.0 * 16 * 32 * 2 * (mode ? 2 : 1);
  printf("%-46s %8.3f ms  %7.1f TFLOP/s  clock %4.0f MHz  clk per MFMA per wave %6.2f\n", name, best, flop / best / 1e9,
         (double)h[0] / (double)h[1] * 100.0, (double)h[0] / nmfma);
}


// random operands: 4 A and 4 B register sets of hashed bf16 bit patterns (sign, 3 exponent bits, full mantissa vary: ~N(0,1)-like
// magnitudes in [0.25, 4)), successive MFMAs alternate between them -> the multiplier inputs toggle on every issue.
__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int NACC, int NSET>
__global__ __launch_bounds__(256) void k_mfma_rand(int iters, int zero_frac_256, int zero_frac_b, float* out, unsigned long long* clk) {
  f32x4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  u32x4_t a[NSET], b[NSET];
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned ha = hash32(threadIdx.x * 131u + s * 17u + i), hb = hash32(threadIdx.x * 977u + s * 29u + i + 1000u);
      // two bf16 per word: sign | exponent 125..128 | 7 mantissa bits
      auto mk = [&](unsigned h, int zero_frac_256) { unsigned lo = (h & 0x807fu) | ((125u + ((h >> 8) & 3u)) << 7); unsigned hi = ((h >> 16) & 0x807fu) | ((125u + ((h >> 24) & 3u)) << 7);
                                  if ((int)(h >> 20 & 255u) < zero_frac_256) lo = 0; if ((int)(h >> 4 & 255u) < zero_frac_256) hi = 0; return lo | (hi << 16); };
      a[s][i] = mk(ha, zero_frac_256); b[s][i] = mk(hb, zero_frac_b);
    }
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i % NSET]), "v"(b[(i / NSET) % NSET]));
    if ((it & 63) == 63) {   // keep the accumulators finite
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] *= 1e-3f;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (sum == 123.456f) out[0] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

void run_rand(const char* name, int iters, int zero_frac_256, int zero_frac_b = -1) {
  if (zero_frac_b < 0) zero_frac_b = zero_frac_256;
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f; unsigned long long h[2] = {0, 1};
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    k_mfma_rand<32, 4><<<256, 256, 100 * 1024>>>(iters, zero_frac_256, zero_frac_b, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
  }
  const double nmfma = (double)iters * 32, flop = 256.0 * 4 * nmfma * 16384.0;
  printf("%-46s %8.3f ms  %7.1f TFLOP/s  clock %4.0f MHz  clk per MFMA per wave %6.2f\n", name, best, flop / best / 1e9,
         (double)h[0] / (double)h[1] * 100.0, (double)h[0] / nmfma);
}

int main() {
  hipFuncSetAttribute((const void*)k_mfma_rand<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int it : {20000, 200000}) {
    run_rand("16x16x32 bf16, RANDOM operands, 4x4 register sets", it, 0);
    run_rand("  same, half of the operand values zero", it, 128);
    run_rand("  same, all operand values zero", it, 256);
    run_rand("  A operand (srcA) half zeros, B dense", it, 128, 0);
    run_rand("  B operand (srcB) half zeros, A dense", it, 0, 128);
    run_rand("  A operand all zeros, B dense", it, 256, 0);
    run_rand("  B operand all zeros, A dense", it, 0, 256);
  }
  const int it = 20000;
  run<32>("16x16x32 bf16, 1 wave/SIMD, data", 1, it, 100 * 1024);
  run<32>("16x16x32 bf16, 2 waves/SIMD, data", 2, it, 60 * 1024);
  run<32>("16x16x32 bf16, 2 waves/SIMD, ZERO operands", 2, it, 60 * 1024, 0.f);
  run<32>("16x16x32 bf16, 1 wave/SIMD, ZERO operands", 1, it, 100 * 1024, 0.f);
  run<32>("32x32x16 bf16, 1 wave/SIMD, data", 1, it, 100 * 1024, 1.f, 256, 1);
  run<32>("32x32x16 bf16, 2 waves/SIMD, data", 2, it, 60 * 1024, 1.f, 256, 1);
  run<32>("32x32x16 bf16, 2 waves/SIMD, ZERO operands", 2, it, 60 * 1024, 0.f, 256, 1);
  run<32>("16x16x32 bf16, ONE wave per CU (1 SIMD of 4)", 1, it, 100 * 1024, 1.f, 64);
  run<32>("16x16x32 bf16, 2 waves on ONE SIMD per CU?", 2, it, 60 * 1024, 1.f, 64);
  run<32>("16x16x32 bf16, 1 wave/SIMD, 32 CUs only", 1, it, 100 * 1024, 1.f, 256, 0, 32);
  run<32>("16x16x32 bf16, 1 wave/SIMD, 128 CUs only", 1, it, 100 * 1024, 1.f, 256, 0, 128);
  return 0;
}

// JAX-compatible counter-based PRNG on the device: Threefry-2x32-20 with jax's counter
// layout, so `split` / `uniform` reproduce the reference's jax.random streams bit for bit
// (train_utils.py:408 / models.py:38-43 random.split, stepfun.py:207-209 random.uniform).
// Algorithm and layout: oracle/threefry_ref.py (pinned by Random123 KATs and the reference's datasets_test golden).
// Integer work, HBM-write bound: 4 B written per 32-bit draw, ~110 integer ops per pair of draws.
#include "hugs_common.h"

namespace {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0];
  x1 += ks[1];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, rot[i & 1][j]) ^ x0;
    }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
}

// jax threefry_2x32(key, iota(n)): the counters are padded to even length, the first half feeds word 0, the
// second half word 1; output = concat(word-0 results, word-1 results)[:n].  Thread i owns the pair (i, h+i).
template <bool UNIFORM>
__global__ void __launch_bounds__(256)
k_threefry(const uint32_t* __restrict__ key, long long n, float lo, float hi, void* __restrict__ out) {
  long long h = (n + 1) >> 1;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= h) return;
  uint32_t x0 = (uint32_t)i, x1 = (h + i < n) ? (uint32_t)(h + i) : 0u;
  threefry2x32(key[0], key[1], x0, x1);
  if (UNIFORM) {   // random.py uniform: mantissa bits -> [1,2) - 1 -> scale, shift, clamp at minval
    float scale = __fsub_rn(hi, lo);
    float f0 = __fsub_rn(__uint_as_float((x0 >> 9) | 0x3F800000u), 1.f);
    float f1 = __fsub_rn(__uint_as_float((x1 >> 9) | 0x3F800000u), 1.f);
    ((float*)out)[i] = fmaxf(lo, __fadd_rn(__fmul_rn(f0, scale), lo));
    if (h + i < n) ((float*)out)[h + i] = fmaxf(lo, __fadd_rn(__fmul_rn(f1, scale), lo));
  } else {
    ((uint32_t*)out)[i] = x0;
    if (h + i < n) ((uint32_t*)out)[h + i] = x1;
  }
}

}  // namespace

extern "C" int hugs_prng_bits(const uint32_t* key, long long n, uint32_t* out, void* stream) {
  HUGS_REQUIRE(key && (out || n == 0), -2, "hugs_prng_bits: null pointer");
  HUGS_REQUIRE(n >= 0 && n < (1ll << 32), -2, "hugs_prng_bits: n=%lld outside [0, 2^32)", n);
  if (n == 0) return 0;
  long long h = (n + 1) >> 1;
  k_threefry<false><<<(unsigned)((h + 255) / 256), 256, 0, (hipStream_t)stream>>>(key, n, 0.f, 1.f, out);
  HUGS_CHECK_LAUNCH("k_threefry");
  return 0;
}

extern "C" int hugs_prng_uniform(const uint32_t* key, long long n, float minval, float maxval, float* out,
                                 void* stream) {
  HUGS_REQUIRE(key && (out || n == 0), -2, "hugs_prng_uniform: null pointer");
  HUGS_REQUIRE(n >= 0 && n < (1ll << 32), -2, "hugs_prng_uniform: n=%lld outside [0, 2^32)", n);
  if (n == 0) return 0;
  long long h = (n + 1) >> 1;
  k_threefry<true><<<(unsigned)((h + 255) / 256), 256, 0, (hipStream_t)stream>>>(key, n, minval, maxval, out);
  HUGS_CHECK_LAUNCH("k_threefry");
  return 0;
}

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
// erf^-1 in single precision: M. Giles, "Approximating the erfinv function" (2012), the polynomial pair XLA's ErfInv uses for
// f32 (w = -log((1 - x)(1 + x)); central branch w < 5, tail branch in sqrt(w)).  jax.random.normal is
// sqrt(2) * erf_inv(uniform(key, shape, minval = nextafter(-1, 0), maxval = 1)).
__device__ __forceinline__ float erfinv_f32(float x) {
  float w = -logf((1.f - x) * (1.f + x)), p;
  if (w < 5.f) {
    w -= 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, w, 3.43273939e-07f); p = fmaf(p, w, -3.5233877e-06f); p = fmaf(p, w, -4.39150654e-06f);
    p = fmaf(p, w, 0.00021858087f); p = fmaf(p, w, -0.00125372503f); p = fmaf(p, w, -0.00417768164f);
    p = fmaf(p, w, 0.246640727f); p = fmaf(p, w, 1.50140941f);
  } else {
    w = sqrtf(w) - 3.f;
    p = -0.000200214257f;
    p = fmaf(p, w, 0.000100950558f); p = fmaf(p, w, 0.00134934322f); p = fmaf(p, w, -0.00367342844f);
    p = fmaf(p, w, 0.00573950773f); p = fmaf(p, w, -0.0076224613f); p = fmaf(p, w, 0.00943887047f);
    p = fmaf(p, w, 1.00167406f); p = fmaf(p, w, 2.83297682f);
  }
  return p * x;
}

// MODE 0: raw bits, 1: uniform [lo, hi), 2: standard normal (lo / hi ignored)
template <int MODE>
__global__ void __launch_bounds__(256)
k_threefry(const uint32_t* __restrict__ key, long long n, float lo, float hi, void* __restrict__ out) {
  long long h = (n + 1) >> 1;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= h) return;
  uint32_t x0 = (uint32_t)i, x1 = (h + i < n) ? (uint32_t)(h + i) : 0u;
  threefry2x32(key[0], key[1], x0, x1);
  if (MODE == 2) {   // jax _normal_real: u in [nextafter(-1, 0), 1), then sqrt(2) erf_inv(u)
    const float nlo = -0.99999994f, scale = __fsub_rn(1.f, nlo);
    const float f0 = __fsub_rn(__uint_as_float((x0 >> 9) | 0x3F800000u), 1.f);
    const float f1 = __fsub_rn(__uint_as_float((x1 >> 9) | 0x3F800000u), 1.f);
    ((float*)out)[i] = 1.41421356f * erfinv_f32(fmaxf(nlo, __fadd_rn(__fmul_rn(f0, scale), nlo)));
    if (h + i < n) ((float*)out)[h + i] = 1.41421356f * erfinv_f32(fmaxf(nlo, __fadd_rn(__fmul_rn(f1, scale), nlo)));
  } else if (MODE == 1) {   // random.py uniform: mantissa bits -> [1,2) - 1 -> scale, shift, clamp at minval
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

// The whole key chain one training step consumes, in ONE launch (train_utils.py:408 `rng, key = random.split(rng)`, then per
// level models.py:196 `key, rng = random.split(rng)` -> stepfun.py:207-209 random.uniform(key, [n_l], maxval_l) and
// models.py:230 `key, rng = random.split(rng)` for the MLP key): every thread re-derives the (1 + 2 L) splits -- two
// threefry evaluations each, a few hundred integer ops -- and writes its pair of uniform draws of every level.
#define HUGS_STEP_JITTER_MAX_LEVELS 8      // the ONE definition of the cap (exported: hugs_prng_step_jitter_max_levels)
struct StepJitter {
  int L;
  long long n[HUGS_STEP_JITTER_MAX_LEVELS];
  float maxval[HUGS_STEP_JITTER_MAX_LEVELS];
  float* out[HUGS_STEP_JITTER_MAX_LEVELS];
};

__device__ __forceinline__ void split2(uint32_t k0, uint32_t k1, uint32_t (&a)[2], uint32_t (&b)[2]) {
  // bits(key, 4): thread pairs (0, 2) and (1, 3); first key = (out0, out1), second = (out2, out3)
  uint32_t x0 = 0u, x1 = 2u, y0 = 1u, y1 = 3u;
  threefry2x32(k0, k1, x0, x1);
  threefry2x32(k0, k1, y0, y1);
  a[0] = x0; a[1] = y0; b[0] = x1; b[1] = y1;
}

__global__ void __launch_bounds__(256)
k_step_jitter(const uint32_t* __restrict__ key_in, StepJitter J, uint32_t* __restrict__ key_out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  uint32_t rng[2], r[2], k[2], t_[2];
  split2(key_in[0], key_in[1], rng, r);             // rng, key = split(rng): `rng` goes back to the caller, `key` feeds the levels
  if (i == 0) { key_out[0] = rng[0]; key_out[1] = rng[1]; }
  for (int l = 0; l < J.L; ++l) {
    split2(r[0], r[1], k, t_);                       // key, rng = split(rng)
    const long long n = J.n[l], h = (n + 1) >> 1;
    if (i < h) {
      uint32_t x0 = (uint32_t)i, x1 = (h + i < n) ? (uint32_t)(h + i) : 0u;
      threefry2x32(k[0], k[1], x0, x1);
      const float hi = J.maxval[l];
      const float f0 = __fsub_rn(__uint_as_float((x0 >> 9) | 0x3F800000u), 1.f);
      const float f1 = __fsub_rn(__uint_as_float((x1 >> 9) | 0x3F800000u), 1.f);
      J.out[l][i] = fmaxf(0.f, __fadd_rn(__fmul_rn(f0, hi), 0.f));
      if (h + i < n) J.out[l][h + i] = fmaxf(0.f, __fadd_rn(__fmul_rn(f1, hi), 0.f));
    }
    split2(t_[0], t_[1], k, r);                      // _, rng = split(rng)
  }
}

}  // namespace

namespace {
__global__ void k_fold_in(const uint32_t* __restrict__ key, uint32_t data, uint32_t* __restrict__ out) {
  if (threadIdx.x == 0) {      // jax threefry_fold_in: threefry_2x32(key, threefry_seed(data)) with seed words (0, data)
    uint32_t x0 = 0u, x1 = data;
    threefry2x32(key[0], key[1], x0, x1);
    out[0] = x0; out[1] = x1;
  }
}
}  // namespace
extern "C" int hugs_prng_fold_in(const uint32_t* key, uint32_t data, uint32_t* key_out, void* stream) {
  HUGS_REQUIRE(key && key_out, -2, "hugs_prng_fold_in: null pointer");
  k_fold_in<<<1, 64, 0, (hipStream_t)stream>>>(key, data, key_out);
  HUGS_CHECK_LAUNCH("k_fold_in");
  return 0;
}

/* One training step's draws from the reference's jax.random stream in one launch: key_out = first half of split(key_in)
 * (train_utils.py:408), and for every level l < L (<= 8) out[l][0..n[l]) = random.uniform(k_l, [n[l]], maxval = maxval[l]) with
 * the keys of models.py:196,230.  n / maxval / out are HOST arrays.  Bit-identical to the split / uniform entry points. */
extern "C" int hugs_prng_step_jitter_max_levels(void) { return HUGS_STEP_JITTER_MAX_LEVELS; }
extern "C" int hugs_prng_step_jitter(const uint32_t* key_in, int L, const long long* n, const float* maxval, float* const* out,
                                     uint32_t* key_out, void* stream) {
  HUGS_REQUIRE(key_in && key_out && n && maxval && out && L >= 0 && L <= HUGS_STEP_JITTER_MAX_LEVELS, -2,
               "hugs_prng_step_jitter: null pointer or L=%d outside 0..%d", L, HUGS_STEP_JITTER_MAX_LEVELS);
  StepJitter J;
  J.L = L;
  long long hmax = 1;
  for (int l = 0; l < L; ++l) {
    HUGS_REQUIRE(n[l] >= 0 && n[l] < (1ll << 32) && (out[l] || n[l] == 0), -2, "hugs_prng_step_jitter: level %d", l);
    J.n[l] = n[l]; J.maxval[l] = maxval[l]; J.out[l] = out[l];
    const long long h = (n[l] + 1) >> 1;
    if (h > hmax) hmax = h;
  }
  k_step_jitter<<<(unsigned)((hmax + 255) / 256), 256, 0, (hipStream_t)stream>>>(key_in, J, key_out);
  HUGS_CHECK_LAUNCH("k_step_jitter");
  return 0;
}

extern "C" int hugs_prng_bits(const uint32_t* key, long long n, uint32_t* out, void* stream) {
  HUGS_REQUIRE(key && (out || n == 0), -2, "hugs_prng_bits: null pointer");
  HUGS_REQUIRE(n >= 0 && n < (1ll << 32), -2, "hugs_prng_bits: n=%lld outside [0, 2^32)", n);
  if (n == 0) return 0;
  long long h = (n + 1) >> 1;
  k_threefry<0><<<(unsigned)((h + 255) / 256), 256, 0, (hipStream_t)stream>>>(key, n, 0.f, 1.f, out);
  HUGS_CHECK_LAUNCH("k_threefry");
  return 0;
}

extern "C" int hugs_prng_uniform(const uint32_t* key, long long n, float minval, float maxval, float* out,
                                 void* stream) {
  HUGS_REQUIRE(key && (out || n == 0), -2, "hugs_prng_uniform: null pointer");
  HUGS_REQUIRE(n >= 0 && n < (1ll << 32), -2, "hugs_prng_uniform: n=%lld outside [0, 2^32)", n);
  if (n == 0) return 0;
  long long h = (n + 1) >> 1;
  k_threefry<1><<<(unsigned)((h + 255) / 256), 256, 0, (hipStream_t)stream>>>(key, n, minval, maxval, out);
  HUGS_CHECK_LAUNCH("k_threefry");
  return 0;
}

/* jax.random.normal(key, [n]) (models.py:458-460,478-481 density / bottleneck noise, and the flax initialisers' draws): the
 * uniform stream above through XLA's single-precision erf_inv polynomial.  Agrees with the float64 evaluation of the same
 * definition (oracle/threefry_ref.py normal) to a few ulp. */
extern "C" int hugs_prng_normal(const uint32_t* key, long long n, float* out, void* stream) {
  HUGS_REQUIRE(key && (out || n == 0), -2, "hugs_prng_normal: null pointer");
  HUGS_REQUIRE(n >= 0 && n < (1ll << 32), -2, "hugs_prng_normal: n=%lld outside [0, 2^32)", n);
  if (n == 0) return 0;
  long long h = (n + 1) >> 1;
  k_threefry<2><<<(unsigned)((h + 255) / 256), 256, 0, (hipStream_t)stream>>>(key, n, 0.f, 1.f, out);
  HUGS_CHECK_LAUNCH("k_threefry");
  return 0;
}

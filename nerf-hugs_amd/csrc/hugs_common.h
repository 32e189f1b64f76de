// Shared host/device helpers for the hugs HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define HUGS_EPS 1.1920928955078125e-07f /* finfo(float32).eps */
#define HUGS_WAVE 64

extern "C" void hugs_set_error(const char* fmt, ...);

#define HUGS_CHECK_LAUNCH(name)                                           \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess) {                                               \
      hugs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return -100;                                                        \
    }                                                                     \
  } while (0)

#define HUGS_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      hugs_set_error(__VA_ARGS__);     \
      return code;                     \
    }                                  \
  } while (0)

// ---- wave-level primitives (64 lanes) ----
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v = fminf(v, __shfl_xor(v, d));
  return v;
}
// inclusive Kogge-Stone scan across lanes
__device__ __forceinline__ float wave_incl_scan_f(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
// inclusive suffix scan (lane l gets sum over lanes >= l)
__device__ __forceinline__ float wave_incl_suffix_scan_f(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float t = __shfl_down(v, d);
    if (lane + d < 64) v += t;
  }
  return v;
}

__device__ __forceinline__ float bf16_to_f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f_to_bf16(float f) {  // round-to-nearest-even, NaN preserved
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// IEEE half (dtype 2: the nerfacto path's fp16 mode) and the run-time 16-bit operand format switch (dt: 1 = bf16, 2 = half)
__device__ __forceinline__ float h16_to_f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f_to_h16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // RNE; > 65504 -> inf
__device__ __forceinline__ float op16_to_f(uint16_t h, int dt) { return dt == 2 ? h16_to_f(h) : bf16_to_f(h); }
__device__ __forceinline__ uint16_t f_to_op16(float f, int dt) { return dt == 2 ? f_to_h16(f) : f_to_bf16(f); }
__device__ __forceinline__ uint32_t f2_to_op16(float a, float b, int dt) { return (uint32_t)f_to_op16(a, dt) | ((uint32_t)f_to_op16(b, dt) << 16); }

// nerfacto density activation (models/nerfacto.py:702-710, 910-918): act 0 = trunc_exp (custom_functions.py:38-52: exp forward, exp of the
// input clamped to [-15, 15] in the backward), act 1 = softplus(raw + density_bias) (F.softplus: linear above threshold 20).
__device__ __forceinline__ float nf_density_value(float raw, int act, float bias) {
  if (!act) return expf(raw);
  const float x = raw + bias;
  return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float nf_density_slope(float raw, int act, float bias) {
  if (!act) return expf(fminf(fmaxf(raw, -15.f), 15.f));
  const float x = raw + bias;
  return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
}

// Ray sections -> Gaussians -> (optional contraction) -> basis lift -> integrated positional
// encoding, fused: the [N,S,3,3] covariances and [N,S,21] lifted moments never touch HBM; the
// only output is the MLP input panel X[N*S, 512] (504 features + 8 zero columns so that rows
// are 16-byte aligned K-tiles for the MFMA GEMMs), written as whole 1 KiB / 2 KiB rows.
//
// Replaces (reference, MipNeRF360/internal): render.py:103-127 cast_rays -> :44-78 / :81-100 ->
// :21-41 lift_gaussian(diag=False); coord.py:21-27 contract + :39-60 track_linearize (closed-form
// Jacobian instead of jax.linearize); coord.py:129-133 lift_and_diagonalize; :102-126
// integrated_pos_enc with math.py:26-38 safe_sin; coord.py:136-147 pos_enc (view directions).
#include "hugs_common.h"

#define ENC_SAMPLES_F32 32    // samples per workgroup, fp32 parity build (unchanged since round 1: its instruction stream is pinned)
#define ENC_SAMPLES_BF16 64   // bf16 (round 5): the per-sample Gaussian step fills one whole wave
#define ENC_NB 21        // max basis directions (icosahedron, 2 subdivisions)

typedef float enc_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 enc_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t enc_pk_bf16(float a, float b) {      // one v_cvt_pk_bf16_f32 (RNE)
  const enc_f32x2_t f = {a, b};
  const enc_bf16x2_t h = __builtin_convertvector(f, enc_bf16x2_t);
  return *(const uint32_t*)&h;
}

__device__ __forceinline__ float safe_sinf(float x) {
  // math.py:26-38: sin(where(|x| < 100pi, x, x mod 100pi)), python-style mod
  const float t = 314.159265358979323846f;
  if (!(fabsf(x) < t)) { float r = fmodf(x, t); if (r != 0.0f && (r < 0.0f)) r += t; x = r; }
  return sinf(x);
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_cast_ipe(int nrays, int S, const float* __restrict__ tdist,
                                                  const float* __restrict__ origins, const float* __restrict__ dirs,
                                                  const float* __restrict__ radii, const float* __restrict__ basis, int nb,
                                                  int ray_shape, int warp, int max_deg, int kp, void* __restrict__ out) {
  constexpr int ENC_SAMPLES = BF16 ? ENC_SAMPLES_BF16 : ENC_SAMPLES_F32;
  __shared__ float s_mean[ENC_SAMPLES][3];
  __shared__ float s_cov[ENC_SAMPLES][6];
  __shared__ float s_lm[ENC_SAMPLES][ENC_NB + 1];
  __shared__ float s_lv[ENC_SAMPLES][ENC_NB + 1];
  __shared__ float s_basis[3 * ENC_NB];
  const int tid = threadIdx.x;
  const long long total = (long long)nrays * S;
  const long long base = (long long)blockIdx.x * ENC_SAMPLES;
  if (tid < 3 * nb) s_basis[tid] = basis[tid];
  if (tid < ENC_SAMPLES && base + tid < total) {
    const long long m = base + tid;
    const int ray = (int)(m / S), s = (int)(m % S);
    const float t0 = tdist[(size_t)ray * (S + 1) + s], t1 = tdist[(size_t)ray * (S + 1) + s + 1];
    const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float rad = radii[ray];
    float t_mean, t_var, r_var;
    if ((ray_shape & 3) == 0) {  // cone, render.py:62-70
      const float mu = (t0 + t1) / 2, hw = (t1 - t0) / 2;
      const float den = fmaxf(HUGS_EPS, 3 * mu * mu + hw * hw);
      const float hw2 = hw * hw, hw4 = hw2 * hw2;
      t_mean = mu + (2 * mu * hw2) / den;
      t_var = hw2 / 3 - (4.0f / 15.0f) * hw4 * (12 * mu * mu - hw2) / (den * den);
      r_var = (mu * mu) / 4 + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / den;
      r_var *= rad * rad;
    } else {               // cylinder, render.py:96-99
      t_mean = (t0 + t1) / 2;
      r_var = rad * rad / 4;
      t_var = (t1 - t0) * (t1 - t0) / 12;
    }
    float mx = dx * t_mean + origins[ray * 3], my = dy * t_mean + origins[ray * 3 + 1],
          mz = dz * t_mean + origins[ray * 3 + 2];
    const float dms = fmaxf(1e-10f, dx * dx + dy * dy + dz * dz);
    const float ex = dx / dms, ey = dy / dms, ez = dz / dms;
    // cov = t_var d d^T + r_var (I - d (d/|d|^2)^T); symmetric -> 6 entries xx xy xz yy yz zz
    float c[6];
    c[0] = t_var * dx * dx + r_var * (1 - dx * ex);
    c[1] = t_var * dx * dy + r_var * (0 - dx * ey);
    c[2] = t_var * dx * dz + r_var * (0 - dx * ez);
    c[3] = t_var * dy * dy + r_var * (1 - dy * ey);
    c[4] = t_var * dy * dz + r_var * (0 - dy * ez);
    c[5] = t_var * dz * dz + r_var * (1 - dz * ez);
    if (ray_shape & 4) {   // Model.disable_integration (models.py:223-226): zero covariances, "PE instead of IPE"
#pragma unroll
      for (int k = 0; k < 6; ++k) c[k] = 0.f;
    }
    if (warp) {  // coord.py:21-27,39-60 with the closed-form Jacobian of contract
      const float n2 = fmaxf(HUGS_EPS, mx * mx + my * my + mz * mz);
      if (n2 > 1.0f) {
        const float n = sqrtf(n2);
        const float sc = (2 * n - 1) / n2;
        const float ds2 = 2 * ((1 / n) / n2 - (2 * n - 1) / (n2 * n2));
        const float x[3] = {mx, my, mz};
        float J[3][3], C[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}}, T[3][3];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) J[a][b] = (a == b ? sc : 0.0f) + ds2 * x[a] * x[b];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) T[a][b] = J[a][0] * C[0][b] + J[a][1] * C[1][b] + J[a][2] * C[2][b];
        float R[3][3];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) R[a][b] = T[a][0] * J[b][0] + T[a][1] * J[b][1] + T[a][2] * J[b][2];
        c[0] = R[0][0]; c[1] = R[0][1]; c[2] = R[0][2]; c[3] = R[1][1]; c[4] = R[1][2]; c[5] = R[2][2];
        mx *= sc; my *= sc; mz *= sc;
      }
    }
    s_mean[tid][0] = mx; s_mean[tid][1] = my; s_mean[tid][2] = mz;
    for (int k = 0; k < 6; ++k) s_cov[tid][k] = c[k];
  }
  __syncthreads();
  for (int e = tid; e < ENC_SAMPLES * nb; e += 256) {  // coord.py:129-133
    const int s = e / nb, j = e % nb;
    const float b0 = s_basis[j], b1 = s_basis[nb + j], b2 = s_basis[2 * nb + j];
    const float* c = s_cov[s];
    s_lm[s][j] = s_mean[s][0] * b0 + s_mean[s][1] * b1 + s_mean[s][2] * b2;
    const float r0 = c[0] * b0 + c[1] * b1 + c[2] * b2;
    const float r1 = c[1] * b0 + c[3] * b1 + c[4] * b2;
    const float r2 = c[2] * b0 + c[4] * b1 + c[5] * b2;
    s_lv[s][j] = b0 * r0 + b1 * r1 + b2 * r2;
    if (BF16) {      // the bf16 feature loop's operands: revolutions and the base-2 exponent (powers of two scale both exactly)
      s_lm[s][j] *= 0.15915494309189533577f;
      s_lv[s][j] *= -0.72134752044448170368f;      // -0.5 log2(e)
    }
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6;
  const int nfeat_half = nb * max_deg;
  if constexpr (BF16) {
    if ((nfeat_half & 3) == 0) {
      // Round 5 (the loop below spent ~30 vector instructions per feature and ran at 2.9 TB/s of output at 1 M samples): a lane owns
      // FOUR (degree k, basis j) columns and produces both halves of each -- sin(x) and sin(x + pi/2) share the attenuation
      // exp(-var/2), the argument is kept in revolutions (a power-of-two scale is exact: fract, v_sin_f32) and the exponent in base 2
      // (v_exp_f32), pairs leave through v_cvt_pk_bf16_f32.  8 bytes per lane into the sin half, 8 into the cos half of the row.
      for (int c0 = lane * 4; c0 < nfeat_half; c0 += 256) {
        int jq[4];
        float scs[4], sc2[4];
        {
          int k = c0 / nb, j = c0 - k * nb;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            jq[q] = j;
            scs[q] = __int_as_float((127 + ((ray_shape >> 8) & 0xff) + k) << 23);      // 2^(min_deg + k)
            sc2[q] = scs[q] * scs[q];
            if (++j == nb) { j = 0; ++k; }
          }
        }
        for (int s = wv; s < ENC_SAMPLES; s += 4) {
          const long long m = base + s;
          if (m >= total) break;
          float vs[4], vc[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float rev = s_lm[s][jq[q]] * scs[q];
            const float att = __builtin_amdgcn_exp2f(s_lv[s][jq[q]] * sc2[q]);
            const float rc = rev + 0.25f;          // sin(x + pi/2), as the reference does
            vs[q] = att * __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev));
            vc[q] = att * __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rc));
          }
          uint16_t* row = (uint16_t*)out + (size_t)m * kp;
          *(uint2*)(row + c0) = make_uint2(enc_pk_bf16(vs[0], vs[1]), enc_pk_bf16(vs[2], vs[3]));
          *(uint2*)(row + nfeat_half + c0) = make_uint2(enc_pk_bf16(vc[0], vc[1]), enc_pk_bf16(vc[2], vc[3]));
        }
      }
      // padding columns [2 nfeat_half, kp): zeros (kp and 2 nfeat_half are multiples of 8 and 4: 8-byte pieces)
      const int npad4 = (kp - 2 * nfeat_half) >> 2;
      for (int e = tid; e < ENC_SAMPLES * npad4; e += 256) {
        const int s = e / npad4, c = e - s * npad4;
        const long long m = base + s;
        if (m < total) *(uint2*)((uint16_t*)out + (size_t)m * kp + 2 * nfeat_half + 4 * c) = make_uint2(0u, 0u);
      }
      return;
    }
  }
  // one wave writes one whole row: lane owns 8 consecutive features.  Which (half, degree k, basis j) a feature column
  // is does not depend on the sample: decoded once per lane (no integer division in the per-sample loop).
  for (int f0 = lane * 8; f0 < kp; f0 += 512) {
    int jj[8];         // j | k << 8 | half << 16; half == 2: padding column
    float scs[8];
    {
      int half = f0 >= nfeat_half ? (f0 >= 2 * nfeat_half ? 2 : 1) : 0;
      const int r = f0 - (half == 1 ? nfeat_half : 0);
      int k = half < 2 ? r / nb : 0, j = half < 2 ? r - k * nb : 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        jj[q] = j | (half << 16);
        scs[q] = __int_as_float((127 + ((ray_shape >> 8) & 0xff) + k) << 23);      // 2^(min_deg + k)
        if (half < 2 && ++j == nb) { j = 0; if (++k == max_deg) { k = 0; ++half; } }
      }
    }
    for (int s = wv; s < ENC_SAMPLES; s += 4) {
      const long long m = base + s;
      if (m >= total) break;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int half = jj[q] >> 16, j = jj[q] & 0xffff;
        float val = 0.0f;
        if (half < 2) {
          const float sc = scs[q];
          float x = s_lm[s][j] * sc;
          if (half) x = x + (BF16 ? 0.25f : 1.57079632679489661923f);   // sin(x + pi/2), as the reference does
          const float var = s_lv[s][j] * (sc * sc);
          if (BF16) {
            // (general basis sizes: x / var are already in revolutions / base-2 exponents, see the lift step above)
            val = __builtin_amdgcn_exp2f(var) * __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x));
          } else {
            val = expf(-0.5f * var) * safe_sinf(x);
          }
        }
        v[q] = val;
      }
      if (BF16) {
        uint4 pk;
        pk.x = enc_pk_bf16(v[0], v[1]); pk.y = enc_pk_bf16(v[2], v[3]); pk.z = enc_pk_bf16(v[4], v[5]); pk.w = enc_pk_bf16(v[6], v[7]);
        *(uint4*)((uint16_t*)out + (size_t)m * kp + f0) = pk;
      } else {
        float4* o = (float4*)((float*)out + (size_t)m * kp + f0);
        o[0] = make_float4(v[0], v[1], v[2], v[3]);
        o[1] = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
  }
}

// coord.py:136-147 pos_enc(viewdirs, 0, deg, append_identity=True) -> [N, 3 + 6*deg]
__global__ void k_dir_enc(int nrays, int deg, const float* __restrict__ viewdirs, float* __restrict__ out) {
  const int W = 3 + 6 * deg;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrays * W) return;
  const int ray = i / W, f = i % W;
  float val;
  if (f < 3) {
    val = viewdirs[ray * 3 + f];
  } else {
    const int r = f - 3, half = r >= 3 * deg, q = r - half * 3 * deg;
    const int k = q / 3, c = q % 3;
    float x = viewdirs[ray * 3 + c] * (float)(1 << k);
    if (half) x = x + 1.57079632679489661923f;
    val = sinf(x);
  }
  out[i] = val;
}

extern "C" int hugs_cast_ipe_fwd(int nrays, int num_samples, const float* tdist, const float* origins,
                                 const float* directions, const float* radii, const float* basis, int num_basis,
                                 int ray_shape, int warp_contract, int max_deg, int out_bf16, int row_pitch, void* out,
                                 void* stream) {
  HUGS_REQUIRE((ray_shape & 3) <= 1 && (ray_shape & ~0xff07) == 0 && ((ray_shape >> 8) & 0xff) + max_deg <= 30, -2,
               "ray_shape must be 'cone' or 'cylinder' (+4: zero covariances; bits 8-15: min_deg)");
  HUGS_REQUIRE(num_basis >= 1 && num_basis <= ENC_NB && max_deg >= 1 && max_deg <= 24, -3,
               "hugs_cast_ipe_fwd: basis size %d / max_deg %d unsupported", num_basis, max_deg);
  HUGS_REQUIRE(row_pitch % 8 == 0 && row_pitch >= 2 * num_basis * max_deg, -3,
               "hugs_cast_ipe_fwd: row pitch %d too small / not a multiple of 8", row_pitch);
  const long long total = (long long)nrays * num_samples;
  if (total <= 0) return 0;
  const int grid = (int)((total + ENC_SAMPLES_F32 - 1) / ENC_SAMPLES_F32), grid_b = (int)((total + ENC_SAMPLES_BF16 - 1) / ENC_SAMPLES_BF16);
  if (out_bf16)
    hipLaunchKernelGGL(k_cast_ipe<true>, dim3(grid_b), dim3(256), 0, (hipStream_t)stream, nrays, num_samples, tdist,
                       origins, directions, radii, basis, num_basis, ray_shape, warp_contract, max_deg, row_pitch, out);
  else
    hipLaunchKernelGGL(k_cast_ipe<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, nrays, num_samples, tdist,
                       origins, directions, radii, basis, num_basis, ray_shape, warp_contract, max_deg, row_pitch, out);
  HUGS_CHECK_LAUNCH("hugs_cast_ipe_fwd");
  return 0;
}

extern "C" int hugs_dir_enc_fwd(int nrays, int deg, const float* viewdirs, float* out, void* stream) {
  const int n = nrays * (3 + 6 * deg);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_dir_enc, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, nrays, deg, viewdirs, out);
  HUGS_CHECK_LAUNCH("hugs_dir_enc_fwd");
  return 0;
}

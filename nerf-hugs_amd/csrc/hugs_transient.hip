// HA-NeRF branch of the hot path: the per-ray ImplicitMask MLP's non-GEMM pieces and its loss.
//   models.py:651-674 ImplicitMask: x = [pos_enc(pix_coords, 0, deg, identity) | tra_vec] -> (Dense+relu) x depth
//   -> sigmoid(Dense(1)); train_utils.py:186-225 compute_hanerf_loss.  The Dense+relu layers run on the shared
//   GEMM kernels (hugs_gemm_nt / hugs_gemm_tn); everything here is per-ray and HBM-trivial (<= 8192 rays).
#include "hugs_common.h"

namespace {

// X[n, :] = [x0, x1, sin(2^k x_c), sin(2^k x_c + pi/2), tra[n, :], 0 ...]   (coord.py:136-147 layout: k-major, c-minor)
template <bool BF16>
__global__ void __launch_bounds__(256)
k_mask_input(int N, int T, int deg, const float* __restrict__ pix, const float* __restrict__ tra, int kpad,
             void* __restrict__ X) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * kpad) return;
  const int n = i / kpad, f = i - n * kpad;
  const int E = 2 + 4 * deg;
  float v = 0.f;
  if (f < 2) {
    v = pix[n * 2 + f];
  } else if (f < E) {
    const int r = f - 2, half = r >= 2 * deg, q = r - half * 2 * deg;
    const int k = q >> 1, c = q & 1;
    float x = pix[n * 2 + c] * (float)(1 << k);
    if (half) x = x + 1.57079632679489661923f;
    v = sinf(x);
  } else if (f < E + T) {
    v = tra ? tra[(size_t)n * T + (f - E)] : 0.f;
  }
  if (BF16) ((uint16_t*)X)[i] = f_to_bf16(v);
  else ((float*)X)[i] = v;
}

template <bool BF16>
__device__ __forceinline__ float ldx(const void* X, size_t i) {
  return BF16 ? bf16_to_f(((const uint16_t*)X)[i]) : ((const float*)X)[i];
}

// mask[n] = sigmoid(X[n,:] . w + b), one wave per ray
template <bool BF16>
__global__ void __launch_bounds__(256)
k_mask_head_fwd(int N, int W, const void* __restrict__ X, int ldxv, const float* __restrict__ w,
                const float* __restrict__ b, float* __restrict__ mask) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float a = 0.f;
  for (int k = lane; k < W; k += 64) a += ldx<BF16>(X, (size_t)n * ldxv + k) * w[k];
  a = wave_sum_f(a);
  if (lane == 0) mask[n] = 1.f / (1.f + expf(-(a + b[0])));
}

// d_raw[n] = d_mask[n] * m (1 - m)   (rows >= N of the padded batch get 0)
__global__ void __launch_bounds__(256)
k_mask_head_draw(int N, int Npad, const float* __restrict__ mask, const float* __restrict__ d_mask,
                 float* __restrict__ d_raw) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= Npad) return;
  float m = n < N ? mask[n] : 0.f;
  d_raw[n] = n < N ? d_mask[n] * m * (1.f - m) : 0.f;
}

// dW[k] = sum_n d_raw[n] X[n,k]; db = sum_n d_raw[n].  One thread per column, fixed order (deterministic).
template <bool BF16>
__global__ void __launch_bounds__(64)
k_mask_head_dw(int N, int W, const void* __restrict__ X, int ldxv, const float* __restrict__ d_raw,
               float* __restrict__ dW, float* __restrict__ db) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k < W) {
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += d_raw[n] * ldx<BF16>(X, (size_t)n * ldxv + k);
    dW[k] = a;
  }
  if (k == W) {
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += d_raw[n];
    db[0] = a;
  }
}

// d_embedding[embed_idx[n], :] += dX[n, col0 : col0+T]
template <bool BF16>
__global__ void __launch_bounds__(256)
k_embed_scatter(int N, int T, const void* __restrict__ dX, int ldxv, int col0, const int* __restrict__ embed_idx,
                float* __restrict__ d_emb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * T) return;
  const int n = i / T, t = i - n * T;
  atomicAdd(d_emb + (size_t)embed_idx[n] * T + t, ldx<BF16>(dX, (size_t)n * ldxv + col0 + t));
}

__device__ __forceinline__ float block_sum1024(float v, float* red) {
  v = wave_sum_f(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// train_utils.py:186-225.  stats: [2l] = mean resid^2, [2l+1] = mean((1-m) * dl) per level, [2L] = mean(m^2),
// [2L+1] = mean(m).  d_pred[l,n,c] = coef[l] (1-m) dl'/(3N); d_mask = -coef[L-1] sum_c dl /(3N) + 2 mult m / N.
__global__ void __launch_bounds__(1024)
k_hanerf_loss(int N, int L, const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ mask,
              int charb, float pad, const float* __restrict__ coef, float mask_mult, float* __restrict__ d_pred,
              float* __restrict__ d_mask, float* __restrict__ stats) {
  __shared__ float red[16];
  const float inv = 1.f / (3.f * (float)N);
  for (int l = 0; l < L; ++l) {
    const float cf = coef[l];
    float s_mse = 0.f, s_loss = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      const float m = mask[n], wgt = 1.f - m;
      float row = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t ix = ((size_t)l * N + n) * 3 + c;
        const float r = pred[ix] - gt[(size_t)n * 3 + c];
        const float r2 = r * r;
        float dl, ddl;
        if (charb) { dl = sqrtf(r2 + pad * pad); ddl = r / dl; } else { dl = r2; ddl = 2.f * r; }
        s_mse += r2;
        s_loss += wgt * dl;
        row += dl;
        d_pred[ix] = cf * wgt * ddl * inv;
      }
      if (l == L - 1) d_mask[n] = -cf * row * inv + 2.f * mask_mult * m / (float)N;
    }
    s_mse = block_sum1024(s_mse, red);
    s_loss = block_sum1024(s_loss, red);
    if (threadIdx.x == 0) { stats[2 * l] = s_mse * inv; stats[2 * l + 1] = s_loss * inv; }
  }
  float s2 = 0.f, s1 = 0.f;
  for (int n = threadIdx.x; n < N; n += blockDim.x) { const float m = mask[n]; s2 += m * m; s1 += m; }
  s2 = block_sum1024(s2, red);
  s1 = block_sum1024(s1, red);
  if (threadIdx.x == 0) { stats[2 * L] = s2 / (float)N; stats[2 * L + 1] = s1 / (float)N; }
}

}  // namespace

extern "C" int hugs_mask_input_fwd(int N, int T, int deg, const float* pix_coords, const float* tra_vec, int kpad,
                                   int dtype, void* X, void* stream) {
  HUGS_REQUIRE(N >= 0 && T >= 0 && deg >= 0 && deg < 31 && kpad >= 2 + 4 * deg + T, -2,
               "hugs_mask_input_fwd: N=%d T=%d deg=%d kpad=%d", N, T, deg, kpad);
  if (N == 0) return 0;
  const int n = N * kpad;
  if (dtype) k_mask_input<true><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, deg, pix_coords, tra_vec, kpad, X);
  else k_mask_input<false><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, deg, pix_coords, tra_vec, kpad, X);
  HUGS_CHECK_LAUNCH("k_mask_input");
  return 0;
}

extern "C" int hugs_mask_head_fwd(int dtype, int N, int W, const void* X, int ldx, const float* w, const float* b,
                                  float* mask, void* stream) {
  if (N <= 0) return 0;
  if (dtype) k_mask_head_fwd<true><<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(N, W, X, ldx, w, b, mask);
  else k_mask_head_fwd<false><<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(N, W, X, ldx, w, b, mask);
  HUGS_CHECK_LAUNCH("k_mask_head_fwd");
  return 0;
}

/* d_raw [Npad] (rows >= N zero), dW [W], db [1]; the caller forms G = (d_raw (x) w) * (X > 0) with hugs_rank1_mask */
extern "C" int hugs_mask_head_bwd(int dtype, int N, int Npad, int W, const void* X, int ldx, const float* mask,
                                  const float* d_mask, float* d_raw, float* dW, float* db, void* stream) {
  HUGS_REQUIRE(Npad >= N && N >= 0, -2, "hugs_mask_head_bwd: N=%d Npad=%d", N, Npad);
  if (Npad == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  k_mask_head_draw<<<(Npad + 255) / 256, 256, 0, st>>>(N, Npad, mask, d_mask, d_raw);
  if (dtype) k_mask_head_dw<true><<<W / 64 + 1, 64, 0, st>>>(N, W, X, ldx, d_raw, dW, db);
  else k_mask_head_dw<false><<<W / 64 + 1, 64, 0, st>>>(N, W, X, ldx, d_raw, dW, db);
  HUGS_CHECK_LAUNCH("k_mask_head_bwd");
  return 0;
}

extern "C" int hugs_embed_scatter_add(int dtype, int N, int T, const void* dX, int ldx, int col0, const int* embed_idx,
                                      float* d_embedding, void* stream) {
  if (N <= 0 || T <= 0) return 0;
  const int n = N * T;
  if (dtype) k_embed_scatter<true><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, dX, ldx, col0, embed_idx, d_embedding);
  else k_embed_scatter<false><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, dX, ldx, col0, embed_idx, d_embedding);
  HUGS_CHECK_LAUNCH("k_embed_scatter");
  return 0;
}

extern "C" int hugs_hanerf_loss(int N, int L, const float* pred, const float* gt, const float* mask, int charb,
                                float charb_pad, const float* coef, float mask_size_mult, float* d_pred, float* d_mask,
                                float* out_stats, void* stream) {
  HUGS_REQUIRE(N > 0 && L >= 1 && L <= 8, -2, "hugs_hanerf_loss: N=%d L=%d", N, L);
  k_hanerf_loss<<<1, 1024, 0, (hipStream_t)stream>>>(N, L, pred, gt, mask, charb, charb_pad, coef, mask_size_mult, d_pred,
                                                     d_mask, out_stats);
  HUGS_CHECK_LAUNCH("k_hanerf_loss");
  return 0;
}

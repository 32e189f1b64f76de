// HA-NeRF branch of the hot path: the per-ray ImplicitMask MLP's non-GEMM pieces and its loss.
//   models.py:651-674 ImplicitMask: x = [pos_enc(pix_coords, 0, deg, identity) | tra_vec] -> (Dense+relu) x depth
//   -> sigmoid(Dense(1)); train_utils.py:186-225 compute_hanerf_loss.  The Dense+relu layers run on the shared
//   GEMM kernels (hugs_gemm_nt / hugs_gemm_tn); everything here is per-ray and HBM-trivial (<= 8192 rays).
#include "hugs_common.h"

namespace {

// X[n, :] = [x0, x1, sin(2^k x_c), sin(2^k x_c + pi/2), tra[n, :], 0 ...]   (coord.py:136-147 layout: k-major, c-minor)
template <int BF16>
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
  if (BF16) ((uint16_t*)X)[i] = f_to_op16(v, BF16);
  else ((float*)X)[i] = v;
}

template <int BF16>
__device__ __forceinline__ float ldx(const void* X, size_t i) {
  return BF16 ? op16_to_f(((const uint16_t*)X)[i], BF16) : ((const float*)X)[i];
}

// mask[n] = sigmoid(X[n,:] . w + b), one wave per ray
template <int BF16>
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

// dW[k] = sum_n d_raw[n] X[n,k]; db = sum_n d_raw[n] (column W).  256 threads = 16 columns x 16 row groups, the groups' partial sums
// added in LDS in group order: deterministic.  (Round 5: one thread per column walked all N rows -- 1024 dependent loads, 386 us
// of the HA-NeRF step.)
template <int BF16>
__global__ void __launch_bounds__(256)
k_mask_head_dw(int N, int W, const void* __restrict__ X, int ldxv, const float* __restrict__ d_raw,
               float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + c;
  float a = 0.f;
  if (k < W) {
#pragma unroll 4
    for (int n = rg; n < N; n += 16) a += d_raw[n] * ldx<BF16>(X, (size_t)n * ldxv + k);
  } else if (k == W) {
#pragma unroll 4
    for (int n = rg; n < N; n += 16) a += d_raw[n];
  }
  red[rg][c] = a;
  __syncthreads();
  if (rg == 0 && k <= W) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][c];
    if (k < W) dW[k] = t; else db[0] = t;
  }
}

// d_embedding[embed_idx[n], :] += dX[n, col0 : col0+T]
template <int BF16>
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
              float* __restrict__ d_mask, float* __restrict__ stats, const float* __restrict__ mask_mult_dev) {
  __shared__ float red[16];
  if (mask_mult_dev) mask_mult = *mask_mult_dev;      // hugs_hanerf_loss_dyn: the step's value lives in device memory (captured step)
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
  if (dtype == 2) k_mask_input<2><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, deg, pix_coords, tra_vec, kpad, X);
  else if (dtype) k_mask_input<1><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, deg, pix_coords, tra_vec, kpad, X);
  else k_mask_input<0><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, deg, pix_coords, tra_vec, kpad, X);
  HUGS_CHECK_LAUNCH("k_mask_input");
  return 0;
}

extern "C" int hugs_mask_head_fwd(int dtype, int N, int W, const void* X, int ldx, const float* w, const float* b,
                                  float* mask, void* stream) {
  if (N <= 0) return 0;
  if (dtype == 2) k_mask_head_fwd<2><<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(N, W, X, ldx, w, b, mask);
  else if (dtype) k_mask_head_fwd<1><<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(N, W, X, ldx, w, b, mask);
  else k_mask_head_fwd<0><<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(N, W, X, ldx, w, b, mask);
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
  if (dtype == 2) k_mask_head_dw<2><<<W / 16 + 1, 256, 0, st>>>(N, W, X, ldx, d_raw, dW, db);
  else if (dtype) k_mask_head_dw<1><<<W / 16 + 1, 256, 0, st>>>(N, W, X, ldx, d_raw, dW, db);
  else k_mask_head_dw<0><<<W / 16 + 1, 256, 0, st>>>(N, W, X, ldx, d_raw, dW, db);
  HUGS_CHECK_LAUNCH("k_mask_head_bwd");
  return 0;
}

extern "C" int hugs_embed_scatter_add(int dtype, int N, int T, const void* dX, int ldx, int col0, const int* embed_idx,
                                      float* d_embedding, void* stream) {
  if (N <= 0 || T <= 0) return 0;
  const int n = N * T;
  if (dtype == 2) k_embed_scatter<2><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, dX, ldx, col0, embed_idx, d_embedding);
  else if (dtype) k_embed_scatter<1><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, dX, ldx, col0, embed_idx, d_embedding);
  else k_embed_scatter<0><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, T, dX, ldx, col0, embed_idx, d_embedding);
  HUGS_CHECK_LAUNCH("k_embed_scatter");
  return 0;
}

extern "C" int hugs_hanerf_loss(int N, int L, const float* pred, const float* gt, const float* mask, int charb,
                                float charb_pad, const float* coef, float mask_size_mult, float* d_pred, float* d_mask,
                                float* out_stats, void* stream) {
  HUGS_REQUIRE(N > 0 && L >= 1 && L <= 8, -2, "hugs_hanerf_loss: N=%d L=%d", N, L);
  k_hanerf_loss<<<1, 1024, 0, (hipStream_t)stream>>>(N, L, pred, gt, mask, charb, charb_pad, coef, mask_size_mult, d_pred,
                                                     d_mask, out_stats, nullptr);
  HUGS_CHECK_LAUNCH("k_hanerf_loss");
  return 0;
}
// mask_size_mult (train_utils.py:190-193: it decays with the step) read from one device float: a captured step's arguments are fixed
extern "C" int hugs_hanerf_loss_dyn(int N, int L, const float* pred, const float* gt, const float* mask, int charb,
                                    float charb_pad, const float* coef, const float* mask_size_mult_dev, float* d_pred, float* d_mask,
                                    float* out_stats, void* stream) {
  HUGS_REQUIRE(N > 0 && L >= 1 && L <= 8 && mask_size_mult_dev, -2, "hugs_hanerf_loss_dyn: N=%d L=%d", N, L);
  k_hanerf_loss<<<1, 1024, 0, (hipStream_t)stream>>>(N, L, pred, gt, mask, charb, charb_pad, coef, 0.f, d_pred, d_mask, out_stats,
                                                     mask_size_mult_dev);
  HUGS_CHECK_LAUNCH("k_hanerf_loss");
  return 0;
}

// =====================================================================================================
// NeRF-W branch: static + transient compositing, one WAVEFRONT per ray (<= 256 samples, 4 per
// lane in registers, wave prefix / suffix scans), and compute_nerfw_loss.
//   render.py:154-182 compute_dual_alpha_weights, :246-273 volumetric_rendering_combined_color,
//   models.py:299-307 uncertainty = sum_i w^t_i u_i + beta_min (w^t from the transient density alone),
//   train_utils.py:150-183 compute_nerfw_loss.
// =====================================================================================================
namespace {

// DC_MAXC = samples per lane (template parameter): 4 -> <= 256 samples per level, 8 -> 512, 16 -> 1024
template <int DC_MAXC>
struct DualScan {
  float a[DC_MAXC], b[DC_MAXC];      // sigma_s * delta, sigma_t * delta (+inf on the last sample when opaque)
  float T[DC_MAXC], Tt[DC_MAXC];     // transmittance of (a+b) and of b alone, before the sample
  float dl[DC_MAXC];                 // delta
  bool ok[DC_MAXC], last[DC_MAXC];
};

template <int DC_MAXC>
__device__ __forceinline__ void dual_scan(int S, int C, int lane, const float* __restrict__ ds, const float* __restrict__ dt_,
                                          const float* __restrict__ td, float dnorm, int opaque, DualScan<DC_MAXC>& R) {
  float tg = 0.f, tb = 0.f;
#pragma unroll
  for (int k = 0; k < DC_MAXC; ++k) {
    const int i = lane * C + k;
    R.ok[k] = k < C && i < S;
    R.last[k] = R.ok[k] && i == S - 1;
    float a = 0.f, b = 0.f, d = 0.f;
    if (R.ok[k]) {
      d = (td[i + 1] - td[i]) * dnorm;
      a = ds[i] * d;
      b = dt_[i] * d;
    }
    R.dl[k] = d;
    if (R.ok[k] && !R.last[k]) { tg += a + b; tb += b; }   // cumsum(...[:-1]): the last interval never enters a prefix
    if (R.last[k] && opaque) { a = __builtin_inff(); b = __builtin_inff(); }
    R.a[k] = a; R.b[k] = b;
  }
  const float ig = wave_incl_scan_f(tg, lane), ib = wave_incl_scan_f(tb, lane);
  float rg = __shfl_up(ig, 1), rb = __shfl_up(ib, 1);
  if (lane == 0) { rg = 0.f; rb = 0.f; }
#pragma unroll
  for (int k = 0; k < DC_MAXC; ++k) {
    R.T[k] = expf(-rg);
    R.Tt[k] = expf(-rb);
    if (R.ok[k] && !R.last[k]) { rg += R.a[k] + R.b[k]; rb += R.b[k]; }
  }
}

template <int DC_MAXC>
__global__ void __launch_bounds__(256)
k_dual_composite_fwd(int nrays, int S, const float* __restrict__ dens_s, const float* __restrict__ dens_t,
                     const float* __restrict__ rgb_s, const float* __restrict__ rgb_t, const float* __restrict__ unc,
                     const float* __restrict__ tdist, const float* __restrict__ dirs, int opaque, float bg, float beta_min,
                     float* __restrict__ rgb_comb, float* __restrict__ rgb_static, float* __restrict__ rgb_trans,
                     float* __restrict__ beta) {
  const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= nrays) return;
  const int C = (S + 63) >> 6;
  const float* td = tdist + (size_t)ray * (S + 1);
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  DualScan<DC_MAXC> R;
  dual_scan<DC_MAXC>(S, C, lane, dens_s + (size_t)ray * S, dens_t + (size_t)ray * S, td, dnorm, opaque, R);
  float acc = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, be = 0.f;
#pragma unroll
  for (int k = 0; k < DC_MAXC; ++k) {
    if (!R.ok[k]) continue;
    const size_t i = (size_t)ray * S + lane * C + k;
    const float w1 = (1.f - expf(-R.a[k])) * R.T[k], w2 = (1.f - expf(-R.b[k])) * R.T[k];
    acc += (1.f - expf(-(R.a[k] + R.b[k]))) * R.T[k];
    s0 += w1 * rgb_s[i * 3]; s1 += w1 * rgb_s[i * 3 + 1]; s2 += w1 * rgb_s[i * 3 + 2];
    t0 += w2 * rgb_t[i * 3]; t1 += w2 * rgb_t[i * 3 + 1]; t2 += w2 * rgb_t[i * 3 + 2];
    be += (1.f - expf(-R.b[k])) * R.Tt[k] * unc[i];
  }
  acc = wave_sum_f(acc); be = wave_sum_f(be);
  s0 = wave_sum_f(s0); s1 = wave_sum_f(s1); s2 = wave_sum_f(s2);
  t0 = wave_sum_f(t0); t1 = wave_sum_f(t1); t2 = wave_sum_f(t2);
  if (lane == 0) {
    const float bgw = fmaxf(0.f, 1.f - acc) * bg;
    rgb_static[ray * 3] = s0; rgb_static[ray * 3 + 1] = s1; rgb_static[ray * 3 + 2] = s2;
    rgb_trans[ray * 3] = t0; rgb_trans[ray * 3 + 1] = t1; rgb_trans[ray * 3 + 2] = t2;
    rgb_comb[ray * 3] = s0 + t0 + bgw; rgb_comb[ray * 3 + 1] = s1 + t1 + bgw; rgb_comb[ray * 3 + 2] = s2 + t2 + bgw;
    beta[ray] = be + beta_min;
  }
}

// d_dens_s += ..., d_rgb_s = ..., d_dens_t = ... (+ dens_t_const: the density regulariser's constant gradient),
// d_rgb_t = ..., d_unc = ...
template <int DC_MAXC>
__global__ void __launch_bounds__(256)
k_dual_composite_bwd(int nrays, int S, const float* __restrict__ dens_s, const float* __restrict__ dens_t,
                     const float* __restrict__ rgb_s, const float* __restrict__ rgb_t, const float* __restrict__ unc,
                     const float* __restrict__ tdist, const float* __restrict__ dirs, int opaque, float bg,
                     const float* __restrict__ d_rgb_comb, const float* __restrict__ d_beta, float dens_t_const,
                     float* __restrict__ d_dens_s, float* __restrict__ d_rgb_s, float* __restrict__ d_dens_t,
                     float* __restrict__ d_rgb_t, float* __restrict__ d_unc) {
  const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= nrays) return;
  const int C = (S + 63) >> 6;
  const float* td = tdist + (size_t)ray * (S + 1);
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  DualScan<DC_MAXC> R;
  dual_scan<DC_MAXC>(S, C, lane, dens_s + (size_t)ray * S, dens_t + (size_t)ray * S, td, dnorm, opaque, R);
  const float g0 = d_rgb_comb[ray * 3], g1 = d_rgb_comb[ray * 3 + 1], g2 = d_rgb_comb[ray * 3 + 2];
  const float gbeta = d_beta[ray];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < DC_MAXC; ++k)
    if (R.ok[k]) acc += (1.f - expf(-(R.a[k] + R.b[k]))) * R.T[k];
  acc = wave_sum_f(acc);
  const float one_m = 1.f - acc;
  const float gb = one_m > 0.f ? 1.f : (one_m == 0.f ? 0.5f : 0.f);   // jnp.maximum's tie rule
  const float gbg = gb * bg * (g0 + g1 + g2);
  float p[DC_MAXC], q[DC_MAXC], u[DC_MAXC], lane_v = 0.f, lane_u = 0.f;
#pragma unroll
  for (int k = 0; k < DC_MAXC; ++k) {
    p[k] = q[k] = u[k] = 0.f;
    if (!R.ok[k]) continue;
    const size_t i = (size_t)ray * S + lane * C + k;
    const float A1 = 1.f - expf(-R.a[k]), A2 = 1.f - expf(-R.b[k]), A = 1.f - expf(-(R.a[k] + R.b[k]));
    p[k] = g0 * rgb_s[i * 3] + g1 * rgb_s[i * 3 + 1] + g2 * rgb_s[i * 3 + 2];
    q[k] = g0 * rgb_t[i * 3] + g1 * rgb_t[i * 3 + 1] + g2 * rgb_t[i * 3 + 2];
    u[k] = unc[i];
    const float w1 = A1 * R.T[k], w2 = A2 * R.T[k], wt = A2 * R.Tt[k];
    d_rgb_s[i * 3] = w1 * g0; d_rgb_s[i * 3 + 1] = w1 * g1; d_rgb_s[i * 3 + 2] = w1 * g2;
    d_rgb_t[i * 3] = w2 * g0; d_rgb_t[i * 3 + 1] = w2 * g1; d_rgb_t[i * 3 + 2] = w2 * g2;
    d_unc[i] = gbeta * wt;
    lane_v += R.T[k] * (A1 * p[k] + A2 * q[k] - gbg * A);
    lane_u += wt * u[k];
  }
  const float sv = wave_incl_suffix_scan_f(lane_v, lane), su = wave_incl_suffix_scan_f(lane_u, lane);
  float sufv = __shfl_down(sv, 1), sufu = __shfl_down(su, 1);      // sums over the lanes after this one
  if (lane == 63) { sufv = 0.f; sufu = 0.f; }
#pragma unroll
  for (int k = DC_MAXC - 1; k >= 0; --k) {
    if (!R.ok[k]) continue;
    const size_t i = (size_t)ray * S + lane * C + k;
    const float ea = expf(-R.a[k]), eb = expf(-R.b[k]), eg = expf(-(R.a[k] + R.b[k]));
    float da = R.T[k] * (ea * p[k] - gbg * eg) - sufv;
    float db = R.T[k] * (eb * q[k] - gbg * eg) - sufv + gbeta * (R.Tt[k] * eb * u[k] - sufu);
    if (R.last[k] && opaque) { da = 0.f; db = 0.f; }              // replaced by +inf: no gradient
    if (R.last[k] && !opaque) {                                    // the last interval is in no prefix: only its own alpha
      da = R.T[k] * (ea * p[k] - gbg * eg);
      db = R.T[k] * (eb * q[k] - gbg * eg) + gbeta * R.Tt[k] * eb * u[k];
    }
    d_dens_s[i] += da * R.dl[k];
    d_dens_t[i] = db * R.dl[k] + dens_t_const;
    const float A1 = 1.f - ea, A2 = 1.f - eb, A = 1.f - eg;
    sufv += R.T[k] * (A1 * p[k] + A2 * q[k] - gbg * A);
    sufu += A2 * R.Tt[k] * u[k];
  }
}

// G[m,n] += (r1[m] c1[n] + r2[m] c2[n]) * (X[m,n] > 0)
template <int BF16>
__global__ void __launch_bounds__(256)
k_rank1_add2_mask(int M, int N, const float* __restrict__ r1, const float* __restrict__ c1, const float* __restrict__ r2,
                  const float* __restrict__ c2, const void* __restrict__ X, int ldxv, void* __restrict__ G, int ldg) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
  const float x = ldx<BF16>(X, (size_t)m * ldxv + n);
  if (x > 0.f) {
    const float add = r1[m] * c1[n] + r2[m] * c2[n];
    if (BF16) {
      uint16_t* g = (uint16_t*)G + (size_t)m * ldg + n;
      *g = f_to_op16(op16_to_f(*g, BF16) + add, BF16);
    } else {
      ((float*)G)[(size_t)m * ldg + n] += add;
    }
  }
}

// train_utils.py:150-183.  pred [L,N,3] (last level = rgb_combined).  stats: [2l] mean resid^2, [2l+1] data loss of
// level l (the last one divided by 2 beta^2), [2L] mean(log beta).
__global__ void __launch_bounds__(1024)
k_nerfw_loss(int N, int L, const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ beta,
             int charb, float pad, const float* __restrict__ coef, float beta_mult, float* __restrict__ d_pred,
             float* __restrict__ d_beta, float* __restrict__ stats) {
  __shared__ float red[16];
  const float inv = 1.f / (3.f * (float)N);
  for (int l = 0; l < L; ++l) {
    const float cf = coef[l];
    const bool fin = l == L - 1;
    float s_mse = 0.f, s_loss = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      const float b = fin ? beta[n] : 1.f;
      const float sc = fin ? 1.f / (2.f * b * b) : 1.f;
      float row = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t ix = ((size_t)l * N + n) * 3 + c;
        const float r = pred[ix] - gt[(size_t)n * 3 + c];
        const float r2 = r * r;
        float dl, ddl;
        if (charb) { dl = sqrtf(r2 + pad * pad); ddl = r / dl; } else { dl = r2; ddl = 2.f * r; }
        s_mse += r2;
        s_loss += dl * sc;
        row += dl;
        d_pred[ix] = cf * sc * ddl * inv;
      }
      if (fin) d_beta[n] = -cf * row / (b * b * b) * inv + beta_mult / (b * (float)N);
    }
    s_mse = block_sum1024(s_mse, red);
    s_loss = block_sum1024(s_loss, red);
    if (threadIdx.x == 0) { stats[2 * l] = s_mse * inv; stats[2 * l + 1] = s_loss * inv; }
  }
  float sl = 0.f;
  for (int n = threadIdx.x; n < N; n += blockDim.x) sl += logf(beta[n]);
  sl = block_sum1024(sl, red);
  if (threadIdx.x == 0) stats[2 * L] = sl / (float)N;
}

}  // namespace

extern "C" int hugs_dual_composite_fwd(int nrays, int S, const float* dens_s, const float* dens_t, const float* rgb_s,
                                       const float* rgb_t, const float* unc, const float* tdist, const float* dirs,
                                       int opaque_background, float bg, float beta_min, float* rgb_combined,
                                       float* rgb_static, float* rgb_transient, float* beta, void* stream) {
  HUGS_REQUIRE(S >= 1 && S <= 1024, -3, "hugs_dual_composite_fwd: %d samples per ray unsupported (<= 1024)", S);
  if (nrays <= 0) return 0;
#define HUGS_DCF(C_) k_dual_composite_fwd<C_><<<(nrays + 3) / 4, 256, 0, (hipStream_t)stream>>>(nrays, S, dens_s, dens_t, rgb_s, rgb_t, unc, \
    tdist, dirs, opaque_background, bg, beta_min, rgb_combined, rgb_static, rgb_transient, beta)
  if (S <= 256) HUGS_DCF(4); else if (S <= 512) HUGS_DCF(8); else HUGS_DCF(16);
#undef HUGS_DCF
  HUGS_CHECK_LAUNCH("k_dual_composite_fwd");
  return 0;
}

extern "C" int hugs_dual_composite_bwd(int nrays, int S, const float* dens_s, const float* dens_t, const float* rgb_s,
                                       const float* rgb_t, const float* unc, const float* tdist, const float* dirs,
                                       int opaque_background, float bg, const float* d_rgb_combined, const float* d_beta,
                                       float dens_t_const, float* d_dens_s_accum, float* d_rgb_s, float* d_dens_t,
                                       float* d_rgb_t, float* d_unc, void* stream) {
  HUGS_REQUIRE(S >= 1 && S <= 1024, -3, "hugs_dual_composite_bwd: %d samples per ray unsupported (<= 1024)", S);
  if (nrays <= 0) return 0;
#define HUGS_DCB(C_) k_dual_composite_bwd<C_><<<(nrays + 3) / 4, 256, 0, (hipStream_t)stream>>>( \
      nrays, S, dens_s, dens_t, rgb_s, rgb_t, unc, tdist, dirs, opaque_background, bg, d_rgb_combined, d_beta, dens_t_const, \
      d_dens_s_accum, d_rgb_s, d_dens_t, d_rgb_t, d_unc)
  if (S <= 256) HUGS_DCB(4); else if (S <= 512) HUGS_DCB(8); else HUGS_DCB(16);
#undef HUGS_DCB
  HUGS_CHECK_LAUNCH("k_dual_composite_bwd");
  return 0;
}

extern "C" int hugs_rank1_add2_mask(int dtype, int M, int N, const float* r1, const float* c1, const float* r2,
                                    const float* c2, const void* X, int ldx, void* G, int ldg, void* stream) {
  const size_t n = (size_t)M * N;
  if (n == 0) return 0;
  if (dtype == 2) k_rank1_add2_mask<2><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(M, N, r1, c1, r2, c2, X, ldx, G, ldg);
  else if (dtype) k_rank1_add2_mask<1><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(M, N, r1, c1, r2, c2, X, ldx, G, ldg);
  else k_rank1_add2_mask<0><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(M, N, r1, c1, r2, c2, X, ldx, G, ldg);
  HUGS_CHECK_LAUNCH("k_rank1_add2_mask");
  return 0;
}

extern "C" int hugs_nerfw_loss(int N, int L, const float* pred, const float* gt, const float* beta, int charb,
                               float charb_pad, const float* coef, float beta_mult, float* d_pred, float* d_beta,
                               float* out_stats, void* stream) {
  HUGS_REQUIRE(N > 0 && L >= 1 && L <= 8, -2, "hugs_nerfw_loss: N=%d L=%d", N, L);
  k_nerfw_loss<<<1, 1024, 0, (hipStream_t)stream>>>(N, L, pred, gt, beta, charb, charb_pad, coef, beta_mult, d_pred, d_beta,
                                                    out_stats);
  HUGS_CHECK_LAUNCH("k_nerfw_loss");
  return 0;
}

// Pixel / regulariser losses and their gradients (tiny, latency-bound kernels).
//
// Replaces (reference, MipNeRF360/internal): train_utils.py:72-111 compute_data_loss (incl. the HuGS
// static-mask variant and its [..,1]-denominator quirk), :114-147 compute_robustnerf_loss with
// :251-348 robustnerf_mask, :228-239 interlevel_loss -> stepfun.py:30-86 (searchsorted / inner_outer /
// lossfun_outer), :242-248 distortion_loss -> stepfun.py:266-276 lossfun_distortion.
#include "hugs_common.h"

__device__ __forceinline__ float block_sum(float v, float* red) {  // blockDim.x == 1024
  v = wave_sum_f(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];  // fixed order
  return t;
}

// One workgroup.  For each level l: resid = pred_l - gt;  lm[n] = per-pixel multiplier;
//   denom = max(cmult * sum lm, eps);  mse_l = sum lm*resid^2 / denom;  loss_l = sum lm*dl / denom
//   d_pred_l = coef_l * lm * dl' / denom.       out_stats[l*2+0] = mse_l, [l*2+1] = loss_l
// mode 0: lm = lossmult[n] (or 1 if disable_multiscale), cmult 3
// mode 1: static mask: lm = m + (1-m)*wt with m = (static_mask >= .5), cmult 1 (reference quirk)
// mode 2: lm = robust mask of level l (mask[l*N + n]), cmult 3
__global__ __launch_bounds__(1024) void k_data_loss(int N, int L, const float* __restrict__ pred /*[L,N,3]*/,
                                                    const float* __restrict__ gt, const float* __restrict__ lm_src,
                                                    int mode, float transient_weight, int charb, float charb_pad,
                                                    const float* __restrict__ coef /*[L]*/, float* __restrict__ d_pred,
                                                    float* __restrict__ out_stats) {
  __shared__ float red[16];
  // one workgroup per level (round 5: one workgroup walked the levels one after the other -- 89 us at 16 384 rays x 3 levels); the
  // sums of a level keep their order
  for (int l = blockIdx.x; l < L; l += gridDim.x) {
    float s_lm = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      float lm;
      if (mode == 0) lm = lm_src ? lm_src[n] : 1.f;
      else if (mode == 1) { const float m = lm_src[n] >= 0.5f ? 1.f : 0.f; lm = m + (1.f - m) * transient_weight; }
      else lm = lm_src[(size_t)l * N + n];
      s_lm += lm;
    }
    s_lm = block_sum(s_lm, red);
    const float denom = fmaxf((mode == 1 ? 1.f : 3.f) * s_lm, HUGS_EPS);
    const float cf = coef[l];
    float s_mse = 0.f, s_loss = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      float lm;
      if (mode == 0) lm = lm_src ? lm_src[n] : 1.f;
      else if (mode == 1) { const float m = lm_src[n] >= 0.5f ? 1.f : 0.f; lm = m + (1.f - m) * transient_weight; }
      else lm = lm_src[(size_t)l * N + n];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t ix = ((size_t)l * N + n) * 3 + c;
        const float r = pred[ix] - gt[(size_t)n * 3 + c];
        const float r2 = r * r;
        float dl, ddl;
        if (charb) { dl = sqrtf(r2 + charb_pad * charb_pad); ddl = r / dl; } else { dl = r2; ddl = 2.f * r; }
        s_mse += lm * r2;
        s_loss += lm * dl;
        d_pred[ix] = cf * lm * ddl / denom;
      }
    }
    s_mse = block_sum(s_mse, red);
    s_loss = block_sum(s_loss, red);
    if (threadIdx.x == 0) { out_stats[l * 2] = s_mse / denom; out_stats[l * 2 + 1] = s_loss / denom; }
  }
}

// RobustNeRF mask: one 256-thread workgroup per P x P patch (P == 16).  err = mean_c |pred - gt|.
// per-patch partial stats: [inlier, neighbours, patch, mask] counts -> stats_part[patch*4 + k]
__global__ __launch_bounds__(256) void k_robust_mask(int npatch, int P, const float* __restrict__ pred,
                                                     const float* __restrict__ gt, const float* __restrict__ thr_ptr, int fsize,
                                                     float smoothed_q, int inner, float inner_q,
                                                     float* __restrict__ mask, float* __restrict__ err_out,
                                                     float* __restrict__ stats_part, int squared) {
  __shared__ float s_inl[16][16];
  __shared__ float red[4];
  const int p = blockIdx.x, y = threadIdx.x / 16, x = threadIdx.x % 16;
  const size_t n = (size_t)p * P * P + y * P + x;
  float e = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) { const float d = pred[n * 3 + c] - gt[n * 3 + c]; e += squared ? d * d : fabsf(d); }
  e = e / 3.f;
  err_out[n] = e;
  const float inl = e < thr_ptr[0] ? 1.f : 0.f;
  s_inl[y][x] = inl;
  __syncthreads();
  // f x f box filter, zero padded SAME (lax.conv: f-1 pad rows in total, (f-1)/2 before, the rest after -- the
  // window is asymmetric for an even f), then binarise
  const int hlo = (fsize - 1) / 2, hhi = fsize - 1 - hlo;
  float nb = 0.f;
  for (int dy = -hlo; dy <= hhi; ++dy)
    for (int dx_ = -hlo; dx_ <= hhi; ++dx_) {
      const int yy = y + dy, xx = x + dx_;
      if (yy >= 0 && yy < P && xx >= 0 && xx < P) nb += s_inl[yy][xx];
    }
  nb = nb / (float)(fsize * fsize);
  const float has_nb = nb > 1.f - smoothed_q ? 1.f : 0.f;
  // patch vote
  float cnt = wave_sum_f(inl);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  const float pmean = (red[0] + red[1] + red[2] + red[3]) / (float)(P * P);
  const int lo = (P - inner) / 2;
  const float in_inner = (y >= lo && y < lo + inner && x >= lo && x < lo + inner) ? 1.f : 0.f;
  const float is_patch = (pmean > 1.f - inner_q ? 1.f : 0.f) * in_inner;
  const float m = (is_patch + has_nb + inl > 1e-3f) ? 1.f : 0.f;
  mask[n] = m;
  // stats
  float a = wave_sum_f(has_nb), b = wave_sum_f(is_patch), c = wave_sum_f(m);
  __shared__ float red2[4][3];
  if ((threadIdx.x & 63) == 0) { red2[threadIdx.x >> 6][0] = a; red2[threadIdx.x >> 6][1] = b; red2[threadIdx.x >> 6][2] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = stats_part + (size_t)p * 4;
    o[0] = red[0] + red[1] + red[2] + red[3];
    o[1] = red2[0][0] + red2[1][0] + red2[2][0] + red2[3][0];
    o[2] = red2[0][1] + red2[1][1] + red2[2][1] + red2[3][1];
    o[3] = red2[0][2] + red2[1][2] + red2[2][2] + red2[3][2];
  }
}

// single-workgroup bitonic sort of n <= 32768 floats in LDS + linear-interpolated quantile (jnp.quantile),
// and the mean of the four per-patch stat counts.  stats_out[0] = quantile, [1..4] = means.
__global__ __launch_bounds__(1024) void k_quantile_stats(int n, const float* __restrict__ vals, float q, int npatch,
                                                         const float* __restrict__ stats_part, float* __restrict__ stats_out) {
  extern __shared__ float sv[];
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = threadIdx.x; i < np2; i += blockDim.x) sv[i] = i < n ? vals[i] : __builtin_inff();
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float a = sv[i], b = sv[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { sv[i] = b; sv[ixj] = a; }
        }
      }
      __syncthreads();
    }
  if (threadIdx.x == 0) {
    const float pos = q * (float)(n - 1);
    const int lo = (int)floorf(pos);
    const int hi = min(lo + 1, n - 1);
    const float fr = pos - (float)lo;
    stats_out[0] = sv[lo] * (1.f - fr) + sv[hi] * fr;
  }
  if (threadIdx.x >= 64 && threadIdx.x < 68) {
    const int k = threadIdx.x - 64;
    float a = 0.f;
    for (int p = 0; p < npatch; ++p) a += stats_part[(size_t)p * 4 + k];
    stats_out[1 + k] = a / (float)n;
  }
}

// ---- interlevel loss, one wave per ray -----------------------------------------------------------
// t[S+1], w[S]: final level (constants);  te[Sp+1], we[Sp]: proposal level (gradient flows to we).
// loss_ray = sum_i max(0, w_i - wo_i)^2 / (w_i + eps);  d_we = scale * d loss_ray / d we
// MAXC = samples per lane of the blocked scan: 4 -> <= 256 samples per level, 8 -> 512, 16 -> 1024 (LDS 5 x 4 x (64 MAXC + 4) floats)
template <int MAXC>
__global__ __launch_bounds__(256) void k_interlevel(int nrays, int S, int Sp, const float* __restrict__ t,
                                                    const float* __restrict__ w, const float* __restrict__ te,
                                                    const float* __restrict__ we, float scale,
                                                    float* __restrict__ loss_ray, float* __restrict__ d_we) {
  constexpr int IL_CAP = 64 * MAXC + 4;
  __shared__ float s_te[4][IL_CAP], s_cy[4][IL_CAP], s_g[4][IL_CAP];
  __shared__ int s_a[4][IL_CAP], s_b[4][IL_CAP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ray = blockIdx.x * 4 + wv;
  const bool live = ray < nrays;
  if (live) {
    for (int i = lane; i <= Sp; i += 64) s_te[wv][i] = te[(size_t)ray * (Sp + 1) + i];
    // cy[0] = 0, cy[j+1] = cumsum(we)[j]: blocked scan, C elements per lane
    const int C = (Sp + 63) >> 6;
    float v[MAXC], tot = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) { const int i = lane * C + k; v[k] = (k < C && i < Sp) ? we[(size_t)ray * Sp + i] : 0.f; tot += v[k]; }
    const float incl = wave_incl_scan_f(tot, lane);
    float run = __shfl_up(incl, 1);
    if (lane == 0) { run = 0.f; s_cy[wv][0] = 0.f; }
    // The lane-blocked scan is monotone inside a lane but its lane-boundary values come from differently associated
    // sums, so the prefix could step DOWN by an ulp there; on a stretch of zero weights that would make an outer
    // measure of -1e-7 and, through 1/(w + eps), an O(1) spurious gradient.  A sequential cumsum of non-negative
    // numbers (the reference's) never decreases: enforce that with a running maximum across lanes.
    float cyv[MAXC], last = run;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) { last += v[k]; cyv[k] = last; }
    float pm = last;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const float o = __shfl_up(pm, dd); if (lane >= dd) pm = fmaxf(pm, o); }
    float carry = __shfl_up(pm, 1);
    if (lane == 0) carry = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) { const int i = lane * C + k; if (k < C && i < Sp) s_cy[wv][i + 1] = fmaxf(cyv[k], carry); }
  }
  __syncthreads();
  float lsum = 0.f;
  if (live) {
    for (int i = lane; i < S; i += 64) {
      const float t0 = t[(size_t)ray * (S + 1) + i], t1 = t[(size_t)ray * (S + 1) + i + 1];
      // idx_lo(t0) = last j with te[j] <= t0 (0 if none); idx_hi(t1) = first j with te[j] > t1 (Sp if none)
      int lo = 0, hi = Sp + 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_te[wv][mid] <= t0) lo = mid + 1; else hi = mid; }
      const int a = lo > 0 ? lo - 1 : 0;
      lo = 0; hi = Sp + 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_te[wv][mid] <= t1) lo = mid + 1; else hi = mid; }
      const int b = lo <= Sp ? lo : Sp;
      const float wo = s_cy[wv][b] - s_cy[wv][a];
      const float wi = w[(size_t)ray * S + i];
      const float ex = fmaxf(0.f, wi - wo);
      lsum += ex * ex / (wi + HUGS_EPS);
      s_g[wv][i] = -2.f * ex / (wi + HUGS_EPS) * scale;   // d(scale*loss)/d wo
      s_a[wv][i] = a; s_b[wv][i] = b;
    }
  }
  lsum = wave_sum_f(lsum);
  __syncthreads();
  if (live) {
    if (lane == 0) loss_ray[ray] = lsum;
    // d_we[j] = sum_{i : a_i <= j < b_i} g_i ; a_i, b_i are non-decreasing in i -> contiguous i range
    for (int j = lane; j < Sp; j += 64) {
      int lo = 0, hi = S;   // first i with b_i > j
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_b[wv][mid] > j) hi = mid; else lo = mid + 1; }
      const int i0 = lo;
      lo = 0; hi = S;       // first i with a_i > j
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_a[wv][mid] > j) hi = mid; else lo = mid + 1; }
      const int i1 = lo;
      float a = 0.f;
      for (int i = i0; i < i1; ++i) a += s_g[wv][i];
      d_we[(size_t)ray * Sp + j] = a;
    }
  }
}

// ---- distortion loss, one wave per ray (O(S^2) in LDS) -------------------------------------------
template <int MAXC>
__global__ __launch_bounds__(256) void k_distortion(int nrays, int S, const float* __restrict__ t,
                                                    const float* __restrict__ w, float scale,
                                                    float* __restrict__ loss_ray, float* __restrict__ d_w) {
  constexpr int IL_CAP = 64 * MAXC + 4;
  __shared__ float s_u[4][IL_CAP], s_w[4][IL_CAP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ray = blockIdx.x * 4 + wv;
  const bool live = ray < nrays;
  if (live)
    for (int i = lane; i < S; i += 64) {
      s_u[wv][i] = (t[(size_t)ray * (S + 1) + i + 1] + t[(size_t)ray * (S + 1) + i]) / 2.f;
      s_w[wv][i] = w[(size_t)ray * S + i];
    }
  __syncthreads();
  float lsum = 0.f;
  if (live)
    for (int i = lane; i < S; i += 64) {
      const float ui = s_u[wv][i], wi = s_w[wv][i];
      float inner = 0.f;
      for (int j = 0; j < S; ++j) inner += s_w[wv][j] * fabsf(ui - s_u[wv][j]);
      const float dt = t[(size_t)ray * (S + 1) + i + 1] - t[(size_t)ray * (S + 1) + i];
      lsum += wi * inner + wi * wi * dt / 3.f;
      if (d_w) d_w[(size_t)ray * S + i] = scale * (2.f * inner + 2.f * wi * dt / 3.f);
    }
  lsum = wave_sum_f(lsum);
  if (live && lane == 0) loss_ray[ray] = lsum;
}

// out[0] = scale * sum(x[0..n))   single workgroup, fixed order
__global__ __launch_bounds__(1024) void k_sum(int n, const float* __restrict__ x, float scale, float* __restrict__ out) {
  __shared__ float red[16];
  float a = 0.f;
  int i = threadIdx.x;
  if (n > 8192) {      // (NeRF-W's mean transient density: 131 072 samples took 37 us as one dependent chain of loads per thread)
    float b = 0.f, c = 0.f, d = 0.f;
    const int B = blockDim.x;
    for (; i + 3 * B < n; i += 4 * B) { a += x[i]; b += x[i + B]; c += x[i + 2 * B]; d += x[i + 3 * B]; }
    a = (a + b) + (c + d);
  }
  for (; i < n; i += blockDim.x) a += x[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = a * scale;
}

// dst[i] += src[i]
__global__ void k_axpy1(size_t n, const float* __restrict__ src, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

extern "C" int hugs_data_loss(int N, int L, const float* pred, const float* gt, const float* lm_src, int mode,
                              float transient_weight, int charb, float charb_pad, const float* coef, float* d_pred,
                              float* out_stats, void* stream) {
  HUGS_REQUIRE(mode >= 0 && mode <= 2, -3, "hugs_data_loss: bad mode %d", mode);
  HUGS_REQUIRE(mode == 0 || lm_src, -3, "hugs_data_loss: mode %d needs a mask", mode);
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_data_loss, dim3(L), dim3(1024), 0, (hipStream_t)stream, N, L, pred, gt, lm_src, mode,
                     transient_weight, charb, charb_pad, coef, d_pred, out_stats);
  HUGS_CHECK_LAUNCH("hugs_data_loss");
  return 0;
}

// pred/gt: [npatch*P*P, 3]. Outputs mask[n], err[n] (workspace), stats[5] = {next threshold (quantile of err),
// mean is_inlier_loss, mean has_inlier_neighbors, mean is_inlier_patch, mean mask}; stats_part ws [npatch*4].
static int robust_mask_impl(int squared, int npatch, int P, const float* pred, const float* gt, const float* inlier_threshold,
                            float quantile, int filter_size, float smoothed_q, int inner_patch, float inner_q,
                            float* mask, float* err_ws, float* stats_part_ws, float* stats, void* stream) {
  HUGS_REQUIRE(P == 16, -5, "hugs_robust_mask: patch_size must be 16 (got %d)", P);
  const int n = npatch * P * P;
  HUGS_REQUIRE(n <= 32768, -3, "hugs_robust_mask: %d rays per device > 32768", n);
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_robust_mask, dim3(npatch), dim3(256), 0, st, npatch, P, pred, gt, inlier_threshold, filter_size,
                     smoothed_q, inner_patch, inner_q, mask, err_ws, stats_part_ws, squared);
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  hipLaunchKernelGGL(k_quantile_stats, dim3(1), dim3(1024), np2 * sizeof(float), st, n, err_ws, quantile, npatch,
                     stats_part_ws, stats);
  HUGS_CHECK_LAUNCH("hugs_robust_mask");
  return 0;
}
extern "C" int hugs_robust_mask(int npatch, int P, const float* pred, const float* gt, const float* inlier_threshold /*device, [1]*/,
                                float quantile, int filter_size, float smoothed_q, int inner_patch, float inner_q,
                                float* mask, float* err_ws, float* stats_part_ws, float* stats, void* stream) {
  return robust_mask_impl(0, npatch, P, pred, gt, inlier_threshold, quantile, filter_size, smoothed_q, inner_patch, inner_q, mask,
                          err_ws, stats_part_ws, stats, stream);
}
// nerfacto's get_robustnerf_mask (nerfacto/utils/loss_utils.py:88-150) is the same mask on SQUARED residuals
// (models/nerfacto.py:505: errors = resid_sq.detach(); Mip-NeRF 360 passes |resid|).
extern "C" int hugs_nf_robust_mask(int npatch, int P, const float* pred, const float* gt, const float* inlier_threshold,
                                   float quantile, int filter_size, float smoothed_q, int inner_patch, float inner_q,
                                   float* mask, float* err_ws, float* stats_part_ws, float* stats, void* stream) {
  return robust_mask_impl(1, npatch, P, pred, gt, inlier_threshold, quantile, filter_size, smoothed_q, inner_patch, inner_q, mask,
                          err_ws, stats_part_ws, stats, stream);
}

extern "C" int hugs_interlevel(int nrays, int S, int Sp, const float* t, const float* w, const float* t_env,
                               const float* w_env, float scale, float* loss_ray, float* d_w_env, void* stream) {
  HUGS_REQUIRE(S >= 1 && Sp >= 1 && S <= 1024 && Sp <= 1024, -3, "hugs_interlevel: S=%d Sp=%d exceed capacity (1024 samples per level)", S, Sp);
  if (nrays <= 0) return 0;
  const int big = S > Sp ? S : Sp;
#define HUGS_IL_LAUNCH(C_) hipLaunchKernelGGL(k_interlevel<C_>, dim3((nrays + 3) / 4), dim3(256), 0, (hipStream_t)stream, nrays, S, Sp, t, w, \
    t_env, w_env, scale, loss_ray, d_w_env)
  if (big <= 256) HUGS_IL_LAUNCH(4); else if (big <= 512) HUGS_IL_LAUNCH(8); else HUGS_IL_LAUNCH(16);
#undef HUGS_IL_LAUNCH
  HUGS_CHECK_LAUNCH("hugs_interlevel");
  return 0;
}

extern "C" int hugs_distortion(int nrays, int S, const float* t, const float* w, float scale, float* loss_ray, float* d_w,
                               void* stream) {
  HUGS_REQUIRE(S >= 1 && S <= 1024, -3, "hugs_distortion: S=%d exceeds capacity (1024 samples per level)", S);
  if (nrays <= 0) return 0;
#define HUGS_DL_LAUNCH(C_) hipLaunchKernelGGL(k_distortion<C_>, dim3((nrays + 3) / 4), dim3(256), 0, (hipStream_t)stream, nrays, S, t, w, scale, loss_ray, d_w)
  if (S <= 256) HUGS_DL_LAUNCH(4); else if (S <= 512) HUGS_DL_LAUNCH(8); else HUGS_DL_LAUNCH(16);
#undef HUGS_DL_LAUNCH
  HUGS_CHECK_LAUNCH("hugs_distortion");
  return 0;
}

extern "C" int hugs_sum(int n, const float* x, float scale, float* out, void* stream) {
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, x, scale, out);
  HUGS_CHECK_LAUNCH("hugs_sum");
  return 0;
}

// dst[i] += alpha * src[i]
__global__ void k_axpy(size_t n, float alpha, const float* __restrict__ src, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += alpha * src[i];
}

extern "C" int hugs_axpy(long long n, float alpha, const float* src, float* dst, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_axpy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (size_t)n, alpha, src, dst);
  HUGS_CHECK_LAUNCH("hugs_axpy");
  return 0;
}

extern "C" int hugs_add_inplace(long long n, const float* src, float* dst, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_axpy1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (size_t)n, src, dst);
  HUGS_CHECK_LAUNCH("hugs_add_inplace");
  return 0;
}

// Per-ray and per-sample kernels of the NERFACTO path (reference: /root/reference/nerfacto).
//   hugs_nf_sample        utils/ray_utils.py:112-231 sample / sample_intervals (+ s_to_t of models/nerfacto.py:243-248)
//   hugs_nf_positions     models/nerfacto.py:326-328 sample positions, :822-829 / :975-982 normalisation + selector,
//                         models/custom_functions.py:17-24 contraction
//   hugs_nf_weights_fwd/bwd  utils/ray_utils.py:234-257 density_to_weight (deltas from the FIRST bin edge -- the
//                         reference's own arithmetic), :300-314 render_features, :340-347 render_depth
//   hugs_nf_interlevel    utils/loss_utils.py:7-62 (searchsorted-right `outer`, EPS = 1e-7) with the gradient to w_env
//   hugs_nf_density_act / hugs_nf_base_grad / hugs_nf_head_input / hugs_nf_app_bwd / hugs_nf_rgb_act / hugs_nf_rgb_grad
//                         the element-wise glue between the padded GEMMs of the fields (models/nerfacto.py:818-876,
//                         :971-988; custom_functions.py:38-52 trunc_exp)
//   hugs_nf_adam          torch.optim.Adam step (train.py:183) on the flat parameter buffer
// One wavefront owns one ray in the per-ray kernels (up to 1024 bins: 16 per lane, staged in LDS).
#include "hugs_common.h"

#define NF_CAP 1025

// (`bf16` is the dtype code of the C ABI: 0 = fp32, 1 = bf16, 2 = IEEE half)
__device__ __forceinline__ float nf_load(const void* p, size_t i, int bf16) {
  return bf16 ? op16_to_f(((const uint16_t*)p)[i], bf16) : ((const float*)p)[i];
}
__device__ __forceinline__ void nf_store(void* p, size_t i, int bf16, float v) {
  if (bf16) ((uint16_t*)p)[i] = f_to_op16(v, bf16); else ((float*)p)[i] = v;
}

// spacing functions of models/nerfacto.py:231-241: 0 uniform, 1 piecewise, 2 reciprocal
__device__ __forceinline__ float nf_spacing(float x, int mode, bool inverse) {
  if (mode == 1) return inverse ? (x < 0.5f ? 2.f * x : 1.f / (2.f - 2.f * x)) : (x < 1.f ? x / 2.f : 1.f - 1.f / (2.f * x));
  if (mode == 2) return 1.f / x;
  return x;
}

// ------------------------------------------------------------------------------------------------
// sampler
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_nf_sample(int nrays, int nb, int ns, const float* __restrict__ bins,
                                                  const float* __restrict__ weights, float anneal, float padding,
                                                  const float* __restrict__ u_base, const float* __restrict__ jitter,
                                                  int jitter_stride, float lo, float hi, int spacing,
                                                  const float* __restrict__ near, const float* __restrict__ far,
                                                  float* __restrict__ sbins, float* __restrict__ ebins) {
  __shared__ float s_bin[NF_CAP], s_cdf[NF_CAP], s_cen[NF_CAP];
  const int ray = blockIdx.x, lane = threadIdx.x;
  const float* b = bins + (size_t)ray * (nb + 1);
  const float* w = weights + (size_t)ray * nb;
  for (int i = lane; i <= nb; i += 64) s_bin[i] = b[i];
  __syncthreads();
  // logits (ray_utils.py:146-150), softmax
  float mx = -__builtin_inff();
  for (int i = lane; i < nb; i += 64) {
    const float lg = s_bin[i + 1] > s_bin[i] ? anneal * logf(w[i] + padding) : -__builtin_inff();
    s_cen[i] = lg;
    mx = fmaxf(mx, lg);
  }
  mx = wave_max_f(mx);
  const bool dead = !(mx > -__builtin_inff());          // every logit -inf: the reference sets them all to 1
  float sum = 0.f;
  for (int i = lane; i < nb; i += 64) {
    const float e = dead ? 1.f : expf(s_cen[i] - mx);
    s_cen[i] = e;
    sum += e;
  }
  sum = wave_sum_f(sum);
  __syncthreads();
  // cdf = [0, cumsum(pdf[:-1]).clamp_max(1), 1]: lane owns a contiguous chunk, wave scan of the chunk totals
  const int per = (nb + 63) / 64;
  float loc = 0.f;
  for (int k = 0; k < per; ++k) { const int i = lane * per + k; if (i < nb) loc += s_cen[i] / sum; }
  const float incl = wave_incl_scan_f(loc, lane);
  float run = incl - loc;
  for (int k = 0; k < per; ++k) {
    const int i = lane * per + k;
    if (i < nb) { run += s_cen[i] / sum; if (i + 1 < nb) s_cdf[i + 1] = fminf(run, 1.f); }
  }
  if (lane == 0) { s_cdf[0] = 0.f; s_cdf[nb] = 1.f; }
  __syncthreads();
  // inverse cdf: inds = searchsorted(cdf, u, right); below / above clamped (ray_utils.py:188-199)
  for (int j = lane; j < ns; j += 64) {
    const float u = u_base[j] + (jitter ? jitter[(size_t)ray * jitter_stride + (jitter_stride > 1 ? j : 0)] : 0.f);
    int l = 0, r = nb + 1;                              // first index with cdf > u
    while (l < r) { const int m = (l + r) >> 1; if (s_cdf[m] > u) r = m; else l = m + 1; }
    const int below = min(max(l - 1, 0), nb), above = min(max(l, 0), nb);
    const float c0 = s_cdf[below], c1 = s_cdf[above], b0 = s_bin[below], b1 = s_bin[above];
    float t = (u - c0) / (c1 - c0);
    if (t != t) t = 0.f;
    t = fminf(fmaxf(t, 0.f), 1.f);
    s_cen[j] = b0 + t * (b1 - b0);
  }
  __syncthreads();
  const float nr = near[ray], fr = far[ray];
  const float s_near = nf_spacing(nr, spacing, false), s_far = nf_spacing(fr, spacing, false);
  for (int j = lane; j <= ns; j += 64) {
    float s;
    if (j == 0) s = fmaxf(2.f * s_cen[0] - (s_cen[1] + s_cen[0]) / 2.f, lo);
    else if (j == ns) s = fminf(2.f * s_cen[ns - 1] - (s_cen[ns - 1] + s_cen[ns - 2]) / 2.f, hi);
    else s = (s_cen[j] + s_cen[j - 1]) / 2.f;
    sbins[(size_t)ray * (ns + 1) + j] = s;
    ebins[(size_t)ray * (ns + 1) + j] = nf_spacing(s * s_far + (1.f - s) * s_near, spacing, true);
  }
}

extern "C" int hugs_nf_sample(int nrays, int nb, int ns, const float* bins, const float* weights, float anneal, float padding,
                              const float* u_base, const float* jitter, int jitter_stride, float lo, float hi, int spacing,
                              const float* near, const float* far, float* sbins, float* ebins, void* stream) {
  if (ns <= 1) { hugs_set_error("num_samples must be > 1, is %d.", ns); return -2; }
  HUGS_REQUIRE(nb >= 1 && nb < NF_CAP && ns < NF_CAP, -3, "hugs_nf_sample: %d bins / %d samples exceed the capacity %d", nb, ns, NF_CAP - 1);
  HUGS_REQUIRE(spacing >= 0 && spacing <= 2, -2, "hugs_nf_sample: spacing must be 0 uniform, 1 piecewise or 2 reciprocal");
  if (nrays <= 0) return 0;
  hipLaunchKernelGGL(k_nf_sample, dim3(nrays), dim3(64), 0, (hipStream_t)stream, nrays, nb, ns, bins, weights, anneal, padding,
                     u_base, jitter, jitter_stride, lo, hi, spacing, near, far, sbins, ebins);
  HUGS_CHECK_LAUNCH("hugs_nf_sample");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// positions
// ------------------------------------------------------------------------------------------------
__global__ void k_nf_positions(long long M, int S, const float* __restrict__ ebins, const float* __restrict__ origins,
                               const float* __restrict__ dirs, int contract, float bound, float* __restrict__ x01,
                               float* __restrict__ sel) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int ray = (int)(m / S), s = (int)(m % S);
  const float t = (ebins[(size_t)ray * (S + 1) + s + 1] + ebins[(size_t)ray * (S + 1) + s]) / 2.f;
  float p[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) p[c] = origins[ray * 3 + c] + dirs[ray * 3 + c] * t;
  if (contract) {   // custom_functions.py:17-24, then (x + 2) / 4
    const float m2 = fmaxf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2], HUGS_EPS);
    if (!(m2 <= 1.f)) { const float k = (2.f * sqrtf(m2) - 1.f) / m2; p[0] *= k; p[1] *= k; p[2] *= k; }
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = (p[c] + 2.f) / 4.f;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = (p[c] + bound) / (2.f * bound);
  }
  const bool in = p[0] >= 0.f && p[0] <= 1.f && p[1] >= 0.f && p[1] <= 1.f && p[2] >= 0.f && p[2] <= 1.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) x01[m * 3 + c] = in ? p[c] : 0.f;
  sel[m] = in ? 1.f : 0.f;
}

extern "C" int hugs_nf_positions(int nrays, int S, const float* ebins, const float* origins, const float* dirs, int contract,
                                 float bound, float* x01, float* sel, void* stream) {
  const long long M = (long long)nrays * S;
  if (M <= 0) return 0;
  hipLaunchKernelGGL(k_nf_positions, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, S, ebins, origins,
                     dirs, contract, bound, x01, sel);
  HUGS_CHECK_LAUNCH("hugs_nf_positions");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// density -> weights, rendering (one wave per ray, samples chunked contiguously per lane)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_nf_weights_fwd(int nrays, int S, const float* __restrict__ density,
                                                       const float* __restrict__ ebins, const float* __restrict__ dirs,
                                                       int opaque, const float* __restrict__ rgb_s, const float* __restrict__ bg,
                                                       float* __restrict__ weights, float* __restrict__ rgb_out,
                                                       float* __restrict__ acc_out, float* __restrict__ depth_out) {
  const int ray = blockIdx.x, lane = threadIdx.x;
  const float* eb = ebins + (size_t)ray * (S + 1);
  const float dn = sqrtf(dirs[ray * 3] * dirs[ray * 3] + dirs[ray * 3 + 1] * dirs[ray * 3 + 1] + dirs[ray * 3 + 2] * dirs[ray * 3 + 2]);
  const float e0 = eb[0];
  const int per = (S + 63) / 64;
  float loc = 0.f;
  for (int k = 0; k < per; ++k) {
    const int i = lane * per + k;
    if (i < S) { const float dd = (opaque && i == S - 1) ? __builtin_inff() : density[(size_t)ray * S + i] * ((eb[i + 1] - e0) * dn); if (i < S - 1) loc += dd; }
  }
  float run = wave_incl_scan_f(loc, lane) - loc;        // sum of dd over samples before this lane's chunk
  float acc = 0.f, dep = 0.f, r = 0.f, g = 0.f, bl = 0.f, smax = 0.f;
  for (int k = 0; k < per; ++k) {
    const int i = lane * per + k;
    if (i < S) {
      const float dd = (opaque && i == S - 1) ? __builtin_inff() : density[(size_t)ray * S + i] * ((eb[i + 1] - e0) * dn);
      float wv = (1.f - expf(-dd)) * expf(-run);
      if (wv != wv) wv = 0.f;                            // nan_to_num (inf * 0)
      weights[(size_t)ray * S + i] = wv;
      run += dd;
      acc += wv;
      const float step = (eb[i + 1] + eb[i]) / 2.f;
      dep += wv * step;
      smax = fmaxf(smax, step);
      if (rgb_s) { const float* c = rgb_s + ((size_t)ray * S + i) * 3; r += wv * c[0]; g += wv * c[1]; bl += wv * c[2]; }
    }
  }
  acc = wave_sum_f(acc); dep = wave_sum_f(dep); smax = wave_max_f(smax);
  if (rgb_s) { r = wave_sum_f(r); g = wave_sum_f(g); bl = wave_sum_f(bl); }
  if (lane == 0) {
    if (acc_out) acc_out[ray] = acc;
    if (depth_out) depth_out[ray] = dep / (acc > 0.f ? acc : HUGS_EPS);   // (the reference clips to the BATCH max step: see host)
    if (rgb_s && rgb_out) {
      const float ba = fmaxf(1.f - acc, 0.f);
      rgb_out[ray * 3] = r + (bg ? bg[ray * 3] * ba : 0.f);
      rgb_out[ray * 3 + 1] = g + (bg ? bg[ray * 3 + 1] * ba : 0.f);
      rgb_out[ray * 3 + 2] = bl + (bg ? bg[ray * 3 + 2] * ba : 0.f);
    }
  }
}

// dL/d density and dL/d rgb_s from dL/d rgb_out [N,3] and an extra dL/d weights [N,S] (losses on the histogram).
// w_i = (1 - e^{-dd_i}) T_i, T_i = e^{-sum_{j<i} dd_j}:  dL/d dd_i = g_i T_i e^{-dd_i} - sum_{k>i} g_k w_k.
__global__ __launch_bounds__(64) void k_nf_weights_bwd(int nrays, int S, const float* __restrict__ density,
                                                       const float* __restrict__ ebins, const float* __restrict__ dirs,
                                                       int opaque, const float* __restrict__ rgb_s, const float* __restrict__ bg,
                                                       const float* __restrict__ weights, const float* __restrict__ d_rgb_out,
                                                       const float* __restrict__ d_w_extra, float* __restrict__ d_density,
                                                       float* __restrict__ d_rgb_s) {
  const int ray = blockIdx.x, lane = threadIdx.x;
  const float* eb = ebins + (size_t)ray * (S + 1);
  const float dn = sqrtf(dirs[ray * 3] * dirs[ray * 3] + dirs[ray * 3 + 1] * dirs[ray * 3 + 1] + dirs[ray * 3 + 2] * dirs[ray * 3 + 2]);
  const float e0 = eb[0];
  const int per = (S + 63) / 64;
  float dr[3] = {0.f, 0.f, 0.f};
  if (d_rgb_out) { dr[0] = d_rgb_out[ray * 3]; dr[1] = d_rgb_out[ray * 3 + 1]; dr[2] = d_rgb_out[ray * 3 + 2]; }
  // background term: rgb_out += bg * max(1 - acc, 0)  ->  g_i -= <bg, dr> while acc < 1
  float accp = 0.f;
  for (int k = 0; k < per; ++k) { const int i = lane * per + k; if (i < S) accp += weights[(size_t)ray * S + i]; }
  const float acc = wave_sum_f(accp);
  const float gbg = (bg && d_rgb_out && acc < 1.f) ? -(bg[ray * 3] * dr[0] + bg[ray * 3 + 1] * dr[1] + bg[ray * 3 + 2] * dr[2]) : 0.f;
  // pass 1: g_i, suffix sums of g_k w_k, prefix sums of dd
  float loc_gw = 0.f, loc_dd = 0.f;
  for (int k = 0; k < per; ++k) {
    const int i = lane * per + k;
    if (i < S) {
      const size_t m = (size_t)ray * S + i;
      float gi = gbg + (d_w_extra ? d_w_extra[m] : 0.f);
      if (rgb_s && d_rgb_out) gi += dr[0] * rgb_s[m * 3] + dr[1] * rgb_s[m * 3 + 1] + dr[2] * rgb_s[m * 3 + 2];
      loc_gw += gi * weights[m];
      if (i < S - 1) loc_dd += (opaque && i == S - 1) ? 0.f : density[m] * ((eb[i + 1] - e0) * dn);
    }
  }
  float suf = wave_incl_suffix_scan_f(loc_gw, lane) - loc_gw;      // sum over later lanes' chunks
  float run = wave_incl_scan_f(loc_dd, lane) - loc_dd;
  // walk the chunk backwards for the suffix part, forwards for T: two small loops
  float Tpre[16], ddv[16], giv[16];
  for (int k = 0; k < per; ++k) {
    const int i = lane * per + k;
    if (i < S) {
      const size_t m = (size_t)ray * S + i;
      const float delta = (eb[i + 1] - e0) * dn;
      ddv[k] = (opaque && i == S - 1) ? __builtin_inff() : density[m] * delta;
      Tpre[k] = expf(-run);
      run += ddv[k];
      float gi = gbg + (d_w_extra ? d_w_extra[m] : 0.f);
      if (rgb_s && d_rgb_out) gi += dr[0] * rgb_s[m * 3] + dr[1] * rgb_s[m * 3 + 1] + dr[2] * rgb_s[m * 3 + 2];
      giv[k] = gi;
      if (d_rgb_s) { const float wv = weights[m]; d_rgb_s[m * 3] = wv * dr[0]; d_rgb_s[m * 3 + 1] = wv * dr[1]; d_rgb_s[m * 3 + 2] = wv * dr[2]; }
    }
  }
  for (int k = per - 1; k >= 0; --k) {
    const int i = lane * per + k;
    if (i < S) {
      const size_t m = (size_t)ray * S + i;
      const float delta = (eb[i + 1] - e0) * dn;
      const bool last_opaque = opaque && i == S - 1;
      // d w_i / d dd_i = T_i e^{-dd_i} (0 for the infinitely wide last interval)
      const float own = last_opaque ? 0.f : giv[k] * Tpre[k] * expf(-ddv[k]);
      float dd_grad = own - suf;
      if (dd_grad != dd_grad) dd_grad = 0.f;
      d_density[m] = last_opaque ? 0.f : dd_grad * delta;
      suf += giv[k] * weights[m];
    }
  }
}

extern "C" int hugs_nf_weights_fwd(int nrays, int S, const float* density, const float* ebins, const float* dirs, int opaque,
                                   const float* rgb_s, const float* bg, float* weights, float* rgb_out, float* acc, float* depth,
                                   void* stream) {
  HUGS_REQUIRE(S >= 1 && S < NF_CAP, -3, "hugs_nf_weights_fwd: %d samples per ray unsupported", S);
  if (nrays <= 0) return 0;
  hipLaunchKernelGGL(k_nf_weights_fwd, dim3(nrays), dim3(64), 0, (hipStream_t)stream, nrays, S, density, ebins, dirs, opaque, rgb_s,
                     bg, weights, rgb_out, acc, depth);
  HUGS_CHECK_LAUNCH("hugs_nf_weights_fwd");
  return 0;
}

extern "C" int hugs_nf_weights_bwd(int nrays, int S, const float* density, const float* ebins, const float* dirs, int opaque,
                                   const float* rgb_s, const float* bg, const float* weights, const float* d_rgb_out,
                                   const float* d_w_extra, float* d_density, float* d_rgb_s, void* stream) {
  HUGS_REQUIRE(S >= 1 && S <= 1024, -3, "hugs_nf_weights_bwd: %d samples per ray unsupported (<= 1024)", S);
  if (nrays <= 0) return 0;
  hipLaunchKernelGGL(k_nf_weights_bwd, dim3(nrays), dim3(64), 0, (hipStream_t)stream, nrays, S, density, ebins, dirs, opaque, rgb_s,
                     bg, weights, d_rgb_out, d_w_extra, d_density, d_rgb_s);
  HUGS_CHECK_LAUNCH("hugs_nf_weights_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// interlevel loss (loss_utils.py:7-62): per ray sum_i max(w_i - w_outer_i, 0)^2 / (w_i + 1e-7), and its gradient to w_env
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_nf_interlevel(int nrays, int S, int Sp, const float* __restrict__ t,
                                                      const float* __restrict__ w, const float* __restrict__ t_env,
                                                      const float* __restrict__ w_env, float scale, float* __restrict__ loss_ray,
                                                      float* __restrict__ d_w_env) {
  __shared__ float s_te[NF_CAP], s_cy[NF_CAP], s_diff[NF_CAP];
  const int ray = blockIdx.x, lane = threadIdx.x;
  const float* te = t_env + (size_t)ray * (Sp + 1);
  const float* we = w_env + (size_t)ray * Sp;
  for (int i = lane; i <= Sp; i += 64) { s_te[i] = te[i]; s_diff[i] = 0.f; }
  // cy1 = [0, cumsum(w_env)]
  const int per = (Sp + 63) / 64;
  float loc = 0.f;
  for (int k = 0; k < per; ++k) { const int i = lane * per + k; if (i < Sp) loc += we[i]; }
  float run = wave_incl_scan_f(loc, lane) - loc;
  for (int k = 0; k < per; ++k) { const int i = lane * per + k; if (i < Sp) { run += we[i]; s_cy[i + 1] = run; } }
  if (lane == 0) s_cy[0] = 0.f;
  __syncthreads();
  float ls = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float t0 = t[(size_t)ray * (S + 1) + i], t1 = t[(size_t)ray * (S + 1) + i + 1], wi = w[(size_t)ray * S + i];
    // idx_lo = searchsorted(t_env[:-1], t0, right) - 1 ; idx_hi = searchsorted(t_env[1:], t1, right) ; both clamped to [0, Sp-1]
    int l = 0, r = Sp;
    while (l < r) { const int m = (l + r) >> 1; if (s_te[m] > t0) r = m; else l = m + 1; }
    const int lo = min(max(l - 1, 0), Sp - 1);
    l = 0; r = Sp;
    while (l < r) { const int m = (l + r) >> 1; if (s_te[m + 1] > t1) r = m; else l = m + 1; }
    const int hi = min(max(l, 0), Sp - 1);
    const float w_outer = s_cy[hi + 1] - s_cy[lo];
    const float d = fmaxf(wi - w_outer, 0.f);
    ls += d * d / (wi + 1.0e-7f);
    if (d > 0.f && d_w_env) {
      const float c = -2.f * d / (wi + 1.0e-7f) * scale;     // dL/d w_outer, spread over env bins lo..hi
      if (hi >= lo) { atomicAdd(&s_diff[lo], c); atomicAdd(&s_diff[hi + 1], -c); }
      // hi < lo: w_outer = cy[hi+1] - cy[lo] = -(sum of w_env[hi+1 .. lo-1]): the same difference form with the sign flipped
      else { atomicAdd(&s_diff[hi + 1], -c); atomicAdd(&s_diff[lo], c); }
    }
  }
  ls = wave_sum_f(ls);
  if (lane == 0 && loss_ray) loss_ray[ray] = ls;
  __syncthreads();
  if (d_w_env) {
    float l2 = 0.f;
    for (int k = 0; k < per; ++k) { const int i = lane * per + k; if (i < Sp) l2 += s_diff[i]; }
    float r2 = wave_incl_scan_f(l2, lane) - l2;
    for (int k = 0; k < per; ++k) { const int i = lane * per + k; if (i < Sp) { r2 += s_diff[i]; d_w_env[(size_t)ray * Sp + i] = r2; } }
  }
}

extern "C" int hugs_nf_interlevel(int nrays, int S, int Sp, const float* t, const float* w, const float* t_env, const float* w_env,
                                  float scale, float* loss_ray, float* d_w_env, void* stream) {
  HUGS_REQUIRE(S >= 1 && Sp >= 1 && S < NF_CAP && Sp < NF_CAP, -3, "hugs_nf_interlevel: S=%d Sp=%d exceed the capacity (%d per level)", S, Sp, NF_CAP - 1);      // (s_te / s_cy / s_diff hold Sp + 1 entries)
  if (nrays <= 0) return 0;
  hipLaunchKernelGGL(k_nf_interlevel, dim3(nrays), dim3(64), 0, (hipStream_t)stream, nrays, S, Sp, t, w, t_env, w_env, scale, loss_ray, d_w_env);
  HUGS_CHECK_LAUNCH("hugs_nf_interlevel");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// element-wise glue around the fields' GEMMs.  Matrices are row-major [M, ld] in the compute dtype (bf16 or fp32).
// ------------------------------------------------------------------------------------------------
// density = trunc_exp(Y[:, col]) * selector  (nerfacto.py:833-836 / 984-987)
__global__ void k_nf_density_act(long long M, int bf16, const void* __restrict__ Y, int ldy, int col, const float* __restrict__ sel,
                                 float* __restrict__ density, int act, float dbias) {
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  density[m] = nf_density_value(nf_load(Y, (size_t)m * ldy + col, bf16), act, dbias) * sel[m];
}

// eight consecutive columns per thread (one 16-byte store in bf16): the per-element form of these glue kernels spent its time
// in 64-bit index divisions and 2-byte stores (0.6 TB/s); ld must be a multiple of 8
__device__ __forceinline__ void nf_store8(void* p, size_t i, int bf16, const float (&v)[8]) {
  if (bf16) {
    uint4 u;
    u.x = f2_to_op16(v[0], v[1], bf16); u.y = f2_to_op16(v[2], v[3], bf16);
    u.z = f2_to_op16(v[4], v[5], bf16); u.w = f2_to_op16(v[6], v[7], bf16);
    *(uint4*)((uint16_t*)p + i) = u;
  } else {
    *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)p + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// element index -> (row, 8-column chunk): a shift when the chunks-per-row count is a power of two (every pitch the model uses),
// one 32-bit division otherwise (a 64-bit division per thread was most of these kernels' time); totals stay below 2^32
// (checked by the launchers).
__device__ __forceinline__ void nf_row_chunk(unsigned e, unsigned cpr, unsigned& m, unsigned& c0) {
  if ((cpr & (cpr - 1u)) == 0u) { const unsigned sh = 31u - (unsigned)__clz((int)cpr); m = e >> sh; c0 = (e & (cpr - 1u)) * 8u; }
  else { m = e / cpr; c0 = (e - m * cpr) * 8u; }
}

// G[M, ldg] = gradient at the base MLP's output: column 0 = d_density * exp(clamp(raw, -15, 15)) * selector
// (custom_functions.py:46-50), columns 1 .. ngeo = dXhead[:, geo_col0 ...] (the head's input gradient), the rest 0.
__global__ void k_nf_base_grad(long long M, int bf16, const void* __restrict__ Y, int ldy, const float* __restrict__ sel,
                               const float* __restrict__ d_density, const void* __restrict__ dXh, int ldx, int geo_col0, int ngeo,
                               void* __restrict__ G, int ldg, int act, float dbias) {
  if (ldg & 7) {                                  // narrow / odd pitches (the fused proposal path uses ldg = 1): one element per thread
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = e / ldg;
    const int c = (int)(e % ldg);
    if (m >= M) return;
    float v = 0.f;
    if (c == 0) v = d_density[m] * nf_density_slope(nf_load(Y, (size_t)m * ldy, bf16), act, dbias) * sel[m];
    else if (c <= ngeo && dXh) v = nf_load(dXh, (size_t)m * ldx + geo_col0 + c - 1, bf16);
    nf_store(G, (size_t)m * ldg + c, bf16, v);
    return;
  }
  unsigned m, c0;
  nf_row_chunk(blockIdx.x * blockDim.x + threadIdx.x, (unsigned)ldg >> 3, m, c0);
  if (m >= M) return;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if ((int)c0 <= ngeo) {                           // (chunks behind the last real column are all zero: no loads)
    if (bf16 && dXh && (geo_col0 & 7) == 0 && geo_col0 >= 8 && (ldx & 7) == 0 && geo_col0 + (int)c0 + 8 <= ldx) {
      // G columns c0 .. c0+7 = head-input-gradient columns geo_col0 + c0 - 1 ..: the 16 columns from geo_col0 + c0 - 8 as two
      // aligned 16-byte loads, shifted by one element
      const uint16_t* xr = (const uint16_t*)dXh + (size_t)m * ldx + geo_col0 + c0 - 8;
      const uint4 lo = *(const uint4*)xr, hi = *(const uint4*)(xr + 8);
      const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = (int)c0 + q, j = q + 7;                      // element geo_col0 + c0 - 8 + j = geo_col0 + c - 1
        const uint16_t h = (uint16_t)((j & 1) ? (w[j >> 1] >> 16) : w[j >> 1]);
        if (c == 0) v[q] = d_density[m] * nf_density_slope(nf_load(Y, (size_t)m * ldy, bf16), act, dbias) * sel[m];
        else if (c <= ngeo) v[q] = op16_to_f(h, bf16);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = (int)c0 + q;
        if (c == 0) v[q] = d_density[m] * nf_density_slope(nf_load(Y, (size_t)m * ldy, bf16), act, dbias) * sel[m];
        else if (c <= ngeo && dXh) v[q] = nf_load(dXh, (size_t)m * ldx + geo_col0 + c - 1, bf16);
      }
    }
  }
  nf_store8(G, (size_t)m * ldg + c0, bf16, v);
}

// head input X[M, ldx] = [SH(viewdir) (16, per ray) | geo = Ybase[:, 1 .. ngeo] | appearance embedding (per ray) | 0 ...]
// One workgroup per ray: everything but the geo columns is the same for the ray's S samples, so the row is built once as a
// template in LDS (in the output format) and every (sample, 8-column chunk) is a 16-byte copy of it -- only the chunks that
// hold geo columns touch the base output.  (One thread per chunk re-reading sh / app for every sample: 447 us for the
// cfg5 batch; a plain fill of the same 0.54 GB is 81 us.)
__global__ __launch_bounds__(256) void k_nf_head_input(int nrays, int S, int bf16, const float* __restrict__ sh, const void* __restrict__ Yb,
                                                       int ldy, int ngeo, const float* __restrict__ app, int napp, void* __restrict__ X,
                                                       int ldx) {
  __shared__ __attribute__((aligned(16))) float tmpl[1024];       // the row in fp32 (ldx <= 1024, checked by the launcher)
  const int ray = blockIdx.x;
  for (int c = threadIdx.x; c < ldx; c += 256)
    tmpl[c] = c < 16 ? sh[(size_t)ray * 16 + c] : (c >= 16 + ngeo && c < 16 + ngeo + napp) ? app[(size_t)ray * napp + (c - 16 - ngeo)] : 0.f;
  __syncthreads();
  const int cpr = ldx >> 3;
  for (int e = threadIdx.x; e < S * cpr; e += 256) {
    const int k = e / cpr, c0 = (e - k * cpr) * 8;
    const size_t m = (size_t)ray * S + k;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = tmpl[c0 + q];
    if (c0 < 16 + ngeo && c0 + 8 > 16) {                          // this chunk holds geo columns
      if (bf16 && (ldy & 7) == 0 && c0 <= ldy) {                  // (the 16-column window [c0-16, c0) lies inside the row)
        // output columns c0 .. c0+7 = base-output columns c0-15 .. c0-8: the 16 columns from c0-16 as two aligned 16-byte loads,
        // shifted by one element (eight 2-byte loads at odd element offsets before)
        const uint16_t* yr = (const uint16_t*)Yb + m * ldy + (c0 - 16);
        const uint4 lo = *(const uint4*)yr;
        const uint4 hi = *(const uint4*)(yr + 8);
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int j = q + 1;                                     // element c0 - 16 + j
          const uint16_t h = (uint16_t)((j & 1) ? (w[j >> 1] >> 16) : w[j >> 1]);
          if (c0 + q < 16 + ngeo) v[q] = op16_to_f(h, bf16);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int c = c0 + q;
          if (c >= 16 && c < 16 + ngeo) v[q] = nf_load(Yb, m * ldy + 1 + (c - 16), bf16);
        }
      }
    }
    nf_store8(X, m * ldx + c0, bf16, v);
  }
}

// d_app[ray, :] = sum over the ray's samples of dX[:, col0 .. col0 + napp); scatter-added into the embedding row
__global__ void k_nf_app_bwd(int nrays, int S, int bf16, const void* __restrict__ dX, int ldx, int col0, int napp,
                             const int* __restrict__ embed_idx, float* __restrict__ d_embedding) {
  // one wave per ray (4 rays per block): lane = embedding column, so a sample's napp columns are one contiguous segment per load
  // (one thread per (ray, column) walking S rows read a separate 64-byte sector per element)
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ray >= nrays) return;
  for (int c = lane; c < napp; c += 64) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four loads in flight (fixed combination order: deterministic per ray)
    int k = 0;
    for (; k + 4 <= S; k += 4) {
      const size_t r0 = ((size_t)ray * S + k) * ldx + col0 + c;
      s0 += nf_load(dX, r0, bf16); s1 += nf_load(dX, r0 + ldx, bf16); s2 += nf_load(dX, r0 + 2 * (size_t)ldx, bf16); s3 += nf_load(dX, r0 + 3 * (size_t)ldx, bf16);
    }
    for (; k < S; ++k) s0 += nf_load(dX, ((size_t)ray * S + k) * ldx + col0 + c, bf16);
    atomicAdd(&d_embedding[(size_t)embed_idx[ray] * napp + c], (s0 + s1) + (s2 + s3));
  }
}

// rgb = sigmoid(Y[:, 0..2] + rgb_bias)
__global__ void k_nf_rgb_act(long long M, int bf16, const void* __restrict__ Y, int ldy, float rgb_bias, float* __restrict__ rgb) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * 3) return;
  const long long m = e / 3;
  const int c = (int)(e % 3);
  rgb[e] = 1.f / (1.f + expf(-(nf_load(Y, (size_t)m * ldy + c, bf16) + rgb_bias)));
}

// G[M, ldg] = gradient at the rgb layer's output: columns 0..2 = d_rgb * rgb * (1 - rgb), the rest 0
__global__ void k_nf_rgb_grad(long long M, int bf16, const float* __restrict__ rgb, const float* __restrict__ d_rgb, void* __restrict__ G, int ldg) {
  const int cpr = ldg >> 3;                       // ldg is a multiple of 8 (checked by the launcher)
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long m = e / cpr;
  const int c0 = (int)(e - m * cpr) * 8;
  if (m >= M) return;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float r = rgb[m * 3 + c]; v[c] = d_rgb[m * 3 + c] * r * (1.f - r); }
  }
  nf_store8(G, (size_t)m * ldg + c0, bf16, v);
}

#define NF_LAUNCH1D(kern, total, ...)                                                                      \
  do {                                                                                                     \
    const long long tot_ = (total);                                                                        \
    if (tot_ > 0) hipLaunchKernelGGL(kern, dim3((unsigned)((tot_ + 255) / 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
  } while (0)

extern "C" int hugs_nf_density_act(long long M, int dtype, const void* Y, int ldy, int col, const float* sel, float* density,
                                   int density_act, float density_bias, void* stream) {
  HUGS_REQUIRE(density_act == 0 || density_act == 1, -2, "hugs_nf_density_act: density_act %d (0 trunc_exp, 1 softplus)", density_act);
  NF_LAUNCH1D(k_nf_density_act, M, M, dtype, Y, ldy, col, sel, density, density_act, density_bias);
  HUGS_CHECK_LAUNCH("hugs_nf_density_act");
  return 0;
}
extern "C" int hugs_nf_base_grad(long long M, int dtype, const void* Y, int ldy, const float* sel, const float* d_density,
                                 const void* dXh, int ldx, int geo_col0, int ngeo, void* G, int ldg, int density_act, float density_bias,
                                 void* stream) {
  HUGS_REQUIRE(M * (long long)((ldg & 7) ? ldg : (ldg >> 3)) < (1ll << 32), -3, "hugs_nf_base_grad: %lld x %d elements exceed the 32-bit index", M, ldg);
  NF_LAUNCH1D(k_nf_base_grad, (ldg & 7) ? M * ldg : M * (ldg >> 3), M, dtype, Y, ldy, sel, d_density, dXh, ldx, geo_col0, ngeo, G, ldg, density_act, density_bias);
  HUGS_CHECK_LAUNCH("hugs_nf_base_grad");
  return 0;
}
extern "C" int hugs_nf_head_input(long long M, int S, int dtype, const float* sh, const void* Yb, int ldy, int ngeo, const float* app,
                                  int napp, void* X, int ldx, void* stream) {
  HUGS_REQUIRE(16 + ngeo + napp <= ldx && ldx % 8 == 0 && ldx <= 1024, -3, "hugs_nf_head_input: %d columns do not fit the pitch %d (a multiple of 8, <= 1024)", 16 + ngeo + napp, ldx);
  HUGS_REQUIRE(S > 0 && M % S == 0 && M / S < (1ll << 31), -3, "hugs_nf_head_input: %lld rows are not whole rays of %d samples", M, S);
  if (M > 0) hipLaunchKernelGGL(k_nf_head_input, dim3((unsigned)(M / S)), dim3(256), 0, (hipStream_t)stream, (int)(M / S), S, dtype, sh, Yb, ldy, ngeo, app, napp, X, ldx);
  HUGS_CHECK_LAUNCH("hugs_nf_head_input");
  return 0;
}
extern "C" int hugs_nf_app_bwd(int nrays, int S, int dtype, const void* dX, int ldx, int col0, int napp, const int* embed_idx,
                               float* d_embedding, void* stream) {
  if (nrays > 0 && napp > 0)
    hipLaunchKernelGGL(k_nf_app_bwd, dim3((nrays + 3) / 4), dim3(256), 0, (hipStream_t)stream, nrays, S, dtype, dX, ldx, col0, napp, embed_idx, d_embedding);
  HUGS_CHECK_LAUNCH("hugs_nf_app_bwd");
  return 0;
}
extern "C" int hugs_nf_rgb_act(long long M, int dtype, const void* Y, int ldy, float rgb_bias, float* rgb, void* stream) {
  NF_LAUNCH1D(k_nf_rgb_act, M * 3, M, dtype, Y, ldy, rgb_bias, rgb);
  HUGS_CHECK_LAUNCH("hugs_nf_rgb_act");
  return 0;
}
extern "C" int hugs_nf_rgb_grad(long long M, int dtype, const float* rgb, const float* d_rgb, void* G, int ldg, void* stream) {
  HUGS_REQUIRE(ldg % 8 == 0, -3, "hugs_nf_rgb_grad: pitch %d is not a multiple of 8", ldg);
  NF_LAUNCH1D(k_nf_rgb_grad, M * (ldg >> 3), M, dtype, rgb, d_rgb, G, ldg);
  HUGS_CHECK_LAUNCH("hugs_nf_rgb_grad");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Fused proposal network (models/nerfacto.py:927-990 HashMLPDensityField: hash-grid features -> Linear -> ReLU -> Linear(1)
// -> trunc_exp, times the selector).  The reference runs it as a tiny-cuda-nn fully fused MLP; as two 128-padded GEMMs a
// 10 -> 64 -> 1 net moved ~4 KB per sample through HBM in a training step (8.4 M samples at the first proposal level).
// Here one thread owns one sample: features in registers (KP = padded input width), the hidden layer in 64 registers, fp32
// weights broadcast from LDS; nothing but the features, the raw density and the feature gradient touches memory.
// Weight gradients: a sum over samples per (k, n) -- each wave turns its 64 x 64 block of hidden gradients around through
// LDS so that lane n owns hidden unit n and walks the 64 samples (features broadcast from LDS), keeping dW0[:, n], db0[n],
// dw1[n] in registers for the whole kernel; workgroup partials go to a slab and are summed in a fixed order.
// ------------------------------------------------------------------------------------------------
#define PM_H 64
__host__ __device__ __forceinline__ constexpr int pm_slab_width(int KP) { return KP * PM_H + 2 * PM_H + 4; }

template <int BF16, int KP>
__device__ __forceinline__ void pm_load_x(const void* X, long long m, int ldx, float (&x)[KP]) {
  if (BF16) {
#pragma unroll
    for (int c = 0; c < KP / 8; ++c) {
      const uint4 u = *(const uint4*)((const uint16_t*)X + (size_t)m * ldx + c * 8);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (BF16 == 2) { x[c * 8 + 2 * q] = h16_to_f((uint16_t)w[q]); x[c * 8 + 2 * q + 1] = h16_to_f((uint16_t)(w[q] >> 16)); }
        else { x[c * 8 + 2 * q] = __uint_as_float(w[q] << 16); x[c * 8 + 2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < KP / 4; ++c) {
      const float4 u = *(const float4*)((const float*)X + (size_t)m * ldx + c * 4);
      x[c * 4] = u.x; x[c * 4 + 1] = u.y; x[c * 4 + 2] = u.z; x[c * 4 + 3] = u.w;
    }
  }
}

template <int KP>
__device__ __forceinline__ void pm_hidden(const float (&x)[KP], const float* sW0, const float* sb0, float (&h)[PM_H]) {
#pragma unroll
  for (int n = 0; n < PM_H; n += 4) { const float4 b = *(const float4*)(sb0 + n); h[n] = b.x; h[n + 1] = b.y; h[n + 2] = b.z; h[n + 3] = b.w; }
#pragma unroll
  for (int k = 0; k < KP; ++k)
#pragma unroll
    for (int n = 0; n < PM_H; n += 4) {
      const float4 w = *(const float4*)(sW0 + k * PM_H + n);
      h[n] = fmaf(x[k], w.x, h[n]); h[n + 1] = fmaf(x[k], w.y, h[n + 1]); h[n + 2] = fmaf(x[k], w.z, h[n + 2]); h[n + 3] = fmaf(x[k], w.w, h[n + 3]);
      if ((n & 15) == 12) __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from hoisting all 16 * KP weight reads (spills)
    }
}

template <int KP>
__device__ __forceinline__ void pm_stage_weights(int in_dim, int H, const float* W0, int ldw0, const float* b0, const float* w1, int ldw1,
                                                 float* sW0, float* sb0, float* sw1) {
  for (int e = threadIdx.x; e < KP * PM_H; e += blockDim.x) {
    const int k = e / PM_H, n = e % PM_H;
    sW0[e] = (k < in_dim && n < H) ? W0[(size_t)k * ldw0 + n] : 0.f;
  }
  if (threadIdx.x < PM_H) {
    const int n = threadIdx.x;
    sb0[n] = n < H ? b0[n] : 0.f;
    sw1[n] = n < H ? w1[(size_t)n * ldw1] : 0.f;
  }
}

template <int BF16, int KP>
__global__ __launch_bounds__(256) void k_nf_prop_fwd(long long M, int in_dim, int H, const void* __restrict__ X, int ldx,
                                                     const float* __restrict__ W0, int ldw0, const float* __restrict__ b0,
                                                     const float* __restrict__ w1, int ldw1, const float* __restrict__ b1,
                                                     const float* __restrict__ sel, float* __restrict__ raw, float* __restrict__ density, int act, float dbias) {
  __shared__ __attribute__((aligned(16))) float sW0[KP * PM_H];
  __shared__ __attribute__((aligned(16))) float sb0[PM_H];
  __shared__ __attribute__((aligned(16))) float sw1[PM_H];
  pm_stage_weights<KP>(in_dim, H, W0, ldw0, b0, w1, ldw1, sW0, sb0, sw1);
  __syncthreads();
  const float b1v = b1[0];
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    float x[KP], h[PM_H];
    pm_load_x<BF16, KP>(X, m, ldx, x);
    pm_hidden<KP>(x, sW0, sb0, h);
    float o = b1v;
#pragma unroll
    for (int n = 0; n < PM_H; n += 4) {
      const float4 w = *(const float4*)(sw1 + n);
      o = fmaf(fmaxf(h[n], 0.f), w.x, o); o = fmaf(fmaxf(h[n + 1], 0.f), w.y, o);
      o = fmaf(fmaxf(h[n + 2], 0.f), w.z, o); o = fmaf(fmaxf(h[n + 3], 0.f), w.w, o);
    }
    raw[m] = o;
    density[m] = nf_density_value(o, act, dbias) * sel[m];      // custom_functions.py:38-44 trunc_exp forward (or softplus), nerfacto.py:984-988 selector
  }
}

// (KP = 32 -- rows of 17..32 features, no shipped config -- holds x, dx and two 32-wide accumulator rows per lane: one workgroup
//  per CU, i.e. the whole register file, instead of 25 spilled registers at two)
template <int BF16, int KP>
__global__ __launch_bounds__(256, KP == 32 ? 1 : 2) void k_nf_prop_bwd(long long M, int in_dim, int H, const void* __restrict__ X, int ldx,
                                                        const float* __restrict__ W0, int ldw0, const float* __restrict__ b0,
                                                        const float* __restrict__ w1, int ldw1, const float* __restrict__ raw,
                                                        const float* __restrict__ sel, const float* __restrict__ d_density,
                                                        void* __restrict__ dX, float* __restrict__ slab, int act, float dbias) {
  // The hidden layer is walked in two halves of 32 units, the per-sample values of a half parked in LDS (T[unit][sample],
  // pitch 65: conflict-free by sample and by unit); only x, dx and the weight-gradient accumulators live in registers (the
  // fully unrolled 64-register form of the forward kernel spilled here; a whole-layer T left one workgroup per CU).
  // Turn-around: lane l owns unit 32*half + (l & 31) for the samples of its lane half (l >> 5); the two lane halves are
  // added at the end.
  constexpr int TP = 65, TW = 32 * TP, XW = 64 * KP;
  __shared__ __attribute__((aligned(16))) float sW0[KP * PM_H];
  __shared__ __attribute__((aligned(16))) float sb0[PM_H];
  __shared__ __attribute__((aligned(16))) float sw1[PM_H];
  __shared__ __attribute__((aligned(16))) float sBuf[4 * (TW + XW + 64)];
  pm_stage_weights<KP>(in_dim, H, W0, ldw0, b0, w1, ldw1, sW0, sb0, sw1);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* xs = sBuf + wv * (TW + XW + 64);         // [64][KP] (first: 16-byte aligned)
  float* rs = xs + XW;                            // [64]
  float* T = rs + 64;                             // [32][TP]
  float aW0[2][KP], ab0[2] = {0.f, 0.f}, aw1[2] = {0.f, 0.f}, ab1 = 0.f;
#pragma unroll
  for (int k = 0; k < KP; ++k) { aW0[0][k] = 0.f; aW0[1][k] = 0.f; }
  const int nl = lane & 31, sb = (lane >> 5) * 32;
  const float w1n[2] = {sw1[nl], sw1[32 + nl]};
  const long long ntile = (M + 255) / 256;
#define PM_WAVE_SYNC() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
  for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
    const long long m = t * 256 + threadIdx.x;
    const bool valid = m < M;
    const long long mm = valid ? m : M - 1;
    float x[KP], dx[KP];
    pm_load_x<BF16, KP>(X, mm, ldx, x);
    // d raw = d density * exp(clamp(raw, -15, 15)) * selector (custom_functions.py:46-50)
    const float r = valid ? d_density[mm] * nf_density_slope(raw[mm], act, dbias) * sel[mm] : 0.f;
    ab1 += r;
#pragma unroll
    for (int k = 0; k < KP; ++k) dx[k] = 0.f;
    if (__ballot(r != 0.f) == 0ull) {             // a wave of samples without gradient (outside the box, zero weight): dX = 0, done
      if (valid) {
        if (BF16) {
#pragma unroll
          for (int c = 0; c < KP / 8; ++c) *(uint4*)((uint16_t*)dX + (size_t)m * ldx + c * 8) = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
          for (int c = 0; c < KP / 4; ++c) *(float4*)((float*)dX + (size_t)m * ldx + c * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      continue;
    }
    rs[lane] = r;
#pragma unroll
    for (int k = 0; k < KP; k += 4) *(float4*)(xs + lane * KP + k) = make_float4(x[k], x[k + 1], x[k + 2], x[k + 3]);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      // ---- hidden pre-activations of this lane's sample (units 32 hh ..) -> T, and their share of dx ----
#pragma unroll 1
      for (int n = 0; n < 32; n += 4) {
        const int ng = hh * 32 + n;
        float4 h = *(const float4*)(sb0 + ng);
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const float4 w = *(const float4*)(sW0 + k * PM_H + ng);
          h.x = fmaf(x[k], w.x, h.x); h.y = fmaf(x[k], w.y, h.y); h.z = fmaf(x[k], w.z, h.z); h.w = fmaf(x[k], w.w, h.w);
        }
        T[n * TP + lane] = h.x; T[(n + 1) * TP + lane] = h.y; T[(n + 2) * TP + lane] = h.z; T[(n + 3) * TP + lane] = h.w;
        const float4 wv1 = *(const float4*)(sw1 + ng);
        const float d0 = h.x > 0.f ? r * wv1.x : 0.f, d1 = h.y > 0.f ? r * wv1.y : 0.f;
        const float d2 = h.z > 0.f ? r * wv1.z : 0.f, d3 = h.w > 0.f ? r * wv1.w : 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {          // dx[k] = sum_n W0[k][n] dh[n]
          const float4 w = *(const float4*)(sW0 + k * PM_H + ng);
          dx[k] = fmaf(w.x, d0, dx[k]); dx[k] = fmaf(w.y, d1, dx[k]); dx[k] = fmaf(w.z, d2, dx[k]); dx[k] = fmaf(w.w, d3, dx[k]);
        }
      }
      PM_WAVE_SYNC()
      // ---- lane (unit nl, sample half sb) walks 32 samples: dw1 += relu(h) r, db0 += dh, dW0[k] += x[k] dh ----
#pragma unroll 2
      for (int s_ = 0; s_ < 32; ++s_) {
        const float hv = T[nl * TP + sb + s_], rr = rs[sb + s_];
        aw1[hh] = fmaf(fmaxf(hv, 0.f), rr, aw1[hh]);
        const float d = hv > 0.f ? rr * w1n[hh] : 0.f;
        ab0[hh] += d;
#pragma unroll
        for (int k = 0; k < KP; k += 4) {
          const float4 xv = *(const float4*)(xs + (sb + s_) * KP + k);
          aW0[hh][k] = fmaf(xv.x, d, aW0[hh][k]); aW0[hh][k + 1] = fmaf(xv.y, d, aW0[hh][k + 1]);
          aW0[hh][k + 2] = fmaf(xv.z, d, aW0[hh][k + 2]); aW0[hh][k + 3] = fmaf(xv.w, d, aW0[hh][k + 3]);
        }
      }
      PM_WAVE_SYNC()
    }
    if (valid) {
      if (BF16) {
#pragma unroll
        for (int c = 0; c < KP / 8; ++c) {
          uint4 u;
          u.x = f2_to_op16(dx[c * 8], dx[c * 8 + 1], BF16); u.y = f2_to_op16(dx[c * 8 + 2], dx[c * 8 + 3], BF16);
          u.z = f2_to_op16(dx[c * 8 + 4], dx[c * 8 + 5], BF16); u.w = f2_to_op16(dx[c * 8 + 6], dx[c * 8 + 7], BF16);
          *(uint4*)((uint16_t*)dX + (size_t)m * ldx + c * 8) = u;
        }
      } else {
#pragma unroll
        for (int c = 0; c < KP / 4; ++c) *(float4*)((float*)dX + (size_t)m * ldx + c * 4) = make_float4(dx[c * 4], dx[c * 4 + 1], dx[c * 4 + 2], dx[c * 4 + 3]);
      }
    }
  }
#undef PM_WAVE_SYNC
  // ---- the two lane halves, then the four waves (fixed order), -> slab row ----
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
    for (int k = 0; k < KP; ++k) aW0[hh][k] += __shfl_xor(aW0[hh][k], 32);
    ab0[hh] += __shfl_xor(ab0[hh], 32);
    aw1[hh] += __shfl_xor(aw1[hh], 32);
  }
  __syncthreads();
  float* red = sBuf;                              // [4][KP + 3][64]
  if (lane < 32) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int k = 0; k < KP; ++k) red[(wv * (KP + 3) + k) * 64 + hh * 32 + lane] = aW0[hh][k];
      red[(wv * (KP + 3) + KP) * 64 + hh * 32 + lane] = ab0[hh];
      red[(wv * (KP + 3) + KP + 1) * 64 + hh * 32 + lane] = aw1[hh];
    }
  }
  red[(wv * (KP + 3) + KP + 2) * 64 + lane] = ab1;
  __syncthreads();
  float* row = slab + (size_t)blockIdx.x * pm_slab_width(KP);
  for (int e = threadIdx.x; e < (KP + 2) * 64; e += 256) {
    const int k = e >> 6, n = e & 63;
    row[e] = red[(0 * (KP + 3) + k) * 64 + n] + red[(1 * (KP + 3) + k) * 64 + n] + red[(2 * (KP + 3) + k) * 64 + n] + red[(3 * (KP + 3) + k) * 64 + n];
  }
  if (threadIdx.x == 0) {
    float b = 0.f;
    for (int w_ = 0; w_ < 4; ++w_)
      for (int l = 0; l < 64; ++l) b += red[(w_ * (KP + 3) + KP + 2) * 64 + l];
    row[(KP + 2) * 64] = b;
  }
}

// ------------------------------------------------------------------------------------------------
// The same proposal networks on the matrix cores, for 16-bit feature rows (in_dim <= 16, hidden <= 64): v_mfma_f32_16x16x16
// f16 / bf16, a wave per 64 samples.  Everything is arranged so that NO cross-lane transpose of the hidden layer is needed:
//   (1) Ht[n][s] = W0^T X^T   (A = W0^T rows n, B = X^T: a lane's B operand is 4 consecutive k of one sample = 8 contiguous
//       bytes of the feature row) leaves lane l with units 4(l/16)+j of sample l%16 -- which IS the B-operand layout of
//   (2) dX^T[k][s] = W0[k][:] dHt[:][s]   (K = units), whose result is 4 consecutive k of one sample: an 8-byte store;
//   (3) H[s][n] = X W0 computed a second time in the other orientation (A = X rows s: the registers of (1)'s B operand)
//       leaves lane l with unit l%16 of samples 4(l/16)+j -- the B-operand layout of
//   (4) dW0[k][n] += X^T[k][s] dH[s][n]   (K = samples), accumulated in registers over all tiles of the wave; only X^T
//       (64 x 16 values) goes through LDS.
// 64 MFMAs of 16 passes per 64 samples in the backward (16 in the forward) against ~400 k lane-FMAs on the VALU.  Operands
// are the 16-bit format of the mode (the weights are rounded to it: what the reference's autocast does to these Linear
// layers); accumulation, bias, relu and the reductions are fp32.  Weight-gradient slabs have the layout of the VALU kernels.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float pmf4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 pmh4_t;
typedef __attribute__((ext_vector_type(4))) short pms4_t;
template <int DT>
__device__ __forceinline__ pmf4_t pm_mfma(uint2 a, uint2 b, pmf4_t c) {
  if (DT == 2) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(pmh4_t, a), __builtin_bit_cast(pmh4_t, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(pms4_t, a), __builtin_bit_cast(pms4_t, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ uint2 pm_pack4(float a, float b, float c, float d) { return make_uint2(f2_to_op16(a, b, DT), f2_to_op16(c, d, DT)); }

// per-lane constants of both kernels: weight fragments and the bias / head vectors in the two accumulator layouts
template <int DT>
struct PmConst {
  uint2 wA[4];      // [nb]: W0[k = 4(l/16)+j][n = 16 nb + l%16]      (A of (1), B of (3))
  uint2 wX[4];      // [nb]: W0[k = l%16][n = 16 nb + 4(l/16)+j]      (A of (2))
  float b0a[4][4], w1a[4][4];   // units 16 nb + 4(l/16)+j  (layout of (1))
  float b0b[4], w1b[4];         // unit 16 nb + l%16        (layout of (3))
  __device__ __forceinline__ void load(int lane, int in_dim, int H, const float* W0, int ldw0, const float* b0, const float* w1, int ldw1) {
    const int lr = lane & 15, lq = lane >> 4;
    auto w = [&](int k, int n) { return (k < in_dim && n < H) ? W0[(size_t)k * ldw0 + n] : 0.f; };
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      wA[nb] = pm_pack4<DT>(w(4 * lq, 16 * nb + lr), w(4 * lq + 1, 16 * nb + lr), w(4 * lq + 2, 16 * nb + lr), w(4 * lq + 3, 16 * nb + lr));
      wX[nb] = pm_pack4<DT>(w(lr, 16 * nb + 4 * lq), w(lr, 16 * nb + 4 * lq + 1), w(lr, 16 * nb + 4 * lq + 2), w(lr, 16 * nb + 4 * lq + 3));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = 16 * nb + 4 * lq + j;
        b0a[nb][j] = n < H ? b0[n] : 0.f;
        w1a[nb][j] = n < H ? w1[(size_t)n * ldw1] : 0.f;
      }
      const int n = 16 * nb + lr;
      b0b[nb] = n < H ? b0[n] : 0.f;
      w1b[nb] = n < H ? w1[(size_t)n * ldw1] : 0.f;
    }
  }
};

template <int DT>
__global__ __launch_bounds__(256) void k_nf_prop_fwd_mfma(long long M, int in_dim, int H, const uint16_t* __restrict__ X, int ldx,
                                                          const float* __restrict__ W0, int ldw0, const float* __restrict__ b0,
                                                          const float* __restrict__ w1, int ldw1, const float* __restrict__ b1,
                                                          const float* __restrict__ sel, float* __restrict__ raw, float* __restrict__ density, int act, float dbias) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  PmConst<DT> C;
  C.load(lane, in_dim, H, W0, ldw0, b0, w1, ldw1);
  const float b1v = b1[0];
  const long long ntile = (M + 63) >> 6, wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
  auto load_tile = [&](long long t, uint2 (&xq)[4]) {
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
      const long long s = (t << 6) + 16 * sb + lr;
      xq[sb] = (t < ntile && s < M) ? *(const uint2*)(X + (size_t)s * ldx + 4 * lq) : make_uint2(0u, 0u);
    }
  };
  uint2 xn[4];
  load_tile(wave, xn);
  for (long long t = wave; t < ntile; t += nwave) {
    const long long s0 = t << 6;
    const uint2 xc[4] = {xn[0], xn[1], xn[2], xn[3]};
    load_tile(t + nwave, xn);                      // the next tile's rows are in flight during this tile's MFMAs
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
      const long long s = s0 + 16 * sb + lr;
      const uint2 xf = xc[sb];
      float o = 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        pmf4_t acc = {C.b0a[nb][0], C.b0a[nb][1], C.b0a[nb][2], C.b0a[nb][3]};
        acc = pm_mfma<DT>(C.wA[nb], xf, acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) o = fmaf(fmaxf(acc[j], 0.f), C.w1a[nb][j], o);
      }
      o += __shfl_xor(o, 16);
      o += __shfl_xor(o, 32);
      if (lq == 0 && s < M) {
        o += b1v;
        raw[s] = o;
        density[s] = nf_density_value(o, act, dbias) * sel[s];      // custom_functions.py:38-44 trunc_exp forward (or softplus), nerfacto.py:984-988 selector
      }
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void k_nf_prop_bwd_mfma(long long M, int in_dim, int H, const uint16_t* __restrict__ X, int ldx,
                                                          const float* __restrict__ W0, int ldw0, const float* __restrict__ b0,
                                                          const float* __restrict__ w1, int ldw1, const float* __restrict__ raw,
                                                          const float* __restrict__ sel, const float* __restrict__ d_density,
                                                          uint16_t* __restrict__ dX, float* __restrict__ slab, int dx_f32, int act, float dbias) {
  constexpr int KP = 16;
  __shared__ __attribute__((aligned(16))) uint16_t sXT[4][64 * 16];     // per wave: the tile's feature rows [sample][k]
  __shared__ float sR[4][64];
  __shared__ float sRed[4][(KP + 2) * 64 + 4];
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4, wv = threadIdx.x >> 6;
  PmConst<DT> C;
  C.load(lane, in_dim, H, W0, ldw0, b0, w1, ldw1);
  pmf4_t aW0[4];
  float ab0[4] = {0.f, 0.f, 0.f, 0.f}, aw1[4] = {0.f, 0.f, 0.f, 0.f}, ab1 = 0.f;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) aW0[nb] = pmf4_t{0.f, 0.f, 0.f, 0.f};
  uint16_t* xt = sXT[wv];
  float* rs = sR[wv];
  const long long ntile = (M + 63) >> 6, wave = (long long)blockIdx.x * 4 + wv, nwave = (long long)gridDim.x * 4;
#define PM_WAVE_SYNC() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
  // the next tile's inputs are requested before the current tile's MFMAs (one tile of loads in flight per wave: without it a
  // wave sat through the full global-load latency at the top of every tile)
  auto load_tile = [&](long long t, uint2 (&xq)[4], float& dq, float& rq, float& sq) {
    const long long s0 = t << 6, sm = s0 + lane;
    const bool in = t < ntile && sm < M;
    dq = in ? d_density[sm] : 0.f; rq = in ? raw[sm] : 0.f; sq = in ? sel[sm] : 0.f;
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
      const long long s = s0 + 16 * sb + lr;
      xq[sb] = (t < ntile && s < M) ? *(const uint2*)(X + (size_t)s * ldx + 4 * lq) : make_uint2(0u, 0u);
    }
  };
  uint2 xn[4];
  float dn, rn, sn;
  load_tile(wave, xn, dn, rn, sn);
  for (long long t = wave; t < ntile; t += nwave) {
    const long long s0 = t << 6;
    // d raw = d density * exp(clamp(raw, -15, 15)) * selector (custom_functions.py:46-50), sample s0 + lane
    const float r = dn * nf_density_slope(rn, act, dbias) * sn;
    uint2 xf[4] = {xn[0], xn[1], xn[2], xn[3]};
    load_tile(t + nwave, xn, dn, rn, sn);
    ab1 += r;
    if (__ballot(r != 0.f) == 0ull) {             // a tile without gradient (outside the box, zero weight): dX = 0, done
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const long long s = s0 + 16 * sb + lr;
        if (s < M) {
          if (dx_f32) *(float4*)((float*)dX + (size_t)s * ldx + 4 * lq) = make_float4(0.f, 0.f, 0.f, 0.f);
          else *(uint2*)(dX + (size_t)s * ldx + 4 * lq) = make_uint2(0u, 0u);
        }
      }
      continue;
    }
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) *(uint2*)(xt + (16 * sb + lr) * 16 + 4 * lq) = xf[sb];
    rs[lane] = r;
    PM_WAVE_SYNC()
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
      // ---- (1) + (2): dX rows of the 16 samples of this block ----
      const float r1 = rs[16 * sb + lr];
      pmf4_t dxacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        pmf4_t acc = {C.b0a[nb][0], C.b0a[nb][1], C.b0a[nb][2], C.b0a[nb][3]};
        acc = pm_mfma<DT>(C.wA[nb], xf[sb], acc);
        const uint2 dh = pm_pack4<DT>(acc[0] > 0.f ? r1 * C.w1a[nb][0] : 0.f, acc[1] > 0.f ? r1 * C.w1a[nb][1] : 0.f,
                                      acc[2] > 0.f ? r1 * C.w1a[nb][2] : 0.f, acc[3] > 0.f ? r1 * C.w1a[nb][3] : 0.f);
        dxacc = pm_mfma<DT>(C.wX[nb], dh, dxacc);
      }
      const long long s = s0 + 16 * sb + lr;
      if (s < M) {      // (dx_f32, round 5: the grid-input gradient leaves in fp32 -- a 16-bit store flushes scaled gradients below 6e-8 in the half mode)
        if (dx_f32) *(float4*)((float*)dX + (size_t)s * ldx + 4 * lq) = make_float4(dxacc[0], dxacc[1], dxacc[2], dxacc[3]);
        else *(uint2*)(dX + (size_t)s * ldx + 4 * lq) = pm_pack4<DT>(dxacc[0], dxacc[1], dxacc[2], dxacc[3]);
      }
      // ---- (3) + (4): this block's 16 samples into the weight gradients ----
      const float4 r4 = *(const float4*)(rs + 16 * sb + 4 * lq);
      const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
      // X^T fragment: k = lr, samples 16 sb + 4 lq + j
      const uint16_t* xcol = xt + (16 * sb + 4 * lq) * 16 + lr;
      const uint2 xtf = make_uint2((uint32_t)xcol[0] | ((uint32_t)xcol[16] << 16), (uint32_t)xcol[32] | ((uint32_t)xcol[48] << 16));
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        pmf4_t acc = {C.b0b[nb], C.b0b[nb], C.b0b[nb], C.b0b[nb]};
        acc = pm_mfma<DT>(xf[sb], C.wA[nb], acc);               // H[s = 4 lq + j][n = 16 nb + lr]
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          aw1[nb] = fmaf(fmaxf(acc[j], 0.f), rr[j], aw1[nb]);
          d[j] = acc[j] > 0.f ? rr[j] * C.w1b[nb] : 0.f;
          ab0[nb] += d[j];
        }
        aW0[nb] = pm_mfma<DT>(xtf, pm_pack4<DT>(d[0], d[1], d[2], d[3]), aW0[nb]);    // dW0[k = 4 lq + j][n = 16 nb + lr]
      }
    }
    PM_WAVE_SYNC()
  }
#undef PM_WAVE_SYNC
  // ---- lane groups (db0 / dw1 are partial over the group's samples), then the four waves in a fixed order -> slab row ----
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    ab0[nb] += __shfl_xor(ab0[nb], 16); ab0[nb] += __shfl_xor(ab0[nb], 32);
    aw1[nb] += __shfl_xor(aw1[nb], 16); aw1[nb] += __shfl_xor(aw1[nb], 32);
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) ab1 += __shfl_xor(ab1, d);
  float* red = sRed[wv];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(4 * lq + j) * 64 + 16 * nb + lr] = aW0[nb][j];
    if (lq == 0) { red[KP * 64 + 16 * nb + lr] = ab0[nb]; red[(KP + 1) * 64 + 16 * nb + lr] = aw1[nb]; }
  }
  if (lane == 0) red[(KP + 2) * 64] = ab1;
  __syncthreads();
  float* row = slab + (size_t)blockIdx.x * pm_slab_width(KP);
  for (int e = threadIdx.x; e <= (KP + 2) * 64; e += 256) row[e] = (sRed[0][e] + sRed[1][e]) + (sRed[2][e] + sRed[3][e]);
}

// slab [nblk][width] -> the four gradient leaves (fixed summation order over the workgroups)
__global__ __launch_bounds__(1024) void k_nf_prop_reduce(const float* __restrict__ slab, int nblk, int KP, int in_dim, int H,
                                                         float* __restrict__ gW0, int ldw0, float* __restrict__ gb0,
                                                         float* __restrict__ gw1, int ldw1, float* __restrict__ gb1) {
  // 64 columns x 16 row groups per workgroup (with 4 row groups the ~20 workgroups of this launch walked 256 slab rows each:
  // 84 us for 4.7 MB); the partial sums are combined in a fixed order
  const int width = pm_slab_width(KP);
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  __shared__ float red[16][64];
  float a = 0.f;
  if (e < width)
    for (int b = part; b < nblk; b += 16) a += slab[(size_t)b * width + e];
  red[part][threadIdx.x & 63] = a;
  __syncthreads();
  if (part != 0 || e >= width) return;
  a = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) a += red[q][threadIdx.x];
  const int k = e >> 6, n = e & 63;
  if (k < KP) { if (k < in_dim && n < H) gW0[(size_t)k * ldw0 + n] = a; }
  else if (k == KP) { if (n < H) gb0[n] = a; }
  else if (k == KP + 1) { if (n < H) gw1[(size_t)n * ldw1] = a; }
  else if (e == (KP + 2) * 64) gb1[0] = a;
}

extern "C" long long hugs_nf_prop_ws_bytes(int in_dim) {
  const int KP = in_dim <= 16 ? 16 : 32;
  return (long long)1024 * pm_slab_width(KP) * 4;
}
extern "C" int hugs_nf_prop_fwd(long long M, int in_dim, int hidden, int dtype, const void* X, int ldx, const float* W0, int ldw0,
                                const float* b0, const float* w1, int ldw1, const float* b1, const float* sel, float* raw,
                                float* density, int density_act, float density_bias, void* stream) {
  HUGS_REQUIRE(in_dim >= 1 && in_dim <= 32 && hidden >= 1 && hidden <= PM_H, -3, "hugs_nf_prop_fwd: %d -> %d -> 1 unsupported (<= 32, <= 64)", in_dim, hidden);
  const int KP = in_dim <= 16 ? 16 : 32;
  HUGS_REQUIRE(ldx >= KP && ldx % 8 == 0, -3, "hugs_nf_prop_fwd: feature pitch %d (needs >= %d, multiple of 8)", ldx, KP);
  if (M <= 0) return 0;
  const int grid = (int)((M + 255) / 256 < 2048 ? (M + 255) / 256 : 2048);
  hipStream_t st = (hipStream_t)stream;
  if (dtype && KP == 16 && ldx % 4 == 0) {        // 16-bit rows of <= 16 features: the matrix-core form
    const long long ntile = (M + 63) / 64;
    const int gm = (int)((ntile + 3) / 4 < 4096 ? (ntile + 3) / 4 : 4096);
    if (dtype == 2) hipLaunchKernelGGL(k_nf_prop_fwd_mfma<2>, dim3(gm), dim3(256), 0, st, M, in_dim, hidden, (const uint16_t*)X, ldx, W0, ldw0, b0, w1, ldw1, b1, sel, raw, density, density_act, density_bias);
    else hipLaunchKernelGGL(k_nf_prop_fwd_mfma<1>, dim3(gm), dim3(256), 0, st, M, in_dim, hidden, (const uint16_t*)X, ldx, W0, ldw0, b0, w1, ldw1, b1, sel, raw, density, density_act, density_bias);
    HUGS_CHECK_LAUNCH("hugs_nf_prop_fwd(mfma)");
    return 0;
  }
#define PM_FWD(B, K) hipLaunchKernelGGL((k_nf_prop_fwd<B, K>), dim3(grid), dim3(256), 0, st, M, in_dim, hidden, X, ldx, W0, ldw0, b0, w1, ldw1, b1, sel, raw, density, density_act, density_bias)
  if (dtype == 2) { if (KP == 16) PM_FWD(2, 16); else PM_FWD(2, 32); }
  else if (dtype) { if (KP == 16) PM_FWD(1, 16); else PM_FWD(1, 32); }
  else { if (KP == 16) PM_FWD(0, 16); else PM_FWD(0, 32); }
#undef PM_FWD
  HUGS_CHECK_LAUNCH("hugs_nf_prop_fwd");
  return 0;
}
extern "C" int hugs_nf_prop_bwd(long long M, int in_dim, int hidden, int dtype, const void* X, int ldx, const float* W0, int ldw0,
                                const float* b0, const float* w1, int ldw1, const float* raw, const float* sel,
                                const float* d_density, void* dX, float* gW0, float* gb0, float* gw1, float* gb1, void* ws,
                                int dx_f32, int density_act, float density_bias, void* stream) {
  HUGS_REQUIRE(in_dim >= 1 && in_dim <= 32 && hidden >= 1 && hidden <= PM_H, -3, "hugs_nf_prop_bwd: %d -> %d -> 1 unsupported (<= 32, <= 64)", in_dim, hidden);
  const int KP = in_dim <= 16 ? 16 : 32;
  HUGS_REQUIRE(ldx >= KP && ldx % 8 == 0, -3, "hugs_nf_prop_bwd: feature pitch %d (needs >= %d, multiple of 8)", ldx, KP);
  if (M <= 0) return 0;
  const long long ntile = (M + 255) / 256;
  const int grid = (int)(ntile < 1024 ? ntile : 1024);
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)ws;
  if (dtype && KP == 16 && ldx % 4 == 0) {        // 16-bit rows of <= 16 features: the matrix-core form (same slab layout)
    const long long nt64 = (M + 63) / 64;
    const int gm = (int)((nt64 + 3) / 4 < 1024 ? (nt64 + 3) / 4 : 1024);
    if (dtype == 2) hipLaunchKernelGGL(k_nf_prop_bwd_mfma<2>, dim3(gm), dim3(256), 0, st, M, in_dim, hidden, (const uint16_t*)X, ldx, W0, ldw0, b0, w1, ldw1, raw, sel, d_density, (uint16_t*)dX, slab, dx_f32, density_act, density_bias);
    else hipLaunchKernelGGL(k_nf_prop_bwd_mfma<1>, dim3(gm), dim3(256), 0, st, M, in_dim, hidden, (const uint16_t*)X, ldx, W0, ldw0, b0, w1, ldw1, raw, sel, d_density, (uint16_t*)dX, slab, dx_f32, density_act, density_bias);
    hipLaunchKernelGGL(k_nf_prop_reduce, dim3((pm_slab_width(KP) + 63) / 64), dim3(1024), 0, st, slab, gm, KP, in_dim, hidden, gW0, ldw0,
                       gb0, gw1, ldw1, gb1);
    HUGS_CHECK_LAUNCH("hugs_nf_prop_bwd(mfma)");
    return 0;
  }
  HUGS_REQUIRE(!dx_f32 || dtype == 0, -3, "hugs_nf_prop_bwd: fp32 feature gradients with 16-bit rows need the matrix-core form (<= 16 features, pitch %% 4 == 0)");
#define PM_BWD(B, K) hipLaunchKernelGGL((k_nf_prop_bwd<B, K>), dim3(grid), dim3(256), 0, st, M, in_dim, hidden, X, ldx, W0, ldw0, b0, w1, ldw1, raw, sel, d_density, dX, slab, density_act, density_bias)
  if (dtype == 2) { if (KP == 16) PM_BWD(2, 16); else PM_BWD(2, 32); }
  else if (dtype) { if (KP == 16) PM_BWD(1, 16); else PM_BWD(1, 32); }
  else { if (KP == 16) PM_BWD(0, 16); else PM_BWD(0, 32); }
#undef PM_BWD
  hipLaunchKernelGGL(k_nf_prop_reduce, dim3((pm_slab_width(KP) + 63) / 64), dim3(1024), 0, st, slab, grid, KP, in_dim, hidden, gW0, ldw0,
                     gb0, gw1, ldw1, gb1);
  HUGS_CHECK_LAUNCH("hugs_nf_prop_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, train.py:183: eps outside the sqrt of the bias-corrected second moment)
// ------------------------------------------------------------------------------------------------
__global__ void k_nf_adam(long long n, float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                          float* __restrict__ v, float lr, float b1, float b2, float eps, float bc1, float bc2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i];
  if (g != g) g = 0.f;
  const float mi = b1 * m[i] + (1.f - b1) * g, vi = b2 * v[i] + (1.f - b2) * g * g;
  m[i] = mi; v[i] = vi;
  theta[i] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
}

extern "C" int hugs_nf_adam(long long n, float* theta, const float* grad, float* m, float* v, float lr, float b1, float b2,
                            float eps, float bc1, float bc2, void* stream) {
  NF_LAUNCH1D(k_nf_adam, n, n, theta, grad, m, v, lr, b1, b2, eps, bc1, bc2);
  HUGS_CHECK_LAUNCH("hugs_nf_adam");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Dynamic loss scaling for the fp16 mode: torch.cuda.amp.GradScaler as nerfacto/train.py:168,210-213 drives it
// (scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()), entirely on the device -- no host read of
// found_inf.  state[0] = scale, state[1] = growth tracker, state[2] = found_inf of the step in flight (fp32).
//   hugs_amp_check   : state[2] = 1 if any gradient in the range is non-finite (the unscale_ pass's inf check)
//   hugs_amp_prepare : per parameter group, the bias corrections of ITS next update from its own update count (torch keeps
//                      `step` per parameter; a skipped step does not advance it), in double like torch's Python scalars
//   hugs_nf_adam_amp : torch.optim.Adam on grad / scale; does nothing when state[2] != 0 (scaler.step skips optimizer.step)
//   hugs_amp_update  : counts of the groups that took part += 1 unless skipped; scale *= backoff on overflow, *= growth
//                      after `interval` clean steps in a row (torch._amp_update_scale_); clears found_inf
// ------------------------------------------------------------------------------------------------
__global__ void k_amp_check(long long n, const float* __restrict__ grad, float* __restrict__ state) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  bool bad = false;
  if (i + 4 <= n) {
    const float4 g = *(const float4*)(grad + i);
    bad = !(fabsf(g.x) <= 3.4028234664e38f) || !(fabsf(g.y) <= 3.4028234664e38f) || !(fabsf(g.z) <= 3.4028234664e38f) || !(fabsf(g.w) <= 3.4028234664e38f);
  } else {
    for (long long k = i; k < n; ++k) bad = bad || !(fabsf(grad[k]) <= 3.4028234664e38f);
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) state[2] = 1.f;
}
__global__ void k_amp_prepare(int ngroups, const float* __restrict__ counts, double b1, double b2, float* __restrict__ bc) {
  const int g = threadIdx.x;
  if (g >= ngroups) return;
  const double k = (double)counts[g] + 1.0;
  bc[2 * g] = (float)(1.0 - pow(b1, k));
  bc[2 * g + 1] = (float)(1.0 - pow(b2, k));
}
__global__ void k_nf_adam_amp(long long n, float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                              float* __restrict__ v, float lr, float b1, float b2, float eps, const float* __restrict__ state,
                              const float* __restrict__ bc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || state[2] != 0.f) return;
  const float inv = 1.f / state[0], bc1 = bc[0], bc2 = bc[1];
  float g = grad[i] * inv;
  if (g != g) g = 0.f;
  const float mi = b1 * m[i] + (1.f - b1) * g, vi = b2 * v[i] + (1.f - b2) * g * g;
  m[i] = mi; v[i] = vi;
  theta[i] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
}
__global__ void k_amp_update(float* __restrict__ state, float* __restrict__ counts, unsigned group_mask, float growth, float backoff,
                             float interval) {
  if (threadIdx.x || blockIdx.x) return;
  if (state[2] != 0.f) { state[0] *= backoff; state[1] = 0.f; }
  else {
    for (int g = 0; g < 32; ++g) if (group_mask >> g & 1u) counts[g] += 1.f;
    const float t = state[1] + 1.f;
    if (t >= interval) { state[0] *= growth; state[1] = 0.f; } else state[1] = t;
  }
  state[2] = 0.f;
}

extern "C" int hugs_amp_check(long long n, const float* grad, float* state, void* stream) {
  HUGS_REQUIRE(grad && state && ((uintptr_t)grad & 15) == 0, -2, "hugs_amp_check: null or unaligned pointer");
  NF_LAUNCH1D(k_amp_check, (n + 3) / 4, n, grad, state);
  HUGS_CHECK_LAUNCH("hugs_amp_check");
  return 0;
}
extern "C" int hugs_amp_prepare(int ngroups, const float* counts, float b1, float b2, float* bc, void* stream) {
  HUGS_REQUIRE(ngroups >= 1 && ngroups <= 32 && counts && bc, -2, "hugs_amp_prepare: %d groups (1..32)", ngroups);
  hipLaunchKernelGGL(k_amp_prepare, dim3(1), dim3(32), 0, (hipStream_t)stream, ngroups, counts, (double)b1, (double)b2, bc);
  HUGS_CHECK_LAUNCH("hugs_amp_prepare");
  return 0;
}
extern "C" int hugs_nf_adam_amp(long long n, float* theta, const float* grad, float* m, float* v, float lr, float b1, float b2,
                                float eps, const float* state, const float* bc, void* stream) {
  NF_LAUNCH1D(k_nf_adam_amp, n, n, theta, grad, m, v, lr, b1, b2, eps, state, bc);
  HUGS_CHECK_LAUNCH("hugs_nf_adam_amp");
  return 0;
}
extern "C" int hugs_amp_update(float* state, float* counts, unsigned group_mask, float growth, float backoff, float interval,
                               void* stream) {
  HUGS_REQUIRE(state && counts, -2, "hugs_amp_update: null pointer");
  hipLaunchKernelGGL(k_amp_update, dim3(1), dim3(64), 0, (hipStream_t)stream, state, counts, group_mask, growth, backoff, interval);
  HUGS_CHECK_LAUNCH("hugs_amp_update");
  return 0;
}

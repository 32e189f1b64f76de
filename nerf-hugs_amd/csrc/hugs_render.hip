// Volumetric compositing, one WAVEFRONT per ray: density -> alpha / transmittance (wave prefix scan
// over the ray's samples held 4 / 8 / 16 per lane in registers: <= 256 / 512 / 1024 samples per level) -> weights -> colour, and the backward
// pass (wave suffix scan).  Nothing but the per-sample weights round-trips through HBM.
//
// Replaces (reference, MipNeRF360/internal): render.py:130-151 compute_alpha_weights, :185-244
// volumetric_rendering (incl. compute_extras: acc, distance_mean, distance percentiles via
// stepfun.py:298-308 weighted_percentile), and their autodiff (train_utils.py:454).
#include "hugs_common.h"

// RC_MAXC = samples per lane (template parameter): 4 -> S <= 256, 8 -> 512, 16 -> 1024, chosen per launch from S

template <int RC_MAXC>
struct RayScan {
  float sd[RC_MAXC];   // density * delta
  float pre[RC_MAXC];  // exclusive prefix of sd
  float w[RC_MAXC];
};

template <int RC_MAXC>
__device__ __forceinline__ void ray_weights(int S, int C, int lane, const float* __restrict__ dens,
                                            const float* __restrict__ td, float dnorm, int opaque, RayScan<RC_MAXC>& R) {
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < RC_MAXC; ++k) {
    const int i = lane * C + k;
    float v = 0.f;
    if (k < C && i < S) {
      v = dens[i] * ((td[i + 1] - td[i]) * dnorm);
      if (opaque && i == S - 1) v = __builtin_inff();
    }
    R.sd[k] = v;
    // the scan must not see the +inf of the last sample (it only matters for its own alpha)
    tot += (k < C && i < S - 1) ? v : 0.f;
  }
  // NB: cumsum(dd[:-1]) in the reference excludes the last interval, so do we.
  const float incl = wave_incl_scan_f(tot, lane);
  float run = __shfl_up(incl, 1);
  if (lane == 0) run = 0.f;
#pragma unroll
  for (int k = 0; k < RC_MAXC; ++k) {
    const int i = lane * C + k;
    R.pre[k] = run;
    const bool ok = k < C && i < S;
    const float alpha = 1.f - expf(-R.sd[k]);
    R.w[k] = ok ? alpha * expf(-run) : 0.f;
    run += (ok && i < S - 1) ? R.sd[k] : 0.f;
  }
}

// forward.  rgb_s may be null (proposal levels: rgb == 0).  extras may be null.
template <int RC_MAXC>
__global__ __launch_bounds__(256) void k_composite_fwd(int nrays, int S, const float* __restrict__ density,
                                                       const float* __restrict__ rgb_s, const float* __restrict__ tdist,
                                                       const float* __restrict__ dirs, int opaque, float bg,
                                                       const float* __restrict__ t_far, float* __restrict__ weights,
                                                       float* __restrict__ rgb_out, float* __restrict__ extras) {
  const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= nrays) return;
  const int C = (S + 63) >> 6;
  const float* td = tdist + (size_t)ray * (S + 1);
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  RayScan<RC_MAXC> R;
  ray_weights<RC_MAXC>(S, C, lane, density + (size_t)ray * S, td, dnorm, opaque, R);
  float acc = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, elog = 0.f;
#pragma unroll
  for (int k = 0; k < RC_MAXC; ++k) {
    const int i = lane * C + k;
    if (k < C && i < S) {
      const float w = R.w[k];
      weights[(size_t)ray * S + i] = w;
      acc += w;
      if (rgb_s) {
        const float* c = rgb_s + ((size_t)ray * S + i) * 3;
        c0 += w * c[0]; c1 += w * c[1]; c2 += w * c[2];
      }
      if (extras) elog += w * logf(0.5f * (td[i] + td[i + 1]));
    }
  }
  // inclusive per-lane weight sums for the percentile CDF
  float lane_w = 0.f;
#pragma unroll
  for (int k = 0; k < RC_MAXC; ++k) lane_w += R.w[k];
  const float incl_w = wave_incl_scan_f(lane_w, lane);
  float excl_w = __shfl_up(incl_w, 1);
  if (lane == 0) excl_w = 0.f;
  acc = wave_sum_f(acc); c0 = wave_sum_f(c0); c1 = wave_sum_f(c1); c2 = wave_sum_f(c2);
  const float bgw = fmaxf(0.f, 1.f - acc);
  if (lane == 0) {
    rgb_out[ray * 3] = c0 + bgw * bg; rgb_out[ray * 3 + 1] = c1 + bgw * bg; rgb_out[ray * 3 + 2] = c2 + bgw * bg;
  }
  if (extras) {
    elog = wave_sum_f(elog);
    // distance_mean (render.py:219-226)
    float dm = expf(elog / fmaxf(HUGS_EPS, acc));
    if (dm != dm) dm = __builtin_inff();
    dm = fminf(fmaxf(dm, td[0]), td[S]);
    // percentiles of [tdist, t_far] with weights [w, bg_w]: cw = [0, min(1, cumsum(w)), 1], S+2 posts
    float res[3];
    const float ps[3] = {0.05f, 0.5f, 0.95f};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float x = ps[q];
      // cnt = #{posts with cw <= x}; post 0 (=0) always counts, post S+1 (=1) never for x < 1
      int cnt = 0;
      float run = excl_w;
#pragma unroll
      for (int k = 0; k < RC_MAXC; ++k) {
        const int i = lane * C + k;
        run += R.w[k];
        if (k < C && i < S && fminf(run, 1.f) <= x) ++cnt;
      }
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) cnt += __shfl_xor(cnt, d);
      cnt += 1;                       // post 0
      // segment [cnt-1, cnt]: cw values and t values
      const int i1 = cnt;             // 1..S+1
      // cw at post p (1..S) = min(1, cumsum_w[p-1]); fetch cw[i1-1], cw[i1] by broadcasting from the owning lane
      float cw_lo = 0.f, cw_hi = 1.f;
      {
        float run2 = excl_w;
        float lo_c = -1.f, hi_c = -1.f;
#pragma unroll
        for (int k = 0; k < RC_MAXC; ++k) {
          const int i = lane * C + k;
          run2 += R.w[k];
          if (k < C && i < S) {
            if (i + 1 == i1 - 1) lo_c = fminf(run2, 1.f);
            if (i + 1 == i1) hi_c = fminf(run2, 1.f);
          }
        }
        lo_c = wave_max_f(lo_c); hi_c = wave_max_f(hi_c);
        if (i1 - 1 >= 1) cw_lo = lo_c;
        if (i1 <= S) cw_hi = hi_c;
      }
      const float t_lo = td[i1 - 1];
      const float t_hi = i1 <= S ? td[i1] : t_far[ray];
      const float dx_ = cw_hi - cw_lo;
      res[q] = dx_ > 0.f ? t_lo + (x - cw_lo) / dx_ * (t_hi - t_lo) : t_lo;
    }
    if (lane == 0) {
      float* e = extras + (size_t)ray * 5;
      e[0] = acc; e[1] = dm; e[2] = res[1]; e[3] = res[0]; e[4] = res[2];  // acc, mean, median, p5, p95
    }
  }
}

// backward: d_density, d_rgb_s from d_rgb_out [N,3] and d_w_extra [N,S] (nullable)
template <int RC_MAXC>
__global__ __launch_bounds__(256) void k_composite_bwd(int nrays, int S, const float* __restrict__ density,
                                                       const float* __restrict__ rgb_s, const float* __restrict__ tdist,
                                                       const float* __restrict__ dirs, int opaque, float bg,
                                                       const float* __restrict__ d_rgb_out,
                                                       const float* __restrict__ d_w_extra, float* __restrict__ d_density,
                                                       float* __restrict__ d_rgb_s, const float* __restrict__ raw,
                                                       float density_bias, float* __restrict__ d_raw) {
  const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= nrays) return;
  const int C = (S + 63) >> 6;
  const float* td = tdist + (size_t)ray * (S + 1);
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  RayScan<RC_MAXC> R;
  ray_weights<RC_MAXC>(S, C, lane, density + (size_t)ray * S, td, dnorm, opaque, R);
  float lane_w = 0.f;
#pragma unroll
  for (int k = 0; k < RC_MAXC; ++k) lane_w += R.w[k];
  const float acc = wave_sum_f(lane_w);
  const float g0 = d_rgb_out ? d_rgb_out[ray * 3] : 0.f, g1 = d_rgb_out ? d_rgb_out[ray * 3 + 1] : 0.f,
              g2 = d_rgb_out ? d_rgb_out[ray * 3 + 2] : 0.f;
  // d max(0, 1-acc)/d acc with jnp.maximum's tie rule (1/2 at equality)
  const float one_m = 1.f - acc;
  const float gb = one_m > 0.f ? 1.f : (one_m == 0.f ? 0.5f : 0.f);
  const float gbg = gb * bg * (g0 + g1 + g2);
  float gw[RC_MAXC], lane_gw = 0.f;
#pragma unroll
  for (int k = 0; k < RC_MAXC; ++k) {
    const int i = lane * C + k;
    float g = 0.f;
    if (k < C && i < S) {
      g = -gbg;
      if (d_w_extra) g += d_w_extra[(size_t)ray * S + i];
      if (rgb_s) {
        const float* c = rgb_s + ((size_t)ray * S + i) * 3;
        g += g0 * c[0] + g1 * c[1] + g2 * c[2];
        if (d_rgb_s) {
          float* o = d_rgb_s + ((size_t)ray * S + i) * 3;
          o[0] = R.w[k] * g0; o[1] = R.w[k] * g1; o[2] = R.w[k] * g2;
        }
      }
    }
    gw[k] = g;
    lane_gw += g * R.w[k];
  }
  // exclusive suffix sums of g_i w_i
  const float suf_incl = wave_incl_suffix_scan_f(lane_gw, lane);
  float suf = __shfl_down(suf_incl, 1);  // sum over lanes > this
  if (lane == 63) suf = 0.f;
#pragma unroll
  for (int k = RC_MAXC - 1; k >= 0; --k) {
    const int i = lane * C + k;
    if (k < C && i < S) {
      // d w_i / d sd_i = T_i e^{-sd_i};  d w_j / d sd_i = -w_j for j > i
      float dsd = gw[k] * expf(-(R.pre[k] + R.sd[k])) - suf;
      if (opaque && i == S - 1) dsd = 0.f;   // the last interval was replaced by +inf: no gradient
      const float dd = dsd * ((td[i + 1] - td[i]) * dnorm);
      d_density[(size_t)ray * S + i] = dd;
      // (hugs_composite_bwd_raw: the softplus head's pre-activation gradient in the same pass -- k_density_bwd_raw's arithmetic)
      if (d_raw) d_raw[(size_t)ray * S + i] = dd / (1.f + expf(-(raw[(size_t)ray * S + i] + density_bias)));
      suf += gw[k] * R.w[k];
    }
  }
}

extern "C" int hugs_composite_fwd(int nrays, int S, const float* density, const float* rgb_s, const float* tdist,
                                  const float* dirs, int opaque_background, float bg, const float* t_far, float* weights,
                                  float* rgb_out, float* extras, void* stream) {
  HUGS_REQUIRE(S >= 1 && S <= 1024, -3, "hugs_composite_fwd: %d samples per ray unsupported (<= 1024)", S);
  HUGS_REQUIRE(!extras || t_far, -3, "hugs_composite_fwd: extras need t_far");
  if (nrays <= 0) return 0;
#define HUGS_CF_LAUNCH(C_) hipLaunchKernelGGL(k_composite_fwd<C_>, dim3((nrays + 3) / 4), dim3(256), 0, (hipStream_t)stream, nrays, S, \
    density, rgb_s, tdist, dirs, opaque_background, bg, t_far, weights, rgb_out, extras)
  if (S <= 256) HUGS_CF_LAUNCH(4); else if (S <= 512) HUGS_CF_LAUNCH(8); else HUGS_CF_LAUNCH(16);
#undef HUGS_CF_LAUNCH
  HUGS_CHECK_LAUNCH("hugs_composite_fwd");
  return 0;
}

static int composite_bwd_impl(int nrays, int S, const float* density, const float* rgb_s, const float* tdist,
                              const float* dirs, int opaque_background, float bg, const float* d_rgb_out,
                              const float* d_w_extra, float* d_density, float* d_rgb_s, const float* raw, float density_bias,
                              float* d_raw, void* stream) {
  HUGS_REQUIRE(S >= 1 && S <= 1024, -3, "hugs_composite_bwd: %d samples per ray unsupported (<= 1024)", S);
  if (nrays <= 0) return 0;
#define HUGS_CB_LAUNCH(C_) hipLaunchKernelGGL(k_composite_bwd<C_>, dim3((nrays + 3) / 4), dim3(256), 0, (hipStream_t)stream, nrays, S, \
    density, rgb_s, tdist, dirs, opaque_background, bg, d_rgb_out, d_w_extra, d_density, d_rgb_s, raw, density_bias, d_raw)
  if (S <= 256) HUGS_CB_LAUNCH(4); else if (S <= 512) HUGS_CB_LAUNCH(8); else HUGS_CB_LAUNCH(16);
#undef HUGS_CB_LAUNCH
  HUGS_CHECK_LAUNCH("hugs_composite_bwd");
  return 0;
}
extern "C" int hugs_composite_bwd(int nrays, int S, const float* density, const float* rgb_s, const float* tdist,
                                  const float* dirs, int opaque_background, float bg, const float* d_rgb_out,
                                  const float* d_w_extra, float* d_density, float* d_rgb_s, void* stream) {
  return composite_bwd_impl(nrays, S, density, rgb_s, tdist, dirs, opaque_background, bg, d_rgb_out, d_w_extra, d_density, d_rgb_s, nullptr, 0.f,
                            nullptr, stream);
}
// + d_raw[m] = d_density[m] * sigmoid(raw[m] + density_bias): the density head's pre-activation gradient (hugs_density_bwd's first half)
// in the same pass -- one launch less on the way from the loss to the trunk's output gradient
extern "C" int hugs_composite_bwd_raw(int nrays, int S, const float* density, const float* rgb_s, const float* tdist,
                                      const float* dirs, int opaque_background, float bg, const float* d_rgb_out,
                                      const float* d_w_extra, float* d_density, float* d_rgb_s, const float* raw, float density_bias,
                                      float* d_raw, void* stream) {
  HUGS_REQUIRE(raw && d_raw, -2, "hugs_composite_bwd_raw: raw / d_raw is null");
  return composite_bwd_impl(nrays, S, density, rgb_s, tdist, dirs, opaque_background, bg, d_rgb_out, d_w_extra, d_density, d_rgb_s, raw,
                            density_bias, d_raw, stream);
}

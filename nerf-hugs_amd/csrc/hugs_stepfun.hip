// Hierarchical sampling level: [max-dilate] -> annealed logits -> softmax CDF -> inverse-CDF
// intervals -> s_to_t, one WAVEFRONT per ray, everything staged in LDS.
//
// Replaces (reference, MipNeRF360/internal): models.py:155-212 level prologue,
// stepfun.py:99-128 max_dilate_weights, :131-161 integrate_weights/invert_cdf,
// math.py:108-127 sorted_interp, stepfun.py:164-263 sample/sample_intervals,
// coord.py:63-99 construct_ray_warps (s_to_t).
//
// Bit-exact contract (DESIGN.md "canonical arithmetic"): every float op here is one IEEE
// binary32 op (this TU is built with -ffp-contract=off and correctly-rounded div), exp/log
// are the polynomial versions below, and the three order-sensitive sums (dilation renormaliser, softmax
// denominator, CDF prefix sum) follow the `sum_order` argument: 1 = reference order (numpy pairwise sums, sequential
// cumsum: what the reference-executed fixtures pin; the default of the Python layer), 0 = wave order (lane l owns
// elements 4l..4l+3; xor-butterfly reduce / Kogge-Stone scan across lanes + running max).
// Algorithms differ from the oracle's on purpose: 3-way rank merge instead of a sort,
// binary searches instead of compare matrices, window max over an index range.
#include "hugs_common.h"

#define SF_CAP 1024  // max bins handled per ray; a level whose largest array is <= 256 uses 4 elements per lane, <= 512: 8, else 16

__device__ __forceinline__ float sf_expf(float x) {
  if (x != x) return x;
  if (x > 88.72283172607421875f) return __builtin_inff();
  if (x < -103.97f) return 0.0f;
  float t = x * 1.44269504088896341f;
  float n = __builtin_floorf(t + 0.5f);
  float r = x - n * 0.693359375f;
  r = r - n * -2.12194440e-4f;
  float z = r * r;
  float p = 1.9875691500E-4f;
  p = p * r + 1.3981999507E-3f;
  p = p * r + 8.3334519073E-3f;
  p = p * r + 4.1665795894E-2f;
  p = p * r + 1.6666665459E-1f;
  p = p * r + 5.0000001201E-1f;
  p = p * z + r;
  p = p + 1.0f;
  int ni = (int)n;
  int n1 = ni / 2, n2 = ni - n1;
  float s1 = __uint_as_float((uint32_t)(n1 + 127) << 23);
  float s2 = __uint_as_float((uint32_t)(n2 + 127) << 23);
  return (p * s1) * s2;
}

__device__ __forceinline__ float sf_logf(float x);
__device__ __forceinline__ float sf_expf(float x);
// fn_fwd (inverse = false) / fn_inv (inverse = true) of Model.raydist_fn (coord.py:84-93)
__device__ __forceinline__ float sf_raywarp(float x, int raydist, bool inverse) {
  switch (raydist) {
    case 1: return 1.0f / x;
    case 2: return inverse ? sf_expf(x) : sf_logf(x);
    case 3: return inverse ? sf_logf(x) : sf_expf(x);
    case 4: return inverse ? x * x : sqrtf(x);
    case 5: return inverse ? sqrtf(x) : x * x;
    case 6: return inverse ? (x < 0.5f ? 2.0f * x : 0.5f / (1.0f - x)) : (x < 1.0f ? 0.5f * x : 1.0f - 0.5f / x);   // 'piecewise', coord.py:81-84
    default: return x;
  }
}

__device__ __forceinline__ float sf_logf(float x) {
  if (x != x || x < 0.0f) return __builtin_nanf("");
  if (x == 0.0f) return -__builtin_inff();
  if (x == __builtin_inff()) return x;
  int e = 0;
  uint32_t u = __float_as_uint(x);
  if ((u >> 23) == 0) { x = x * 8388608.0f; u = __float_as_uint(x); e = -23; }
  e += (int)(u >> 23) - 126;
  float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float y = 7.0376836292E-2f;
  y = y * m - 1.1514610310E-1f;
  y = y * m + 1.1676998740E-1f;
  y = y * m - 1.2420140846E-1f;
  y = y * m + 1.4249322787E-1f;
  y = y * m - 1.6668057665E-1f;
  y = y * m + 2.0000714765E-1f;
  y = y * m - 2.4999993993E-1f;
  y = y * m + 3.3333331174E-1f;
  y = y * m * z;
  float fe = (float)e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  float r = m + y;
  r = r + 0.693359375f * fe;
  return r;
}

// canonical wave-order sum over arr[0..n) (n <= 256), arr in LDS, zero padded reads
template <int C>
__device__ __forceinline__ float sf_wave_sum(const float* arr, int n, int lane) {
  float p = 0.0f;
#pragma unroll
  for (int k = 0; k < C; ++k) { int i = C * lane + k; float v = i < n ? arr[i] : 0.0f; p = k == 0 ? v : p + v; }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) p = p + __shfl_xor(p, d);
  return p;
}

// ORDER 1 ("reference order"): the order the reference's own calls have when its source is executed under numpy
// float32 (what tests/golden pins): jnp.sum and the softmax denominator = numpy's pairwise reduction -- blocks of <= 128
// elements on 8 interleaved accumulators combined ((0+1)+(2+3))+((4+5)+(6+7)), a sequential tail, halves split at
// a multiple of 8 above 128 -- and jnp.cumsum (stepfun.py:145) = a sequential left-to-right running sum.
// Lanes 0..7 hold the 8 accumulators of a block; IEEE addition is commutative, so the xor-butterfly over those 8
// lanes reproduces the fixed tree bit for bit.
__device__ __forceinline__ float sf_np_block(const float* a, int n, int lane) {      // n <= 128
  if (n < 8) { float r = 0.0f; for (int i = 0; i < n; ++i) r = r + a[i]; return r; }
  const int nb = n - (n & 7);
  float r = lane < 8 ? a[lane] : 0.0f;
#pragma unroll 4
  for (int i = 8; i < nb; i += 8) { if (lane < 8) r = r + a[i + lane]; }
  float t = r + __shfl_xor(r, 1);
  t = t + __shfl_xor(t, 2);
  t = t + __shfl_xor(t, 4);
  float res = __shfl(t, 0);
  for (int i = nb; i < n; ++i) res = res + a[i];
  return res;
}
template <int DEPTH>
__device__ __forceinline__ float sf_np_sum(const float* a, int n, int lane) {
  if constexpr (DEPTH == 0) return sf_np_block(a, n, lane);
  else {
    if (n <= 128) return sf_np_block(a, n, lane);
    int n2 = n >> 1; n2 -= n2 & 7;
    const float lo = sf_np_sum<DEPTH - 1>(a, n2, lane);
    const float hi = sf_np_sum<DEPTH - 1>(a + n2, n - n2, lane);
    return lo + hi;
  }
}

// # of j in [0,len) with (base[j] + off) <= x   (base ascending)
__device__ __forceinline__ int sf_count_le(const float* base, int len, float off, float x) {
  int lo = 0, hi = len;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (base[mid] + off <= x) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ int sf_count_lt(const float* base, int len, float off, float x) {
  int lo = 0, hi = len;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (base[mid] + off < x) lo = mid + 1; else hi = mid; }
  return lo;
}

template <int C>
struct SfLds {
  static constexpr int CAP = 64 * C;
  float tp[CAP + 4];   // previous fenceposts
  float p[CAP];        // pdf of previous bins / softmax weights
  float td[CAP + 4];   // dilated fenceposts, later cw0
  float wd[CAP];       // dilated weights
  float tin[CAP + 4];  // fenceposts fed to the sampler
  float cen[CAP];      // sampled centers
};

template <int C, int ORDER>
__global__ __launch_bounds__(256) void k_level_sample(
    int nrays, const float* __restrict__ t_prev, const float* __restrict__ w_prev, int n_prev, int do_dilate,
    float dilation, float dlo, float dhi, float anneal, float pad, const float* __restrict__ u_base,
    const float* __restrict__ jitter, int jitter_stride, int ns, int raydist, const float* __restrict__ near,
    const float* __restrict__ far, float* __restrict__ sdist, float* __restrict__ tdist, int32_t* __restrict__ idx_out,
    float* __restrict__ t_in_out, float* __restrict__ w_in_out, const float* __restrict__ anneal_dev) {
  __shared__ __attribute__((aligned(16))) SfLds<C> lds[4];
  if (anneal_dev) anneal = *anneal_dev;      // hugs_level_sample_fwd_dyn: the per-step value lives in device memory (captured step)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  const bool live = ray < nrays;
  SfLds<C>& L = lds[wv];
  const float eps2 = HUGS_EPS * HUGS_EPS;
  const float ninf = -__builtin_inff();

  if (live) {
    for (int i = lane; i <= n_prev; i += 64) L.tp[i] = t_prev[(size_t)ray * (n_prev + 1) + i];
  }
  __syncthreads();
  int n_in;
  if (do_dilate) {
    const int n = n_prev, m = 3 * n + 1;
    if (live) {
      for (int j = lane; j < n; j += 64) {
        float dt = L.tp[j + 1] - L.tp[j];
        L.p[j] = w_prev[(size_t)ray * n + j] / (dt > eps2 ? dt : eps2);
      }
      // 3-way rank merge of A = tp[0..n], B = tp[j]-dil (j<n), C = tp[j+1]+dil (j<n)
      for (int e = lane; e < m; e += 64) {
        float val; int rank;
        if (e <= n) {
          val = L.tp[e];
          rank = e + sf_count_lt(L.tp, n, -dilation, val) + sf_count_lt(L.tp + 1, n, dilation, val);
        } else if (e < 2 * n + 1) {
          int k = e - (n + 1);
          val = L.tp[k] + -dilation;
          rank = k + sf_count_le(L.tp, n + 1, 0.0f, val) + sf_count_lt(L.tp + 1, n, dilation, val);
        } else {
          int k = e - (2 * n + 1);
          val = L.tp[k + 1] + dilation;
          rank = k + sf_count_le(L.tp, n + 1, 0.0f, val) + sf_count_le(L.tp, n, -dilation, val);
        }
        val = val < dlo ? dlo : val;
        val = val > dhi ? dhi : val;
        L.td[rank] = val;
      }
    }
    __syncthreads();
    if (live) {
      for (int i = lane; i < m - 1; i += 64) {
        float x = L.td[i];
        int j0 = sf_count_le(L.tp + 1, n, dilation, x);       // first j with t1_j > x
        int j1 = sf_count_le(L.tp, n, -dilation, x) - 1;      // last j with t0_j <= x
        float best = 0.0f;
        for (int j = j0; j <= j1; ++j) { float pj = L.p[j]; best = pj > best ? pj : best; }
        L.wd[i] = best * (L.td[i + 1] - x);
      }
    }
    __syncthreads();
    float s = 1.0f;
    if (live) s = ORDER == 1 ? sf_np_sum<4>(L.wd, m - 1, lane) : sf_wave_sum<C>(L.wd, m - 1, lane);
    float den = s > eps2 ? s : eps2;
    n_in = 3 * n - 2;
    if (live) {
      for (int i = lane; i <= n_in; i += 64) L.tin[i] = L.td[i + 1];
      for (int i = lane; i < n_in; i += 64) L.p[i] = L.wd[i + 1] / den;   // p now holds w_in
    }
    __syncthreads();
  } else {
    n_in = n_prev;
    if (live) {
      for (int i = lane; i <= n_in; i += 64) L.tin[i] = L.tp[i];
      for (int i = lane; i < n_in; i += 64) L.p[i] = w_prev[(size_t)ray * n_prev + i];
    }
    __syncthreads();
  }
  if (live && t_in_out) {  // test hook: the (dilated, trimmed) step function the sampler sees
    for (int i = lane; i <= n_in; i += 64) t_in_out[(size_t)ray * (n_in + 1) + i] = L.tin[i];
    for (int i = lane; i < n_in; i += 64) w_in_out[(size_t)ray * n_in + i] = L.p[i];
  }
  // annealed logits (models.py:191-193) -> wd ; softmax -> p ; cw0 -> td
  float mx = ninf;
  if (live) {
    for (int i = lane; i < n_in; i += 64) {
      float lg = L.tin[i + 1] > L.tin[i] ? anneal * sf_logf(L.p[i] + pad) : ninf;
      L.wd[i] = lg;
      mx = lg > mx ? lg : mx;
    }
  }
  mx = wave_max_f(mx);
  __syncthreads();
  if (live) for (int i = lane; i < n_in; i += 64) L.wd[i] = sf_expf(L.wd[i] - mx);
  __syncthreads();
  float den = 1.0f;
  if (live) den = ORDER == 1 ? sf_np_sum<4>(L.wd, n_in, lane) : sf_wave_sum<C>(L.wd, n_in, lane);
  if (live) for (int i = lane; i < n_in; i += 64) L.p[i] = L.wd[i] / den;
  __syncthreads();
  if (live && ORDER == 1) {
    // jnp.cumsum: one lane walks the <= 1023 weights left to right (a float running sum cannot be re-associated)
    if (lane == 0) {
      float run = 0.0f;
      L.td[0] = 0.0f;
      // (four weights per LDS read: the additions stay strictly left to right, only the read latency is paid once per four)
      int i = 0;
      for (; i + 4 <= n_in - 1; i += 4) {
        const float4 v = *(const float4*)&L.p[i];
        run = run + v.x; L.td[i + 1] = run < 1.0f ? run : 1.0f;
        run = run + v.y; L.td[i + 2] = run < 1.0f ? run : 1.0f;
        run = run + v.z; L.td[i + 3] = run < 1.0f ? run : 1.0f;
        run = run + v.w; L.td[i + 4] = run < 1.0f ? run : 1.0f;
      }
      for (; i < n_in - 1; ++i) { run = run + L.p[i]; L.td[i + 1] = run < 1.0f ? run : 1.0f; }
      L.td[n_in] = 1.0f;
    }
  } else if (live) {
    // canonical inclusive scan of p[0..n_in-2]
    float v[C];
    float tot = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) { int i = C * lane + k; v[k] = i < n_in - 1 ? L.p[i] : 0.0f; tot = k == 0 ? v[k] : tot + v[k]; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { float t = __shfl_up(tot, d); if (lane >= d) tot = tot + t; }
    float run = __shfl_up(tot, 1);
    if (lane == 0) run = 0.0f;
    float cs[C];
#pragma unroll
    for (int k = 0; k < C; ++k) { run = run + v[k]; cs[k] = run; }
    // The tree-ordered prefix of lane l+1 can round below the sequential tail of lane l when the
    // next weight is tiny; a running max (exact, order independent) restores the monotone CDF the
    // interval search relies on.
    float pm = cs[C - 1];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { float t = __shfl_up(pm, d); if (lane >= d) pm = fmaxf(pm, t); }
    float pme = __shfl_up(pm, 1);
    if (lane == 0) pme = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      int i = C * lane + k;
      float c = fmaxf(cs[k], pme);
      if (i < n_in - 1) L.td[i + 1] = c < 1.0f ? c : 1.0f;
    }
    if (lane == 0) { L.td[0] = 0.0f; L.td[n_in] = 1.0f; }
  }
  __syncthreads();
  if (live) {
    const float jit = jitter ? jitter[(size_t)ray * jitter_stride] : 0.0f;
    for (int j = lane; j < ns; j += 64) {
      float u = u_base[j] + (jitter && jitter_stride > 1 ? jitter[(size_t)ray * jitter_stride + j] : jit);
      int cnt = sf_count_le(L.td, n_in + 1, 0.0f, u);
      int i0 = cnt > 0 ? cnt - 1 : 0;
      int i1 = cnt <= n_in ? cnt : n_in;
      float xp0 = L.td[i0], xp1 = L.td[i1], fp0 = L.tin[i0], fp1 = L.tin[i1];
      float off = (u - xp0) / (xp1 - xp0);
      if (off != off) off = 0.0f;
      off = off < 0.0f ? 0.0f : (off > 1.0f ? 1.0f : off);
      L.cen[j] = fp0 + off * (fp1 - fp0);
      if (idx_out) idx_out[(size_t)ray * ns + j] = i0;
    }
  }
  __syncthreads();
  if (live) {
    const float nr = near[ray], fr = far[ray];
    // coord.py:63-99: fn_fwd on the bounds, fn_inv on the blend.  raydist: 0 None, 1 reciprocal, 2 log (inverse exp),
    // 3 exp (inverse log), 4 sqrt (inverse square), 5 square (inverse sqrt); exp / log are the canonical polynomial
    // versions shared with the C oracle, sqrt is the correctly rounded one.
    const float s_near = sf_raywarp(nr, raydist, false);
    const float s_far = sf_raywarp(fr, raydist, false);
    for (int j = lane; j <= ns; j += 64) {
      float s;
      if (j == 0) {
        float mid0 = (L.cen[1] + L.cen[0]) / 2.0f;
        float first = 2.0f * L.cen[0] - mid0;
        s = first > dlo ? first : dlo;
      } else if (j == ns) {
        float midl = (L.cen[ns - 1] + L.cen[ns - 2]) / 2.0f;
        float last = 2.0f * L.cen[ns - 1] - midl;
        s = last < dhi ? last : dhi;
      } else {
        s = (L.cen[j] + L.cen[j - 1]) / 2.0f;
      }
      sdist[(size_t)ray * (ns + 1) + j] = s;
      float v = s * s_far + (1.0f - s) * s_near;
      tdist[(size_t)ray * (ns + 1) + j] = sf_raywarp(v, raydist, true);
    }
  }
}

__global__ void k_explog(const float* x, int n, float* ye, float* yl) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ye[i] = sf_expf(x[i]); yl[i] = sf_logf(x[i]); }
}

static int level_sample_impl(int nrays, const float* t_prev, const float* w_prev, int n_prev, int do_dilate,
                             float dilation, float domain_lo, float domain_hi, float anneal,
                             float resample_padding, const float* u_base, const float* jitter,
                             int jitter_stride, int num_samples, int raydist, int sum_order, const float* near,
                             const float* far, float* sdist, float* tdist, int32_t* idx_out,
                             float* t_in_out, float* w_in_out, const float* anneal_dev, void* stream) {
  HUGS_REQUIRE(num_samples > 1, -2, "num_samples must be > 1, is %d.", num_samples);
  HUGS_REQUIRE(num_samples <= SF_CAP, -3, "hugs_level_sample_fwd: num_samples %d > capacity %d", num_samples, SF_CAP);
  int n_in = do_dilate ? 3 * n_prev : n_prev;
  HUGS_REQUIRE(n_prev >= 1 && n_in <= SF_CAP, -3, "hugs_level_sample_fwd: %d input bins (%d after dilation) > capacity %d",
               n_prev, n_in, SF_CAP);
  HUGS_REQUIRE(sum_order == 0 || sum_order == 1, -4, "hugs_level_sample_fwd: sum_order must be 0 (wave order) or 1 (reference order)");
  HUGS_REQUIRE(raydist >= 0 && raydist <= 6, -4, "hugs_level_sample_fwd: raydist must be 0 (None), 1 reciprocal, 2 log, 3 exp, 4 sqrt, 5 square or 6 piecewise");
  if (nrays <= 0) return 0;
  // one lane-chunk for the whole level, chosen from its largest array (the oracle applies the same rule)
  const int big = n_in > num_samples ? n_in : num_samples;
#define HUGS_LS_LAUNCH(C_, O_) hipLaunchKernelGGL((k_level_sample<C_, O_>), dim3((nrays + 3) / 4), dim3(256), 0, (hipStream_t)stream, nrays, \
    t_prev, w_prev, n_prev, do_dilate, dilation, domain_lo, domain_hi, anneal, resample_padding, u_base, jitter, jitter_stride, \
    num_samples, raydist, near, far, sdist, tdist, idx_out, t_in_out, w_in_out, anneal_dev)
  if (big <= 256) { if (sum_order) HUGS_LS_LAUNCH(4, 1); else HUGS_LS_LAUNCH(4, 0); }
  else if (big <= 512) { if (sum_order) HUGS_LS_LAUNCH(8, 1); else HUGS_LS_LAUNCH(8, 0); }
  else { if (sum_order) HUGS_LS_LAUNCH(16, 1); else HUGS_LS_LAUNCH(16, 0); }      // 256 samples per level dilate to 766 bins
#undef HUGS_LS_LAUNCH
  HUGS_CHECK_LAUNCH("hugs_level_sample_fwd");
  return 0;
}
extern "C" int hugs_level_sample_fwd(int nrays, const float* t_prev, const float* w_prev, int n_prev, int do_dilate,
                                     float dilation, float domain_lo, float domain_hi, float anneal,
                                     float resample_padding, const float* u_base, const float* jitter,
                                     int jitter_stride, int num_samples, int raydist, int sum_order, const float* near,
                                     const float* far, float* sdist, float* tdist, int32_t* idx_out,
                                     float* t_in_out, float* w_in_out, void* stream) {
  return level_sample_impl(nrays, t_prev, w_prev, n_prev, do_dilate, dilation, domain_lo, domain_hi, anneal, resample_padding, u_base,
                           jitter, jitter_stride, num_samples, raydist, sum_order, near, far, sdist, tdist, idx_out, t_in_out, w_in_out,
                           nullptr, stream);
}
extern "C" int hugs_level_sample_fwd_dyn(int nrays, const float* t_prev, const float* w_prev, int n_prev, int do_dilate,
                                         float dilation, float domain_lo, float domain_hi, const float* anneal_dev,
                                         float resample_padding, const float* u_base, const float* jitter,
                                         int jitter_stride, int num_samples, int raydist, int sum_order, const float* near,
                                         const float* far, float* sdist, float* tdist, void* stream) {
  HUGS_REQUIRE(anneal_dev, -2, "hugs_level_sample_fwd_dyn: anneal_dev is null");
  return level_sample_impl(nrays, t_prev, w_prev, n_prev, do_dilate, dilation, domain_lo, domain_hi, 0.f, resample_padding, u_base,
                           jitter, jitter_stride, num_samples, raydist, sum_order, near, far, sdist, tdist, nullptr, nullptr, nullptr,
                           anneal_dev, stream);
}

__global__ void k_arith(const float* a, const float* b, int n, float* o) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { o[i] = a[i] / b[i]; o[n + i] = a[i] * b[i]; o[2 * n + i] = a[i] + b[i]; o[3 * n + i] = a[i] - b[i]; }
}
extern "C" int hugs_test_arith(const float* a, const float* b, int n, float* out4n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_arith, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, n, out4n);
  HUGS_CHECK_LAUNCH("hugs_test_arith");
  return 0;
}

extern "C" int hugs_test_explog(const float* x, int n, float* y_exp, float* y_log, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_explog, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n, y_exp, y_log);
  HUGS_CHECK_LAUNCH("hugs_test_explog");
  return 0;
}

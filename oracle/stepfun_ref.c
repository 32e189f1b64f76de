/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's step-function path
 * (MipNeRF360/internal/stepfun.py, math.py:108-127, models.py:155-212 of
 * cnhaox/NeRF-HuGS).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (nerf-hugs_amd) never does.
 *
 * It follows the reference's *formulation* (compare-matrix searches, full sort,
 * O(n*m) windowed max) -- deliberately not the algorithms the HIP kernels use
 * (rank-merge, binary search) -- so that the two are independent.
 *
 * Bit-exactness contract with the HIP path (DESIGN.md "canonical arithmetic"):
 *   - every float op is an IEEE-754 binary32 +,-,*,/ (no FMA contraction:
 *     build with -ffp-contract=off), so CPU and gfx950 agree bit for bit;
 *   - exp/log are the portable polynomial versions below (orc_expf/orc_logf),
 *     restated independently in csrc/hugs_stepfun.hip;
 *   - the three order-sensitive sums (softmax denominator, CDF prefix sum,
 *     dilation renormaliser) use the "wave order": 64 virtual lanes, lane l owns
 *     elements 4l..4l+3 summed left to right, lanes combined by an xor-butterfly
 *     (reduce) or a Kogge-Stone scan (prefix).  Capacity 256 elements; a level whose largest array exceeds
 *     256 (e.g. 128 proposal samples -> 382 dilated bins) uses 8 elements per lane (capacity 512) throughout.
 * XLA's own summation order is unknowable here -- "parity
 * unpinned" at that level; the golden fixtures pin this file against the
 * reference source executed under numpy float32 to ~1e-6.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_CAP 1024
#define ORC_EPS 1.1920928955078125e-07f /* finfo(float32).eps */

static float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ---- portable exp / log (Cephes-style polynomials, IEEE basic ops only) ---- */
float orc_expf(float x) {
  if (x != x) return x;
  if (x > 88.72283172607421875f) return INFINITY;
  if (x < -103.97f) return 0.0f;
  float t = x * 1.44269504088896341f;
  float n = floorf(t + 0.5f);
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
  /* scale by 2^ni in two exact-power steps so denormal results round once */
  int n1 = ni / 2, n2 = ni - n1;
  float s1 = bits2f((uint32_t)(n1 + 127) << 23);
  float s2 = bits2f((uint32_t)(n2 + 127) << 23);
  return (p * s1) * s2;
}

float orc_logf(float x) {
  if (x != x || x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (x == INFINITY) return x;
  int e = 0;
  uint32_t u = f2bits(x);
  if ((u >> 23) == 0) { /* subnormal: scale up exactly */
    x = x * 8388608.0f; u = f2bits(x); e = -23;
  }
  e += (int)(u >> 23) - 126;
  float m = bits2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
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

/* ---- canonical "wave order" sums over n <= 256 values ---- */
static int g_chunk = 4;   /* elements per virtual lane for the current level (4: capacity 256, 8: 512, 16: 1024) */
static int chunk_for(int nmax) { return nmax <= 256 ? 4 : (nmax <= 512 ? 8 : 16); }
static void lane_partials(const float* x, int n, float lane[64]) {
  for (int l = 0; l < 64; ++l) {
    float s = 0.0f;
    for (int k = 0; k < g_chunk; ++k) {
      float v = (g_chunk * l + k) < n ? x[g_chunk * l + k] : 0.0f;
      s = k == 0 ? v : s + v;          /* ((v0+v1)+v2)+... left to right */
    }
    lane[l] = s;
  }
}
/* Summation order of the three order-sensitive sums: 0 = wave order (above), 1 = the order the reference's own calls
 * have when its source runs under numpy float32 (tests/golden fixtures): jnp.cumsum = sequential left to right
 * (stepfun.py:145), jnp.sum / softmax denominator = numpy's pairwise reduction (8 accumulators over blocks of <= 128,
 * halves above).  XLA's own order is not knowable here; this is the only pinned one. */
static int g_order = 0;
void orc_set_sum_order(int order) { g_order = order; }
static float np_pairwise(const float* a, int n) {
  if (n < 8) { float r = 0.0f; for (int i = 0; i < n; ++i) r = r + a[i]; return r; }
  if (n <= 128) {
    float r[8];
    int i;
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
  }
  int n2 = n / 2; n2 -= n2 % 8;
  return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
float orc_wave_sum(const float* x, int n) {
  if (g_order == 1) return np_pairwise(x, n);
  float a[64], b[64];
  lane_partials(x, n, a);
  for (int d = 1; d < 64; d <<= 1) {
    for (int l = 0; l < 64; ++l) b[l] = a[l] + a[l ^ d];
    memcpy(a, b, sizeof a);
  }
  return a[0];
}
/* inclusive prefix sum, out[i] = x[0]+..+x[i] in wave order */
void orc_wave_cumsum(const float* x, int n, float* out) {
  if (g_order == 1) { float run = 0.0f; for (int i = 0; i < n; ++i) { run = run + x[i]; out[i] = run; } return; }
  float a[64], b[64];
  lane_partials(x, n, a);
  for (int d = 1; d < 64; d <<= 1) { /* Kogge-Stone inclusive scan of lane totals */
    for (int l = 0; l < 64; ++l) b[l] = l >= d ? a[l] + a[l - d] : a[l];
    memcpy(a, b, sizeof a);
  }
  for (int l = 0; l < 64; ++l) {
    float run = l ? a[l - 1] : 0.0f;
    for (int k = 0; k < g_chunk; ++k) {
      int i = g_chunk * l + k;
      if (i >= n) break;
      run = run + x[i];
      out[i] = run;
    }
  }
}

static int cmp_float(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

/* stepfun.py:99-128 max_dilate_weights(renormalize=True): one ray.
 * t[n+1], w[n] -> t_dil[3n+1], w_dil[3n].  Returns 0, or -1 if 3n > capacity. */
static int max_dilate_weights_(const float* t, const float* w, int n, float dilation,
                               float lo, float hi, float* t_dil, float* w_dil) {
  const float eps2 = ORC_EPS * ORC_EPS;
  if (3 * n > ORC_CAP) return -1;
  float p[ORC_CAP], t0[ORC_CAP], t1[ORC_CAP];
  for (int j = 0; j < n; ++j) {
    float dt = t[j + 1] - t[j];
    p[j] = w[j] / (dt > eps2 ? dt : eps2);           /* weight_to_pdf, stepfun.py:89-91 */
    t0[j] = t[j] - dilation;
    t1[j] = t[j + 1] + dilation;
  }
  int m = 3 * n + 1;
  for (int j = 0; j <= n; ++j) t_dil[j] = t[j];
  for (int j = 0; j < n; ++j) { t_dil[n + 1 + j] = t0[j]; t_dil[2 * n + 1 + j] = t1[j]; }
  qsort(t_dil, m, sizeof(float), cmp_float);          /* jnp.sort, stepfun.py:103 */
  for (int i = 0; i < m; ++i) {                        /* jnp.clip, stepfun.py:104 */
    float v = t_dil[i]; v = v < lo ? lo : v; v = v > hi ? hi : v; t_dil[i] = v;
  }
  for (int i = 0; i < m - 1; ++i) {                    /* windowed max, stepfun.py:105-113 */
    float best = 0.0f;
    for (int j = 0; j < n; ++j)
      if (t0[j] <= t_dil[i] && t1[j] > t_dil[i] && p[j] > best) best = p[j];
    w_dil[i] = best * (t_dil[i + 1] - t_dil[i]);     /* pdf_to_weight, stepfun.py:94-96 */
  }
  float s = orc_wave_sum(w_dil, m - 1);
  float den = s > eps2 ? s : eps2;
  for (int i = 0; i < m - 1; ++i) w_dil[i] = w_dil[i] / den;  /* stepfun.py:126-127 */
  return 0;
}

int orc_max_dilate_weights(const float* t, const float* w, int n, float dilation,
                           float lo, float hi, float* t_dil, float* w_dil) {
  g_chunk = chunk_for(3 * n);
  return max_dilate_weights_(t, w, n, dilation, lo, hi, t_dil, w_dil);
}

/* stepfun.py:131-161 + math.py:108-127: invert the CDF of softmax(logits) on t at u.
 * Writes centers[ns] and the interval index idx[ns] (last i with cw0[i] <= u). */
int orc_invert_cdf(const float* u, int ns, const float* t, const float* logits, int n,
                   float* centers, int32_t* idx) {
  if (n > ORC_CAP) return -1;
  float w[ORC_CAP], cw0[ORC_CAP + 1], cs[ORC_CAP];
  float mx = -INFINITY;
  for (int i = 0; i < n; ++i) if (logits[i] > mx) mx = logits[i];
  for (int i = 0; i < n; ++i) w[i] = orc_expf(logits[i] - mx);
  float den = orc_wave_sum(w, n);
  for (int i = 0; i < n; ++i) w[i] = w[i] / den;
  cw0[0] = 0.0f;
  if (n > 1) orc_wave_cumsum(w, n - 1, cs);
  /* canonical CDF = running max of the wave-order prefix sums: the tree-ordered scan is not
   * monotone to the last ulp, the reference's max/min interval search assumes it is. */
  for (int i = 1; i < n - 1; ++i) if (cs[i] < cs[i - 1]) cs[i] = cs[i - 1];
  for (int i = 0; i < n - 1; ++i) cw0[i + 1] = cs[i] < 1.0f ? cs[i] : 1.0f;
  cw0[n] = 1.0f;
  for (int j = 0; j < ns; ++j) {
    /* compare-matrix form of math.sorted_interp */
    float x = u[j];
    float xp0 = cw0[0], xp1 = cw0[n], fp0 = t[0], fp1 = t[n];
    int last_true = 0;
    for (int i = 0; i <= n; ++i) {
      int mask = x >= cw0[i];
      float a = mask ? cw0[i] : cw0[0]; if (a > xp0) xp0 = a;
      float b = mask ? t[i] : t[0];     if (b > fp0) fp0 = b;
      float c = !mask ? cw0[i] : cw0[n]; if (c < xp1) xp1 = c;
      float d = !mask ? t[i] : t[n];     if (d < fp1) fp1 = d;
      if (mask) last_true = i;
    }
    float off = (x - xp0) / (xp1 - xp0);
    if (off != off) off = 0.0f;                      /* nan_to_num(.,0) */
    off = off < 0.0f ? 0.0f : (off > 1.0f ? 1.0f : off);
    centers[j] = fp0 + off * (fp1 - fp0);
    idx[j] = last_true;
  }
  return 0;
}

/* stepfun.py:214-263 sample_intervals given explicit u (= sample()'s u, a7). */
static int sample_intervals_(const float* u, int ns, const float* t, const float* logits, int n,
                             float lo, float hi, float* out /*ns+1*/, int32_t* idx /*ns*/) {
  if (ns <= 1 || ns > ORC_CAP) return -2;
  float c[ORC_CAP];
  int rc = orc_invert_cdf(u, ns, t, logits, n, c, idx);
  if (rc) return rc;
  for (int j = 0; j < ns - 1; ++j) out[j + 1] = (c[j + 1] + c[j]) / 2.0f;
  float first = 2.0f * c[0] - out[1];
  float last = 2.0f * c[ns - 1] - out[ns - 1];
  out[0] = first > lo ? first : lo;
  out[ns] = last < hi ? last : hi;
  return 0;
}

int orc_sample_intervals(const float* u, int ns, const float* t, const float* logits, int n,
                         float lo, float hi, float* out, int32_t* idx) {
  g_chunk = chunk_for(n > ns ? n : ns);
  return sample_intervals_(u, ns, t, logits, n, lo, hi, out, idx);
}

/* coord.py:84-93: fn_fwd (inverse = 0) / fn_inv (inverse = 1) of Model.raydist_fn.
 * raydist: 0 None, 1 reciprocal, 2 log <-> exp, 3 exp <-> log, 4 sqrt <-> square, 5 square <-> sqrt. */
static float orc_raywarp(float x, int raydist, int inverse) {
  switch (raydist) {
    case 1: return 1.0f / x;
    case 2: return inverse ? orc_expf(x) : orc_logf(x);
    case 3: return inverse ? orc_logf(x) : orc_expf(x);
    case 4: return inverse ? x * x : sqrtf(x);
    case 5: return inverse ? sqrtf(x) : x * x;
    case 6: return inverse ? (x < 0.5f ? 2.0f * x : 0.5f / (1.0f - x)) : (x < 1.0f ? 0.5f * x : 1.0f - 0.5f / x);   /* 'piecewise', coord.py:81-84 */
    default: return x;
  }
}

/* One sampling level for one ray, following models.py:155-212:
 *   [dilate (level>0)] -> trim [1:-1] -> annealed logits -> sample_intervals -> s_to_t.
 * u = u_base[j] + jitter (one float add, as stepfun.py:203-209 does with the
 * linspace + uniform draw).  raydist: 0 = linear (fn None), 1 = reciprocal
 * (coord.py:63-99). */
int orc_level_sample(const float* t_prev, const float* w_prev, int n_prev, int do_dilate,
                     float dilation, float lo, float hi, float anneal, float resample_padding,
                     const float* u_base, float jitter, int ns, int raydist, float near,
                     float far, float* sdist, float* tdist, int32_t* idx,
                     float* t_in_out, float* w_in_out, int* n_in_out) {
  float tb[ORC_CAP + 1], wb[ORC_CAP], lg[ORC_CAP], u[ORC_CAP];
  const float* t_in = t_prev; const float* w_in = w_prev; int n = n_prev;
  { /* one lane-chunk for the whole level, from its largest array (same rule as the HIP launch) */
    int big = do_dilate ? 3 * n_prev : n_prev;
    if (ns > big) big = ns;
    g_chunk = chunk_for(big);
  }
  if (do_dilate) {
    float td[ORC_CAP + 1], wd[ORC_CAP];
    int rc = max_dilate_weights_(t_prev, w_prev, n_prev, dilation, lo, hi, td, wd);
    if (rc) return rc;
    n = 3 * n_prev - 2;                       /* sdist[1:-1] has 3n-1 posts -> 3n-2 bins */
    memcpy(tb, td + 1, (n + 1) * sizeof(float));
    memcpy(wb, wd + 1, n * sizeof(float));
    t_in = tb; w_in = wb;
  }
  if (n > ORC_CAP || ns > ORC_CAP) return -1;
  for (int i = 0; i < n; ++i)                 /* models.py:191-193 */
    lg[i] = t_in[i + 1] > t_in[i] ? anneal * orc_logf(w_in[i] + resample_padding) : -INFINITY;
  for (int j = 0; j < ns; ++j) u[j] = u_base[j] + jitter;
  int rc = sample_intervals_(u, ns, t_in, lg, n, lo, hi, sdist, idx);
  if (rc) return rc;
  float s_near = orc_raywarp(near, raydist, 0);
  float s_far = orc_raywarp(far, raydist, 0);
  for (int j = 0; j <= ns; ++j) {             /* coord.py:98: fn_inv(s*s_far + (1-s)*s_near) */
    float v = sdist[j] * s_far + (1.0f - sdist[j]) * s_near;
    tdist[j] = orc_raywarp(v, raydist, 1);
  }
  if (t_in_out) memcpy(t_in_out, t_in, (n + 1) * sizeof(float));
  if (w_in_out) memcpy(w_in_out, w_in, n * sizeof(float));
  if (n_in_out) *n_in_out = n;
  return 0;
}

/* The same level with one jitter draw PER SAMPLE (stepfun.py:203-209 with single_jitter = False: uniform(rng, t.shape[:-1] + (num_samples,))):
 * u[j] = u_base[j] + jitter[j].  Implemented by folding the per-sample draws into the u grid it hands on (one float add per sample, as
 * the reference's `linspace + uniform`). */
int orc_level_sample_pj(const float* t_prev, const float* w_prev, int n_prev, int do_dilate,
                        float dilation, float lo, float hi, float anneal, float resample_padding,
                        const float* u_base, const float* jitter_ps, int ns, int raydist, float near,
                        float far, float* sdist, float* tdist, int32_t* idx) {
  float u[ORC_CAP];
  if (ns > ORC_CAP) return -1;
  for (int j = 0; j < ns; ++j) u[j] = u_base[j] + jitter_ps[j];
  return orc_level_sample(t_prev, w_prev, n_prev, do_dilate, dilation, lo, hi, anneal, resample_padding, u, 0.0f, ns, raydist, near, far,
                          sdist, tdist, idx, 0, 0, 0);
}

/* batched wrapper with per-sample jitter [nrays, ns] */
int orc_level_sample_batch_pj(int nrays, const float* t_prev, const float* w_prev, int n_prev,
                              int do_dilate, float dilation, float lo, float hi, float anneal,
                              float resample_padding, const float* u_base, const float* jitter,
                              int ns, int raydist, const float* near, const float* far,
                              float* sdist, float* tdist, int32_t* idx) {
  for (int r = 0; r < nrays; ++r) {
    int rc = orc_level_sample_pj(t_prev + (size_t)r * (n_prev + 1), w_prev + (size_t)r * n_prev, n_prev,
                                 do_dilate, dilation, lo, hi, anneal, resample_padding, u_base,
                                 jitter + (size_t)r * ns, ns, raydist, near[r], far[r],
                                 sdist + (size_t)r * (ns + 1), tdist + (size_t)r * (ns + 1), idx + (size_t)r * ns);
    if (rc) return rc;
  }
  return 0;
}

/* batched wrapper: rays are rows. jitter may be NULL (deterministic). */
int orc_level_sample_batch(int nrays, const float* t_prev, const float* w_prev, int n_prev,
                           int do_dilate, float dilation, float lo, float hi, float anneal,
                           float resample_padding, const float* u_base, const float* jitter,
                           int ns, int raydist, const float* near, const float* far,
                           float* sdist, float* tdist, int32_t* idx) {
  for (int r = 0; r < nrays; ++r) {
    int rc = orc_level_sample(t_prev + (size_t)r * (n_prev + 1), w_prev + (size_t)r * n_prev, n_prev,
                              do_dilate, dilation, lo, hi, anneal, resample_padding, u_base,
                              jitter ? jitter[r] : 0.0f, ns, raydist, near[r], far[r],
                              sdist + (size_t)r * (ns + 1), tdist + (size_t)r * (ns + 1),
                              idx + (size_t)r * ns, 0, 0, 0);
    if (rc) return rc;
  }
  return 0;
}

/* stepfun.py:30-53 searchsorted (compare-matrix definition), one row. */
void orc_searchsorted(const float* a, int na, const float* v, int nv, int32_t* lo, int32_t* hi) {
  for (int j = 0; j < nv; ++j) {
    int l = 0, h = na - 1;
    int lmax = 0, hmin = na - 1;
    for (int i = 0; i < na; ++i) {
      int ge = v[j] >= a[i];
      l = ge ? i : 0;       if (l > lmax) lmax = l;
      h = !ge ? i : na - 1; if (h < hmin) hmin = h;
    }
    lo[j] = lmax; hi[j] = hmin;
  }
}

void orc_expf_vec(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) y[i] = orc_expf(x[i]); }
void orc_logf_vec(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) y[i] = orc_logf(x[i]); }

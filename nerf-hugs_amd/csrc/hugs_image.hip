// Eval-side image metrics on the device: the two numbers eval.py writes per test image
// (image.py:127-141 MetricHarness): PSNR from the mean squared error and dm_pix.ssim.
//   SSIM (dm_pix.ssim defaults, the mip-NeRF recipe): 11-tap Gaussian (sigma 1.5), 'valid' separable filtering of
//   a, b, a^2, b^2, ab per channel; variances clamped at eps^2, covariance clamped to the Cauchy-Schwarz bound;
//   mean over the (H-10) x (W-10) x C map.
// One block = a 32x8 output tile of one channel: (42x18) input tile of both images staged in LDS, horizontal
// pass into LDS (5 planes), vertical pass from LDS; per-block partial sums, reduced in a fixed order.
// HBM-bound by construction: each input pixel is read ~(42*18)/(32*8) = 2.95x from L2, once from HBM.
#include "hugs_common.h"

namespace {

constexpr int FS = 11, HW = 5, TX = 32, TY = 8, IX = TX + FS - 1, IY = TY + FS - 1;

struct SsimFilt { float w[FS]; };

__global__ void __launch_bounds__(TX * TY)
k_ssim(int H, int W, int C, const float* __restrict__ a, const float* __restrict__ b, SsimFilt filt, float c1, float c2,
       float* __restrict__ partial) {
  __shared__ float sa[IY][IX + 1], sb[IY][IX + 1];
  __shared__ float hp[5][IY][TX + 1];
  __shared__ float red[TX * TY / 64];
  const int c = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const int tid = threadIdx.y * TX + threadIdx.x;
  for (int i = tid; i < IX * IY; i += TX * TY) {
    int ly = i / IX, lx = i - ly * IX;
    int gy = y0 + ly, gx = x0 + lx;
    bool in = gy < H && gx < W;
    size_t g = ((size_t)gy * W + gx) * C + c;
    sa[ly][lx] = in ? a[g] : 0.f;
    sb[ly][lx] = in ? b[g] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < TX * IY; i += TX * TY) {   // horizontal pass: 18 rows x 32 columns
    int ly = i / TX, lx = i - ly * TX;
    float m0 = 0.f, m1 = 0.f, s00 = 0.f, s11 = 0.f, s01 = 0.f;
#pragma unroll
    for (int k = 0; k < FS; ++k) {
      float va = sa[ly][lx + k], vb = sb[ly][lx + k], w = filt.w[k];
      m0 += w * va; m1 += w * vb; s00 += w * (va * va); s11 += w * (vb * vb); s01 += w * (va * vb);
    }
    hp[0][ly][lx] = m0; hp[1][ly][lx] = m1; hp[2][ly][lx] = s00; hp[3][ly][lx] = s11; hp[4][ly][lx] = s01;
  }
  __syncthreads();
  float v = 0.f;
  const int ox = x0 + threadIdx.x, oy = y0 + threadIdx.y;
  if (ox < W - (FS - 1) && oy < H - (FS - 1)) {
    float q[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < FS; ++k) {
      float w = filt.w[k];
#pragma unroll
      for (int j = 0; j < 5; ++j) q[j] += w * hp[j][threadIdx.y + k][threadIdx.x];
    }
    float mu00 = q[0] * q[0], mu11 = q[1] * q[1], mu01 = q[0] * q[1];
    const float eps2 = HUGS_EPS * HUGS_EPS;
    float s00 = fmaxf(eps2, q[2] - mu00), s11 = fmaxf(eps2, q[3] - mu11), s01 = q[4] - mu01;
    float bound = sqrtf(s00 * s11);
    float mag = fminf(bound, fabsf(s01));
    s01 = s01 > 0.f ? mag : (s01 < 0.f ? -mag : 0.f);
    v = ((2.f * mu01 + c1) * (2.f * s01 + c2)) / ((mu00 + mu11 + c1) * (s00 + s11 + c2));
  }
  v = wave_sum_f(v);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < TX * TY / 64; ++i) s += red[i];
    partial[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
  }
}

// out[0] = scale * sum(partial[0..n)) in a fixed order (double accumulation: the map has up to ~1e7 terms)
__global__ void __launch_bounds__(256) k_reduce_partials(int n, const float* __restrict__ partial, double scale,
                                                          float* __restrict__ out) {
  __shared__ double sm[256];
  double s = 0.;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sm[threadIdx.x] += sm[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sm[0] * scale);
}

__global__ void __launch_bounds__(256) k_sqerr_partial(long long n, const float* __restrict__ a,
                                                        const float* __restrict__ b, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float d = a[i] - b[i];
    s += d * d;
  }
  s = wave_sum_f(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace

extern "C" long long hugs_ssim_ws_bytes(int H, int W, int C) {
  if (H < FS || W < FS || C < 1) return 0;
  long long gx = (W - (FS - 1) + TX - 1) / TX, gy = (H - (FS - 1) + TY - 1) / TY;
  return gx * gy * C * (long long)sizeof(float);
}

extern "C" int hugs_ssim(int H, int W, int C, const float* a, const float* b, float max_val, float filter_sigma, float k1,
                         float k2, float* ws, float* out, void* stream) {
  HUGS_REQUIRE(a && b && ws && out, -2, "hugs_ssim: null pointer");
  HUGS_REQUIRE(H >= FS && W >= FS && C >= 1 && C <= 65535, -2,
               "hugs_ssim: image %dx%dx%d smaller than the %d-tap window", H, W, C, FS);
  SsimFilt f;
  double sum = 0., tmp[FS];
  for (int i = 0; i < FS; ++i) {
    double z = (i - HW) / (double)filter_sigma;
    tmp[i] = exp(-0.5 * z * z);
    sum += tmp[i];
  }
  for (int i = 0; i < FS; ++i) f.w[i] = (float)(tmp[i] / sum);
  int gx = (W - (FS - 1) + TX - 1) / TX, gy = (H - (FS - 1) + TY - 1) / TY;
  HUGS_REQUIRE(gy <= 65535, -2, "hugs_ssim: image too tall");
  float c1 = (k1 * max_val) * (k1 * max_val), c2 = (k2 * max_val) * (k2 * max_val);
  k_ssim<<<dim3(gx, gy, C), dim3(TX, TY), 0, (hipStream_t)stream>>>(H, W, C, a, b, f, c1, c2, ws);
  HUGS_CHECK_LAUNCH("k_ssim");
  double cnt = (double)(H - (FS - 1)) * (W - (FS - 1)) * C;
  k_reduce_partials<<<1, 256, 0, (hipStream_t)stream>>>(gx * gy * C, ws, 1. / cnt, out);
  HUGS_CHECK_LAUNCH("k_reduce_partials");
  return 0;
}

/* out[0] = mean((a-b)^2) over n elements; ws: 1024 floats */
extern "C" int hugs_mse(long long n, const float* a, const float* b, float* ws, float* out, void* stream) {
  HUGS_REQUIRE(a && b && ws && out && n > 0, -2, "hugs_mse: bad arguments (n=%lld)", n);
  int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  k_sqerr_partial<<<blocks, 256, 0, (hipStream_t)stream>>>(n, a, b, ws);
  HUGS_CHECK_LAUNCH("k_sqerr_partial");
  k_reduce_partials<<<1, 256, 0, (hipStream_t)stream>>>(blocks, ws, 1. / (double)n, out);
  HUGS_CHECK_LAUNCH("k_reduce_partials");
  return 0;
}

// MLP heads around the MFMA trunks: density head (N=1 "GEMV"), view-branch per-ray bias, rgb head,
// and their backward passes.  All are HBM-streaming kernels (one pass over an [M, K] activation).
//
// Replaces (reference, MipNeRF360/internal/models.py): :456 raw_density = Dense(1), :467 softplus(raw-1);
// :488-512 view-direction / GLO concat feeding Dense(128) -- the per-ray constant part of that matmul is
// hoisted into a per-ray bias; :514-519 rgb = sigmoid(Dense(3)) * (1+2*pad) - pad.
#include "hugs_common.h"
#include <algorithm>

template <int BF16>
__device__ __forceinline__ void decode8(const uint4 p, float v[8]) {      // eight 16-bit operands (BF16 = 1 bf16, 2 half) -> fp32
  if (BF16 == 2) {
    v[0] = h16_to_f((uint16_t)p.x); v[1] = h16_to_f((uint16_t)(p.x >> 16)); v[2] = h16_to_f((uint16_t)p.y); v[3] = h16_to_f((uint16_t)(p.y >> 16));
    v[4] = h16_to_f((uint16_t)p.z); v[5] = h16_to_f((uint16_t)(p.z >> 16)); v[6] = h16_to_f((uint16_t)p.w); v[7] = h16_to_f((uint16_t)(p.w >> 16));
  } else {
    v[0] = __uint_as_float(p.x << 16); v[1] = __uint_as_float(p.x & 0xffff0000u);
    v[2] = __uint_as_float(p.y << 16); v[3] = __uint_as_float(p.y & 0xffff0000u);
    v[4] = __uint_as_float(p.z << 16); v[5] = __uint_as_float(p.z & 0xffff0000u);
    v[6] = __uint_as_float(p.w << 16); v[7] = __uint_as_float(p.w & 0xffff0000u);
  }
}
template <int BF16>
__device__ __forceinline__ void load8(const void* base, size_t elem_off, float v[8]) {
  if (BF16 == 2) {
    const uint4 p = *(const uint4*)((const uint16_t*)base + elem_off);
    v[0] = h16_to_f((uint16_t)p.x); v[1] = h16_to_f((uint16_t)(p.x >> 16)); v[2] = h16_to_f((uint16_t)p.y); v[3] = h16_to_f((uint16_t)(p.y >> 16));
    v[4] = h16_to_f((uint16_t)p.z); v[5] = h16_to_f((uint16_t)(p.z >> 16)); v[6] = h16_to_f((uint16_t)p.w); v[7] = h16_to_f((uint16_t)(p.w >> 16));
  } else if (BF16) {
    const uint4 p = *(const uint4*)((const uint16_t*)base + elem_off);
    v[0] = __uint_as_float(p.x << 16); v[1] = __uint_as_float(p.x & 0xffff0000u);
    v[2] = __uint_as_float(p.y << 16); v[3] = __uint_as_float(p.y & 0xffff0000u);
    v[4] = __uint_as_float(p.z << 16); v[5] = __uint_as_float(p.z & 0xffff0000u);
    v[6] = __uint_as_float(p.w << 16); v[7] = __uint_as_float(p.w & 0xffff0000u);
  } else {
    const float4 a = *(const float4*)((const float*)base + elem_off);
    const float4 b = *(const float4*)((const float*)base + elem_off + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
template <int BF16>
__device__ __forceinline__ void store8(void* base, size_t elem_off, const float v[8]) {
  if (BF16) {
    uint4 pk;
    pk.x = f2_to_op16(v[0], v[1], BF16); pk.y = f2_to_op16(v[2], v[3], BF16);
    pk.z = f2_to_op16(v[4], v[5], BF16); pk.w = f2_to_op16(v[6], v[7], BF16);
    *(uint4*)((uint16_t*)base + elem_off) = pk;
  } else {
    *(float4*)((float*)base + elem_off) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)base + elem_off + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

__device__ __forceinline__ float softplusf(float x) {  // logaddexp(x, 0)
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

// ---- density head forward: raw[m] = Y[m,:] . w + b ; density = softplus(raw + density_bias) ----
// LPR = lanes that share a row = min(64, K/8 rounded up to a power of two): a 256-wide PropMLP row is 32 lanes x 16 bytes, so a
// wave reads TWO rows per load instruction (round 4: with one row per wave half the lanes idled and every row paid a 6-step
// shuffle reduction: 300 us per 1 M x 256 rows = 1.7 TB/s); two row groups are kept in flight.
template <int BF16>
__global__ __launch_bounds__(256) void k_density_fwd(int M, int K, int lpr, const void* __restrict__ Y, int ldy,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     float density_bias, float* __restrict__ raw,
                                                     float* __restrict__ density) {
  const int lane = threadIdx.x & 63;
  const int rpw = 64 / lpr, sub = lane / lpr, l = lane % lpr;          // rows per wave, this lane's row of the group, its column lane
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const float bb = b[0];
  auto dot = [&](long long m) {
    float acc = 0.f;
    if (m < M)
      for (int k0 = l * 8; k0 < K; k0 += lpr * 8) {
        float v[8];
        load8<BF16>(Y, (size_t)m * ldy + k0, v);
        const float4 w0 = *(const float4*)(w + k0), w1 = *(const float4*)(w + k0 + 4);
        acc += v[0] * w0.x + v[1] * w0.y + v[2] * w0.z + v[3] * w0.w + v[4] * w1.x + v[5] * w1.y + v[6] * w1.z + v[7] * w1.w;
      }
    return acc;
  };
  for (long long g = wave * 2; g * rpw < M; g += nwaves * 2) {
    const long long m0 = g * rpw + sub, m1 = (g + 1) * rpw + sub;
    float a0 = dot(m0), a1 = dot(m1);
    for (int d = 1; d < lpr; d <<= 1) { a0 += __shfl_xor(a0, d); a1 += __shfl_xor(a1, d); }
    if (l == 0) {
      if (m0 < M) { const float r = a0 + bb; raw[m0] = r; density[m0] = softplusf(r + density_bias); }
      if (m1 < M) { const float r = a1 + bb; raw[m1] = r; density[m1] = softplusf(r + density_bias); }
    }
  }
}

// d_raw[m] = d_density[m] * sigmoid(raw[m] + density_bias)
__global__ void k_density_bwd_raw(int M, const float* __restrict__ d_density, const float* __restrict__ raw,
                                  float density_bias, float* __restrict__ d_raw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) d_raw[i] = d_density[i] / (1.f + expf(-(raw[i] + density_bias)));
}

// weighted column sums: slab[blk][k] = sum_{m in blk} r[m] * Y[m,k]; slab[blk][K] = sum r[m]   (K <= 2048)
// Thread (rr, c): column chunk c of C = K/8, row lane rr of R = 256 / C: a block reads R rows per step, 16 bytes per lane, every
// lane busy (round 4: with one thread per chunk walking all rows a 256-wide PropMLP used 32 of 256 threads: 0.36 TB/s).  The R
// partial sums of a column are added in LDS in row-lane order: deterministic.
template <int BF16>
__global__ __launch_bounds__(256) void k_wcolsum(int M, int K, int rows_per_blk, const void* __restrict__ Y, int ldy,
                                                 const float* __restrict__ r, float* __restrict__ slab) {
  __shared__ float red[256 * 9];
  const int t = threadIdx.x, C = K >> 3, R = C >= 256 ? 1 : 256 / C;
  const int c = t % C, rr = t / C;
  const int m0 = blockIdx.x * rows_per_blk, m1 = min(M, m0 + rows_per_blk);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float rs = 0.f;
  if (rr < R) {
#pragma unroll 4
    for (int m = m0 + rr; m < m1; m += R) {
      float v[8];
      load8<BF16>(Y, (size_t)m * ldy + c * 8, v);
      const float rm = r[m];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += rm * v[q];
      rs += rm;
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) red[t * 9 + q] = acc[q];
  red[t * 9 + 8] = rs;
  __syncthreads();
  if (t < C) {
    float o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = 0; g < R; ++g)
#pragma unroll
      for (int q = 0; q < 9; ++q) o[q] += red[(g * C + t) * 9 + q];
    float* dst = slab + (size_t)blockIdx.x * (K + 4) + t * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = o[q];
    if (t == 0) slab[(size_t)blockIdx.x * (K + 4) + K] = o[8];
  }
}

// out[i] = sum_b slab[b*stride + i], i < width; columns width .. width+width2-1 go to out2 (the bias gradient next to the
// weight gradient: one launch).  1024 threads = 16 columns x 64 row groups, fixed order (round 4: 64 columns x 16 groups put a
// 388-column reduction of 1024 partial rows on SEVEN workgroups with 64 dependent loads per thread: 20-26 us, twice on the
// backward's critical chain).
#define SRS_COLS 16
__global__ __launch_bounds__(1024) void k_slab_reduce_small(const float* __restrict__ slab, int nblk, int width, int stride,
                                                            float* __restrict__ out, int width2 = 0, float* __restrict__ out2 = nullptr) {
  __shared__ float red[64][SRS_COLS + 1];
  const int cx = threadIdx.x & (SRS_COLS - 1), rg = threadIdx.x / SRS_COLS;
  const int i = blockIdx.x * SRS_COLS + cx;
  float a = 0.f;
  if (i < width + width2) {
#pragma unroll 4
    for (int b = rg; b < nblk; b += 64) a += slab[(size_t)b * stride + i];
  }
  red[rg][cx] = a;
  __syncthreads();
  if (rg == 0 && i < width + width2) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 64; ++g) t += red[g][cx];
    if (i < width) out[i] = t; else out2[i - width] = t;
  }
}

// out[m,n] = r[m] * c[n] * (Y[m,n] > 0)   (PropMLP: gradient entering the last trunk layer)
template <int BF16>
__global__ void k_rank1_mask(int M, int N, const float* __restrict__ r, const float* __restrict__ c,
                             const void* __restrict__ Y, int ldy, void* __restrict__ out, int ldo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = N / 8;
  if (i >= (size_t)M * per_row) return;
  const int m = (int)(i / per_row), n0 = (int)(i % per_row) * 8;
  float y[8], o[8];
  load8<BF16>(Y, (size_t)m * ldy + n0, y);
  const float rm = r[m];
#pragma unroll
  for (int q = 0; q < 8; ++q) o[q] = y[q] > 0.f ? rm * c[n0 + q] : 0.f;
  store8<BF16>(out, (size_t)m * ldo + n0, o);
}

// ---- view branch: per-ray bias rb[ray, j] = b[j] + sum_c enc[ray,c] * Wv[c_off + c, j] ----
// enc = [dir_enc(nd) | glo(ng)] ; Wv is the fp32 master [*, H]
__global__ void k_raybias_fwd(int nrays, int H, int nd, int ng, const float* __restrict__ dir_enc,
                              const float* __restrict__ glo, const float* __restrict__ Wv_tail,
                              const float* __restrict__ bias, float* __restrict__ rb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrays * H) return;
  const int ray = i / H, j = i % H;
  float a = bias[j];
  for (int c = 0; c < nd; ++c) a += dir_enc[ray * nd + c] * Wv_tail[(size_t)c * H + j];
  for (int c = 0; c < ng; ++c) a += glo[ray * ng + c] * Wv_tail[(size_t)(nd + c) * H + j];
  rb[i] = a;
}

// backward of the per-ray part: d_rb[ray,j] = sum_s G[ray*S+s, j];  then
//   dWv_tail[c,j] = sum_ray enc[ray,c] d_rb[ray,j];  d_glo[ray,g] = sum_j d_rb[ray,j] Wv_tail[nd+g, j]
template <int BF16>
__global__ void k_segsum(int nrays, int S, int H, const void* __restrict__ G, int ldg, float* __restrict__ d_rb) {
  // thread = (ray, 8-column chunk): 16-byte loads, H/8 lanes per row (round 4; a thread per column read 2 bytes per lane)
  const int i = blockIdx.x * blockDim.x + threadIdx.x, C = H >> 3;
  if (i >= nrays * C) return;
  const int ray = i / C, c = i % C;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int s = 0; s < S; ++s) {
    float v[8];
    load8<BF16>(G, (size_t)(ray * S + s) * ldg + c * 8, v);
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] += v[q];
  }
  float* o = d_rb + (size_t)ray * H + c * 8;
  *(float4*)o = make_float4(a[0], a[1], a[2], a[3]);
  *(float4*)(o + 4) = make_float4(a[4], a[5], a[6], a[7]);
}
#define RBW_CHUNK 512      // rays per workgroup of k_raybias_bwd_w
__global__ __launch_bounds__(1024) void k_raybias_bwd_w(int nrays, int H, int nd, int ng, const float* __restrict__ dir_enc,
                                                        const float* __restrict__ glo, const float* __restrict__ d_rb,
                                                        float* __restrict__ dWv_tail, float* __restrict__ part) {
  // workgroup (c, chunk): encoding column c, rays [chunk * RBW_CHUNK, +RBW_CHUNK): 128 j-lanes x 8 ray groups, LDS reduce in
  // fixed order (H == 128).  One chunk (<= RBW_CHUNK rays): the result goes straight to dWv_tail; more: to part[chunk][c][j],
  // summed by k_slab_reduce_small in chunk order (round 4: one workgroup per column walked ALL rays -- 1.2 ms at 16 384 rays).
  __shared__ float red[8][129];
  const int c = blockIdx.x, j = threadIdx.x & 127, rg = threadIdx.x >> 7;
  const int r0 = blockIdx.y * RBW_CHUNK, r1 = min(nrays, r0 + RBW_CHUNK);
  float a = 0.f;
  if (c < nd) for (int r = r0 + rg; r < r1; r += 8) a += dir_enc[r * nd + c] * d_rb[(size_t)r * H + j];
  else for (int r = r0 + rg; r < r1; r += 8) a += glo[r * ng + (c - nd)] * d_rb[(size_t)r * H + j];
  red[rg][j] = a;
  __syncthreads();
  if (rg == 0) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][j];
    if (gridDim.y == 1) dWv_tail[c * H + j] = t;
    else part[((size_t)blockIdx.y * gridDim.x + c) * H + j] = t;
  }
}
__global__ void k_glo_bwd(int nrays, int H, int nd, int ng, const float* __restrict__ d_rb,
                          const float* __restrict__ Wv_tail, const int* __restrict__ embed_idx,
                          float* __restrict__ d_embedding) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrays * ng) return;
  const int ray = i / ng, g = i % ng;
  float a = 0.f;
  for (int j = 0; j < H; ++j) a += d_rb[(size_t)ray * H + j] * Wv_tail[(size_t)(nd + g) * H + j];
  atomicAdd(d_embedding + (size_t)embed_idx[ray] * ng + g, a);
}
__global__ void k_glo_gather(int nrays, int ng, const float* __restrict__ embedding, const int* __restrict__ embed_idx,
                             int zero_glo, float* __restrict__ glo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrays * ng) return;
  glo[i] = zero_glo ? 0.f : embedding[(size_t)embed_idx[i / ng] * ng + (i % ng)];
}

// ---- rgb head: 16 lanes per row, lane `sub` owns the chunks of 8 consecutive k at p * 128 + sub * 8 ----
// P > 0: H = 128 P and the lane's 24 P weights stay in registers while its group walks ROWS rows (P = 1 the Mip-NeRF 360 view
// layer, P = 2 nerfacto's colour MLP: with the weights re-read from L1 for every row the pass ran at 2.6 TB/s); P = 0: any H.
template <int BF16, int P>
__global__ __launch_bounds__(256) void k_rgb_fwd(int M, int H, const void* __restrict__ Hact, int ldh,
                                                 const float* __restrict__ W /*[H,3]*/, const float* __restrict__ b,
                                                 float pad, float* __restrict__ rgb) {
  constexpr int ROWS = P > 0 ? 8 : 1;
  const int sub = threadIdx.x & 15;
  const long long row0 = (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4) * ROWS;
  if (row0 >= M) return;   // whole 16-lane groups exit together
  float w[P > 0 ? P : 1][8][3];
  if (P > 0) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = p * 128 + sub * 8 + q;
        w[p][q][0] = W[k * 3]; w[p][q][1] = W[k * 3 + 1]; w[p][q][2] = W[k * 3 + 2];
      }
  }
  const float sc = 1.f + 2.f * pad, b0 = b[0], b1 = b[1], b2 = b[2];
#pragma unroll 2
  for (int r = 0; r < ROWS; ++r) {
    const long long row = row0 + r;
    if (row >= M) break;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (P > 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float v[8];
        load8<BF16>(Hact, (size_t)row * ldh + p * 128 + sub * 8, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) { a0 += v[q] * w[p][q][0]; a1 += v[q] * w[p][q][1]; a2 += v[q] * w[p][q][2]; }
      }
    } else {
      for (int k0 = sub * 8; k0 < H; k0 += 128) {
        float v[8];
        load8<BF16>(Hact, (size_t)row * ldh + k0, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          a0 += v[q] * W[(k0 + q) * 3]; a1 += v[q] * W[(k0 + q) * 3 + 1]; a2 += v[q] * W[(k0 + q) * 3 + 2];
        }
      }
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { a0 += __shfl_xor(a0, d); a1 += __shfl_xor(a1, d); a2 += __shfl_xor(a2, d); }
    if (sub == 0) {
      rgb[row * 3] = sc / (1.f + expf(-(a0 + b0))) - pad;
      rgb[row * 3 + 1] = sc / (1.f + expf(-(a1 + b1))) - pad;
      rgb[row * 3 + 2] = sc / (1.f + expf(-(a2 + b2))) - pad;
    }
  }
}

// backward: dz[c] = d_rgb[c] * (1+2pad) * s(1-s), s = (rgb+pad)/(1+2pad)
//   G[m,j] = (h[m,j] > 0) * sum_c dz[c] W[j,c]     (gradient at the view layer's pre-activation)
//   slab[blk] accumulates dW[j,c] = sum_m h[m,j] dz[m,c] and db[c] = sum_m dz[m,c]
// H = 128 P (P = 1: the Mip-NeRF 360 view layer; P = 2: nerfacto's 256-wide colour MLP): 16 lanes per row, lane `sub` owns the
// P chunks of 8 consecutive j at p * 128 + sub * 8.
template <int BF16, int P>
__global__ __launch_bounds__(256) void k_rgb_bwd(int M, int rows_per_blk, const void* __restrict__ Hact, int ldh,
                                                 const float* __restrict__ W, const float* __restrict__ rgb,
                                                 const float* __restrict__ d_rgb, float pad, void* __restrict__ G,
                                                 int ldg, float* __restrict__ slab) {
  constexpr int H = 128 * P;
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int m0 = blockIdx.x * rows_per_blk, m1 = min(M, m0 + rows_per_blk);
  float w[P][8][3], dW[P][8][3];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = p * 128 + sub * 8 + q;
      w[p][q][0] = W[j * 3]; w[p][q][1] = W[j * 3 + 1]; w[p][q][2] = W[j * 3 + 2];
      dW[p][q][0] = dW[p][q][1] = dW[p][q][2] = 0.f;
    }
  float db[3] = {0.f, 0.f, 0.f};
  const float sc = 1.f + 2.f * pad;
  // (round 4, 16-bit activations: the next row's chunks and colours are requested before this row's arithmetic -- one row in flight
  //  per 16-lane group left the pass at 3.5 TB/s)
  uint4 nraw[P];
  float nrgb[3], ndr[3];
  auto fetch = [&](int m) {
#pragma unroll
    for (int p = 0; p < P; ++p) nraw[p] = *(const uint4*)((const uint16_t*)Hact + (size_t)m * ldh + p * 128 + sub * 8);
#pragma unroll
    for (int c = 0; c < 3; ++c) { nrgb[c] = rgb[(size_t)m * 3 + c]; ndr[c] = d_rgb[(size_t)m * 3 + c]; }
  };
  constexpr bool PF = BF16 != 0 && P == 2;      // (the 128-wide head at 131 k rows is 8 % slower with it)
  if (PF && m0 + grp < m1) fetch(m0 + grp);
  for (int m = m0 + grp; m < m1; m += 16) {
    float dz[3];
    uint4 raw[P];
    if (PF) {
#pragma unroll
      for (int p = 0; p < P; ++p) raw[p] = nraw[p];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float s = (nrgb[c] + pad) / sc;
        dz[c] = ndr[c] * sc * s * (1.f - s);
      }
      if (m + 16 < m1) fetch(m + 16);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float s = (rgb[(size_t)m * 3 + c] + pad) / sc;
        dz[c] = d_rgb[(size_t)m * 3 + c] * sc * s * (1.f - s);
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int j0 = p * 128 + sub * 8;
      float h[8], g[8];
      if (PF) decode8<BF16>(raw[p], h);
      else load8<BF16>(Hact, (size_t)m * ldh + j0, h);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        g[q] = h[q] > 0.f ? dz[0] * w[p][q][0] + dz[1] * w[p][q][1] + dz[2] * w[p][q][2] : 0.f;
        dW[p][q][0] += h[q] * dz[0]; dW[p][q][1] += h[q] * dz[1]; dW[p][q][2] += h[q] * dz[2];
      }
      store8<BF16>(G, (size_t)m * ldg + j0, g);
    }
    if (sub == 0) { db[0] += dz[0]; db[1] += dz[1]; db[2] += dz[2]; }
  }
  // reduce the 16 row-groups of the block through LDS in a fixed order
  __shared__ float red[16][H * 3 + 4];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = p * 128 + sub * 8 + q;
      red[grp][j * 3] = dW[p][q][0]; red[grp][j * 3 + 1] = dW[p][q][1]; red[grp][j * 3 + 2] = dW[p][q][2];
    }
  if (sub == 0) { red[grp][H * 3] = db[0]; red[grp][H * 3 + 1] = db[1]; red[grp][H * 3 + 2] = db[2]; }
  __syncthreads();
  for (int i = threadIdx.x; i < H * 3 + 3; i += 256) {
    float a = 0.f;
#pragma unroll
    for (int gph = 0; gph < 16; ++gph) a += red[gph][i];
    slab[(size_t)blockIdx.x * (H * 3 + 4) + i] = a;
  }
}

// ------------------------------------------------------------------------------------------------
extern "C" int hugs_density_fwd(int dtype, int M, int K, const void* Y, int ldy, const float* w, const float* b,
                                float density_bias, float* raw, float* density, void* stream) {
  HUGS_REQUIRE(K % 8 == 0, -3, "hugs_density_fwd: K=%d must be a multiple of 8", K);
  if (M <= 0) return 0;
  int lpr = 1;
  while (lpr < 64 && lpr * 8 < K) lpr <<= 1;
  const long long groups = ((long long)M * lpr + 63) / 64;                   // wave-loads of 64 / lpr rows
  const int grid = (int)std::min<long long>((groups + 7) / 8, 4096);          // 4 waves per block, 2 groups per wave iteration
  if (dtype == 2) hipLaunchKernelGGL(k_density_fwd<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, K, lpr, Y, ldy, w, b, density_bias, raw, density);
  else if (dtype) hipLaunchKernelGGL(k_density_fwd<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, K, lpr, Y, ldy, w, b, density_bias, raw, density);
  else hipLaunchKernelGGL(k_density_fwd<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, K, lpr, Y, ldy, w, b, density_bias, raw, density);
  HUGS_CHECK_LAUNCH("hugs_density_fwd");
  return 0;
}

#define WCS_BLOCKS 1024
extern "C" long long hugs_density_bwd_ws_bytes(int K) { return (long long)WCS_BLOCKS * (K + 4) * 4; }

// d_raw = d_density * sigmoid(raw + bias); dw[K] = Y^T d_raw; db = sum d_raw
extern "C" int hugs_density_bwd(int dtype, int M, int K, const void* Y, int ldy, const float* d_density, const float* raw,
                                float density_bias, float* d_raw, float* dw, float* db, void* ws, void* stream) {
  HUGS_REQUIRE(K % 8 == 0 && K <= 2048, -3, "hugs_density_bwd: K=%d unsupported", K);
  HUGS_REQUIRE(d_raw && (d_density || dw), -2, "hugs_density_bwd: d_raw and at least one of d_density / dw are required");
  if (M <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // d_density == null: d_raw is an INPUT (computed by an earlier call); dw == null: only d_raw is wanted.  The two halves can
  // then run on different streams: d_raw feeds the trunk's input gradient (critical path), the weight gradient is a 268 MB
  // pass over Y that nothing downstream waits for.
  if (d_density) hipLaunchKernelGGL(k_density_bwd_raw, dim3((M + 255) / 256), dim3(256), 0, st, M, d_density, raw, density_bias, d_raw);
  if (!dw) { HUGS_CHECK_LAUNCH("hugs_density_bwd"); return 0; }
  HUGS_REQUIRE(db && ws, -2, "hugs_density_bwd: db and ws are required with dw");
  const int rpb = (M + WCS_BLOCKS - 1) / WCS_BLOCKS;
  const int nblk = (M + rpb - 1) / rpb;
  float* slab = (float*)ws;
  if (dtype == 2) hipLaunchKernelGGL(k_wcolsum<2>, dim3(nblk), dim3(256), 0, st, M, K, rpb, Y, ldy, d_raw, slab);
  else if (dtype) hipLaunchKernelGGL(k_wcolsum<1>, dim3(nblk), dim3(256), 0, st, M, K, rpb, Y, ldy, d_raw, slab);
  else hipLaunchKernelGGL(k_wcolsum<0>, dim3(nblk), dim3(256), 0, st, M, K, rpb, Y, ldy, d_raw, slab);
  hipLaunchKernelGGL(k_slab_reduce_small, dim3((K + 1 + SRS_COLS - 1) / SRS_COLS), dim3(1024), 0, st, slab, nblk, K, K + 4, dw, 1, db);
  HUGS_CHECK_LAUNCH("hugs_density_bwd");
  return 0;
}

extern "C" int hugs_rank1_mask(int dtype, int M, int N, const float* r, const float* c, const void* Y, int ldy, void* out,
                               int ldo, void* stream) {
  HUGS_REQUIRE(N % 8 == 0, -3, "hugs_rank1_mask: N=%d must be a multiple of 8", N);
  const size_t n = (size_t)M * (N / 8);
  if (n == 0) return 0;
  if (dtype == 2) hipLaunchKernelGGL(k_rank1_mask<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, N, r, c, Y, ldy, out, ldo);
  else if (dtype) hipLaunchKernelGGL(k_rank1_mask<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, N, r, c, Y, ldy, out, ldo);
  else hipLaunchKernelGGL(k_rank1_mask<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, N, r, c, Y, ldy, out, ldo);
  HUGS_CHECK_LAUNCH("hugs_rank1_mask");
  return 0;
}

extern "C" int hugs_glo_gather(int nrays, int ng, const float* embedding, const int* embed_idx, int zero_glo, float* glo,
                               void* stream) {
  if (nrays * ng <= 0) return 0;
  hipLaunchKernelGGL(k_glo_gather, dim3((nrays * ng + 255) / 256), dim3(256), 0, (hipStream_t)stream, nrays, ng, embedding, embed_idx, zero_glo, glo);
  HUGS_CHECK_LAUNCH("hugs_glo_gather");
  return 0;
}

extern "C" int hugs_raybias_fwd(int nrays, int H, int nd, int ng, const float* dir_enc, const float* glo,
                                const float* Wv_tail, const float* bias, float* rb, void* stream) {
  if (nrays <= 0) return 0;
  hipLaunchKernelGGL(k_raybias_fwd, dim3((nrays * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, nrays, H, nd, ng, dir_enc, glo, Wv_tail, bias, rb);
  HUGS_CHECK_LAUNCH("hugs_raybias_fwd");
  return 0;
}

// G: gradient at the view layer pre-activation [nrays*S, H]. Outputs dWv_tail[(nd+ng), H] and scatter-adds into
// d_embedding (must be zeroed by the caller once per step). d_rb is workspace [hugs_raybias_bwd_ws_rows(nrays, nd, ng), H].
extern "C" long long hugs_raybias_bwd_ws_rows(int nrays, int nd, int ng) {
  const long long nchunk = ((long long)nrays + RBW_CHUNK - 1) / RBW_CHUNK;
  return (long long)nrays + (nchunk > 1 ? nchunk * (nd + ng) : 0);
}
extern "C" int hugs_raybias_bwd(int dtype, int nrays, int S, int H, int nd, int ng, const void* G, int ldg,
                                const float* dir_enc, const float* glo, const float* Wv_tail, const int* embed_idx,
                                float* d_rb, float* dWv_tail, float* d_embedding, void* stream) {
  HUGS_REQUIRE(H == 128, -3, "hugs_raybias_bwd: view width %d unsupported (128)", H);
  if (nrays <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int nseg = nrays * (H / 8);
  if (dtype == 2) hipLaunchKernelGGL(k_segsum<2>, dim3((nseg + 255) / 256), dim3(256), 0, st, nrays, S, H, G, ldg, d_rb);
  else if (dtype) hipLaunchKernelGGL(k_segsum<1>, dim3((nseg + 255) / 256), dim3(256), 0, st, nrays, S, H, G, ldg, d_rb);
  else hipLaunchKernelGGL(k_segsum<0>, dim3((nseg + 255) / 256), dim3(256), 0, st, nrays, S, H, G, ldg, d_rb);
  // ray chunks: their partial sums live behind the [nrays, H] ray sums in d_rb (hugs_raybias_bwd_ws_rows rows in all)
  const int nchunk = (nrays + RBW_CHUNK - 1) / RBW_CHUNK;
  float* part = nchunk > 1 ? d_rb + (size_t)nrays * H : nullptr;
  hipLaunchKernelGGL(k_raybias_bwd_w, dim3(nd + ng, nchunk), dim3(1024), 0, st, nrays, H, nd, ng, dir_enc, glo, d_rb, dWv_tail, part);
  if (nchunk > 1)
    hipLaunchKernelGGL(k_slab_reduce_small, dim3(((nd + ng) * H + SRS_COLS - 1) / SRS_COLS), dim3(1024), 0, st, part, nchunk, (nd + ng) * H, (nd + ng) * H, dWv_tail);
  if (ng > 0 && d_embedding)
    hipLaunchKernelGGL(k_glo_bwd, dim3((nrays * ng + 255) / 256), dim3(256), 0, st, nrays, H, nd, ng, d_rb, Wv_tail, embed_idx, d_embedding);
  HUGS_CHECK_LAUNCH("hugs_raybias_bwd");
  return 0;
}

extern "C" int hugs_rgb_fwd(int dtype, int M, int H, const void* Hact, int ldh, const float* W, const float* b, float pad,
                            float* rgb, void* stream) {
  HUGS_REQUIRE(H % 8 == 0, -3, "hugs_rgb_fwd: H=%d must be a multiple of 8", H);
  if (M <= 0) return 0;
  const int P = H == 128 ? 1 : H == 256 ? 2 : 0;
  const long long groups = P ? ((long long)M + 7) / 8 : (long long)M;
  const int grid = (int)((groups * 16 + 255) / 256);
#define RGB_FWD(DT_, P_) hipLaunchKernelGGL((k_rgb_fwd<DT_, P_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, M, H, Hact, ldh, W, b, pad, rgb)
#define RGB_FWD_DT(P_) do { if (dtype == 2) RGB_FWD(2, P_); else if (dtype) RGB_FWD(1, P_); else RGB_FWD(0, P_); } while (0)
  if (P == 1) RGB_FWD_DT(1); else if (P == 2) RGB_FWD_DT(2); else RGB_FWD_DT(0);
#undef RGB_FWD_DT
#undef RGB_FWD
  HUGS_CHECK_LAUNCH("hugs_rgb_fwd");
  return 0;
}

#define RGB_BLOCKS 1024
/* workspace of hugs_rgb_bwd for the widest head it takes (H = 256) */
extern "C" long long hugs_rgb_bwd_ws_bytes(void) { return (long long)RGB_BLOCKS * (256 * 3 + 4) * 4; }

extern "C" int hugs_rgb_bwd(int dtype, int M, int H, const void* Hact, int ldh, const float* W, const float* rgb,
                            const float* d_rgb, float pad, void* G, int ldg, float* dW, float* db, void* ws, void* stream) {
  HUGS_REQUIRE(H > 0 && H % 128 == 0, -3, "hugs_rgb_bwd: head width %d unsupported (a multiple of 128)", H);
  HUGS_REQUIRE(dW || H <= 256, -3, "hugs_rgb_bwd: dW == NULL (reduction deferred to hugs_rgb_bwd_reduce) needs a single column slab, H = %d", H);
  if (M <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int rpb = (M + RGB_BLOCKS - 1) / RGB_BLOCKS;
  rpb = (rpb + 15) / 16 * 16;
  const int nblk = (M + rpb - 1) / rpb;
  float* slab = (float*)ws;
  // wider heads (use_viewdirs=False puts the rgb layer on the trunk: models.py:486-516) go in column slabs of 256 / 128: the
  // pre-activation gradient dz is recomputed per slab (6 floats per row), everything else is per column
  const int step = H % 256 == 0 ? 256 : 128;
  const size_t esz = dtype ? 2 : 4;
  for (int c0 = 0; c0 < H; c0 += step) {
    const void* Hc = (const char*)Hact + (size_t)c0 * esz;
    void* Gc = (char*)G + (size_t)c0 * esz;
    const float* Wc = W + (size_t)c0 * 3;
#define RGB_BWD(DT_, P_) hipLaunchKernelGGL((k_rgb_bwd<DT_, P_>), dim3(nblk), dim3(256), 0, st, M, rpb, Hc, ldh, Wc, rgb, d_rgb, pad, Gc, ldg, slab)
    if (step == 128) { if (dtype == 2) RGB_BWD(2, 1); else if (dtype) RGB_BWD(1, 1); else RGB_BWD(0, 1); }
    else { if (dtype == 2) RGB_BWD(2, 2); else if (dtype) RGB_BWD(1, 2); else RGB_BWD(0, 2); }
#undef RGB_BWD
    // (the bias gradient is the same in every slab: written by the first)
    if (dW)
      hipLaunchKernelGGL(k_slab_reduce_small, dim3((step * 3 + 3 + SRS_COLS - 1) / SRS_COLS), dim3(1024), 0, st, slab, nblk, step * 3, step * 3 + 4,
                         dW + (size_t)c0 * 3, c0 == 0 ? 3 : 0, c0 == 0 ? db : nullptr);
  }
  HUGS_CHECK_LAUNCH("hugs_rgb_bwd");
  return 0;
}

// The second half of hugs_rgb_bwd called with dW == NULL (H <= 256: one column slab, its partial sums stay in ws): dW [H,3], db [3] from
// the workspace.  Round 5: the 387-column reduction of the view head sat between rgb_bwd and the G_last product on the step's critical
// chain -- 5 us alone, 54 us next to the proposal level's persistent backward kernel, whose workgroups it waited for; nothing before the
// gradient exchange reads dW, so the step launches it on the head weight-gradient stream.
extern "C" int hugs_rgb_bwd_reduce(int M, int H, float* dW, float* db, const void* ws, void* stream) {
  HUGS_REQUIRE(H > 0 && H % 128 == 0 && H <= 256 && dW && db && ws, -3, "hugs_rgb_bwd_reduce: head width %d (128 or 256), non-null dW / db / ws", H);
  if (M <= 0) return 0;
  int rpb = (M + RGB_BLOCKS - 1) / RGB_BLOCKS;
  rpb = (rpb + 15) / 16 * 16;
  const int nblk = (M + rpb - 1) / rpb;
  hipLaunchKernelGGL(k_slab_reduce_small, dim3((H * 3 + 3 + SRS_COLS - 1) / SRS_COLS), dim3(1024), 0, (hipStream_t)stream, (const float*)ws, nblk, H * 3,
                     H * 3 + 4, dW, 3, db);
  HUGS_CHECK_LAUNCH("hugs_rgb_bwd_reduce");
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Round 5: the option branches' element-wise steps as kernels (they were torch ops in engine.py -- per-sample arithmetic outside the
// library and outside a captured step).
// ------------------------------------------------------------------------------------------------
// models.py:458-460,467: raw += noise_scale * noise (noise null: nothing added) ; density = softplus(raw + density_bias)
__global__ void k_noise_softplus(long long n, long long n_noise, float* __restrict__ raw, const float* __restrict__ noise, float scale,
                                 float density_bias, float* __restrict__ density) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r = raw[i];
  if (noise && i < n_noise) { r = fmaf(scale, noise[i], r); raw[i] = r; }
  density[i] = softplusf(r + density_bias);
}
// y (compute dtype) += a * x (fp32)            (models.py:478-481 bottleneck noise)
template <int DT>
__global__ void k_axpy_op(long long n, float a, const float* __restrict__ x, void* __restrict__ y) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (DT) { uint16_t* p = (uint16_t*)y + i; *p = f_to_op16(fmaf(a, x[i], op16_to_f(*p, DT)), DT); }
  else ((float*)y)[i] = fmaf(a, x[i], ((float*)y)[i]);
}
// dst (compute dtype) += src (compute dtype)
template <int DT>
__global__ void k_add_op(long long n, const void* __restrict__ src, void* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (DT) { uint16_t* p = (uint16_t*)dst + i; *p = f_to_op16(op16_to_f(*p, DT) + op16_to_f(((const uint16_t*)src)[i], DT), DT); }
  else ((float*)dst)[i] += ((const float*)src)[i];
}
// dst = a * src + b (fp32; src may be dst)      (rgb_premultiplier / rgb_bias folded into the rgb head's weights, models.py:514-516)
__global__ void k_affine(long long n, const float* __restrict__ src, float a, float b, float* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = fmaf(a, src[i], b);
}
// render.py:219-221 with a per-ray, per-channel background (models.py:256-261): bgw = max(0, 1 - sum_s w), rgb_out += bgw * bg
__global__ void k_bg_blend_fwd(int N, int S, const float* __restrict__ w, const float* __restrict__ bg, float* __restrict__ rgb_out,
                               float* __restrict__ bgw) {
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ray >= N) return;
  float a = 0.f;
  for (int s_ = lane; s_ < S; s_ += 64) a += w[(size_t)ray * S + s_];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
  const float b = fmaxf(0.f, 1.f - a);
  if (lane == 0) bgw[ray] = b;
  if (lane < 3) rgb_out[ray * 3 + lane] += b * bg[ray * 3 + lane];
}
// its backward: d/dw_s of bgw * bg = -(bg . d_rgb_out) where 1 - sum w > 0, the same for every sample of the ray, added to d_w_extra
__global__ void k_bg_blend_bwd(int N, int S, const float* __restrict__ d_rgb_out, const float* __restrict__ bg, const float* __restrict__ bgw,
                               const float* __restrict__ d_w_extra, float* __restrict__ d_w_total) {
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ray >= N) return;
  const float dacc = bgw[ray] > 0.f ? -(d_rgb_out[ray * 3] * bg[ray * 3] + d_rgb_out[ray * 3 + 1] * bg[ray * 3 + 1] + d_rgb_out[ray * 3 + 2] * bg[ray * 3 + 2]) : 0.f;
  for (int s_ = lane; s_ < S; s_ += 64) d_w_total[(size_t)ray * S + s_] = (d_w_extra ? d_w_extra[(size_t)ray * S + s_] : 0.f) + dacc;
}

#define HH_1D(kern, n_, ...) do { const long long nn_ = (n_); if (nn_ > 0) hipLaunchKernelGGL(kern, dim3((unsigned)((nn_ + 255) / 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); } while (0)
extern "C" int hugs_noise_softplus(long long n, long long n_noise, float* raw, const float* noise, float noise_scale, float density_bias,
                                   float* density, void* stream) {
  HUGS_REQUIRE(raw && density && n_noise <= n, -2, "hugs_noise_softplus: null pointer / more noise values than rows");
  HH_1D(k_noise_softplus, n, n, n_noise, raw, noise, noise_scale, density_bias, density);
  HUGS_CHECK_LAUNCH("hugs_noise_softplus");
  return 0;
}
extern "C" int hugs_axpy_op(int dtype, long long n, float a, const float* x, void* y, void* stream) {
  HUGS_REQUIRE(dtype >= 0 && dtype <= 2 && (n == 0 || (x && y)), -2, "hugs_axpy_op: dtype %d / null pointer", dtype);
  if (dtype == 2) HH_1D(k_axpy_op<2>, n, n, a, x, y); else if (dtype) HH_1D(k_axpy_op<1>, n, n, a, x, y); else HH_1D(k_axpy_op<0>, n, n, a, x, y);
  HUGS_CHECK_LAUNCH("hugs_axpy_op");
  return 0;
}
extern "C" int hugs_add_op(int dtype, long long n, const void* src, void* dst, void* stream) {
  HUGS_REQUIRE(dtype >= 0 && dtype <= 2 && (n == 0 || (src && dst)), -2, "hugs_add_op: dtype %d / null pointer", dtype);
  if (dtype == 2) HH_1D(k_add_op<2>, n, n, src, dst); else if (dtype) HH_1D(k_add_op<1>, n, n, src, dst); else HH_1D(k_add_op<0>, n, n, src, dst);
  HUGS_CHECK_LAUNCH("hugs_add_op");
  return 0;
}
extern "C" int hugs_affine(long long n, const float* src, float a, float b, float* dst, void* stream) {
  HUGS_REQUIRE(n == 0 || (src && dst), -2, "hugs_affine: null pointer");
  HH_1D(k_affine, n, n, src, a, b, dst);
  HUGS_CHECK_LAUNCH("hugs_affine");
  return 0;
}
extern "C" int hugs_bg_blend_fwd(int N, int S, const float* w, const float* bg_rgb, float* rgb_out, float* bgw, void* stream) {
  HUGS_REQUIRE(N >= 0 && S > 0 && w && bg_rgb && rgb_out && bgw, -2, "hugs_bg_blend_fwd: null pointer");
  if (N > 0) hipLaunchKernelGGL(k_bg_blend_fwd, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, S, w, bg_rgb, rgb_out, bgw);
  HUGS_CHECK_LAUNCH("hugs_bg_blend_fwd");
  return 0;
}
extern "C" int hugs_bg_blend_bwd(int N, int S, const float* d_rgb_out, const float* bg_rgb, const float* bgw, const float* d_w_extra,
                                 float* d_w_total, void* stream) {
  HUGS_REQUIRE(N >= 0 && S > 0 && d_rgb_out && bg_rgb && bgw && d_w_total, -2, "hugs_bg_blend_bwd: null pointer");
  if (N > 0) hipLaunchKernelGGL(k_bg_blend_bwd, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, S, d_rgb_out, bg_rgb, bgw, d_w_extra, d_w_total);
  HUGS_CHECK_LAUNCH("hugs_bg_blend_bwd");
  return 0;
}

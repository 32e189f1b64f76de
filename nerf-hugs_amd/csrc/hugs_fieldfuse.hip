// Fused nerfacto field (round 4): the base MLP and the colour MLP of models/nerfacto.py:693-759 (HashMLPDensityField-style base
// network: hash features -> Linear(256) -> ReLU -> Linear(1 + geo_feat_dim); colour network: [SH(dir) | geo features | appearance
// embedding] -> Linear(256) -> ReLU -> Linear(256) -> ReLU -> Linear(3) -> sigmoid), run per 64-sample tile with the activations
// kept in LDS from layer to layer.  As separate GEMM launches every one of these 256-wide layers is HBM-bound (0.5 KB in + 0.5 KB
// out per sample and layer, the narrow sides padded to 128 columns: profiles/r04_cfg5_bench.json, 0.55-0.70 of 8 TB/s); here
//   forward : reads the 32 hash features of a sample (64 B), writes each activation the weight-gradient GEMMs need exactly once
//             (Y0, head input, H0, H1) + 1-bit relu masks + density + rgb; nothing is read back;
//   backward: (k_field_bwd) reads H1 / the masks / d_rgb / d_density, writes each layer's output gradient exactly once (the G
//             operands of the weight-gradient GEMMs) + the 32 feature gradients; the appearance-embedding and rgb-layer
//             gradients are reduced in the kernel.
// Geometry (both kernels): 64-row tile, 4 waves, two workgroups per CU (70 KiB of LDS each: the epilogue of one overlaps the MFMAs
// of the other).  Wave wn owns output columns [wn * 16 NJ, (wn + 1) * 16 NJ) of all 64 rows: NJ x 4 fragments of
// v_mfma_f32_16x16x32 with the WEIGHTS as the A operand, so a lane holds 4 consecutive output columns of one row
// (hugs_gemm.hip's convention: the K-stage LDS layout, its XOR swizzle and the mask-bit layout are that file's).  A layer's
// weights stream from L2 straight into registers, three K-stages ahead; the first three stages of the NEXT layer are requested
// before the current layer's epilogue.  Compiled for both 16-bit operand formats (dtype 1 = bf16, 2 = IEEE half).
#include "hugs_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float ff_f32x4_t;
typedef float __attribute__((ext_vector_type(2))) ff_f32x2_t;

template <int F16> struct FfOps;
template <> struct FfOps<0> {
  typedef __attribute__((ext_vector_type(8))) __bf16 x8_t;
  typedef __bf16 __attribute__((ext_vector_type(2))) x2_t;
  static __device__ __forceinline__ ff_f32x4_t mfma(x8_t a, x8_t b, ff_f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float lo(uint32_t u) { return __uint_as_float(u << 16); }
  static __device__ __forceinline__ float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
};
template <> struct FfOps<1> {
  typedef __attribute__((ext_vector_type(8))) _Float16 x8_t;
  typedef _Float16 __attribute__((ext_vector_type(2))) x2_t;
  static __device__ __forceinline__ ff_f32x4_t mfma(x8_t a, x8_t b, ff_f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
  static __device__ __forceinline__ float hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
};
template <int F16> __device__ __forceinline__ uint32_t ff_cvt_pk(float a, float b) {       // one v_cvt_pk_{bf16,f16}_f32 (RNE)
  const ff_f32x2_t f = {a, b};
  const typename FfOps<F16>::x2_t h = __builtin_convertvector(f, typename FfOps<F16>::x2_t);
  return *(const uint32_t*)&h;
}

// A lane offset the optimiser may not treat as loop-invariant: with the whole tile as straight-line code, every (uniform base +
// lane offset) pair of its ~150 loads and stores was turned into a 64-bit per-lane address, hoisted out of the tile loop and
// spilled (264 B of scratch per lane); an opaque copy per use keeps them as one 32-bit VGPR next to a scalar base.
__device__ __forceinline__ unsigned ff_fresh(unsigned x) { asm volatile("" : "+v"(x)); return x; }

// x summed over the lanes r16 + {0, 16, 32, 48} (every lane gets the sum)
__device__ __forceinline__ float ff_sum_kb(float x) {
  typedef unsigned __attribute__((ext_vector_type(2))) u2;
  unsigned u = __float_as_uint(x);
  const u2 s = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(s[0]) + __uint_as_float(s[1]));
  const u2 t = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

#define FF_ROWS 64
#define FF_STAGE (FF_ROWS * 64)        // one K-stage: 64 rows x 32 k x 2 B
#define FF_ACT (8 * FF_STAGE)          // a 256-wide activation tile

// byte offset of element (row, col) of an activation tile in the K-stage layout
__device__ __forceinline__ int ff_at(int row, int col) {
  return (col >> 5) * FF_STAGE + row * 64 + ((((col & 31) >> 3) ^ (3 * ((row >> 2) & 1))) << 4) + (col & 7) * 2;
}

// the first min(3, KST) K-stages of a layer's weights -> the register ring.  W = the matrix (uniform), wo = the lane's BYTE
// offset: row (col0 + r16), k offset kb * 8 (32-bit lane offsets on scalar bases: 64-bit per-lane addresses of every load and
// store of the straight-line tile were hoisted out of the tile loop and spilled)
template <int F16, int KST, int NJ>
__device__ __forceinline__ void ff_prefetch(const uint16_t* W, unsigned wo, int ldw, typename FfOps<F16>::x8_t (&wq)[4][4]) {
  typedef typename FfOps<F16>::x8_t x8_t;
  wo = ff_fresh(wo);
#pragma unroll
  for (int s = 0; s < (KST < 3 ? KST : 3); ++s)
#pragma unroll
    for (int j = 0; j < NJ; ++j) wq[s][j] = *(const x8_t*)((const char*)W + (wo + (unsigned)(j * 16 * ldw + s * 32) * 2u));
}

// acc[i][j] += sum over KST stages of W-fragment(j) x X-fragment(i); A = the activation tile (stage s0 first)
template <int F16, int KST, int NJ>
__device__ __forceinline__ void ff_mma(const unsigned char* A, int frag_off, const uint16_t* W, unsigned wo, int ldw,
                                       typename FfOps<F16>::x8_t (&wq)[4][4], ff_f32x4_t (&acc)[4][4]) {
  typedef typename FfOps<F16>::x8_t x8_t;
  wo = ff_fresh(wo);
#ifdef FF_NOMMA      // (timing builds, scratch/r4_ffuse_var.sh: -DFF_NOMMA / FF_NOSTORE / FF_NOWLOAD)
  return;
#endif
#pragma unroll
  for (int s = 0; s < KST; ++s) {
#ifndef FF_NOWLOAD
    if (s + 3 < KST)
#else
    if (false)
#endif
    {
#pragma unroll
      for (int j = 0; j < NJ; ++j) wq[(s + 3) & 3][j] = *(const x8_t*)((const char*)W + (wo + (unsigned)(j * 16 * ldw + (s + 3) * 32) * 2u));
    }
    x8_t xa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = *(const x8_t*)(A + s * FF_STAGE + frag_off + i * 16 * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = FfOps<F16>::mfma(wq[s & 3][j], xa[i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);      // (keeps the weight loads three stages ahead, not eight: the straight-line tile spilled)
  }
}

template <int NJ>
__device__ __forceinline__ void ff_acc_bias(ff_f32x4_t (&acc)[4][4], const float* b /* + col0 + kb*4 */) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float4 bb = *(const float4*)(b + j * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] = ff_f32x4_t{bb.x, bb.y, bb.z, bb.w};
  }
}
template <int NJ>
__device__ __forceinline__ void ff_acc_zero(ff_f32x4_t (&acc)[4][4]) {
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] = ff_f32x4_t{0.f, 0.f, 0.f, 0.f};
}

struct FieldFwd {
  int M, S, ldx0, ldw0, ngeo;
  const uint16_t* X0;                       // [M, ldx0] hash features, columns 0..31 real
  const uint16_t *W0t, *W1t, *C0t, *C1t;    // [256][ldw0], [128][256], [256][128], [256][256]  ([n][k], 16-bit)
  const float *b0, *b1, *cb0, *cb1;         // [256], [128], [256], [256]
  const float* c2;                          // [256, 3] fp32 (the rgb layer), cb2: 3 floats (rgb_bias folded in)
  const float* cb2;
  const uint16_t* tmpl;                     // [M / S, 128] per-ray head-input template (hugs_nf_head_template)
  const float* sel;                         // [M]
  uint16_t *Y0, *raw, *Xh, *H0, *H1;        // [M,256], [M] (16-bit raw density), [M,128], [M,256], [M,256]
  uint32_t *bY0, *bH0;                      // 1-bit relu masks (M * 256 / 8 bytes each) or null
  float* density;                           // [M]
  float* rgb;                               // [M, 3]
};

// relu + 16-bit rounding + {HBM row, next layer's LDS tile, mask bits} of one 64 x 64 wave block (NJ = 4)
template <int F16, bool RELU, bool KEEP>
__device__ __forceinline__ void ff_emit256(ff_f32x4_t (&acc)[4][4], int m0, int wn, int r16, int kb, int lane, uint16_t* Y /* [M,256] */,
                                           unsigned char* An, uint32_t* bout, uint32_t (&keep)[4][4][2]) {
  const int swz = 3 * ((r16 >> 2) & 1);
  uint16_t* Yout = Y ? Y + (size_t)m0 * 256 : nullptr;                    // (uniform)
  const unsigned yo = ff_fresh((unsigned)(r16 * 256 + wn * 64 + kb * 4) * 2u);      // (bytes)
  const unsigned bo = ff_fresh((unsigned)lane * 4u);
  // hugs_gemm.hip nt_epilogue_direct bit layout: NT tile = 256 rows; its wave (wm_nt, wn) covers 128 rows = fragment rows i_nt 0..7
  const int i_nt0 = ((m0 >> 6) & 1) * 4;
  uint32_t* btile = bout ? bout + ((size_t)(m0 >> 8) * 8 + (size_t)(((m0 >> 7) & 1) * 4 + wn)) * 256 : nullptr;      // (uniform)
  uint32_t bw = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0 = acc[i][j][0], x1 = acc[i][j][1], x2 = acc[i][j][2], x3 = acc[i][j][3];
      if (RELU) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); x2 = fmaxf(x2, 0.f); x3 = fmaxf(x3, 0.f); }
      uint2 u;
      u.x = ff_cvt_pk<F16>(x0, x1); u.y = ff_cvt_pk<F16>(x2, x3);
      if (KEEP) { keep[i][j][0] = u.x; keep[i][j][1] = u.y; }
#ifndef FF_NOSTORE
      if (Yout) *(uint2*)((char*)Yout + (yo + (unsigned)(i * 16 * 256 + j * 16) * 2u)) = u;
#endif
      if (An) {
        const int st = wn * 2 + (j >> 1), ch = (j & 1) * 2 + (kb >> 1);
        *(uint2*)(An + st * FF_STAGE + (i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
      }
      if (bout) {
        const int k = (i & 1) * 8 + j * 2;
        bw |= ((u.x + 0x7fff7fffu) >> (15 - k)) & (0x00010001u << k);
        bw |= ((u.y + 0x7fff7fffu) >> (14 - k)) & (0x00010001u << (k + 1));
      }
    }
    if (bout && (i & 1)) { *(uint32_t*)((char*)btile + (bo + (unsigned)(((i_nt0 + i) >> 1) * 64) * 4u)) = bw; bw = 0u; }
  }
}

template <int F16>
__global__ __launch_bounds__(256, 2) void k_field_fwd(const FieldFwd P) {
  typedef typename FfOps<F16>::x8_t x8_t;
  __shared__ __attribute__((aligned(16))) unsigned char act[2][FF_ACT];
  __shared__ float red[FF_ROWS][4][3];
  __shared__ __attribute__((aligned(16))) float c2s[256 * 3 + 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kb = lane >> 4;
  const int swz = 3 * ((r16 >> 2) & 1);
  const int frag_off = r16 * 64 + ((kb ^ swz) << 4);
  const int ntile = P.M / FF_ROWS;
  for (int e = tid; e < 256 * 3; e += 256) c2s[e] = P.c2[e];
  if (tid < 3) c2s[768 + tid] = P.cb2[tid];
  // lane bases of the four weight matrices
  const unsigned w0o = (unsigned)((wn * 64 + r16) * P.ldw0 + kb * 8) * 2u, w1o = (unsigned)((wn * 32 + r16) * 256 + kb * 8) * 2u;      // bytes
  const unsigned c0o = (unsigned)((wn * 64 + r16) * 128 + kb * 8) * 2u, c1o = (unsigned)((wn * 64 + r16) * 256 + kb * 8) * 2u;
  const bool l1_live = wn * 32 < 16 + P.ngeo;      // (wave-uniform) this wave's 32 columns of layer 1 hold real outputs

#ifdef FF_STAGGER      // (timing experiment: the second workgroup of a CU starts ~half a tile late)
  if (blockIdx.x >= gridDim.x / 2)
    for (int q = 0; q < FF_STAGGER; ++q) __builtin_amdgcn_s_sleep(127);
#endif
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int m0 = t * FF_ROWS;
    x8_t wq[4][4];
    ff_f32x4_t acc[4][4];
    uint32_t keep[4][4][2];
    ff_prefetch<F16, 1, 4>(P.W0t, w0o, P.ldw0, wq);
    __syncthreads();      // the previous tile's readers of act[] / red are done (first tile: c2s is published)
    {
      // hash features -> act[0] stage 7 (the head input assembled below uses stages 0..3 of the same buffer)
      const int row = tid >> 2, cc = tid & 3;
      const uint4 v = *(const uint4*)((const char*)(P.X0 + (size_t)m0 * P.ldx0) + ff_fresh((unsigned)(row * P.ldx0 + cc * 8) * 2u));
      *(uint4*)(act[0] + 7 * FF_STAGE + row * 64 + ((cc ^ (3 * ((row >> 2) & 1))) << 4)) = v;
      // head-input template of the row's ray (hugs_nf_head_template: [SH16 | 0 (geo) | appearance | 0], 128 columns, 16-bit) ->
      // stages 0..3; the geo columns are written by layer 1's epilogue, two barriers later.  thread = (row, 32-column stage cc)
      const unsigned ray = (unsigned)(m0 + row) / (unsigned)P.S;
      const char* tp = (const char*)P.tmpl + ff_fresh(ray * 256u + (unsigned)cc * 64u);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(uint4*)(act[0] + cc * FF_STAGE + row * 64 + ((q ^ (3 * ((row >> 2) & 1))) << 4)) = *(const uint4*)(tp + q * 16);
    }
    __syncthreads();
    // ---- base layer 0: 32 -> 256, relu ------------------------------------------------------------------------------
    ff_acc_bias<4>(acc, P.b0 + wn * 64 + kb * 4);
    ff_mma<F16, 1, 4>(act[0] + 7 * FF_STAGE, frag_off, P.W0t, w0o, P.ldw0, wq, acc);
    if (l1_live) ff_prefetch<F16, 8, 2>(P.W1t, w1o, 256, wq);
    ff_emit256<F16, true, false>(acc, m0, wn, r16, kb, lane, P.Y0, act[1], P.bY0, keep);
    __syncthreads();
    // ---- base layer 1: 256 -> 1 + ngeo (no activation), computed in HEAD-INPUT column order (W1t / b1 rows: 0 = raw density,
    // 16 .. 16 + ngeo = the geo features, the rest zero): the geo block lands in the head-input tile as aligned 8-byte writes
    if (l1_live) {
      ff_acc_bias<2>(acc, P.b1 + wn * 32 + kb * 4);
      ff_mma<F16, 8, 2>(act[1], frag_off, P.W1t, w1o, 256, wq, acc);
    }
    ff_prefetch<F16, 4, 4>(P.C0t, c0o, 128, wq);
    if (l1_live) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = wn * 32 + j * 16 + kb * 4;
        const bool geo = n >= 16 && n < 16 + P.ngeo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 16 + r16;
          uint2 u;
          u.x = ff_cvt_pk<F16>(acc[i][j][0], acc[i][j][1]); u.y = ff_cvt_pk<F16>(acc[i][j][2], acc[i][j][3]);
          if (geo) *(uint2*)(act[0] + ((n >> 5) * FF_STAGE + row * 64 + ((((n & 31) >> 3) ^ swz) << 4) + (n & 4) * 2)) = u;
          if (n == 0) {
            *(uint16_t*)((char*)(P.raw + m0) + ff_fresh((unsigned)row * 2u)) = (uint16_t)u.x;
            const unsigned ro = ff_fresh((unsigned)row * 4u);
            *(float*)((char*)(P.density + m0) + ro) = expf(FfOps<F16>::lo(u.x)) * *(const float*)((const char*)(P.sel + m0) + ro);
          }
        }
      }
    }
    __syncthreads();
    // ---- head input -> HBM (the colour network's first weight gradient reads it); colour layer 0: 128 -> 256, relu ------
    {
      const int row = tid >> 2, st = tid & 3;
      const unsigned char* src = act[0] + st * FF_STAGE + row * 64;
      char* dst = (char*)(P.Xh + (size_t)m0 * 128) + ff_fresh((unsigned)(row * 128 + st * 32) * 2u);
      const int sw = 3 * ((row >> 2) & 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) *(uint4*)(dst + q * 16) = *(const uint4*)(src + ((q ^ sw) << 4));
    }
    ff_acc_bias<4>(acc, P.cb0 + wn * 64 + kb * 4);
    ff_mma<F16, 4, 4>(act[0], frag_off, P.C0t, c0o, 128, wq, acc);
    ff_prefetch<F16, 8, 4>(P.C1t, c1o, 256, wq);
    ff_emit256<F16, true, false>(acc, m0, wn, r16, kb, lane, P.H0, act[1], P.bH0, keep);
    __syncthreads();
    // ---- colour layer 1: 256 -> 256, relu; rgb = sigmoid(H1 c2 + cb2) on the rounded activations ---------------------------
    ff_acc_bias<4>(acc, P.cb1 + wn * 64 + kb * 4);
    ff_mma<F16, 8, 4>(act[1], frag_off, P.C1t, c1o, 256, wq, acc);
    ff_emit256<F16, true, true>(acc, m0, wn, r16, kb, lane, P.H1, nullptr, nullptr, keep);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* w = c2s + (wn * 64 + j * 16 + kb * 4) * 3;
        const float4 w0 = *(const float4*)w, w1 = *(const float4*)(w + 4), w2 = *(const float4*)(w + 8);
        const float v0 = FfOps<F16>::lo(keep[i][j][0]), v1 = FfOps<F16>::hi(keep[i][j][0]);
        const float v2 = FfOps<F16>::lo(keep[i][j][1]), v3 = FfOps<F16>::hi(keep[i][j][1]);
        a0 += v0 * w0.x + v1 * w0.w + v2 * w1.z + v3 * w2.y;
        a1 += v0 * w0.y + v1 * w1.x + v2 * w1.w + v3 * w2.z;
        a2 += v0 * w0.z + v1 * w1.y + v2 * w2.x + v3 * w2.w;
      }
      // sum over the four kb lane groups: v_permlane16_swap / v_permlane32_swap of a value with itself leave (x, neighbour's x) in
      // the two results.  (__shfl_xor's ds_bpermute_b32 returned a stale FIRST operand here now and then -- the low half of the
      // v_pk_add_f32 pair the compiler forms from a0 / a1 -- on ~0.7 % of the rows, run-to-run different: scratch/ffuse_bench.py.)
      a0 = ff_sum_kb(a0); a1 = ff_sum_kb(a1); a2 = ff_sum_kb(a2);
      if (kb == 0) { red[i * 16 + r16][wn][0] = a0; red[i * 16 + r16][wn][1] = a1; red[i * 16 + r16][wn][2] = a2; }
    }
    __syncthreads();
    if (tid < FF_ROWS * 3) {
      const int row = tid / 3, c = tid - row * 3;
      const float a = ((red[row][0][c] + red[row][1][c]) + (red[row][2][c] + red[row][3][c])) + c2s[768 + c];

      *(float*)((char*)(P.rgb + (size_t)m0 * 3) + ff_fresh((unsigned)tid * 4u)) = 1.f / (1.f + expf(-a));
    }
  }
}


// per-ray head-input template: out[ray, 0..127] = [SH16 | 0 x ngeo | appearance (napp) | 0 ...] in the 16-bit operand format
__global__ void k_head_template(int nrays, int f16, const float* __restrict__ sh, const float* __restrict__ app, int ngeo, int napp,
                                uint16_t* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nrays * 128) return;
  const int ray = e >> 7, c = e & 127, a0 = 16 + ngeo;
  const float v = c < 16 ? sh[(size_t)ray * 16 + c] : (c >= a0 && c < a0 + napp) ? app[(size_t)ray * napp + (c - a0)] : 0.f;
  out[e] = f_to_op16(v, f16 ? 2 : 1);
}

}  // namespace

// include/hugs.h hugs_nf_head_template
extern "C" int hugs_nf_head_template(int dtype, int nrays, const float* sh, const float* app, int ngeo, int napp, void* out, void* stream) {
  HUGS_REQUIRE(dtype == 1 || dtype == 2, -2, "hugs_nf_head_template: 16-bit operands only (dtype 1 = bf16, 2 = half), got %d", dtype);
  HUGS_REQUIRE(nrays >= 0 && nrays < (1 << 24) && ngeo >= 0 && napp >= 0 && 16 + ngeo + napp <= 128 && sh && out && (napp == 0 || app), -3,
               "hugs_nf_head_template: %d rays, %d geo / %d appearance columns (16 + ngeo + napp <= 128)", nrays, ngeo, napp);
  if (nrays > 0)
    hipLaunchKernelGGL(k_head_template, dim3((nrays * 128 + 255) / 256), dim3(256), 0, (hipStream_t)stream, nrays, dtype == 2, sh, app, ngeo, napp, (uint16_t*)out);
  HUGS_CHECK_LAUNCH("hugs_nf_head_template");
  return 0;
}

// include/hugs.h hugs_nf_field_fwd
extern "C" int hugs_nf_field_fwd(int dtype, long long M, int S, const void* X0, int ldx0, const void* W0t, int ldw0, const void* W1x,
                                 const void* C0t, const void* C1t, const float* b0, const float* b1, const float* cb0,
                                 const float* cb1, const float* c2, const float* cb2, const void* tmpl, int ngeo, const float* sel,
                                 void* Y0, void* raw, void* Xh, void* H0, void* H1, uint32_t* bY0, uint32_t* bH0, float* density,
                                 float* rgb, void* stream) {
  HUGS_REQUIRE(dtype == 1 || dtype == 2, -2, "hugs_nf_field_fwd: 16-bit operands only (dtype 1 = bf16, 2 = half), got %d", dtype);
  HUGS_REQUIRE(M > 0 && M % 256 == 0 && M < (1ll << 31) && S > 0 && M % S == 0, -3,
               "hugs_nf_field_fwd: %lld rows (a positive multiple of 256, whole rays of %d samples)", M, S);
  HUGS_REQUIRE(ldx0 >= 32 && ldx0 % 8 == 0 && ldw0 >= 32 && ldw0 % 8 == 0 && ldx0 <= 4096 && ldw0 <= 4096, -3,
               "hugs_nf_field_fwd: pitches %d / %d (32 .. 4096, multiples of 8)", ldx0, ldw0);
  HUGS_REQUIRE(ngeo >= 0 && 16 + ngeo <= 128 && ngeo % 4 == 0, -3,
               "hugs_nf_field_fwd: %d geo features (a multiple of 4, 16 + ngeo <= 128)", ngeo);
  HUGS_REQUIRE(X0 && W0t && W1x && C0t && C1t && b0 && b1 && cb0 && cb1 && c2 && cb2 && tmpl && sel && Y0 && raw && Xh && H0 && H1 && density && rgb,
               -2, "hugs_nf_field_fwd: null pointer");
  FieldFwd P;
  P.M = (int)M; P.S = S; P.ldx0 = ldx0; P.ldw0 = ldw0; P.ngeo = ngeo;
  P.X0 = (const uint16_t*)X0;
  P.W0t = (const uint16_t*)W0t; P.W1t = (const uint16_t*)W1x; P.C0t = (const uint16_t*)C0t; P.C1t = (const uint16_t*)C1t;
  P.b0 = b0; P.b1 = b1; P.cb0 = cb0; P.cb1 = cb1; P.c2 = c2; P.cb2 = cb2; P.tmpl = (const uint16_t*)tmpl; P.sel = sel;
  P.Y0 = (uint16_t*)Y0; P.raw = (uint16_t*)raw; P.Xh = (uint16_t*)Xh; P.H0 = (uint16_t*)H0; P.H1 = (uint16_t*)H1;
  P.bY0 = bY0; P.bH0 = bH0; P.density = density; P.rgb = rgb;
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
  const int ntile = (int)(M / FF_ROWS);
  const dim3 grid(ntile < 2 * ncu ? ntile : 2 * ncu), block(256);
  if (dtype == 2) hipLaunchKernelGGL(k_field_fwd<1>, grid, block, 0, (hipStream_t)stream, P);
  else hipLaunchKernelGGL(k_field_fwd<0>, grid, block, 0, (hipStream_t)stream, P);
  HUGS_CHECK_LAUNCH("hugs_nf_field_fwd");
  return 0;
}

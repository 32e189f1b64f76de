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
#include <type_traits>
#include <stdlib.h>
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
// relu of a packed pair of 16-bit floats: a signed 16-bit max with 0 (negative values have the sign bit set in both formats)
__device__ __forceinline__ uint32_t ff_relu_pk(uint32_t u) {
  typedef short __attribute__((ext_vector_type(2))) s2;
  const s2 z = {0, 0};
  const s2 r = __builtin_elementwise_max(__builtin_bit_cast(s2, u), z);
  return __builtin_bit_cast(uint32_t, r);
}
// 1 in each 16-bit half that is not zero (relu outputs: positive or +0): v_pk_min_u16 with 1 -- bits 0 and 16, the mask layout's pair
// (as inline asm -- round 5: from __builtin_elementwise_min the compiler made two 16-bit compares, two selects and a v_perm per
//  pair, ~300 vector instructions per tile of the forward kernel; the inline constant 1 feeds both halves through op_sel_hi 0)
__device__ __forceinline__ uint32_t ff_nz_pk(uint32_t u) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(u));
  return r;
}
template <int F16> __device__ __forceinline__ uint32_t ff_cvt_pk(float a, float b) {       // one v_cvt_pk_{bf16,f16}_f32 (RNE)
  const ff_f32x2_t f = {a, b};
  const typename FfOps<F16>::x2_t h = __builtin_convertvector(f, typename FfOps<F16>::x2_t);
  return *(const uint32_t*)&h;
}

// A lane offset the optimiser may not treat as loop-invariant: with the whole tile as straight-line code, every (uniform base +
// lane offset) pair of its ~150 loads and stores was turned into a 64-bit per-lane address, hoisted out of the tile loop and
// spilled (264 B of scratch per lane); an opaque copy per use keeps them as one 32-bit VGPR next to a scalar base.
__device__ __forceinline__ unsigned ff_fresh(unsigned x) { asm volatile("" : "+v"(x)); return x; }

#define FF_ROWS 64
#define FF_STAGE (FF_ROWS * 64)        // one K-stage: 64 rows x 32 k x 2 B
#define FF_ACT (8 * FF_STAGE)          // a 256-wide activation tile

// byte offset of element (row, col) of an activation tile in the K-stage layout
__device__ __forceinline__ int ff_at(int row, int col) {
  return (col >> 5) * FF_STAGE + row * 64 + ((((col & 31) >> 3) ^ (3 * ((row >> 2) & 1))) << 4) + (col & 7) * 2;
}

// This thread's share of a finished activation tile (NST K-stages = 32 NST columns of all 64 rows) LDS -> HBM rows of 64 NST bytes:
// 16 bytes per lane, a row's 64 NST bytes contiguous over 4 NST lanes.  Loads and stores retire through ONE in-order counter, so
// a store issued ahead of a weight load is waited for with it: the copies sit in the NEXT layer's MFMA loop right after that
// loop instead of in the epilogue that produced the tile, and the only loads of the tile loop (the next tile's inputs) are
// requested ahead of them and consumed in straight-line code (counted vmcnt: the stores behind them stay in flight).
template <int NST>
__device__ __forceinline__ void ff_copy_out(const unsigned char* src, char* dst_tile, int tid) {
#pragma unroll
  for (int q = 0; q < NST; ++q) {
    const int id = q * 256 + tid, row = id / (NST * 4), cc = id % (NST * 4);
    const uint4 v = *(const uint4*)(src + (cc >> 2) * FF_STAGE + row * 64 + (((cc & 3) ^ (3 * ((row >> 2) & 1))) << 4));
    *(uint4*)(dst_tile + ff_fresh((unsigned)(row * (NST * 64) + cc * 16))) = v;
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

#ifdef FF_TRACE      // (timing builds: s_memtime at the phase edges of workgroup 0's first tiles, waves 0 and 3)
__device__ long long ff_trace_buf[2 * 8 * 16];
#define FF_TP(k_) do { if (blockIdx.x == 0 && ti < 8 && (tid == 0 || tid == 192)) ff_trace_buf[((tid != 0) * 8 + ti) * 16 + (k_)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FF_TP(k_) do { } while (0)
#endif

struct FieldFwd {
  int M, S, ldx0, ldw0, ngeo;
  int dact; float dbias;                    // density activation (hugs_common.h nf_density_value)
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

// thread (row = tid >> 2, cc = tid & 3) requests its part of a tile's inputs: 16 bytes of the row's hash features and the 64
// bytes [32 cc, 32 cc + 32) of its ray's head-input template (hugs_nf_head_template: [SH16 | 0 (geo) | appearance | 0], 16-bit)
struct FfIn { uint4 x, t0, t1, t2, t3; };
__device__ __forceinline__ FfIn ff_load_inputs(const FieldFwd& P, int m0, int tid) {
  const int row = tid >> 2, cc = tid & 3;
  FfIn r;
  r.x = *(const uint4*)((const char*)(P.X0 + (size_t)m0 * P.ldx0) + ff_fresh((unsigned)(row * P.ldx0 + cc * 8) * 2u));
  const unsigned ray = (unsigned)(m0 + row) / (unsigned)P.S;
  const char* tp = (const char*)P.tmpl + ff_fresh(ray * 256u + (unsigned)cc * 64u);
  r.t0 = *(const uint4*)tp; r.t1 = *(const uint4*)(tp + 16); r.t2 = *(const uint4*)(tp + 32); r.t3 = *(const uint4*)(tp + 48);
  return r;
}

// One 256-wide relu layer of a tile for this wave (its 64 output columns of all 64 rows), 16-row block by 16-row block: the MFMAs
// of row block i + 1 are issued ahead of the epilogue of block i (relu, 16-bit rounding, the next layer's LDS tile, mask bits;
// LAST: the HBM row + the rgb head's partial sums), so that the epilogue's VALU work fills the issue slots between independent
// MFMAs -- with one wave per SIMD nothing else would.  (The weights sit in registers, so the loop order is free.)
template <int F16, int KST, int CP, bool LAST>
__device__ __forceinline__ void ff_layer256(const unsigned char* A, int frag_off, const typename FfOps<F16>::x8_t (&w)[KST][4],
                                            const float* bias /* + wn*64 + kb*4 */, int m0, int wn, int r16, int kb, int lane,
                                            unsigned char* An, uint32_t* bout, uint16_t* Y, const unsigned char* cp_src, char* cp_dst,
                                            int tid, const typename FfOps<F16>::x8_t (&c2f)[2][3], float (*red)[4][3]) {
  typedef typename FfOps<F16>::x8_t x8_t;
  const int swz = 3 * ((r16 >> 2) & 1);
  float4 bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bb[j] = *(const float4*)(bias + j * 16);
  ff_f32x4_t acc[2][4];
  auto mma_rows = [&](int i, ff_f32x4_t (&a)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ff_f32x4_t{bb[j].x, bb[j].y, bb[j].z, bb[j].w};
#ifndef FF_NOMMA
    x8_t xa[KST];      // (all of the row block's fragments requested before the first MFMA: one LDS latency per block, not per stage)
#pragma unroll
    for (int s = 0; s < KST; ++s) xa[s] = *(const x8_t*)(A + s * FF_STAGE + frag_off + i * 16 * 64);
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler sinks each read back in front of its MFMAs otherwise)
#pragma unroll
    for (int s = 0; s < KST; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = FfOps<F16>::mfma(w[s][j], xa[s], a[j]);
#endif
  };
  // (LAST <=> the layer's rows go to HBM from the registers, no LDS tile; the mask bits are always computed and only their store
  //  depends on the pointer: a null test per fragment split the layer into ~35 basic blocks, and MFMAs do not move across those)
  uint16_t* Yout = Y + (size_t)m0 * 256;                                  // (uniform)
  const unsigned yo = ff_fresh((unsigned)(r16 * 256 + wn * 64 + kb * 4) * 2u);      // (bytes)
  const unsigned bo = ff_fresh((unsigned)lane * 4u);
  // hugs_gemm.hip nt_epilogue_direct bit layout: NT tile = 256 rows; its wave (wm_nt, wn) covers 128 rows = fragment rows i_nt 0..7
  const int i_nt0 = ((m0 >> 6) & 1) * 4;
  uint32_t* btile = bout ? bout + ((size_t)(m0 >> 8) * 8 + (size_t)(((m0 >> 7) & 1) * 4 + wn)) * 256 : nullptr;      // (uniform)
  uint32_t bw = 0u;
  mma_rows(0, acc[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < 3) mma_rows(i + 1, acc[(i + 1) & 1]);
    if constexpr (CP > 0) { if (i == 0) ff_copy_out<CP>(cp_src, cp_dst, tid); }
    uint32_t uk[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const ff_f32x4_t v = acc[i & 1][j];
      uint2 u;
      u.x = ff_relu_pk(ff_cvt_pk<F16>(v[0], v[1])); u.y = ff_relu_pk(ff_cvt_pk<F16>(v[2], v[3]));
      uk[j][0] = u.x; uk[j][1] = u.y;
      if constexpr (LAST) {
#ifndef FF_NOSTORE
        *(uint2*)((char*)Yout + (yo + (unsigned)(i * 16 * 256 + j * 16) * 2u)) = u;
#endif
      } else {
        const int st = wn * 2 + (j >> 1), ch = (j & 1) * 2 + (kb >> 1);
        *(uint2*)(An + st * FF_STAGE + (i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
        const int k = (i & 1) * 8 + j * 2;
        bw |= ff_nz_pk(u.x) << k;
        bw |= ff_nz_pk(u.y) << (k + 1);
      }
    }
    if (!LAST && (i & 1)) { if (bout) *(uint32_t*)((char*)btile + (bo + (unsigned)(((i_nt0 + i) >> 1) * 64) * 4u)) = bw; bw = 0u; }
    if (LAST) {
      // rgb head on the rounded activations, on the matrix cores: the packed outputs of two neighbouring 16-column fragments ARE a
      // B operand (lane (row, kb) holds k-slots kb*8 .. +7 = columns (2a)*16 + kb*4 .. +3 and (2a+1)*16 + kb*4 .. +3) once the
      // A operand c2f[a] carries the rgb layer's weights in the same slot order; its fp32 weights enter as 16-bit hi + lo (+ a third
      // slice in bf16) operands.
      // D[n][row]: the lanes kb == 0 end up with (r, g, b) partial sums of this wave's 64 columns for row r16 -- no lane shuffles.
      ff_f32x4_t pr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        typedef unsigned __attribute__((ext_vector_type(4))) u4;
        const u4 bw4 = {uk[2 * a][0], uk[2 * a][1], uk[2 * a + 1][0], uk[2 * a + 1][1]};
        const x8_t bf = __builtin_bit_cast(x8_t, bw4);
        pr = FfOps<F16>::mfma(c2f[a][0], bf, pr);
        pr = FfOps<F16>::mfma(c2f[a][1], bf, pr);
        if (!F16) pr = FfOps<F16>::mfma(c2f[a][2], bf, pr);      // (bf16: a third 8-bit slice, 24 bits of the fp32 weight in all)
      }
      if (kb == 0) { red[i * 16 + r16][wn][0] = pr[0]; red[i * 16 + r16][wn][1] = pr[1]; red[i * 16 + r16][wn][2] = pr[2]; }
    }
  }
}

// register-resident weights: stage s, fragment j of a [n][k] matrix for this lane (W uniform, wo = the lane's byte offset)
template <int F16, int KST, int NJ>
__device__ __forceinline__ void ff_load_w(const uint16_t* W, unsigned wo, int ldw, typename FfOps<F16>::x8_t (&w)[KST][NJ]) {
  typedef typename FfOps<F16>::x8_t x8_t;
#pragma unroll
  for (int s = 0; s < KST; ++s)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      w[s][j] = *(const x8_t*)((const char*)W + (wo + (unsigned)(j * 16 * ldw + s * 32) * 2u));
#ifndef FF_W_ANY
      // park the weights in the accumulation-register half of the file (MFMA takes its A operand from there): left to itself the
      // allocator filled all 256 VGPRs with them, kept the ACCUMULATORS in AGPRs (a v_accvgpr_read per value in every epilogue)
      // and had ONE 4-register buffer for the activation fragments -- every K-stage was ds_read, wait, 4 MFMAs, in series
      asm volatile("" : "+a"(w[s][j]));
#endif
    }
}

// acc[i][j] += sum over KST stages of W-fragment(s, j) x X-fragment(i, s); A = the activation tile (its first stage).  CP > 0: the
// copy-out job of an earlier tile (cp_src, CP stages) rides in this loop (ff_copy_out).
template <int F16, int KST, int NJ, int CP = 0>
__device__ __forceinline__ void ff_mma_r(const unsigned char* A, int frag_off, const typename FfOps<F16>::x8_t (&w)[KST][NJ],
                                         ff_f32x4_t (&acc)[4][4], const unsigned char* cp_src = nullptr, char* cp_dst = nullptr,
                                         int tid = 0) {
  typedef typename FfOps<F16>::x8_t x8_t;
#ifdef FF_NOMMA      // (timing builds, scratch/r4_ffuse_var.sh)
  if constexpr (CP > 0) ff_copy_out<CP>(cp_src, cp_dst, tid);
  return;
#endif
  x8_t xa[2][4];      // (the next stage's fragments are requested before this stage's MFMAs)
#pragma unroll
  for (int i = 0; i < 4; ++i) xa[0][i] = *(const x8_t*)(A + frag_off + i * 16 * 64);
#pragma unroll
  for (int s = 0; s < KST; ++s) {
    if (s + 1 < KST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xa[(s + 1) & 1][i] = *(const x8_t*)(A + (s + 1) * FF_STAGE + frag_off + i * 16 * 64);
    }
    if constexpr (CP > 0) { if (s == 1) ff_copy_out<CP>(cp_src, cp_dst, tid); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = FfOps<F16>::mfma(w[s][j], xa[s & 1][i], acc[i][j]);
  }
}

// a tile's inputs -> LDS: the thread's 16 bytes of hash features -> stage 7 of `A`, its 64 bytes of the head-input template ->
// stage cc (the geo columns of stages 0..2 are overwritten by layer 1's epilogue), the selector of row tid -> sel_s
__device__ __forceinline__ void ff_put_inputs(unsigned char* A, float* sel_s, const FfIn& nx, float nsel, int tid) {
  const int row = tid >> 2, cc = tid & 3, sw = 3 * ((row >> 2) & 1);
  *(uint4*)(A + 7 * FF_STAGE + row * 64 + ((cc ^ sw) << 4)) = nx.x;
  unsigned char* tb = A + cc * FF_STAGE + row * 64;
  *(uint4*)(tb + ((0 ^ sw) << 4)) = nx.t0; *(uint4*)(tb + ((1 ^ sw) << 4)) = nx.t1;
  *(uint4*)(tb + ((2 ^ sw) << 4)) = nx.t2; *(uint4*)(tb + ((3 ^ sw) << 4)) = nx.t3;
  if (tid < FF_ROWS) sel_s[tid] = nsel;
}

// One workgroup per CU, one wave per SIMD: the lane's share of ALL FOUR weight matrices (16 + 64 + 64 + 128 = 272 registers of the
// 512) is loaded once and stays in registers for the whole kernel -- with two workgroups per CU and the weights streamed from L2
// three K-stages ahead (the first form of this kernel, 1.6-1.9 ms for 2 M samples) every stage of every layer waited ~1.5 k
// cycles for its weights (s_memtime trace: 14 k cycles in colour layer 1's loop for 2 k cycles of MFMAs) and the 272 KB per
// 64-row tile added up to 9 GB of L2 reads.  No global load is waited for inside the tile loop except the NEXT tile's inputs,
// requested a layer and a half before they are written to LDS (counted vmcnt in straight-line code: the stores behind them
// stay in flight).
template <int F16>
__global__ __launch_bounds__(256, 1) void k_field_fwd(const FieldFwd P) {
  typedef typename FfOps<F16>::x8_t x8_t;
  __shared__ __attribute__((aligned(16))) unsigned char act[2][FF_ACT];
  __shared__ float red[FF_ROWS][4][3];
  __shared__ float sel_s[FF_ROWS];
  __shared__ float cb2s[4];
  __shared__ __attribute__((aligned(16))) float bs[256 + 128 + 256 + 256];      // b0 | b1x | cb0 | cb1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kb = lane >> 4;
  const int swz = 3 * ((r16 >> 2) & 1);
  const int frag_off = r16 * 64 + ((kb ^ swz) << 4);
  const int ntile = P.M / FF_ROWS, G = (int)gridDim.x;
  if (tid < 3) cb2s[tid] = P.cb2[tid];
  // the rgb layer [256, 3] fp32 as MFMA A operands in the slot order of ff_layer256's B operands: lane (n = r16, kb), slot e of
  // half a -> column wn*64 + (2a + (e >> 2))*16 + kb*4 + (e & 3); rows n >= 3 are zero; each value as 16-bit hi + lo (+ third) slices
  x8_t c2f[2][3];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    u4 hw, lw, tw;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int c0 = wn * 64 + (2 * a + (e2 >> 1)) * 16 + kb * 4 + (e2 & 1) * 2;
      const float v0 = r16 < 3 ? P.c2[c0 * 3 + r16] : 0.f, v1 = r16 < 3 ? P.c2[(c0 + 1) * 3 + r16] : 0.f;
      const uint32_t h = ff_cvt_pk<F16>(v0, v1);
      const float r0 = v0 - FfOps<F16>::lo(h), r1 = v1 - FfOps<F16>::hi(h);
      const uint32_t l = ff_cvt_pk<F16>(r0, r1);
      hw[e2] = h; lw[e2] = l;
      tw[e2] = ff_cvt_pk<F16>(r0 - FfOps<F16>::lo(l), r1 - FfOps<F16>::hi(l));
    }
    c2f[a][0] = __builtin_bit_cast(x8_t, hw); c2f[a][1] = __builtin_bit_cast(x8_t, lw); c2f[a][2] = __builtin_bit_cast(x8_t, tw);
  }
  bs[tid] = P.b0[tid]; bs[384 + tid] = P.cb0[tid]; bs[640 + tid] = P.cb1[tid];
  if (tid < 128) bs[256 + tid] = P.b1[tid];
  const bool l1_live = wn * 32 < 16 + P.ngeo;      // (wave-uniform) this wave's 32 columns of layer 1 hold real outputs
  x8_t w0r[1][4], w1r[8][2], c0r[4][4], c1r[8][4];
  ff_load_w<F16, 1, 4>(P.W0t, (unsigned)((wn * 64 + r16) * P.ldw0 + kb * 8) * 2u, P.ldw0, w0r);
  ff_load_w<F16, 8, 2>(P.W1t, (unsigned)((wn * 32 + r16) * 256 + kb * 8) * 2u, 256, w1r);      // (rows past 16 + ngeo are zero)
  ff_load_w<F16, 4, 4>(P.C0t, (unsigned)((wn * 64 + r16) * 128 + kb * 8) * 2u, 128, c0r);
  ff_load_w<F16, 8, 4>(P.C1t, (unsigned)((wn * 64 + r16) * 256 + kb * 8) * 2u, 256, c1r);

  FfIn nx;
  float nsel = 0.f;
  if ((int)blockIdx.x < ntile) {
    nx = ff_load_inputs(P, (int)blockIdx.x * FF_ROWS, tid);
    if (tid < FF_ROWS) nsel = P.sel[(size_t)blockIdx.x * FF_ROWS + tid];
    ff_put_inputs(act[0], sel_s, nx, nsel, tid);
  }
  __syncthreads();
  int ti = -1;
  for (int t = blockIdx.x; t < ntile; t += G) {
    const int m0 = t * FF_ROWS;
    const bool has_next = t + G < ntile;
    ++ti;
    FF_TP(0);
    ff_f32x4_t acc[4][4];
    // ---- base layer 0: 32 -> 256, relu ------------------------------------------------------------------------------
    ff_layer256<F16, 1, 0, false>(act[0] + 7 * FF_STAGE, frag_off, w0r, bs + wn * 64 + kb * 4, m0, wn, r16, kb, lane, act[1], P.bY0, nullptr,
                                  nullptr, nullptr, tid, c2f, red);      // (Y0 -> HBM: in layer 1's loop)
    FF_TP(1);
    FF_TP(2);
    __syncthreads();
    FF_TP(3);
    // ---- base layer 1: 256 -> 1 + ngeo (no activation), computed in HEAD-INPUT column order (W1t / b1 rows: 0 = raw density,
    // 16 .. 16 + ngeo = the geo features, the rest zero): the geo block lands in the head-input tile as aligned 8-byte writes
    if (l1_live) {
      ff_acc_bias<2>(acc, bs + 256 + wn * 32 + kb * 4);
      ff_mma_r<F16, 8, 2, 8>(act[1], frag_off, w1r, acc, act[1], (char*)(P.Y0 + (size_t)m0 * 256), tid);
      FF_TP(4);
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
            *(float*)((char*)(P.density + m0) + ff_fresh((unsigned)row * 4u)) = nf_density_value(FfOps<F16>::lo(u.x), P.dact, P.dbias) * sel_s[row];
          }
        }
      }
    } else {
      ff_copy_out<8>(act[1], (char*)(P.Y0 + (size_t)m0 * 256), tid);
      FF_TP(4);
    }
    FF_TP(5);
    __syncthreads();
    FF_TP(6);
    // ---- colour layer 0: 128 -> 256, relu (the head input goes to HBM in its loop: the first colour weight gradient reads it);
    // the next tile's inputs are requested here and written to LDS after colour layer 1's loop
    if (has_next) {
      nx = ff_load_inputs(P, m0 + G * FF_ROWS, tid);
      if (tid < FF_ROWS) nsel = *(const float*)((const char*)(P.sel + (size_t)m0 + (size_t)G * FF_ROWS) + ff_fresh((unsigned)tid * 4u));
    }
    ff_layer256<F16, 4, 4, false>(act[0], frag_off, c0r, bs + 384 + wn * 64 + kb * 4, m0, wn, r16, kb, lane, act[1], P.bH0, nullptr,
                                  act[0], (char*)(P.Xh + (size_t)m0 * 128), tid, c2f, red);      // (H0 -> HBM: in the next loop)
    FF_TP(7);
    FF_TP(8);
    __syncthreads();
    FF_TP(9);
    // ---- colour layer 1: 256 -> 256, relu; rgb = sigmoid(H1 c2 + cb2) on the rounded activations ---------------------------
    if (has_next) ff_put_inputs(act[0], sel_s, nx, nsel, tid);      // (act[0] is free since the barrier above)
    FF_TP(10);
    ff_layer256<F16, 8, 8, true>(act[1], frag_off, c1r, bs + 640 + wn * 64 + kb * 4, m0, wn, r16, kb, lane, nullptr, nullptr, P.H1,
                                 act[1], (char*)(P.H0 + (size_t)m0 * 256), tid, c2f, red);
    FF_TP(11);
    FF_TP(12);
    __syncthreads();
    FF_TP(13);
    if (tid < FF_ROWS * 3) {
      const int row = tid / 3, c = tid - row * 3;
      const float a = ((red[row][0][c] + red[row][1][c]) + (red[row][2][c] + red[row][3][c])) + cb2s[c];
      *(float*)((char*)(P.rgb + (size_t)m0 * 3) + ff_fresh((unsigned)tid * 4u)) = 1.f / (1.f + expf(-a));
    }
    FF_TP(14);
  }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the same forward on EIGHT waves (two per SIMD), as k_mlp256_chain3_fwd8 (hugs_mlpfuse.hip): wave wq owns output columns
// [32 wq, 32 wq + 32) of the 256-wide layers (16 of layer 1's 128) for all 64 rows -- 136 registers of weights instead of 272, so two
// waves fit a SIMD and one wave's epilogue / LDS waits sit under the other's MFMAs (one wave per SIMD: ~0.5 of the byte bound).  Costs:
// every wave reads the whole activation tile as its B operand (twice the LDS fragment traffic), the two waves of a pair each hold half
// of the pair's mask words (OR-ed through LDS behind the layer's barrier), the rgb head's partial sums come from 8 waves.
// Same arithmetic per output element as k_field_fwd (same K order, same rounding points): bit-identical activations, densities,
// mask bits; rgb sums its 8 partial dot products in a different order than the 4-wave kernel's 4.
// ------------------------------------------------------------------------------------------------
// a tile's inputs over 512 threads: thread (row = tid >> 3, cc = tid & 7) requests 8 bytes of the row's hash features and the 32 bytes
// [32 cc, 32 cc + 32) of its ray's head-input template (half of k_field_fwd's per-thread share: 10 registers in flight instead of 20)
struct Ff8In { uint2 x; uint4 t0, t1; };
__device__ __forceinline__ Ff8In ff8_load_inputs(const FieldFwd& P, int m0, int tid) {
  const int row = tid >> 3, cc = tid & 7;
  Ff8In r;
  r.x = *(const uint2*)((const char*)(P.X0 + (size_t)m0 * P.ldx0) + ff_fresh((unsigned)(row * P.ldx0 + cc * 4) * 2u));
  const unsigned ray = (unsigned)(m0 + row) / (unsigned)P.S;
  const char* tp = (const char*)P.tmpl + ff_fresh(ray * 256u + (unsigned)cc * 32u);
  r.t0 = *(const uint4*)tp; r.t1 = *(const uint4*)(tp + 16);
  return r;
}
__device__ __forceinline__ void ff8_put_inputs(unsigned char* A, float* sel_s, const Ff8In& nx, float nsel, int tid) {
  const int row = tid >> 3, cc = tid & 7, sw = 3 * ((row >> 2) & 1);
  *(uint2*)(A + 7 * FF_STAGE + row * 64 + (((cc >> 1) ^ sw) << 4) + (cc & 1) * 8) = nx.x;
  unsigned char* tb = A + (cc >> 1) * FF_STAGE + row * 64;
  const int k0 = (cc & 1) * 2;
  *(uint4*)(tb + ((k0 ^ sw) << 4)) = nx.t0; *(uint4*)(tb + (((k0 + 1) ^ sw) << 4)) = nx.t1;
  if (tid < FF_ROWS) sel_s[tid] = nsel;
}

template <int NST>
__device__ __forceinline__ void ff8_copy_out(const unsigned char* src, char* dst_tile, int tid) {
  // a finished tile (NST K-stages of all 64 rows) LDS -> HBM rows of 64 NST bytes, 16 bytes per thread and step, 512 threads
#pragma unroll
  for (int q = 0; q < NST / 2; ++q) {
    const int id = q * 512 + tid, row = id / (NST * 4), cc = id % (NST * 4);
    const uint4 v = *(const uint4*)(src + (cc >> 2) * FF_STAGE + row * 64 + (((cc & 3) ^ (3 * ((row >> 2) & 1))) << 4));
    *(uint4*)(dst_tile + ff_fresh((unsigned)(row * (NST * 64) + cc * 16))) = v;
  }
}

template <int F16, int KST, int CP, bool LAST>
__device__ __forceinline__ void ff8_layer256(const unsigned char* A, int frag_off, const typename FfOps<F16>::x8_t (&w)[KST][2],
                                             const float* bias /* + wq*32 + kb*4 */, int m0, int wq, int r16, int kb, int lane,
                                             unsigned char* An, uint32_t* bpart /* LDS [2][64] of this wave */, uint16_t* Y,
                                             const unsigned char* cp_src, char* cp_dst, int tid,
                                             const typename FfOps<F16>::x8_t (&c2f)[3], float (*red)[8][3]) {
  typedef typename FfOps<F16>::x8_t x8_t;
  const int swz = 3 * ((r16 >> 2) & 1);
  float4 bb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bb[j] = *(const float4*)(bias + j * 16);
  const int jj0 = (wq & 1) * 2;      // this wave's fragments are j = jj0, jj0 + 1 of its NT wave's 64-column block
  uint32_t bw = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ff_f32x4_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[j] = ff_f32x4_t{bb[j].x, bb[j].y, bb[j].z, bb[j].w};
    {
      x8_t xa[KST];
#pragma unroll
      for (int s = 0; s < KST; ++s) xa[s] = *(const x8_t*)(A + s * FF_STAGE + frag_off + i * 16 * 64);
#pragma unroll
      for (int s = 0; s < KST; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = FfOps<F16>::mfma(w[s][j], xa[s], acc[j]);
    }
    if constexpr (CP > 0) { if (i == 0) ff8_copy_out<CP>(cp_src, cp_dst, tid); }
    uint32_t uk[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const ff_f32x4_t v = acc[j];
      uint2 u;
      u.x = ff_relu_pk(ff_cvt_pk<F16>(v[0], v[1])); u.y = ff_relu_pk(ff_cvt_pk<F16>(v[2], v[3]));
      uk[j][0] = u.x; uk[j][1] = u.y;
      // (LAST: H1 goes through LDS too -- as 8-byte stores from the accumulator layout a row block's 16 rows x 32 bytes cost the
      //  vector-memory path ~100 cycles per instruction; the tile leaves as whole 512-byte rows from inside the NEXT tile's first layers)
      const int ch = j * 2 + (kb >> 1);
      *(uint2*)(An + wq * FF_STAGE + (i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
      if constexpr (!LAST) {
        const int k = (i & 1) * 8 + (jj0 + j) * 2;
        bw |= ff_nz_pk(u.x) << k;
        bw |= ff_nz_pk(u.y) << (k + 1);
      }
    }
    if (!LAST && (i & 1)) { bpart[(i >> 1) * 64 + lane] = bw; bw = 0u; }
    if (LAST) {
      // rgb head on the rounded activations (k_field_fwd's scheme): this wave's two fragments are ONE B operand
      typedef unsigned __attribute__((ext_vector_type(4))) u4;
      const u4 bw4 = {uk[0][0], uk[0][1], uk[1][0], uk[1][1]};
      const x8_t bf = __builtin_bit_cast(x8_t, bw4);
      ff_f32x4_t pr = {0.f, 0.f, 0.f, 0.f};
      pr = FfOps<F16>::mfma(c2f[0], bf, pr);
      pr = FfOps<F16>::mfma(c2f[1], bf, pr);
      if (!F16) pr = FfOps<F16>::mfma(c2f[2], bf, pr);
      if (kb == 0) { red[i * 16 + r16][wq][0] = pr[0]; red[i * 16 + r16][wq][1] = pr[1]; red[i * 16 + r16][wq][2] = pr[2]; }
    }
    __builtin_amdgcn_sched_barrier(0);      // (nothing of the next row block is hoisted above this epilogue: the other wave fills the pipe)
  }
}

template <int F16>
__global__ __launch_bounds__(512, 1) void k_field_fwd8(const FieldFwd P) {
  typedef typename FfOps<F16>::x8_t x8_t;
  __shared__ __attribute__((aligned(16))) unsigned char act[3][FF_ACT];      // [2]: the finished H1 tile on its way out
  __shared__ float red[FF_ROWS][8][3];
  __shared__ float sel_s[FF_ROWS];
  __shared__ float cb2s[4];
  __shared__ __attribute__((aligned(16))) float bs[256 + 128 + 256 + 256];      // b0 | b1x | cb0 | cb1
  __shared__ uint32_t bparts[2][8][2][64];                                       // [layer parity][wave][word][lane]: mask-bit halves
  const int tid = threadIdx.x, lane = tid & 63;
  const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kb = lane >> 4;
  const int swz = 3 * ((r16 >> 2) & 1);
  const int frag_off = r16 * 64 + ((kb ^ swz) << 4);
  const int ntile = P.M / FF_ROWS, G = (int)gridDim.x;
  if (tid < 3) cb2s[tid] = P.cb2[tid];
  // the rgb layer's weights for this wave's 32 columns as an MFMA A operand in the slot order of the packed outputs: lane (n = r16, kb),
  // slot e -> column wq*32 + (e >> 2)*16 + kb*4 + (e & 3); hi + lo (+ third) 16-bit slices of the fp32 value
  x8_t c2f[3];
  {
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    u4 hw, lw, tw;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int c0 = wq * 32 + (e2 >> 1) * 16 + kb * 4 + (e2 & 1) * 2;
      const float v0 = r16 < 3 ? P.c2[c0 * 3 + r16] : 0.f, v1 = r16 < 3 ? P.c2[(c0 + 1) * 3 + r16] : 0.f;
      const uint32_t h = ff_cvt_pk<F16>(v0, v1);
      const float r0 = v0 - FfOps<F16>::lo(h), r1 = v1 - FfOps<F16>::hi(h);
      const uint32_t l = ff_cvt_pk<F16>(r0, r1);
      hw[e2] = h; lw[e2] = l;
      tw[e2] = ff_cvt_pk<F16>(r0 - FfOps<F16>::lo(l), r1 - FfOps<F16>::hi(l));
    }
    c2f[0] = __builtin_bit_cast(x8_t, hw); c2f[1] = __builtin_bit_cast(x8_t, lw); c2f[2] = __builtin_bit_cast(x8_t, tw);
  }
  for (int e = tid; e < 256; e += 512) { bs[e] = P.b0[e]; bs[384 + e] = P.cb0[e]; bs[640 + e] = P.cb1[e]; }
  if (tid < 128) bs[256 + tid] = P.b1[tid];
  const bool l1_live = wq * 16 < 16 + P.ngeo;      // (wave-uniform) this wave's 16 columns of layer 1 hold real outputs
  x8_t w0r[1][2], w1r[8][1], c0r[4][2], c1r[8][2];
  {
    auto load_w = [&](auto& w, const uint16_t* W, unsigned wo, int ldw, auto kst, auto nj) {
#pragma unroll
      for (int s = 0; s < decltype(kst)::value; ++s)
#pragma unroll
        for (int j = 0; j < decltype(nj)::value; ++j) w[s][j] = *(const x8_t*)((const char*)W + (wo + (unsigned)(j * 16 * ldw + s * 32) * 2u));
    };
    load_w(w0r, P.W0t, (unsigned)((wq * 32 + r16) * P.ldw0 + kb * 8) * 2u, P.ldw0, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
    load_w(w1r, P.W1t, (unsigned)((wq * 16 + r16) * 256 + kb * 8) * 2u, 256, std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
    load_w(c0r, P.C0t, (unsigned)((wq * 32 + r16) * 128 + kb * 8) * 2u, 128, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
    load_w(c1r, P.C1t, (unsigned)((wq * 32 + r16) * 256 + kb * 8) * 2u, 256, std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});
  }
  // the two waves of a pair each hold half of the pair's mask words: wave 2p stores word 0, wave 2p + 1 word 1 (hugs_gemm.hip's layout)
  auto bits_out = [&](int par, uint32_t* bits, int m0) {
    if (!bits) return;
    const int p2 = wq & ~1, wd_ = wq & 1;
    const uint32_t v = bparts[par][p2][wd_][lane] | bparts[par][p2 + 1][wd_][lane];
    uint32_t* btile = bits + ((size_t)(m0 >> 8) * 8 + (size_t)(((m0 >> 7) & 1) * 4 + (wq >> 1))) * 256;
    const int i_nt0 = ((m0 >> 6) & 1) * 4;
    *(uint32_t*)((char*)btile + (ff_fresh((unsigned)lane * 4u) + (unsigned)(((i_nt0 >> 1) + wd_) * 64) * 4u)) = v;
  };
  Ff8In nx;
  float nsel = 0.f;
  if ((int)blockIdx.x < ntile) {
    nx = ff8_load_inputs(P, (int)blockIdx.x * FF_ROWS, tid);
    if (tid < FF_ROWS) nsel = P.sel[(size_t)blockIdx.x * FF_ROWS + tid];
    ff8_put_inputs(act[0], sel_s, nx, nsel, tid);
  }
  __syncthreads();
  int prev_m0 = -1, ti = -1;
  for (int t = blockIdx.x; t < ntile; t += G) {
    const int m0 = t * FF_ROWS;
    const bool has_next = t + G < ntile;
    if (prev_m0 >= 0) ff8_copy_out<8>(act[2], (char*)(P.H1 + (size_t)prev_m0 * 256), tid);      // (act[2] is rewritten three barriers from here)
    prev_m0 = m0;
    ++ti;
    FF_TP(0);
    // ---- base layer 0: 32 -> 256, relu ------------------------------------------------------------------------------
    ff8_layer256<F16, 1, 0, false>(act[0] + 7 * FF_STAGE, frag_off, w0r, bs + wq * 32 + kb * 4, m0, wq, r16, kb, lane, act[1], &bparts[0][wq][0][0],
                                   nullptr, nullptr, nullptr, tid, c2f, red);      // (Y0 -> HBM: in layer 1's loop)
    FF_TP(1);
    __syncthreads();
    FF_TP(2);
    bits_out(0, P.bY0, m0);
    // ---- base layer 1: 256 -> 1 + ngeo (no activation) in HEAD-INPUT column order: this wave's 16 columns ----------------
    ff8_copy_out<8>(act[1], (char*)(P.Y0 + (size_t)m0 * 256), tid);
    FF_TP(3);
    if (l1_live) {
      const float4 b1v = *(const float4*)(bs + 256 + wq * 16 + kb * 4);
      const int n = wq * 16 + kb * 4;
      const bool geo = n >= 16 && n < 16 + P.ngeo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ff_f32x4_t a = {b1v.x, b1v.y, b1v.z, b1v.w};
        x8_t xa[8];
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) xa[s_] = *(const x8_t*)(act[1] + s_ * FF_STAGE + frag_off + i * 16 * 64);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) a = FfOps<F16>::mfma(w1r[s_][0], xa[s_], a);
        const int row = i * 16 + r16;
        uint2 u;
        u.x = ff_cvt_pk<F16>(a[0], a[1]); u.y = ff_cvt_pk<F16>(a[2], a[3]);
        if (geo) *(uint2*)(act[0] + ((n >> 5) * FF_STAGE + row * 64 + ((((n & 31) >> 3) ^ swz) << 4) + (n & 4) * 2)) = u;
        if (n == 0) {
          *(uint16_t*)((char*)(P.raw + m0) + ff_fresh((unsigned)row * 2u)) = (uint16_t)u.x;
          *(float*)((char*)(P.density + m0) + ff_fresh((unsigned)row * 4u)) = nf_density_value(FfOps<F16>::lo(u.x), P.dact, P.dbias) * sel_s[row];
        }
      }
    }
    FF_TP(4);
    __syncthreads();
    FF_TP(5);
    // ---- colour layer 0: 128 -> 256, relu (the head input goes to HBM in its loop); the next tile's inputs are requested here ----
    if (has_next) {
      nx = ff8_load_inputs(P, m0 + G * FF_ROWS, tid);
      if (tid < FF_ROWS) nsel = *(const float*)((const char*)(P.sel + (size_t)m0 + (size_t)G * FF_ROWS) + ff_fresh((unsigned)tid * 4u));
    }
    ff8_layer256<F16, 4, 4, false>(act[0], frag_off, c0r, bs + 384 + wq * 32 + kb * 4, m0, wq, r16, kb, lane, act[1], &bparts[1][wq][0][0], nullptr,
                                   act[0], (char*)(P.Xh + (size_t)m0 * 128), tid, c2f, red);      // (H0 -> HBM: in the next loop)
    FF_TP(6);
    __syncthreads();
    FF_TP(7);
    bits_out(1, P.bH0, m0);
    // ---- colour layer 1: 256 -> 256, relu; rgb = sigmoid(H1 c2 + cb2) on the rounded activations ---------------------------
    if (has_next) ff8_put_inputs(act[0], sel_s, nx, nsel, tid);      // (act[0] is free since the barrier above)
    FF_TP(8);
    ff8_layer256<F16, 8, 8, true>(act[1], frag_off, c1r, bs + 640 + wq * 32 + kb * 4, m0, wq, r16, kb, lane, act[2], nullptr, nullptr,
                                  act[1], (char*)(P.H0 + (size_t)m0 * 256), tid, c2f, red);
    FF_TP(9);
    __syncthreads();
    FF_TP(10);
    if (tid < FF_ROWS * 3) {
      const int row = tid / 3, c = tid - row * 3;
      const float a = (((red[row][0][c] + red[row][1][c]) + (red[row][2][c] + red[row][3][c])) +
                       ((red[row][4][c] + red[row][5][c]) + (red[row][6][c] + red[row][7][c]))) + cb2s[c];
      *(float*)((char*)(P.rgb + (size_t)m0 * 3) + ff_fresh((unsigned)tid * 4u)) = 1.f / (1.f + expf(-a));
    }
    FF_TP(11);
  }
  if (prev_m0 >= 0) ff8_copy_out<8>(act[2], (char*)(P.H1 + (size_t)prev_m0 * 256), tid);      // (behind the last tile's final barrier)
}

// ------------------------------------------------------------------------------------------------
// Backward of the same two networks from the colour layer 1 gradient down to the hash-feature gradient (models/nerfacto.py's
// autograd through :693-759), one launch:
//   G0  = (G1 c1^T) * relu'(H0)            [M,256]   (G1 = the gradient at colour layer 1's pre-activation: hugs_rgb_bwd writes it)
//   dXh = G0 c0^T                          [M,128]   columns [16, 16+ngeo) = d geo features, [16+ngeo, +napp) summed per ray into the
//                                                    appearance-embedding gradient, [0,16) (the SH of the direction) dropped
//   Gb  = [d_raw | 0 x 15 | d geo | 0 ..]  [M,128]   d_raw = d_density * exp(clamp(raw, +-15)) * sel; HEAD-INPUT column order (as W1x)
//   Gy0 = (Gb W1x) * relu'(Y0)             [M,256]
//   dX0 = Gy0 w0^T                         [M,<=32]
// Same geometry as the forward: one workgroup per CU, the lane's share of the four TRANSPOSED-use weight matrices ([k_out][n], the
// `wn` copies: 128 + 64 + 64 + 32 = 288 registers) loaded once; every G written to HBM exactly once (the weight-gradient GEMMs
// read them), riding in the next layer's loop; relu' from the forward's 1-bit masks (hugs_gemm_nt_bits lane layout).
// Whole rays per tile: S must be a multiple of 64.
// ------------------------------------------------------------------------------------------------
struct FieldBwd {
  int M, S, ldx0, ngeo, napp;
  int dact; float dbias;                    // density activation (hugs_common.h nf_density_slope)
  int dx_f32;                               // dX0 is a float [M, ldx0] buffer (round 5: the grid-input gradient without a 16-bit rounding)
  const uint16_t* G1;                       // [M,256]
  const uint16_t *C1n, *C0n, *W1xn, *W0n;   // [256][256], [128][256], [256][128], [>=32][256]   ([k_out][n], 16-bit)
  const uint32_t *bH0, *bY0;                // relu masks of H0 / Y0 (forward layout)
  const float *d_density, *sel;             // [M]
  const uint16_t* raw;                      // [M] 16-bit raw density (forward output)
  const int* embed_idx;                     // [M / S]
  uint16_t *G0, *Gb, *Gy0, *dX0;            // [M,256], [M,128], [M,256], [M, ldx0] (columns 0..31 written)
  float* d_embedding;                       // [n_embeddings, napp] (+=, float atomics) or null
};

// mask-bit words of this lane for a 64-row tile of a 256-wide activation (forward layout): words (i_nt0 >> 1) and + 1
__device__ __forceinline__ void ff_load_bits(const uint32_t* bits, int m0, int wn, int lane, uint32_t (&bin)[2]) {
  const int i_nt0 = ((m0 >> 6) & 1) * 4;
  const uint32_t* btile = bits + ((size_t)(m0 >> 8) * 8 + (size_t)(((m0 >> 7) & 1) * 4 + wn)) * 256;
  const unsigned bo = ff_fresh((unsigned)lane * 4u);
  bin[0] = *(const uint32_t*)((const char*)btile + (bo + (unsigned)((i_nt0 >> 1) * 64) * 4u));
  bin[1] = *(const uint32_t*)((const char*)btile + (bo + (unsigned)(((i_nt0 >> 1) + 1) * 64) * 4u));
}

// One 256-wide masked layer of the backward for this wave (its 64 output columns): out = (A W^T) & mask -> the next LDS tile.
// Row blocks pipelined as in ff_layer256.
template <int F16, int KST, int CP>
__device__ __forceinline__ void ff_layer256_bwd(const unsigned char* A, int frag_off, const typename FfOps<F16>::x8_t (&w)[KST][4],
                                                const uint32_t (&bin)[2], int wn, int r16, int kb, unsigned char* An,
                                                const unsigned char* cp_src, char* cp_dst, int tid) {
  typedef typename FfOps<F16>::x8_t x8_t;
  const int swz = 3 * ((r16 >> 2) & 1);
  ff_f32x4_t acc[2][4];
  auto mma_rows = [&](int i, ff_f32x4_t (&a)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ff_f32x4_t{0.f, 0.f, 0.f, 0.f};
    x8_t xa[KST];
#pragma unroll
    for (int s = 0; s < KST; ++s) xa[s] = *(const x8_t*)(A + s * FF_STAGE + frag_off + i * 16 * 64);
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler sinks each read back in front of its MFMAs otherwise)
#pragma unroll
    for (int s = 0; s < KST; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = FfOps<F16>::mfma(w[s][j], xa[s], a[j]);
  };
  mma_rows(0, acc[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < 3) mma_rows(i + 1, acc[(i + 1) & 1]);
    if constexpr (CP > 0) { if (i == 0) ff_copy_out<CP>(cp_src, cp_dst, tid); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const ff_f32x4_t v = acc[i & 1][j];
      uint2 u;
      u.x = ff_cvt_pk<F16>(v[0], v[1]); u.y = ff_cvt_pk<F16>(v[2], v[3]);
      // hugs_gemm.hip nt_epilogue_direct: word (i_nt >> 1), pair k = (i_nt & 1) * 8 + j * 2 owns bit k (even column) and 16 + k (odd)
      const int k = (i & 1) * 8 + j * 2;
      const uint32_t t = bin[i >> 1] >> k;
      u.x &= (t & 0x00010001u) * 0xffffu;
      u.y &= ((t >> 1) & 0x00010001u) * 0xffffu;
      const int st = wn * 2 + (j >> 1), ch = (j & 1) * 2 + (kb >> 1);
      *(uint2*)(An + st * FF_STAGE + (i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
    }
  }
}

template <int F16>
__global__ __launch_bounds__(256, 1) void k_field_bwd(const FieldBwd P) {
  typedef typename FfOps<F16>::x8_t x8_t;
  __shared__ __attribute__((aligned(16))) unsigned char act[2][FF_ACT];
  __shared__ float appsum[16][64];          // per-lane-row partial column sums of the appearance block (<= 64 columns used)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kb = lane >> 4;
  const int swz = 3 * ((r16 >> 2) & 1);
  const int frag_off = r16 * 64 + ((kb ^ swz) << 4);
  const int ntile = P.M / FF_ROWS, G = (int)gridDim.x;
  const int a0c = 16 + P.ngeo, a1c = a0c + P.napp;      // appearance columns of the head input
  x8_t c1r[8][4], c0r[8][2], w1r[4][4], w0r[8][1];
  ff_load_w<F16, 8, 4>(P.C1n, (unsigned)((wn * 64 + r16) * 256 + kb * 8) * 2u, 256, c1r);
  ff_load_w<F16, 8, 2>(P.C0n, (unsigned)((wn * 32 + r16) * 256 + kb * 8) * 2u, 256, c0r);
  ff_load_w<F16, 4, 4>(P.W1xn, (unsigned)((wn * 64 + r16) * 128 + kb * 8) * 2u, 128, w1r);
  ff_load_w<F16, 8, 1>(P.W0n, (unsigned)(((wn & 1) * 16 + r16) * 256 + kb * 8) * 2u, 256, w0r);      // (waves 0, 1: hash features 0..15 / 16..31)

  // a tile's inputs: this thread's 128 bytes of the G1 tile (chunk id = q * 256 + tid -> row id >> 5, 16-byte chunk id & 31) and the
  // lane's mask words of both masked layers; requested one tile ahead (during the dXh layer) and written to LDS in the last phase
  // (scalars, not an array: a loop-carried array assigned under a condition stays in scratch memory)
  uint4 g0, g1, g2, g3, g4, g5, g6, g7;
  uint32_t binH[2], binY[2], nH0 = 0u, nH1 = 0u, nY0 = 0u, nY1 = 0u;
  auto load_tile = [&](int m0_) {
    const char* src = (const char*)(P.G1 + (size_t)m0_ * 256);
    auto ld = [&](int q) { return *(const uint4*)(src + ff_fresh((unsigned)(((q * 256 + tid) >> 5) * 512 + ((q * 256 + tid) & 31) * 16))); };
    g0 = ld(0); g1 = ld(1); g2 = ld(2); g3 = ld(3); g4 = ld(4); g5 = ld(5); g6 = ld(6); g7 = ld(7);
    uint32_t b[2];
    ff_load_bits(P.bH0, m0_, wn, lane, b); nH0 = b[0]; nH1 = b[1];
    ff_load_bits(P.bY0, m0_, wn, lane, b); nY0 = b[0]; nY1 = b[1];
  };
  auto put_tile = [&]() {
    auto st = [&](int q, const uint4& v) {
      const int id = q * 256 + tid, row = id >> 5, cc = id & 31;
      *(uint4*)(act[0] + (cc >> 2) * FF_STAGE + row * 64 + (((cc & 3) ^ (3 * ((row >> 2) & 1))) << 4)) = v;
    };
    st(0, g0); st(1, g1); st(2, g2); st(3, g3); st(4, g4); st(5, g5); st(6, g6); st(7, g7);
  };
  if ((int)blockIdx.x < ntile) { load_tile((int)blockIdx.x * FF_ROWS); put_tile(); }
  __syncthreads();
  for (int t = blockIdx.x; t < ntile; t += G) {
    const int m0 = t * FF_ROWS;
    const bool has_next = t + G < ntile;
    binH[0] = nH0; binH[1] = nH1; binY[0] = nY0; binY[1] = nY1;
    // ---- G0 = (G1 c1^T) * relu'(H0) -> act[1] -------------------------------------------------------------------------------
    ff_layer256_bwd<F16, 8, 0>(act[0], frag_off, c1r, binH, wn, r16, kb, act[1], nullptr, nullptr, tid);
    __syncthreads();
    // ---- dXh = G0 c0^T (this wave: head-input columns [32 wn, 32 wn + 32)); G0 -> HBM rides here ------------------------------
    if (has_next) load_tile(m0 + G * FF_ROWS);
    {
      ff_f32x4_t acc[4][4];
      ff_acc_zero<2>(acc);
      ff_mma_r<F16, 8, 2, 8>(act[1], frag_off, c0r, acc, act[1], (char*)(P.G0 + (size_t)m0 * 256), tid);
      // Gb tile (head-input column order) -> act[0] stages 0..3: geo columns from the registers; column 0 = d_raw, zeros elsewhere
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = wn * 32 + j * 16 + kb * 4;
        const bool geo = n >= 16 && n < a0c;
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 16 + r16;
          if (geo) {
            uint2 u;
            u.x = ff_cvt_pk<F16>(acc[i][j][0], acc[i][j][1]); u.y = ff_cvt_pk<F16>(acc[i][j][2], acc[i][j][3]);
            *(uint2*)(act[0] + ((n >> 5) * FF_STAGE + row * 64 + ((((n & 31) >> 3) ^ swz) << 4) + (n & 4) * 2)) = u;
          }
          // (the appearance block sums the 16-bit rounded gradient, as the stand-alone kernel reads it)
          const uint32_t p0 = ff_cvt_pk<F16>(acc[i][j][0], acc[i][j][1]), p1 = ff_cvt_pk<F16>(acc[i][j][2], acc[i][j][3]);
          cs[0] += FfOps<F16>::lo(p0); cs[1] += FfOps<F16>::hi(p0); cs[2] += FfOps<F16>::lo(p1); cs[3] += FfOps<F16>::hi(p1);
        }
        if (n >= a0c && n < a1c && P.d_embedding) *(float4*)&appsum[r16][n - a0c] = make_float4(cs[0], cs[1], cs[2], cs[3]);
      }
      // per row: chunk (stage 0, 0) = [d_raw, 0 x 7]; zero chunks for columns 8..15 and [16 + ngeo, 128)
      {
        const int row = tid >> 2, part = tid & 3, sw = 3 * ((row >> 2) & 1);
        unsigned char* rb = act[0] + row * 64;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        if (part == 0) {
          const float raw = FfOps<F16>::lo((uint32_t)*(const uint16_t*)((const char*)(P.raw + m0) + ff_fresh((unsigned)row * 2u)));
          const unsigned ro = ff_fresh((unsigned)row * 4u);
          const float dr = *(const float*)((const char*)(P.d_density + m0) + ro) * nf_density_slope(raw, P.dact, P.dbias) *
                           *(const float*)((const char*)(P.sel + m0) + ro);
          *(uint4*)(rb + ((0 ^ sw) << 4)) = make_uint4(ff_cvt_pk<F16>(dr, 0.f), 0u, 0u, 0u);
          *(uint4*)(rb + ((1 ^ sw) << 4)) = z;
        }
        // columns [a0c, 128): 8-column chunks (a0c is a multiple of 8 here: the launcher checks ngeo % 8 == 0)
        for (int c8 = a0c / 8 + part; c8 < 16; c8 += 4) *(uint4*)(rb + (c8 >> 2) * FF_STAGE + (((c8 & 3) ^ sw) << 4)) = z;
      }
    }
    __syncthreads();
    // ---- appearance-embedding gradient of the tile's ray; Gy0 = (Gb W1x) * relu'(Y0) -> act[1]; Gb -> HBM rides here ---------------
    if (P.d_embedding && tid < P.napp) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += appsum[r][tid];
      const int ray = m0 / P.S;
      atomicAdd(P.d_embedding + (size_t)P.embed_idx[ray] * P.napp + tid, a);
    }
    ff_layer256_bwd<F16, 4, 4>(act[0], frag_off, w1r, binY, wn, r16, kb, act[1], act[0], (char*)(P.Gb + (size_t)m0 * 128), tid);
    __syncthreads();
    // ---- dX0 = Gy0 w0^T (waves 0, 1: 16 hash features each) straight to HBM; Gy0 -> HBM rides here; the next tile's G1 -> act[0]
    if (has_next) put_tile();
    if (wn < 2) {
      ff_f32x4_t acc[4][4];
      ff_acc_zero<1>(acc);
      ff_mma_r<F16, 8, 1, 8>(act[1], frag_off, w0r, acc, act[1], (char*)(P.Gy0 + (size_t)m0 * 256), tid);
      if (P.dx_f32) {      // fp32 feature gradients: in the half mode a scaled gradient below 6e-8 would flush to zero here
        char* dst = (char*)((float*)P.dX0 + (size_t)m0 * P.ldx0);
        const unsigned xo = ff_fresh((unsigned)(r16 * P.ldx0 + wn * 16 + kb * 4) * 4u);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *(float4*)(dst + (xo + (unsigned)(i * 16 * P.ldx0) * 4u)) = make_float4(acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3]);
      } else {
        char* dst = (char*)(P.dX0 + (size_t)m0 * P.ldx0);
        const unsigned xo = ff_fresh((unsigned)(r16 * P.ldx0 + wn * 16 + kb * 4) * 2u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint2 u;
          u.x = ff_cvt_pk<F16>(acc[i][0][0], acc[i][0][1]); u.y = ff_cvt_pk<F16>(acc[i][0][2], acc[i][0][3]);
          *(uint2*)(dst + (xo + (unsigned)(i * 16 * P.ldx0) * 2u)) = u;
        }
      }
    } else {
      ff_copy_out<8>(act[1], (char*)(P.Gy0 + (size_t)m0 * 256), tid);
    }
    __syncthreads();      // the next tile's G1 is complete in act[0]; act[1] (Gy0) is free for its G0
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

#ifdef FF_TRACE
extern "C" int hugs_ff_trace_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_trace_buf), sizeof(long long) * 256); }
#endif

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
                                 float* rgb, int density_act, float density_bias, void* stream) {
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
  P.M = (int)M; P.S = S; P.ldx0 = ldx0; P.ldw0 = ldw0; P.ngeo = ngeo; P.dact = density_act; P.dbias = density_bias;
  P.X0 = (const uint16_t*)X0;
  P.W0t = (const uint16_t*)W0t; P.W1t = (const uint16_t*)W1x; P.C0t = (const uint16_t*)C0t; P.C1t = (const uint16_t*)C1t;
  P.b0 = b0; P.b1 = b1; P.cb0 = cb0; P.cb1 = cb1; P.c2 = c2; P.cb2 = cb2; P.tmpl = (const uint16_t*)tmpl; P.sel = sel;
  P.Y0 = (uint16_t*)Y0; P.raw = (uint16_t*)raw; P.Xh = (uint16_t*)Xh; P.H0 = (uint16_t*)H0; P.H1 = (uint16_t*)H1;
  P.bY0 = bY0; P.bH0 = bH0; P.density = density; P.rgb = rgb;
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
  const int ntile = (int)(M / FF_ROWS);
  const dim3 grid(ntile < ncu ? ntile : ncu), block(256);
  static int waves8 = -1;      // HUGS_FF_WAVES=4: the one-wave-per-SIMD kernel of round 4
  if (waves8 < 0) { const char* e = getenv("HUGS_FF_WAVES"); waves8 = !(e && e[0] == '4'); }
  if (waves8) {
    if (dtype == 2) hipLaunchKernelGGL(k_field_fwd8<1>, grid, dim3(512), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(k_field_fwd8<0>, grid, dim3(512), 0, (hipStream_t)stream, P);
  }
  else if (dtype == 2) hipLaunchKernelGGL(k_field_fwd<1>, grid, block, 0, (hipStream_t)stream, P);
  else hipLaunchKernelGGL(k_field_fwd<0>, grid, block, 0, (hipStream_t)stream, P);
  HUGS_CHECK_LAUNCH("hugs_nf_field_fwd");
  return 0;
}

// include/hugs.h hugs_nf_field_bwd
extern "C" int hugs_nf_field_bwd(int dtype, long long M, int S, const void* G1, const void* C1n, const void* C0n, const void* W1xn,
                                 const void* W0n, const uint32_t* bH0, const uint32_t* bY0, const float* d_density, const float* sel,
                                 const void* raw, int ngeo, int napp, const int* embed_idx, void* G0, void* Gb, void* Gy0, void* dX0,
                                 int ldx0, float* d_embedding, int dx_f32, int density_act, float density_bias, void* stream) {
  HUGS_REQUIRE(dtype == 1 || dtype == 2, -2, "hugs_nf_field_bwd: 16-bit operands only (dtype 1 = bf16, 2 = half), got %d", dtype);
  HUGS_REQUIRE(M > 0 && M % 256 == 0 && M < (1ll << 31) && S > 0 && S % FF_ROWS == 0 && M % S == 0, -3,
               "hugs_nf_field_bwd: %lld rows (a positive multiple of 256), rays of %d samples (a multiple of 64)", M, S);
  HUGS_REQUIRE(ngeo >= 0 && ngeo % 8 == 0 && napp >= 0 && napp <= 64 && napp % 4 == 0 && 16 + ngeo + napp <= 128 && ldx0 >= 32 && ldx0 % 8 == 0 && ldx0 <= 4096, -3,
               "hugs_nf_field_bwd: %d geo (a multiple of 8) / %d appearance (a multiple of 4, <= 64) columns, feature pitch %d", ngeo, napp, ldx0);
  HUGS_REQUIRE(G1 && C1n && C0n && W1xn && W0n && bH0 && bY0 && d_density && sel && raw && G0 && Gb && Gy0 && dX0 && (!d_embedding || napp == 0 || embed_idx), -2,
               "hugs_nf_field_bwd: null pointer");
  FieldBwd P;
  P.M = (int)M; P.S = S; P.ldx0 = ldx0; P.ngeo = ngeo; P.napp = napp; P.dx_f32 = dx_f32 ? 1 : 0; P.dact = density_act; P.dbias = density_bias;
  P.G1 = (const uint16_t*)G1; P.C1n = (const uint16_t*)C1n; P.C0n = (const uint16_t*)C0n; P.W1xn = (const uint16_t*)W1xn; P.W0n = (const uint16_t*)W0n;
  P.bH0 = bH0; P.bY0 = bY0; P.d_density = d_density; P.sel = sel; P.raw = (const uint16_t*)raw; P.embed_idx = embed_idx;
  P.G0 = (uint16_t*)G0; P.Gb = (uint16_t*)Gb; P.Gy0 = (uint16_t*)Gy0; P.dX0 = (uint16_t*)dX0; P.d_embedding = napp ? d_embedding : nullptr;
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
  const int ntile = (int)(M / FF_ROWS);
  const dim3 grid(ntile < ncu ? ntile : ncu), block(256);
  if (dtype == 2) hipLaunchKernelGGL(k_field_bwd<1>, grid, block, 0, (hipStream_t)stream, P);
  else hipLaunchKernelGGL(k_field_bwd<0>, grid, block, 0, (hipStream_t)stream, P);
  HUGS_CHECK_LAUNCH("hugs_nf_field_bwd");
  return 0;
}

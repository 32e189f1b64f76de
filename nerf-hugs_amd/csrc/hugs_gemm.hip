// MFMA GEMM family for the PropMLP / NerfMLP trunks (reference: MipNeRF360/internal/models.py:451-456
// trunk Dense+relu+skip-concat, :475 bottleneck, :508-512 view layer; their backward is what
// jax.value_and_grad derives in train_utils.py:454).
//
//   NT : C[M,N]  = epi( [A1|A2][M,K1+K2] * Bt[N,K1+K2]^T )   forward layers and dX (both operands K-contiguous)
//   TN : D[Kc,N] = sum_m X[m,Kc]^T * G[m,N]                  weight gradients, split over M into fp32 slabs
//
// bf16 path: 128x128x64 tiles, 4 waves (2x2), v_mfma_f32_16x16x32_bf16, operands staged with
// global_load_lds (16 B/lane DMA, no VGPR round trip) into XOR-swizzled LDS (swizzle applied on the
// per-lane SOURCE address, LDS image stays lane-linear), double buffered.  The MFMA operands are swapped
// (weights as A, activations as B) so each lane ends up with 4 consecutive output columns of one row:
// row-major stores without a cross-lane transpose.  TN reads both operands with ds_read_b64_tr_b16
// (hardware transpose) from row-major tiles.  Workgroup ids are remapped so each XCD owns a contiguous
// band of row tiles (activations stream once through that XCD's L2, weights stay L2 resident).
// fp32 path (parity mode): same tiling on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
#include "hugs_common.h"
#include <type_traits>

// The 16-bit operand kernels are compiled twice from this one source: as written (bf16, dtype 1) and, through
// hugs_gemm_f16.hip (#define HUGS_GEMM_F16, #include "hugs_gemm.hip"), with IEEE half operands (dtype 2: the nerfacto
// path's fp16 mode, the reference's `enable_amp`).  Only the MFMA opcode, the transpose-read builtin, the fp32 -> operand
// conversion and the constant 1.0 differ; relu (signed 16-bit max), the "> 0" mask tests and the 1-bit masks are sign /
// magnitude tests that hold for both formats.  The names keep their bf16 spelling in both passes.
#ifdef HUGS_GEMM_F16
typedef _Float16 hugs_op_t;
#define HUGS_GEMM_NS gemm_f16
#define HUGS_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
__device__ __forceinline__ uint16_t f_to_op16(float f) { const _Float16 h = (_Float16)f; return *(const uint16_t*)&h; }
#else
typedef __bf16 hugs_op_t;
#define HUGS_GEMM_NS gemm_bf16
#define HUGS_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
__device__ __forceinline__ uint16_t f_to_op16(float f) { return f_to_bf16(f); }
#endif
typedef __attribute__((ext_vector_type(8))) hugs_op_t bf16x8_t;
typedef __attribute__((ext_vector_type(4))) hugs_op_t bf16x4_t;
// (the transposing LDS read moves 16-bit patterns: the bf16-typed builtin serves both formats)
typedef __attribute__((ext_vector_type(4))) __bf16 hugs_raw16x4_t;
#define HUGS_DS_READ_TR16(p_) __builtin_bit_cast(bf16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((hugs_raw16x4_t __attribute__((address_space(3)))*)(p_)))
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace HUGS_GEMM_NS {

struct GemmEpi {
  const float* bias;      // [N] added per column, or null
  const float* row_bias;  // [M/row_div, ld_rb] added per row group (per-ray bias), or null
  int row_div, ld_rb;
  int relu;               // max(0, .)
  const void* mask;       // activation tensor [M, ld_mask] (same dtype as out): multiply by (mask > 0)
  int ld_mask;
  const float* r1_row;    // rank-1 term r1_row[m] * r1_col[n], or null
  const float* r1_col;
  void* out;              // [M, ldc]
  int ldc;
  // 1-bit relu masks in the 256x256 kernels' own register layout: per tile (row-major tile index), wave and lane one
  // 4 words = 128 bits = (fragment row i 0..7) x (fragment column j 0..3) x (4 values); bit = that output is > 0.
  // bits_out: written by a relu epilogue; bits_in: multiplies the output by the bit (the backward's relu mask, in
  // place of re-reading the bf16 activation: 16 B per lane instead of 256 B).
  uint32_t* bits_out;
  const uint32_t* bits_in;
};

// Per-phase timestamps / staggered starts for the measurement builds live in scratch/hugs_gemm_trace.h (scratch/build_trace.sh
// compiles with -DHUGS_TRACE -I scratch); the product build sees empty macros.
#ifdef HUGS_TRACE
#include "hugs_gemm_trace.h"
#else
#define HUGS_TR(i)
#define HUGS_TRP(i, k)
#define HUGS_TRH(i, h)
#define HUGS_TRH_DECL
#define HUGS_TRH_END()
#define HUGS_TR_ID()
#define HUGS_STAGGER()
#endif
#ifndef HUGS_EPI_STORE
#define HUGS_EPI_STORE(v_, p_) __builtin_nontemporal_store(v_, p_)
#endif

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // contiguous band of tiles per XCD; bijective for any nwg (cdna guide T1)
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <bool BF16OUT>
__device__ __forceinline__ void epi_store4(const GemmEpi& E, int m, int n, f32x4_t v) {
  float x[4] = {v[0], v[1], v[2], v[3]};
  if (E.bias) { const float4 b = *(const float4*)(E.bias + n); x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w; }
  if (E.row_bias) {
    const float4 b = *(const float4*)(E.row_bias + (size_t)(m / E.row_div) * E.ld_rb + n);
    x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
  }
  if (E.r1_row) {
    const float r = E.r1_row[m];
    const float4 c = *(const float4*)(E.r1_col + n);
    x[0] += r * c.x; x[1] += r * c.y; x[2] += r * c.z; x[3] += r * c.w;
  }
  if (E.relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
  if (BF16OUT) {
    if (E.mask) {
      const uint2 mk = *(const uint2*)((const uint16_t*)E.mask + (size_t)m * E.ld_mask + n);
      // bf16 > 0  <=>  sign clear and not zero
      if (!((mk.x & 0x7fffu) && !(mk.x & 0x8000u))) x[0] = 0.f;
      if (!((mk.x >> 16 & 0x7fffu) && !(mk.x >> 31))) x[1] = 0.f;
      if (!((mk.y & 0x7fffu) && !(mk.y & 0x8000u))) x[2] = 0.f;
      if (!((mk.y >> 16 & 0x7fffu) && !(mk.y >> 31))) x[3] = 0.f;
    }
    uint2 pk;
    pk.x = f_to_op16(x[0]) | ((uint32_t)f_to_op16(x[1]) << 16);
    pk.y = f_to_op16(x[2]) | ((uint32_t)f_to_op16(x[3]) << 16);
    *(uint2*)((uint16_t*)E.out + (size_t)m * E.ldc + n) = pk;
  } else {
    if (E.mask) {
      const float4 mk = *(const float4*)((const float*)E.mask + (size_t)m * E.ld_mask + n);
      if (!(mk.x > 0.f)) x[0] = 0.f;
      if (!(mk.y > 0.f)) x[1] = 0.f;
      if (!(mk.z > 0.f)) x[2] = 0.f;
      if (!(mk.w > 0.f)) x[3] = 0.f;
    }
    *(float4*)((float*)E.out + (size_t)m * E.ldc + n) = make_float4(x[0], x[1], x[2], x[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 NT
// ------------------------------------------------------------------------------------------------
#define GB_BM 128
#define GB_BN 128
#define GB_BK 64
#define GB_TILE_BYTES (128 * 64 * 2)  // 16 KiB per operand tile

__global__ __launch_bounds__(256, 2) void k_gemm_nt_bf16(int M, int N, int K1, int K2, const uint16_t* __restrict__ A1,
                                                          int lda1, const uint16_t* __restrict__ A2, int lda2,
                                                          const uint16_t* __restrict__ Bt, int ldb, GemmEpi E) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * GB_TILE_BYTES];  // [buf][A|B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / GB_BN, ntm = M / GB_BM;
  const int t = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (t / ntn) * GB_BM, n0 = (t % ntn) * GB_BN;
  const int wm = wv >> 1, wn = wv & 1;
  const int nk = (K1 + K2) / GB_BK;

  // staging: tile = 128 rows x 8 chunks(16 B).  chunk id p = it*256 + tid -> row p/8, physical pos p%8,
  // which holds logical chunk (pos ^ (row&7)).
  auto stage = [&](int kt, int buf) {
    const int kglob = kt * GB_BK;
    const uint16_t* Abase; int lda, kcol;
    if (kglob < K1) { Abase = A1; lda = lda1; kcol = kglob; } else { Abase = A2; lda = lda2; kcol = kglob - K1; }
    unsigned char* la = lds + buf * 2 * GB_TILE_BYTES;
    unsigned char* lb = la + GB_TILE_BYTES;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int p = it * 256 + tid;
      const int row = p >> 3, pos = p & 7, c = pos ^ (row & 7);
      glds16(Abase + (size_t)(m0 + row) * lda + kcol + c * 8, la + (it * 256 + wv * 64) * 16);
      glds16(Bt + (size_t)(n0 + row) * ldb + kglob + c * 8, lb + (it * 256 + wv * 64) * 16);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  const int r16 = lane & 15, kb = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    __syncthreads();  // drains the outstanding LDS-DMA of tile kt and fences the previous compute
    if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
    const unsigned char* la = lds + buf * 2 * GB_TILE_BYTES;
    const unsigned char* lb = la + GB_TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t xa[4], wb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + r16;
        xa[i] = *(const bf16x8_t*)(la + row * 128 + (((kk * 4 + kb) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + r16;
        wb[j] = *(const bf16x8_t*)(lb + row * 128 + (((kk * 4 + kb) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = HUGS_MFMA_16X16X32(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      epi_store4<true>(E, m0 + wm * 64 + i * 16 + r16, n0 + wn * 64 + j * 16 + kb * 4, acc[i][j]);
}

// ------------------------------------------------------------------------------------------------
// bf16 NT, large tile: 256x256 output tile, 8 waves (2 x 4, each 128 x 64 = 8 x 4 MFMA fragments), K walked
// in 32-wide stages through a 4-deep LDS ring (4 x 32 KiB).  Three stages stay in flight across the
// per-stage s_barrier (counted s_waitcnt vmcnt, raw barrier -- a __syncthreads would drain the LDS-DMA
// queue).  Versus the 128^2 kernel: half the L2->LDS bytes and 3/4 of the LDS-read bytes per flop.
// Epilogue: bias / rank-1 / relu in registers -> bf16 tile staged in the (now free) ring with a 528-byte
// row pitch -> whole 512-byte rows stored 16 B per lane, the relu mask of the backward pass applied on
// the coalesced side.
// ------------------------------------------------------------------------------------------------
typedef unsigned __attribute__((ext_vector_type(2))) u32x2_t;
#define GL_CPAD 16   // bytes added to each row of the staged C tile (bank spread for the 8-byte fragment writes)

// WN = wave columns: 4 -> 256x256 tile, 8 waves, 4-slot ring (128 KiB, 1 workgroup/CU);
//                    2 -> 256x128 tile, 4 waves, 3-slot ring (72 KiB, 2 workgroups/CU).  The launch bound must
//                         ask for 2 waves/SIMD here too: with "1" the compiler spread the kernel over 366
//                         registers (AGPRs included), only ONE workgroup fitted a CU and it ran at 610 instead
//                         of 850 TFLOP/s.
// EPI: -1 = every epilogue term decided at run time (GemmEpi pointers); >= 0 = compile-time set of terms
// (bit 0 bias, 1 relu, 2 mask, 3 rank-1; never a per-ray row bias): the run-time form compiles to ~60 branches, each
// with its own global load + s_waitcnt vmcnt(0) -- a chain of serialised L2 latencies (9.6k cycles per tile by
// s_memtime, scratch/nt_trace.py) -- while the specialised forms issue all their loads in one batch.
#define EPI_BIAS 1
#define EPI_RELU 2
#define EPI_MASK 4
#define EPI_R1 8
#define EPI_BIN 16    // 1-bit relu mask read (GemmEpi.bits_in)
#define EPI_BOUT 32   // 1-bit relu mask written (GemmEpi.bits_out)
// Epilogue of the 256x256 NT tile straight from the accumulator registers (wave (wm, wn) owns rows wm*128.. and
// columns wn*64..; lane (r16, kb) of fragment (i, j) holds row i*16 + r16, columns j*16 + kb*4 .. +3).
// lds_bias / lds_r1col: optional LDS-resident copies of E.bias / E.r1_col (indexed by absolute column): the persistent
// kernel keeps them there so that the epilogue issues no global load that would queue behind the prefetched stages.
// The epilogue is VALU-bound (two waves per SIMD walk 128 accumulator registers each: ~1300 vector instructions per wave
// cost ~10k of a tile's ~50k cycles), so it is written for instruction count: the bias arrives inside the accumulators
// when BIAS_IN_ACC (the persistent kernel initialises them with it), relu is one v_pk_max_i16 per bf16 PAIR after the
// conversion (sign-magnitude bf16 read as int16: negative <=> sign bit; rounding commutes with the clamp), the 1-bit masks
// are made / applied on the packed pairs (3-4 instructions per pair), terms that are absent are not added as zeros.
typedef short __attribute__((ext_vector_type(2))) i16x2_t;
__device__ __forceinline__ uint32_t pk_relu_bf16(uint32_t u) {
  const i16x2_t z = {0, 0};
  const i16x2_t r = __builtin_elementwise_max(*(const i16x2_t*)&u, z);
  return *(const uint32_t*)&r;
}
typedef unsigned short __attribute__((ext_vector_type(2))) u16x2_t;
// min(half, 1) of both 16-bit halves: ONE v_pk_min_u16 (the inline constant 1 feeds both halves: op_sel_hi 0 on it).  Written as
// asm: from __builtin_elementwise_min the compiler makes two compares, two selects and a v_perm per pair.
__device__ __forceinline__ uint32_t pk_min1_u16(uint32_t a) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t pk_mul_u16(uint32_t a, uint32_t b) {
  const u16x2_t r = *(const u16x2_t*)&a * *(const u16x2_t*)&b;
  return *(const uint32_t*)&r;
}
typedef float __attribute__((ext_vector_type(2))) f32x2_t;
typedef hugs_op_t __attribute__((ext_vector_type(2))) bf16x2_t;
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {      // one v_cvt_pk_bf16_f32
  const f32x2_t f = {a, b};
  const bf16x2_t h = __builtin_convertvector(f, bf16x2_t);
  return *(const uint32_t*)&h;
}
// (after_row(i): called once row block i of the accumulators has been converted and its stores issued -- k_gemm_nt_bf16_p64 starts the
//  NEXT tile's first MFMAs on that row block there, under the rest of the epilogue's vector work)
struct NtNoRowFn { __device__ __forceinline__ void operator()(f32x4_t (&)[8][4], int) const {} };
template <int EPI, bool BIAS_IN_ACC = false, int FRESH_LANE = 0, int MASK_AHEAD = 2, class RowFn = NtNoRowFn>
__device__ __forceinline__ void nt_epilogue_direct(f32x4_t (&acc)[8][4], const GemmEpi& E, int m0, int n0, int wm, int wn,
                                                   int r16, int kb, const float* lds_bias, const float* lds_r1col,
                                                   RowFn after_row = RowFn()) {
    // ---- epilogue straight from registers: bias / rank-1 / relu, v_cvt_pk_bf16_f32, one v_permlane16_swap pair
    // per two neighbouring 16-column fragments turns the 8-byte-per-lane MFMA layout into 16 contiguous bytes per
    // lane (64-byte runs per row): no LDS round trip, no barriers.
    constexpr bool GEN = EPI < 0;
    const bool has_bias = GEN ? E.bias != nullptr : bool(EPI & EPI_BIAS);
    const bool has_relu = GEN ? E.relu != 0 : bool(EPI & EPI_RELU);
    const bool has_mask = GEN ? E.mask != nullptr : bool(EPI & EPI_MASK);
    const bool has_r1 = GEN ? E.r1_row != nullptr : bool(EPI & EPI_R1);
    const bool has_rowb = GEN ? E.row_bias != nullptr : false;
    // lane-private mask words (see GemmEpi): word = i >> 1; packed bf16 pair k = (i & 1) * 8 + j * 2 + (c >> 1) of the word
    // owns bit k (even c) and bit 16 + k (odd c)
    // layout [tile][wave][word 0..3][lane]: every word is one 256-byte wave store / load
    // (FRESH_LANE: the five-slot kernel has no register left to carry a per-lane 64-bit address across its main loop -- the lane
    // index is re-derived from an opaque copy of the thread index here, so that nothing of it is live before the epilogue)
    int lane_ = threadIdx.x & 63;
    if (FRESH_LANE == 1) { int t_ = threadIdx.x; asm volatile("" : "+v"(t_)); lane_ = t_ & 63; }
    // (2, the layer-chained kernel: the lane index from mbcnt inside a volatile asm -- not even threadIdx's register stays live)
    if (FRESH_LANE == 2) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
    // (the bf16-mask specialisations' load / store geometry -- ccol, srow below -- is loop invariant in the persistent kernel and was
    //  hoisted out of its tile loop into scratch: derive it from the opaque lane index there)
    if (FRESH_LANE && EPI >= 0 && (EPI & EPI_MASK)) { r16 = lane_ & 15; kb = lane_ >> 4; }
    const size_t bits_at = (((size_t)(m0 >> 8) * (size_t)(E.ldc >> 8) + (size_t)(n0 >> 8)) * 8 + (size_t)(wm * 4 + wn)) * 256 + (size_t)lane_;
    const bool has_bin = GEN ? E.bits_in != nullptr : bool(EPI & EPI_BIN);
    const bool has_bout = GEN ? E.bits_out != nullptr : bool(EPI & EPI_BOUT);
    uint32_t bin[4] = {0u, 0u, 0u, 0u}, bw = 0u;
    if (has_bin) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bin[q] = E.bits_in[bits_at + q * 64];
    }
    float4 bj[4], cj[4];
    float r1v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + kb * 4;
      bj[j] = has_bias ? (lds_bias ? *(const float4*)(lds_bias + n) : *(const float4*)(E.bias + n)) : make_float4(0.f, 0.f, 0.f, 0.f);
      cj[j] = has_r1 ? (lds_r1col ? *(const float4*)(lds_r1col + n) : *(const float4*)(E.r1_col + n)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r1v[i] = has_r1 ? E.r1_row[m0 + wm * 128 + i * 16 + r16] : 0.f;
    // Store geometry.  After the permlane16 swaps lane (r16, kb) holds, for each jp, 16 contiguous bytes of row r16:
    // columns jp*32 + chunk(kb) .. +7.  Storing that directly writes 64-byte runs of 16 rows per instruction --
    // HALF cache lines, which cost 62 cycles of store issue each (scratch/store_bench.hip: 8.0k cycles per 128 KiB tile
    // and CU) against 18.5 for whole 128-byte lines (2.4k).  So lanes r16 and r16+8 trade one chunk (DPP row_ror:8):
    // afterwards lanes r16<8 hold the jp=0 chunks of rows r16 and r16+8, lanes r16>=8 the jp=1 chunks of rows r16-8
    // and r16, and every store instruction writes 8 rows x 128 B.
    const bool hi8 = (r16 & 8) != 0;
    const int srow = r16 & 7;                                            // row (within the 16-row group) of store A; B = +8
    const int ccol = n0 + wn * 64 + (kb & 1) * 16 + (kb >> 1) * 8 + (hi8 ? 32 : 0);
    char* const out_base = (char*)E.out + ((size_t)(m0 + wm * 128) * (size_t)E.ldc + (size_t)(n0 + wn * 64)) * 2;
    const unsigned out_voff = (unsigned)(srow * E.ldc + (ccol - n0 - wn * 64)) * 2u;
    // relu-mask chunks (same whole-line geometry as the stores), fetched MASK_AHEAD fragment rows ahead of their use: all 16
    // in flight at once would cost 64 registers on top of the 128 accumulators (the persistent kernel, whose loader state is live
    // across the epilogue, affords one row ahead: with two its bf16-mask specialisations spilled 2-3 registers)
    uint4 mkv[8][2];
    auto mask_load = [&](int i) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        mkv[i][h] = *(const uint4*)((const uint16_t*)E.mask + (size_t)(m0 + wm * 128 + i * 16 + h * 8 + srow) * E.ld_mask + ccol);
    };
    if (has_mask) {
#pragma unroll
      for (int a = 0; a < MASK_AHEAD; ++a) mask_load(a);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (has_mask && i + MASK_AHEAD < 8) mask_load(i + MASK_AHEAD);
      const int m = m0 + wm * 128 + i * 16 + r16;
      const float r1 = r1v[i];
      const float* rbp = has_rowb ? E.row_bias + (size_t)(m / E.row_div) * E.ld_rb + n0 + wn * 64 + kb * 4 : nullptr;
      uint32_t pk[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (has_bias && !BIAS_IN_ACC) { x[0] += bj[j].x; x[1] += bj[j].y; x[2] += bj[j].z; x[3] += bj[j].w; }
        if (has_r1) { x[0] += r1 * cj[j].x; x[1] += r1 * cj[j].y; x[2] += r1 * cj[j].z; x[3] += r1 * cj[j].w; }
        if (rbp) { const float4 b = *(const float4*)(rbp + j * 16); x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w; }
        uint2 u;
        u.x = cvt_pk_bf16(x[0], x[1]); u.y = cvt_pk_bf16(x[2], x[3]);
        if (has_relu) { u.x = pk_relu_bf16(u.x); u.y = pk_relu_bf16(u.y); }
        const int k = (i & 1) * 8 + j * 2;
        if (has_bin) {      // each 16-bit half times its bit (v_pk_mul_lo_u16): 3 instructions per packed pair (round 5; was 4)
          const uint32_t t = bin[i >> 1] >> k;
          u.x = pk_mul_u16(u.x, t & 0x00010001u);
          u.y = pk_mul_u16(u.y, (t >> 1) & 0x00010001u);
        }
        pk[j][0] = u.x; pk[j][1] = u.y;
        if (has_bout) {
          if (has_relu) {   // the relu left every half in [0, 0x7fff]: min(half, 1) IS the bit (v_pk_min_u16), one v_lshl_or_b32 files it
            bw |= pk_min1_u16(u.x) << k;
            bw |= pk_min1_u16(u.y) << (k + 1);
          } else {          // (unreachable through the C ABI: bits_out requires a relu epilogue) magnitude != 0: half + 0x7fff carries into bit 15
            const uint32_t ax = u.x & 0x7fff7fffu, ay = u.y & 0x7fff7fffu;
            bw |= ((ax + 0x7fff7fffu) >> (15 - k)) & (0x00010001u << k);
            bw |= ((ay + 0x7fff7fffu) >> (14 - k)) & (0x00010001u << (k + 1));
          }
        }
      }
      if (has_bout && (i & 1)) { E.bits_out[bits_at + (i >> 1) * 64] = bw; bw = 0u; }
      uint32_t vv[2][4];
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(pk[2 * jp][0], pk[2 * jp + 1][0], false, false);
        const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(pk[2 * jp][1], pk[2 * jp + 1][1], false, false);
        vv[jp][0] = s0[0]; vv[jp][1] = s1[0]; vv[jp][2] = s0[1]; vv[jp][3] = s1[1];
      }
      // low lanes give away their jp=1 chunk and receive the partner row's jp=0 chunk; high lanes the other way round:
      // two bank-masked DPP moves (row_ror:8 reads lane (l + 8) % 16; bank_mask 0xc / 0x3 writes lanes 8-15 / 0-7 only)
      uint32_t sa[4], sb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sa[q] = (uint32_t)__builtin_amdgcn_update_dpp((int)vv[0][q], (int)vv[1][q], 0x128, 0xf, 0xc, false);   // store A: row srow
        sb[q] = (uint32_t)__builtin_amdgcn_update_dpp((int)vv[1][q], (int)vv[0][q], 0x128, 0xf, 0x3, false);   // store B: row srow + 8
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t* vw = h ? sb : sa;
        if (has_mask) {
          const uint4 mk = mkv[i][h];
          const uint32_t mw[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t lo = mw[q] & 0xffffu, hi = mw[q] >> 16;
            const uint32_t keep = (((lo & 0x7fffu) && !(lo & 0x8000u)) ? 0x0000ffffu : 0u) |
                                  (((hi & 0x7fffu) && !(hi & 0x8000u)) ? 0xffff0000u : 0u);
            vw[q] &= keep;
          }
        }
        // streaming (nontemporal) stores: the 32 MB all workgroups write at the same time do not push the operand panels
        // out of the 4 MB L2s (in-step A/B: step -1.8 %, forward layer 264 -> 254 us; explicit sc0 / sc1 / nt policy
        // bits measured within noise of it)
        // address = (wave-uniform 64-bit base of row block (i, h): scalar arithmetic) + (ONE per-lane 32-bit byte offset, the same for
        // all 16 stores): the store takes the SGPR-base form and costs no vector address arithmetic (round 5: ~50 of the epilogue's
        // ~540 vector instructions were 64-bit per-lane address computations)
        { typedef unsigned __attribute__((ext_vector_type(4))) u32x4_t; const u32x4_t v_ = {vw[0], vw[1], vw[2], vw[3]};
          char* rb_ = out_base + (size_t)(i * 16 + h * 8) * (size_t)E.ldc * 2;
          u32x4_t* p_ = (u32x4_t*)(rb_ + out_voff);
          HUGS_EPI_STORE(v_, p_);
        }
      }
      after_row(acc, i);
    }
}

template <int WN, int EPI = -1>
__global__ __launch_bounds__(128 * WN, 2) void k_gemm_nt_bf16_big(
    int M, int N, int K1, int K2, const uint16_t* __restrict__ A1, int lda1, const uint16_t* __restrict__ A2, int lda2,
    const uint16_t* __restrict__ Bt, int ldb, GemmEpi E) {
  constexpr int NT = 128 * WN;                 // threads
  constexpr int TN_ = 64 * WN;                 // tile columns
  constexpr int NSLOT = WN == 4 ? 4 : 3;
  constexpr int A_BYTES = 256 * 64, B_BYTES = TN_ * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int AIT = 1024 / NT, BIT = (TN_ * 4) / NT;   // 16-byte chunks per thread per stage
  constexpr int CPITCH = TN_ * 2 + GL_CPAD;
  constexpr int LDS_BYTES = (NSLOT * STAGE) > (256 * CPITCH) ? (NSLOT * STAGE) : (256 * CPITCH);
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / TN_, ntm = M >> 8;
  const int t = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (t / ntn) << 8, n0 = (t % ntn) * TN_;
  const int wm = wv / WN, wn = wv % WN;
  const int ns = (K1 + K2) >> 5;

  // staging: rows of 4 chunks(16 B); chunk id p = it*NT + tid -> row p>>2, physical pos p&3 holding logical
  // chunk pos ^ (3*((row>>2)&1)) (keeps the fragment ds_read_b128 conflict-free).
  // Every LDS-DMA takes its global address as (64-bit SGPR base) + (32-bit per-lane byte offset), as the persistent kernel's
  // does: chunk p = it*NT + tid sits in row it*(NT/4) + (tid >> 2) -- NT/4 is a multiple of 8, so the swizzle bit (row >> 2) & 1
  // and with it the per-lane offset do not depend on `it` (3 VGPRs for the whole kernel), and the base is scalar arithmetic.
  // (Round 5: the builtin's 64-bit per-lane addresses cost 3 VALU + a VGPR pair per load; every one-tile specialisation spilled
  // 16-84 bytes per lane around them.)
  const int prow = tid >> 2, pcol = ((tid & 3) ^ (3 * ((prow >> 2) & 1))) * 8;
  const unsigned oA1 = (unsigned)(prow * lda1 + pcol) * 2u, oA2 = (unsigned)(prow * lda2 + pcol) * 2u;
  const unsigned oB = (unsigned)(prow * ldb + pcol) * 2u;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)wv * 1024u;
  auto dma = [&](const char* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  };
  auto stage = [&](int st) {
    const int kglob = st << 5;
    const unsigned la = lds_base + (unsigned)(st % NSLOT) * STAGE;
    if (kglob < K1) {
      const char* ab = (const char*)A1 + ((size_t)m0 * lda1 + kglob) * 2;
#pragma unroll
      for (int it = 0; it < AIT; ++it) dma(ab + (size_t)lda1 * 2 * (it * (NT / 4)), oA1, la + it * NT * 16);
    } else {
      const char* ab = (const char*)A2 + ((size_t)m0 * lda2 + (kglob - K1)) * 2;
#pragma unroll
      for (int it = 0; it < AIT; ++it) dma(ab + (size_t)lda2 * 2 * (it * (NT / 4)), oA2, la + it * NT * 16);
    }
    const char* bb = (const char*)Bt + ((size_t)n0 * ldb + kglob) * 2;
#pragma unroll
    for (int it = 0; it < BIT; ++it) dma(bb + (size_t)ldb * 2 * (it * (NT / 4)), oB, la + A_BYTES + it * NT * 16);
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, kb = lane >> 4;
  const int frag_off = r16 * 64 + ((kb ^ (3 * ((r16 >> 2) & 1))) << 4);

  // Fragments are double-buffered in registers: the ds_reads of stage st+1 are issued before the MFMAs of
  // stage st, so LDS latency/bandwidth hides under the matrix pipe inside each wave (all waves of a
  // workgroup are barrier-locked to the same phase, so there is no other wave to hide it under).
  struct Frags { bf16x8_t wb[4], xa[8]; };
  auto load_frags = [&](Frags& f, int st) {
    const unsigned char* la = lds + (st % NSLOT) * STAGE + (wm * 128) * 64 + frag_off;
    const unsigned char* lb = lds + (st % NSLOT) * STAGE + A_BYTES + (wn * 64) * 64 + frag_off;
#pragma unroll
    for (int j = 0; j < 4; ++j) f.wb[j] = *(const bf16x8_t*)(lb + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) f.xa[i] = *(const bf16x8_t*)(la + i * 16 * 64);
  };
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = HUGS_MFMA_16X16X32(f.wb[j], f.xa[i], acc[i][j], 0, 0, 0);
  };
  constexpr int G = AIT + BIT;   // LDS-DMA instructions per thread per stage
  // iteration st: frags(st) are in `cur`; make stage st+1 visible, refill slot st%NSLOT with stage st+NSLOT,
  // start reading frags(st+1) into `nxt`, then run the MFMAs of stage st.
#define GL_ITER(cur, nxt, st, VM)                                                        \
  {                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_s_barrier();                                                         \
    asm volatile("" ::: "memory");                                                        \
    if ((st) + NSLOT < ns) stage((st) + NSLOT);                                           \
    load_frags(nxt, (st) + 1);                                                            \
    mfmas(cur);                                                                           \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {   /* 8 MFMA : 3 ds_read interleave (A/B: +5 %) */ \
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                  \
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                  \
    }                                                                                     \
  }
  Frags f0, f1;
  HUGS_STAGGER()
  HUGS_TR(0) HUGS_TR_ID()
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) stage(q);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  HUGS_TR(1)
  load_frags(f0, 0);
  int st = 0;
  if (WN == 4) {
    for (; st + 5 < ns; st += 2) { GL_ITER(f0, f1, st, 8) GL_ITER(f1, f0, st + 1, 8) }   // st = 0 .. ns-5
    GL_ITER(f0, f1, st, 8)                                                                // st = ns-4
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    load_frags(f0, st + 2); mfmas(f1);                                                    // ns-3
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    load_frags(f1, st + 3); mfmas(f0);                                                    // ns-2
    mfmas(f1);                                                                            // ns-1
  } else {
    for (; st + 3 < ns; st += 2) { GL_ITER(f0, f1, st, 6) GL_ITER(f1, f0, st + 1, 6) }   // st = 0 .. ns-3
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    load_frags(f1, st + 1); mfmas(f0);                                                    // ns-2
    mfmas(f1);                                                                            // ns-1
  }
#undef GL_ITER
  HUGS_TR(2)
  if (WN == 4) {     // register-direct epilogue (A/B: +10 % over staging the C tile through LDS)
    // (fenced off from the last MFMAs, lane index re-derived: scheduled into the tail iterations the epilogue's address arithmetic
    //  and first conversions pushed 5-21 registers -- an accumulator fragment among them -- into scratch in most specialisations)
    __builtin_amdgcn_sched_barrier(0);
    nt_epilogue_direct<EPI, false, true>(acc, E, m0, n0, wm, wn, r16, kb, nullptr, nullptr);
    HUGS_TR(3)
    return;
  }
  __syncthreads();   // everyone is done reading the ring: reuse it as the C staging tile

  // ---- epilogue: registers -> (bias, rank-1, relu) -> bf16 tile in LDS -> whole rows, 16 B per lane ----
  float4 bj[4], cj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + kb * 4;
    bj[j] = E.bias ? *(const float4*)(E.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    cj[j] = E.r1_row ? *(const float4*)(E.r1_col + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ml = wm * 128 + i * 16 + r16;
    const int m = m0 + ml;
    const float r1 = E.r1_row ? E.r1_row[m] : 0.f;
    const float* rbp = E.row_bias ? E.row_bias + (size_t)(m / E.row_div) * E.ld_rb + n0 + wn * 64 + kb * 4 : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nl = wn * 64 + j * 16 + kb * 4;
      float x[4] = {acc[i][j][0] + bj[j].x + r1 * cj[j].x, acc[i][j][1] + bj[j].y + r1 * cj[j].y,
                    acc[i][j][2] + bj[j].z + r1 * cj[j].z, acc[i][j][3] + bj[j].w + r1 * cj[j].w};
      if (rbp) { const float4 b = *(const float4*)(rbp + j * 16); x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w; }
      if (E.relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
      bf16x4_t pk;   // v_cvt_pk_bf16_f32 (round to nearest even)
      pk[0] = (hugs_op_t)x[0]; pk[1] = (hugs_op_t)x[1]; pk[2] = (hugs_op_t)x[2]; pk[3] = (hugs_op_t)x[3];
      *(bf16x4_t*)(lds + ml * CPITCH + nl * 2) = pk;
    }
  }
  __syncthreads();
  constexpr int CPR = TN_ / 8;                  // 16-byte chunks per row
  constexpr int EIT = 256 * CPR / NT;           // = 16
#pragma unroll 4
  for (int it = 0; it < EIT; ++it) {
    const int p = it * NT + tid;
    const int row = p / CPR, c = p % CPR;
    uint4 v = *(const uint4*)(lds + row * CPITCH + c * 16);
    if (E.mask) {
      const uint4 mk = *(const uint4*)((const uint16_t*)E.mask + (size_t)(m0 + row) * E.ld_mask + n0 + c * 8);
      const uint32_t mw[4] = {mk.x, mk.y, mk.z, mk.w};
      uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // bf16 > 0 <=> sign clear and magnitude non-zero
        const uint32_t lo = mw[q] & 0xffffu, hi = mw[q] >> 16;
        const uint32_t keep = (((lo & 0x7fffu) && !(lo & 0x8000u)) ? 0x0000ffffu : 0u) |
                              (((hi & 0x7fffu) && !(hi & 0x8000u)) ? 0xffff0000u : 0u);
        vw[q] &= keep;
      }
      v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
    }
    *(uint4*)((uint16_t*)E.out + (size_t)(m0 + row) * E.ldc + n0 + c * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 NT, PERSISTENT form of the 256x256 kernel: one workgroup per CU walks its tiles (bid = blockIdx.x + i*grid, the
// same XCD band as the one-tile-per-workgroup launch) and the 4-slot K-stage ring runs straight on from one tile into
// the next.  s_memtime bracketing of the one-tile kernel (scratch/nt_trace.py, trunk shape) gave per tile: prologue
// 9.0k cycles (first stage of a cold ring, every CU fetching at once) + main loop 45k + epilogue 6.3k + store
// acknowledgement 2.9k + relaunch gap 2.2k.  Here the next tile's first four stages are already in flight when the
// epilogue starts, the stores drain under the next tile's first iterations, and there is no relaunch:
//   * the epilogue's 16 stores per lane sit in the SAME in-order vmcnt queue as the LDS-DMA stages, so the three
//     iterations that follow an epilogue wait with vmcnt(8 + 16): stage landed <=> at most the two younger stages AND
//     the 16 stores are still outstanding.  (Waiting with the steady-state vmcnt(8) there would drain the stores --
//     what made an earlier persistent attempt no faster.)
//   * bias / rank-1 column vectors live in LDS (the ring leaves 32 KiB): an epilogue global load would have to wait
//     for every prefetched stage ahead of it in the queue.
//   * past the last tile the loader re-issues the final stages (valid addresses, dead ring slots) so that every
//     iteration issues exactly one stage and the counted waits stay uniform.
// ------------------------------------------------------------------------------------------------
// In-launch cycle account (bench.py's measured `roofline.per_cycle_frac`, round 6): when hugs_debug_set_nt_cycles() has handed the
// library a device buffer, thread 0 of every workgroup adds the shader cycles (s_memtime) between its first instruction and the
// retirement of its last K stage, and the number of tiles it walked, to the record of its (epilogue specialisation, K class):
// buf[(EPI * 4 + kclass) * 2 + {0, 1}], kclass = 0: K <= 256, 1: K = 512, 2: K = 1024, 3: anything else.  Two 64-bit atomics per
// workgroup and launch; a null pointer (the default) costs one scalar load at the end of the kernel.
__device__ unsigned long long* g_nt_cycles;

template <int EPI>
__global__ __launch_bounds__(512, 2) void k_gemm_nt_bf16_pers(
    int M, int N, int K1, int K2, const uint16_t* __restrict__ A1, int lda1, const uint16_t* __restrict__ A2, int lda2,
    const uint16_t* __restrict__ Bt, int ldb, GemmEpi E, int ntiles) {
  constexpr int NSLOT = 4, A_BYTES = 256 * 64, STAGE = 2 * A_BYTES;
  constexpr int VEC_OFF = NSLOT * STAGE;                      // bias / r1_col copies: 2 x 16 KiB behind the ring
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * STAGE + 2 * 16384];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N >> 8;
  const int wm = wv >> 2, wn = wv & 3;
  const int ns = (K1 + K2) >> 5;
  const int G = gridDim.x;
  const int nmine = (ntiles - (int)blockIdx.x + G - 1) / G;
  const unsigned long long cyc_begin = __builtin_readcyclecounter();

  float* lds_bias = (float*)(lds + VEC_OFF);
  float* lds_r1 = (float*)(lds + VEC_OFF + 16384);
  {
    const bool has_bias = EPI < 0 ? E.bias != nullptr : bool(EPI & EPI_BIAS);
    const bool has_r1 = EPI < 0 ? E.r1_row != nullptr : bool(EPI & EPI_R1);
    if (has_bias) for (int n = tid * 4; n < N; n += 2048) *(float4*)(lds_bias + n) = *(const float4*)(E.bias + n);
    if (has_r1) for (int n = tid * 4; n < N; n += 2048) *(float4*)(lds_r1 + n) = *(const float4*)(E.r1_col + n);
  }

  // ---- loader state (wave-uniform, SGPRs): tile being fetched, stage within it, ring slot ----
  // Every LDS-DMA takes its global address as (64-bit SGPR base) + (32-bit per-lane byte offset): the per-lane part
  // is constant for the whole kernel (6 VGPRs), the base is scalar arithmetic -- no VALU and no address VGPR pairs in
  // the loop (the builtin's 64-bit per-lane addresses cost 3 VALU + a VGPR pair per load and pushed the kernel into
  // scratch spills, whose reloads are VMEM operations that drain the counted DMA queue).
  // K rotation (round 5): a tile walks its K stages starting at l_k = rot(tile) and wraps.  The 8 workgroups of an XCD that
  // share a weight column panel in a round (8 row bands x ntn column tiles on its 32 CUs) get 8 different rotations, the ntn
  // workgroups that share a row band the same one: every K slice of the weight panel is then touched 8 times per round at evenly
  // spaced moments instead of by all 8 at once and not again for a whole round (30 us, during which 4 MB of activations stream
  // through the XCD's 4 MB L2 and push it out: the forward layer read 409 MB for 270 MB algorithmic,
  // profiles/r04_gemm_traffic.json), while the activation slices are still fetched once and shared.  fp32 accumulation order
  // changes per tile (deterministically); nothing else does.
#ifndef HUGS_NT_KROT
#define HUGS_NT_KROT 1
#endif
  auto rot_of = [&](int t) { return HUGS_NT_KROT ? ((((t / ntn) & 7) * ns) >> 3) : 0; };
  int l_bid = blockIdx.x, l_st = 0, l_slot = 0, l_k;
  int lm0, ln0;
  { const int t = xcd_remap(l_bid, ntiles); lm0 = (t / ntn) << 8; ln0 = (t % ntn) << 8; l_k = rot_of(t); }
  const int prow = tid >> 2, pcol = ((tid & 3) ^ (3 * ((prow >> 2) & 1))) * 8;   // this thread's chunk of rows 0..127
  // (rows 128..255 of a stage: the same per-lane offset on a base advanced by 128 rows)
  const unsigned oA1 = (unsigned)(prow * lda1 + pcol) * 2u, oA2 = (unsigned)(prow * lda2 + pcol) * 2u;
  const unsigned oB = (unsigned)(prow * ldb + pcol) * 2u;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)wv * 1024u;
  auto dma = [&](const char* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  };
  auto issue = [&]() {
    const int kglob = l_k << 5;
    const unsigned la = lds_base + (unsigned)l_slot * STAGE;
    const char* bb = (const char*)Bt + ((size_t)ln0 * ldb + kglob) * 2;
    if (kglob < K1) {
      const char* ab = (const char*)A1 + ((size_t)lm0 * lda1 + kglob) * 2;
      dma(ab, oA1, la); dma(ab + (size_t)lda1 * 256, oA1, la + 8192);
    } else {
      const char* ab = (const char*)A2 + ((size_t)lm0 * lda2 + (kglob - K1)) * 2;
      dma(ab, oA2, la); dma(ab + (size_t)lda2 * 256, oA2, la + 8192);
    }
    dma(bb, oB, la + A_BYTES); dma(bb + (size_t)ldb * 256, oB, la + A_BYTES + 8192);
    l_slot = (l_slot + 1) & 3;
    if (++l_k == ns) l_k = 0;
    if (++l_st == ns) {
      l_st = 0;
      if (l_bid + G < ntiles) {
        l_bid += G;
        const int t = xcd_remap(l_bid, ntiles);
        lm0 = (t / ntn) << 8; ln0 = (t % ntn) << 8; l_k = rot_of(t);
      }   // else: keep re-reading the last tile (dead slots, uniform counts)
    }
  };

  f32x4_t acc[8][4];
  const int r16 = lane & 15, kb = lane >> 4;
  const int frag_off = r16 * 64 + ((kb ^ (3 * ((r16 >> 2) & 1))) << 4);
  struct Frags { bf16x8_t wb[4], xa[8]; };
  int c_slot = 0;       // ring slot of the stage whose fragments are loaded next
  auto load_frags = [&](Frags& f) {
    const unsigned char* la = lds + c_slot * STAGE + (wm * 128) * 64 + frag_off;
    const unsigned char* lb = lds + c_slot * STAGE + A_BYTES + (wn * 64) * 64 + frag_off;
#pragma unroll
    for (int j = 0; j < 4; ++j) f.wb[j] = *(const bf16x8_t*)(lb + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) f.xa[i] = *(const bf16x8_t*)(la + i * 16 * 64);
    c_slot = (c_slot + 1) & 3;
  };
#ifdef HUGS_MFMA32_STANDIN
  // TIMING STAND-IN (scratch/r4_mfma32.sh): the stage's 32 v_mfma_f32_16x16x32 replaced by 16
  // v_mfma_f32_32x32x16 on the same fragment registers -- same flops, LDS bytes and accumulator registers, half the operand reads
  // per flop; the products are NOT the GEMM's (the fragment layout of the 32x32 shape differs): for the clock only
  typedef __attribute__((ext_vector_type(16))) float f32x16_t;
  f32x16_t acc16[4][2];
  auto mfma32_q = [&](const Frags& f, int q) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
        acc16[q][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wb[cb * 2 + kh], f.xa[2 * q + kh], acc16[q][cb], 0, 0, 0);
  };
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) mfma32_q(f, q);
  };
#else
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = HUGS_MFMA_16X16X32(f.wb[j], f.xa[i], acc[i][j], 0, 0, 0);
  };
#endif
  // The same work in four fenced quarters of {1 LDS-DMA, 3 fragment reads, 8 MFMAs}: the four DMAs
  // of a stage are not pushed into the CU's vector-memory path back to back by all 8 waves at once.
  auto issue_piece = [&](int q) {
    const int kglob = l_k << 5;
    const unsigned la = lds_base + (unsigned)l_slot * STAGE;
    if (q < 2) {
      if (kglob < K1) dma((const char*)A1 + ((size_t)lm0 * lda1 + kglob) * 2 + (q ? (size_t)lda1 * 256 : 0), oA1, la + q * 8192);
      else dma((const char*)A2 + ((size_t)lm0 * lda2 + (kglob - K1)) * 2 + (q ? (size_t)lda2 * 256 : 0), oA2, la + q * 8192);
    } else {
      dma((const char*)Bt + ((size_t)ln0 * ldb + kglob) * 2 + (q == 3 ? (size_t)ldb * 256 : 0), oB, la + A_BYTES + (q - 2) * 8192);
    }
    if (q == 3) {
      l_slot = (l_slot + 1) & 3;
      if (++l_k == ns) l_k = 0;
      if (++l_st == ns) {
        l_st = 0;
        if (l_bid + G < ntiles) {
          l_bid += G;
          const int t = xcd_remap(l_bid, ntiles);
          lm0 = (t / ntn) << 8; ln0 = (t % ntn) << 8; l_k = rot_of(t);
        }
      }
    }
  };
  auto frags_piece = [&](Frags& f, int q) {      // reads 3q .. 3q+2 of {wb[0..3], xa[0..7]}
    const unsigned char* la = lds + c_slot * STAGE + wm * 128 * 64 + frag_off;
    const unsigned char* lb = lds + c_slot * STAGE + A_BYTES + wn * 64 * 64 + frag_off;
#pragma unroll
    for (int r = 3 * q; r < 3 * q + 3; ++r) {
      if (r < 4) f.wb[r] = *(const bf16x8_t*)(lb + r * 16 * 64);
      else f.xa[r - 4] = *(const bf16x8_t*)(la + (r - 4) * 16 * 64);
    }
    if (q == 3) c_slot = (c_slot + 1) & 3;
  };
#ifdef HUGS_MFMA32_STANDIN
  auto mfma_piece = [&](const Frags& f, int q) { mfma32_q(f, q); };
#else
  auto mfma_piece = [&](const Frags& f, int q) {
#pragma unroll
    for (int i = 2 * q; i < 2 * q + 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = HUGS_MFMA_16X16X32(f.wb[j], f.xa[i], acc[i][j], 0, 0, 0);
  };
#endif
#define GP_Q(cur, nxt, q)                                                                                   \
    issue_piece(q); frags_piece(nxt, q); mfma_piece(cur, q);                                                 \
    __builtin_amdgcn_sched_barrier(0);
  // iteration for stage g: frags(g) are in `cur`; make stage g+1 visible, refill the slot of stage g with stage g+4,
  // start reading frags(g+1) into `nxt`, run the MFMAs of stage g.
#define GP_ITERQ(cur, nxt, VM)                                                           \
  {                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_s_barrier();                                                         \
    asm volatile("" ::: "memory");                                                        \
    GP_Q(cur, nxt, 0) GP_Q(cur, nxt, 1) GP_Q(cur, nxt, 2) GP_Q(cur, nxt, 3)               \
  }
  // Round 5 experiment (HUGS_NT_HALF=1): the wait + barrier of an iteration sits in the MIDDLE of the stage's MFMAs.  Quarters 0, 1
  // (fragments already in registers) are issued BEFORE it, so the matrix pipe has 16 MFMAs per wave queued while the eight waves
  // straggle into the barrier; the LDS-DMA of stage g+4 and the reads of stage g+1's first fragments (wb[0..3], xa[0..3]) follow it
  // next to quarters 2, 3; the other four fragments (xa[4..7], needed from quarter 2 on) are read at the top of the NEXT iteration,
  // ahead of its barrier -- their ring slot stays valid until that barrier releases the slot's refill.
  auto frags_late = [&](Frags& f) {
    const unsigned char* la = lds + ((c_slot + 3) & 3) * STAGE + wm * 128 * 64 + frag_off;
#pragma unroll
    for (int i = 4; i < 8; ++i) f.xa[i] = *(const bf16x8_t*)(la + i * 16 * 64);
  };
  auto frags_early = [&](Frags& f, int h) {
    const unsigned char* la = lds + c_slot * STAGE + wm * 128 * 64 + frag_off;
    const unsigned char* lb = lds + c_slot * STAGE + A_BYTES + wn * 64 * 64 + frag_off;
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) f.wb[j] = *(const bf16x8_t*)(lb + j * 16 * 64);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) f.xa[i] = *(const bf16x8_t*)(la + i * 16 * 64);
      c_slot = (c_slot + 1) & 3;
    }
  };
  auto frags_early2 = [&](Frags& f, int e) {      // e = 0, 1: wb[2e], wb[2e+1]; e = 2, 3: xa[2(e-2)], xa[2(e-2)+1]; e == 3 advances the slot
    const unsigned char* la = lds + c_slot * STAGE + wm * 128 * 64 + frag_off;
    const unsigned char* lb = lds + c_slot * STAGE + A_BYTES + wn * 64 * 64 + frag_off;
    if (e < 2) { f.wb[2 * e] = *(const bf16x8_t*)(lb + (2 * e) * 16 * 64); f.wb[2 * e + 1] = *(const bf16x8_t*)(lb + (2 * e + 1) * 16 * 64); }
    else { f.xa[2 * (e - 2)] = *(const bf16x8_t*)(la + (2 * (e - 2)) * 16 * 64); f.xa[2 * (e - 2) + 1] = *(const bf16x8_t*)(la + (2 * (e - 2) + 1) * 16 * 64); }
    if (e == 3) c_slot = (c_slot + 1) & 3;
  };
  auto mfma_row = [&](const Frags& f, int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = HUGS_MFMA_16X16X32(f.wb[j], f.xa[i], acc[i][j], 0, 0, 0);
  };
#define GP_ITERH(cur, nxt, VM)                                                           \
  {                                                                                       \
    frags_late(cur);                                                                      \
    mfma_piece(cur, 0); __builtin_amdgcn_sched_barrier(0);                                \
    mfma_piece(cur, 1); __builtin_amdgcn_sched_barrier(0);                                \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_s_barrier();                                                         \
    asm volatile("" ::: "memory");                                                        \
    issue_piece(0); frags_early2(nxt, 0); mfma_row(cur, 4); __builtin_amdgcn_sched_barrier(0); \
    issue_piece(1); frags_early2(nxt, 1); mfma_row(cur, 5); __builtin_amdgcn_sched_barrier(0); \
    issue_piece(2); frags_early2(nxt, 2); mfma_row(cur, 6); __builtin_amdgcn_sched_barrier(0); \
    issue_piece(3); frags_early2(nxt, 3); mfma_row(cur, 7); __builtin_amdgcn_sched_barrier(0); \
  }
#ifndef HUGS_NT_HALF
#define HUGS_NT_HALF 0
#endif
#define GP_ITER(cur, nxt, VM)                                                            \
  {                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_s_barrier();                                                         \
    asm volatile("" ::: "memory");                                                        \
    issue();                                                                              \
    load_frags(nxt);                                                                      \
    mfmas(cur);                                                                           \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                  \
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                  \
    }                                                                                     \
  }
  Frags f0, f1;
  HUGS_STAGGER()
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) issue();
  // all four stages of the cold ring land before the first iteration: the first tile can then run the same three
  // vmcnt(24) iterations as every later tile (which has 16 epilogue stores in the queue at that point) -- one code path
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                        // also publishes the bias / r1 vectors
  load_frags(f0);
  int c_bid = blockIdx.x;
  for (int i = 0; i < nmine; ++i, c_bid += G) {
    int m0, n0;
    { const int t = xcd_remap(c_bid, ntiles); m0 = (t / ntn) << 8; n0 = (t % ntn) << 8; }
    if constexpr (EPI >= 0 && (EPI & EPI_BIAS) != 0) {      // the accumulators start from the bias (same for all 8 row blocks)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float4 bb = *(const float4*)(lds_bias + n0 + wn * 64 + b * 16 + kb * 4);
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a][b] = f32x4_t{bb.x, bb.y, bb.z, bb.w};
      }
    } else {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#ifdef HUGS_MFMA32_STANDIN
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc16[q][cb][e] = acc[2 * q + (e >> 3)][2 * cb + ((e >> 2) & 1)][e & 3];
#endif
    HUGS_TRP(i, 0)
    // the previous tile's 16 stores (+ 4 mask-bit words) are in the queue behind the two younger stages
    // Round 3: iterations 1-3 in the quartered form too (issued back to back, the four DMAs of a stage cost ~0.7k cycles more
    // per iteration, scratch/ntp_trace.py: first four iterations 8.7k -> 8.0k cycles, forward layer 255 -> 252 us in-step).
    // Iteration 0 keeps the unfenced form: the bias is still live as the first MFMAs' C operand there, and the fenced quarters
    // leave the compiler no order in which 16 bias + 96 fragment + 128 accumulator registers fit (every specialisation spilled).
    // What remains of the tile-boundary cost is NOT mainly a store-acknowledgement stall: a five-slot ring that requests
    // stage 4 ahead of the stores (bias from VGPRs, all 160 KiB of LDS) took only ~0.5k off these iterations and lost 1.9k in
    // its epilogue -- measured, removed; DESIGN.md section 4.
#if HUGS_NT_HALF
    if (EPI >= 0 && (EPI & EPI_BOUT)) { GP_ITERH(f0, f1, 28) GP_ITERH(f1, f0, 28) GP_ITERH(f0, f1, 28) }
    else { GP_ITERH(f0, f1, 24) GP_ITERH(f1, f0, 24) GP_ITERH(f0, f1, 24) }
    GP_ITERH(f1, f0, 8)
    HUGS_TRP(i, 1)
#pragma unroll 1
    for (int st = 4; st < ns; st += 2) { GP_ITERH(f0, f1, 8) GP_ITERH(f1, f0, 8) }
#else
    if (EPI >= 0 && (EPI & EPI_BOUT)) { GP_ITER(f0, f1, 28) GP_ITERQ(f1, f0, 28) GP_ITERQ(f0, f1, 28) }
    else { GP_ITER(f0, f1, 24) GP_ITERQ(f1, f0, 24) GP_ITERQ(f0, f1, 24) }
    GP_ITERQ(f1, f0, 8)
    HUGS_TRP(i, 1)
#pragma unroll 1
    for (int st = 4; st < ns; st += 2) { GP_ITERQ(f0, f1, 8) GP_ITERQ(f1, f0, 8) }
#endif
    HUGS_TRP(i, 2)
#ifdef HUGS_MFMA32_STANDIN
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[2 * q + (e >> 3)][2 * cb + ((e >> 2) & 1)][e & 3] = acc16[q][cb][e];
#endif
    nt_epilogue_direct<EPI, (EPI >= 0 && (EPI & EPI_BIAS) != 0), true, 1>(acc, E, m0, n0, wm, wn, r16, kb, lds_bias, lds_r1);
    HUGS_TRP(i, 3)
  }
#undef GP_ITER
#undef GP_ITERQ
#undef GP_ITERH
#undef GP_Q
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dead stages past the last tile must land before the LDS is released
  if (tid == 0) {
    unsigned long long* const cy = g_nt_cycles;
    if (cy) {
      const int kc = ns <= 8 ? 0 : ns == 16 ? 1 : ns == 32 ? 2 : 3;
      unsigned long long* const rec = cy + ((EPI < 0 ? 63 : EPI) * 4 + kc) * 2;
      atomicAdd(rec, __builtin_readcyclecounter() - cyc_begin);
      atomicAdd(rec + 1, (unsigned long long)nmine);
    }
  }
}

#include "hugs_gemm_p64.inc"
#include "hugs_gemm_dq.inc"
// the four-wave form (one wave per SIMD, 128 x 128 per wave, accumulators in AGPRs): built, bit-identical, measured SLOWER (259 vs 241 us,
// profiles/r06_nt_w4_ab.txt) and it spills 12-24 bytes per lane -- compiled only with -DHUGS_BUILD_W4 (scratch/build_variant.sh), then
// selected by HUGS_NT_W4=1
#ifdef HUGS_BUILD_W4
#include "hugs_gemm_w4.inc"
#endif
#include "hugs_gemm_chain.inc"

// ------------------------------------------------------------------------------------------------
// bf16 TN: slab[split][Kc, N] (fp32) = sum over this split's rows of X[m,Kc]^T G[m,N]
//   tile: 128 (Kc) x 128 (N), reduction step 64 rows; both tiles row-major [64][128] bf16 in LDS,
//   32-byte granules XOR-swizzled by (row & 7) so the 8 rows a transpose-read pair touches hit 8
//   distinct bank groups.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_gemm_tn_bf16(int Mrows, int Kc, int N, int nsplit,
                                                          const uint16_t* __restrict__ X, int ldx,
                                                          const uint16_t* __restrict__ G, int ldg,
                                                          float* __restrict__ slab, int lds_out,
                                                          float* __restrict__ colsum_slab) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * GB_TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / 128, ntk = Kc / 128;
  const int tiles = ntn * ntk;
  const int bid = blockIdx.x;
  // Round 4: the tiles of one M-split share an operand (a [rows,128] x [rows,256] product reads X in both of its tiles): with
  // consecutive blocks dealt round-robin over the 8 XCDs, tile 0 and tile 1 of a split sat behind different L2s and the shared
  // rows came from HBM twice (1024 B per row for 768).  In groups of 8 * tiles blocks, block r of a group is tile r / 8 of the
  // group's split r % 8: a split's tiles are consecutive blocks of ONE XCD.
  int split = bid / tiles, tt = bid % tiles;
  if (nsplit % 8 == 0 && tiles > 1) {
    const int grp = bid / (8 * tiles), r = bid - grp * 8 * tiles;
    split = grp * 8 + (r & 7); tt = r >> 3;
  }
  const int c0 = (tt / ntn) * 128, n0 = (tt % ntn) * 128;
  const int rows_per = Mrows / nsplit;
  const int mbeg = split * rows_per;
  const int nk = rows_per / 64;
  const int wk = wv >> 1, wn = wv & 1;  // wave owns Kc rows [wk*64, +64), N cols [wn*64, +64)

  auto stage = [&](int kt, int buf) {
    const int mrow0 = mbeg + kt * 64;
    unsigned char* lx = lds + buf * 2 * GB_TILE_BYTES;
    unsigned char* lg = lx + GB_TILE_BYTES;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int p = it * 256 + tid;          // 1024 chunks: row p/16, physical pos p%16
      const int row = p >> 4, pos = p & 15;
      const int c = ((((pos >> 1) ^ (row & 7)) << 1) | (pos & 1));
      glds16(X + (size_t)(mrow0 + row) * ldx + c0 + c * 8, lx + (it * 256 + wv * 64) * 16);
      glds16(G + (size_t)(mrow0 + row) * ldg + n0 + c * 8, lg + (it * 256 + wv * 64) * 16);
    }
  };

  f32x4_t acc[4][4];  // [i over N frags][j over Kc frags]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // Column sums of G (the bias gradient), tiles with c0 == 0 only: four extra MFMAs per 32 rows against an all-ones
  // fragment in the waves with wk == 0 (both wk waves read the same G fragments).  The role is chosen ONCE outside
  // the loop so that the loop body stays a single basic block.
  const bool do_colsum = colsum_slab && c0 == 0;
  f32x4_t accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t ones;
#pragma unroll
  for (int q = 0; q < 8; ++q) ones[q] = (hugs_op_t)1.0f;

  stage(0, 0);
  const int g = lane >> 4, s = lane & 15;
  auto run = [&](auto role) {
    constexpr bool COLSUM = decltype(role)::value;
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      __syncthreads();
      if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
      const unsigned char* lx = lds + buf * 2 * GB_TILE_BYTES;
      const unsigned char* lg = lx + GB_TILE_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8_t ga[4], xb[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = kk * 32 + h * 16 + g * 4 + (s >> 2);
          const int sw = row & 7;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int q = (wn * 64 + i * 16) >> 4;  // 32-byte granule of the fragment's 16 columns
            bf16x4_t v = HUGS_DS_READ_TR16(
                (bf16x4_t __attribute__((address_space(3)))*)(lg + row * 256 + ((q ^ sw) << 5) + ((s & 3) << 3)));
            ga[i][h * 4 + 0] = v[0]; ga[i][h * 4 + 1] = v[1]; ga[i][h * 4 + 2] = v[2]; ga[i][h * 4 + 3] = v[3];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = (wk * 64 + j * 16) >> 4;
            bf16x4_t v = HUGS_DS_READ_TR16(
                (bf16x4_t __attribute__((address_space(3)))*)(lx + row * 256 + ((q ^ sw) << 5) + ((s & 3) << 3)));
            xb[j][h * 4 + 0] = v[0]; xb[j][h * 4 + 1] = v[1]; xb[j][h * 4 + 2] = v[2]; xb[j][h * 4 + 3] = v[3];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = HUGS_MFMA_16X16X32(ga[i], xb[j], acc[i][j], 0, 0, 0);
        if constexpr (COLSUM) {
#pragma unroll
          for (int i = 0; i < 4; ++i) accb[i] = HUGS_MFMA_16X16X32(ga[i], ones, accb[i], 0, 0, 0);
        }
      }
    }
  };
  if (do_colsum && wk == 0) run(std::true_type{});
  else run(std::false_type{});
  float* out = slab + (size_t)split * Kc * lds_out;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + wk * 64 + j * 16 + s;
      const int n = n0 + wn * 64 + i * 16 + g * 4;
      *(float4*)(out + (size_t)c * lds_out + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  if (do_colsum && wk == 0 && s == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *(float4*)(colsum_slab + (size_t)split * N + n0 + wn * 64 + i * 16 + g * 4) =
          make_float4(accb[i][0], accb[i][1], accb[i][2], accb[i][3]);
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 TN, large tile: 256 (Kc) x 256 (N) output tile per workgroup, 8 waves (2 over N x 4 over Kc, each
// 128 x 64), reduction rows walked 32 at a time through the same 4-slot ring / counted-vmcnt / register
// double-buffer pipeline as the NT kernel.  Both operands are read with ds_read_b64_tr_b16 from row-major
// [32][256] stages stored as 1 KiB row pairs at a 1056-byte pitch (bank spread by padding, not by an address
// swizzle: fragment reads are base + immediate, which frees ~25 VGPRs and the XOR arithmetic).  The bias gradient (column sums of G) costs two
// extra MFMAs per stage against an all-ones fragment instead of a scalar LDS pass.
// ------------------------------------------------------------------------------------------------
// (Round-2 measurement builds -- no DMA / no fragment reads / no MFMA / L2-resident loads / no barrier / uncounted waits,
// 8:6 / 4:3 / DS-first interleaves, an L2 prefetch of the XCD siblings' lines -- are in the git history; DESIGN.md
// section 4 has their numbers.)
#ifndef HUGS_TN_SYNC
#define HUGS_TN_SYNC 128      // batched launch: stages between two re-alignments of a tile group (even)
#endif
// One 256x256 tile of D = X^T G over the reduction rows [mbeg, mbeg + 32 ns): the body shared by the per-layer kernel
// (k_gemm_tn_bf16_big) and the batched one (k_gemm_tn_bf16_batch).  out: this split's fp32 slab [Kc, lds_out]; colsum:
// this split's bias-gradient row [N] or null.  ns must be even and >= 8.
__device__ __forceinline__ void tn_big_tile(const uint16_t* __restrict__ X, int ldx, const uint16_t* __restrict__ G, int ldg,
                                            int mbeg, int ns, int c0, int n0, float* __restrict__ out, int lds_out,
                                            float* __restrict__ colsum, unsigned* sync_ctr = nullptr, int sync_n = 0) {
  // stage layout: per operand 16 row PAIRS of 1 KiB (rows 2p, 2p+1 as the LDS-DMA writes them) at a 1056-byte
  // pitch: the 8 rows one 32-lane group of a transpose read touches have distinct p, so they land 8 banks apart
  // (1056 B = 264 dwords = 8 mod 64) with NO address swizzle: every fragment read is base + immediate.
  constexpr int NSLOT = 4, PAIR = 1056, XB = 16 * PAIR, STAGE = 2 * XB;
  constexpr int RING = 128 * 1040 > NSLOT * STAGE ? 128 * 1040 : NSLOT * STAGE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[RING];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wv >> 2, wk = wv & 3;
  const bool do_colsum = colsum && c0 == 0;
  // LDS-DMA as inline asm: (64-bit SGPR base) + (32-bit per-lane byte offset, constant for the whole kernel), M0 = LDS
  // destination.  Through the builtin the compiler knows the instruction writes LDS and puts an `s_waitcnt vmcnt(0)` in
  // front of the first transpose read of EVERY iteration (it cannot prove the ds_read_tr intrinsics do not alias the
  // DMA destination): the ring drained each stage and the counted waits below never counted anything.  The explicit
  // vmcnt / barrier protocol is the synchronisation; the "memory" clobber keeps the compiler from moving reads across.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned voX = (unsigned)((lane >> 5) * ldx + (lane & 31) * 8) * 2u, voG = (unsigned)((lane >> 5) * ldg + (lane & 31) * 8) * 2u;
  auto dma16 = [&](const char* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  };
  auto stage = [&](int st) {
    const int mrow0 = mbeg + (st << 5);
    const unsigned l = lds0 + (unsigned)(st % NSLOT) * STAGE;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pair = it * 8 + wv;
      dma16((const char*)(X + (size_t)(mrow0 + 2 * pair) * ldx + c0), voX, l + pair * PAIR);
      dma16((const char*)(G + (size_t)(mrow0 + 2 * pair) * ldg + n0), voG, l + XB + pair * PAIR);
    }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t accb[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  bf16x8_t ones;
#pragma unroll
  for (int q = 0; q < 8; ++q) ones[q] = (hugs_op_t)1.0f;

  const int g = lane >> 4, s = lane & 15;
  struct Frags { bf16x8_t ga[8], xb[4]; };
  auto load_frags = [&](Frags& f, int st) {
    const unsigned char* lx = lds + (st % NSLOT) * STAGE;
    const unsigned char* lg = lx + XB;
    // reduction row of (lane group g, r = s>>2, read h): (g>>1)*16 + 2*((g&1)*4 + r) + h  (a bijection on the 32 rows)
    const int lo = ((g >> 1) * 8 + (g & 1) * 4 + (s >> 2)) * PAIR + ((s & 3) << 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bf16x4_t v = HUGS_DS_READ_TR16((bf16x4_t __attribute__((address_space(3)))*)(lg + lo + h * 512 + ((wn * 8 + i) << 5)));
        f.ga[i][h * 4 + 0] = v[0]; f.ga[i][h * 4 + 1] = v[1]; f.ga[i][h * 4 + 2] = v[2]; f.ga[i][h * 4 + 3] = v[3];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x4_t v = HUGS_DS_READ_TR16((bf16x4_t __attribute__((address_space(3)))*)(lx + lo + h * 512 + ((wk * 4 + j) << 5)));
        f.xb[j][h * 4 + 0] = v[0]; f.xb[j][h * 4 + 1] = v[1]; f.xb[j][h * 4 + 2] = v[2]; f.xb[j][h * 4 + 3] = v[3];
      }
    }
  };
  // The main loop is instantiated per bias-gradient role (WK = -1: none; 0..3: this wave's two G fragments), chosen
  // by ONE wave-uniform switch outside the loop: a branch inside an iteration splits it into basic blocks and the
  // scheduler can then no longer overlap the next stage's ds_reads with this stage's MFMAs across iterations.
  Frags f0, f1;
  auto run = [&](auto role) {
    constexpr int WK = decltype(role)::value;
    auto mfmas = [&](const Frags& f) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = HUGS_MFMA_16X16X32(f.ga[i], f.xb[j], acc[i][j], 0, 0, 0);
      if constexpr (WK >= 0) {   // wave (wn, wk) owns the column sums of its N fragments 2wk, 2wk+1
        accb[0] = HUGS_MFMA_16X16X32(f.ga[2 * WK], ones, accb[0], 0, 0, 0);
        accb[1] = HUGS_MFMA_16X16X32(f.ga[2 * WK + 1], ones, accb[1], 0, 0, 0);
      }
    };
    // An iteration in four fenced quarters of {1 LDS-DMA, 6 transpose reads, 8 MFMAs}.  With the four
    // DMAs issued back to back behind the barrier, 8 waves push 32 KB into the CU's one vector-memory path at once and
    // every wave sits in VMEM issue (no MFMA behind it can go: in-order issue) until the queue has drained.
    auto stage_piece = [&](int st, int q) {
      const int mrow0 = mbeg + (st << 5);
      const unsigned l = lds0 + (unsigned)(st % NSLOT) * STAGE;
      const int pair = (q >> 1) * 8 + wv;
      if (q & 1) dma16((const char*)(G + (size_t)(mrow0 + 2 * pair) * ldg + n0), voG, l + XB + pair * PAIR);
      else dma16((const char*)(X + (size_t)(mrow0 + 2 * pair) * ldx + c0), voX, l + pair * PAIR);
    };
    auto frags_piece = [&](Frags& f, int st, int q) {
      const unsigned char* lx = lds + (st % NSLOT) * STAGE;
      const unsigned char* lg = lx + XB;
      const int lo = ((g >> 1) * 8 + (g & 1) * 4 + (s >> 2)) * PAIR + ((s & 3) << 3);
#pragma unroll
      for (int r = 6 * q; r < 6 * q + 6; ++r) {
        const int h = r / 12, idx = r % 12;
        if (idx < 8) {
          bf16x4_t v = HUGS_DS_READ_TR16((bf16x4_t __attribute__((address_space(3)))*)(lg + lo + h * 512 + ((wn * 8 + idx) << 5)));
          f.ga[idx][h * 4 + 0] = v[0]; f.ga[idx][h * 4 + 1] = v[1]; f.ga[idx][h * 4 + 2] = v[2]; f.ga[idx][h * 4 + 3] = v[3];
        } else {
          bf16x4_t v = HUGS_DS_READ_TR16((bf16x4_t __attribute__((address_space(3)))*)(lx + lo + h * 512 + ((wk * 4 + idx - 8) << 5)));
          f.xb[idx - 8][h * 4 + 0] = v[0]; f.xb[idx - 8][h * 4 + 1] = v[1]; f.xb[idx - 8][h * 4 + 2] = v[2]; f.xb[idx - 8][h * 4 + 3] = v[3];
        }
      }
    };
    auto mfma_piece = [&](const Frags& f, int q) {
#pragma unroll
      for (int i = 2 * q; i < 2 * q + 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = HUGS_MFMA_16X16X32(f.ga[i], f.xb[j], acc[i][j], 0, 0, 0);
      if constexpr (WK >= 0) {
        if (q == WK) {
          accb[0] = HUGS_MFMA_16X16X32(f.ga[2 * WK], ones, accb[0], 0, 0, 0);
          accb[1] = HUGS_MFMA_16X16X32(f.ga[2 * WK + 1], ones, accb[1], 0, 0, 0);
        }
      }
    };
#define GT_Q(cur, nxt, st, q, DOST)                                       \
    if (DOST) stage_piece((st) + NSLOT, q);                               \
    frags_piece(nxt, (st) + 1, q);                                        \
    mfma_piece(cur, q);                                                   \
    __builtin_amdgcn_sched_barrier(0);
#define GT_ITER4(cur, nxt, st, VM, DOST)                                  \
  {                                                                       \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                         \
    asm volatile("" ::: "memory");                                        \
    GT_Q(cur, nxt, st, 0, DOST) GT_Q(cur, nxt, st, 1, DOST) GT_Q(cur, nxt, st, 2, DOST) GT_Q(cur, nxt, st, 3, DOST) \
  }
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) stage(q);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(f0, 0);
    int st = 0;
    if (sync_ctr) {
      // Best-effort re-alignment of the workgroups that share operand panels (the tiles of one (item, piece): same X / G row
      // block, one XCD): over a 2048-stage loop they drift apart and the panel a neighbour fetched has left the 4 MB L2 by the
      // time the next one asks for it (PMC: 1.58x the algorithmic bytes, against 1.28x for the 256-stage per-layer launch).
      // Every HUGS_TN_SYNC stages one lane announces the group's arrival count and waits -- a BOUNDED number of polls, so a
      // group member that is not resident yet (another kernel holds its CU) delays nobody for long.
      unsigned epoch = 0;
      for (int lim = HUGS_TN_SYNC; lim + 5 < ns; lim += HUGS_TN_SYNC) {
        for (; st < lim; st += 2) { GT_ITER4(f0, f1, st, 8, 1) GT_ITER4(f1, f0, st + 1, 8, 1) }
        ++epoch;
        if (threadIdx.x == 0) {
          __hip_atomic_fetch_add(sync_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned want = epoch * (unsigned)sync_n;
          for (int spin = 0; spin < 512 && __hip_atomic_load(sync_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_s_barrier();
      }
    }
    for (; st + 5 < ns; st += 2) { GT_ITER4(f0, f1, st, 8, 1) GT_ITER4(f1, f0, st + 1, 8, 1) }
    GT_ITER4(f0, f1, st, 8, 0)
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    load_frags(f0, st + 2); mfmas(f1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    load_frags(f1, st + 3); mfmas(f0);
    mfmas(f1);
#undef GT_ITER4
#undef GT_Q
  };
  if (!do_colsum) run(std::integral_constant<int, -1>{});
  else if (wk == 0) run(std::integral_constant<int, 0>{});
  else if (wk == 1) run(std::integral_constant<int, 1>{});
  else if (wk == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});

  // Epilogue: the fp32 tile leaves through LDS (the ring is free now) so that every wave instruction stores one
  // whole 1 KiB slab row instead of sixteen 64-byte runs.  Two passes of 128 Kc-rows (pitch 1040 B).
  constexpr int TP = 1040;
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if ((wk >> 1) == h) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rl = (wk & 1) * 64 + j * 16 + s;            // row inside this pass
          const int cl = wn * 128 + i * 16 + g * 4;
          *(float4*)(lds + rl * TP + cl * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {                            // 128 rows x 64 chunks(16 B) / 512 threads
      const int p = it * 512 + tid, rl = p >> 6, cq = p & 63;
      const float4 v = *(const float4*)(lds + rl * TP + cq * 16);
      *(float4*)(out + (size_t)(c0 + h * 128 + rl) * lds_out + n0 + cq * 4) = v;
    }
    __syncthreads();
  }
  if (do_colsum && s == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = n0 + wn * 128 + (2 * wk + q) * 16 + g * 4;
      *(float4*)(colsum + n) = make_float4(accb[q][0], accb[q][1], accb[q][2], accb[q][3]);
    }
  }
}

__global__ __launch_bounds__(512, 2) void k_gemm_tn_bf16_big(int Mrows, int Kc, int N, int nsplit,
                                                              const uint16_t* __restrict__ X, int ldx,
                                                              const uint16_t* __restrict__ G, int ldg,
                                                              float* __restrict__ slab, int lds_out,
                                                              float* __restrict__ colsum_slab) {
  const int ntn = N >> 8, ntk = Kc >> 8, tiles = ntn * ntk;
  // all tiles of one M-split read the same X / G row block: keep them on one XCD so its L2 serves the re-reads
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vb / tiles, tt = vb % tiles;
  const int c0 = (tt / ntn) << 8, n0 = (tt % ntn) << 8;
  const int rows_per = Mrows / nsplit, mbeg = split * rows_per;
  tn_big_tile(X, ldx, G, ldg, mbeg, rows_per >> 5, c0, n0, slab + (size_t)split * Kc * lds_out, lds_out,
              colsum_slab ? colsum_slab + (size_t)split * N : nullptr);
}

// Batched form (round 4): the weight gradients of SEVERAL layers in one launch -- every item an X^T G product with its own
// operands and slab -- so that nsplit = #CUs / (total tiles) instead of #CUs / (one layer's tiles): the NerfMLP trunk's
// 8 layers are 128 tiles, i.e. TWO reduction halves per tile instead of sixteen (8 MB of fp32 slabs per layer instead of 64,
// 2048-stage main loops instead of 256, one reduce launch instead of eight).  The reduction rows of an item are cut into
// nsplit near-equal pieces in units of 64 rows (unit u of piece s: [s U / nsplit, (s+1) U / nsplit)): any nsplit <= U.
// vb order: item-major, then split, then tile -- the tiles of one (item, split) are neighbours in vb and therefore share
// an XCD (xcd_remap), whose L2 serves their re-reads of the X / G row block.
#define HUGS_TN_BATCH_MAX 16
struct TnBatchItem {
  const uint16_t* X; const uint16_t* G;
  float* slab;          // [nsplit][Kc][N] fp32
  float* colsum;        // [nsplit][N] or null
  int ldx, ldg, Kc, N, units, wg0, tiles, pad_;
};
struct TnBatch { int nitems, nsplit; unsigned* sync; TnBatchItem it[HUGS_TN_BATCH_MAX]; };
__global__ __launch_bounds__(512, 2) void k_gemm_tn_bf16_batch(const TnBatch B) {
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  int i = 0;
  while (i + 1 < B.nitems && vb >= B.it[i + 1].wg0) ++i;
  const TnBatchItem& I = B.it[i];
  const int r = vb - I.wg0;
  const int split = r / I.tiles, tt = r % I.tiles;
  const int ntn = I.N >> 8;
  const int c0 = (tt / ntn) << 8, n0 = (tt % ntn) << 8;
  const int u0 = (int)(((long long)split * I.units) / B.nsplit), u1 = (int)(((long long)(split + 1) * I.units) / B.nsplit);
  tn_big_tile(I.X, I.ldx, I.G, I.ldg, u0 << 6, (u1 - u0) << 1, c0, n0, I.slab + (size_t)split * I.Kc * I.N, I.N,
              I.colsum ? I.colsum + (size_t)split * I.N : nullptr, B.sync ? B.sync + (i * B.nsplit + split) : nullptr, I.tiles);
}

#ifndef HUGS_GEMM_F16
// ------------------------------------------------------------------------------------------------
// fp32 parity path: 128x128x16 tiles on v_mfma_f32_16x16x4_f32, register-staged, padded LDS.
// ------------------------------------------------------------------------------------------------
#define GF_BK 16
#define GF_PITCH 20  // floats per LDS row (16 + 4 pad): keeps float4 alignment, spreads banks

__global__ __launch_bounds__(256, 2) void k_gemm_nt_f32(int M, int N, int K1, int K2, const float* __restrict__ A1, int lda1,
                                                         const float* __restrict__ A2, int lda2,
                                                         const float* __restrict__ Bt, int ldb, GemmEpi E) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 128 * GF_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ntn = N / 128, ntm = M / 128;
  const int t = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (t / ntn) * 128, n0 = (t % ntn) * 128;
  const int wm = wv >> 1, wn = wv & 1;
  const int nk = (K1 + K2) / GF_BK;
  // (staging registers as four named values: as `float4 ra[2], rb[2]` indexed inside the lambdas' loops they lived in scratch --
  //  64 bytes per lane written and re-read every K step)
  float4 ra0, ra1, rb0, rb1;
  const int prow = tid >> 2, pc = (tid & 3) * 4;      // chunk p = it * 256 + tid: row it * 64 + prow
  auto gload = [&](int kt) {
    const int kglob = kt * GF_BK;
    const float* Abase; int lda, kcol;
    if (kglob < K1) { Abase = A1; lda = lda1; kcol = kglob; } else { Abase = A2; lda = lda2; kcol = kglob - K1; }
    ra0 = *(const float4*)(Abase + (size_t)(m0 + prow) * lda + kcol + pc);
    ra1 = *(const float4*)(Abase + (size_t)(m0 + 64 + prow) * lda + kcol + pc);
    rb0 = *(const float4*)(Bt + (size_t)(n0 + prow) * ldb + kglob + pc);
    rb1 = *(const float4*)(Bt + (size_t)(n0 + 64 + prow) * ldb + kglob + pc);
  };
  auto lstore = [&](int buf) {
    float* la = lds + buf * 2 * 128 * GF_PITCH;
    float* lb = la + 128 * GF_PITCH;
    *(float4*)(la + prow * GF_PITCH + pc) = ra0;
    *(float4*)(la + (64 + prow) * GF_PITCH + pc) = ra1;
    *(float4*)(lb + prow * GF_PITCH + pc) = rb0;
    *(float4*)(lb + (64 + prow) * GF_PITCH + pc) = rb1;
  };
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  gload(0);
  lstore(0);
  const int r16 = lane & 15, kb = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
    const float* la = lds + buf * 2 * 128 * GF_PITCH;
    const float* lb = la + 128 * GF_PITCH;
    float4 xa[4], wb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = *(const float4*)(la + (wm * 64 + i * 16 + r16) * GF_PITCH + kb * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) wb[j] = *(const float4*)(lb + (wn * 64 + j * 16 + r16) * GF_PITCH + kb * 4);
    // lane (r16, kb) holds k = kb*4 + q; MFMA step q consumes the same k set on both operands
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j].x, xa[i].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j].y, xa[i].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j].z, xa[i].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j].w, xa[i].w, acc[i][j], 0, 0, 0);
      }
    if (kt + 1 < nk) lstore(buf ^ 1);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      epi_store4<false>(E, m0 + wm * 64 + i * 16 + r16, n0 + wn * 64 + j * 16 + kb * 4, acc[i][j]);
}

#define GF_TPITCH 132  // floats per row of a [16][128] tile
__global__ __launch_bounds__(256, 2) void k_gemm_tn_f32(int Mrows, int Kc, int N, int nsplit, const float* __restrict__ X,
                                                         int ldx, const float* __restrict__ G, int ldg,
                                                         float* __restrict__ slab, int lds_out,
                                                         float* __restrict__ colsum_slab) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 16 * GF_TPITCH];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ntn = N / 128, ntk = Kc / 128, tiles = ntn * ntk;
  const int split = blockIdx.x / tiles, tt = blockIdx.x % tiles;
  const int c0 = (tt / ntn) * 128, n0 = (tt % ntn) * 128;
  const int rows_per = Mrows / nsplit, mbeg = split * rows_per, nk = rows_per / 16;
  const int wk = wv >> 1, wn = wv & 1;
  float4 rx0, rx1, rg0, rg1;      // (named, not arrays: see k_gemm_nt_f32)
  const int prow = tid >> 5, pc = (tid & 31) * 4;     // chunk p = it * 256 + tid: row it * 8 + prow
  auto gload = [&](int kt) {
    const int mrow0 = mbeg + kt * 16;
    rx0 = *(const float4*)(X + (size_t)(mrow0 + prow) * ldx + c0 + pc);
    rx1 = *(const float4*)(X + (size_t)(mrow0 + 8 + prow) * ldx + c0 + pc);
    rg0 = *(const float4*)(G + (size_t)(mrow0 + prow) * ldg + n0 + pc);
    rg1 = *(const float4*)(G + (size_t)(mrow0 + 8 + prow) * ldg + n0 + pc);
  };
  auto lstore = [&](int buf) {
    float* lx = lds + buf * 2 * 16 * GF_TPITCH;
    float* lg = lx + 16 * GF_TPITCH;
    *(float4*)(lx + prow * GF_TPITCH + pc) = rx0;
    *(float4*)(lx + (8 + prow) * GF_TPITCH + pc) = rx1;
    *(float4*)(lg + prow * GF_TPITCH + pc) = rg0;
    *(float4*)(lg + (8 + prow) * GF_TPITCH + pc) = rg1;
  };
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;
  const bool do_colsum = colsum_slab && c0 == 0;
  gload(0);
  lstore(0);
  const int g = lane >> 4, s = lane & 15;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
    const float* lx = lds + buf * 2 * 16 * GF_TPITCH;
    const float* lg = lx + 16 * GF_TPITCH;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float ga[4], xb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ga[i] = lg[(kk * 4 + g) * GF_TPITCH + wn * 64 + i * 16 + s];
#pragma unroll
      for (int j = 0; j < 4; ++j) xb[j] = lx[(kk * 4 + g) * GF_TPITCH + wk * 64 + j * 16 + s];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i], xb[j], acc[i][j], 0, 0, 0);
    }
    if (do_colsum && tid < 128) {
      float a = 0.f;
#pragma unroll
      for (int row = 0; row < 16; ++row) a += lg[row * GF_TPITCH + tid];
      csum += a;
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
  }
  float* out = slab + (size_t)split * Kc * lds_out;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + wk * 64 + j * 16 + s;
      const int n = n0 + wn * 64 + i * 16 + g * 4;
      *(float4*)(out + (size_t)c * lds_out + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  if (do_colsum && tid < 128) colsum_slab[(size_t)split * N + n0 + tid] = csum;
}

#endif  // !HUGS_GEMM_F16 (the fp32 kernels exist once)

// sum fp32 slabs: out[i] = sum_s slab[s][i]   (fixed order: deterministic).  Eight slab loads are kept in flight
// per thread; a 16-slab, 4 MB-per-slab reduction is an HBM stream, not a latency chain.
// A second (small) range -- the bias-gradient slabs -- rides in the same launch: blocks past the first range's reduce it
// (one kernel boundary less on the weight-gradient stream per layer).
__global__ __launch_bounds__(256) void k_slab_reduce(const float* __restrict__ slab, int nsplit, size_t per,
                                                     float* __restrict__ out, unsigned nblk1,
                                                     const float* __restrict__ slab2, size_t per2, float* __restrict__ out2) {
  unsigned blk = blockIdx.x;
  if (blk >= nblk1) { blk -= nblk1; slab = slab2; per = per2; out = out2; }
  const size_t i = ((size_t)blk * blockDim.x + threadIdx.x) * 4;
  if (i >= per) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 8 <= nsplit; s += 8) {
    f32x4_t b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = __builtin_nontemporal_load((const f32x4_t*)(slab + (size_t)(s + q) * per + i));
#pragma unroll
    for (int q = 0; q < 8; ++q) { a.x += b[q][0]; a.y += b[q][1]; a.z += b[q][2]; a.w += b[q][3]; }
  }
  for (; s < nsplit; ++s) {
    const float4 b = *(const float4*)(slab + (size_t)s * per + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  *(float4*)(out + i) = a;
}

// The same reduction for a table of ranges (the weight and bias gradients of every item of a batched TN launch) in one launch.
struct RedItem { const float* slab; float* out; unsigned long long per; unsigned blk0, pad_; };
struct RedBatch { int nitems, nsplit; RedItem it[2 * HUGS_TN_BATCH_MAX]; };
__global__ __launch_bounds__(256) void k_slab_reduce_batch(const RedBatch B) {
  int k = 0;
  while (k + 1 < B.nitems && blockIdx.x >= B.it[k + 1].blk0) ++k;
  const RedItem& I = B.it[k];
  const size_t per = I.per;
  const size_t i = ((size_t)(blockIdx.x - I.blk0) * 256 + threadIdx.x) * 4;
  if (i >= per) return;
  const float* slab = I.slab;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 4 <= B.nsplit; s += 4) {
    f32x4_t b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = __builtin_nontemporal_load((const f32x4_t*)(slab + (size_t)(s + q) * per + i));
#pragma unroll
    for (int q = 0; q < 4; ++q) { a.x += b[q][0]; a.y += b[q][1]; a.z += b[q][2]; a.w += b[q][3]; }
  }
  for (; s < B.nsplit; ++s) {
    const f32x4_t b = __builtin_nontemporal_load((const f32x4_t*)(slab + (size_t)s * per + i));
    a.x += b[0]; a.y += b[1]; a.z += b[2]; a.w += b[3];
  }
  *(float4*)(I.out + i) = a;
}

}  // namespace HUGS_GEMM_NS
using namespace HUGS_GEMM_NS;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// tile_mode (hugs_gemm_nt_tiles / hugs_gemm_tn_tiles; the plain entry points pass 0): 0 = default kernel selection
// (persistent 256x256, else 256x256, else 256x128, else 128x128), 1 = force 128x128, 3 = force 256x128 where 256x256
// would be chosen, 5 = 256x256 without the persistent form.
#ifdef HUGS_GEMM_F16
#define HUGS_NT_IMPL hugs_gemm_nt_impl_f16
#define HUGS_TN_IMPL hugs_gemm_tn_impl_f16
#define HUGS_OP_DTYPE 2
#else
#define HUGS_NT_IMPL hugs_gemm_nt_impl_bf16
#define HUGS_TN_IMPL hugs_gemm_tn_impl_bf16
#define HUGS_OP_DTYPE 1
#endif
#define HUGS_NT_ARGS int tile_mode, int dtype, int M, int N, int K1, int K2, const void* A1, int lda1, const void* A2, int lda2,            \
                     const void* Bt, int ldb, const float* bias, const float* row_bias, int row_div, int ld_rb, int relu, const void* mask, \
                     int ld_mask, const float* r1_row, const float* r1_col, void* out, int ldc, uint32_t* bits_out,                         \
                     const uint32_t* bits_in, void* stream
#define HUGS_TN_ARGS int tile_mode, int dtype, int Mrows, int Kc, int N, int nsplit, const void* X, int ldx, const void* G, int ldg, \
                     float* dW, float* dbias, void* ws, void* stream
int hugs_gemm_nt_impl_bf16(HUGS_NT_ARGS);
int hugs_gemm_nt_impl_f16(HUGS_NT_ARGS);       // hugs_gemm_f16.hip
int hugs_gemm_tn_impl_bf16(HUGS_TN_ARGS);
int hugs_gemm_tn_impl_f16(HUGS_TN_ARGS);

#ifndef HUGS_GEMM_F16
static int gemm_nt_impl(HUGS_NT_ARGS) {
  return dtype == 2 ? hugs_gemm_nt_impl_f16(tile_mode, dtype, M, N, K1, K2, A1, lda1, A2, lda2, Bt, ldb, bias, row_bias, row_div, ld_rb, relu, mask,
                                            ld_mask, r1_row, r1_col, out, ldc, bits_out, bits_in, stream)
                    : hugs_gemm_nt_impl_bf16(tile_mode, dtype, M, N, K1, K2, A1, lda1, A2, lda2, Bt, ldb, bias, row_bias, row_div, ld_rb, relu, mask,
                                             ld_mask, r1_row, r1_col, out, ldc, bits_out, bits_in, stream);
}
static int gemm_tn_impl(HUGS_TN_ARGS) {
  return dtype == 2 ? hugs_gemm_tn_impl_f16(tile_mode, dtype, Mrows, Kc, N, nsplit, X, ldx, G, ldg, dW, dbias, ws, stream)
                    : hugs_gemm_tn_impl_bf16(tile_mode, dtype, Mrows, Kc, N, nsplit, X, ldx, G, ldg, dW, dbias, ws, stream);
}

// Dynamic tile queues of the persistent NT launches (hugs_gemm_dq.inc): between hugs_gemm_nt_queue_begin and _end every eligible
// hugs_gemm_nt / hugs_gemm_nt_bits launch takes the next 32-byte slot (8 counters, one per XCD) of the caller's region, which _begin
// zeroes on the stream.  Process-global, like the launch sequence it numbers; outside a begin / end pair the static tile walk runs.
static unsigned* g_ntq_base = nullptr;
static int g_ntq_slots = 0, g_ntq_next = 0;
unsigned* hugs_gemm_nt_queue_take() {
  if (!g_ntq_base || g_ntq_next >= g_ntq_slots) return nullptr;
  return g_ntq_base + 8 * (g_ntq_next++);
}
extern "C" int hugs_gemm_nt_queue_begin(void* region, long long bytes, void* stream) {
  HUGS_REQUIRE(region && bytes >= 32 && bytes % 32 == 0, -2, "hugs_gemm_nt_queue_begin: region of %lld bytes (a multiple of 32)", bytes);
  HUGS_REQUIRE(hipMemsetAsync(region, 0, (size_t)bytes, (hipStream_t)stream) == hipSuccess, -100, "hugs_gemm_nt_queue_begin: hipMemsetAsync");
  g_ntq_base = (unsigned*)region; g_ntq_slots = (int)(bytes / 32); g_ntq_next = 0;
  return 0;
}
extern "C" int hugs_gemm_nt_queue_end(int) {
  const int used = g_ntq_next;
  g_ntq_base = nullptr; g_ntq_slots = 0; g_ntq_next = 0;
  return used < 0 ? 0 : 0;
}

// measurement hook (include/hugs.h): buf = 64 x 4 x 2 device uint64 (4 KiB, zeroed by the caller) or NULL to switch the account off
extern "C" int hugs_debug_set_nt_cycles(void* buf) {
  HUGS_REQUIRE(hipMemcpyToSymbol(HIP_SYMBOL(gemm_bf16::g_nt_cycles), &buf, sizeof(buf)) == hipSuccess, -100, "hugs_debug_set_nt_cycles: hipMemcpyToSymbol");
  return 0;
}

extern "C" int hugs_gemm_nt(int dtype, int M, int N, int K1, int K2, const void* A1, int lda1, const void* A2, int lda2,
                            const void* Bt, int ldb, const float* bias, const float* row_bias, int row_div, int ld_rb,
                            int relu, const void* mask, int ld_mask, const float* r1_row, const float* r1_col,
                            void* out, int ldc, void* stream) {
  return gemm_nt_impl(0, dtype, M, N, K1, K2, A1, lda1, A2, lda2, Bt, ldb, bias, row_bias, row_div, ld_rb, relu, mask, ld_mask,
                      r1_row, r1_col, out, ldc, nullptr, nullptr, stream);
}
extern "C" int hugs_gemm_nt_tiles(int tile_mode, int dtype, int M, int N, int K1, int K2, const void* A1, int lda1, const void* A2,
                                  int lda2, const void* Bt, int ldb, const float* bias, const float* row_bias, int row_div, int ld_rb,
                                  int relu, const void* mask, int ld_mask, const float* r1_row, const float* r1_col,
                                  void* out, int ldc, void* stream) {
  HUGS_REQUIRE(tile_mode == 0 || tile_mode == 1 || tile_mode == 3 || tile_mode == 5, -2, "hugs_gemm_nt_tiles: tile_mode %d", tile_mode);
  return gemm_nt_impl(tile_mode, dtype, M, N, K1, K2, A1, lda1, A2, lda2, Bt, ldb, bias, row_bias, row_div, ld_rb, relu, mask,
                      ld_mask, r1_row, r1_col, out, ldc, nullptr, nullptr, stream);
}

// hugs_gemm_nt with 1-bit relu masks (bf16, 256x256-tile kernels only: M, N multiples of 256, ldc == N).  bits_out
// (relu epilogue): M*N/8 bytes written; bits_in: the output is multiplied by the bit (in place of `mask`).
extern "C" long long hugs_gemm_nt_bits_bytes(int M, int N) { return (long long)M * N / 8; }
extern "C" int hugs_gemm_nt_bits(int dtype, int M, int N, int K1, int K2, const void* A1, int lda1, const void* A2, int lda2,
                                 const void* Bt, int ldb, const float* bias, int relu, const float* r1_row,
                                 const float* r1_col, void* out, int ldc, uint32_t* bits_out, const uint32_t* bits_in,
                                 void* stream) {
  HUGS_REQUIRE((dtype == 1 || dtype == 2) && M % 256 == 0 && N % 256 == 0 && ldc == N && (K1 + K2) % 64 == 0 && K1 + K2 >= 128, -3,
               "hugs_gemm_nt_bits: needs bf16 / fp16, M=%d N=%d multiples of 256, ldc == N, K=%d a multiple of 64 and >= 128", M, N, K1 + K2);
  HUGS_REQUIRE(!bits_out || relu, -3, "hugs_gemm_nt_bits: bits_out needs a relu epilogue");
  return gemm_nt_impl(0, dtype, M, N, K1, K2, A1, lda1, A2, lda2, Bt, ldb, bias, nullptr, 1, 0, relu, bits_in ? (const void*)1 : nullptr, 0,
                      r1_row, r1_col, out, ldc, bits_out, bits_in, stream);
}

// One launch for all trunk layers of an MLP (k_gemm_nt_bf16_chain, hugs_gemm_chain.inc): nl layers of [M, N] = relu(A_l W_l^T + b_l) with
// the 1-bit relu masks, layer l+1 reading layer l's output.  tab: 12 host words per layer {A1, A2, Bt, bias, out, bits, lda1, lda2, ldb,
// K1, K2, 0}; flags: nl * M / 256 device words (zeroed here, on the stream).  bf16 only; M, N multiples of 256, N <= 1024 * 8 / nl,
// every K a multiple of 64 and >= 128, every leading dimension a multiple of 512, at least four tiles per CU.  Returns -3 when the shape does not qualify (the caller then
// launches layer by layer).
extern "C" int hugs_gemm_nt_chain(int dtype, int M, int N, int nl, const unsigned long long* tab, unsigned* flags, void* stream) {
  HUGS_REQUIRE(dtype == 1 && nl >= 2 && nl <= HUGS_NT_CHAIN_MAX && M % 256 == 0 && N % 256 == 0 && (long long)nl * N * 4 <= 32768, -3,
               "hugs_gemm_nt_chain: bf16, 2..%d layers, M=%d N=%d multiples of 256, nl * N <= 8192", HUGS_NT_CHAIN_MAX, M, N);
  int dev = 0, ncu = 0;
  HUGS_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess,
               -100, "hugs_gemm_nt_chain: device query");
  ncu &= ~7;
  const int ntiles = (M / 256) * (N / 256);
  HUGS_REQUIRE(ncu >= 8 && ntiles % ncu == 0 && ntiles / ncu >= 4 && (ntiles / 8) % (N / 256) == 0 && (ncu / 8) % (N / 256) == 0, -3,
               "hugs_gemm_nt_chain: %d tiles on %d CUs (needs a whole number >= 4 of tiles per CU, whole row bands per XCD and round)", ntiles, ncu);
  NtChainTab T;
  T.nl = nl; T.pad_ = 0; T.flags = flags;
  for (int l = 0; l < nl; ++l) {
    const unsigned long long* w = tab + 12 * l;
    NtChainLayer& L = T.L[l];
    L.A1 = (const uint16_t*)w[0]; L.A2 = (const uint16_t*)w[1]; L.Bt = (const uint16_t*)w[2]; L.bias = (const float*)w[3];
    L.out = (void*)w[4]; L.bits = (uint32_t*)w[5];
    L.lda1 = (int)w[6]; L.lda2 = (int)w[7]; L.ldb = (int)w[8]; L.K1 = (int)w[9]; L.K2 = (int)w[10]; L.pad_ = 0;
    HUGS_REQUIRE(L.A1 && L.Bt && L.bias && L.out && L.bits && L.K1 > 0 && L.K1 % 32 == 0 && L.K2 % 32 == 0 && (L.K1 + L.K2) % 64 == 0 &&
                 L.K1 + L.K2 >= 128 && (L.K2 == 0 || L.A2) && L.lda1 % 512 == 0 && L.lda2 % 512 == 0 && L.ldb % 512 == 0 && L.lda1 > 0 && L.lda2 > 0 &&
                 (long long)127 * 1024 * (L.ldb >> 9) + 64 < (1ll << 24) * 1 && L.lda1 <= 8192 && L.lda2 <= 8192 && L.ldb <= 8192, -3,
                 "hugs_gemm_nt_chain: layer %d operands / K = %d + %d / leading dimensions %d %d %d (multiples of 512, <= 8192)", l, L.K1, L.K2,
                 L.lda1, L.lda2, L.ldb);
  }
  HUGS_REQUIRE(hipMemsetAsync(flags, 0, (size_t)nl * (M / 256) * sizeof(unsigned), (hipStream_t)stream) == hipSuccess, -100,
               "hugs_gemm_nt_chain: hipMemsetAsync");
  hipLaunchKernelGGL(k_gemm_nt_bf16_chain, dim3(ncu), dim3(512), 0, (hipStream_t)stream, M, N, T, ntiles);
  HUGS_CHECK_LAUNCH("hugs_gemm_nt_chain");
  return 0;
}
#endif  // !HUGS_GEMM_F16

unsigned* hugs_gemm_nt_queue_take();      // (defined once, in the bf16 pass)
int HUGS_NT_IMPL(HUGS_NT_ARGS) {
  HUGS_REQUIRE(dtype == HUGS_OP_DTYPE || (HUGS_OP_DTYPE == 1 && dtype == 0), -2, "hugs_gemm_nt: dtype must be 0 (fp32), 1 (bf16) or 2 (fp16)");
  const int bk = dtype ? GB_BK : 16 /* GF_BK */;
  HUGS_REQUIRE(M % 128 == 0 && N % 128 == 0 && K1 % bk == 0 && K2 % bk == 0 && K1 > 0, -3,
               "hugs_gemm_nt: shape M=%d N=%d K=%d+%d not tile aligned (128,128,%d)", M, N, K1, K2, bk);
  HUGS_REQUIRE(!row_bias || row_div > 0, -3, "hugs_gemm_nt: row_div must be > 0");
  if (M == 0 || N == 0) return 0;
  const bool bits = bits_out || bits_in;
  HUGS_REQUIRE(!bits || (tile_mode != 1 && tile_mode != 3), -3, "hugs_gemm_nt_bits: needs the 256x256 kernels (tile_mode 0 or 5)");
  if (bits_in) mask = nullptr;       // (the EPI_MASK specialisation is selected through `epi_mask` below)
  GemmEpi E{bias, row_bias, row_div, ld_rb, relu, mask, ld_mask, r1_row, r1_col, out, ldc, bits_out, bits_in};
  const int grid = (M / 128) * (N / 128);
  // (four K-stages of 32 are the shortest pipeline the ring kernels run: K >= 128)
  if (dtype && M % 256 == 0 && N % 256 == 0 && (K1 + K2) % 64 == 0 && K1 + K2 >= 128 && tile_mode != 1 && tile_mode != 3)
  {
    // epilogue specialisations for the combinations the trunks use (bit set = term present); anything else -> generic
    const int epi = row_bias ? -1 : (bias ? EPI_BIAS : 0) | (relu ? EPI_RELU : 0) | (mask ? EPI_MASK : 0) | (r1_row ? EPI_R1 : 0) |
                                    (bits_in ? EPI_BIN : 0) | (bits_out ? EPI_BOUT : 0);
    const int ntiles = (M / 256) * (N / 256), nstage = (K1 + K2) / 32;
    static int ncu_of[64] = {0};      // per device (a process may drive GPUs of different sizes)
    int dev = 0;
    HUGS_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, -100, "hugs_gemm_nt: hipGetDevice");
    if (!ncu_of[dev]) {
      int n = 0;
      HUGS_REQUIRE(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess,
                   -100, "hugs_gemm_nt: cannot query the device's CU count");
      n &= ~7;
      ncu_of[dev] = n < 8 ? 8 : n;
    }
    const int ncu = ncu_of[dev];
    // more than one tile per CU: the persistent kernel (ring carried across tiles).  Needs an even number of stages
    // (fragment double buffer parity) and N <= 4096 (bias / r1 vectors in the 32 KiB the ring leaves).
    // (an epilogue combination without a specialisation -- a per-ray row bias, bias without relu + mask, ... -- stays on the
    // one-tile-per-workgroup kernel: the run-time-flag epilogue on top of the persistent loop's live state spilled 44 VGPRs)
    const bool pers_epi = epi == (EPI_BIAS | EPI_RELU) || epi == (EPI_BIAS | EPI_RELU | EPI_BOUT) || epi == EPI_BIN || epi == (EPI_BIN | EPI_R1) ||
                          epi == EPI_BIAS || epi == EPI_MASK || epi == (EPI_MASK | EPI_R1) || epi == 0;
    // (nstage >= 4: a tile of four stages is exactly one ring fill -- the next tile's stages are requested during this tile's four
    //  iterations and the first three waits see the same two younger stages + 16 stores as with longer tiles.  Round 4: the
    //  trunk's output gradient is a K = 128 product, 8192 tiles at the reference-default shape)
    // round 6: grids of at most one tile per CU (the 128-ray step of the fixed-global-batch curve: 16 trunk launches of 256 tiles) also
    // take the whole-line kernel, one tile per workgroup, instead of k_gemm_nt_bf16_big's 32-wide stages: 128 rays 1.166 / 1.170 ->
    // 1.123 / 1.122 ms same box (HUGS_NT_P64_SMALL=0: the old selection)
    const char* small_env = getenv("HUGS_NT_P64_SMALL");
    const char* k64_env0 = getenv("HUGS_NT_K64");
    // (only together with the whole-line kernel: k_gemm_nt_bf16_pers has never run on a grid of less than one tile per CU)
    const bool small_ok = !(small_env && small_env[0] == '0') && !(k64_env0 && k64_env0[0] == '0') && K1 % 64 == 0 && K2 % 64 == 0 &&
                          (K1 + K2) % 512 == 0 && N <= 2048;
    if ((ntiles > ncu || small_ok) && nstage % 2 == 0 && nstage >= 4 && N <= 4096 && tile_mode != 5 && pers_epi) {
      const dim3 gp(ntiles < ncu ? ntiles : ncu), bp(512);
      // round 6: K in 64-wide super-stages of whole cache lines (hugs_gemm_p64.inc) where the K split allows it and the K rotation of
      // the two kernels coincides (K a multiple of 512): bit-identical results.  HUGS_NT_K64=0 keeps the 32-wide stages (A/B switch,
      // read per call).
      const char* k64_env = getenv("HUGS_NT_K64");
      const bool k64 = !(k64_env && k64_env[0] == '0') && K1 % 64 == 0 && K2 % 64 == 0 && (K1 + K2) % 512 == 0 && N <= 2048;      // (N: its arrival counter sits in the rank-1 vector's upper half)
#ifdef HUGS_BUILD_W4
      // HUGS_NT_W4=1: the four-wave form of the same loop (one wave per SIMD, 128 x 128 per wave, hugs_gemm_w4.inc)
      const char* w4_env = getenv("HUGS_NT_W4");
      if (k64 && w4_env && w4_env[0] == '1') {
        const dim3 bp4(256);
#define HUGS_NTP_LAUNCH(EPI_) hipLaunchKernelGGL((k_gemm_nt_bf16_w4<EPI_>), gp, bp4, 0, (hipStream_t)stream, M, N, K1, K2, \
                       (const uint16_t*)A1, lda1, (const uint16_t*)A2, lda2, (const uint16_t*)Bt, ldb, E, ntiles)
        switch (epi) {
          case EPI_BIAS | EPI_RELU: HUGS_NTP_LAUNCH(EPI_BIAS | EPI_RELU); break;
          case EPI_BIAS | EPI_RELU | EPI_BOUT: HUGS_NTP_LAUNCH(EPI_BIAS | EPI_RELU | EPI_BOUT); break;
          case EPI_BIN: HUGS_NTP_LAUNCH(EPI_BIN); break;
          case EPI_BIN | EPI_R1: HUGS_NTP_LAUNCH(EPI_BIN | EPI_R1); break;
          case EPI_BIAS: HUGS_NTP_LAUNCH(EPI_BIAS); break;
          case EPI_MASK: HUGS_NTP_LAUNCH(EPI_MASK); break;
          case EPI_MASK | EPI_R1: HUGS_NTP_LAUNCH(EPI_MASK | EPI_R1); break;
          default: HUGS_NTP_LAUNCH(0); break;      // epi == 0
        }
#undef HUGS_NTP_LAUNCH
        HUGS_CHECK_LAUNCH("hugs_gemm_nt(persistent, K64, four waves)");
        return 0;
      }
#endif
      // dynamic tile queue per XCD (hugs_gemm_dq.inc) inside a hugs_gemm_nt_queue_begin / _end pair: at least two tiles per workgroup,
      // whole XCD groups, at least eight super-stages per tile (HUGS_NT_DYNQ=0: off)
      const char* dq_env = getenv("HUGS_NT_DYNQ");
      if (k64 && !(dq_env && dq_env[0] == '0') && ntiles >= 2 * ncu && ncu % 8 == 0 && (K1 + K2) >= 512) {
        unsigned* const slot = hugs_gemm_nt_queue_take();
        if (slot) {
#define HUGS_NTP_LAUNCH(EPI_) hipLaunchKernelGGL((k_gemm_nt_bf16_dq<EPI_>), gp, bp, 0, (hipStream_t)stream, M, N, K1, K2, \
                       (const uint16_t*)A1, lda1, (const uint16_t*)A2, lda2, (const uint16_t*)Bt, ldb, E, ntiles, slot)
          switch (epi) {
            case EPI_BIAS | EPI_RELU: HUGS_NTP_LAUNCH(EPI_BIAS | EPI_RELU); break;
            case EPI_BIAS | EPI_RELU | EPI_BOUT: HUGS_NTP_LAUNCH(EPI_BIAS | EPI_RELU | EPI_BOUT); break;
            case EPI_BIN: HUGS_NTP_LAUNCH(EPI_BIN); break;
            case EPI_BIN | EPI_R1: HUGS_NTP_LAUNCH(EPI_BIN | EPI_R1); break;
            case EPI_BIAS: HUGS_NTP_LAUNCH(EPI_BIAS); break;
            case EPI_MASK: HUGS_NTP_LAUNCH(EPI_MASK); break;
            case EPI_MASK | EPI_R1: HUGS_NTP_LAUNCH(EPI_MASK | EPI_R1); break;
            default: HUGS_NTP_LAUNCH(0); break;      // epi == 0
          }
#undef HUGS_NTP_LAUNCH
          HUGS_CHECK_LAUNCH("hugs_gemm_nt(persistent, K64, tile queue)");
          return 0;
        }
      }
      if (k64) {
#define HUGS_NTP_LAUNCH(EPI_) hipLaunchKernelGGL((k_gemm_nt_bf16_p64<EPI_>), gp, bp, 0, (hipStream_t)stream, M, N, K1, K2, \
                       (const uint16_t*)A1, lda1, (const uint16_t*)A2, lda2, (const uint16_t*)Bt, ldb, E, ntiles)
        switch (epi) {
          case EPI_BIAS | EPI_RELU: HUGS_NTP_LAUNCH(EPI_BIAS | EPI_RELU); break;
          case EPI_BIAS | EPI_RELU | EPI_BOUT: HUGS_NTP_LAUNCH(EPI_BIAS | EPI_RELU | EPI_BOUT); break;
          case EPI_BIN: HUGS_NTP_LAUNCH(EPI_BIN); break;
          case EPI_BIN | EPI_R1: HUGS_NTP_LAUNCH(EPI_BIN | EPI_R1); break;
          case EPI_BIAS: HUGS_NTP_LAUNCH(EPI_BIAS); break;
          case EPI_MASK: HUGS_NTP_LAUNCH(EPI_MASK); break;
          case EPI_MASK | EPI_R1: HUGS_NTP_LAUNCH(EPI_MASK | EPI_R1); break;
          default: HUGS_NTP_LAUNCH(0); break;      // epi == 0
        }
#undef HUGS_NTP_LAUNCH
        HUGS_CHECK_LAUNCH("hugs_gemm_nt(persistent, K64)");
        return 0;
      }
#define HUGS_NTP_LAUNCH(EPI_) hipLaunchKernelGGL((k_gemm_nt_bf16_pers<EPI_>), gp, bp, 0, (hipStream_t)stream, M, N, K1, K2, \
                       (const uint16_t*)A1, lda1, (const uint16_t*)A2, lda2, (const uint16_t*)Bt, ldb, E, ntiles)
      switch (epi) {
        case EPI_BIAS | EPI_RELU: HUGS_NTP_LAUNCH(EPI_BIAS | EPI_RELU); break;
        case EPI_BIAS | EPI_RELU | EPI_BOUT: HUGS_NTP_LAUNCH(EPI_BIAS | EPI_RELU | EPI_BOUT); break;   // forward trunk, mask bits out
        case EPI_BIN: HUGS_NTP_LAUNCH(EPI_BIN); break;                                                 // dX, mask bits in
        case EPI_BIN | EPI_R1: HUGS_NTP_LAUNCH(EPI_BIN | EPI_R1); break;                               // G of the last trunk layer
        case EPI_BIAS: HUGS_NTP_LAUNCH(EPI_BIAS); break;
        case EPI_MASK: HUGS_NTP_LAUNCH(EPI_MASK); break;
        case EPI_MASK | EPI_R1: HUGS_NTP_LAUNCH(EPI_MASK | EPI_R1); break;
        default: HUGS_NTP_LAUNCH(0); break;      // epi == 0
      }
#undef HUGS_NTP_LAUNCH
      HUGS_CHECK_LAUNCH("hugs_gemm_nt(persistent)");
      return 0;
    }
    const dim3 g((M / 256) * (N / 256)), b(512);
#define HUGS_NT_LAUNCH(EPI_) hipLaunchKernelGGL((k_gemm_nt_bf16_big<4, EPI_>), g, b, 0, (hipStream_t)stream, M, N, K1, K2, \
                       (const uint16_t*)A1, lda1, (const uint16_t*)A2, lda2, (const uint16_t*)Bt, ldb, E)
    switch (epi) {
      case EPI_BIAS | EPI_RELU: HUGS_NT_LAUNCH(EPI_BIAS | EPI_RELU); break;   // forward trunk layer
      case EPI_BIAS | EPI_RELU | EPI_BOUT: HUGS_NT_LAUNCH(EPI_BIAS | EPI_RELU | EPI_BOUT); break;
      case EPI_BIN: HUGS_NT_LAUNCH(EPI_BIN); break;
      case EPI_BIN | EPI_R1: HUGS_NT_LAUNCH(EPI_BIN | EPI_R1); break;
      case EPI_BIAS: HUGS_NT_LAUNCH(EPI_BIAS); break;                          // bottleneck
      case EPI_MASK: HUGS_NT_LAUNCH(EPI_MASK); break;                          // dX through a relu
      case EPI_MASK | EPI_R1: HUGS_NT_LAUNCH(EPI_MASK | EPI_R1); break;        // G of the last trunk layer
      case 0: HUGS_NT_LAUNCH(0); break;                                        // dBottleneck
      default: HUGS_NT_LAUNCH(-1); break;
    }
#undef HUGS_NT_LAUNCH
  }
  else if (dtype && M % 256 == 0 && (K1 + K2) % 64 == 0 && K1 + K2 >= 128 && tile_mode != 1)
    hipLaunchKernelGGL(k_gemm_nt_bf16_big<2>, dim3((M / 256) * (N / 128)), dim3(256), 0, (hipStream_t)stream, M, N, K1, K2,
                       (const uint16_t*)A1, lda1, (const uint16_t*)A2, lda2, (const uint16_t*)Bt, ldb, E);
  else if (dtype)
    hipLaunchKernelGGL(k_gemm_nt_bf16, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, N, K1, K2, (const uint16_t*)A1,
                       lda1, (const uint16_t*)A2, lda2, (const uint16_t*)Bt, ldb, E);
#ifndef HUGS_GEMM_F16
  else
    hipLaunchKernelGGL(k_gemm_nt_f32, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, N, K1, K2, (const float*)A1, lda1,
                       (const float*)A2, lda2, (const float*)Bt, ldb, E);
#endif
  HUGS_CHECK_LAUNCH("hugs_gemm_nt");
  return 0;
}

// ---- batched TN (include/hugs.h hugs_gemm_tn_batch) ----
struct HugsTnItem { const void* X; const void* G; float* dW; float* dbias; int ldx, ldg, Mrows, Kc, N, reserved; };
#ifdef HUGS_GEMM_F16
#define HUGS_TNB_IMPL hugs_gemm_tn_batch_impl_f16
#else
#define HUGS_TNB_IMPL hugs_gemm_tn_batch_impl_bf16
#endif
int hugs_gemm_tn_batch_impl_bf16(int nitems, const HugsTnItem* items, int nsplit, void* ws, void* stream);
int hugs_gemm_tn_batch_impl_f16(int nitems, const HugsTnItem* items, int nsplit, void* ws, void* stream);
int HUGS_TNB_IMPL(int nitems, const HugsTnItem* items, int nsplit, void* ws, void* stream) {
  TnBatch B;
  RedBatch R;
  B.nitems = nitems; B.nsplit = nsplit;
  R.nitems = 0; R.nsplit = nsplit;
  float* w = (float*)ws;
  // group arrival counters (one per (item, piece)) in front of the slabs: zeroed by a memset node ahead of the launch.  Only where
  // a group is more than one workgroup and the loop is long enough to drift (HUGS_TN_BATCH_SYNC=0: off)
  static const bool sync_on = []() { const char* e = getenv("HUGS_TN_BATCH_SYNC"); return !(e && e[0] == '0'); }();
  bool want_sync = sync_on && nitems * nsplit <= 256;
  for (int i = 0; i < nitems; ++i)
    if ((items[i].Kc >> 8) * (items[i].N >> 8) < 2 || (items[i].Mrows >> 5) / nsplit <= 2 * HUGS_TN_SYNC) want_sync = false;
  B.sync = nullptr;
  if (want_sync) {
    B.sync = (unsigned*)w;
    if (hipMemsetAsync(w, 0, 256 * sizeof(float), (hipStream_t)stream) != hipSuccess) B.sync = nullptr;
  }
  w += 256;
  int wg = 0;
  unsigned blk = 0;
  for (int i = 0; i < nitems; ++i) {
    const HugsTnItem& h = items[i];
    TnBatchItem& t = B.it[i];
    t.X = (const uint16_t*)h.X; t.G = (const uint16_t*)h.G; t.ldx = h.ldx; t.ldg = h.ldg; t.Kc = h.Kc; t.N = h.N;
    t.units = h.Mrows >> 6; t.wg0 = wg; t.tiles = (h.Kc >> 8) * (h.N >> 8); t.pad_ = 0;
    t.slab = w; w += (size_t)nsplit * h.Kc * h.N;
    t.colsum = nullptr;
    if (h.dbias) { t.colsum = w; w += (size_t)nsplit * h.N; }
    wg += t.tiles * nsplit;
    RedItem& r = R.it[R.nitems++];
    r.slab = t.slab; r.out = h.dW; r.per = (unsigned long long)h.Kc * h.N; r.blk0 = blk; r.pad_ = 0;
    blk += (unsigned)((r.per / 4 + 255) / 256);
    if (h.dbias) {
      RedItem& q = R.it[R.nitems++];
      q.slab = t.colsum; q.out = h.dbias; q.per = (unsigned long long)h.N; q.blk0 = blk; q.pad_ = 0;
      blk += (unsigned)((q.per / 4 + 255) / 256);
    }
  }
  hipLaunchKernelGGL(k_gemm_tn_bf16_batch, dim3(wg), dim3(512), 0, (hipStream_t)stream, B);
  HUGS_CHECK_LAUNCH("hugs_gemm_tn_batch");
  hipLaunchKernelGGL(k_slab_reduce_batch, dim3(blk), dim3(256), 0, (hipStream_t)stream, R);
  HUGS_CHECK_LAUNCH("hugs_gemm_tn_batch(reduce)");
  return 0;
}

#ifndef HUGS_GEMM_F16
static int tn_batch_check(int dtype, int nitems, const HugsTnItem* items) {
  HUGS_REQUIRE(dtype == 1 || dtype == 2, -2, "hugs_gemm_tn_batch: dtype must be 1 (bf16) or 2 (fp16), got %d", dtype);
  HUGS_REQUIRE(items && nitems >= 1 && nitems <= HUGS_TN_BATCH_MAX, -3, "hugs_gemm_tn_batch: 1..%d items, got %d", HUGS_TN_BATCH_MAX, nitems);
  for (int i = 0; i < nitems; ++i) {
    const HugsTnItem& h = items[i];
    HUGS_REQUIRE(h.X && h.G && h.dW && h.Kc > 0 && h.N > 0 && h.Kc % 256 == 0 && h.N % 256 == 0 && h.Mrows >= 512 && h.Mrows % 64 == 0 &&
                     h.ldx % 8 == 0 && h.ldg % 8 == 0 && h.ldx >= h.Kc && h.ldg >= h.N, -3,
                 "hugs_gemm_tn_batch: item %d: rows=%d Kc=%d N=%d ldx=%d ldg=%d (Kc, N multiples of 256; rows a multiple of 64, >= 512)", i,
                 h.Mrows, h.Kc, h.N, h.ldx, h.ldg);
  }
  return 0;
}
// nsplit the launch would use for nsplit_request <= 0: #CUs / (total tiles), at least 1, at most rows/512 of the shortest item
extern "C" int hugs_gemm_tn_batch_nsplit(int nitems, const HugsTnItem* items) {
  if (!items || nitems < 1) return 1;
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
  long long tiles = 0;
  int cap = 1 << 30;
  for (int i = 0; i < nitems; ++i) {
    tiles += (long long)(items[i].Kc >> 8) * (items[i].N >> 8);
    if (items[i].Mrows / 512 < cap) cap = items[i].Mrows / 512;
  }
  long long ns = tiles > 0 ? ncu / tiles : 1;
  if (ns > cap) ns = cap;
  return ns < 1 ? 1 : (int)ns;
}
extern "C" long long hugs_gemm_tn_batch_ws_bytes(int nitems, const HugsTnItem* items, int nsplit) {
  long long n = 0;
  for (int i = 0; i < nitems; ++i) n += (long long)nsplit * ((long long)items[i].Kc * items[i].N + items[i].N);
  return (n + 256) * 4;      // + the tile groups' arrival counters
}
extern "C" int hugs_gemm_tn_batch(int dtype, int nitems, const HugsTnItem* items, int nsplit, void* ws, void* stream) {
  if (int rc = tn_batch_check(dtype, nitems, items)) return rc;
  HUGS_REQUIRE(ws && nsplit >= 1, -3, "hugs_gemm_tn_batch: nsplit %d (use hugs_gemm_tn_batch_nsplit), ws %p", nsplit, ws);
  for (int i = 0; i < nitems; ++i)
    HUGS_REQUIRE(items[i].Mrows / nsplit >= 512, -3, "hugs_gemm_tn_batch: item %d: %d rows in %d pieces: fewer than 512 rows per piece", i,
                 items[i].Mrows, nsplit);
  return dtype == 2 ? hugs_gemm_tn_batch_impl_f16(nitems, items, nsplit, ws, stream)
                    : hugs_gemm_tn_batch_impl_bf16(nitems, items, nsplit, ws, stream);
}

extern "C" long long hugs_gemm_tn_ws_bytes(int Kc, int N, int nsplit) {
  return (long long)nsplit * ((long long)Kc * N + N) * 4;
}

// dW[Kc, ldw(:N)] = X^T G, dbias[N] = colsum(G) (if dbias != null). ws: hugs_gemm_tn_ws_bytes().
extern "C" int hugs_gemm_tn(int dtype, int Mrows, int Kc, int N, int nsplit, const void* X, int ldx, const void* G, int ldg,
                            float* dW, float* dbias, void* ws, void* stream) {
  return gemm_tn_impl(0, dtype, Mrows, Kc, N, nsplit, X, ldx, G, ldg, dW, dbias, ws, stream);
}
extern "C" int hugs_gemm_tn_tiles(int tile_mode, int dtype, int Mrows, int Kc, int N, int nsplit, const void* X, int ldx, const void* G,
                                  int ldg, float* dW, float* dbias, void* ws, void* stream) {
  HUGS_REQUIRE(tile_mode == 0 || tile_mode == 1, -2, "hugs_gemm_tn_tiles: tile_mode %d (0 default, 1 force 128x128)", tile_mode);
  return gemm_tn_impl(tile_mode, dtype, Mrows, Kc, N, nsplit, X, ldx, G, ldg, dW, dbias, ws, stream);
}
#endif  // !HUGS_GEMM_F16

int HUGS_TN_IMPL(HUGS_TN_ARGS) {
  HUGS_REQUIRE(dtype == HUGS_OP_DTYPE || (HUGS_OP_DTYPE == 1 && dtype == 0), -2, "hugs_gemm_tn: dtype must be 0 (fp32), 1 (bf16) or 2 (fp16)");
  const int step = dtype ? 64 : 16;
  HUGS_REQUIRE(Kc % 128 == 0 && N % 128 == 0 && nsplit >= 1 && Mrows % (nsplit * step) == 0, -3,
               "hugs_gemm_tn: shape rows=%d Kc=%d N=%d split=%d not tile aligned", Mrows, Kc, N, nsplit);
  float* slab = (float*)ws;
  float* cs = dbias ? slab + (size_t)nsplit * Kc * N : nullptr;
  const int grid = (Kc / 128) * (N / 128) * nsplit;
  const int rows_per = Mrows / nsplit;
  // (a single 256x256 tile qualifies too when the caller splits the rows deep enough: nerfacto's 256-wide field layers)
  if (dtype && Kc % 256 == 0 && N % 256 == 0 && ((Kc / 256) * (N / 256) >= 4 || rows_per >= 2048) && rows_per % 64 == 0 &&
      rows_per >= 256 && tile_mode != 1)
    hipLaunchKernelGGL(k_gemm_tn_bf16_big, dim3((Kc / 256) * (N / 256) * nsplit), dim3(512), 0, (hipStream_t)stream, Mrows, Kc,
                       N, nsplit, (const uint16_t*)X, ldx, (const uint16_t*)G, ldg, slab, N, cs);
  else if (dtype)
    hipLaunchKernelGGL(k_gemm_tn_bf16, dim3(grid), dim3(256), 0, (hipStream_t)stream, Mrows, Kc, N, nsplit,
                       (const uint16_t*)X, ldx, (const uint16_t*)G, ldg, slab, N, cs);
#ifndef HUGS_GEMM_F16
  else
    hipLaunchKernelGGL(k_gemm_tn_f32, dim3(grid), dim3(256), 0, (hipStream_t)stream, Mrows, Kc, N, nsplit, (const float*)X,
                       ldx, (const float*)G, ldg, slab, N, cs);
#endif
  HUGS_CHECK_LAUNCH("hugs_gemm_tn");
  const size_t per = (size_t)Kc * N;
  const unsigned nb1 = (unsigned)((per / 4 + 255) / 256), nb2 = dbias ? (unsigned)((N / 4 + 255) / 256) : 0u;
  hipLaunchKernelGGL(k_slab_reduce, dim3(nb1 + nb2), dim3(256), 0, (hipStream_t)stream, slab, nsplit, per, dW, nb1,
                     (const float*)cs, (size_t)N, dbias);
  HUGS_CHECK_LAUNCH("hugs_gemm_tn(reduce)");
  return 0;
}

// Fused forward of the 256-wide trunk layers 1 .. depth-1 of an MLP + its density head (round 4): the PropMLP of the
// reference (MipNeRF360/internal/models.py:451-456 Dense + relu, :456 raw_density, :467 softplus) after its first layer.
//
//   Y_l = relu(Y_{l-1} W_l + b_l)  for l = 1 .. nl      (every W_l is [256, 256], operands bf16, fp32 accumulation)
//   raw = Y_nl . w_d + b_d ;  density = softplus(raw + density_bias)
//
// A layer of this width is HBM-bound as a GEMM launch (0.5 GB in + 0.5 GB out per 1 M rows: 220-230 us at 4.6 TB/s) and the
// activations are read back once per layer.  Here a workgroup keeps a 128-row activation tile in LDS from layer to layer
// (two 64 KiB buffers in the K-stage layout the MFMA fragments are read from), streams the 128 KB weight matrix of a layer
// from L2 straight into registers (each wave its 64 output columns, three K-stages ahead), and writes every Y_l -- the
// backward pass needs them all -- plus the 1-bit relu masks in the 256x256 NT kernels' lane layout exactly once; nothing is
// read back.  The density head is the last layer's epilogue (on the bf16-rounded activations, as the stand-alone kernel reads
// them).  Wave (wm, wn) of the 8 owns rows wm*64.. and columns wn*64..: 4 x 4 fragments of v_mfma_f32_16x16x32_bf16 with the
// weights as the A operand, so that a lane holds 4 consecutive output columns of one row (hugs_gemm.hip's convention: the
// mask-bit layout and the K-stage LDS layout are that file's).
#include "hugs_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 mf_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float mf_f32x4_t;

#define MF_MAXL 7
struct MlpTail {
  int M, nl;
  const uint16_t* Y0;              // [M, 256] input of layer 1
  const uint16_t* Wt[MF_MAXL];     // [256 (n), 256 (k)] per layer
  const float* bias[MF_MAXL];      // [256]
  uint16_t* Y[MF_MAXL];            // [M, 256] outputs
  uint32_t* bits[MF_MAXL];         // 1-bit relu masks (M*256/8 bytes) or null
  const float* wd;                 // [256] density head (null: no head)
  const float* bd;
  float density_bias;
  float* raw;
  float* density;
};

__device__ __forceinline__ float mf_softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ uint32_t mf_cvt_pk(float a, float b) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
  typedef float __attribute__((ext_vector_type(2))) f2; typedef __bf16 __attribute__((ext_vector_type(2))) h2;
  const f2 f = {a, b};
  const h2 h = __builtin_convertvector(f, h2);
  return *(const uint32_t*)&h;
}

// WM = waves along M: 2 -> 128-row tile, 8 waves, one workgroup per CU (130 KiB of LDS); 1 -> 64-row tile, 4 waves, TWO workgroups
// per CU (66 KiB each): the epilogue / tile load of one overlaps the MFMAs of the other (what ships: 1)
template <int WM>
__global__ __launch_bounds__(256 * WM, 2) void k_mlp256_tail_fwd(const MlpTail P) {
  constexpr int MF_ROWS = 64 * WM, MF_STAGE = MF_ROWS * 64, MF_ACT = 8 * MF_STAGE;      // 8 K-stages of [rows x 32 k] bf16
  constexpr int NT_ = 256 * WM;
  __shared__ __attribute__((aligned(16))) unsigned char act[2][MF_ACT];
  __shared__ float dred[MF_ROWS][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;
  const int r16 = lane & 15, kb = lane >> 4;
  const int swz = 3 * ((r16 >> 2) & 1);
  const int ntile = P.M / MF_ROWS;
  // fragment read offset inside a stage (hugs_gemm.hip frag_off): row r16 of a 16-row block, 16-byte chunk kb (XOR-swizzled)
  const int frag_off = (wm * 64 + r16) * 64 + ((kb ^ swz) << 4);

  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int m0 = t * MF_ROWS;
    // weights three K-stages ahead in registers; the first three stages of a layer are requested BEFORE the previous layer's
    // epilogue (for layer 0: before the tile load is waited for), so that their L2 latency hides under it
    mf_bf16x8_t wq[4][4];
    const size_t wlane = (size_t)(wn * 64 + r16) * 256 + kb * 8;      // lane's row of fragment j: + j*16*256; stage s: + s*32
    auto prefetch3 = [&](int l_) {
      const uint16_t* W_ = P.Wt[l_] + wlane;
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[s][j] = *(const mf_bf16x8_t*)(W_ + (size_t)j * 16 * 256 + s * 32);
    };
    prefetch3(0);
    __syncthreads();      // (the previous tile's readers of act[] / dred are done)
    // ---- Y0 tile -> act[0] in the stage layout: chunk p = it*512 + tid -> row p>>5, 16-byte chunk p&31 of the 512-byte row
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int p = it * NT_ + tid, row = p >> 5, cc = p & 31;
      const uint4 v = *(const uint4*)(P.Y0 + (size_t)(m0 + row) * 256 + cc * 8);
      *(uint4*)(act[0] + (cc >> 2) * MF_STAGE + row * 64 + (((cc & 3) ^ (3 * ((row >> 2) & 1))) << 4)) = v;
    }
    __syncthreads();
    for (int l = 0; l < P.nl; ++l) {
      const unsigned char* A = act[l & 1];
      unsigned char* An = act[(l + 1) & 1];
      const uint16_t* W = P.Wt[l] + wlane;
      mf_f32x4_t acc[4][4];
      {
        const float* b = P.bias[l] + wn * 64 + kb * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 bb = *(const float4*)(b + j * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = mf_f32x4_t{bb.x, bb.y, bb.z, bb.w};
        }
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#ifndef MF_NOWLOAD
        if (s + 3 < 8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wq[(s + 3) & 3][j] = *(const mf_bf16x8_t*)(W + (size_t)j * 16 * 256 + (s + 3) * 32);
        }
#endif
        mf_bf16x8_t xa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[i] = *(const mf_bf16x8_t*)(A + s * MF_STAGE + frag_off + i * 16 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s & 3][j], xa[i], acc[i][j], 0, 0, 0);
      }
      // ---- epilogue: relu, bf16, -> next layer's LDS tile + HBM + mask bits (+ the density head on the last layer)
      const bool last = l + 1 == P.nl;
      if (!last) prefetch3(l + 1);
      uint16_t* Yout = P.Y[l] + (size_t)(m0 + wm * 64 + r16) * 256 + wn * 64 + kb * 4;
      uint32_t* bout = P.bits[l];
      // bits: NT tile = 256 rows x 256 columns, its wave (wm_nt, wn) covers 128 rows = fragment rows i_nt 0..7; row m0 + wm*64 + i*16
      // is i_nt = ((m0 >> 6) & 1) * 4 + wm*4 + i of wave wm_nt = (m0 >> 7) & 1   (m0 is a multiple of 64 * WM)
      const int i_nt0 = (((m0 >> 6) & 1) + wm) * 4;
      const size_t bits_at = ((size_t)(m0 >> 8) * 8 + (size_t)((((m0 + wm * 64) >> 7) & 1) * 4 + wn)) * 256 + (size_t)lane;
      float4 wdv[4];
      if (last && P.wd) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wdv[j] = *(const float4*)(P.wd + wn * 64 + j * 16 + kb * 4);
      }
      uint32_t bw = 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = fmaxf(acc[i][j][0], 0.f), x1 = fmaxf(acc[i][j][1], 0.f), x2 = fmaxf(acc[i][j][2], 0.f), x3 = fmaxf(acc[i][j][3], 0.f);
          uint2 u;
          u.x = mf_cvt_pk(x0, x1); u.y = mf_cvt_pk(x2, x3);
#ifndef MF_NOSTORE
          *(uint2*)(Yout + (size_t)i * 16 * 256 + j * 16) = u;
#endif
          if (!last) {
            // next layer's k = this layer's column n = wn*64 + j*16 + kb*4: stage n>>5, chunk (n&31)>>3, half (n>>2)&1
            const int st = wn * 2 + (j >> 1), ch = (j & 1) * 2 + (kb >> 1);
            *(uint2*)(An + st * MF_STAGE + (wm * 64 + i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
          }
          if (bout) {       // hugs_gemm.hip nt_epilogue_direct: word (i_nt >> 1), pair k = (i_nt & 1)*8 + j*2 (+1 for the second pair)
            const int k = (i & 1) * 8 + j * 2;
            bw |= ((u.x + 0x7fff7fffu) >> (15 - k)) & (0x00010001u << k);
            bw |= ((u.y + 0x7fff7fffu) >> (14 - k)) & (0x00010001u << (k + 1));
          }
          if (last && P.wd) {
            dsum += __uint_as_float(u.x << 16) * wdv[j].x + __uint_as_float(u.x & 0xffff0000u) * wdv[j].y +
                    __uint_as_float(u.y << 16) * wdv[j].z + __uint_as_float(u.y & 0xffff0000u) * wdv[j].w;
          }
        }
        if (bout && (i & 1)) { bout[bits_at + (size_t)(((i_nt0 & 7) + i) >> 1) * 64] = bw; bw = 0u; }
        if (last && P.wd) {
          dsum += __shfl_xor(dsum, 16);
          dsum += __shfl_xor(dsum, 32);
          if (kb == 0) dred[wm * 64 + i * 16 + r16][wn] = dsum;
        }
      }
      __syncthreads();      // the next layer's tile is complete (and this layer's readers of the old one are long done)
      if (last && P.wd && tid < MF_ROWS) {
        const float r = ((dred[tid][0] + dred[tid][1]) + (dred[tid][2] + dred[tid][3])) + P.bd[0];
        P.raw[m0 + tid] = r;
        P.density[m0 + tid] = mf_softplus(r + P.density_bias);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Round 5: the same three-layer chain (the reference's PropMLP, net_depth 4: layers 1..3 + the density head) with the WEIGHTS
// RESIDENT IN REGISTERS -- csrc/hugs_fieldfuse.hip's design (one workgroup per CU, one wave per SIMD, 512 registers per lane).
// Wave wn owns output columns [64 wn, 64 wn + 64) of all 64 rows of a tile, i.e. a [64 x 256] slice of every weight matrix =
// 128 registers per layer: layers 1 and 2 fill the 256 accumulation registers (MFMA takes its A operand from there), layer 3
// sits in 128 of the 256 architectural ones; the kernel loads its 384 KB of weights ONCE.  (k_mlp256_tail_fwd above re-streams
// 128 KB per layer and 64-row tile from L2 -- 6 GB of L2 reads per 1 M rows -- and loses to the per-layer GEMMs above 32 k rows.)
// Per tile: the Y0 rows arrive by LDS-DMA (global_load_lds, requested a tile ahead, lane-linear image of the K-stage layout
// through the source-side swizzle of hugs_gemm.hip), the activations go X -> P -> Q -> R through four 32 KiB LDS tiles, every
// finished tile leaves as whole 512-byte rows (16 B per lane) from inside the NEXT layer's MFMA loop, the 1-bit relu masks in
// hugs_gemm_nt_bits' layout, the density head as three hi / lo / third-slice MFMAs per row block on the packed outputs.
// Loads and stores retire through one in-order counter: the only load of the tile loop (the next tile's DMA) is waited for with
// a COUNTED vmcnt (the 20 stores issued behind it stay in flight).
// ------------------------------------------------------------------------------------------------
#define MC_ROWS 64
#define MC_STAGE (MC_ROWS * 64)
#define MC_ACT (8 * MC_STAGE)

__device__ __forceinline__ unsigned mc_fresh(unsigned x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ uint32_t mc_relu_pk(uint32_t u) {
  typedef short __attribute__((ext_vector_type(2))) s2;
  const s2 z = {0, 0};
  const s2 r = __builtin_elementwise_max(__builtin_bit_cast(s2, u), z);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t mc_nz_pk(uint32_t u) {      // 1 in each non-zero 16-bit half: one v_pk_min_u16
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(u));
  return r;
}
// this thread's share of a finished [64 x 256] tile, LDS (K-stage layout) -> 64 rows of 512 bytes.  Chunk id = q * 256 + tid is
// row q * 8 + (tid >> 5), 16-byte chunk cc = tid & 31 of the row: the swizzle bit (row >> 2) & 1 does not depend on q, so ONE LDS
// lane offset (+ 512 q as an immediate) and ONE global lane offset (on a scalar base advanced by 4096 q) serve all eight copies.
__device__ __forceinline__ void mc_copy_out(const unsigned char* src, char* dst_tile, int tid) {
  typedef unsigned __attribute__((ext_vector_type(4))) u4;
  tid = (int)mc_fresh((unsigned)tid);      // (nothing derived from it may be hoisted out of the tile loop and kept live across it)
  const int r0 = tid >> 5, cc = tid & 31;
  const unsigned lo = mc_fresh((unsigned)((cc >> 2) * MC_STAGE + r0 * 64 + (((cc & 3) ^ (3 * ((r0 >> 2) & 1))) << 4)));
  const unsigned go = mc_fresh((unsigned)(r0 * 512 + cc * 16));
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const u4 v = *(const u4*)(src + lo + q * 512);
    __builtin_nontemporal_store(v, (u4*)(dst_tile + (size_t)q * 4096 + go));
  }
}

// One relu layer of a tile for this wave, 16-row block by 16-row block: the MFMAs of block i + 1 are issued ahead of the epilogue
// of block i.  HEAD: + the density head's partial sums (kb == 0 lanes: row r16 of the block) -> red[row][wn].
template <bool HEAD>
__device__ __forceinline__ void mc_layer(const unsigned char* A, int frag_off, const mf_bf16x8_t (&w)[8][4], const float* bias /* LDS, + wn*64 + kb*4 */,
                                         int m0, int wn, int r16, int kb, int lane, unsigned char* An, uint32_t* bout, bool do_cp,
                                         const unsigned char* cp_src, char* cp_dst, int tid, const unsigned char* c2f_lds, float (*red)[4]) {
  const int swz = 3 * ((r16 >> 2) & 1);
  mf_f32x4_t acc[2][4];
  // The MFMAs of row block i + 1 run in two halves of four K-stages; the epilogue of row block i is split the same way (output
  // fragments j = 0, 1 / j = 2, 3) and each half shares a scheduling region with one half of the MFMAs, interleaved by
  // sched_group_barrier (1 MFMA : 5 vector instructions): one wave per SIMD has nobody else to fill the 12 issue cycles between
  // two MFMAs, and left to itself the scheduler issues the 16 MFMAs back to back and the epilogue behind them (measured before the
  // interleave: 20 k cycles per tile for 6 k cycles of MFMAs).
  auto acc_init = [&](mf_f32x4_t (&a)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 bb = *(const float4*)(bias + j * 16); a[j] = mf_f32x4_t{bb.x, bb.y, bb.z, bb.w};
      // (into the accumulator's own architectural registers: as the first MFMA's C operand the bias was read into 16 accumulation
      //  registers -- all 256 of which hold weights -- and two weight fragments went to scratch, reloaded in every tile)
      asm volatile("" : "+v"(a[j]));
    }
  };
  auto mma_half = [&](int i, int h, mf_f32x4_t (&a)[4]) {
    mf_bf16x8_t xa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xa[s] = *(const mf_bf16x8_t*)(A + (h * 4 + s) * MC_STAGE + frag_off + i * 16 * 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[h * 4 + s][j], xa[s], a[j], 0, 0, 0);
  };
  const unsigned bo = mc_fresh((unsigned)lane * 4u);
  // hugs_gemm.hip nt_epilogue_direct bit layout: NT tile = 256 rows; its wave (wm_nt, wn) covers 128 rows = fragment rows i_nt 0..7
  const int i_nt0 = ((m0 >> 6) & 1) * 4;
  uint32_t* btile = bout + ((size_t)(m0 >> 8) * 8 + (size_t)(((m0 >> 7) & 1) * 4 + wn)) * 256;      // (uniform)
  uint32_t bw = 0u;
  uint32_t uk[4][2];
  auto epi_half = [&](int i, int jh) {
#pragma unroll
    for (int j = 2 * jh; j < 2 * jh + 2; ++j) {
      const mf_f32x4_t v = acc[i & 1][j];
      uint2 u;
      u.x = mc_relu_pk(mf_cvt_pk(v[0], v[1])); u.y = mc_relu_pk(mf_cvt_pk(v[2], v[3]));
      uk[j][0] = u.x; uk[j][1] = u.y;
      const int st = wn * 2 + (j >> 1), ch = (j & 1) * 2 + (kb >> 1);
      *(uint2*)(An + st * MC_STAGE + (i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
      const int k = (i & 1) * 8 + j * 2;
      bw |= mc_nz_pk(u.x) << k;
      bw |= mc_nz_pk(u.y) << (k + 1);
    }
  };
  auto interleave = [&]() {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // five vector ALU instructions
      if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // an LDS write every other group
    }
  };
  acc_init(acc[0]);
  mma_half(0, 0, acc[0]);
  mma_half(0, 1, acc[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < 3) { acc_init(acc[(i + 1) & 1]); mma_half(i + 1, 0, acc[(i + 1) & 1]); }
    if (i == 0 && do_cp) mc_copy_out(cp_src, cp_dst, tid);
    epi_half(i, 0);
    if (i < 3) interleave();
    if (i < 3) mma_half(i + 1, 1, acc[(i + 1) & 1]);
    epi_half(i, 1);
    if (i & 1) { *(uint32_t*)((char*)btile + (bo + (unsigned)(((i_nt0 + i) >> 1) * 64) * 4u)) = bw; bw = 0u; }
    if (HEAD) {
      // raw density on the rounded activations, on the matrix cores: the packed outputs of two neighbouring 16-column fragments ARE a
      // B operand (lane (row, kb): k-slots = columns (2a)*16 + kb*4 .. +3 and (2a+1)*16 + kb*4 .. +3) once the A operand carries w_d
      // in the same slot order (row n = 0 of 16; fp32 weights as three bf16 slices = 24 bits): D[n][row], n = 0 at the kb == 0 lanes
      mf_f32x4_t pr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        typedef unsigned __attribute__((ext_vector_type(4))) u4;
        const u4 bw4 = {uk[2 * a][0], uk[2 * a][1], uk[2 * a + 1][0], uk[2 * a + 1][1]};
        const mf_bf16x8_t bf = __builtin_bit_cast(mf_bf16x8_t, bw4);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
          pr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const mf_bf16x8_t*)(c2f_lds + ((a * 3 + sl) * 64 + lane) * 16), bf, pr, 0, 0, 0);
      }
      if (kb == 0) red[i * 16 + r16][wn] = pr[0];
    }
    if (i < 3) interleave();
  }
}

__global__ __launch_bounds__(256, 1) void k_mlp256_chain3_fwd(const MlpTail P) {
  __shared__ __attribute__((aligned(16))) unsigned char act[4][MC_ACT];      // X (DMA target), P, Q, R
  __shared__ __attribute__((aligned(16))) unsigned char c2f[4][6 * 64 * 16];   // per wave: w_d as A fragments [half a][slice][lane]
  __shared__ __attribute__((aligned(16))) float bs[3 * 256];
  __shared__ float red[MC_ROWS][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kb = lane >> 4;
  const int swz = 3 * ((r16 >> 2) & 1);
  const int frag_off = r16 * 64 + ((kb ^ swz) << 4);
  const int ntile = P.M / MC_ROWS, G = (int)gridDim.x;
#pragma unroll
  for (int l = 0; l < 3; ++l) bs[l * 256 + tid] = P.bias[l][tid];
  {   // w_d [256] fp32 as MFMA A operands in the slot order of mc_layer's B operands: lane (n = r16, kb), slot e of half a ->
      // column wn*64 + (2a + (e >> 2))*16 + kb*4 + (e & 3); rows n >= 1 are zero; each value as hi + lo + third bf16 slices
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      uint32_t hw[4], lw[4], tw[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int c0 = wn * 64 + (2 * a + (e2 >> 1)) * 16 + kb * 4 + (e2 & 1) * 2;
        const float v0 = (r16 == 0 && P.wd) ? P.wd[c0] : 0.f, v1 = (r16 == 0 && P.wd) ? P.wd[c0 + 1] : 0.f;
        const uint32_t h = mf_cvt_pk(v0, v1);
        const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
        const uint32_t l_ = mf_cvt_pk(r0, r1);
        hw[e2] = h; lw[e2] = l_;
        tw[e2] = mf_cvt_pk(r0 - __uint_as_float(l_ << 16), r1 - __uint_as_float(l_ & 0xffff0000u));
      }
      *(uint4*)(c2f[wn] + ((a * 3 + 0) * 64 + lane) * 16) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *(uint4*)(c2f[wn] + ((a * 3 + 1) * 64 + lane) * 16) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
      *(uint4*)(c2f[wn] + ((a * 3 + 2) * 64 + lane) * 16) = make_uint4(tw[0], tw[1], tw[2], tw[3]);
    }
  }
  // register-resident weights: stage s, fragment j of the [n][k] matrix of layer l for this lane
  mf_bf16x8_t w1[8][4], w2[8][4], w3[8][4];
  {
    const unsigned wo = (unsigned)((wn * 64 + r16) * 256 + kb * 8) * 2u;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w1[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wt[0] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
        w2[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wt[1] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
        w3[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wt[2] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
        // layers 1 and 2 in the accumulation-register half of the file (all 256 of it), layer 3 in architectural registers: left to
        // itself the allocator parks the ACCUMULATORS there and pays a v_accvgpr_read per value in every epilogue
        asm volatile("" : "+a"(w1[s][j]));
        asm volatile("" : "+a"(w2[s][j]));
        asm volatile("" : "+v"(w3[s][j]));
      }
  }
  // LDS-DMA of a tile's Y0 rows into act[0]: unit u = (stage s, 16-row block rb) = 1 KiB = one wave instruction; wave wn moves
  // units 8 wn .. 8 wn + 7.  Lane l lands at row l >> 2, physical chunk l & 3 and therefore FETCHES logical chunk (l & 3) ^ swz(row).
  const unsigned dma_voff = (unsigned)((lane >> 2) * 256 + (((lane & 3) ^ (3 * (((lane >> 2) >> 2) & 1))) << 3)) * 2u;
  const unsigned lds_x = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)act[0];
  auto issue_dma = [&](int t_) {
    const char* base = (const char*)P.Y0 + (size_t)t_ * MC_ROWS * 512;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = wn * 8 + q, s_ = u >> 2, rb = u & 3;
      const char* sb = base + (size_t)rb * 16 * 512 + s_ * 64;
      const unsigned la = lds_x + (unsigned)(s_ * MC_STAGE + rb * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sb), "v"(dma_voff), "s"(la) : "memory", "m0");
    }
  };
  if ((int)blockIdx.x < ntile) issue_dma((int)blockIdx.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int prev_m0 = -1;
  for (int t = blockIdx.x; t < ntile; t += G) {
    const int m0 = t * MC_ROWS;
    const bool has_next = t + G < ntile;
    // ---- layer 1: X -> P (the previous tile's Y3 leaves from R in this loop)
    mc_layer<false>(act[0], frag_off, w1, bs + wn * 64 + kb * 4, m0, wn, r16, kb, lane, act[1], P.bits[0], prev_m0 >= 0, act[3],
                    (char*)(P.Y[2] + (size_t)(prev_m0 < 0 ? 0 : prev_m0) * 256), tid, nullptr, red);
    __syncthreads();
    if (has_next) issue_dma(t + G);      // (X is free since the barrier; it lands under layers 2 and 3)
    // ---- layer 2: P -> Q (Y1 leaves from P)
    mc_layer<false>(act[1], frag_off, w2, bs + 256 + wn * 64 + kb * 4, m0, wn, r16, kb, lane, act[2], P.bits[1], true, act[1],
                    (char*)(P.Y[0] + (size_t)m0 * 256), tid, nullptr, red);
    __syncthreads();
    // ---- layer 3: Q -> R + the density head (Y2 leaves from Q)
    mc_layer<true>(act[2], frag_off, w3, bs + 512 + wn * 64 + kb * 4, m0, wn, r16, kb, lane, act[3], P.bits[2], true, act[2],
                   (char*)(P.Y[1] + (size_t)m0 * 256), tid, c2f[wn], red);
    // the next tile's rows: requested before 8 + 2 + 8 + 2 stores of this thread, all of which may stay in flight
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __syncthreads();
    if (P.wd && tid < MC_ROWS) {
      const float r = ((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3])) + P.bd[0];
      *(float*)((char*)(P.raw + m0) + mc_fresh((unsigned)tid * 4u)) = r;
      *(float*)((char*)(P.density + m0) + mc_fresh((unsigned)tid * 4u)) = mf_softplus(r + P.density_bias);
    }
    prev_m0 = m0;
  }
  if (prev_m0 >= 0) mc_copy_out(act[3], (char*)(P.Y[2] + (size_t)prev_m0 * 256), tid);
}


// ------------------------------------------------------------------------------------------------
// The same forward chain on EIGHT waves (two per SIMD): wave wq owns output columns [32 wq, 32 wq + 32) of all 64 rows, i.e. 64
// registers of every weight matrix (192 of its 256).  One wave per SIMD (the kernel above) leaves every LDS latency, barrier and the
// whole epilogue exposed -- ~20 k cycles per tile for 6 k cycles of MFMAs; with two waves per SIMD one wave's epilogue and waits sit
// under the other's MFMAs.  Costs: every wave reads the whole activation tile as its B operand (LDS fragment traffic 128 -> 256 KB per
// layer and tile, 1 k cycles at 256 B/clk: under the 2 k cycles of MFMAs per SIMD and layer), and two waves share each 32-bit word of
// the mask-bit layout (their halves are OR-ed through LDS behind the layer's barrier).  The density head is an fp32 dot product on the
// lane's 8 rounded outputs (no operand fragments in LDS: the 24 KiB they took do not fit next to four tiles here).
// ------------------------------------------------------------------------------------------------
#ifdef MC_TRACE      // (timing builds: s_memtime at the phase edges of workgroup 0's first tiles, waves 0 and 7; scratch/mc_trace.py)
__device__ long long mc_trace_buf[2 * 8 * 16];
#define MC_TP(k_) do { if (blockIdx.x == 0 && ti < 8 && (tid == 0 || tid == 448)) mc_trace_buf[((tid != 0) * 8 + ti) * 16 + (k_)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define MC_TP(k_) do { } while (0)
#endif
#ifndef MC_DMA_LAYER
#define MC_DMA_LAYER 2      // the layer whose loop carries the next tile's LDS-DMA requests (2: behind layer 2's copy-out; 3: behind layer 3's)
#endif
template <bool HEAD, class Mid>
__device__ __forceinline__ void mc8_layer(const unsigned char* A, int frag_off, const mf_bf16x8_t (&w)[8][2], const float* bias /* LDS, + wq*32 + kb*4 */,
                                          int wq, int r16, int kb, int lane, unsigned char* An, uint32_t* bpart /* LDS [2][64] of this wave */,
                                          bool do_cp, const unsigned char* cp_src, char* cp_dst, int tid, const float* wd_lds /* + wq*32 + kb*4 */,
                                          float (*red)[8][4], Mid&& mid /* runs behind the copy-out's stores (row block 1) */) {
  const int swz = 3 * ((r16 >> 2) & 1);
  // (ONE accumulator set: with two waves per SIMD the other wave's MFMAs cover this wave's epilogue; the double-buffered form of
  //  the four-wave kernel spilled 7 registers here, reloaded inside the tile loop)
  mf_f32x4_t acc[1][2];
  auto acc_init = [&](mf_f32x4_t (&a)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { const float4 bb = *(const float4*)(bias + j * 16); a[j] = mf_f32x4_t{bb.x, bb.y, bb.z, bb.w}; }
  };
  auto mma_half = [&](int i, int h, mf_f32x4_t (&a)[2]) {
    mf_bf16x8_t xa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xa[s] = *(const mf_bf16x8_t*)(A + (h * 4 + s) * MC_STAGE + frag_off + i * 16 * 64);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[h * 4 + s][j], xa[s], a[j], 0, 0, 0);
  };
  const int jj0 = (wq & 1) * 2;      // this wave's fragments are j = jj0, jj0 + 1 of its NT wave's 64-column block
  uint32_t bw = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc_init(acc[0]); mma_half(i, 0, acc[0]); mma_half(i, 1, acc[0]);
    if (i == 0 && do_cp) {
      // this thread's share of a finished tile: 2048 16-byte chunks over 512 threads (chunk id = q * 512 + tid: row q * 16 + (tid >> 5))
      typedef unsigned __attribute__((ext_vector_type(4))) u4;
      const int t_ = (int)mc_fresh((unsigned)tid), r0 = t_ >> 5, cc = t_ & 31;
      const unsigned lo = (unsigned)((cc >> 2) * MC_STAGE + r0 * 64 + (((cc & 3) ^ (3 * ((r0 >> 2) & 1))) << 4));
      const unsigned go = (unsigned)(r0 * 512 + cc * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u4 v = *(const u4*)(cp_src + lo + q * 1024);
        __builtin_nontemporal_store(v, (u4*)(cp_dst + (size_t)q * 8192 + go));
      }
    }
    if (i == 1) mid();
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const mf_f32x4_t v = acc[0][j];
      uint2 u;
      u.x = mc_relu_pk(mf_cvt_pk(v[0], v[1])); u.y = mc_relu_pk(mf_cvt_pk(v[2], v[3]));
      const int ch = j * 2 + (kb >> 1);
      *(uint2*)(An + wq * MC_STAGE + (i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
      const int k = (i & 1) * 8 + (jj0 + j) * 2;
      bw |= mc_nz_pk(u.x) << k;
      bw |= mc_nz_pk(u.y) << (k + 1);
      if (HEAD) {      // (w_d from LDS at use: eight more resident registers spilled)
        const float4 wv = *(const float4*)(wd_lds + j * 16);
        dsum = fmaf(__uint_as_float(u.x << 16), wv.x, dsum); dsum = fmaf(__uint_as_float(u.x & 0xffff0000u), wv.y, dsum);
        dsum = fmaf(__uint_as_float(u.y << 16), wv.z, dsum); dsum = fmaf(__uint_as_float(u.y & 0xffff0000u), wv.w, dsum);
      }
    }
    if (i & 1) { bpart[(i >> 1) * 64 + lane] = bw; bw = 0u; }
    if (HEAD) red[i * 16 + r16][wq][kb] = dsum;
    __builtin_amdgcn_sched_barrier(0);      // (nothing of the next row block -- its 8 fragment reads -- is hoisted above this epilogue)
  }
}

__global__ __launch_bounds__(512, 1) void k_mlp256_chain3_fwd8(const MlpTail P) {
  __shared__ __attribute__((aligned(16))) unsigned char act[4][MC_ACT];      // X (DMA target), P, Q, R
  __shared__ __attribute__((aligned(16))) float bs[3 * 256];
  __shared__ __attribute__((aligned(16))) float wds[256];
  __shared__ uint32_t bparts[2][8][2][64];                                   // [layer parity][wave][word][lane]: mask-bit halves
  __shared__ __attribute__((aligned(16))) float red[MC_ROWS][8][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kb = lane >> 4;
  const int swz = 3 * ((r16 >> 2) & 1);
  const int frag_off = r16 * 64 + ((kb ^ swz) << 4);
  const int ntile = P.M / MC_ROWS, G = (int)gridDim.x;
  for (int e = tid; e < 768; e += 512) bs[e] = P.bias[e >> 8][e & 255];
  if (tid < 256) wds[tid] = P.wd ? P.wd[tid] : 0.f;
  const float* wdv = wds + wq * 32 + kb * 4;
  mf_bf16x8_t w1[8][2], w2[8][2], w3[8][2];
  {
    const unsigned wo = (unsigned)((wq * 32 + r16) * 256 + kb * 8) * 2u;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        w1[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wt[0] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
        w2[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wt[1] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
        w3[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wt[2] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
      }
  }
  // Loads and stores of a CU go through one in-order vector-memory path: a store issued while the next tile's LDS-DMA requests are in
  // flight -- an 8 MB read burst when all CUs do it -- waits for them (phase trace, scratch/mc_trace.py: with the requests issued behind
  // the first barrier, layer 2 took 7.0 k cycles against layer 1's 3.1 k; giving loads and stores to different WAVES did not help:
  // 5.3 / 6.6 k, the path is shared).  So the requests are issued as late as the tile allows: in layer 3, behind its copy-out stores.
  // LDS-DMA of a tile's Y0 rows: 32 units of (stage, 16-row block), four per wave
  const unsigned dma_voff = (unsigned)((lane >> 2) * 256 + (((lane & 3) ^ (3 * (((lane >> 2) >> 2) & 1))) << 3)) * 2u;
  const unsigned lds_x = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)act[0];
  auto issue_dma = [&](int t_) {
    const char* base = (const char*)P.Y0 + (size_t)t_ * MC_ROWS * 512;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = wq * 4 + q, s_ = u >> 2, rb = u & 3;
      const char* sb = base + (size_t)rb * 16 * 512 + s_ * 64;
      const unsigned la = lds_x + (unsigned)(s_ * MC_STAGE + rb * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sb), "v"(dma_voff), "s"(la) : "memory", "m0");
    }
  };
  // the two waves of a pair each hold half of the pair's mask words: wave 2p stores word 0, wave 2p + 1 word 1 (ONE store per wave and layer)
  auto bits_out = [&](int par, uint32_t* bits, int m0) {
    const int p2 = wq & ~1, wd_ = wq & 1;
    const uint32_t v = bparts[par][p2][wd_][lane] | bparts[par][p2 + 1][wd_][lane];
    uint32_t* btile = bits + ((size_t)(m0 >> 8) * 8 + (size_t)(((m0 >> 7) & 1) * 4 + (wq >> 1))) * 256;
    const int i_nt0 = ((m0 >> 6) & 1) * 4;
    *(uint32_t*)((char*)btile + (mc_fresh((unsigned)lane * 4u) + (unsigned)(((i_nt0 >> 1) + wd_) * 64) * 4u)) = v;
  };
  if ((int)blockIdx.x < ntile) issue_dma((int)blockIdx.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int prev_m0 = -1, ti = -1;
  for (int t = blockIdx.x; t < ntile; t += G) {
    const int m0 = t * MC_ROWS;
    const bool has_next = t + G < ntile;
    ++ti;
    MC_TP(0);
    auto nothing = [] {};
    mc8_layer<false>(act[0], frag_off, w1, bs + wq * 32 + kb * 4, wq, r16, kb, lane, act[1], &bparts[0][wq][0][0], prev_m0 >= 0, act[3],
                     (char*)(P.Y[2] + (size_t)(prev_m0 < 0 ? 0 : prev_m0) * 256), tid, wdv, red, nothing);
    MC_TP(1);
    __syncthreads();
    MC_TP(2);
    bits_out(0, P.bits[0], m0);
    MC_TP(3);
    auto dma_next = [&] { if (has_next) issue_dma(t + G); };      // (X is free since the first barrier)
    mc8_layer<false>(act[1], frag_off, w2, bs + 256 + wq * 32 + kb * 4, wq, r16, kb, lane, act[2], &bparts[1][wq][0][0], true, act[1],
#if MC_DMA_LAYER == 2
                     (char*)(P.Y[0] + (size_t)m0 * 256), tid, wdv, red, dma_next);
#else
                     (char*)(P.Y[0] + (size_t)m0 * 256), tid, wdv, red, nothing);
#endif
    MC_TP(4);
    __syncthreads();
    MC_TP(5);
    bits_out(1, P.bits[1], m0);
    mc8_layer<true>(act[2], frag_off, w3, bs + 512 + wq * 32 + kb * 4, wq, r16, kb, lane, act[3], &bparts[0][wq][0][0], true, act[2],
#if MC_DMA_LAYER == 2
                    (char*)(P.Y[1] + (size_t)m0 * 256), tid, wdv, red, nothing);
    // the next tile's rows: requested before 1 + 4 stores of this thread (layer 2's mask word, layer 3's copy-out), which may stay in flight
    MC_TP(6);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
#else
                    (char*)(P.Y[1] + (size_t)m0 * 256), tid, wdv, red, dma_next);
    // the next tile's rows: requested behind every store of the tile
    MC_TP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MC_TP(7);
    __syncthreads();
    MC_TP(8);
    bits_out(0, P.bits[2], m0);
    if (P.wd && wq < 4) {
      // raw density of the tile's 64 rows: four threads per row, each sums two waves' four partial sums, a quad reduction (DPP) adds them
      // (as ONE wave's job -- 32 scalar LDS reads, exp, log per lane -- it made that wave 1.4 k cycles late for the next tile)
      const int row = tid >> 2, p4 = tid & 3;
      const float4 a = *(const float4*)&red[row][p4][0], b = *(const float4*)&red[row][p4 + 4][0];
      float r = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
      r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
      r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
      if (p4 == 0) {
        r += P.bd[0];
        *(float*)((char*)(P.raw + m0) + mc_fresh((unsigned)row * 4u)) = r;
        *(float*)((char*)(P.density + m0) + mc_fresh((unsigned)row * 4u)) = mf_softplus(r + P.density_bias);
      }
    }
    MC_TP(9);
    prev_m0 = m0;
  }
  if (prev_m0 >= 0) {
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    const int r0 = tid >> 5, cc = tid & 31;
    const unsigned lo = (unsigned)((cc >> 2) * MC_STAGE + r0 * 64 + (((cc & 3) ^ (3 * ((r0 >> 2) & 1))) << 4));
    char* dst = (char*)(P.Y[2] + (size_t)prev_m0 * 256);
#pragma unroll
    for (int q = 0; q < 4; ++q) *(u4*)(dst + (size_t)q * 8192 + (unsigned)(r0 * 512 + cc * 16)) = *(const u4*)(act[3] + lo + q * 1024);
  }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the backward twin -- the dX chain of the same three layers from the density head's gradient down to the gradient at
// layer 0's pre-activation (what jax.value_and_grad derives for models.py:451-456,467 with disable_rgb), one launch:
//   G3 = (d_raw (x) w_d) * (Y3 > 0)          G2 = (G3 W3^T) * (Y2 > 0)       G1 = (G2 W2^T) * (Y1 > 0)       G0 = (G1 W1^T) * (Y0 > 0)
// Every G_l [M, 256] is written exactly once (the weight-gradient GEMMs read them: dW_l = Y_{l-1}^T G_l, db_l = colsum G_l); the
// masks are the forward pass's 1-bit masks in hugs_gemm_nt_bits' lane layout.  The launches it replaces: hugs_rank1_mask (reads
// Y3, writes G3) + three masked dX GEMMs (each reads G_l and writes G_{l-1}): 3.5 GB per 1 M rows against 2.1 GB here.
// Same geometry as k_mlp256_chain3_fwd: the three Wn operand copies ([fan_in][fan_out] = [output of dX][reduction]) resident in
// registers, tiles B0 -> B1 -> B2 -> B3 in LDS, finished tiles leave from inside the next phase.  A tile's inputs -- 8 mask words per
// lane and 64 d_raw values -- arrive by LDS-DMA a tile ahead (double-buffered, 9 instructions per wave), waited for with a counted
// vmcnt (24 stores behind them stay in flight).
// ------------------------------------------------------------------------------------------------
struct MlpTailBwd {
  int M;
  const float* d_raw;              // [M]
  const float* wd;                 // [256]
  const uint16_t* Wn[3];           // layers 1..3: [256 (fan_in)][256 (fan_out)] bf16
  const uint32_t* bits[4];         // masks of Y0 .. Y3
  uint16_t* G[4];                  // outputs G0 .. G3 [M, 256]
};

__device__ __forceinline__ uint32_t mc_pk_mul(uint32_t a, uint32_t b) {      // v_pk_mul_lo_u16: each half times its 0 / 1
  typedef unsigned short __attribute__((ext_vector_type(2))) us2;
  const us2 r = __builtin_bit_cast(us2, a) * __builtin_bit_cast(us2, b);
  return __builtin_bit_cast(uint32_t, r);
}

// G_out tile = (G_in tile x W^T) * mask for this wave's 64 output columns; mw: the lane's two mask words of the tile (LDS)
__device__ __forceinline__ void mc_layer_bwd(const unsigned char* A, int frag_off, const mf_bf16x8_t (&w)[8][4], int wn, int r16, int kb,
                                             unsigned char* An, const uint32_t* mw, const unsigned char* cp_src, char* cp_dst, int tid) {
  const int swz = 3 * ((r16 >> 2) & 1);
  mf_f32x4_t acc[2][4];
  auto acc_zero = [&](mf_f32x4_t (&a)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = mf_f32x4_t{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(a[j])); }
  };
  auto mma_half = [&](int i, int h, mf_f32x4_t (&a)[4]) {
    mf_bf16x8_t xa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xa[s] = *(const mf_bf16x8_t*)(A + (h * 4 + s) * MC_STAGE + frag_off + i * 16 * 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[h * 4 + s][j], xa[s], a[j], 0, 0, 0);
  };
  const uint32_t m01[2] = {mw[0], mw[64]};      // words (i >> 1) = 0, 1 of this tile's 64 rows
  auto epi_half = [&](int i, int jh) {
#pragma unroll
    for (int j = 2 * jh; j < 2 * jh + 2; ++j) {
      const mf_f32x4_t v = acc[i & 1][j];
      const int k = (i & 1) * 8 + j * 2;
      const uint32_t t = m01[i >> 1] >> k;
      uint2 u;
      u.x = mc_pk_mul(mf_cvt_pk(v[0], v[1]), t & 0x00010001u);
      u.y = mc_pk_mul(mf_cvt_pk(v[2], v[3]), (t >> 1) & 0x00010001u);
      const int st = wn * 2 + (j >> 1), ch = (j & 1) * 2 + (kb >> 1);
      *(uint2*)(An + st * MC_STAGE + (i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
    }
  };
  auto interleave = [&]() {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };
  acc_zero(acc[0]);
  mma_half(0, 0, acc[0]);
  mma_half(0, 1, acc[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < 3) { acc_zero(acc[(i + 1) & 1]); mma_half(i + 1, 0, acc[(i + 1) & 1]); }
    if (i == 0) mc_copy_out(cp_src, cp_dst, tid);
    epi_half(i, 0);
    if (i < 3) interleave();
    if (i < 3) mma_half(i + 1, 1, acc[(i + 1) & 1]);
    epi_half(i, 1);
    if (i < 3) interleave();
  }
}

__global__ __launch_bounds__(256, 1) void k_mlp256_chain3_bwd(const MlpTailBwd P) {
  __shared__ __attribute__((aligned(16))) unsigned char act[4][MC_ACT];      // B0 (G3), B1 (G2), B2 (G1), B3 (G0)
  __shared__ __attribute__((aligned(16))) uint32_t mws[2][4][4][128];        // [buffer][wave][mask of Y_l][word 0..1][lane]
  __shared__ __attribute__((aligned(16))) float drs[2][4][64];               // [buffer][wave (its own copy)][row]
  __shared__ __attribute__((aligned(16))) float wds[256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kb = lane >> 4;
  const int swz = 3 * ((r16 >> 2) & 1);
  const int frag_off = r16 * 64 + ((kb ^ swz) << 4);
  const int ntile = P.M / MC_ROWS, G = (int)gridDim.x;
  wds[tid] = P.wd[tid];
  mf_bf16x8_t w1[8][4], w2[8][4], w3[8][4];
  {
    const unsigned wo = (unsigned)((wn * 64 + r16) * 256 + kb * 8) * 2u;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w1[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wn[0] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
        w2[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wn[1] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
        w3[s][j] = *(const mf_bf16x8_t*)((const char*)P.Wn[2] + (wo + (unsigned)(j * 16 * 256 + s * 32) * 2u));
        asm volatile("" : "+a"(w1[s][j]));
        asm volatile("" : "+a"(w2[s][j]));
        asm volatile("" : "+v"(w3[s][j]));
      }
  }
  // a tile's inputs by LDS-DMA (4 bytes per lane): the wave's two words of each of the four masks + the tile's 64 d_raw values
  const unsigned voff4 = (unsigned)lane * 4u;
  auto issue_dma = [&](int t_, int buf) {
    const int m0 = t_ * MC_ROWS;
    const size_t wbase = (((size_t)(m0 >> 8) * 8 + (size_t)(((m0 >> 7) & 1) * 4 + wn)) * 256 + (size_t)(((m0 >> 6) & 1) * 2) * 64) * 4;
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int wd_ = 0; wd_ < 2; ++wd_) {
        const char* sb = (const char*)P.bits[l] + wbase + wd_ * 256;
        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)&mws[buf][wn][l][wd_ * 64];
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sb), "v"(voff4), "s"(la) : "memory", "m0");
      }
    {
      const char* sb = (const char*)(P.d_raw + m0);
      const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&drs[buf][wn][0];
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sb), "v"(voff4), "s"(la) : "memory", "m0");
    }
  };
  if ((int)blockIdx.x < ntile) issue_dma((int)blockIdx.x, 0);
  int prev_m0 = -1, buf = 0;
  for (int t = blockIdx.x; t < ntile; t += G, buf ^= 1) {
    const int m0 = t * MC_ROWS;
    // this tile's inputs: requested a tile ago, in front of 24 stores of this thread (first tile: in front of nothing -- the same
    // count only waits for more)
    if (prev_m0 >= 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();      // (also: B3 of the previous tile is complete, wds / the weights' prologue is done)
    if (prev_m0 >= 0) mc_copy_out(act[3], (char*)(P.G[0] + (size_t)prev_m0 * 256), tid);
    if (t + G < ntile) issue_dma(t + G, buf ^ 1);
    // ---- G3 = (d_raw (x) w_d) * mask3 -> B0
    {
      const uint32_t* mw = &mws[buf][wn][3][lane];
      const uint32_t m01[2] = {mw[0], mw[64]};
      float4 wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = *(const float4*)(wds + wn * 64 + j * 16 + kb * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = drs[buf][wn][i * 16 + r16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = (i & 1) * 8 + j * 2;
          const uint32_t t_ = m01[i >> 1] >> k;
          uint2 u;
          u.x = mc_pk_mul(mf_cvt_pk(r * wv[j].x, r * wv[j].y), t_ & 0x00010001u);
          u.y = mc_pk_mul(mf_cvt_pk(r * wv[j].z, r * wv[j].w), (t_ >> 1) & 0x00010001u);
          const int st = wn * 2 + (j >> 1), ch = (j & 1) * 2 + (kb >> 1);
          *(uint2*)(act[0] + st * MC_STAGE + (i * 16 + r16) * 64 + ((ch ^ swz) << 4) + (kb & 1) * 8) = u;
        }
      }
    }
    __syncthreads();
    mc_layer_bwd(act[0], frag_off, w3, wn, r16, kb, act[1], &mws[buf][wn][2][lane], act[0], (char*)(P.G[3] + (size_t)m0 * 256), tid);
    __syncthreads();
    mc_layer_bwd(act[1], frag_off, w2, wn, r16, kb, act[2], &mws[buf][wn][1][lane], act[1], (char*)(P.G[2] + (size_t)m0 * 256), tid);
    __syncthreads();
    mc_layer_bwd(act[2], frag_off, w1, wn, r16, kb, act[3], &mws[buf][wn][0][lane], act[2], (char*)(P.G[1] + (size_t)m0 * 256), tid);
    prev_m0 = m0;
  }
  __syncthreads();
  if (prev_m0 >= 0) mc_copy_out(act[3], (char*)(P.G[0] + (size_t)prev_m0 * 256), tid);
}

}  // namespace

extern "C" int hugs_mlp256_tail_max_layers(void) { return MF_MAXL; }

// include/hugs.h hugs_mlp256_tail_fwd
extern "C" int hugs_mlp256_tail_fwd(int dtype, int M, int nl, const void* Y0, const void* const* Wt, const float* const* bias,
                                    void* const* Y, uint32_t* const* bits, const float* wd, const float* bd, float density_bias,
                                    float* raw, float* density, void* stream) {
  HUGS_REQUIRE(dtype == 1, -2, "hugs_mlp256_tail_fwd: bf16 operands only (dtype 1), got %d", dtype);
  HUGS_REQUIRE(M > 0 && M % 256 == 0 && nl >= 1 && nl <= MF_MAXL && Y0 && Wt && bias && Y, -3,
               "hugs_mlp256_tail_fwd: M=%d (a multiple of 256), %d layers (1..%d)", M, nl, MF_MAXL);
  HUGS_REQUIRE(!wd || (bd && raw && density), -2, "hugs_mlp256_tail_fwd: the density head needs bd, raw and density");
  MlpTail P;
  P.M = M; P.nl = nl; P.Y0 = (const uint16_t*)Y0;
  for (int l = 0; l < MF_MAXL; ++l) {
    P.Wt[l] = l < nl ? (const uint16_t*)Wt[l] : nullptr;
    P.bias[l] = l < nl ? bias[l] : nullptr;
    P.Y[l] = l < nl ? (uint16_t*)Y[l] : nullptr;
    P.bits[l] = (l < nl && bits) ? bits[l] : nullptr;
    HUGS_REQUIRE(l >= nl || (P.Wt[l] && P.bias[l] && P.Y[l]), -2, "hugs_mlp256_tail_fwd: null pointer in layer %d", l);
  }
  P.wd = wd; P.bd = bd; P.density_bias = density_bias; P.raw = raw; P.density = density;
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
  // Round 5: exactly three layers with their mask bits (the reference's PropMLP) -> the register-resident kernel, one workgroup
  // per CU (HUGS_MLPFUSE_CHAIN3=0: the L2-streaming kernel below, the round-4 form)
  static const bool chain3 = []() { const char* e = getenv("HUGS_MLPFUSE_CHAIN3"); return !(e && e[0] == '0'); }();
  if (chain3 && nl == 3 && P.bits[0] && P.bits[1] && P.bits[2]) {
    const int ntile = M / 64;
    // HUGS_MLPFUSE_WAVES=4: the one-wave-per-SIMD form; default 8 (two waves per SIMD)
    static const bool waves8 = []() { const char* e = getenv("HUGS_MLPFUSE_WAVES"); return !(e && e[0] == '4'); }();
    if (waves8) hipLaunchKernelGGL(k_mlp256_chain3_fwd8, dim3(ntile < ncu ? ntile : ncu), dim3(512), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(k_mlp256_chain3_fwd, dim3(ntile < ncu ? ntile : ncu), dim3(256), 0, (hipStream_t)stream, P);
    HUGS_CHECK_LAUNCH("hugs_mlp256_tail_fwd(chain3)");
    return 0;
  }
  static const int wm_sel = []() { const char* e = getenv("HUGS_MLPFUSE_WM"); return e && e[0] == '2' ? 2 : 1; }();
  if (wm_sel == 2) {
    const int ntile = M / 128;
    hipLaunchKernelGGL(k_mlp256_tail_fwd<2>, dim3(ntile < ncu ? ntile : ncu), dim3(512), 0, (hipStream_t)stream, P);
  } else {
    const int ntile = M / 64;
    hipLaunchKernelGGL(k_mlp256_tail_fwd<1>, dim3(ntile < 2 * ncu ? ntile : 2 * ncu), dim3(256), 0, (hipStream_t)stream, P);
  }
  HUGS_CHECK_LAUNCH("hugs_mlp256_tail_fwd");
  return 0;
}

// include/hugs.h hugs_mlp256_tail_bwd
extern "C" int hugs_mlp256_tail_bwd(int dtype, int M, int nl, const float* d_raw, const float* wd, const void* const* Wn,
                                    const uint32_t* const* bits, void* const* G, void* stream) {
  HUGS_REQUIRE(dtype == 1, -2, "hugs_mlp256_tail_bwd: bf16 operands only (dtype 1), got %d", dtype);
  HUGS_REQUIRE(M > 0 && M % 256 == 0 && nl == 3 && d_raw && wd && Wn && bits && G, -3,
               "hugs_mlp256_tail_bwd: M=%d (a multiple of 256), %d layers (exactly 3: callers fall back to per-layer GEMMs otherwise)", M, nl);
  MlpTailBwd P;
  P.M = M; P.d_raw = d_raw; P.wd = wd;
  for (int l = 0; l < 3; ++l) { P.Wn[l] = (const uint16_t*)Wn[l]; HUGS_REQUIRE(P.Wn[l], -2, "hugs_mlp256_tail_bwd: null weight pointer %d", l); }
  for (int l = 0; l < 4; ++l) {
    P.bits[l] = bits[l]; P.G[l] = (uint16_t*)G[l];
    HUGS_REQUIRE(P.bits[l] && P.G[l], -2, "hugs_mlp256_tail_bwd: null mask / output pointer %d", l);
  }
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
  const int ntile = M / 64;
  hipLaunchKernelGGL(k_mlp256_chain3_bwd, dim3(ntile < ncu ? ntile : ncu), dim3(256), 0, (hipStream_t)stream, P);
  HUGS_CHECK_LAUNCH("hugs_mlp256_tail_bwd");
  return 0;
}

#ifdef MC_TRACE
extern "C" int hugs_mc_trace_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mc_trace_buf), sizeof(long long) * 256); }
#endif

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

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
struct GemmEpi { const float* bias; const float* row_bias; int row_div, ld_rb; int relu; const void* mask; int ld_mask; const float* r1_row; const float* r1_col; void* out; int ldc; };
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0); }
__device__ __forceinline__ int xcd_remap(int bid, int nwg) { const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3; return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k; }
#ifndef HUGS_NT_DIRECT_EPI
#define HUGS_NT_DIRECT_EPI 1   // A/B-tested: +10 % over staging the C tile through LDS
#endif
typedef unsigned __attribute__((ext_vector_type(2))) u32x2_t;
#ifndef HUGS_NT_VARIANT
#define HUGS_NT_VARIANT 2   // 8 MFMA : 3 ds_read interleave of next-stage fragment reads (A/B-tested: +5 %)
#endif
#define GL_CPAD 16   // bytes added to each row of the staged C tile (bank spread for the 8-byte fragment writes)

// WN = wave columns: 4 -> 256x256 tile, 8 waves, 4-slot ring (128 KiB, 1 workgroup/CU);
//                    2 -> 256x128 tile, 4 waves, 3-slot ring (72 KiB, 2 workgroups/CU, their epilogues and
//                         prologues overlap each other's main loops).
template <int WN, int MODE>
__global__ __launch_bounds__(128 * WN, WN == 4 ? 2 : 1) void k_gemm_nt_bf16_big(
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
  auto stage = [&](int st) {
    const int kglob = st << 5;
    const uint16_t* Abase; int lda, kcol;
    if (kglob < K1) { Abase = A1; lda = lda1; kcol = kglob; } else { Abase = A2; lda = lda2; kcol = kglob - K1; }
    unsigned char* la = lds + (st % NSLOT) * STAGE;
    unsigned char* lb = la + A_BYTES;
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int p = it * NT + tid, row = p >> 2, col = ((p & 3) ^ (3 * ((row >> 2) & 1))) * 8;
      glds16(Abase + (size_t)(m0 + row) * lda + kcol + col, la + (it * NT + wv * 64) * 16);
    }
#pragma unroll
    for (int it = 0; it < BIT; ++it) {
      const int p = it * NT + tid, row = p >> 2, col = ((p & 3) ^ (3 * ((row >> 2) & 1))) * 8;
      glds16(Bt + (size_t)(n0 + row) * ldb + kglob + col, lb + (it * NT + wv * 64) * 16);
    }
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
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wb[j], f.xa[i], acc[i][j], 0, 0, 0);
  };
  constexpr int G = AIT + BIT;   // LDS-DMA instructions per thread per stage
  // iteration st: frags(st) are in `cur`; make stage st+1 visible, refill slot st%NSLOT with stage st+NSLOT,
  // start reading frags(st+1) into `nxt`, then run the MFMAs of stage st.
#define GL_ITER(cur, nxt, st, VM)                                                        \
  {                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");                     \
    if (!(MODE & 2)) __builtin_amdgcn_s_barrier();                                                         \
    asm volatile("" ::: "memory");                                                        \
    if (!(MODE & 1) && (st) + NSLOT < ns) stage((st) + NSLOT);                                           \
    if (!(MODE & 4)) load_frags(nxt, (st) + 1); else { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) asm volatile("" : "+v"(nxt.xa[q_])); }                                                            \
    mfmas(cur);                                                                           \
    if (HUGS_NT_VARIANT == 2) {                                                           \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                \
      }                                                                                   \
    }                                                                                     \
    if (HUGS_NT_VARIANT == 4) {                                                           \
      _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                \
      }                                                                                   \
    }                                                                                     \
    if (HUGS_NT_VARIANT == 5) {                                                           \
      _Pragma("unroll") for (int q_ = 0; q_ < 12; ++q_) {                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                \
      }                                                                                   \
    }                                                                                     \
    if (HUGS_NT_VARIANT == 6) {                                                           \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                \
      }                                                                                   \
    }                                                                                     \
    if (HUGS_NT_VARIANT == 7) {                                                           \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                \
      }                                                                                   \
    }                                                                                     \
  }
  Frags f0, f1;
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) stage(q);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
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
  if (MODE & 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
#if HUGS_NT_DIRECT_EPI
  if (WN == 4) {
    // ---- epilogue straight from registers: bias / rank-1 / relu, v_cvt_pk_bf16_f32, one v_permlane16_swap pair
    // per two neighbouring 16-column fragments turns the 8-byte-per-lane MFMA layout into 16 contiguous bytes per
    // lane (64-byte runs per row): no LDS round trip, no barriers.
    float4 bj[4], cj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + kb * 4;
      bj[j] = E.bias ? *(const float4*)(E.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      cj[j] = E.r1_row ? *(const float4*)(E.r1_col + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ccol = n0 + wn * 64 + (kb & 1) * 16 + (kb >> 1) * 8;   // + jp*32: this lane's 8 output columns
    uint4 mkv[8][2];
    if (E.mask) {   // all 16 mask chunks in flight before the conversion work starts
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
          mkv[i][jp] = *(const uint4*)((const uint16_t*)E.mask + (size_t)(m0 + wm * 128 + i * 16 + r16) * E.ld_mask + ccol + jp * 32);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * 128 + i * 16 + r16;
      const float r1 = E.r1_row ? E.r1_row[m] : 0.f;
      const float* rbp = E.row_bias ? E.row_bias + (size_t)(m / E.row_div) * E.ld_rb + n0 + wn * 64 + kb * 4 : nullptr;
      uint32_t pk[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[4] = {acc[i][j][0] + bj[j].x + r1 * cj[j].x, acc[i][j][1] + bj[j].y + r1 * cj[j].y,
                      acc[i][j][2] + bj[j].z + r1 * cj[j].z, acc[i][j][3] + bj[j].w + r1 * cj[j].w};
        if (rbp) { const float4 b = *(const float4*)(rbp + j * 16); x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w; }
        if (E.relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
        bf16x4_t h;
        h[0] = (__bf16)x[0]; h[1] = (__bf16)x[1]; h[2] = (__bf16)x[2]; h[3] = (__bf16)x[3];
        const uint2 u = *(const uint2*)&h;
        pk[j][0] = u.x; pk[j][1] = u.y;
      }
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(pk[2 * jp][0], pk[2 * jp + 1][0], false, false);
        const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(pk[2 * jp][1], pk[2 * jp + 1][1], false, false);
        uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        if (E.mask) {
          const uint4 mk = mkv[i][jp];
          const uint32_t mw[4] = {mk.x, mk.y, mk.z, mk.w};
          uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t lo = mw[q] & 0xffffu, hi = mw[q] >> 16;
            const uint32_t keep = (((lo & 0x7fffu) && !(lo & 0x8000u)) ? 0x0000ffffu : 0u) |
                                  (((hi & 0x7fffu) && !(hi & 0x8000u)) ? 0xffff0000u : 0u);
            vw[q] &= keep;
          }
          v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
        }
        *(uint4*)((uint16_t*)E.out + (size_t)m * E.ldc + ccol + jp * 32) = v;
      }
    }
    return;
  }
#endif
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
      pk[0] = (__bf16)x[0]; pk[1] = (__bf16)x[1]; pk[2] = (__bf16)x[2]; pk[3] = (__bf16)x[3];
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


template <int MODE> float run(int M, int N, int K, uint16_t* A, uint16_t* B, float* bias, uint16_t* C) {
  GemmEpi E{bias, nullptr, 1, 0, 1, nullptr, 0, nullptr, nullptr, C, N};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g((M / 256) * (N / 256)), b(512);
  for (int i = 0; i < 5; ++i) k_gemm_nt_bf16_big<4, MODE><<<g, b>>>(M, N, K, 0, A, K, nullptr, 0, B, K, E);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) k_gemm_nt_bf16_big<4, MODE><<<g, b>>>(M, N, K, 0, A, K, nullptr, 0, B, K, E);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main() {
  const int M = 131072, N = 1024, K = 1024;
  std::vector<uint16_t> h((size_t)M * K);
  for (auto& x : h) x = (uint16_t)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15));
  uint16_t *A, *B, *C; float* bias;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
  hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, N * 4);
  const double fl = 2.0 * M * N * K; float t;
#define R(MODE, name) t = run<MODE>(M, N, K, A, B, bias, C); printf("%-44s %.3f ms %.0f TF\n", name, t, fl / t / 1e9);
  R(0, "warm") R(0, "full") R(8, "no-epi") R(9, "no-epi no-glds") R(11, "no-epi no-glds no-barrier") R(15, "no-epi no-glds no-barrier no-dsread (MFMA only)") R(13, "no-epi no-glds no-dsread (MFMA+barrier)") R(0, "full")
  return 0;
}

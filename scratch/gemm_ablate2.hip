#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
struct GemmEpi { const float* bias; const float* row_bias; int row_div, ld_rb; int relu; const void* mask; int ld_mask; const float* r1_row; const float* r1_col; void* out; int ldc; };
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0); }
__device__ __forceinline__ int xcd_remap(int bid, int nwg) { const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3; return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k; }
__device__ __forceinline__ uint16_t f_to_bf16(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
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

  auto compute = [&](int st) {
    const unsigned char* la = lds + (st % NSLOT) * STAGE + (wm * 128) * 64 + frag_off;
    const unsigned char* lb = lds + (st % NSLOT) * STAGE + A_BYTES + (wn * 64) * 64 + frag_off;
    bf16x8_t wb[4], xa[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) wb[j] = *(const bf16x8_t*)(lb + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) xa[i] = *(const bf16x8_t*)(la + i * 16 * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { if (MODE == 3) { asm volatile("" ::"v"(xa[i]), "v"(wb[j])); } else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0); }
  };

  // NSLOT-1 stages in flight; every loop iteration finds exactly that many outstanding at its wait.
#pragma unroll
  for (int st = 0; st < NSLOT - 1; ++st) stage(st);
  for (int st = 0; st < ns - (NSLOT - 2); ++st) {
    if (WN == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (NSLOT-2) * (AIT+BIT) loads may still fly
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // stage st landed for every wave; ring slot (st-1)%NSLOT is free
    asm volatile("" ::: "memory");
    if (MODE != 2 && st + NSLOT - 1 < ns) stage(st + NSLOT - 1);
    compute(st);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q = NSLOT - 2; q >= 1; --q) compute(ns - q);
  __syncthreads();   // everyone is done reading the ring: reuse it as the C staging tile

  if (MODE == 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // ---- epilogue
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
      uint2 pk;
      pk.x = f_to_bf16(x[0]) | ((uint32_t)f_to_bf16(x[1]) << 16);
      pk.y = f_to_bf16(x[2]) | ((uint32_t)f_to_bf16(x[3]) << 16);
      *(uint2*)(lds + ml * CPITCH + nl * 2) = pk;
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
    if (MODE == 1) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); } else *(uint4*)((uint16_t*)E.out + (size_t)(m0 + row) * E.ldc + n0 + c * 8) = v;
  }
}


template <int WN, int MODE> float run(int M, int N, int K, uint16_t* A, uint16_t* B, float* bias, uint16_t* C) {
  GemmEpi E{bias, nullptr, 1, 0, 1, nullptr, 0, nullptr, nullptr, C, N};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g((M / 256) * (N / (64 * WN))), b(128 * WN);
  for (int i = 0; i < 3; ++i) k_gemm_nt_bf16_big<WN, MODE><<<g, b>>>(M, N, K, 0, A, K, nullptr, 0, B, K, E);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) k_gemm_nt_bf16_big<WN, MODE><<<g, b>>>(M, N, K, 0, A, K, nullptr, 0, B, K, E);
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
#define R(WN, MODE, name) t = run<WN, MODE>(M, N, K, A, B, bias, C); printf("WN=%d %-12s %.3f ms %.0f TF\n", WN, name, t, fl / t / 1e9);
  R(2, 0, "full") R(2, 1, "no-gstore") R(2, 4, "no-epilogue") R(2, 2, "no-glds") R(2, 3, "no-mfma")
  R(4, 0, "full") R(4, 1, "no-gstore") R(4, 4, "no-epilogue") R(4, 2, "no-glds") R(4, 3, "no-mfma")
  return 0;
}

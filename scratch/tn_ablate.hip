#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0); }
__device__ __forceinline__ int xcd_remap(int bid, int nwg) { const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3; return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k; }
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_gemm_tn_bf16_big(int Mrows, int Kc, int N, int nsplit,
                                                              const uint16_t* __restrict__ X, int ldx,
                                                              const uint16_t* __restrict__ G, int ldg,
                                                              float* __restrict__ slab, int lds_out,
                                                              float* __restrict__ colsum_slab) {
  constexpr int NSLOT = 4, STAGE = 32768, XB = 16384;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N >> 8, ntk = Kc >> 8, tiles = ntn * ntk;
  // all tiles of one M-split read the same X / G row block: keep them on one XCD so its L2 serves the re-reads
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vb / tiles, tt = vb % tiles;
  const int c0 = (tt / ntn) << 8, n0 = (tt % ntn) << 8;
  const int rows_per = Mrows / nsplit, mbeg = split * rows_per;
  const int ns = rows_per >> 5;
  const int wn = wv >> 2, wk = wv & 3;
  const bool do_colsum = colsum_slab && c0 == 0;

  auto stage = [&](int st) {
    const int mrow0 = mbeg + (st << 5);
    unsigned char* lx = lds + (st % NSLOT) * STAGE;
    unsigned char* lg = lx + XB;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int p = it * 512 + tid, row = p >> 5, pos = p & 31;
      const int c = ((((pos >> 1) ^ (row & 7)) << 1) | (pos & 1)) * 8;
      glds16(X + (size_t)(mrow0 + row) * ldx + c0 + c, lx + (it * 512 + wv * 64) * 16);
      glds16(G + (size_t)(mrow0 + row) * ldg + n0 + c, lg + (it * 512 + wv * 64) * 16);
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
  for (int q = 0; q < 8; ++q) ones[q] = (__bf16)1.0f;

  const int g = lane >> 4, s = lane & 15;
  struct Frags { bf16x8_t ga[8], xb[4]; };
  auto load_frags = [&](Frags& f, int st) {
    const unsigned char* lx = lds + (st % NSLOT) * STAGE;
    const unsigned char* lg = lx + XB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = h * 16 + g * 4 + (s >> 2);
      const int sw = row & 7;
      const int lo = row * 512 + ((s & 3) << 3);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = wn * 8 + i;
        bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4_t __attribute__((address_space(3)))*)(lg + lo + ((q ^ sw) << 5)));
        f.ga[i][h * 4 + 0] = v[0]; f.ga[i][h * 4 + 1] = v[1]; f.ga[i][h * 4 + 2] = v[2]; f.ga[i][h * 4 + 3] = v[3];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = wk * 4 + j;
        bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4_t __attribute__((address_space(3)))*)(lx + lo + ((q ^ sw) << 5)));
        f.xb[j][h * 4 + 0] = v[0]; f.xb[j][h * 4 + 1] = v[1]; f.xb[j][h * 4 + 2] = v[2]; f.xb[j][h * 4 + 3] = v[3];
      }
    }
  };
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { if (MODE & 1) { asm volatile("" ::"v"(f.ga[i]), "v"(f.xb[j])); } else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[i], f.xb[j], acc[i][j], 0, 0, 0); }
    if (do_colsum && !(MODE & 1)) {   // wave (wn, wk) owns the column sums of its N fragments 2wk, 2wk+1
      if (wk == 0) { accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[0], ones, accb[0], 0, 0, 0); accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[1], ones, accb[1], 0, 0, 0); }
      else if (wk == 1) { accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[2], ones, accb[0], 0, 0, 0); accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[3], ones, accb[1], 0, 0, 0); }
      else if (wk == 2) { accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[4], ones, accb[0], 0, 0, 0); accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[5], ones, accb[1], 0, 0, 0); }
      else { accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[6], ones, accb[0], 0, 0, 0); accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ga[7], ones, accb[1], 0, 0, 0); }
    }
  };
#define GT_ITER(cur, nxt, st, VM)                                         \
  {                                                                       \
    asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                         \
    asm volatile("" ::: "memory");                                        \
    if (!(MODE & 2) && (st) + NSLOT < ns) stage((st) + NSLOT);                           \
    load_frags(nxt, (st) + 1);                                            \
    mfmas(cur);                                                           \
  }
  Frags f0, f1;
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) stage(q);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_frags(f0, 0);
  int st = 0;
  for (; st + 5 < ns; st += 2) { GT_ITER(f0, f1, st, 8) GT_ITER(f1, f0, st + 1, 8) }
  GT_ITER(f0, f1, st, 8)
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
  load_frags(f0, st + 2); mfmas(f1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
  load_frags(f1, st + 3); mfmas(f0);
  mfmas(f1);
#undef GT_ITER

  if (MODE & 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  float* out = slab + (size_t)split * Kc * lds_out;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + wk * 64 + j * 16 + s;
      const int n = n0 + wn * 128 + i * 16 + g * 4;
      *(float4*)(out + (size_t)c * lds_out + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  if (do_colsum && s == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = n0 + wn * 128 + (2 * wk + q) * 16 + g * 4;
      *(float4*)(colsum_slab + (size_t)split * N + n) = make_float4(accb[q][0], accb[q][1], accb[q][2], accb[q][3]);
    }
  }
}


template <int MODE> float run(int M, int Kc, int N, int ns, uint16_t* X, uint16_t* G, float* slab, float* cs) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g((Kc / 256) * (N / 256) * ns), b(512);
  for (int i = 0; i < 3; ++i) k_gemm_tn_bf16_big<MODE><<<g, b>>>(M, Kc, N, ns, X, Kc, G, N, slab, N, cs);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) k_gemm_tn_bf16_big<MODE><<<g, b>>>(M, Kc, N, ns, X, Kc, G, N, slab, N, cs);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main() {
  const int M = 131072, N = 1024, K = 1024, ns = 16;
  std::vector<uint16_t> h((size_t)M * K);
  for (auto& x : h) x = (uint16_t)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15));
  uint16_t *X, *G; float *slab, *cs;
  hipMalloc(&X, (size_t)M * K * 2); hipMalloc(&G, (size_t)M * N * 2); hipMalloc(&slab, (size_t)ns * K * N * 4); hipMalloc(&cs, ns * N * 4);
  hipMemcpy(X, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(G, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice);
  const double fl = 2.0 * M * N * K; float t;
#define R(MODE, name) t = run<MODE>(M, K, N, ns, X, G, slab, cs); printf("%-24s %.3f ms %.0f TF\n", name, t, fl / t / 1e9);
  R(0, "full") R(0, "full") R(4, "no-epilogue") R(2, "no-glds") R(1, "no-mfma") R(3, "no-mfma no-glds") R(0, "full")
  return 0;
}

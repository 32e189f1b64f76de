// scratch (round 6): what does the L2/HBM -> LDS path (global_load_lds_dwordx4) deliver per CU as a function of the FOOTPRINT of one
// wave instruction?  The persistent NT kernel stages K in 32-wide steps: a wave instruction covers 16 rows x 64 B (half cache lines);
// the ablation build "LDS-DMA only" of round 2 took 213 us for what the matrix pipe needs 220 us for.  If the path moves whole 128-B
// lines per half-line request, a 64-wide K step (8 rows x 128 B per instruction) halves its cost.
//   PAT 0: 16 rows x 64 B   (today: stage = 256 A rows + 256 B rows, K = 32)
//   PAT 1:  8 rows x 128 B  (stage = 128 A rows + 128 B rows, K = 64; the two row halves alternate)
//   PAT 2:  1 KiB contiguous per wave instruction (a K-panel-major operand; upper bound)
// Same bytes per stage (32 KiB per workgroup), same ring (4 x 32 KiB), same counted wait (3 stages in flight), same tile walk
// (8 row bands x 4 column tiles per XCD round, 8 tiles per CU), M = 131072, N = K = 1024.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int PAT, int BAR>
__global__ __launch_bounds__(512, 2) void k_dma(const char* __restrict__ A, const char* __restrict__ Bt, int M, int N, int K,
                                                 int ntiles, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 32768];
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N >> 8, G = gridDim.x;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)wv * 1024u;
  const size_t lda = (size_t)K * 2;
  unsigned voff;
  if (PAT == 0 || PAT == 3) { const int prow = tid >> 2, pc = ((tid & 3) ^ (3 * ((prow >> 2) & 1))); voff = (unsigned)(prow * lda + pc * 16); }
  else if (PAT == 1 || PAT == 4) { const int prow = tid >> 3, pc = (tid & 7) ^ (prow & 7); voff = (unsigned)(prow * lda + pc * 16); }
  else voff = (unsigned)tid * 16u;
  auto dma = [&](const char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  };
  const unsigned long long t0 = __builtin_readcyclecounter();
  int slot = 0;
  for (int bid = blockIdx.x; bid < ntiles; bid += G) {
    const int t = xcd_remap(bid, ntiles);
    const size_t m0 = (size_t)(t / ntn) << 8, n0 = (size_t)(t % ntn) << 8;
    for (int st = 0; st < (K >> 5); ++st) {
      const unsigned la = lds_base + (unsigned)slot * 32768u;
      if (PAT == 0) {
        const char* ab = A + m0 * lda + (size_t)st * 64;
        const char* bb = Bt + n0 * lda + (size_t)st * 64;
        dma(ab, la); dma(ab + lda * 128, la + 8192); dma(bb, la + 16384); dma(bb + lda * 128, la + 24576);
      } else if (PAT == 3) {
        if (st & 1) {      // both stages of a K64 pair, half-lines of the same rows adjacent in the queue; nothing in even iterations
          const unsigned lb_ = lds_base + (unsigned)((slot + 1) & 3) * 32768u;
          const char* ab = A + m0 * lda + (size_t)(st - 1) * 64;
          const char* bb = Bt + n0 * lda + (size_t)(st - 1) * 64;
          dma(ab, la); dma(ab + 64, lb_); dma(ab + lda * 128, la + 8192); dma(ab + lda * 128 + 64, lb_ + 8192);
          dma(bb, la + 16384); dma(bb + 64, lb_ + 16384); dma(bb + lda * 128, la + 24576); dma(bb + lda * 128 + 64, lb_ + 24576);
        }
      } else if (PAT == 4) {
        if (st & 1) {      // full lines, both row halves of a K64 step in one burst
          const unsigned lb_ = lds_base + (unsigned)((slot + 1) & 3) * 32768u;
          const char* ab = A + m0 * lda + (size_t)(st >> 1) * 128;
          const char* bb = Bt + n0 * lda + (size_t)(st >> 1) * 128;
          dma(ab, la); dma(ab + lda * 64, la + 8192); dma(ab + lda * 128, lb_); dma(ab + lda * 192, lb_ + 8192);
          dma(bb, la + 16384); dma(bb + lda * 64, la + 24576); dma(bb + lda * 128, lb_ + 16384); dma(bb + lda * 192, lb_ + 24576);
        }
      } else if (PAT == 1) {
        const int h = st & 1;                                   // row half; K step of 64 = two stages
        const char* ab = A + (m0 + h * 128) * lda + (size_t)(st >> 1) * 128;
        const char* bb = Bt + (n0 + h * 128) * lda + (size_t)(st >> 1) * 128;
        dma(ab, la); dma(ab + lda * 64, la + 8192); dma(bb, la + 16384); dma(bb + lda * 64, la + 24576);
      } else {
        // panel-major: [K/32][rows][32] -- a stage's 256 rows x 64 B are 16 KiB contiguous
        const char* ab = A + ((size_t)st * M + m0) * 64;
        const char* bb = Bt + ((size_t)st * N + n0) * 64;
        dma(ab, la); dma(ab + 8192, la + 8192); dma(bb, la + 16384); dma(bb + 8192, la + 24576);
      }
      slot = (slot + 1) & 3;
      if (PAT >= 3) { if (st & 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }      // the previous pair has landed
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      if (BAR) __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PAT, int BAR> void run(const char* name, const char* A, const char* Bt, int M, int N, int K, unsigned long long* cyc) {
  const int ntiles = (M >> 8) * (N >> 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_dma<PAT, BAR>), dim3(256), dim3(512), 0, 0, A, Bt, M, N, K, ntiles, cyc);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_dma<PAT, BAR>), dim3(256), dim3(512), 0, 0, A, Bt, M, N, K, ntiles, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double c = 0; for (auto v : h) c += (double)v; c /= 256;
  const double bytes_cu = (double)ntiles / 256 * (K >> 5) * 32768.0;
  printf("%-34s bar %d: %7.1f us per launch   %8.0f cyc per WG   %5.1f B/clk/CU   %6.2f TB/s chip (L2->LDS bytes %.2f GB)\n", name, BAR,
         ms * 1e3 / reps, c, bytes_cu / c, bytes_cu * 256 / (ms * 1e-3 / reps) * 1e-12, bytes_cu * 256 * 1e-9);
}

int main() {
  const int M = 131072, N = 1024, K = 1024;
  char *A, *Bt; unsigned long long* cyc;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&Bt, (size_t)N * K * 2); hipMalloc(&cyc, 256 * 8);
  std::vector<unsigned short> h((size_t)M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff));
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(Bt, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>("16 rows x 64 B (today, K32)", A, Bt, M, N, K, cyc);
    run<1, 0>("8 rows x 128 B (K64 halves)", A, Bt, M, N, K, cyc);
    run<2, 0>("1 KiB contiguous (panel-major)", A, Bt, M, N, K, cyc);
    run<3, 0>("16 rows x 64 B, K32 pairs adjacent", A, Bt, M, N, K, cyc);
    run<4, 0>("8 rows x 128 B, K64 in one burst", A, Bt, M, N, K, cyc);
    run<3, 1>("16 rows x 64 B, K32 pairs adjacent", A, Bt, M, N, K, cyc);
    run<4, 1>("8 rows x 128 B, K64 in one burst", A, Bt, M, N, K, cyc);
    run<0, 1>("16 rows x 64 B (today, K32)", A, Bt, M, N, K, cyc);
    run<1, 1>("8 rows x 128 B (K64 halves)", A, Bt, M, N, K, cyc);
    run<2, 1>("1 KiB contiguous (panel-major)", A, Bt, M, N, K, cyc);
  }
  return 0;
}

// scratch: what does a 128 KiB tile epilogue cost per CU as a function of the store address pattern?
// 512 threads (8 waves) per workgroup, each lane issues 16 x 16-byte stores; s_memtime around issue and around ack.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int PAT>
__global__ __launch_bounds__(512) void k(uint4* out, unsigned long long* tr, int ldc_bytes, int nwaves) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (wv >= nwaves) return;
  char* base = (char*)out + (size_t)blockIdx.x * 256 * ldc_bytes;   // a 256-row band, row pitch ldc_bytes
  uint4 v = make_uint4(tid, lane, wv, blockIdx.x);
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    size_t off;
    const int wm = wv >> 2, wn = wv & 3;
    if (PAT == 0) {        // current epilogue: 16 rows x 64 B per instruction (lane&15 = row, lane>>4 = 16-B chunk)
      const int i = s >> 1, jp = s & 1;
      off = (size_t)(wm * 128 + i * 16 + (lane & 15)) * ldc_bytes + wn * 128 + jp * 64 + (lane >> 4) * 16;
    } else if (PAT == 1) { // 8 rows x 128 B per instruction
      const int i = s >> 1, h = s & 1;
      off = (size_t)(wm * 128 + i * 16 + h * 8 + (lane >> 3)) * ldc_bytes + wn * 128 + (lane & 7) * 16;
    } else if (PAT == 2) { // 2 rows x 512 B per instruction (whole tile rows), wave owns 32 rows
      off = (size_t)(wv * 32 + s * 2 + (lane >> 5)) * ldc_bytes + (lane & 31) * 16;
    } else {               // 1 KiB fully contiguous per instruction (tile stored as a packed 128 KiB block)
      off = (size_t)blockIdx.x * 0 + ((size_t)(wv * 16 + s) * 64 + lane) * 16;
      base = (char*)out + (size_t)blockIdx.x * 131072;
    }
    *(uint4*)(base + off) = v;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t2 = __builtin_readcyclecounter();
  if (lane == 0) { tr[(blockIdx.x * 8 + wv) * 3] = t0; tr[(blockIdx.x * 8 + wv) * 3 + 1] = t1; tr[(blockIdx.x * 8 + wv) * 3 + 2] = t2; }
}
template <int PAT> void run(const char* name, int nwg, int nwaves, uint4* out, unsigned long long* tr) {
  std::vector<unsigned long long> h(nwg * 8 * 3);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(tr, 0, nwg * 8 * 3 * 8);
    hipLaunchKernelGGL(k<PAT>, dim3(nwg), dim3(512), 0, 0, out, tr, 2048, nwaves);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
  double issue = 0, ack = 0; int n = 0;
  for (int b = 0; b < nwg; ++b) {
    unsigned long long t0 = ~0ull, t1 = 0, t2 = 0;
    for (int w = 0; w < nwaves; ++w) { auto* p = &h[(b * 8 + w) * 3]; if (p[0] < t0) t0 = p[0]; if (p[1] > t1) t1 = p[1]; if (p[2] > t2) t2 = p[2]; }
    issue += (double)(t1 - t0); ack += (double)(t2 - t0); ++n;
  }
  printf("%-44s wgs %4d waves %d: issue %7.0f cyc  issue+ack %7.0f cyc  (%d KiB per WG)\n", name, nwg, nwaves, issue / n, ack / n, nwaves * 16);
}
int main() {
  uint4* out; unsigned long long* tr;
  hipMalloc(&out, (size_t)2048 * 131072 + (1 << 20)); hipMalloc(&tr, 2048 * 8 * 3 * 8);
  for (int nwg : {32, 256}) {
    run<0>("16 rows x 64 B (current)", nwg, 8, out, tr);
    run<1>("8 rows x 128 B", nwg, 8, out, tr);
    run<2>("2 rows x 512 B", nwg, 8, out, tr);
    run<3>("1 KiB contiguous", nwg, 8, out, tr);
    run<0>("16 rows x 64 B (current), 4 waves", nwg, 4, out, tr);
    run<0>("16 rows x 64 B (current), 1 wave", nwg, 1, out, tr);
    run<3>("1 KiB contiguous, 1 wave", nwg, 1, out, tr);
  }
  return 0;
}

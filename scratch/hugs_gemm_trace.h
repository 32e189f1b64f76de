// scratch: measurement hooks of csrc/hugs_gemm.hip (built only with -DHUGS_TRACE -I scratch; see scratch/build_trace.sh)
__device__ unsigned long long* g_nt_trace;
extern "C" int hugs_debug_set_trace(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nt_trace), &p, sizeof(p));
}
__device__ int g_nt_stagger[2];   // [0] = groups (power of two), [1] = s_sleep(16) (~1k clock) units per group step
extern "C" int hugs_debug_set_stagger(int groups, int iters) {
  int v[2] = {groups, iters};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nt_stagger), v, sizeof(v));
}
// groups > 1: workgroups staggered by their slot inside the XCD ((blockIdx >> 3) & (groups - 1)); groups < 0: by XCD
// (blockIdx & 7; XCDs share no operand data, so nothing pulls them back into lock-step)
#define HUGS_STAGGER() { const int g_ = g_nt_stagger[0]; if ((g_ > 1 || g_ < 0) && blockIdx.x < 256) { \
    const int n_ = (g_ < 0 ? (int)(blockIdx.x & 7) : (int)((blockIdx.x >> 3) & (g_ - 1))) * g_nt_stagger[1]; \
    for (int q_ = 0; q_ < n_; ++q_) __builtin_amdgcn_s_sleep(16); } }   /* unit: 16 x 64 = ~1k clocks */
#ifdef HUGS_TRACE_PLAIN_STORE
#define HUGS_EPI_STORE(v_, p_) (*(p_) = (v_))
#endif
#define HUGS_TR(i) { if (g_nt_trace && threadIdx.x == 0) g_nt_trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); }
#define HUGS_TRP(i, k) { if (g_nt_trace && threadIdx.x == 0 && (i) < 16) g_nt_trace[((size_t)blockIdx.x * 16 + (i)) * 4 + (k)] = __builtin_readcyclecounter(); }
#define HUGS_TR_ID() { if (g_nt_trace && threadIdx.x == 0) { g_nt_trace[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg(63492); \
                                                             g_nt_trace[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg(63508); } }
// per-half-iteration cycle sums of every tile (k_gemm_nt_bf16_p64), kept in registers (a stamp STORED per half sits in the vmcnt queue the
// next barrier's vmcnt(0) waits for: it measured its own acknowledgement).  Stamp h marks the start of half h; sum[0] = H0 halves
// (no barrier), sum[1] = H1 halves, both over super-stages 1 .. ns2-1 of all tiles; written once at the end of the kernel.
#define HUGS_TRH_DECL unsigned long long trh_prev = 0, trh_sum0 = 0, trh_sum1 = 0;
#define HUGS_TRH(i, h) { const unsigned long long n_ = __builtin_readcyclecounter(); if ((h) >= 3) { if ((h) & 1) trh_sum0 += n_ - trh_prev; else trh_sum1 += n_ - trh_prev; } trh_prev = n_; }
#define HUGS_TRH_END() { if (g_nt_trace && threadIdx.x == 0) { g_nt_trace[256 * 16 * 4 + (size_t)blockIdx.x * 2] = trh_sum0; g_nt_trace[256 * 16 * 4 + (size_t)blockIdx.x * 2 + 1] = trh_sum1; } }

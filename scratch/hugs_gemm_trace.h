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

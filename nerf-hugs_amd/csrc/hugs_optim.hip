// Fused optimizer step on the flat fp32 parameter buffer: gradient statistics, per-module clip
// (value then norm), nan_to_num, Adam, update statistics -- two streaming passes over the 36 MB
// buffers instead of the reference's five pytree traversals; plus the compute-dtype weight casts
// (natural [K,N] and transposed [N,K] copies the MFMA GEMMs read).
//
// Replaces (reference, MipNeRF360/internal/train_utils.py): :442 weight_l2s, :461-462 grad norms/maxes,
// :351-369 clip_gradients, :466 nan_to_num, :468 apply_gradients (optax.adam, :487-512), :470-473
// opt_update norms/maxes.
#include "hugs_common.h"

struct OptChunk { int off, len, leaf, module; };

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum_f(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wave_max_f(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// part1[chunk] = {sum g^2, max|g|, sum theta^2, sum clamp(g)^2}  (g already scaled by gscale = 1/world)
__global__ __launch_bounds__(256) void k_opt_stats(const OptChunk* __restrict__ chunks, const float* __restrict__ theta,
                                                   const float* __restrict__ grad, float gscale, float max_val,
                                                   float* __restrict__ part1) {
  __shared__ float red[4];
  const OptChunk c = chunks[blockIdx.x];
  float sg = 0.f, mg = 0.f, st = 0.f, sc = 0.f;
  auto one = [&](float g, float t) {
    g *= gscale;
    sg += g * g;
    mg = fmaxf(mg, fabsf(g));   // NaN-ignoring like jnp.max? jnp.max propagates NaN; handled via sg
    st += t * t;
    const float gc = max_val > 0.f ? fminf(fmaxf(g, -max_val), max_val) : g;
    sc += gc * gc;
  };
  // chunk offsets are multiples of 4 floats (leaves are padded to 4): 16-byte accesses, scalar tail
  const int nv = (c.off & 3) ? 0 : (c.len >> 2);
  for (int i = threadIdx.x; i < nv; i += 256) {
    const float4 g = *(const float4*)(grad + c.off + 4 * i), t = *(const float4*)(theta + c.off + 4 * i);
    one(g.x, t.x); one(g.y, t.y); one(g.z, t.z); one(g.w, t.w);
  }
  for (int i = 4 * nv + threadIdx.x; i < c.len; i += 256) one(grad[c.off + i], theta[c.off + i]);
  sg = block_sum256(sg, red); mg = block_max256(mg, red); st = block_sum256(st, red); sc = block_sum256(sc, red);
  if (threadIdx.x == 0) { float* o = part1 + (size_t)blockIdx.x * 4; o[0] = sg; o[1] = mg; o[2] = st; o[3] = sc; }
}

// single workgroup: leaf_stats[leaf] = {sum g^2, max|g|, sum theta^2, sum clamp(g)^2};
// mod_scale[m] = min(1, max_norm/(eps+|g_m|)).  leaf_info[leaf] = {chunk_begin, chunk_end, module, 0}.
__global__ void k_opt_finalize1(int nleaf, int nmod, const int4* __restrict__ leaf_info,
                                const float* __restrict__ part1, float max_norm, float* __restrict__ leaf_stats,
                                float* __restrict__ mod_scale) {
  // a WAVE per leaf (round 4: a thread per leaf walked up to 96 chunk records one after the other: 24 us of pure latency at the
  // end of every step): lane l sums the chunks l, l + 64, ..., the lanes combine in a fixed xor butterfly -- deterministic
  __shared__ float s_sc[1024];
  __shared__ int s_mod[1024];
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int leaf = threadIdx.x >> 6; leaf < nleaf; leaf += nw) {
    const int4 li = leaf_info[leaf];
    float sg = 0.f, mg = 0.f, st = 0.f, sc = 0.f;
    for (int c = li.x + lane; c < li.y; c += 64) {
      const float4 p = *(const float4*)(part1 + (size_t)c * 4);
      sg += p.x; mg = fmaxf(mg, p.y); st += p.z; sc += p.w;
    }
    sg = wave_sum_f(sg); mg = wave_max_f(mg); st = wave_sum_f(st); sc = wave_sum_f(sc);
    if (lane == 0) {
      leaf_stats[leaf * 4] = sg; leaf_stats[leaf * 4 + 1] = mg; leaf_stats[leaf * 4 + 2] = st; leaf_stats[leaf * 4 + 3] = sc;
      s_sc[leaf] = sc; s_mod[leaf] = li.z;      // (round 5: the module pass below walked the leaves through ~2 dependent global loads each)
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < nmod) {   // fixed leaf order per module
    float sq = 0.f;
    for (int leaf = 0; leaf < nleaf; ++leaf)
      if (s_mod[leaf] == (int)threadIdx.x) sq += s_sc[leaf];
    float mult = 1.f;
    if (max_norm > 0.f) {
      const float x = max_norm / (HUGS_EPS + sqrtf(sq));
      mult = (x != x) ? x : fminf(1.f, x);   // jnp.minimum propagates NaN
    }
    mod_scale[threadIdx.x] = mult;
  }
}

// Adam on one chunk per workgroup.  trainable[leaf] == 0 -> update forced to zero (optax.set_to_zero).
// part2[chunk] = {sum delta^2, max|delta|}
__global__ __launch_bounds__(256) void k_opt_adam(const OptChunk* __restrict__ chunks, float* __restrict__ theta,
                                                  const float* __restrict__ grad, float* __restrict__ m,
                                                  float* __restrict__ v, const float* __restrict__ mod_scale,
                                                  const int* __restrict__ trainable, float gscale, float max_val,
                                                  float lr, float b1, float b2, float eps, float bc1, float bc2,
                                                  float* __restrict__ part2, const float* __restrict__ dyn) {
  __shared__ float red[4];
  if (dyn) { lr = dyn[0]; bc1 = dyn[1]; bc2 = dyn[2]; }      // hugs_opt_adam_dyn: the step's scalars from device memory
  const OptChunk c = chunks[blockIdx.x];
  float sd = 0.f, md = 0.f;
  if (!trainable || trainable[c.leaf]) {
    const float mult = mod_scale[c.module];
    auto one = [&](float g, float& mi, float& vi, float& th) {
      g *= gscale;
      if (max_val > 0.f) g = fminf(fmaxf(g, -max_val), max_val);
      g = mult * g;
      if (g != g) g = 0.f;                               // nan_to_num
      else if (g == __builtin_inff()) g = 3.4028234664e38f;
      else if (g == -__builtin_inff()) g = -3.4028234664e38f;
      mi = b1 * mi + (1.f - b1) * g;
      vi = b2 * vi + (1.f - b2) * g * g;
      const float delta = -lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
      const float t0 = th;
      th = t0 + delta;
      const float d = th - t0;                           // the reference reports new - old
      sd += d * d; md = fmaxf(md, fabsf(d));
    };
    const int nv = (c.off & 3) ? 0 : (c.len >> 2);       // 16-byte accesses (chunk offsets are multiples of 4 floats), scalar tail
    for (int i = threadIdx.x; i < nv; i += 256) {
      const int ix = c.off + 4 * i;
      // (round 5: the gradient and both moments are touched once per step -- streaming loads / stores, so that they do not push the
      //  masters, which the next step's operand casts read, out of the 256 MB memory-side cache)
      typedef float __attribute__((ext_vector_type(4))) f4;
      const f4 gq = __builtin_nontemporal_load((const f4*)(grad + ix));
      const f4 mq = __builtin_nontemporal_load((const f4*)(m + ix)), vq = __builtin_nontemporal_load((const f4*)(v + ix));
      const float4 g = make_float4(gq.x, gq.y, gq.z, gq.w);
      float4 mi = make_float4(mq.x, mq.y, mq.z, mq.w), vi = make_float4(vq.x, vq.y, vq.z, vq.w), th = *(const float4*)(theta + ix);
      one(g.x, mi.x, vi.x, th.x); one(g.y, mi.y, vi.y, th.y); one(g.z, mi.z, vi.z, th.z); one(g.w, mi.w, vi.w, th.w);
      __builtin_nontemporal_store(f4{mi.x, mi.y, mi.z, mi.w}, (f4*)(m + ix));
      __builtin_nontemporal_store(f4{vi.x, vi.y, vi.z, vi.w}, (f4*)(v + ix));
      *(float4*)(theta + ix) = th;
    }
    for (int i = 4 * nv + threadIdx.x; i < c.len; i += 256) {
      const int ix = c.off + i;
      float mi = m[ix], vi = v[ix], th = theta[ix];
      one(grad[ix], mi, vi, th);
      m[ix] = mi; v[ix] = vi; theta[ix] = th;
    }
  }
  sd = block_sum256(sd, red); md = block_max256(md, red);
  if (threadIdx.x == 0) { part2[(size_t)blockIdx.x * 2] = sd; part2[(size_t)blockIdx.x * 2 + 1] = md; }
}

// What a captured step hands back, written by the step's LAST launch (round 5: it was three blit copies of ~13 us each behind the
// graph -- stat tail -> packed, packed -> pinned host memory, the advanced key -> a fresh tensor): packed[0..ntail) = tail * tscale,
// thr_dst[l] = packed[thr_off + l * thr_stride] (RobustNeRF's device-side threshold feedback), the whole packed buffer to the host
// slot and the key to the device buffer whose ADDRESSES this step's staging launch wrote to ptrs[0..1] (hugs_stage_step_pub: a
// captured launch's arguments are fixed, the table's content is not).  A null pointer in the table skips that copy.
struct StepPublish {
  const float* tail; float* packed; int ntail, npacked; float tscale;
  float* thr_dst; int thr_off, thr_stride, thr_n;
  const int* key_src; int* key_dst;
  const unsigned long long* ptrs;
};
__global__ void k_opt_finalize2(int nleaf, const int4* __restrict__ leaf_info, const float* __restrict__ part2,
                                float* __restrict__ leaf_upd, const StepPublish P) {
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;      // a wave per leaf, as k_opt_finalize1
  for (int leaf = threadIdx.x >> 6; leaf < nleaf; leaf += nw) {
    const int4 li = leaf_info[leaf];
    float sd = 0.f, md = 0.f;
    for (int c = li.x + lane; c < li.y; c += 64) { const float2 p = *(const float2*)(part2 + (size_t)c * 2); sd += p.x; md = fmaxf(md, p.y); }
    sd = wave_sum_f(sd); md = wave_max_f(md);
    if (lane == 0) { leaf_upd[leaf * 2] = sd; leaf_upd[leaf * 2 + 1] = md; }
  }
  if (!P.packed) return;
  for (int i = threadIdx.x; i < P.ntail; i += blockDim.x) P.packed[i] = P.tail[i] * P.tscale;
  __syncthreads();      // (one workgroup: the per-leaf sums above and the scaled tail are visible to all of it)
  if (P.thr_dst && (int)threadIdx.x < P.thr_n) P.thr_dst[threadIdx.x] = P.packed[P.thr_off + threadIdx.x * P.thr_stride];
  float* host = P.ptrs ? (float*)P.ptrs[0] : nullptr;
  int* key2 = P.ptrs ? (int*)P.ptrs[1] : nullptr;
  if (host) {
    for (int i = threadIdx.x; i < P.npacked; i += blockDim.x) host[i] = P.packed[i];
    __threadfence_system();
  }
  if (P.key_src && threadIdx.x < 2) {
    const int k = P.key_src[threadIdx.x];
    if (P.key_dst) P.key_dst[threadIdx.x] = k;
    if (key2) key2[threadIdx.x] = k;
  }
}

// W fp32 [K,N] -> Wn (natural, [K,N]) and Wt (transposed, [N,K]) in the compute dtype. 32x32 LDS tiles.
template <int BF16>
__global__ __launch_bounds__(256) void k_cast_weights(int K, int N, const float* __restrict__ W, void* __restrict__ Wn,
                                                      void* __restrict__ Wt) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    float x = 0.f;
    if (k < K && n < N) {
      x = W[(size_t)k * N + n];
      if (Wn) { if (BF16) ((uint16_t*)Wn)[(size_t)k * N + n] = f_to_op16(x, BF16); else ((float*)Wn)[(size_t)k * N + n] = x; }
    }
    tile[r][tx] = x;
  }
  __syncthreads();
  if (Wt)
    for (int r = ty; r < 32; r += 8) {
      const int n = n0 + r, k = k0 + tx;
      if (k < K && n < N) {
        const float x = tile[tx][r];
        if (BF16) ((uint16_t*)Wt)[(size_t)n * K + k] = f_to_op16(x, BF16); else ((float*)Wt)[(size_t)n * K + k] = x;
      }
    }
}

// The same cast for a table of matrices in ONE launch (a train step refreshes 14 operand pairs: 14 launches of ~5 us).
struct CastItem { const float* W; void* Wn; void* Wt; int K, N, blk0, nbx; };
template <int BF16>
__global__ __launch_bounds__(256) void k_cast_weights_batch(int nitems, const CastItem* __restrict__ items) {
  __shared__ float tile[32][33];
  int it = 0;
  while (it + 1 < nitems && (int)blockIdx.x >= items[it + 1].blk0) ++it;
  const CastItem I = items[it];
  const int b = blockIdx.x - I.blk0;
  const int k0 = (b / I.nbx) * 32, n0 = (b % I.nbx) * 32, K = I.K, N = I.N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    float x = 0.f;
    if (k < K && n < N) {
      x = I.W[(size_t)k * N + n];
      if (I.Wn) { if (BF16) ((uint16_t*)I.Wn)[(size_t)k * N + n] = f_to_op16(x, BF16); else ((float*)I.Wn)[(size_t)k * N + n] = x; }
    }
    tile[r][tx] = x;
  }
  __syncthreads();
  if (I.Wt)
    for (int r = ty; r < 32; r += 8) {
      const int n = n0 + r, k = k0 + tx;
      if (k < K && n < N) {
        const float x = tile[tx][r];
        if (BF16) ((uint16_t*)I.Wt)[(size_t)n * K + k] = f_to_op16(x, BF16); else ((float*)I.Wt)[(size_t)n * K + k] = x;
      }
    }
}

extern "C" int hugs_opt_stats(int nchunks, int nleaf, int nmod, const void* chunks, const void* leaf_info,
                              const float* theta, const float* grad,
                              float gscale, float max_val, float max_norm, float* part1_ws, float* leaf_stats,
                              float* mod_scale, void* stream) {
  HUGS_REQUIRE(nmod <= 16 && nleaf <= 1024, -3, "hugs_opt_stats: too many modules/leaves (%d/%d)", nmod, nleaf);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_opt_stats, dim3(nchunks), dim3(256), 0, st, (const OptChunk*)chunks, theta, grad, gscale, max_val, part1_ws);
  hipLaunchKernelGGL(k_opt_finalize1, dim3(1), dim3(1024), 0, st, nleaf, nmod, (const int4*)leaf_info, part1_ws, max_norm,
                     leaf_stats, mod_scale);
  HUGS_CHECK_LAUNCH("hugs_opt_stats");
  return 0;
}

static int opt_adam_impl(int nchunks, int nleaf, const void* chunks, const void* leaf_info, float* theta, const float* grad, float* m, float* v,
                         const float* mod_scale, const int* trainable, float gscale, float max_val, float lr, float b1,
                         float b2, float eps, float bias_corr1, float bias_corr2, float* part2_ws, float* leaf_upd,
                         const float* dyn, const StepPublish& P, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_opt_adam, dim3(nchunks), dim3(256), 0, st, (const OptChunk*)chunks, theta, grad, m, v, mod_scale,
                     trainable, gscale, max_val, lr, b1, b2, eps, bias_corr1, bias_corr2, part2_ws, dyn);
  hipLaunchKernelGGL(k_opt_finalize2, dim3(1), dim3(1024), 0, st, nleaf, (const int4*)leaf_info, part2_ws, leaf_upd, P);
  HUGS_CHECK_LAUNCH("hugs_opt_adam");
  return 0;
}
extern "C" int hugs_opt_adam(int nchunks, int nleaf, const void* chunks, const void* leaf_info, float* theta, const float* grad, float* m, float* v,
                             const float* mod_scale, const int* trainable, float gscale, float max_val, float lr, float b1,
                             float b2, float eps, float bias_corr1, float bias_corr2, float* part2_ws, float* leaf_upd,
                             void* stream) {
  return opt_adam_impl(nchunks, nleaf, chunks, leaf_info, theta, grad, m, v, mod_scale, trainable, gscale, max_val, lr, b1, b2, eps,
                       bias_corr1, bias_corr2, part2_ws, leaf_upd, nullptr, StepPublish{}, stream);
}
// dyn: 3 device floats {lr, bias_corr1, bias_corr2}: the per-step scalars of a captured (hipGraph) train step
extern "C" int hugs_opt_adam_dyn(int nchunks, int nleaf, const void* chunks, const void* leaf_info, float* theta, const float* grad, float* m,
                                 float* v, const float* mod_scale, const int* trainable, float gscale, float max_val, const float* dyn,
                                 float b1, float b2, float eps, float* part2_ws, float* leaf_upd, void* stream) {
  HUGS_REQUIRE(dyn, -2, "hugs_opt_adam_dyn: dyn is null");
  return opt_adam_impl(nchunks, nleaf, chunks, leaf_info, theta, grad, m, v, mod_scale, trainable, gscale, max_val, 0.f, b1, b2, eps, 1.f,
                       1.f, part2_ws, leaf_upd, dyn, StepPublish{}, stream);
}
// hugs_opt_adam_dyn whose last launch also publishes the step's results (include/hugs.h): pub = 12 host words
//   {tail, packed, ntail, npacked, tscale (float bits), thr_dst, thr_off, thr_stride, thr_n, key_src, key_dst, ptrs}
extern "C" int hugs_opt_adam_pub(int nchunks, int nleaf, const void* chunks, const void* leaf_info, float* theta, const float* grad, float* m,
                                 float* v, const float* mod_scale, const int* trainable, float gscale, float max_val, const float* dyn,
                                 float b1, float b2, float eps, float* part2_ws, float* leaf_upd, const unsigned long long* pub, void* stream) {
  HUGS_REQUIRE(dyn && pub, -2, "hugs_opt_adam_pub: dyn / pub is null");
  StepPublish P;
  P.tail = (const float*)pub[0]; P.packed = (float*)pub[1]; P.ntail = (int)pub[2]; P.npacked = (int)pub[3];
  { const unsigned u = (unsigned)pub[4]; __builtin_memcpy(&P.tscale, &u, 4); }
  P.thr_dst = (float*)pub[5]; P.thr_off = (int)pub[6]; P.thr_stride = (int)pub[7]; P.thr_n = (int)pub[8];
  P.key_src = (const int*)pub[9]; P.key_dst = (int*)pub[10]; P.ptrs = (const unsigned long long*)pub[11];
  HUGS_REQUIRE(P.packed && P.tail && P.ntail >= 0 && P.ntail <= P.npacked && P.thr_n >= 0 && P.thr_n <= 1024, -2,
               "hugs_opt_adam_pub: descriptor (ntail %d, npacked %d, thr_n %d)", P.ntail, P.npacked, P.thr_n);
  return opt_adam_impl(nchunks, nleaf, chunks, leaf_info, theta, grad, m, v, mod_scale, trainable, gscale, max_val, 0.f, b1, b2, eps, 1.f,
                       1.f, part2_ws, leaf_upd, dyn, P, stream);
}
__global__ void k_set_floats(float* dst, int n, float a, float b, float c, float d) {
  const float v[4] = {a, b, c, d};
  if ((int)threadIdx.x < n) dst[threadIdx.x] = v[threadIdx.x];
}
// dst[0..n) = {a, b, c, d}[0..n), n <= 4: per-step scalars enter a captured step through the kernel arguments of this launch
extern "C" int hugs_set_floats(float* dst, int n, float a, float b, float c, float d, void* stream) {
  HUGS_REQUIRE(dst && n >= 0 && n <= 4, -2, "hugs_set_floats: n=%d outside 0..4", n);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_set_floats, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, n, a, b, c, d);
  HUGS_CHECK_LAUNCH("hugs_set_floats");
  return 0;
}

// One launch in front of a replayed (captured) step: the step's inputs -- up to 16 flat buffers of 4-byte words (ray fields, target
// colours, the jax key) -- are copied into the buffers the graph was captured on, and the per-step scalars go into dst_f[0..nf).
// The source ADDRESSES change from step to step (whatever batch the caller hands over), so they travel as kernel arguments.
struct StageItems { const uint32_t* src[16]; uint32_t* dst[16]; int words[16]; int n; float f[8]; };
__global__ __launch_bounds__(256) void k_stage_step(StageItems S, float* dst_f, int nf,
                                                    unsigned long long* dst_p, unsigned long long p0, unsigned long long p1) {
  const int it = blockIdx.y;
  if (it == S.n) {      // (the extra row of blocks: the scalars, and the two addresses the step's last launch publishes to)
    if (blockIdx.x == 0 && (int)threadIdx.x < nf) dst_f[threadIdx.x] = S.f[threadIdx.x];
    if (dst_p && blockIdx.x == 0 && threadIdx.x == 64) { dst_p[0] = p0; dst_p[1] = p1; }
    return;
  }
  const uint32_t* __restrict__ s_ = S.src[it];
  uint32_t* __restrict__ d_ = S.dst[it];
  const int n = S.words[it];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d_[i] = s_[i];
}
// include/hugs.h hugs_stage_step: src / dst are HOST arrays of n device pointers, words[i] = 4-byte words of item i
static int stage_step_impl(int n, const void* const* src, void* const* dst, const int* words, float* dst_f, int nf, const float* scal,
                           unsigned long long* dst_p, unsigned long long p0, unsigned long long p1, void* stream) {
  HUGS_REQUIRE(n >= 0 && n <= 16 && nf >= 0 && nf <= 8 && (n == 0 || (src && dst && words)) && (nf == 0 || (dst_f && scal)), -2,
               "hugs_stage_step: %d items (<= 16), %d scalars (<= 8)", n, nf);
  StageItems S;
  S.n = n;
  for (int i = 0; i < 8; ++i) S.f[i] = i < nf ? scal[i] : 0.f;
  int mx = 1;
  for (int i = 0; i < 16; ++i) {
    S.src[i] = i < n ? (const uint32_t*)src[i] : nullptr; S.dst[i] = i < n ? (uint32_t*)dst[i] : nullptr; S.words[i] = i < n ? words[i] : 0;
    HUGS_REQUIRE(i >= n || (S.src[i] && S.dst[i] && S.words[i] >= 0), -2, "hugs_stage_step: item %d", i);
    if (S.words[i] > mx) mx = S.words[i];
  }
  int gx = (mx + 1023) / 1024;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_stage_step, dim3(gx, n + 1), dim3(256), 0, (hipStream_t)stream, S, dst_f, nf, dst_p, p0, p1);
  HUGS_CHECK_LAUNCH("hugs_stage_step");
  return 0;
}
extern "C" int hugs_stage_step(int n, const void* const* src, void* const* dst, const int* words, float* dst_f, int nf, float a, float b,
                               float c, float d, void* stream) {
  HUGS_REQUIRE(nf >= 0 && nf <= 4, -2, "hugs_stage_step: %d scalars (<= 4)", nf);
  const float v[4] = {a, b, c, d};
  return stage_step_impl(n, src, dst, words, dst_f, nf, v, nullptr, 0, 0, stream);
}
// + dst_p[0..1] = {p0, p1}: the addresses (a pinned host slot for the packed stats, a device buffer for the advanced key; 0 = none)
// that hugs_opt_adam_pub's last launch reads back from dst_p
// (scalars: HOST array of nf <= 8 floats)
extern "C" int hugs_stage_step_pub(int n, const void* const* src, void* const* dst, const int* words, float* dst_f, int nf,
                                   const float* scalars, void* dst_p, void* p0, void* p1, void* stream) {
  HUGS_REQUIRE(dst_p, -2, "hugs_stage_step_pub: dst_p is null");
  return stage_step_impl(n, src, dst, words, dst_f, nf, scalars, (unsigned long long*)dst_p, (unsigned long long)p0, (unsigned long long)p1, stream);
}

extern "C" int hugs_cast_weights_batch(int dtype, int nitems, const void* items, int total_blocks, void* stream) {
  if (nitems <= 0 || total_blocks <= 0) return 0;
  if (dtype == 2) hipLaunchKernelGGL(k_cast_weights_batch<2>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, nitems, (const CastItem*)items);
  else if (dtype) hipLaunchKernelGGL(k_cast_weights_batch<1>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, nitems, (const CastItem*)items);
  else hipLaunchKernelGGL(k_cast_weights_batch<0>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, nitems, (const CastItem*)items);
  HUGS_CHECK_LAUNCH("hugs_cast_weights_batch");
  return 0;
}

extern "C" int hugs_cast_weights(int dtype, int K, int N, const float* W, void* Wn, void* Wt, void* stream) {
  if (K <= 0 || N <= 0) return 0;
  dim3 grid((N + 31) / 32, (K + 31) / 32);
  if (dtype == 2) hipLaunchKernelGGL(k_cast_weights<2>, grid, dim3(256), 0, (hipStream_t)stream, K, N, W, Wn, Wt);
  else if (dtype) hipLaunchKernelGGL(k_cast_weights<1>, grid, dim3(256), 0, (hipStream_t)stream, K, N, W, Wn, Wt);
  else hipLaunchKernelGGL(k_cast_weights<0>, grid, dim3(256), 0, (hipStream_t)stream, K, N, W, Wn, Wt);
  HUGS_CHECK_LAUNCH("hugs_cast_weights");
  return 0;
}

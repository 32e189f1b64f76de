/* hugs.h -- C ABI of the MI355X-native Mip-NeRF 360 per-ray hot path (libhugs_hip.so, gfx950).
 *
 * The reference (cnhaox/NeRF-HuGS, MipNeRF360/internal/*.py) has no FFI layer: models.py / train_utils.py
 * call jax.numpy directly and XLA is the backend.  These entry points are what an FFI for that path would
 * bind; each cites the reference function it replaces (paths relative to MipNeRF360/internal/).
 *
 * Conventions: every pointer is a caller-owned DEVICE pointer (host pointers only where stated); nothing is
 * allocated, no stream is created; `stream` is a hipStream_t; kernels are enqueued and the call returns.
 * Process-global state, all of it: (i) the thread-local last-error string; (ii) hugs_gemm_nt caches the device's CU
 * count on first use (the persistent kernel's grid).  (The Python layer above keeps per-process
 * caches of its own -- sampler abscissae uploaded once per (num_samples, mode), RobustNeRF thresholds fed back on
 * the device between steps: nerf-hugs_amd/internal/stepfun.py `_UB_CACHE`, train_utils.py `cache['thr_dev']`.)  Return 0 = ok, <0 = error (message via hugs_last_error(), thread local):
 *   -2 invalid argument for which the reference raises ValueError, -3 unsupported shape, -100 launch failure.
 * dtype: 0 = float32 (parity mode, v_mfma_f32_16x16x4_f32), 1 = bfloat16 operands with fp32 accumulate.
 *   2 = IEEE half operands with fp32 accumulate (v_mfma_f32_16x16x32_f16): accepted by hugs_gemm_nt / _tn / _nt_bits,
 *   hugs_cast_weights(_batch), the hash-grid / SH entries, the hugs_nf_* glue and fused proposal kernels, hugs_rank1_mask,
 *   hugs_mask_head_*, hugs_embed_scatter_add and the other head kernels -- what the nerfacto path's fp16 mode uses (the
 *   reference's enable_amp); the Mip-NeRF 360 encoder stays bf16 / fp32 (its 2^k x frequencies need the fp32 exponent range).
 * Activations/weights in `dtype`, everything per-ray / per-sample scalar in float32.  Row-major.
 */
#ifndef HUGS_H
#define HUGS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int hugs_version(void);
const char* hugs_last_error(void);
int hugs_device_count(void);

/* models.py:155-212 level prologue = stepfun.py:99-128 max_dilate_weights (renormalize) -> [1:-1] trim ->
 * models.py:191-193 annealed logits -> stepfun.py:131-161 softmax CDF + math.py:108-127 sorted_interp ->
 * stepfun.py:214-263 sample_intervals -> coord.py:63-99 s_to_t.  One wavefront per ray.
 * t_prev [nrays, n_prev+1], w_prev [nrays, n_prev]; u = u_base[j] + jitter[ray*jitter_stride (+j)] (jitter may
 * be NULL = rng None).  raydist (coord.py:78-90): 0 None, 1 reciprocal, 2 log, 3 exp, 4 sqrt, 5 square, 6 piecewise.  Outputs sdist,tdist [nrays, num_samples+1];
 * optional test hooks idx_out [nrays,num_samples] (CDF interval index), t_in_out/w_in_out (the dilated,
 * trimmed step function).  sum_order: order of the three order-sensitive float sums (dilation renormaliser
 * stepfun.py:126, softmax denominator :142, CDF cumsum :145): 1 = the reference's calls executed with numpy float32
 * semantics (pairwise jnp.sum, sequential jnp.cumsum -- what the reference-generated fixtures pin), 0 = wave order
 * (lane-blocked tree).  Bit-exact against oracle/stepfun_ref.c in either order.  -2 if num_samples <= 1 (stepfun.py:239). */
int hugs_level_sample_fwd(int nrays, const float* t_prev, const float* w_prev, int n_prev, int do_dilate,
                          float dilation, float domain_lo, float domain_hi, float anneal, float resample_padding,
                          const float* u_base, const float* jitter, int jitter_stride, int num_samples, int raydist,
                          int sum_order, const float* near, const float* far, float* sdist, float* tdist, int32_t* idx_out,
                          float* t_in_out, float* w_in_out, void* stream);
/* The same launch with `anneal` read from device memory (no test hooks): the form a captured (hipGraph) train step uses, whose
 * per-step scalars must not be baked into kernel arguments. */
int hugs_level_sample_fwd_dyn(int nrays, const float* t_prev, const float* w_prev, int n_prev, int do_dilate,
                              float dilation, float domain_lo, float domain_hi, const float* anneal_dev, float resample_padding,
                              const float* u_base, const float* jitter, int jitter_stride, int num_samples, int raydist,
                              int sum_order, const float* near, const float* far, float* sdist, float* tdist, void* stream);

/* render.py:103-127 cast_rays (cone :44-78 / cylinder :81-100, lift_gaussian :21-41 diag=False) ->
 * coord.py:21-27,39-60 contract + track_linearize (closed-form Jacobian) -> coord.py:129-133
 * lift_and_diagonalize -> coord.py:102-126 integrated_pos_enc.  out [nrays*num_samples, row_pitch] in
 * bf16/fp32, columns >= 2*num_basis*max_deg zero.  basis [3, num_basis] fp32.  ray_shape 0 cone / 1 cylinder
 * (-2 otherwise, render.py:124); + 4: the Gaussians' covariances are set to zero after casting (Model.disable_integration,
 * models.py:223-226); bits 8-15: min_deg_point -- the scales are 2^(min_deg + k), k < max_deg, i.e. `max_deg` counts the degrees
 * (coord.py:107-126 integrated_pos_enc(mean, var, min_deg, max_deg)). */
int hugs_cast_ipe_fwd(int nrays, int num_samples, const float* tdist, const float* origins, const float* directions,
                      const float* radii, const float* basis, int num_basis, int ray_shape, int warp_contract,
                      int max_deg, int out_bf16, int row_pitch, void* out, void* stream);
/* coord.py:136-147 pos_enc(viewdirs, 0, deg, append_identity=True) -> [nrays, 3+6*deg] */
int hugs_dir_enc_fwd(int nrays, int deg, const float* viewdirs, float* out, void* stream);

/* models.py:451-455 Dense+relu(+skip concat as a second A panel), :475 bottleneck, :508-512 view layer, and the
 * dX half of their backward:  out[M,N] = epi([A1|A2][M,K1+K2] * Bt[N,K1+K2]^T) with
 * epi(x) = (x + bias[n] + row_bias[m/row_div][n] + r1_row[m]*r1_col[n]) -> relu? -> * (mask[m][n] > 0)?.
 * M,N multiples of 128; K1,K2 multiples of 64 (bf16) / 16 (fp32). */
int hugs_gemm_nt(int dtype, int M, int N, int K1, int K2, const void* A1, int lda1, const void* A2, int lda2,
                 const void* Bt, int ldb, const float* bias, const float* row_bias, int row_div, int ld_rb, int relu,
                 const void* mask, int ld_mask, const float* r1_row, const float* r1_col, void* out, int ldc,
                 void* stream);
/* weight gradients of the same layers: dW[Kc,N] = X[Mrows,Kc]^T G[Mrows,N], dbias[N] = colsum(G) (optional);
 * Mrows split into nsplit fp32 slabs reduced in fixed order (deterministic). */
long long hugs_gemm_tn_ws_bytes(int Kc, int N, int nsplit);
int hugs_gemm_tn(int dtype, int Mrows, int Kc, int N, int nsplit, const void* X, int ldx, const void* G, int ldg,
                 float* dW, float* dbias, void* ws, void* stream);
/* The weight gradients of SEVERAL layers (train_utils.py:454: everything jax.value_and_grad derives for the Dense kernels of
 * models.py:451-455) in one launch: item i is dW_i[Kc,N] = X_i[Mrows,Kc]^T G_i[Mrows,N] (+ dbias_i = colsum G_i when
 * non-null).  bf16 / fp16 operands (dtype 1 / 2), Kc and N multiples of 256, Mrows a multiple of 64, <= 16 items.  The
 * reduction rows of every item are cut into nsplit pieces (fixed partition, fp32 partial tiles summed in piece order:
 * deterministic); hugs_gemm_tn_batch_nsplit returns #CUs / (total 256x256 tiles), the value that fills the chip once.
 * `items` is a HOST array; ws: hugs_gemm_tn_batch_ws_bytes(nitems, items, nsplit) device bytes. */
typedef struct HugsTnItem {
  const void* X; const void* G; float* dW; float* dbias;
  int ldx, ldg, Mrows, Kc, N, reserved;
} HugsTnItem;
int hugs_gemm_tn_batch_nsplit(int nitems, const HugsTnItem* items);
long long hugs_gemm_tn_batch_ws_bytes(int nitems, const HugsTnItem* items, int nsplit);
int hugs_gemm_tn_batch(int dtype, int nitems, const HugsTnItem* items, int nsplit, void* ws, void* stream);

/* models.py:451-456,467 for a 256-wide trunk behind its first layer, in ONE launch: Y_l = relu(Y_{l-1} W_l + b_l), l = 1..nl (every
 * Wt[l-1] is the [256 (out), 256 (in)] bf16 operand copy, Y[l-1] [M,256] bf16 is written for the backward pass, bits[l-1] -- if
 * non-null -- receives the 1-bit relu masks in hugs_gemm_nt_bits' layout), then raw = Y_nl . wd + bd, density = softplus(raw +
 * density_bias) (wd null: no head).  A 128-row activation tile stays in LDS from layer to layer.  Wt / bias / Y / bits are HOST
 * arrays of nl device pointers; M a multiple of 256, nl <= 7, dtype 1 (bf16). */
int hugs_mlp256_tail_max_layers(void);      /* the nl cap of hugs_mlp256_tail_fwd (callers fall back to per-layer GEMMs above it) */
int hugs_mlp256_tail_fwd(int dtype, int M, int nl, const void* Y0, const void* const* Wt, const float* const* bias,
                         void* const* Y, uint32_t* const* bits, const float* wd, const float* bd, float density_bias, float* raw,
                         float* density, void* stream);
/* Backward twin of hugs_mlp256_tail_fwd for nl == 3 (the reference's PropMLP, models.py:451-456,467 with disable_rgb; what
 * jax.value_and_grad derives, train_utils.py:454): G3 = (d_raw (x) wd) * (Y3 > 0), G_{l-1} = (G_l W_l^T) * (Y_{l-1} > 0) for l = 3, 2, 1 in
 * ONE launch with the three operand copies Wn[l-1] ([256 (fan_in)][256 (fan_out)] bf16) resident in registers.  bits[0..3]: the 1-bit
 * relu masks of Y0 .. Y3 as hugs_gemm_nt_bits / hugs_mlp256_tail_fwd wrote them; G[0..3]: the outputs G0 .. G3 [M,256] bf16 (the G
 * operands of the weight-gradient GEMMs).  Wn / bits / G are HOST arrays of device pointers; M a multiple of 256, dtype 1 (bf16).
 * Replaces hugs_rank1_mask + three masked hugs_gemm_nt_bits launches. */
int hugs_mlp256_tail_bwd(int dtype, int M, int nl, const float* d_raw, const float* wd, const void* const* Wn,
                         const uint32_t* const* bits, void* const* G, void* stream);
/* models.py:456 raw_density = Dense(1)(x)[...,0]; :467 density = softplus(raw + density_bias) */
int hugs_density_fwd(int dtype, int M, int K, const void* Y, int ldy, const float* w, const float* b,
                     float density_bias, float* raw, float* density, void* stream);
long long hugs_density_bwd_ws_bytes(int K);
/* backward of the two lines above: d_raw = d_density * sigmoid(raw + density_bias); dw = Y^T d_raw, db = sum d_raw.
 * d_density == NULL: d_raw is an input (an earlier call made it) and only the weight gradient is computed; dw == NULL: only
 * d_raw -- the two halves may run on different streams. */
int hugs_density_bwd(int dtype, int M, int K, const void* Y, int ldy, const float* d_density, const float* raw,
                     float density_bias, float* d_raw, float* dw, float* db, void* ws, void* stream);
/* out[m,n] = r[m]*c[n]*(Y[m,n] > 0): gradient entering the last trunk layer when there is no colour branch */
int hugs_rank1_mask(int dtype, int M, int N, const float* r, const float* c, const void* Y, int ldy, void* out, int ldo,
                    void* stream);
/* models.py:109-118 GloEmbed gather (zeros if zero_glo) */
int hugs_glo_gather(int nrays, int ng, const float* embedding, const int* embed_idx, int zero_glo, float* glo,
                    void* stream);
/* models.py:488-512: the per-ray constant part of the view layer, rb[ray] = b + [dir_enc|glo] * Wv[bottleneck:] */
int hugs_raybias_fwd(int nrays, int H, int nd, int ng, const float* dir_enc, const float* glo, const float* Wv_tail,
                     const float* bias, float* rb, void* stream);
/* (d_rb: workspace of hugs_raybias_bwd_ws_rows(nrays, nd, ng) rows of H floats: the per-ray sums + the ray chunks' partial products) */
long long hugs_raybias_bwd_ws_rows(int nrays, int nd, int ng);
int hugs_raybias_bwd(int dtype, int nrays, int S, int H, int nd, int ng, const void* G, int ldg, const float* dir_enc,
                     const float* glo, const float* Wv_tail, const int* embed_idx, float* d_rb, float* dWv_tail,
                     float* d_embedding, void* stream);
/* models.py:514-519 rgb = sigmoid(Dense(3)(h)) * (1 + 2 pad) - pad.  H: any multiple of 8 forward (128 / 256 with the weights in
 * registers); backward: any multiple of 128 (128: the Mip-NeRF 360 view layer; 256: nerfacto's colour MLP; wider -- the head on the
 * trunk when Model.use_viewdirs is False -- in column slabs).  G = (h > 0) * dz W^T, dW = h^T dz, db = sum dz. */
int hugs_rgb_fwd(int dtype, int M, int H, const void* Hact, int ldh, const float* W, const float* b, float pad,
                 float* rgb, void* stream);
long long hugs_rgb_bwd_ws_bytes(void);
int hugs_rgb_bwd(int dtype, int M, int H, const void* Hact, int ldh, const float* W, const float* rgb,
                 const float* d_rgb, float pad, void* G, int ldg, float* dW, float* db, void* ws, void* stream);
/* hugs_rgb_bwd with dW == NULL (H <= 256) leaves the weight gradient as per-workgroup partial sums in ws; this is its second half
 * (dW [H,3], db [3]), for a stream that is not the one the G consumer waits on. */
int hugs_rgb_bwd_reduce(int M, int H, float* dW, float* db, const void* ws, void* stream);

/* render.py:130-151 compute_alpha_weights + :185-244 volumetric_rendering.  extras (optional, [nrays,5]) =
 * {acc, distance_mean, distance_median, distance_percentile_5, distance_percentile_95}. */
int hugs_composite_fwd(int nrays, int S, const float* density, const float* rgb_s, const float* tdist,
                       const float* dirs, int opaque_background, float bg, const float* t_far, float* weights,
                       float* rgb_out, float* extras, void* stream);
int hugs_composite_bwd(int nrays, int S, const float* density, const float* rgb_s, const float* tdist,
                       const float* dirs, int opaque_background, float bg, const float* d_rgb_out,
                       const float* d_w_extra, float* d_density, float* d_rgb_s, void* stream);
/* hugs_composite_bwd that also writes the softplus density head's pre-activation gradient d_raw = d_density * sigmoid(raw + density_bias)
 * (models.py:461 density = softplus(raw + bias): what hugs_density_bwd computes first when handed d_density) */
int hugs_composite_bwd_raw(int nrays, int S, const float* density, const float* rgb_s, const float* tdist, const float* dirs,
                           int opaque_background, float bg, const float* d_rgb_out, const float* d_w_extra, float* d_density,
                           float* d_rgb_s, const float* raw, float density_bias, float* d_raw, void* stream);

/* train_utils.py:72-111 compute_data_loss (mode 0 lossmult, 1 static mask, 2 robust mask) value + gradient */
int hugs_data_loss(int N, int L, const float* pred, const float* gt, const float* lm_src, int mode,
                   float transient_weight, int charb, float charb_pad, const float* coef, float* d_pred,
                   float* out_stats, void* stream);
/* train_utils.py:251-348 robustnerf_mask (patch_size 16; -5 otherwise) */
int hugs_robust_mask(int npatch, int P, const float* pred, const float* gt, const float* inlier_threshold,
                     float quantile, int filter_size, float smoothed_q, int inner_patch, float inner_q, float* mask,
                     float* err_ws, float* stats_part_ws, float* stats, void* stream);
/* nerfacto/utils/loss_utils.py:88-150 get_robustnerf_mask as nerfacto/models/nerfacto.py:492-527 calls it: the same mask on
 * squared residuals (errors = resid_sq). */
int hugs_nf_robust_mask(int npatch, int P, const float* pred, const float* gt, const float* inlier_threshold,
                     float quantile, int filter_size, float smoothed_q, int inner_patch, float inner_q, float* mask,
                     float* err_ws, float* stats_part_ws, float* stats, void* stream);
/* Fused proposal network (nerfacto/models/nerfacto.py:927-990 HashMLPDensityField with Linear layers; the reference's
 * tcnn FullyFusedMLP form): density = trunc_exp(relu(X W0 + b0) w1 + b1) * selector for in_dim <= 32, hidden <= 64, one
 * thread per sample, fp32 weights read from the masters (W0 [in_dim, ldw0], w1 = column 0 of a [hidden, ldw1] matrix).
 * X [M, ldx] in `dtype` (ldx >= 16 / 32, multiple of 8; columns >= in_dim ignored).  fwd writes raw [M] and density [M];
 * bwd takes d_density and writes dX [M, ldx] and (=, deterministic slab reduction) the four weight-gradient leaves; ws:
 * hugs_nf_prop_ws_bytes(in_dim). */
long long hugs_nf_prop_ws_bytes(int in_dim);
int hugs_nf_prop_fwd(long long M, int in_dim, int hidden, int dtype, const void* X, int ldx, const float* W0, int ldw0,
                     const float* b0, const float* w1, int ldw1, const float* b1, const float* sel, float* raw,
                     float* density, int density_act, float density_bias, void* stream);
int hugs_nf_prop_bwd(long long M, int in_dim, int hidden, int dtype, const void* X, int ldx, const float* W0, int ldw0,
                     const float* b0, const float* w1, int ldw1, const float* raw, const float* sel,
                     const float* d_density, void* dX, float* gW0, float* gb0, float* gw1, float* gb1, void* ws,
                     int dx_f32 /* dX is a float [M, ldx] buffer (16-bit rows of <= 16 features only: the matrix-core form) */,
                     int density_act, float density_bias, void* stream);
/* train_utils.py:228-239 interlevel_loss -> stepfun.py:30-86 (per-ray loss + d/d w_env) */
int hugs_interlevel(int nrays, int S, int Sp, const float* t, const float* w, const float* t_env, const float* w_env,
                    float scale, float* loss_ray, float* d_w_env, void* stream);
/* train_utils.py:242-248 distortion_loss -> stepfun.py:266-276 */
int hugs_distortion(int nrays, int S, const float* t, const float* w, float scale, float* loss_ray, float* d_w,
                    void* stream);
int hugs_sum(int n, const float* x, float scale, float* out, void* stream);
int hugs_add_inplace(long long n, const float* src, float* dst, void* stream);
/* dst += alpha * src: the gradient of the weight-decay term m * ||theta_group||^2 (train_utils.py:444-447) */
int hugs_axpy(long long n, float alpha, const float* src, float* dst, void* stream);
/* Element-wise steps of the option branches (round 5: they were torch ops).
 *   hugs_noise_softplus: models.py:458-460,467 raw[0..n_noise) += noise_scale * noise (noise null: nothing added); density = softplus(raw + bias)
 *   hugs_axpy_op:        models.py:478-481 y (compute dtype 0 fp32 / 1 bf16 / 2 half) += a * x (fp32) -- bottleneck noise
 *   hugs_add_op:         dst (compute dtype) += src (compute dtype)
 *   hugs_affine:         dst = a * src + b (fp32; src may be dst) -- rgb_premultiplier / rgb_bias folded into the rgb head (models.py:514-516)
 *   hugs_bg_blend_fwd:   render.py:219-221 with a per-ray, per-channel background (models.py:256-261): bgw = max(0, 1 - sum_s w), rgb_out += bgw * bg
 *   hugs_bg_blend_bwd:   d_w_total[ray, s] = (d_w_extra or 0) - (bg . d_rgb_out)[ray] where bgw > 0 */
int hugs_noise_softplus(long long n, long long n_noise, float* raw, const float* noise, float noise_scale, float density_bias,
                        float* density, void* stream);
int hugs_axpy_op(int dtype, long long n, float a, const float* x, void* y, void* stream);
int hugs_add_op(int dtype, long long n, const void* src, void* dst, void* stream);
int hugs_affine(long long n, const float* src, float a, float b, float* dst, void* stream);
int hugs_bg_blend_fwd(int N, int S, const float* w, const float* bg_rgb, float* rgb_out, float* bgw, void* stream);
int hugs_bg_blend_bwd(int N, int S, const float* d_rgb_out, const float* bg_rgb, const float* bgw, const float* d_w_extra,
                      float* d_w_total, void* stream);

/* train_utils.py:442 weight_l2s, :461-462 grad norms/maxes, :351-369 clip_gradients, :466 nan_to_num, :468
 * optax.adam (:487-512), :470-473 update norms/maxes on the flat fp32 parameter buffer.
 * chunks: int4 {offset, len, leaf, module}[nchunks]; leaf_info: int4 {chunk_begin, chunk_end, module, 0}[nleaf]. */
int hugs_opt_stats(int nchunks, int nleaf, int nmod, const void* chunks, const void* leaf_info, const float* theta,
                   const float* grad, float gscale, float max_val, float max_norm, float* part1_ws, float* leaf_stats,
                   float* mod_scale, void* stream);
int hugs_opt_adam(int nchunks, int nleaf, const void* chunks, const void* leaf_info, float* theta, const float* grad,
                  float* m, float* v, const float* mod_scale, const int* trainable, float gscale, float max_val,
                  float lr, float b1, float b2, float eps, float bias_corr1, float bias_corr2, float* part2_ws,
                  float* leaf_upd, void* stream);
/* hugs_opt_adam with {lr, bias_corr1, bias_corr2} read from 3 device floats, and the launch that writes up to 4 such per-step
 * scalars from its kernel arguments: together they let the train step be replayed as a captured hipGraph. */
int hugs_opt_adam_dyn(int nchunks, int nleaf, const void* chunks, const void* leaf_info, float* theta, const float* grad,
                      float* m, float* v, const float* mod_scale, const int* trainable, float gscale, float max_val,
                      const float* dyn, float b1, float b2, float eps, float* part2_ws, float* leaf_upd, void* stream);
int hugs_set_floats(float* dst, int n, float a, float b, float c, float d, void* stream);
/* hugs_opt_adam_dyn whose last launch also publishes what the captured step hands back (train_utils.py:475-477 returns new_state,
 * stats, rng): packed[0..ntail) = tail * tscale (the pmean'ed stat tail), thr_dst[l] = packed[thr_off + l * thr_stride] (train.py:145-148's
 * inlier-threshold feedback, on the device), the packed stats to the pinned host slot ptrs[0] and the advanced key to key_dst and to the
 * device buffer ptrs[1] -- ptrs is a DEVICE table of two addresses that the step's staging launch (hugs_stage_step_pub) fills, since a
 * captured launch's own arguments cannot change.  pub: 12 host words {tail, packed, ntail, npacked, tscale (float bits), thr_dst (or 0),
 * thr_off, thr_stride, thr_n, key_src (or 0), key_dst (or 0), ptrs (or 0)}. */
int hugs_opt_adam_pub(int nchunks, int nleaf, const void* chunks, const void* leaf_info, float* theta, const float* grad,
                      float* m, float* v, const float* mod_scale, const int* trainable, float gscale, float max_val,
                      const float* dyn, float b1, float b2, float eps, float* part2_ws, float* leaf_upd,
                      const unsigned long long* pub, void* stream);
/* One launch in front of a replayed (captured) train step (train_utils.py:386-477's inputs: rays, batch.rgb, rng; train_frac through
 * the scalars): item i copies words[i] 4-byte words src[i] -> dst[i] (n <= 16; HOST arrays of device pointers), dst_f[0..nf) =
 * {a, b, c, d}[0..nf) as hugs_set_floats. */
int hugs_stage_step(int n, const void* const* src, void* const* dst, const int* words, float* dst_f, int nf, float a, float b, float c,
                    float d, void* stream);
/* The same launch with up to 8 scalars (scalars: HOST array of nf floats), which also writes dst_p[0..1] = {p0, p1} (device table of two
 * addresses read back by hugs_opt_adam_pub's last launch; 0 = nothing to publish there). */
int hugs_stage_step_pub(int n, const void* const* src, void* const* dst, const int* words, float* dst_f, int nf, const float* scalars,
                        void* dst_p, void* p0, void* p1, void* stream);
/* fp32 master [K,N] -> compute-dtype copies Wn [K,N] and Wt [N,K] (either may be NULL) */
int hugs_cast_weights(int dtype, int K, int N, const float* W, void* Wn, void* Wt, void* stream);
/* The same cast for a device table of matrices in one launch.  items: nitems records of 40 bytes
 * {const float* W; void* Wn; void* Wt; int32 K, N, blk0, nbx} with nbx = ceil(N/32), blk0 = first 32x32 block of the
 * item in the launch grid (ascending), total_blocks = sum of ceil(K/32)*nbx. */
int hugs_cast_weights_batch(int dtype, int nitems, const void* items, int total_blocks, void* stream);

/* ---- batch assembly on the device (the reference does this in a host numpy thread) ----
 * camera_utils.py:503-607 pixels_to_rays (+ :462-495 Newton undistort, :561-570 fisheye, :32-100 convert_to_ndc)
 * and the pix_coords of :649-652 cast_ray_batch.  pixtocams [ncams,3,3], camtoworlds [ncams,3,4] fp32 (ncams == 1:
 * shared, cam_idx may be NULL); dist NULL or {k1,k2,k3,k4,p1,p2}[1 or ncams] (dist_per_cam); pixtocam_ndc NULL or
 * [3,3]; camtype 0 perspective / 1 fisheye (-2 otherwise); widths/heights int32 [ncams], needed only for
 * pix_coords (may be NULL with it).  Outputs [n,3],[n,3],[n,3],[n,1],[n,2] fp32. */
int hugs_pixels_to_rays(int n, const int32_t* pix_x, const int32_t* pix_y, const int32_t* cam_idx, int ncams,
                        const float* pixtocams, const float* camtoworlds, const float* dist, int dist_per_cam,
                        const float* pixtocam_ndc, int camtype, const int32_t* widths, const int32_t* heights,
                        float* origins, float* directions, float* viewdirs, float* radii, float* pix_coords,
                        void* stream);
/* datasets.py:474-491 `arr[cam_idx][pix_y, pix_x]` for images / static masks / near / far kept resident in HBM as
 * one flat buffer: dst[i,:] = src[(offsets[cam]+y*widths[cam]+x)*channels : +channels]; per_pixel == 0 reads
 * src[cam,:] (one value per image).  src_u8: bytes, divided by 255 in binary32 as the loaders do. */
int hugs_gather_pixels(int n, int channels, const int32_t* pix_x, const int32_t* pix_y, const int32_t* cam_idx,
                       const int64_t* offsets, const int32_t* widths, int per_pixel, int src_u8, const void* src,
                       float* dst, void* stream);
/* datasets.py:498-524 patch origins -> pixel coordinates: origin + (dx,dy)*dilation, row-major inside a patch */
int hugs_expand_patches(int npatch, int patch_size, int dilation, const int32_t* org_x, const int32_t* org_y,
                        const int32_t* cam_of_patch, int32_t* pix_x, int32_t* pix_y, int32_t* cam_idx, void* stream);

/* ---- JAX-compatible PRNG: Threefry-2x32-20 with jax's counter layout (jax/_src/prng.py
 * threefry_2x32 on iota(n): first half of the padded counters -> word 0, second half -> word 1).  key: 2 uint32 on
 * the device.  hugs_prng_bits == jax.random.bits(key, (n,)); split(key, m) is hugs_prng_bits(key, 2m) viewed [m,2]
 * (train_utils.py:408, models.py:38-43).  hugs_prng_uniform == jax.random.uniform(key, (n,), float32, minval,
 * maxval) (stepfun.py:207-209).  n < 2^32. */
int hugs_prng_bits(const uint32_t* key, long long n, uint32_t* out, void* stream);
int hugs_prng_uniform(const uint32_t* key, long long n, float minval, float maxval, float* out, void* stream);
/* jax.random.normal(key, [n]): sqrt(2) erf_inv(uniform(key, minval = nextafter(-1, 0), maxval = 1)) with XLA's f32 erf_inv
 * polynomial (models.py:458-460,478-481 noise draws; the draws of flax's initialisers). */
int hugs_prng_normal(const uint32_t* key, long long n, float* out, void* stream);
/* One training step's consumption of the jax.random stream in ONE launch: key_out = first key of split(key_in)
 * (train_utils.py:408 `rng, key = random.split(rng)`); with the second key, per level l < L (<= 8): models.py:196 split ->
 * stepfun.py:207-209 random.uniform(key, [n[l]], maxval = maxval[l]) into out[l], models.py:230 split.  n, maxval and out
 * are HOST arrays (out[l] device pointers).  Bit-identical to the chain of hugs_prng_bits / hugs_prng_uniform calls. */
int hugs_prng_step_jitter(const uint32_t* key_in, int L, const long long* n, const float* maxval, float* const* out,
                          uint32_t* key_out, void* stream);
/* jax.random.fold_in(key, data) (threefry_2x32(key, (0, data))): how flax derives a parameter's initialiser key from the
 * `params` rng -- the hash of the module path and the per-scope counter folded in (models.py:333-357 `model.init(rng, ...)`) */
int hugs_prng_fold_in(const uint32_t* key, uint32_t data, uint32_t* key_out, void* stream);
/* The largest L hugs_prng_step_jitter takes (the host side gates its fused launch on this, not on a literal). */
int hugs_prng_step_jitter_max_levels(void);

/* ---- eval metrics: image.py:127-141 MetricHarness.  hugs_ssim == dm_pix.ssim(a, b) for one
 * [H,W,C] fp32 image pair with dm_pix's defaults passed explicitly (max_val 1, 11 taps fixed, filter_sigma 1.5,
 * k1 .01, k2 .03): mean of the 'valid' SSIM map.  ws: hugs_ssim_ws_bytes(H,W,C) bytes.  H, W >= 11 (-2 otherwise).
 * hugs_mse: mean squared error over n floats (image.py:135), ws 1024 floats. */
long long hugs_ssim_ws_bytes(int H, int W, int C);
int hugs_ssim(int H, int W, int C, const float* a, const float* b, float max_val, float filter_sigma, float k1, float k2,
              float* ws, float* out, void* stream);
int hugs_mse(long long n, const float* a, const float* b, float* ws, float* out, void* stream);

/* ---- HA-NeRF branch.  models.py:651-674 ImplicitMask: X = [pos_enc(pix_coords,0,deg,identity) |
 * tra_vec | 0-pad] [N,kpad] in the compute dtype (tra_vec NULL = zero_tra); its Dense+relu layers run on
 * hugs_gemm_nt / hugs_gemm_tn; head mask = sigmoid(X w + b) and its backward (d_raw [Npad] with zero tail rows,
 * dW [W], db [1]; G = hugs_rank1_mask(d_raw, w, X)).  hugs_embed_scatter_add: TransientEmbed gradient,
 * d_embedding[embed_idx[n], :] += dX[n, col0:col0+T] (caller zeroes d_embedding once per step).
 * hugs_hanerf_loss: train_utils.py:186-225; out_stats [2L+2] = {mean resid^2, mean((1-m) loss)} per level,
 * mean(m^2), mean(m); d_pred [L,N,3], d_mask [N] include the level multipliers coef[L] and mask_size_mult. */
int hugs_mask_input_fwd(int N, int T, int deg, const float* pix_coords, const float* tra_vec, int kpad, int dtype, void* X,
                        void* stream);
int hugs_mask_head_fwd(int dtype, int N, int W, const void* X, int ldx, const float* w, const float* b, float* mask,
                       void* stream);
int hugs_mask_head_bwd(int dtype, int N, int Npad, int W, const void* X, int ldx, const float* mask, const float* d_mask,
                       float* d_raw, float* dW, float* db, void* stream);
int hugs_embed_scatter_add(int dtype, int N, int T, const void* dX, int ldx, int col0, const int* embed_idx,
                           float* d_embedding, void* stream);
int hugs_hanerf_loss(int N, int L, const float* pred, const float* gt, const float* mask, int charb, float charb_pad,
                     const float* coef, float mask_size_mult, float* d_pred, float* d_mask, float* out_stats, void* stream);
/* mask_size_mult (train_utils.py:190-193) read from one device float: the form a captured (hipGraph) train step uses */
int hugs_hanerf_loss_dyn(int N, int L, const float* pred, const float* gt, const float* mask, int charb, float charb_pad,
                         const float* coef, const float* mask_size_mult_dev, float* d_pred, float* d_mask, float* out_stats, void* stream);

/* ---- NeRF-W branch.  render.py:154-182 compute_dual_alpha_weights + :246-273
 * volumetric_rendering_combined_color + models.py:299-307 uncertainty (beta = sum_i w^t_i u_i + beta_min, w^t from
 * the transient density alone).  All per-sample arrays [nrays*S(,3)] fp32.  The backward ADDS into d_dens_s (the
 * static-only hugs_composite_bwd runs first: interlevel / distortion reach sigma_s through `weights`), overwrites
 * the others; dens_t_const = d(density regulariser)/d sigma_t.  hugs_rank1_add2_mask: G += (r1 c1^T + r2 c2^T) *
 * (X > 0), the two scalar heads' contribution to the gradient entering the transient trunk.
 * hugs_nerfw_loss: train_utils.py:150-183 (pred[L-1] = rgb_combined); out_stats [2L+1]. */
int hugs_dual_composite_fwd(int nrays, int S, const float* dens_s, const float* dens_t, const float* rgb_s,
                            const float* rgb_t, const float* unc, const float* tdist, const float* dirs,
                            int opaque_background, float bg, float beta_min, float* rgb_combined, float* rgb_static,
                            float* rgb_transient, float* beta, void* stream);
int hugs_dual_composite_bwd(int nrays, int S, const float* dens_s, const float* dens_t, const float* rgb_s,
                            const float* rgb_t, const float* unc, const float* tdist, const float* dirs,
                            int opaque_background, float bg, const float* d_rgb_combined, const float* d_beta,
                            float dens_t_const, float* d_dens_s_accum, float* d_rgb_s, float* d_dens_t, float* d_rgb_t,
                            float* d_unc, void* stream);
int hugs_rank1_add2_mask(int dtype, int M, int N, const float* r1, const float* c1, const float* r2, const float* c2,
                         const void* X, int ldx, void* G, int ldg, void* stream);
int hugs_nerfw_loss(int N, int L, const float* pred, const float* gt, const float* beta, int charb, float charb_pad,
                    const float* coef, float beta_mult, float* d_pred, float* d_beta, float* out_stats, void* stream);

/* ---- nerfacto encodings (groundwork; PARITY UNPINNED: tiny-cuda-nn is not available, the
 * algorithm is restated in oracle/hashgrid_ref.py).  nerfacto/models/nerfacto.py:714-733,761-770,921-947 HashGrid:
 * x01 [n,3] in [0,1]; table fp32 [level_offsets[L], features]; level tables are HOST arrays (offsets [L+1] in entries,
 * resolutions [L], scales [L]); out [n, row_pitch] (first L*features columns written) in the dtype code given (0 fp32, 1 bf16, 2 fp16; the
 * `*_bf16` arguments below are that code).  The backward
 * ADDS into d_table (fp32 atomics; runs of consecutive samples inside one cell are summed in the wavefront first).
 * nerfacto.py:693-700 SphericalHarmonics degree 4: 16 columns from col0. */
int hugs_hashgrid_fwd(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                      const float* level_scales, const float* x01, const float* table, int out_bf16, int row_pitch,
                      void* out, void* stream);
int hugs_hashgrid_bwd(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                      const float* level_scales, const float* x01, const void* d_out, int d_out_bf16, int row_pitch,
                      float* d_table_accum, void* stream);
/* hugs_hashgrid_bwd with a caller-owned device workspace of hugs_hashgrid_bwd_ws_bytes(n, n_levels, features) bytes (0 when the form
 * does not apply: features != 2): the levels behind the LDS-resident coarse ones are binned by table slot and summed per bin in LDS
 * (count -> scan -> scatter into per-bin record runs -> one workgroup per bin), then added to the table with plain coalesced
 * read-modify-writes, instead of one L2 float atomic per corner and feature -- the scatter sits at the chip's atomic-transaction rate
 * (nerfacto.py:693-733 gets the same gradient from tiny-cuda-nn's atomics).  Same result up to fp32 summation order.  Falls back to the
 * scatter when ws is NULL / too small, n < 65536, or a level has more than 2^20 entries. */
long long hugs_hashgrid_bwd_ws_bytes(int n, int n_levels, int features);
int hugs_hashgrid_bwd_ws(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                         const float* level_scales, const float* x01, const void* d_out, int d_out_dtype, int row_pitch,
                         float* d_table_accum, void* ws, long long ws_bytes, void* stream);
int hugs_sh4_fwd(int n, const float* dirs01, int out_bf16, int row_pitch, int col0, void* out, void* stream);
/* hugs_hashgrid_fwd with the table given as an IEEE-half copy (table_dtype 2; 0 = fp32): the fp16 mode, the reference's
 * `enable_amp: True` (nerfacto/configs/phototourism_nerfacto_base.yml:3; tiny-cuda-nn evaluates the grid on half
 * parameters, nerfacto.py:699,770 dtype=None).  Interpolation accumulates in fp32; out_dtype 0 fp32 / 1 bf16 / 2 fp16. */
int hugs_hashgrid_fwd_t(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                        const float* level_scales, const float* x01, const void* table, int table_dtype, int out_dtype,
                        int row_pitch, void* out, void* stream);
/* nerfacto.py:1036-1047,1080-1091 (HA-NeRF ImplicitMask of the nerfacto model): 2-D hash grid of per-RAY image coordinates
 * x01 [n,2] (resolution^2 dense entries / two-prime hash, bilinear).  The forward writes the mask MLP's whole input row
 * out[n, :row_pitch] = [grid (n_levels*features) | extra[n, :T] (the ray's transient embedding row) | zeros]; the backward
 * ADDS the table gradient (fp32 atomics) from the first n_levels*features columns of d_out. */
int hugs_hashgrid2d_fwd(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                        const float* level_scales, const float* x01, const float* table, const float* extra, int T,
                        int out_bf16, int row_pitch, void* out, void* stream);
int hugs_hashgrid2d_bwd(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                        const float* level_scales, const float* x01, const void* d_out, int d_out_bf16, int row_pitch,
                        float* d_table_accum, void* stream);

/* hugs_gemm_nt with 1-bit relu masks in the 256x256 kernels' own register layout (bf16; M, N multiples of 256, ldc == N,
 * K a multiple of 64 and >= 128): a relu epilogue writes bits_out (hugs_gemm_nt_bits_bytes(M, N) = M*N/8 bytes: per
 * tile, wave and lane one 16-byte word), the backward GEMM that produces the same [M, N] shape multiplies its output by
 * bits_in instead of re-reading the bf16 activation (models.py:451-456 relu; its autodiff, train_utils.py:454). */
long long hugs_gemm_nt_bits_bytes(int M, int N);
int hugs_gemm_nt_bits(int dtype, int M, int N, int K1, int K2, const void* A1, int lda1, const void* A2, int lda2,
                      const void* Bt, int ldb, const float* bias, int relu, const float* r1_row, const float* r1_col,
                      void* out, int ldc, uint32_t* bits_out, const uint32_t* bits_in, void* stream);
/* The trunk of an MLP (models.py:432-461: for i < net_depth: x = relu(Dense(x)); skip concat at i % skip_layer == 0) as ONE launch:
 * nl layers  out_l[M, N] = relu([A1_l | A2_l] Bt_l^T + bias_l)  with their 1-bit relu masks, layer l + 1 reading layer l's output
 * (the caller names it as that layer's A1).  tab: 12 HOST words per layer {A1, A2, Bt, bias, out, bits (device pointers), lda1, lda2,
 * ldb, K1, K2, 0}; flags: nl * M / 256 device words of scratch (zeroed by the call, on the stream).  Results are bit-identical to nl
 * hugs_gemm_nt_bits calls.  bf16 (dtype 1); M, N multiples of 256; nl * N <= 8192; every K a multiple of 64 and >= 128; every
 * leading dimension a multiple of 512; a whole number >= 4 of 256 x 256 tiles per CU.  Returns -3 when the shape does not qualify. */
int hugs_gemm_nt_chain(int dtype, int M, int N, int nl, const unsigned long long* tab, unsigned* flags, void* stream);

/* ---- nerfacto path (reference /root/reference/nerfacto).  One wavefront per ray in the per-ray
 * kernels (<= 1024 bins / samples).  Matrices [M, ld] row-major in `dtype` (0 fp32, 1 bf16).
 * hugs_nf_sample: utils/ray_utils.py:112-231 sample + sample_intervals (softmax of anneal*log(w+padding) with -inf on
 *   zero-width bins, all -inf -> uniform; cdf = [0, cumsum(pdf[:-1]).clamp_max(1), 1]; searchsorted right; midpoints,
 *   clamped end posts) and models/nerfacto.py:231-248 s_to_t (spacing 0 uniform / 1 piecewise / 2 reciprocal).
 *   u = u_base[j] + jitter[ray*jitter_stride (+j)]; jitter may be NULL (perturb=False).  -2 if ns <= 1.
 * hugs_nf_positions: nerfacto.py:326-328 o + d * t_mid, :822-829 (x+bound)/(2 bound) or contraction (custom_functions.py
 *   :17-24) then (x+2)/4; selector; positions outside [0,1]^3 zeroed.
 * hugs_nf_weights_fwd/bwd: ray_utils.py:234-257 density_to_weight -- deltas are (edge[i+1] - edge[0]) |d|, the
 *   reference's own arithmetic -- + :300-314 render_features (bg * max(1 - acc, 0)) + :340-347 depth (unclipped: the
 *   reference clips to the batch's largest step).  bwd: dL/d density [M], dL/d rgb_s [M,3] from dL/d rgb_out [nrays,3]
 *   and an extra dL/d weights [nrays,S]; any of rgb_s / bg / d_rgb_out / d_w_extra may be NULL.
 * hugs_nf_interlevel: utils/loss_utils.py:7-62 lossfun_outer (searchsorted-right `outer`, EPS 1e-7): loss_ray[nrays] =
 *   sum_i max(w - w_outer, 0)^2 / (w + 1e-7), d_w_env = scale * d loss / d w_env.
 * hugs_nf_density_act / base_grad / head_input / app_bwd / rgb_act / rgb_grad: element-wise glue of the fields around
 *   the padded GEMMs (nerfacto.py:818-876, :971-988; trunc_exp custom_functions.py:38-52).
 * hugs_nf_adam: torch.optim.Adam step (nerfacto/train.py:183) on a flat buffer; bc1 = 1 - b1^t, bc2 = 1 - b2^t. */
int hugs_nf_sample(int nrays, int nb, int ns, const float* bins, const float* weights, float anneal, float padding,
                   const float* u_base, const float* jitter, int jitter_stride, float lo, float hi, int spacing,
                   const float* near, const float* far, float* sbins, float* ebins, void* stream);
int hugs_nf_positions(int nrays, int S, const float* ebins, const float* origins, const float* dirs, int contract,
                      float bound, float* x01, float* sel, void* stream);
int hugs_nf_weights_fwd(int nrays, int S, const float* density, const float* ebins, const float* dirs, int opaque,
                        const float* rgb_s, const float* bg, float* weights, float* rgb_out, float* acc, float* depth,
                        void* stream);
int hugs_nf_weights_bwd(int nrays, int S, const float* density, const float* ebins, const float* dirs, int opaque,
                        const float* rgb_s, const float* bg, const float* weights, const float* d_rgb_out,
                        const float* d_w_extra, float* d_density, float* d_rgb_s, void* stream);
int hugs_nf_interlevel(int nrays, int S, int Sp, const float* t, const float* w, const float* t_env, const float* w_env,
                       float scale, float* loss_ray, float* d_w_env, void* stream);
/* density_act / density_bias (here and in hugs_nf_prop_*, hugs_nf_field_*): models/nerfacto.py:36,702-710,910-918 density_activation --
 * 0 = trunc_exp (custom_functions.py:38-52), 1 = softplus(raw + density_bias), density_bias = -1 by the fields' constructors. */
int hugs_nf_density_act(long long M, int dtype, const void* Y, int ldy, int col, const float* sel, float* density, int density_act,
                        float density_bias, void* stream);
int hugs_nf_base_grad(long long M, int dtype, const void* Y, int ldy, const float* sel, const float* d_density,
                      const void* dXh, int ldx, int geo_col0, int ngeo, void* G, int ldg, int density_act, float density_bias,
                      void* stream);
int hugs_nf_head_input(long long M, int S, int dtype, const float* sh, const void* Yb, int ldy, int ngeo, const float* app,
                       int napp, void* X, int ldx, void* stream);
int hugs_nf_app_bwd(int nrays, int S, int dtype, const void* dX, int ldx, int col0, int napp, const int* embed_idx,
                    float* d_embedding, void* stream);
int hugs_nf_rgb_act(long long M, int dtype, const void* Y, int ldy, float rgb_bias, float* rgb, void* stream);
int hugs_nf_rgb_grad(long long M, int dtype, const float* rgb, const float* d_rgb, void* G, int ldg, void* stream);
/* Fused nerfacto field forward (models/nerfacto.py:693-759 base network + colour network; csrc/hugs_fieldfuse.hip), the shape class
 * of phototourism_nerfacto_base.yml: <= 32 hash features -> 256 relu -> 1 + ngeo; [SH16 | ngeo | napp] (<= 128) -> 256 relu -> 256
 * relu -> 3 sigmoid; tmpl = hugs_nf_head_template's per-ray rows.  16-bit [n][k] weight copies: W0t [256][ldw0] (columns 0..31
 * read); W1x [128][256] = the base network's second layer in HEAD-INPUT row order (row 0 = raw density, rows 16 .. 16 + ngeo = the
 * geo features, the rest zero) with b1x [128] ordered alike; C0t [256][128]; C1t [256][256]; c2 [256,3] fp32, cb2 3 floats
 * (rgb_bias already added).  Outputs: Y0, H0, H1 [M,256], raw [M] (16-bit raw density), Xh [M,128] (the head input), relu mask bits
 * of Y0 / H0 in the hugs_gemm_nt_bits layout (or null), density = exp(raw) * sel, rgb [M,3].  M: a multiple of 256, whole rays of
 * S samples; ngeo a multiple of 4. */
int hugs_nf_field_fwd(int dtype, long long M, int S, const void* X0, int ldx0, const void* W0t, int ldw0, const void* W1x,
                      const void* C0t, const void* C1t, const float* b0, const float* b1x, const float* cb0, const float* cb1,
                      const float* c2, const float* cb2, const void* tmpl, int ngeo, const float* sel, void* Y0, void* raw, void* Xh,
                      void* H0, void* H1, uint32_t* bY0, uint32_t* bH0, float* density, float* rgb, int density_act,
                      float density_bias, void* stream);
/* Backward of the two networks above from colour layer 1's pre-activation gradient G1 [M,256] (hugs_rgb_bwd writes it) down to the
 * hash-feature gradient, one launch: G0 = (G1 c1^T) relu'(H0), dXh = G0 c0^T, appearance columns summed per ray into d_embedding
 * (+=, float atomics; null: skipped), Gb = [d_raw | 0 | d geo | 0] in HEAD-INPUT column order (d_raw = d_density exp(clamp(raw,
 * +-15)) sel), Gy0 = (Gb W1x) relu'(Y0), dX0 = Gy0 w0^T (columns 0..31 of the pitch-ldx0 rows).  Weights as 16-bit [k_out][n]
 * copies: C1n [256][256], C0n [128][256], W1xn [256][128] (= W1x transposed), W0n [>= 32][256]; bH0 / bY0 the forward's mask bits;
 * raw its 16-bit raw density.  G0, Gb, Gy0 are the G operands of the weight-gradient GEMMs.  M a multiple of 256, S of 64, ngeo
 * of 8, napp of 4 (<= 64). */
int hugs_nf_field_bwd(int dtype, long long M, int S, const void* G1, const void* C1n, const void* C0n, const void* W1xn,
                      const void* W0n, const uint32_t* bH0, const uint32_t* bY0, const float* d_density, const float* sel,
                      const void* raw, int ngeo, int napp, const int* embed_idx, void* G0, void* Gb, void* Gy0, void* dX0, int ldx0,
                      float* d_embedding, int dx_f32 /* dX0 is a float [M, ldx0] buffer instead of a 16-bit one */, int density_act,
                      float density_bias, void* stream);
/* per-ray head-input template of the kernel above: out[ray, 0..127] = [SH16 | 0 x ngeo | appearance | 0 ..] (16-bit) */
int hugs_nf_head_template(int dtype, int nrays, const float* sh, const float* app, int ngeo, int napp, void* out, void* stream);
int hugs_nf_adam(long long n, float* theta, const float* grad, float* m, float* v, float lr, float b1, float b2, float eps,
                 float bc1, float bc2, void* stream);
/* Dynamic loss scaling of the fp16 mode = torch.cuda.amp.GradScaler as nerfacto/train.py:168,210-213 drives it
 * (scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()), with found_inf kept on the device.
 * state: 3 floats {scale, growth tracker, found_inf}; counts: per parameter group, the number of Adam updates it has taken
 * (torch keeps `step` per parameter; a skipped step does not advance it); bc: 2 floats per group {1 - b1^k, 1 - b2^k}.
 *   hugs_amp_check(n, grad, state)      state[2] = 1 if any of grad[0..n) is inf / nan        (unscale_'s inf check)
 *   hugs_amp_prepare(G, counts, b1, b2, bc)   bias corrections of each group's NEXT update (double arithmetic)
 *   hugs_nf_adam_amp(...)               Adam on grad / state[0]; a no-op when state[2] != 0   (scaler.step)
 *   hugs_amp_update(state, counts, mask, growth, backoff, interval)   counts[g] += 1 for g in mask unless skipped; scale *=
 *       backoff on overflow, *= growth after `interval` clean steps in a row; found_inf cleared   (scaler.update) */
int hugs_amp_check(long long n, const float* grad, float* state, void* stream);
int hugs_amp_prepare(int ngroups, const float* counts, float b1, float b2, float* bc, void* stream);
int hugs_nf_adam_amp(long long n, float* theta, const float* grad, float* m, float* v, float lr, float b1, float b2, float eps,
                     const float* state, const float* bc, void* stream);
int hugs_amp_update(float* state, float* counts, unsigned group_mask, float growth, float backoff, float interval, void* stream);

/* test/bench forms of hugs_gemm_nt / hugs_gemm_tn with an explicit kernel selection (a call argument: no process
 * state): tile_mode 0 = the default choice (what the plain entry points use), 1 = force the 128x128-tile kernels,
 * 3 = the 256x128 NT kernel where a 256x256 one would be chosen, 5 = 256x256 without the persistent form */
int hugs_gemm_nt_tiles(int tile_mode, int dtype, int M, int N, int K1, int K2, const void* A1, int lda1, const void* A2, int lda2,
                       const void* Bt, int ldb, const float* bias, const float* row_bias, int row_div, int ld_rb, int relu,
                       const void* mask, int ld_mask, const float* r1_row, const float* r1_col, void* out, int ldc, void* stream);
int hugs_gemm_tn_tiles(int tile_mode, int dtype, int Mrows, int Kc, int N, int nsplit, const void* X, int ldx, const void* G, int ldg,
                       float* dW, float* dbias, void* ws, void* stream);
/* Dynamic tile queues of the persistent NT launches (no reference counterpart: the reference's Dense layers, models.py:451-456, are
 * XLA's).  Between hugs_gemm_nt_queue_begin and hugs_gemm_nt_queue_end every eligible hugs_gemm_nt / hugs_gemm_nt_bits launch (bf16 /
 * fp16, the 64-wide whole-line kernel's shapes, at least two 256 x 256 tiles per CU) draws its tiles from per-XCD ticket counters in
 * the next 32-byte slot of `region` instead of the static walk tile = workgroup + i x grid: a workgroup that gets its CU late (a
 * kernel of another stream held it) computes fewer tiles and the launch ends when the tiles do.  Results are bit-identical.  _begin
 * zeroes the region on `stream` (every launch that uses it must be ordered behind that point -- the train step calls it first thing, so
 * a captured step re-zeroes its slots on every replay); `bytes` a multiple of 32, one slot per launch, launches beyond the region
 * use the static walk.  Process-global sequence state; _end(0) closes the pair. */
int hugs_gemm_nt_queue_begin(void* region, long long bytes, void* stream);
int hugs_gemm_nt_queue_end(int reserved);
/* measurement hook of the persistent NT kernel (bench.py's measured roofline.per_cycle_frac; no reference counterpart: the reference's
 * Dense layers, models.py:451-456, are XLA's): buf = 64 x 4 x 2 device uint64 words, zeroed by the caller, or NULL (default: off).
 * While set, every persistent hugs_gemm_nt / hugs_gemm_nt_bits launch adds, per workgroup, {shader cycles from its first instruction
 * to the retirement of its last K stage, tiles walked} to record (epilogue specialisation * 4 + K class) -- K class 0: K <= 256,
 * 1: 512, 2: 1024, 3: other; specialisation = bit set {1 bias, 2 relu, 4 bf16 mask, 8 rank-1, 16 mask bits in, 32 mask bits out}.
 * Process-global (a __device__ pointer of the bf16 instantiation); not stream-ordered: set it while the device is idle. */
int hugs_debug_set_nt_cycles(void* buf);
/* test hooks: the portable exp/log of the sampler and raw IEEE ops as the device executes them */
int hugs_test_explog(const float* x, int n, float* y_exp, float* y_log, void* stream);
int hugs_test_arith(const float* a, const float* b, int n, float* out4n, void* stream);

#ifdef __cplusplus
}
#endif
#endif

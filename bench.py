"""Headline benchmark: training rays/sec of the Mip-NeRF 360 hot path (BASELINE.json: 1024-ray batch,
64 proposal + 128 fine samples, NerfMLP 8x1024 + PropMLP 4x256, Kubric base gin) on N MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A step = forward + losses + backward + RCCL all-reduce + clip/Adam + weight re-cast on one batch of
synthetic rays already resident in HBM (data: synthetic).  Weak scaling: 1024 rays per GPU.
Prints ONE JSON line (rank 0) with `roofline` for the dominant kernel (bf16 NT GEMM of the NerfMLP trunk,
timed live with HIP events around its launches INSIDE extra train steps, on the stream they run on) and `cpu_baseline`
(the oracle = CPU restatement of the reference, timed on a bounded sample at N=1)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIN = ["Config.patch_size = 16", "Config.data_loss_type = 'mse'", "Config.distortion_loss_mult = 0.",
       "Config.near = 0.1", "Config.far = 1.2", "Model.opaque_background = True", "Model.num_levels = 2",
       "Model.num_prop_samples = 64", "Model.num_nerf_samples = 128", "PropMLP.net_depth = 4",
       "PropMLP.net_width = 256", "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 8", "NerfMLP.net_width = 1024"]
# BASELINE.json configs[2] / configs[3] as synthetic restatements; --config selects them (not bench lines)
GIN_CFG3 = GIN[:2] + ["Config.distortion_loss_mult = 0.001", "Config.transient_type = 'withmask'", "Config.data_loss_type = 'charb'",
                      "Model.num_glo_features = 48"] + GIN[5:]
GIN_CFG4 = GIN[:2] + ["Config.distortion_loss_mult = 0.001", "Config.transient_type = 'robustnerf'",
                      "Config.robustnerf_inlier_quantile = 0.8", "Model.raydist_fn = @jnp.reciprocal", "Model.num_glo_features = 4",
                      "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract"] + GIN[5:]
# the reference-default shape (MipNeRF360/configs/360.gin + the Config / Model defaults, internal/configs.py:50,
# models.py:50-52): L=3, S=(64,64,32), 16384 rays, contract + reciprocal spacing, charb, interlevel + distortion losses, GLO 0
GIN_REF360 = ["Config.near = 0.2", "Config.far = 1e6", "Model.raydist_fn = @jnp.reciprocal", "Model.opaque_background = True",
              "PropMLP.warp_fn = @coord.contract", "PropMLP.net_depth = 4", "PropMLP.net_width = 256", "PropMLP.disable_rgb = True",
              "NerfMLP.warp_fn = @coord.contract", "NerfMLP.net_depth = 8", "NerfMLP.net_width = 1024"]
FLOP_TRAIN_PER_RAY = 6.5036e9   # SURVEY 8d / BASELINE.md work model, cfg2
FLOP_TRAIN_PER_RAY_REF360 = 1.8160e9   # BASELINE.md section 2 work model, L=3 S=(64,64,32)
PEAK_BF16 = 2.5e15              # dense MFMA peak (MI355X_MICROARCH.md)


def synth_batch(n_patch, P, seed, device):
  from nerf_hugs_amd.internal import utils
  rng = np.random.default_rng(seed)
  shp = (n_patch, P, P)
  d = rng.normal(size=shp + (3,))
  d = d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.8, 1.2, shp + (1,))
  f = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(device)
  rays = utils.Rays(pix_coords=f(rng.uniform(size=shp + (2,))), origins=f(rng.normal(size=shp + (3,)) * 0.5),
                    directions=f(d), viewdirs=f(d / np.linalg.norm(d, axis=-1, keepdims=True)),
                    radii=f(rng.uniform(5e-4, 2e-3, shp + (1,))), lossmult=f(np.ones(shp + (1,))),
                    static_mask=f(np.ones(shp + (1,))), near=f(np.full(shp + (1,), 0.1)), far=f(np.full(shp + (1,), 1.2)),
                    embed_idx=f(np.zeros(shp + (1,)), np.int32), cam_idx=f(np.zeros(shp + (1,)), np.int32))
  return utils.Batch(rays=rays, rgb=f(rng.uniform(size=shp + (3,))))


def instep_roofline(train_step, state, next_batch, gen, thr, steps=5):
  """Dominant kernel, measured INSIDE the train steps: every GEMM launch of `steps` extra steps is bracketed by HIP
  events on the stream it is launched on (nerf_hugs_amd/_lib.py PROFILE hook; the side stream for the weight-gradient
  GEMMs).  `roofline` is the forward NerfMLP trunk layer [131072,1024]x[1024,1024]^T + bias + relu; the masked dX and
  the dW GEMM of the same shape are reported next to it.  `traffic` = HBM bytes per launch from the rocprofv3 PMC
  passes of the shipped kernels, committed as profiles/r06_gemm_traffic.json (r05_... / r04_... when absent; 2 x FETCH_SIZE + WRITE_SIZE,
  MI355X_MICROARCH.md HBM section; `traffic_source` names the file and the kernel's duration under the profiler next to
  the in-step one), null when that file is absent."""
  from nerf_hugs_amd import _lib
  # the persistent NT kernel's own cycle account (csrc/hugs_gemm.hip g_nt_cycles; include/hugs.h hugs_debug_set_nt_cycles): s_memtime
  # cycles and tiles per (epilogue specialisation, K class), summed over every workgroup of every launch of these steps
  cyc = torch.zeros(64 * 4 * 2, dtype=torch.int64, device='cuda')
  torch.cuda.synchronize()
  _lib.call('hugs_debug_set_nt_cycles', cyc.data_ptr())
  _lib.PROFILE = []
  try:
    for _ in range(steps):
      state, stats, gen = train_step(gen, state, next_batch(), 0.5, thr)
    torch.cuda.synchronize()
  finally:
    _lib.call('hugs_debug_set_nt_cycles', 0)
  cyc = cyc.cpu().numpy().reshape(64, 4, 2)
  recs, _lib.PROFILE = _lib.PROFILE, None
  agg = {}
  for name, key, e0, e1 in recs:
    agg.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)     # us
  traffic = {}
  tname = next((n for n in ('r06_gemm_traffic.json', 'r05_gemm_traffic.json', 'r04_gemm_traffic.json') if os.path.exists(os.path.join(ROOT, 'profiles', n))), None)
  if tname is not None:
    traffic = json.load(open(os.path.join(ROOT, 'profiles', tname)))

  def entry(key, tkey, label, flops, alg_bytes):
    if key not in agg:
      return None
    us = float(np.mean(agg[key]))
    tf = flops / (us * 1e-6) / 1e12
    t = traffic.get(tkey, {})
    lo = float(np.min(agg[key]))      # the fastest launch of the profiled steps: the kernel when no side-stream kernel shares the chip with it
    return {"bound": "mfma", "kernel": label, "launches": len(agg[key]), "avg_us": round(us, 1), "min_us": round(lo, 1),
            "frac_at_min_us": round(flops / (lo * 1e-6) / PEAK_BF16, 4), "achieved": round(tf, 1),
            "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": round(tf * 1e12 / PEAK_BF16, 4),
            "traffic": t.get("hbm_bytes_per_launch"), "algorithmic_bytes": alg_bytes if alg_bytes is not None else t.get("algorithmic_bytes"),
            "traffic_source": (f"profiles/{tname}[{tkey}]: separate rocprofv3 --pmc passes, {t.get('avg_us_profiled')} us per launch "
                               f"under the profiler vs {round(us, 1)} us in-step") if t else None}

  W = 1024
  # the forward trunk may run in row chunks (engine._mlp_forward): take whatever M the W x W relu layers ran at
  fwd = [k for k in agg if k[0] == 'nt' and k[2] == W and k[3] == W and k[4] == 'relu']
  main = None
  if fwd:
    k = max(fwd, key=lambda k: len(agg[k]))
    main = entry(k, 'nt_fwd', f"NT forward trunk [{k[1]}x1024]x[1024x1024]^T +bias +relu, writes 1-bit relu masks "
                 "(gemm_bf16::k_gemm_nt_bf16_p64<35>; HUGS_NT_K64=0: k_gemm_nt_bf16_pers<35>)", 2.0 * k[1] * W * W, 2.0 * k[1] * W * 2 + W * W * 2 + k[1] * W / 8)
  def per_cycle(ent, epi, K):
    # what the fraction is made of, MEASURED by these very launches: cycles per 256 x 256 x K tile on a CU against the 4096 flop / cycle / CU
    # that 2.5 PFLOP/s at 2.4 GHz on 256 CUs means (workgroup start to its last stage's retirement, prologue included); the rest is the
    # clock the power manager sustains under this load: frac = per_cycle_frac x clock / 2.4 GHz
    c, t = cyc[epi, {512: 1, 1024: 2}.get(K, 3)]
    if ent is None or t <= 0:
      return
    cpt = float(c) / float(t)
    ent["cycles_per_tile"] = round(cpt, 0)
    ent["per_cycle_frac"] = round(2.0 * 256 * 256 * K / cpt / 4096.0, 4)
    ent["implied_clock_ghz"] = round(ent["frac"] / ent["per_cycle_frac"] * 2.4, 3)
    ent["per_cycle_source"] = (f"measured in these {steps} steps: s_memtime account of k_gemm_nt_bf16_pers<{epi}> (hugs_debug_set_nt_cycles), "
                               f"{int(t)} tiles of 256x256x{K}")
  per_cycle(main, 35, W)
  M = max(fwd, key=lambda k: len(agg[k]))[1] if fwd else 131072      # rows of the NerfMLP level (131072 at cfg2)
  fl = 2.0 * M * W * W
  tnk = [k for k in agg if k[0] == 'tn' and k[1] == M and k[2] == W and k[3] == W]
  tnb = [k for k in agg if k[0] == 'tnb' and k[1] > 4 * fl]      # the NerfMLP trunk's batched weight-gradient launches
  others = [entry(('nt', M, W, W, 'mask'), 'nt_dx', f"NT dX [{M}x1024]x[1024x1024] *relu-mask bits (gemm_bf16::k_gemm_nt_bf16_p64<16>)", fl,
                  2.0 * M * W * 2 + W * W * 2 + M * W / 8)] + [
            entry(k, 'tn_dw', f"TN dW [1024x{M}]x[{M}x1024] {k[4]} (gemm_bf16::k_gemm_tn_bf16_big)", fl, 2.0 * M * W * 2 + W * W * 4) for k in tnk] + [
            entry(k, 'tn_dw_batch', f"TN dW of {k[2]} trunk items in one launch, {k[3]} reduction pieces per tile + the reduce "
                                    f"(gemm_bf16::k_gemm_tn_bf16_batch)", k[1], None) for k in tnb]
  per_cycle(others[0], 16, W)
  shapes = {(f"{k[0]} M={k[1]} {k[2]}x{k[3]} {k[4]}" if k[0] != 'tnb' else f"tnb {k[4]} GFLOP={k[1] / 1e9:.1f}"): [len(v), round(float(np.mean(v)), 1)]
            for k, v in sorted(agg.items(), key=str)}
  return main, [o for o in others if o], shapes, state, gen


def cpu_baseline(seed):
  """Oracle (CPU restatement of the reference) FULL training steps at the stated workload -- 1024 rays x (64 + 128) samples, the same
  nets -- on the box's host cores (the reference's convention: train.py:162-165 times whole steps), >= 2 timed steps per thread
  count tried; `small_sample` keeps rounds 1-5's 64-ray figure.  Budget ~40 s."""
  from oracle import torch_ref as R
  host = os.cpu_count()
  model_name = next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), 'unknown') \
      if os.path.exists('/proc/cpuinfo') else 'unknown'
  try:      # physical cores: one thread per core is what a dense fp32 GEMM wants
    smt = open('/sys/devices/system/cpu/smt/active').read().strip() == '1'
  except OSError:
    smt = False
  phys = max(1, host // 2 if smt else host)
  cfg = R.kubric_cfg(num_levels=2, num_prop_samples=64, num_nerf_samples=128)
  params = R.init_params(cfg, seed)
  T = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))
  leaves = {nme: v for nme, v in R.flat_leaves(params['params'])}
  m = {k: torch.zeros_like(v) for k, v in leaves.items()}
  v_ = {k: torch.zeros_like(v) for k, v in leaves.items()}

  def make(n):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3)); d = d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.8, 1.2, (n, 1))
    rays = dict(origins=T(rng.normal(size=(n, 3)) * 0.5), directions=T(d), viewdirs=T(d / np.linalg.norm(d, axis=-1, keepdims=True)),
                radii=T(rng.uniform(5e-4, 2e-3, (n, 1))), lossmult=T(np.ones((n, 1))), static_mask=T(np.ones((n, 1))),
                near=T(np.full((n, 1), 0.1)), far=T(np.full((n, 1), 1.2)), embed_idx=torch.zeros(n, 1, dtype=torch.int32))
    return rays, T(rng.uniform(size=(n, 3)))

  kk = 0

  def step(n, rays, gt):
    nonlocal kk
    u01 = [torch.rand(n) for _ in range(2)]
    stats, grads, _, _ = R.loss_and_grad(cfg, params, rays, gt, 0.5, u01)
    g = R.clip_gradients(cfg, grads)
    R.adam_update(cfg, leaves, g, m, v_, kk)
    kk += 1

  t_start = time.time()

  def timed(n, ncores, nsteps, min_time=0.0):
    """(rays/s, steps) over `nsteps` timed steps (more while `min_time` has not passed), behind one 64-ray warm-up step at this
    thread count (thread pool, allocator: a full-size warm-up step would cost a third of the budget)."""
    rays, gt = make(n)
    torch.set_num_threads(ncores)
    rw, gw = make(64)
    step(64, rw, gw)
    t0, ts = time.time(), []
    while len(ts) < nsteps or time.time() - t0 < min_time:
      t1 = time.time(); step(n, rays, gt); ts.append(time.time() - t1)
    return n / min(ts), len(ts)      # the FASTEST step (the first full-size one still grows the allocator: the CPU is not to be under-reported)

  # rounds 1-5's sample: 64 rays, 16 threads (8192-row GEMMs: too small to occupy the box -- kept for continuity)
  small, ks = timed(64, min(host, 16), 2, 2.0)
  # the stated workload: 1024 rays; thread counts tried: HUGS_CPU_THREADS, or 64 and the physical core count.  Budget ~40 s of CPU work:
  # the first candidate gets two timed steps, a further candidate ONE, and only while 0.8 x the best step time is left of the budget
  tries = [int(os.environ['HUGS_CPU_THREADS'])] if 'HUGS_CPU_THREADS' in os.environ else sorted({min(phys, 64), min(phys, 128)})
  best, tried = None, {}
  for ncores in tries:
    if best is not None and 40.0 - (time.time() - t_start) < 0.8 * 1024 / best[0]:
      tried[str(ncores)] = 'not run: budget'
      continue
    r, k = timed(1024, ncores, 2 if best is None else 1)
    tried[str(ncores)] = round(r, 2)
    if best is None or r > best[0]:
      best = (r, ncores, k)
  return {"value": round(best[0], 2), "unit": "rays/s", "cores": best[1], "host_cpu_count": host, "host_physical_cores": phys,
          "cpu_model": model_name, "kind": "port", "rays_per_s_by_threads": tried,
          "sample": f"fastest of {best[2]} full train steps of 1024 rays x (64+128) samples (the stated workload), oracle/torch_ref.py fp32, {best[1]} threads",
          "small_sample": {"value": round(small, 2), "cores": min(host, 16), "sample": f"fastest of {ks} full train steps of 64 rays (the rounds 1-5 sample)"}}


def eval_psnr_vs_oracle(model, state, batch, dtype, flat=None):
  """'eval PSNR' leg of the metric (no dataset ships): PSNR of the HIP render (deterministic eval forward, the
  benchmarked compute dtype, trained weights of this run) against the oracle's fp32 CPU render of the same 256
  rays with the same weights."""
  from oracle import torch_ref as R
  from nerf_hugs_amd.internal import models
  flat = state.flat if flat is None else flat                 # (flat: another parameter buffer of the same model, e.g. the initial weights)
  rays = batch.rays.map(lambda x: x[:1])                      # one 16 x 16 patch
  rend, _ = model.apply(flat, None, rays, 1.0, False)
  hip = rend[-1]['rgb'].reshape(-1, 3).float().cpu()
  cfg = R.kubric_cfg(num_levels=2, num_prop_samples=64, num_nerf_samples=128)
  tree = model.variables(flat)['params']
  P = {m: {k: {kk: vv.detach().float().cpu() for kk, vv in v.items()} for k, v in sub.items()} for m, sub in tree.items()}
  r = rays.flat()
  orays = dict(origins=r.origins.cpu(), directions=r.directions.cpu(), viewdirs=r.viewdirs.cpu(), radii=r.radii.cpu(),
               lossmult=r.lossmult.cpu(), static_mask=r.static_mask.cpu(), near=r.near.cpu(), far=r.far.cpu(),
               embed_idx=r.embed_idx.cpu())
  torch.set_num_threads(min(os.cpu_count(), 16))
  with torch.no_grad():
    orend, _ = R.model_forward(cfg, {'params': P}, orays, 1.0, None, False)
  mse = float(((hip - orend[-1]['rgb'])**2).mean())
  return round(-10.0 * np.log10(max(mse, 1e-20)), 2)


def nerfacto_roofline(model, step_fn, N, steps=3):
  """Per-kernel figures of the nerfacto step, measured INSIDE extra train steps (HIP events around every launch of the
  hash-grid, fused-proposal and GEMM entry points on the stream they run on).  Bounds: hash-grid forward = gathered table
  bytes (8 corners x features x 4 B per sample and level) + the row written, against HBM 8 TB/s; hash-grid backward = the
  gradient rows read + one read-modify-write of the touched table entries against HBM 8 TB/s (its real limiter, the L2 float-atomic
  rate, is reported beside the bound as `atomic_updates_per_s`); GEMMs = 2 M K N against the dense 16-bit MFMA peak; the fused proposal networks = their algorithmic
  flops against the 16x16x16 MFMA rate (16-bit modes) or the fp32 VALU peak (parity mode)."""
  from nerf_hugs_amd import _lib
  # round 4: the profiled steps run on ONE stream (HUGS_NF_BWD_STREAMS=0 for their duration): with the per-level backward streams
  # of the timed steps every bracketed kernel shares the chip with two others and its event time is inflated by them (round 3's
  # `instep_profiled_ms_per_step` 14.5 > the 12.3 ms step); the dominant kernel is picked from these stand-alone durations
  old_env = os.environ.get('HUGS_NF_BWD_STREAMS')
  os.environ['HUGS_NF_BWD_STREAMS'] = '0'
  try:
    step_fn(); torch.cuda.synchronize()
    _lib.PROFILE = []
    for _ in range(steps):
      step_fn()
    torch.cuda.synchronize()
  finally:
    if old_env is None:
      os.environ.pop('HUGS_NF_BWD_STREAMS', None)
    else:
      os.environ['HUGS_NF_BWD_STREAMS'] = old_env
  recs, _lib.PROFILE = _lib.PROFILE, None
  agg = {}
  for name, key, e0, e1 in recs:
    agg.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
  # the fraction of samples that carry a gradient into each grid (rows of d_out with any non-zero): one extra step with the table-gradient
  # calls watched -- what the atomic scatter's work is proportional to (round 6: profiles/r06_cfg5_hashgrid_levels.txt)
  live_frac = {}
  orig_call = _lib.call
  def spy(name, *a):
    if name in ('hugs_hashgrid_bwd', 'hugs_hashgrid_bwd_ws'):
      live_frac[(a[0], a[1], a[2])] = float((a[7].float().abs().sum(-1) != 0).float().mean())
    return orig_call(name, *a)
  _lib.call = spy
  try:
    step_fn(); torch.cuda.synchronize()
  finally:
    _lib.call = orig_call
  grids = {(g.n_levels, g.features): g for g in model.grids.values()}
  HBM = 8e12

  def two_bounds(ent, us, by, fl, peak_flops, label):
    """bound = whichever of (algorithmic bytes / 8 TB/s, algorithmic flops / peak) is the longer; frac = that time / measured."""
    t_hbm, t_cmp = by / HBM, fl / peak_flops
    hb = t_hbm >= t_cmp
    ent.update(bound="hbm" if hb else label, achieved=round((by / (us * 1e-6) / 1e9) if hb else (fl / (us * 1e-6) / 1e12), 2),
               peak=8000.0 if hb else round(peak_flops / 1e12, 1), unit="GB/s" if hb else "TFLOP/s",
               frac=round(max(t_hbm, t_cmp) / (us * 1e-6), 4), algorithmic_bytes=by, algorithmic_flops=fl,
               hbm_time_us=round(t_hbm * 1e6, 1), compute_time_us=round(t_cmp * 1e6, 1))
  out = []
  for key, v in agg.items():
    us = float(np.mean(v))
    per_step = len(v) / steps
    ent = {"kernel": None, "launches_per_step": per_step, "avg_us": round(us, 1), "ms_per_step": round(us * per_step * 1e-3, 3)}
    if key[0] in ('hg_fwd', 'hg_bwd'):
      n, L_, F = key[1:]
      g = grids.get((L_, F))
      dense = sum(1 for l in range(L_) if g is not None and int(g.resolutions[l]) ** 3 <= int(g.offsets[l + 1] - g.offsets[l]))
      if key[0] == 'hg_fwd':
        by = n * L_ * 8 * F * (2 if model.amp else 4) + n * L_ * F * 2 + n * 12
        ent.update(kernel=f"k_hashgrid_fwd {n} samples x {L_} levels", bound="hbm", achieved=round(by / (us * 1e-6) / 1e9, 1), peak=8000.0,
                   unit="GB/s", frac=round(by / (us * 1e-6) / 8e12, 4), algorithmic_bytes=by)
      else:
        # bound (every kernel by max(bytes / 8 TB/s, flops / peak)): the gradient rows read + the sample
        # positions + one read-modify-write of every table entry the samples can touch (at most the levels' tables).  The
        # kernel's own limiter is the L2 float-atomic rate, reported beside it: `atomic_updates` = the 8 x F products per sample
        # and level BEFORE the kernel merges runs of lanes that hit one cell (its issued count is data-dependent and lower).
        tbl = float(g.offsets[L_] - g.offsets[0]) * F * 4 if g is not None else float('inf')
        by = n * L_ * F * (2 if model.dt else 4) + n * 12 + 2 * min(float(n) * L_ * 8 * F * 4, tbl)
        ent.update(kernel=f"k_hashgrid_bwd(+_l0) {n} samples x {L_} levels ({dense} dense)")
        two_bounds(ent, us, by, 0.0, PEAK_BF16, "mfma")
        ent.update(atomic_updates=n * L_ * 8 * F, atomic_updates_per_s=round(n * L_ * 8 * F / (us * 1e-6) / 1e9, 1),
                   limiter="L2 float atomics (scratch/atomic_pair.hip: ~21 G distinct-address transactions/s on this chip)")
        lf = live_frac.get((n, L_, F))
        if lf is not None and F == 2:
          # the roof this kernel actually sits under (round 6): an x-adjacent corner pair of 2 features = one 16-byte transaction when the
          # pair is contiguous (dense levels always; hashed levels for even cx: every other cell), else two: 4 per sample on a dense
          # level, 6 on average on a hashed one, for the samples that carry a gradient; chip rate 21 G/s for isolated addresses, 42 G/s for
          # adjacent pairs (scratch/atomic_pair.hip).  Runs of lanes in one cell that the kernel merges in the wave lower the count at
          # coarse levels; the estimate is an upper bound on the work and the fraction therefore a lower bound.
          tr = n * lf * (4.0 * dense + 6.0 * (L_ - dense))
          ent["atomic_roofline"] = {"live_sample_fraction": round(lf, 4), "transactions_est": round(tr), "achieved_G_per_s": round(tr / (us * 1e-6) / 1e9, 1),
                                    "peak_G_per_s": [21.0, 42.0], "frac_of_42": round(tr / (us * 1e-6) / 42e9, 3),
                                    "source": "profiles/r06_cfg5_hashgrid_levels.txt"}
    elif key[0] in ('field_fwd', 'field_bwd'):
      # csrc/hugs_fieldfuse.hip, per sample (16-bit operands): forward reads 32 hash features (64 B) and writes Y0, H0, H1 (3 x 512),
      # the head input (256), 2 x 32 B of mask bits, raw (2), density (4), rgb (12); backward reads G1 (512), the masks (64), raw /
      # d_density / sel (10) and writes G0, Gy0 (2 x 512), Gb (256), the 32 feature gradients (64).  flops: the four matrices.
      n, g_, a_ = key[1:]
      mats = 32 * 256 + 256 * (1 + g_) + (16 + g_ + a_) * 256 + 256 * 256
      if key[0] == 'field_fwd':
        by, fl = n * (64 + 3 * 512 + 256 + 64 + 18), 2.0 * n * (mats + 256 * 3)
      else:
        by, fl = n * (512 + 64 + 10 + 2 * 512 + 256 + 64), 2.0 * n * mats
      ent.update(kernel=f"k_{key[0]} {n} samples (32 -> 256 -> {1 + g_}; {16 + g_ + a_} -> 256 -> 256 -> 3), activations in LDS, weights in registers")
      two_bounds(ent, us, by, fl, PEAK_BF16, "mfma")
    elif key[0] in ('prop_fwd', 'prop_bwd'):
      n, i_, h_ = key[1:]
      fl = 2.0 * n * (i_ * h_ + h_) * (1 if key[0] == 'prop_fwd' else 3)
      if model.dt and i_ <= 16:      # 16-bit feature rows: the matrix-core kernels (v_mfma_f32_16x16x16: half the 16x16x32 rate)
        # per sample: the 16-wide feature row in (32 B) + density / raw out (8 B); backward: + d_density in (4 B) + the row's gradient out (32 B)
        by = n * (40 if key[0] == 'prop_fwd' else 76)
        ent.update(kernel=f"k_nf_{key[0]}_mfma {n} samples {i_}->{h_}->1 (algorithmic flops; the kernel pads {i_} -> 16 inputs)")
        two_bounds(ent, us, by, fl, PEAK_BF16 / 2, "mfma")
      else:
        ent.update(kernel=f"k_nf_{key[0]} {n} samples {i_}->{h_}->1 (fp32 VALU)", bound="valu", achieved=round(fl / (us * 1e-6) / 1e12, 2),
                   peak=157.3, unit="TFLOP/s", frac=round(fl / (us * 1e-6) / 157.3e12, 4))
    else:
      fl = 2.0 * key[1] * key[2] * key[3]
      # nt key = (M, N, K): A [M,K] in + out [M,N]; tn key = (M, Kc, N): X [M,Kc] + G [M,N] in, fp32 [Kc,N] out (16-bit activations)
      by = (2.0 * key[1] * (key[2] + key[3]) + (2.0 if key[0] == 'nt' else 4.0) * key[2] * key[3])
      ent.update(kernel=f"gemm_{key[0]} M={key[1]} {key[2]}x{key[3]} {key[4]} (padded shape)")
      two_bounds(ent, us, by, fl, PEAK_BF16, "mfma")
    out.append(ent)
  out.sort(key=lambda e: -e["ms_per_step"])
  return out


def bench_nerfacto(args, device, world, rank):
  """Informational line for BASELINE configs[4] (nerfacto hash-grid path): 16384 rays per GPU, full train step."""
  import torch.distributed as dist
  from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
  from nerf_hugs_amd.nerfacto.configs import PHOTOTOURISM_NERFACTO_BASE as CFG5     # BASELINE.json configs[4]: the yml's sizes
  model = NerfactoModel(NerfactoConfig(**CFG5), device=device, compute_dtype=args.dtype, seed=20200823)
  N = 16384
  g = torch.Generator(device=device).manual_seed(100 + rank)
  d = torch.randn(N, 3, generator=g, device=device); d = d / d.norm(dim=-1, keepdim=True)
  batch = dict(origin=(torch.rand(N, 3, generator=g, device=device) - 0.5) * 0.6, direction=d, viewdir=d,
               near=torch.full((N,), 0.05, device=device), far=torch.full((N,), 3.0, device=device),
               embed_idx=torch.randint(0, 3500, (N,), generator=g, device=device).int(), bg_rgb=torch.ones(N, 3, device=device),
               rgb=torch.rand(N, 3, generator=g, device=device))
  draws = lambda: [torch.rand(N, generator=g, device=device) for _ in range(3)]
  res = None

  def step():
    nonlocal res
    res = model.train_step(batch, u01=draws(), world=world)      # (`model` is rebound for the second measurement below)
  for _ in range(args.warmup):
    step()

  def window():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      step()
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    w = time.perf_counter() - t0
    if world > 1:
      tmax = torch.tensor([w], device=device, dtype=torch.float64)
      dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
      w = float(tmax.item())
    return w
  wins = [window()]
  for _ in range(int(min(max(1, math.ceil(args.min_time / wins[0])), args.max_windows)) - 1):
    wins.append(window())
  dt = float(np.median(wins))
  kernels = nerfacto_roofline(model, step, N) if world == 1 else None
  # the same step with the grid-input gradients stored in 16 bits (what rounds 3-4 timed: in the half mode scaled gradients below
  # 6e-8 flush to zero there and skip their table atomics) -- a second model, a quarter of the timed budget
  half_grad = None
  if world == 1 and model.dt and model.grid_grad_f32:
    m_main = model
    model = NerfactoModel(NerfactoConfig(**CFG5), device=device, compute_dtype=args.dtype, seed=20200823, grid_grad_f32=False)
    for _ in range(args.warmup):
      step()
    w2 = [window() for _ in range(max(2, len(wins) // 4))]
    half_grad = {"ms_per_step": round(float(np.median(w2)) / args.steps * 1e3, 3), "value": round(N * args.steps / float(np.median(w2)), 1),
                 "note": "HUGS_NF_GRID_GRAD_F32=0: the fused kernels' feature gradients rounded to the 16-bit operand type before the table scatter"}
    model = m_main
    step(); torch.cuda.synchronize()      # (`res` below is the main model's again)
  if rank == 0:
    st = res['stats'].cpu().numpy()
    extra = {"loss_scale_last": model.loss_scale()} if model.amp else {}
    line = {"metric": "train rays/sec (nerfacto, 16384-ray batch per GPU, 512+256+128 samples)", "value": round(N * world * args.steps / dt, 1),
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "windows": len(wins), "value_min": round(N * world * args.steps / max(wins), 1), "value_max": round(N * world * args.steps / min(wins), 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[4] restatement: nerfacto hash-grid fields (phototourism_nerfacto_base.yml sizes), "
                                   "16384 rays/GPU, full train step", "params": int(model.flat.numel()), "parallelism": f"dp{world}"},
            "loss_rgb_last": round(float(st[1]), 6), **extra}
    line["grid_input_gradient"] = "fp32" if (model.dt and model.grid_grad_f32) else ("16-bit" if model.dt else "fp32 (parity mode)")
    if half_grad is not None:
      line["with_16bit_grid_input_gradient"] = half_grad
    if kernels:
      line["roofline"] = kernels[0]
      line["instep_kernels"] = kernels[1:12]
      line["instep_profiled_ms_per_step"] = round(sum(k["ms_per_step"] for k in kernels), 3)
      line["roofline_note"] = ("per-kernel durations from steps enqueued on ONE stream (stand-alone times, not inflated by the per-level "
                               "backward streams of the timed steps); every kernel is bounded by max(algorithmic bytes / 8 TB/s, flops / peak)")
    print(json.dumps(line))
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=30)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--dtype', default=None, choices=['bf16', 'fp32', 'fp16'],
                  help="compute dtype; default bf16 (BASELINE config 2) and, for --config cfg5, fp16 (the reference's enable_amp: half operands,\n"
                       "half table copies, dynamic loss scaling)")
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--min-time', type=float, default=8.0,
                  help='repeat the K-step timed window until this many seconds of timed steps have run; value = median window')
  ap.add_argument('--max-windows', type=int, default=400)
  ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                  help='weak: 1024 rays per GPU (the default the driver runs); strong: 1024 rays in total, split over the GPUs '
                       '(BASELINE.md promises both curves)')
  ap.add_argument('--rays-per-gpu', type=int, default=None,
                  help='override the per-GPU batch (a multiple of 64: whole 8x8 patches): the per-rank workload of the fixed-global-batch\n'
                       'scaling curve on ONE GPU, e.g. 128 = what --scaling strong hands each of 8 ranks')
  ap.add_argument('--step-graph', default=None, choices=['0', '1'],
                  help='replay the train step as a captured hipGraph (HUGS_STEP_GRAPH); default: the library default')
  ap.add_argument('--batch-pool', type=int, default=64,
                  help='number of pre-generated synthetic batches (resident in HBM) cycled one per step, so that no batch is seen\n'
                       'twice inside a timed window of <= this many steps (VERDICT r4 item 1a: the old bench trained ~1300 steps on ONE\n'
                       'memorised batch; GEMM time depends on operand values).  The fixed-batch rate is reported beside it.')
  ap.add_argument('--targets', default='scene', choices=['scene', 'random'],
                  help='scene (default): the batch pool is drawn from an analytic scene (nerf_hugs_amd/internal/synthetic.py), so the network\n'
                       'learns and train PSNR rises; random: independent random colours (the rounds 1-5 pool).  The other one is timed\n'
                       'beside the headline as `random_targets` / `scene_targets`.')
  ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg3', 'cfg4', 'cfg5', 'ref360'],
                  help='cfg2 = the headline workload; cfg3 (static masks, 4096 rays, GLO 48, charb) and cfg4 (RobustNeRF 0.8,\n'
                       'contract + reciprocal, GLO 4, 1024 rays/GPU) and cfg5 (nerfacto hash-grid path, 16384 rays/GPU,\n'
                       'phototourism_nerfacto_base.yml sizes) are informational; ref360 = the reference-default shape (360.gin: L=3,\n'
                       'S=(64,64,32), 16384 rays, contract + reciprocal), the only shape with a published number (BASELINE.md section 1)')
  args = ap.parse_args()
  if args.dtype is None:
    args.dtype = 'fp16' if args.config == 'cfg5' else 'bf16'
  if args.step_graph is not None:      # (read when nerf_hugs_amd.internal.train_utils is imported)
    os.environ['HUGS_STEP_GRAPH'] = args.step_graph
  if args.dtype == 'fp16' and args.config != 'cfg5':
    ap.error('fp16 is the nerfacto path (cfg5) mode; the Mip-NeRF 360 path runs bf16 or fp32')
  import torch.distributed as dist
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus > 1 and world == 1:
    raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...')
  if 'HUGS_FORCE_DEVICE' in os.environ:     # test hook: several ranks on one GPU (with HUGS_DIST_BACKEND=gloo)
    local = int(os.environ['HUGS_FORCE_DEVICE'])
  torch.cuda.set_device(local)
  device = torch.device('cuda', local)
  if world > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    backend = os.environ.get('HUGS_DIST_BACKEND', 'nccl')   # "nccl" is RCCL on ROCm
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=device)
    else:
      dist.init_process_group(backend)
  if args.config == 'cfg5':
    return bench_nerfacto(args, device, world, rank)
  from nerf_hugs_amd.internal import configs, train_utils
  configs.clear_config()
  # the CPU baseline runs FIRST (it used to run last: the driver's 5-s SMI poll then saw 12 s of idle GPU next to 2 s of work)
  cpu_base = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 'cfg2' and args.rays_per_gpu is None:
    cpu_base = cpu_baseline(20200823)
  gin = {'cfg2': GIN, 'cfg3': GIN_CFG3, 'cfg4': GIN_CFG4, 'ref360': GIN_REF360}[args.config]
  configs.parse_config_files_and_bindings(None, gin)
  rays_per_gpu = {'cfg3': 4096, 'ref360': 16384}.get(args.config, 1024)
  if args.rays_per_gpu is not None:
    if args.rays_per_gpu % 64 or args.scaling == 'strong':
      raise SystemExit('--rays-per-gpu: a multiple of 64 (whole 8x8 patches), not together with --scaling strong')
    rays_per_gpu = args.rays_per_gpu
  P = 16
  if args.scaling == 'strong':
    if rays_per_gpu % world or (rays_per_gpu // world) % 64:
      raise SystemExit(f'--scaling strong: {rays_per_gpu} rays do not split into whole 8x8 patches over {world} GPUs')
    rays_per_gpu //= world
    if rays_per_gpu % 256:
      if args.config == 'cfg4':
        raise SystemExit('RobustNeRF needs whole 16x16 patches per GPU: strong scaling stops at 4 GPUs for 1024 rays')
      P = 8                      # 128 rays per GPU = two 8x8 patches (the plain / static-mask losses have no patch structure)
  if rays_per_gpu % 256:
    if args.config == 'cfg4':
      raise SystemExit('RobustNeRF needs whole 16x16 patches per GPU')
    P = 8
  config = configs.make_config(batch_size=rays_per_gpu * world)
  model, state, _, train_step, _ = train_utils.setup_model(config, 20200823, compute_dtype=args.dtype, device=device)
  theta_init = state.flat.clone()      # (for the eval-PSNR leg: the render of a network that has structure, see below)
  from nerf_hugs_amd.internal import synthetic

  def make_batch(i, targets=None):
    # targets 'scene' (default): rays of pinhole cameras around an analytic textured sphere, colours a function of the ray -- the
    # network LEARNS, so the GEMM operands (post-relu activations, gradients) are those of a network fitting a scene;
    # 'random': directions / origins / colours drawn independently (rounds 1-5: the network collapses to the mean colour)
    if (targets or args.targets) == 'scene':
      batch = synthetic.scene_batch(np.random.default_rng(1000 + rank + 7919 * i), rays_per_gpu // (P * P), P, device)
    else:
      batch = synth_batch(rays_per_gpu // (P * P), P, 1000 + rank + 7919 * i, device)
    if args.config == 'ref360':     # 360.gin: near 0.2, far 1e6 (contracted space)
      batch.rays.near.fill_(0.2)
      batch.rays.far.fill_(1e6)
    if args.config == 'cfg4':       # distractor-like geometry: near in [0.05, 0.3], far 1e6
      batch.rays.near.uniform_(0.05, 0.3)
      batch.rays.far.fill_(1e6)
    if args.config in ('cfg3', 'cfg4'):
      batch.rays.embed_idx.copy_(torch.randint(0, 3500, (rays_per_gpu // (P * P), 1, 1, 1), device=device).expand_as(batch.rays.embed_idx))
      batch.rays.static_mask.copy_((torch.rand(rays_per_gpu // (P * P), P, P, 1, device=device) < 0.8).float())
    return batch
  # a pool of distinct batches, all resident in HBM before the timed region; step i trains on pool[i % len(pool)]
  pool = [make_batch(i) for i in range(max(1, args.batch_pool))]
  batch = pool[0]
  nstep = 0

  def next_batch():
    nonlocal nstep
    b = pool[nstep % len(pool)]
    nstep += 1
    return b
  # the reference's stream: PRNGKey(20200823) split over the devices (train.py:46,80), threefry on the GPU
  from nerf_hugs_amd.internal import random as hrandom
  gen = hrandom.split(hrandom.PRNGKey(20200823, device), world)[rank].clone()
  thr = None      # RobustNeRF: thresholds are fed back on the device (first step: ones, train.py:130)
  for _ in range(args.warmup):
    state, stats, gen = train_step(gen, state, next_batch(), 0.5, thr)

  def window(fixed=None):
    """EXACTLY --steps steps between barrier + synchronize on both sides; returns the max over ranks (seconds).
    fixed: None = one pool batch per step (the headline), a batch = every step on that one batch."""
    nonlocal state, stats, gen
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      state, stats, gen = train_step(gen, state, next_batch() if fixed is None else fixed, 0.5, thr)
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    w = time.perf_counter() - t0
    if world > 1:
      tmax = torch.tensor([w], device=device, dtype=torch.float64)
      dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
      w = float(tmax.item())
    return w

  # One K-step window is 0.2 s at the default K: too short for a 5-s SMI poll or a +-2 % box-to-box drift to average
  # out.  The K-step window is repeated until --min-time seconds of timed steps have run (the count is fixed from the
  # first window's max-over-ranks time, so every rank runs the same number); `value` is the MEDIAN window.
  wins = [window()]
  nwin = int(min(max(1, math.ceil(args.min_time / wins[0])), args.max_windows))
  for _ in range(nwin - 1):
    wins.append(window())
  dt = float(np.median(wins))
  loss = float(stats['loss'])
  psnr = float(stats['psnr'])
  # the trunk's post-relu active fraction at this point of training: popcount of the 1-bit relu masks the forward NT GEMMs wrote
  # (they are in HBM for the backward pass): the operand statistics the GEMM times below belong to
  active = None
  if rank == 0 and args.dtype == 'bf16':
    try:
      eng = model.engine(device)
      lut = torch.tensor([bin(i).count('1') for i in range(256)], dtype=torch.int64, device=device)
      fr = {}
      for k_, t_ in eng.ws.bufs.items():
        if torch.is_tensor(t_) and isinstance(k_, tuple) and isinstance(k_[0], str) and k_[0].startswith(f'NerfMLP_0/L{model.num_levels - 1}/bits'):
          fr[k_[0].rsplit('/', 1)[1]] = round(float(lut[t_.view(torch.uint8).long()].sum()) / (t_.numel() * 32), 4)
      if fr:
        active = {"mean": round(float(np.mean(list(fr.values()))), 4), "per_layer": dict(sorted(fr.items()))}
    except Exception as e:      # (a measurement beside the headline: never fails the bench)
      active = {"error": repr(e)}
  eval_psnr = eval_psnr_init = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 'cfg2' and args.rays_per_gpu is None:
    eval_psnr = eval_psnr_vs_oracle(model, state, batch, args.dtype)
    # the same comparison at the INITIAL weights (he_uniform trunk), kept from round 5 for continuity
    eval_psnr_init = eval_psnr_vs_oracle(model, state, batch, args.dtype, flat=theta_init)
  # the same windows on ONE fixed batch (what rounds 1-4 reported): a quarter of the timed budget
  fwins = [window(batch) for _ in range(max(3, nwin // 4))] if len(pool) > 1 else list(wins)
  dt_fixed = float(np.median(fwins))
  # host side of a step: wall time of enqueueing 8 steps onto an idle GPU without waiting for them (the launch queue is
  # far deeper than 8 steps); host_enqueue_ms >= ms_per_step means the step is host-bound
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(8):
    state, stats, gen = train_step(gen, state, next_batch(), 0.5, thr)
  host_ms = (time.perf_counter() - t0) / 8 * 1e3
  torch.cuda.synchronize()
  # N > 1: the part of the gradient exchange the backward pass does not hide (events on the compute stream around the point
  # where the step waits for its all-reduces; train_utils.AR_PROFILE), over 10 extra untimed steps -- so that a SCALE run
  # explains itself: ms_per_step(N) - ms_per_step(1) should be about this number
  ar_exposed = None
  if world > 1:
    train_utils.AR_PROFILE = []
    for _ in range(10):
      state, stats, gen = train_step(gen, state, next_batch(), 0.5, thr)
    torch.cuda.synchronize()
    evs, train_utils.AR_PROFILE = train_utils.AR_PROFILE, None
    if evs:
      t_ = torch.tensor([float(np.mean([a.elapsed_time(b) for a, b in evs]))], device=device, dtype=torch.float64)
      dist.all_reduce(t_, op=dist.ReduceOp.MAX)
      ar_exposed = round(float(t_.item()), 4)
  roof = None
  if args.dtype == 'bf16' and args.config in ('cfg2', 'ref360'):
    # after the timed region: a few more steps with the GEMM launches bracketed by HIP events (every rank runs them:
    # the steps contain the gradient all-reduce; rank 0 reports)
    roof, roof_others, roof_shapes, state, gen = instep_roofline(train_step, state, next_batch, gen, thr)
  # LAST (it un-trains the network): the same windows on the OTHER target kind -- random colours when the headline pool is the scene
  other = 'random' if args.targets == 'scene' else 'scene'
  other_line = None
  if len(pool) > 1:
    pool_main, pool = pool, [make_batch(i, other) for i in range(min(len(pool), 32))]
    for _ in range(args.warmup):
      state, stats, gen = train_step(gen, state, next_batch(), 0.5, thr)
    owins = [window() for _ in range(max(3, nwin // 4))]
    dt_o = float(np.median(owins))
    other_line = {"value": round(rays_per_gpu * world * args.steps / dt_o, 1), "ms_per_step": round(dt_o / args.steps * 1e3, 3),
                  "windows": len(owins), "over_headline": round(dt_o / dt, 4), "train_psnr_last": round(float(stats['psnr']), 3),
                  "loss_last": round(float(stats['loss']), 6),
                  "note": f"{other} targets, {len(pool)} batches, timed after everything else on the same (already trained) network"}
    pool = pool_main
  if rank == 0:
    rps = rays_per_gpu * world * args.steps / dt
    line = {
        "metric": "train rays/sec (%d-ray batch per GPU, %s samples)" % (rays_per_gpu, "64+64+32" if args.config == 'ref360' else "64+128"), "value": round(rps, 1), "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "windows": len(wins), "timed_s": round(float(np.sum(wins)), 3), "gpu_timed_s": round(float(np.sum(wins)), 3),
        "host_enqueue_ms_per_step": round(host_ms, 3), "step_graph": bool(getattr(train_step, 'graph_active', lambda: False)()),
        "value_min": round(rays_per_gpu * world * args.steps / max(wins), 1), "value_max": round(rays_per_gpu * world * args.steps / min(wins), 1),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype,
        "data": (f"synthetic: {len(pool)} distinct pre-generated batches resident in HBM, one per step (no batch twice inside a window); "
                 + ("rays + colours of an analytic scene (textured sphere, cameras on a ring): the network learns" if args.targets == 'scene'
                    else "independent random rays and colours")),
        ("random_targets" if other == 'random' else "scene_targets"): other_line,
        "trunk_relu_active_fraction": active,
        "fixed_batch": {"value": round(rays_per_gpu * world * args.steps / dt_fixed, 1), "ms_per_step": round(dt_fixed / args.steps * 1e3, 3),
                        "windows": len(fwins), "fresh_over_fixed": round(dt_fixed / dt, 4),
                        "note": "every step on pool[0] (the rounds 1-4 protocol), timed after the headline windows"},
        "config": {"workload": {"cfg2": "configs[1]: MipNeRF360 base (kubric_1024_base.gin nets), 1024 rays x (64 prop + 128 fine) per GPU, "
                                        "full train step",
                                "cfg3": "configs[2] restatement: + HuGS static masks, GLO 48, charb, 4096 rays x (64+128), full train step",
                                "cfg4": "configs[3] restatement: RobustNeRF 0.8, contract + reciprocal, GLO 4, 1024 rays/GPU x (64+128)",
                                "ref360": "reference-default shape (360.gin + Config/Model defaults): L=3, S=(64,64,32), 16384 rays/GPU, contract + "
                                          "reciprocal, charb + interlevel + distortion losses, full train step (BASELINE.md section 1: ~177 k rays/s "
                                          "derived from the upstream table, hardware not stated)"}[args.config],
                   "rays_per_gpu": rays_per_gpu, "global_batch": rays_per_gpu * world,
                   "parallelism": f"dp{world}", "params": model.layout.num_params()},
        "train_psnr_last": round(psnr, 3), "loss_last": round(loss, 6),
        "eval_psnr_vs_cpu_fp32_db": eval_psnr, "eval_psnr_vs_cpu_fp32_init_weights_db": eval_psnr_init,
        "step_mfma_frac": (round(rps / world * (FLOP_TRAIN_PER_RAY_REF360 if args.config == 'ref360' else FLOP_TRAIN_PER_RAY) /
                                 (PEAK_BF16 if args.dtype == 'bf16' else 157.3e12), 4) if args.config in ('cfg2', 'ref360') else None),
    }
    if world > 1:
      line["allreduce_exposed_ms_per_step"] = ar_exposed
      line["allreduce_form"] = ("two hipGraphs around one eager all-reduce" if line["step_graph"] else
                                ("bucketed, issued under the backward pass" if os.environ.get('HUGS_AR_BUCKETS', '1') != '0' else "one all-reduce after the backward pass"))
    if roof is not None:
      line["roofline"] = roof
      line["instep_kernels"] = roof_others
      line["instep_gemm_shapes_count_avg_us"] = roof_shapes
    if cpu_base is not None:
      line["cpu_baseline"] = cpu_base
    print(json.dumps(line))
  if world > 1:
    dist.barrier()          # rank 0 measures the roofline / prints after the timed region: keep the group alive until then
    dist.destroy_process_group()


if __name__ == '__main__':
  main()

"""Shared helpers for the GPU parity tests: synthetic rays (SURVEY 8d distributions), model construction in
both the product (nerf_hugs_amd) and the oracle (oracle.torch_ref) from the same weights."""
import numpy as np
import torch

from nerf_hugs_amd.internal import configs, models, train_utils, utils
from oracle import torch_ref as R


def synth_rays(n_patch, P, seed, near=0.1, far=1.2, num_embed=3500, mask_p=0.8):
  rng = np.random.default_rng(seed)
  shp = (n_patch, P, P)
  o = (rng.normal(size=shp + (3,)) * 0.5).astype(np.float32)
  d = rng.normal(size=shp + (3,))
  d = (d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.8, 1.2, shp + (1,))).astype(np.float32)
  v = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
  f = lambda a: torch.from_numpy(np.ascontiguousarray(a))
  nr = np.full(shp + (1,), near, np.float32) if np.isscalar(near) else rng.uniform(near[0], near[1], shp + (1,)).astype(np.float32)
  fr = np.full(shp + (1,), far, np.float32)
  sm = (rng.uniform(size=(n_patch, 1, 1, 1)) < mask_p).astype(np.float32) * np.ones(shp + (1,), np.float32)
  sm = np.where(rng.uniform(size=shp + (1,)) < 0.9, sm, 1 - sm).astype(np.float32)
  rays = utils.Rays(
      pix_coords=f(rng.uniform(size=shp + (2,)).astype(np.float32)), origins=f(o), directions=f(d), viewdirs=f(v),
      radii=f(rng.uniform(5e-4, 2e-3, shp + (1,)).astype(np.float32)), lossmult=f(np.ones(shp + (1,), np.float32)),
      static_mask=f(sm), near=f(nr), far=f(fr),
      embed_idx=f(rng.integers(0, num_embed, (n_patch, 1, 1, 1)).astype(np.int32) * np.ones(shp + (1,), np.int32)),
      cam_idx=f(np.zeros(shp + (1,), np.int32)))
  rgb = f(rng.uniform(size=shp + (3,)).astype(np.float32))
  return utils.Batch(rays=rays, rgb=rgb)


def make_pair(gin_lines, seed=3, compute_dtype='fp32'):
  """Build the product model + state and the oracle cfg + params from the same gin bindings/weights."""
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, gin_lines)
  config = configs.make_config()
  model, state, render_fn, train_step, lr_fn = train_utils.setup_model(config, seed, compute_dtype=compute_dtype)
  mb, nb, pb = configs.bindings('Model'), configs.bindings('NerfMLP'), configs.bindings('PropMLP')
  cfg = R.ModelCfg(
      num_prop_samples=model.num_prop_samples, num_nerf_samples=model.num_nerf_samples, num_levels=model.num_levels,
      raydist_fn=model.raydist, ray_shape=model.ray_shape, num_glo_features=model.num_glo_features,
      opaque_background=model.opaque_background, warp=model.nerf_spec.warp_fn is not None,
      nerf_depth=model.nerf_spec.net_depth, nerf_width=model.nerf_spec.net_width, prop_depth=model.prop_spec.net_depth,
      prop_width=model.prop_spec.net_width, prop_disable_rgb=model.prop_spec.disable_rgb,
      data_loss_type=config.data_loss_type, data_loss_mult=config.data_loss_mult, charb_padding=config.charb_padding,
      withmask_transient_weight=config.withmask_transient_weight, grad_max_val=config.grad_max_val,
      disable_multiscale_loss=config.disable_multiscale_loss, data_coarse_loss_mult=config.data_coarse_loss_mult,
      interlevel_loss_mult=config.interlevel_loss_mult, distortion_loss_mult=config.distortion_loss_mult,
      transient_type=config.transient_type, patch_size=config.patch_size,
      robustnerf_inlier_quantile=config.robustnerf_inlier_quantile, grad_max_norm=config.grad_max_norm,
      lr_init=config.lr_init, lr_final=config.lr_final, max_steps=config.max_steps,
      lr_delay_steps=config.lr_delay_steps, lr_delay_mult=config.lr_delay_mult, adam_eps=config.adam_eps,
      basis_shape=model.nerf_spec.basis_shape, basis_subdivisions=model.nerf_spec.basis_subdivisions,
      max_deg_point=model.nerf_spec.max_deg_point, num_transient_features=model.num_transient_features,
      hanerf_mask_size_loss_mult_min=config.hanerf_mask_size_loss_mult_min,
      hanerf_mask_size_loss_mult_max=config.hanerf_mask_size_loss_mult_max,
      hanerf_mask_size_loss_mult_k=config.hanerf_mask_size_loss_mult_k)
  cfg.transient_depth, cfg.transient_width, cfg.beta_min = (model.nerf_spec.net_depth_transient,
                                                            model.nerf_spec.net_width_transient, model.beta_min)
  cfg.nerfw_beta_loss_mult, cfg.nerfw_beta_loss_bias, cfg.nerfw_density_loss_mult = (
      config.nerfw_beta_loss_mult, config.nerfw_beta_loss_bias, config.nerfw_density_loss_mult)
  cfg.weight_decay_mults = dict(config.weight_decay_mults or {})
  cfg.disable_integration = bool(model.disable_integration)
  cfg.use_viewdirs = bool(model.use_viewdirs)
  cfg.depth_viewdirs = int(model.nerf_spec.net_depth_viewdirs)
  cfg.min_deg_point = int(model.nerf_spec.min_deg_point)
  cfg.rgb_premultiplier, cfg.rgb_bias = float(model.nerf_spec.rgb_premultiplier), float(model.nerf_spec.rgb_bias)
  # the remaining gin-visible knobs the oracle knows (scratch/config_fuzz2.py walks them): sampler, heads, background
  cfg.anneal_slope, cfg.single_jitter = float(model.anneal_slope), bool(model.single_jitter)
  cfg.dilation_multiplier, cfg.dilation_bias, cfg.resample_padding = float(model.dilation_multiplier), float(model.dilation_bias), float(model.resample_padding)
  cfg.bg_intensity = float(model.bg_intensity)
  cfg.near_anneal_rate, cfg.near_anneal_init = model.near_anneal_rate, float(model.near_anneal_init)
  cfg.skip_layer, cfg.deg_view = int(model.nerf_spec.skip_layer), int(model.nerf_spec.deg_view)
  cfg.density_bias, cfg.rgb_padding = float(model.nerf_spec.density_bias), float(model.nerf_spec.rgb_padding)
  cfg.bottleneck_width, cfg.width_viewdirs = int(model.nerf_spec.bottleneck_width), int(model.nerf_spec.net_width_viewdirs)
  if model.mask_spec is not None:
    cfg.mask_depth, cfg.mask_width, cfg.mask_deg_coord = (model.mask_spec.net_depth, model.mask_spec.net_width,
                                                           model.mask_spec.deg_coord)
  # oracle params = copies of the product's (logical, unpadded) leaves
  tree = model.variables(state.flat)['params']
  P = {m: ({k: {kk: vv.detach().cpu().clone() for kk, vv in v.items()} for k, v in sub.items()} if 'Embed' not in m
           else {'embedding': sub['embedding'].detach().cpu().clone()}) for m, sub in tree.items()}
  return config, model, state, render_fn, train_step, cfg, {'params': P}


def oracle_rays(batch):
  r = batch.rays.flat()
  return dict(pix_coords=r.pix_coords, origins=r.origins, directions=r.directions, viewdirs=r.viewdirs, radii=r.radii, lossmult=r.lossmult,
              static_mask=r.static_mask, near=r.near, far=r.far, embed_idx=r.embed_idx)


def relerr(a, b):
  a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
  return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))

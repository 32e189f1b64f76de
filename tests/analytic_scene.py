"""Synthetic analytic scene for training-equivalence checks (no dataset ships): a textured, shaded unit sphere on a white
background seen by pinhole cameras on a ring of radius 3.  `scene_batch` draws whole P x P patches; `psnr_run` trains
the Mip-NeRF 360 model on it and reports the PSNR on a fixed validation set."""
import numpy as np
import torch

from nerf_hugs_amd.internal import configs, train_utils, utils

GIN = ["Config.patch_size = 16", "Config.data_loss_type = 'mse'", "Config.distortion_loss_mult = 0.", "Config.lr_delay_steps = 100",
       "Model.opaque_background = True", "Model.num_levels = 2", "Model.num_prop_samples = 64", "Model.num_nerf_samples = 128",
       "PropMLP.net_depth = 4", "PropMLP.net_width = 256", "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 8"]


def scene_batch(rng, n_patch, P, device):
  o = np.zeros((n_patch, P, P, 3), np.float32); d = np.zeros_like(o)
  for i in range(n_patch):
    th = rng.uniform(0, 2 * np.pi); c = np.array([3 * np.cos(th), 3 * np.sin(th), rng.uniform(-0.5, 0.5)], np.float32)
    f = -c / np.linalg.norm(c); r = np.cross(f, [0, 0, 1]); r /= np.linalg.norm(r); u = np.cross(r, f)
    x0, y0 = rng.uniform(-0.35, 0.35, 2)
    px = x0 + (np.arange(P) - P / 2) * 0.004; py = y0 + (np.arange(P) - P / 2) * 0.004
    X, Y = np.meshgrid(px, py)
    o[i] = c; d[i] = f[None, None] + X[..., None] * r + Y[..., None] * u
  v = d / np.linalg.norm(d, axis=-1, keepdims=True)
  b = (o * v).sum(-1); cc = (o * o).sum(-1) - 1.0; disc = b * b - cc
  hit = disc > 0
  t = -b - np.sqrt(np.maximum(disc, 0))
  p = o + v * t[..., None]
  tex = 0.5 + 0.5 * np.stack([np.sin(6 * p[..., 0]), np.sin(6 * p[..., 1] + 1), np.sin(6 * p[..., 2] + 2)], -1)
  shade = np.clip((p * np.array([0.5, 0.3, 0.8])).sum(-1, keepdims=True) * 0.5 + 0.6, 0.2, 1.0)
  rgb = np.where(hit[..., None], tex * shade, 1.0).astype(np.float32)
  f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(device)
  shp = (n_patch, P, P)
  rays = utils.Rays(pix_coords=f32(np.zeros(shp + (2,))), origins=f32(o), directions=f32(d), viewdirs=f32(v),
                    radii=f32(np.full(shp + (1,), 0.004 * 2 / np.sqrt(12))), lossmult=f32(np.ones(shp + (1,))),
                    static_mask=f32(np.ones(shp + (1,))), near=f32(np.full(shp + (1,), 1.5)), far=f32(np.full(shp + (1,), 4.5)),
                    embed_idx=torch.zeros(shp + (1,), dtype=torch.int32, device=device),
                    cam_idx=torch.zeros(shp + (1,), dtype=torch.int32, device=device))
  return utils.Batch(rays=rays, rgb=f32(rgb))


def psnr_run(dtype, steps, seed, width=1024, max_steps=2000, every=100):
  """Train `steps` steps (1024 rays each: four 16 x 16 patches) from init / batch / jitter seeds derived from `seed`; returns
  [(step, validation PSNR on a fixed 4096-ray set, train loss)]."""
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, GIN + [f"NerfMLP.net_width = {width}", f"Config.max_steps = {max_steps}"])
  config = configs.make_config()
  model, state, _, train_step, _ = train_utils.setup_model(config, 1234 + seed, compute_dtype=dtype)
  rng = np.random.default_rng(seed)
  gen = torch.Generator(device='cuda').manual_seed(seed)
  val = scene_batch(np.random.default_rng(999), 16, 16, 'cuda')
  out = []
  for s in range(steps):
    batch = scene_batch(rng, 4, 16, 'cuda')
    state, stats, gen = train_step(gen, state, batch, s / (config.max_steps - 1), None)
    if s % every == 0 or s == steps - 1:
      rend, _ = model.apply(state.flat, None, val.rays, s / (config.max_steps - 1), False)
      mse = float(((rend[-1]['rgb'] - val.rgb)**2).mean())
      out.append((s, float(-10 * np.log10(mse)), float(stats['loss'])))
  torch.cuda.synchronize()
  configs.clear_config()
  return out

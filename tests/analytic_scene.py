"""Synthetic analytic scene for training-equivalence checks (no dataset ships): a textured, shaded unit sphere on a white
background seen by pinhole cameras on a ring of radius 3.  `scene_batch` draws whole P x P patches; `psnr_run` trains
the Mip-NeRF 360 model on it and reports the PSNR on a fixed validation set."""
import numpy as np
import torch

from nerf_hugs_amd.internal import configs, train_utils, utils

GIN = ["Config.patch_size = 16", "Config.data_loss_type = 'mse'", "Config.distortion_loss_mult = 0.", "Config.lr_delay_steps = 100",
       "Model.opaque_background = True", "Model.num_levels = 2", "Model.num_prop_samples = 64", "Model.num_nerf_samples = 128",
       "PropMLP.net_depth = 4", "PropMLP.net_width = 256", "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 8"]


from nerf_hugs_amd.internal.synthetic import scene_batch  # noqa: F401  (moved under the package in round 6: bench.py trains on it)


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

"""Model.apply in test mode (compute_extras) against the oracle's model_forward under gin variants, fp32 and bf16, ragged ray counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import hugs_testlib as H
from tests.test_gpu_train_step import SMALL
from oracle import torch_ref as R
V = {
  'base': [],
  'glo 4 contract levels 3': ["Model.num_glo_features = 4", "Model.num_levels = 3", "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract", "Model.raydist_fn = @jnp.reciprocal"],
  'no viewdirs': ["Model.use_viewdirs = False"],
  'view depth 4': ["NerfMLP.net_depth_viewdirs = 4"],
  'no opaque bg 0.5': ["Model.opaque_background = False", "Model.bg_intensity_range = (0.5, 0.5)"],
  'cylinder': ["Model.ray_shape = 'cylinder'"],
  'hanerf': ["Config.transient_type = 'hanerf'", "Model.num_transient_features = 16", "Model.num_glo_features = 4", "NerfMLP.bottleneck_width = 128"],
  'nerfw': ["Config.transient_type = 'nerfw'", "Model.num_transient_features = 16", "Model.num_glo_features = 4", "NerfMLP.bottleneck_width = 128"],
  'samples 256/512': ["Model.num_prop_samples = 256", "Model.num_nerf_samples = 512"],
}
for name, extra in V.items():
  keys = {e.split('=')[0].strip() for e in extra}
  gin = [g for g in SMALL if g.split('=')[0].strip() not in keys] + extra
  for dtp in ('fp32', 'bf16'):
    for n_rays in (64, 37):
      try:
        config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(gin, compute_dtype=dtp)
        batch = H.synth_rays(1, 8, 9)
        rays = batch.rays.map(lambda x: x.reshape(64, -1)[:n_rays])
        rend, _ = model.apply(state.flat, None, rays, 1.0, True)
        ob = H.oracle_rays(batch)
        ob = {k: v[:n_rays] for k, v in ob.items()}
        orend, _ = R.model_forward(cfg, oparams, ob, 1.0, None, True)
        worst = 0.
        for k in ['rgb', 'acc', 'distance_mean', 'distance_median']:
          a = rend[-1][k].cpu().reshape(n_rays, -1).double(); b = orend[-1][k].detach().reshape(n_rays, -1).double()
          worst = max(worst, float((a - b).abs().max()) / max(1.0, float(b.abs().max())))
        tol = 2e-4 if dtp == 'fp32' else 5e-2
        print(f'{name:28s} {dtp} n={n_rays:3d} worst rel {worst:.2e} {"ok" if worst < tol else "MISMATCH"}', flush=True)
      except Exception as e:
        print(f'{name:28s} {dtp} n={n_rays:3d} {type(e).__name__}: {str(e)[:140]}', flush=True)

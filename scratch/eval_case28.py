import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import hugs_testlib as H
from tests.test_gpu_train_step import SMALL
from oracle import torch_ref as R
extra = ["Config.transient_type = 'hanerf'", "Model.num_transient_features = 16", "NerfMLP.bottleneck_width = 128", "Model.num_prop_samples = 32",
         "Model.num_nerf_samples = 32", "Model.raydist_fn = @jnp.reciprocal", "Model.opaque_background = False", "Model.bg_intensity_range = (0.5, 0.5)",
         "Config.data_loss_type = 'charb'", "Config.charb_padding = 0.01", "NerfMLP.skip_layer = 2", "PropMLP.skip_layer = 2", "NerfMLP.rgb_padding = 0.",
         "PropMLP.rgb_padding = 0."]
gin = list(SMALL)
for e in extra:
  gin = [g for g in gin if g.split('=')[0].strip() != e.split('=')[0].strip()] + [e]
config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(gin)
batch = H.synth_rays(1, 8, 9, near=(0.05, 0.3), far=1e6)
n = 37
rays = batch.rays.map(lambda x: x.reshape(64, -1)[:n])
rend, hist = model.apply(state.flat, None, rays, 1.0, True)
ob = {k: v[:n] for k, v in H.oracle_rays(batch).items()}
orend, ohist = R.model_forward(cfg, oparams, ob, 1.0, None, True)
for k in ['rgb', 'acc', 'distance_mean', 'distance_median', 'distance_percentile_5', 'distance_percentile_95']:
  a = rend[-1][k].cpu().reshape(n, -1).double(); b = orend[-1][k].detach().reshape(n, -1).double()
  d = (a - b).abs().max(-1).values
  bad = (d > 2e-4 * max(1.0, float(b.abs().max()))).nonzero().flatten().tolist()
  print(k, 'max abs diff', float(d.max()), 'scale', float(b.abs().max()), 'bad rays', bad)
  for r in bad[:4]:
    print('   ray', r, 'prod', a[r].tolist(), 'oracle', b[r].tolist(), 'acc', float(orend[-1]['acc'][r]))
w, ow = hist[-1]['weights'].cpu().reshape(n, -1), ohist[-1]['weights'].detach()
print('weights max diff', float((w - ow).abs().max()), 'sdist max diff', float((hist[-1]['sdist'].cpu().reshape(n, -1) - ohist[-1]['sdist']).abs().max()))

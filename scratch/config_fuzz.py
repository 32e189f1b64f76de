"""Train two steps under a range of gin variants the reference accepts; report which run, which raise what."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import hugs_testlib as H
from tests.test_gpu_train_step import SMALL
VARIANTS = {
  'deg_view 2': ["NerfMLP.deg_view = 2"],
  'deg_view 6': ["NerfMLP.deg_view = 6"],
  'max_deg 16': ["NerfMLP.max_deg_point = 16", "PropMLP.max_deg_point = 16"],
  'max_deg 3': ["NerfMLP.max_deg_point = 3", "PropMLP.max_deg_point = 3"],
  'glo 1': ["Model.num_glo_features = 1"],
  'glo 127': ["Model.num_glo_features = 127"],
  'nerf depth 2': ["NerfMLP.net_depth = 2"],
  'nerf depth 5 (skip after 4 = last)': ["NerfMLP.net_depth = 5"],
  'nerf depth 9': ["NerfMLP.net_depth = 9"],
  'skip_layer 2': ["NerfMLP.skip_layer = 2"],
  'prop depth 1': ["PropMLP.net_depth = 1"],
  'bottleneck 128': ["NerfMLP.bottleneck_width = 128"],
  'bottleneck 384': ["NerfMLP.bottleneck_width = 384"],
  'nerf width 384': ["NerfMLP.net_width = 384"],
  'nerf width 192 (not 128 multiple, depth 4)': ["NerfMLP.net_width = 192", "NerfMLP.net_depth = 4"],
  'prop width 64': ["PropMLP.net_width = 64"],
  'levels 4': ["Model.num_levels = 4"],
  'levels 1': ["Model.num_levels = 1"],
  'samples 48/24': ["Model.num_prop_samples = 48", "Model.num_nerf_samples = 24"],
  'samples 16/8': ["Model.num_prop_samples = 16", "Model.num_nerf_samples = 8"],
  'cylinder': ["Model.ray_shape = 'cylinder'"],
  'single_jitter off': ["Model.single_jitter = False"],
  'no anneal / dilation 0': ["Model.anneal_slope = 0.", "Model.dilation_multiplier = 0.", "Model.dilation_bias = 0."],
  'resample_padding 0.01': ["Model.resample_padding = 0.01"],
  'near_anneal': ["Model.near_anneal_rate = 0.5"],
  'density softplus bias 0': ["NerfMLP.density_bias = 0."],
  'rgb_padding 0': ["NerfMLP.rgb_padding = 0."],
  'charb + coarse loss': ["Config.data_loss_type = 'charb'", "Config.data_coarse_loss_mult = 0.3"],
  'rawnerf loss': ["Config.data_loss_type = 'rawnerf'"],
  'distortion 0 + interlevel 0.5': ["Config.distortion_loss_mult = 0.", "Config.interlevel_loss_mult = 0.5"],
  'grad clip': ["Config.grad_max_norm = 0.01", "Config.grad_max_val = 0.001"],
  'adam betas / eps': ["Config.adam_beta1 = 0.8", "Config.adam_beta2 = 0.99", "Config.adam_eps = 1e-8"],
  'lr delay': ["Config.lr_delay_steps = 100", "Config.lr_delay_mult = 0.1"],
  'patch 4': ["Config.patch_size = 4"],
}
for name, extra in VARIANTS.items():
  keys = {e.split('=')[0].strip() for e in extra}
  gin = [g for g in SMALL if g.split('=')[0].strip() not in keys] + extra
  try:
    config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(gin)
    P = config.patch_size
    batch = H.synth_rays(max(1, 64 // (P * P)), P, 3)
    gen = torch.Generator(device='cuda').manual_seed(5)
    ls = []
    for _ in range(2):
      state, stats, gen = train_step(gen, state, batch, 0.5, None)
      ls.append(float(stats['loss']))
    ok = all(np.isfinite(ls)) and bool(torch.isfinite(state.flat).all())
    print(f'{name:45s} {"ok" if ok else "NON-FINITE"}  loss {ls[0]:.5f} -> {ls[1]:.5f}', flush=True)
  except Exception as e:
    print(f'{name:45s} {type(e).__name__}: {str(e)[:140]}', flush=True)

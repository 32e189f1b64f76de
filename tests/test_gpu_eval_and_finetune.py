"""GPU parity of the eval surface (Model.apply with compute_extras, render_image, create_render_fn) and of the
finetune stage (only GLO embeddings move), against the oracle."""
import functools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GIN = ["Config.patch_size = 8", "Config.data_loss_type = 'mse'", "Model.opaque_background = True", "Model.num_levels = 3",
       "PropMLP.net_depth = 4", "PropMLP.net_width = 128", "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 8",
       "NerfMLP.net_width = 128", "Model.num_glo_features = 4", "Config.render_chunk_size = 96", "Config.vis_num_rays = 4"]


def test_render_image_and_extras_vs_oracle():
  from tests import hugs_testlib as H
  from oracle import torch_ref as R
  from nerf_hugs_amd.internal import models, utils
  config, model, state, render_fn, _, cfg, oparams = H.make_pair(GIN)
  batch = H.synth_rays(1, 16, 9)                       # a 16 x 16 "image"
  rays = batch.rays.map(lambda x: x.reshape(16, 16, -1))
  out = models.render_image(functools.partial(render_fn, state.params, 1.0), rays, None, config, verbose=False)
  orend, ohist = R.model_forward(cfg, oparams, H.oracle_rays(batch), 1.0, None, True)
  ref = orend[-1]
  assert out['rgb'].shape == (16, 16, 3)
  for k in ['rgb', 'acc', 'distance_mean', 'distance_median', 'distance_percentile_5', 'distance_percentile_95']:
    a = out[k].cpu().reshape(-1).double(); b = ref[k].detach().reshape(-1).double()
    assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(b.abs().max())), k
  assert len(out['ray_sdist']) == 3 and out['ray_sdist'][0].shape == (4, 65) and out['ray_rgbs'][2].shape == (4, 32, 3)
  # zero_glo path (Config.enable_render_zero_glo): differs from the embedded one, matches the oracle's
  rend0, _ = model.apply(state.flat, None, batch.rays, 1.0, True, zero_glo=True)
  o0, _ = R.model_forward(cfg, oparams, H.oracle_rays(batch), 1.0, None, True, zero_glo=True)
  assert float((rend0[-1]['rgb'].cpu().reshape(-1, 3) - o0[-1]['rgb'].detach()).abs().max()) < 2e-4
  # history keys / shapes of Model.__call__
  rend, hist = model(state.params, None, batch.rays, 1.0, False)
  assert set(hist[0].keys()) == {'density', 'rgb', 'sdist', 'weights'} and hist[2]['rgb'].shape == (1, 16, 16, 32, 3)


def test_finetune_moves_only_embeddings():
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import train_utils
  config, model, state, _, train_step, cfg, oparams = H.make_pair(GIN + ["Config.finetune_enable = True"])
  fstate, ftrain, lr_fn = train_utils.setup_finetune_model(config, model, state)
  before = fstate.flat.clone()
  batch = H.synth_rays(1, 8, 3)
  gen = torch.Generator(device='cuda').manual_seed(0)
  fstate, stats, gen = ftrain(gen, fstate, batch, 1.0, None)
  torch.cuda.synchronize()
  lay = model.layout
  for lf in lay.leaves:
    d = (lay.view(fstate.flat, lf['path']) - lay.view(before, lf['path'])).abs().max().item()
    if 'embedding' in lf['path']:
      assert d > 0
    else:
      assert d == 0, lf['path']
  assert set(stats['losses'].keys()) == {'data'}        # no interlevel / distortion in the finetune stage
  assert abs(lr_fn(0) / (config.finetune_lr_init * config.finetune_lr_delay_mult) - 1) < 1e-9


def test_leaf_api_wrappers():
  from oracle import torch_ref as R
  from nerf_hugs_amd.internal import coord, render
  g = torch.Generator().manual_seed(0)
  N, S = 16, 32
  td = torch.sort(torch.rand(N, S + 1, generator=g) + 0.1, -1).values
  dens = torch.rand(N, S, generator=g) * 20; rgb = torch.rand(N, S, 3, generator=g); d = torch.randn(N, 3, generator=g)
  far = td[:, -1:] + 1
  w = render.compute_alpha_weights(dens.cuda(), td.cuda(), d.cuda(), True)
  wo = R.compute_alpha_weights(dens, td, d, True)[0]
  assert float((w.cpu() - wo).abs().max()) < 1e-5
  out, w2 = render.volumetric_rendering(rgb.cuda(), dens.cuda(), td.cuda(), d.cuda(), 1.0, far.cuda(), True, True)
  ro = R.volumetric_rendering(rgb, wo, td, 1.0, far, True)
  for k, v in ro.items():
    assert float((out[k].cpu() - v).abs().max()) < 2e-4, k
  v = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
  assert float((coord.pos_enc(v.cuda(), 0, 4).cpu() - R.pos_enc(v, 0, 4)).abs().max()) < 1e-5
  with pytest.raises(ValueError):
    render.cast_rays_ipe(td.cuda(), d.cuda(), d.cuda(), torch.ones(N, 1).cuda(), 'sphere', torch.eye(3).cuda(), 4)


@pytest.mark.parametrize('n_rays', [1, 35, 131])
def test_ragged_ray_counts_render_like_the_oracle(n_rays):
  """The reference renders any ray count (the last chunk of an odd-sized image); the GEMM tiles want rays x samples
  in multiples of 128, so Model.apply pads internally: outputs for 1 / 35 / 131 rays still match the oracle."""
  from tests import hugs_testlib as H
  from oracle import torch_ref as R
  from nerf_hugs_amd.internal import models, utils
  config, model, state, render_fn, _, cfg, oparams = H.make_pair(GIN)
  full = H.synth_rays(3, 8, 21)                          # 192 rays
  rays = full.rays.map(lambda x: x.reshape(-1, x.shape[-1])[:n_rays])
  rend, hist = model.apply(state.flat, None, rays, 0.7, True)
  orays = {k: v[:n_rays] for k, v in H.oracle_rays(full).items()}
  orend, ohist = R.model_forward(cfg, oparams, orays, 0.7, None, True)
  assert rend[-1]['rgb'].shape == (n_rays, 3) and hist[-1]['weights'].shape == (n_rays, 32)
  for k in ['rgb', 'acc', 'distance_mean', 'distance_median']:
    a = rend[-1][k].cpu().reshape(n_rays, -1).double(); b = orend[-1][k].detach().reshape(n_rays, -1).double()
    assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(b.abs().max())), k
  assert float((hist[0]['sdist'].cpu() - ohist[0]['sdist']).abs().max()) < 1e-5
  # an odd-sized image through render_image: 7 x 5 pixels, chunks of 16 -> 16, 16, 3
  img = full.rays.map(lambda x: x.reshape(-1, x.shape[-1])[:35].reshape(7, 5, -1))
  configs_chunk = config.render_chunk_size
  config.render_chunk_size = 16
  try:
    out = models.render_image(functools.partial(render_fn, state.params, 0.7), img, None, config, verbose=False)
  finally:
    config.render_chunk_size = configs_chunk
  o35, _ = R.model_forward(cfg, oparams, {k: v[:35] for k, v in H.oracle_rays(full).items()}, 0.7, None, True)
  assert out['rgb'].shape == (7, 5, 3)
  assert float((out['rgb'].cpu().reshape(35, 3) - o35[-1]['rgb'].detach()).abs().max()) < 2e-4
  with pytest.raises(ValueError):
    model.apply(state.flat, None, rays.map(lambda x: x[:0]), 0.7, False)


@pytest.mark.parametrize('tt,T', [('hanerf', 16), ('nerfw', 16)])
def test_finetune_stage_of_transient_models(tt, T):
  """train.py:97-109 on a HA-NeRF / NeRF-W model: the finetune loss is the plain data loss, so the transient branches get
  zero gradient, only the embedding tables are trainable, and every statistic stays finite."""
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import train_utils
  gin = GIN + [f"Config.transient_type = '{tt}'", f"Model.num_transient_features = {T}", "Config.finetune_enable = True",
               "NerfMLP.bottleneck_width = 128"]
  config, model, state, _, train_step, cfg, oparams = H.make_pair(gin)
  batch = H.synth_rays(2, 8, 3)
  gen = torch.Generator(device='cuda').manual_seed(0)
  state, _, gen = train_step(gen, state, batch, 0.5, None)          # one regular step first (fills every gradient slot)
  fstate, ftrain, _ = train_utils.setup_finetune_model(config, model, state)
  before = fstate.flat.clone()
  for _ in range(2):
    fstate, stats, gen = ftrain(gen, fstate, batch, 1.0, None)
  torch.cuda.synchronize()
  lay = model.layout
  moved = set()
  for lf in lay.leaves:
    d = (lay.view(fstate.flat, lf['path']) - lay.view(before, lf['path'])).abs().max().item()
    if d > 0:
      moved.add(lf['path'][0])
      assert 'embedding' in lf['path'], lf['path']
  assert moved == {'GloEmbed_0'}            # TransientEmbed rows get exactly zero gradient in this stage
  for k in ('grad_norms', 'weight_l2s', 'opt_update_norms'):
    assert all(np.isfinite(float(v)) for v in stats[k].values()), k
  assert np.isfinite(float(stats['loss'])) and set(stats['losses'].keys()) == {'data'}

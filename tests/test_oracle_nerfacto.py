"""oracle/nerfacto_ref.py against vectors recorded by importing the reference's own nerfacto/utils/{ray_utils,
loss_utils,lr_scheduler_utils}.py and models/custom_functions.py (tests/golden/ref_nerfacto.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import nerfacto_ref as NF

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def z():
  return np.load(os.path.join(HERE, 'golden', 'ref_nerfacto.npz'))


T = lambda a: torch.from_numpy(np.array(a))


@pytest.mark.parametrize('tag', ['l0', 'l1', 'l2'])
def test_sample_intervals(z, tag):
  """ray_utils.py:112-231 incl. a zero-width bin, an all-zero-weight ray (logits forced to 1) and half-empty rays."""
  bins, w = T(z[f'samp/{tag}/bins']), T(z[f'samp/{tag}/w'])
  a, p, ns = float(z[f'samp/{tag}/anneal']), float(z[f'samp/{tag}/pad']), int(z[f'samp/{tag}/ns'])
  np.testing.assert_allclose(NF.sample_intervals(bins, w, a, p, ns, None, True, (0., 1.)).numpy(), z[f'samp/{tag}/det'], rtol=0, atol=1e-7)
  np.testing.assert_allclose(NF.sample_intervals(bins, w, a, p, ns, T(z[f'samp/{tag}/u01']), True, (0., 1.)).numpy(),
                             z[f'samp/{tag}/jit'], rtol=0, atol=1e-7)


@pytest.mark.parametrize('ob', [0, 1])
def test_density_to_weight_and_render(z, ob):
  """ray_utils.py:234-257 (deltas from the FIRST edge), :300-314, :340-347."""
  eb, dens, dirs, rgb, bg = (T(z[f'w/{k}']) for k in ('ebins', 'dens', 'dirs', 'rgb', 'bg'))
  w, a, t = NF.density_to_weight(dens, eb, dirs, bool(ob))
  np.testing.assert_allclose(w.numpy(), z[f'w/ob{ob}/weights'], rtol=1e-6, atol=1e-8)
  np.testing.assert_allclose(a.numpy(), z[f'w/ob{ob}/alphas'], rtol=1e-6, atol=1e-8)
  np.testing.assert_allclose(t.numpy(), z[f'w/ob{ob}/trans'], rtol=1e-6, atol=1e-8)
  np.testing.assert_allclose(NF.render_features(w, rgb, bg).numpy(), z[f'w/ob{ob}/rgb'], rtol=1e-6, atol=1e-7)
  np.testing.assert_allclose(NF.render_depth(w, eb).numpy(), z[f'w/ob{ob}/depth'], rtol=1e-6, atol=1e-7)
  # the quirk is real: interval-width deltas (what MipNeRF360/internal/render.py uses) give different weights
  widths = torch.cat([eb[:, :1], eb[:, :1] + torch.cumsum(eb[:, 1:] - eb[:, :1], -1)], -1)
  assert not np.allclose(NF.density_to_weight(dens, widths, dirs, bool(ob))[0].numpy(), w.numpy(), atol=1e-3)


def test_losses_and_gradients(z):
  """loss_utils.py:7-86: lossfun_outer, interlevel_loss (+ its gradient to both proposal histograms), distortion."""
  c, w, cp, cp2 = (T(z[f'loss/{k}']) for k in ('c', 'w', 'cp', 'cp2'))
  wp, wp2 = T(z['loss/wp']).requires_grad_(True), T(z['loss/wp2']).requires_grad_(True)
  np.testing.assert_allclose(NF.lossfun_outer(c, w, cp, wp).detach().numpy(), z['loss/lossfun_outer'], rtol=1e-5, atol=1e-9)
  il = NF.interlevel_loss([wp2, wp, w], [cp2, cp, c])
  assert abs(float(il) - float(z['loss/interlevel'])) <= 1e-6 * abs(float(il))
  il.backward()
  np.testing.assert_allclose(wp.grad.numpy(), z['loss/d_wp'], rtol=1e-5, atol=1e-9)
  np.testing.assert_allclose(wp2.grad.numpy(), z['loss/d_wp2'], rtol=1e-5, atol=1e-9)
  wd = w.clone().requires_grad_(True)
  ld = NF.lossfun_distortion(c, wd)
  np.testing.assert_allclose(ld.detach().numpy(), z['loss/distortion'], rtol=1e-6)
  ld.mean().backward()
  np.testing.assert_allclose(wd.grad.numpy(), z['loss/d_w_distortion'], rtol=1e-5, atol=1e-9)


def test_custom_functions_and_lr(z):
  """custom_functions.py:17-24 contraction, :38-52 trunc_exp with its clipped gradient; lr_scheduler_utils.py:6-27."""
  np.testing.assert_allclose(NF.spatial_distortion_norm2(T(z['cf/x'])).numpy(), z['cf/contract'], rtol=1e-6, atol=1e-7)
  r = T(z['cf/raw']).requires_grad_(True)
  y = NF.trunc_exp(r)
  np.testing.assert_allclose(y.detach().numpy(), z['cf/trunc_exp'], rtol=1e-6)
  y.sum().backward()
  np.testing.assert_allclose(r.grad.numpy(), z['cf/trunc_exp_grad'], rtol=1e-6)
  f = [NF.lr_factor(int(s), 1e-2, 1e-3, 1e-8, 500, 25000) for s in z['lr/steps']]
  np.testing.assert_allclose(f, z['lr/factor'], rtol=1e-12)


def test_forward_rays_runs_and_has_gradients():
  """The wiring (nerfacto.py:286-414): shapes, weights sum to 1 with an opaque background, every parameter gets a
  gradient, zero-density fallbacks.  (The fields' numbers are parity-unpinned: tiny-cuda-nn.)"""
  cfg = NF.Cfg(num_levels=4, max_res=64, log2_hashmap_size=10, hidden_dim=16, geo_feat_dim=7, hidden_dim_color=16,
               num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8, opaque_background=True,
               use_appearance_embedding=True, appearance_embedding_dim=5, num_embedding=4,
               proposal_net_args_list=[dict(hidden_dim=8, log2_hashmap_size=9, num_levels=3, max_res=32)])
  P = NF.init_params(cfg, 1)
  for grp in P.values():
    for v in (grp.values() if isinstance(grp, dict) else [grp]):
      v.requires_grad_(True)
  g = torch.Generator().manual_seed(0)
  N = 6
  d = torch.randn(N, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
  rays = dict(origin=torch.randn(N, 3, generator=g) * 0.3, direction=d, viewdir=d, near=torch.full((N, 1), 0.05),
              far=torch.full((N, 1), 3.0), embed_idx=torch.randint(0, 4, (N, 1), generator=g), bg_rgb=torch.ones(N, 3))
  out = NF.forward_rays(cfg, P, rays, 100, [torch.rand(N, 1, generator=g) for _ in range(3)])
  assert out['rgb'].shape == (N, 3) and [w.shape[1] for w in out['weights_list']] == [32, 16, 8]
  for w in out['weights_list']:
    np.testing.assert_allclose(w.sum(-1).detach().numpy(), 1.0, atol=1e-5)
  loss, info = NF.loss_fn(cfg, out, torch.rand(N, 3, generator=g))
  loss.backward()
  assert set(info) == {'mse', 'rgb_loss', 'interlevel_loss', 'distortion_loss'}
  for name, grp in P.items():
    for k, v in (grp.items() if isinstance(grp, dict) else [('emb', grp)]):
      assert v.grad is not None and torch.isfinite(v.grad).all(), (name, k)
      if k != 'emb':
        assert float(v.grad.abs().max()) > 0, (name, k)


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_robustnerf_mask(z, tag):
  """utils/loss_utils.py:88-150 executed by the reference: odd and even box filters, first-step and fed-back threshold."""
  f, q, thr = z[f'robust/{tag}/cfg']
  cfg = NF.Cfg(transient_type='robustnerf', robustnerf_smoothed_filter_size=int(f), robustnerf_inlier_quantile=float(q))
  mask, info = NF.get_robustnerf_mask(cfg, torch.from_numpy(z[f'robust/{tag}/errors']), 1.0 if thr < 0 else float(thr))
  assert np.array_equal(mask.numpy(), z[f'robust/{tag}/mask'])
  got = np.array([float(info[k]) for k in ('inlier_threshold', 'is_inlier_loss', 'has_inlier_neighbors', 'is_inlier_patch', 'robust_mask')])
  np.testing.assert_allclose(got, z[f'robust/{tag}/info'], rtol=1e-6, atol=0)
  assert abs(got[0] - float(z[f'robust/{tag}/next_thr'])) <= 1e-6 * got[0]

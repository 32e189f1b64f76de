"""Golden vectors for the pure-torch half of the NERFACTO path, recorded by IMPORTING THE REFERENCE'S OWN modules
(/root/reference/nerfacto/utils/{ray_utils,loss_utils,lr_scheduler_utils}.py, models/custom_functions.py -- plain
PyTorch, which is installed here).  Build container only; only data is committed (tests/golden/ref_nerfacto.npz).

    python tests/golden/gen_nerfacto_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/nerfacto'


def main():
  if not os.path.isdir(REF):
    raise SystemExit('needs the reference checkout at ' + REF)
  sys.path.insert(0, REF)
  import warnings
  warnings.filterwarnings('ignore')
  from utils import ray_utils, loss_utils, lr_scheduler_utils
  import importlib.util      # models/__init__.py imports tinycudann (absent): load custom_functions.py by path
  spec = importlib.util.spec_from_file_location('nf_custom_functions', os.path.join(REF, 'models', 'custom_functions.py'))
  cf = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(cf)
  g = torch.Generator().manual_seed(7)
  out = {}
  T = lambda a: np.asarray(a.detach().numpy())
  # ---- sampler: level 0 (one unit interval), level > 0 (histogram incl. zero-width / zero-weight bins), both modes ----
  N = 24
  bins0 = torch.cat([torch.zeros(N, 1), torch.ones(N, 1)], -1)
  w0 = torch.ones(N, 1)
  real_rand = torch.rand
  for tag, bins, w, ns, anneal, pad in (('l0', bins0, w0, 64, 0.31, 0.01), ('l1', None, None, 48, 0.77, 0.005), ('l2', None, None, 96, 1.0, 0.01)):
    if bins is None:
      nb = 64 if tag == 'l1' else 200
      bins = torch.sort(torch.rand(N, nb + 1, generator=g), -1).values
      bins[:, 0], bins[:, -1] = 0., 1.
      bins[3, 10] = bins[3, 9]                       # a zero-width bin
      w = torch.rand(N, nb, generator=g)**3
      w[5] = 0.                                       # an all-zero ray
      w[7, :nb // 2] = 0.
    out[f'samp/{tag}/bins'], out[f'samp/{tag}/w'] = T(bins), T(w)
    out[f'samp/{tag}/anneal'], out[f'samp/{tag}/pad'], out[f'samp/{tag}/ns'] = np.float64(anneal), np.float64(pad), np.int64(ns)
    out[f'samp/{tag}/det'] = T(ray_utils.sample_intervals(bins, w, anneal, pad, ns, False, True, (0., 1.)))
    draws = []

    def rec_rand(*a, **k):
      r = real_rand(*a, **{kk: vv for kk, vv in k.items() if kk != 'device'}, generator=g)
      draws.append(r)
      return r
    torch.rand = rec_rand
    try:
      out[f'samp/{tag}/jit'] = T(ray_utils.sample_intervals(bins, w, anneal, pad, ns, True, True, (0., 1.)))
    finally:
      torch.rand = real_rand
    out[f'samp/{tag}/u01'] = T(draws[0])
  # ---- density_to_weight (first-edge deltas), render_features / depth -----------------------------------------------
  S = 40
  eb = torch.sort(torch.rand(N, S + 1, generator=g) * 3 + 0.1, -1).values
  dens = torch.rand(N, S, generator=g) * 5
  dens[2] = 0.
  dirs = torch.randn(N, 3, generator=g)
  rgb = torch.rand(N, S, 3, generator=g)
  bg = torch.rand(N, 3, generator=g)
  out['w/ebins'], out['w/dens'], out['w/dirs'], out['w/rgb'], out['w/bg'] = T(eb), T(dens), T(dirs), T(rgb), T(bg)
  for ob in (False, True):
    w_, a_, t_ = ray_utils.density_to_weight(dens, eb, dirs, ob)
    out[f'w/ob{int(ob)}/weights'], out[f'w/ob{int(ob)}/alphas'], out[f'w/ob{int(ob)}/trans'] = T(w_), T(a_), T(t_)
    out[f'w/ob{int(ob)}/rgb'] = T(ray_utils.render_features(w_, rgb, bg, False))
    out[f'w/ob{int(ob)}/depth'] = T(ray_utils.render_depth(w_, eb))
  # ---- losses -------------------------------------------------------------------------------------------------------------
  c = torch.sort(torch.rand(N, 49, generator=g), -1).values; c[:, 0], c[:, -1] = 0., 1.
  w = torch.rand(N, 48, generator=g); w = w / w.sum(-1, keepdim=True)
  cp = torch.sort(torch.rand(N, 97, generator=g), -1).values; cp[:, 0], cp[:, -1] = 0., 1.
  wp = (torch.rand(N, 96, generator=g) * 0.02).requires_grad_(True)
  cp2 = torch.sort(torch.rand(N, 257, generator=g), -1).values; cp2[:, 0], cp2[:, -1] = 0., 1.
  wp2 = (torch.rand(N, 256, generator=g) * 0.008).requires_grad_(True)
  lo = loss_utils.lossfun_outer(c, w, cp, wp)
  il = loss_utils.interlevel_loss([wp2, wp, w], [cp2, cp, c])
  il.backward()
  out['loss/c'], out['loss/w'], out['loss/cp'], out['loss/wp'], out['loss/cp2'], out['loss/wp2'] = T(c), T(w), T(cp), T(wp), T(cp2), T(wp2)
  out['loss/lossfun_outer'], out['loss/interlevel'] = T(lo), np.float64(il)
  out['loss/d_wp'], out['loss/d_wp2'] = T(wp.grad), T(wp2.grad)
  wd = w.clone().requires_grad_(True)
  ld = loss_utils.lossfun_distortion(c, wd)
  ld.mean().backward()
  out['loss/distortion'], out['loss/d_w_distortion'] = T(ld), T(wd.grad)
  # ---- custom functions -------------------------------------------------------------------------------------------------
  x = torch.cat([torch.randn(40, 3, generator=g) * 3, torch.randn(20, 3, generator=g) * 0.3, torch.zeros(1, 3)])
  out['cf/x'], out['cf/contract'] = T(x), T(cf.spatial_distortion_norm2(x))
  r = (torch.randn(50, generator=g) * 8).requires_grad_(True)
  y = cf.trunc_exp(r)
  y.sum().backward()
  out['cf/raw'], out['cf/trunc_exp'], out['cf/trunc_exp_grad'] = T(r), T(y), T(r.grad)
  # ---- lr schedule ---------------------------------------------------------------------------------------------------------
  opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-2)
  sch = lr_scheduler_utils.get_warmup_decay_scheduler(opt, 1e-2, 1e-3, 1e-8, 500, 25000)
  steps = [0, 1, 100, 499, 500, 501, 10000, 24999, 25000, 30000]
  out['lr/steps'] = np.array(steps)
  out['lr/factor'] = np.array([sch.lr_lambdas[0](s) for s in steps])
  # ---- get_robustnerf_mask (utils/loss_utils.py:88-150): default filter 3 and an even one, first step (no threshold in
  # extra_infos -> 1.0) and a fed-back threshold; errors shaped so that every criterion fires somewhere ------------------------
  for tag, f, q, thr in (('a', 3, 0.8, None), ('b', 3, 0.6, 0.02), ('c', 4, 0.8, 0.05), ('d', 5, 0.5, 0.004)):
    errs = torch.rand(6, 16, 16, 3, generator=g)**4 * 0.3
    errs[1] *= 0.02; errs[2, 4:12, 4:12] *= 0.01; errs[3] += 0.2
    extra = {} if thr is None else {'inlier_threshold': thr}
    mask, info, extra2 = loss_utils.get_robustnerf_mask(errors=errs, suffix=None, extra_infos=dict(extra), inlier_quantile=q,
                                                        smoothed_filter_size=f, smoothed_inlier_quantile=0.5, inner_patch_size=8,
                                                        inner_patch_inlier_quantile=0.4)
    out[f'robust/{tag}/errors'], out[f'robust/{tag}/mask'] = T(errs), T(mask)
    out[f'robust/{tag}/cfg'] = np.array([f, q, -1.0 if thr is None else thr])
    out[f'robust/{tag}/info'] = np.array([float(info[k]) for k in ('inlier_threshold', 'is_inlier_loss', 'has_inlier_neighbors',
                                                                    'is_inlier_patch', 'robust_mask')])
    out[f'robust/{tag}/next_thr'] = np.float64(extra2['inlier_threshold'])
  np.savez_compressed(os.path.join(HERE, 'ref_nerfacto.npz'), **out)
  print('wrote ref_nerfacto.npz', len(out), 'arrays')


if __name__ == '__main__':
  main()

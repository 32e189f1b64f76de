"""Generate golden input/output vectors by EXECUTING THE REFERENCE'S OWN SOURCE.

Runs only in the build container (needs /root/reference; refuses otherwise).
It installs the numpy-backed `jax` stand-in (`_jax_standin.py`), imports
`MipNeRF360/internal/{math,stepfun,render,coord,geopoly}.py` unmodified from
/root/reference and records small (N<=16 rays) input/output vectors as .npz.
Only DATA is committed; no reference source, bytecode or pickled function.

    python tests/golden/gen_fixtures.py        # rewrites tests/golden/*.npz

Reference entry points exercised (file:line under /root/reference/MipNeRF360):
  internal/stepfun.py:30 searchsorted, :64 inner_outer, :80 lossfun_outer,
  :99 max_dilate, :112 max_dilate_weights, :131 integrate_weights,
  :153 invert_cdf, :164 sample, :214 sample_intervals, :266 lossfun_distortion,
  :298 weighted_percentile
  internal/render.py:103 cast_rays, :130 compute_alpha_weights,
  :185 volumetric_rendering
  internal/coord.py:21 contract, :39 track_linearize, :63 construct_ray_warps,
  :107 integrated_pos_enc, :129 lift_and_diagonalize, :136 pos_enc
  internal/geopoly.py:78 generate_basis
  internal/math.py:66 learning_rate_decay, :108 sorted_interp
The level loop that strings them together follows internal/models.py:145-272
(models.py itself needs flax/gin and cannot be imported).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/MipNeRF360'
f32 = np.float32


def main():
  if not os.path.isdir(REF):
    raise SystemExit('gen_fixtures.py needs the reference checkout at ' + REF)
  sys.path.insert(0, HERE)
  import _jax_standin
  jax = _jax_standin.install()
  jnp = jax.numpy
  sys.path.insert(0, REF)
  from internal import coord, geopoly, math as rmath, render, stepfun

  rng = np.random.default_rng(20200823)

  def rays(n):
    o = (rng.normal(size=(n, 3)) * 0.5).astype(f32)
    d = rng.normal(size=(n, 3))
    d = d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.8, 1.2, (n, 1))
    d = d.astype(f32)
    v = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(f32)
    radii = rng.uniform(5e-4, 2e-3, (n, 1)).astype(f32)
    return o, d, v, radii

  out = {}

  # ---- geopoly / lr / psnr-free constants ---------------------------------
  basis = geopoly.generate_basis('icosahedron', 2)  # [21,3]; models.py:393-396 uses .T
  out['basis_ico2'] = np.asarray(basis, np.float64)
  out['basis_octa1'] = np.asarray(geopoly.generate_basis('octahedron', 1), np.float64)
  steps = np.array([0, 1, 10, 256, 511, 512, 1000, 125000, 249999, 250000], np.float64)
  out['lr_steps'] = steps
  out['lr_vals'] = np.array([
      float(rmath.learning_rate_decay(s, 2e-3, 2e-5, 250000, 512, 0.01)) for s in steps])
  pos_basis_t = np.asarray(basis, np.float64).T.astype(f32)  # [3,21]

  # ---- level loop, following models.py:145-272 ---------------------------
  def level_loop(tag, n, num_levels, s_prop, s_nerf, raydist, near, far, jitter_seed,
                 train_frac, opaque_bg, warp, ray_shape='cone'):
    o, d, v, radii = rays(n)
    near_a = np.full((n, 1), near, f32) if np.isscalar(near) else near
    far_a = np.full((n, 1), far, f32) if np.isscalar(far) else far
    fn = None if raydist is None else jnp.reciprocal
    _, s_to_t = coord.construct_ray_warps(fn, near_a, far_a)
    sdist = np.concatenate([np.zeros_like(near_a), np.ones_like(far_a)], -1)
    weights = np.ones_like(near_a)
    prod = 1
    rec = dict(o=o, d=d, viewdirs=v, radii=radii, near=near_a, far=far_a,
               train_frac=np.float64(train_frac))
    key = None if jitter_seed is None else _jax_standin._Key(jitter_seed)
    for lvl in range(num_levels):
      is_prop = lvl < num_levels - 1
      S = s_prop if is_prop else s_nerf
      dilation = 0.0025 + 0.5 * (1. - 0.) / prod
      prod *= S
      rec[f'l{lvl}_in_sdist'] = sdist.copy()
      rec[f'l{lvl}_in_weights'] = weights.copy()
      if lvl > 0:
        sdist, weights = stepfun.max_dilate_weights(
            sdist, weights, dilation, domain=(0., 1.), renormalize=True)
        rec[f'l{lvl}_dil_t_full'] = sdist.copy()
        rec[f'l{lvl}_dil_w_full'] = weights.copy()
        sdist = sdist[..., 1:-1]
        weights = weights[..., 1:-1]
      rec[f'l{lvl}_dilation'] = np.float64(dilation)
      anneal = (10 * train_frac) / ((10 - 1) * train_frac + 1)
      with np.errstate(divide='ignore'):
        logits = np.where(sdist[..., 1:] > sdist[..., :-1],
                          f32(anneal) * np.log(weights + f32(0.0)), -np.inf).astype(f32)
      rec[f'l{lvl}_logits'] = logits
      ndraw = 0 if key is None else len(key.draws)
      sdist = stepfun.sample_intervals(key, sdist, logits, S, single_jitter=True,
                                       domain=(0., 1.))
      if key is not None:
        rec[f'l{lvl}_u01'] = key.draws[ndraw]
      rec[f'l{lvl}_sdist'] = sdist.copy()
      tdist = s_to_t(sdist)
      rec[f'l{lvl}_tdist'] = tdist.copy()
      means, covs = render.cast_rays(tdist, o, d, radii, ray_shape, diag=False)
      rec[f'l{lvl}_means'] = means
      rec[f'l{lvl}_covs'] = covs
      if warp:
        means, covs = coord.track_linearize(coord.contract, means, covs)
        rec[f'l{lvl}_wmeans'] = means
        rec[f'l{lvl}_wcovs'] = covs
      lm, lv = coord.lift_and_diagonalize(means, covs, pos_basis_t)
      rec[f'l{lvl}_lift_mean'] = lm
      rec[f'l{lvl}_lift_var'] = lv
      # subsampled (rays[:2], every 3rd sample) to keep the committed fixture small
      rec[f'l{lvl}_ipe_sub'] = coord.integrated_pos_enc(lm, lv, 0, 12)[:2, ::3]
      # Stand-in "MLP": a smooth random density / colour field (the flax MLP is
      # not importable); it only has to feed the compositing functions.
      cen = means.mean(-2, keepdims=True)
      density = (np.exp(-4 * np.sum((means - cen)**2, -1)) *
                 rng.uniform(0., 60., means.shape[:-1])).astype(f32)
      rgb = rng.uniform(0, 1, means.shape).astype(f32)
      rec[f'l{lvl}_density'] = density
      rec[f'l{lvl}_rgb'] = rgb
      weights, alpha, trans = render.compute_alpha_weights(
          density, tdist, d, opaque_background=opaque_bg)
      rec[f'l{lvl}_weights'] = weights
      rec[f'l{lvl}_alpha'] = alpha
      rec[f'l{lvl}_trans'] = trans
      rend = render.volumetric_rendering(rgb, weights, tdist, 1.0, far_a, True)
      for k2, v2 in rend.items():
        rec[f'l{lvl}_rend_{k2}'] = np.asarray(v2, f32)
    # losses on the history (train_utils.py:228-248 call pattern)
    c, w = rec[f'l{num_levels-1}_sdist'], rec[f'l{num_levels-1}_weights']
    for lvl in range(num_levels - 1):
      cp, wp = rec[f'l{lvl}_sdist'], rec[f'l{lvl}_weights']
      rec[f'l{lvl}_lossfun_outer'] = stepfun.lossfun_outer(c, w, cp, wp)
      inner, outer = stepfun.inner_outer(c, cp, wp)
      rec[f'l{lvl}_w_inner'] = inner
      rec[f'l{lvl}_w_outer'] = outer
      lo, hi = stepfun.searchsorted(cp, c)
      rec[f'l{lvl}_idx_lo'] = lo.astype(np.int32)
      rec[f'l{lvl}_idx_hi'] = hi.astype(np.int32)
    rec['lossfun_distortion'] = stepfun.lossfun_distortion(c, w)
    rec['dir_enc'] = coord.pos_enc(v, 0, 4, True)
    for k2, v2 in rec.items():
      out[f'{tag}/{k2}'] = v2

  # cfg2 shape (64 + 128), linear spacing, no warp, deterministic and jittered
  level_loop('cfg2_det', 4, 2, 64, 128, None, 0.1, 1.2, None, 0.5, True, False)
  level_loop('cfg2_jit', 4, 2, 64, 128, None, 0.1, 1.2, 7, 0.5, True, False)
  # reference default (64,64,32), reciprocal spacing + contract (360.gin / distractor)
  nr = rng.uniform(0.05, 0.3, (4, 1)).astype(f32)
  level_loop('def3_warp', 4, 3, 64, 64, 'reciprocal', nr, 1e6, 11, 0.25, True, True)
  level_loop('def3_nobg', 4, 3, 64, 64, None, 2., 6., 3, 1.0, False, False)
  level_loop('cyl', 4, 2, 64, 64, None, 2., 6., None, 0.7, False, False, ray_shape='cylinder')

  # ---- known-answer vector from the reference's own test -------------------
  # tests/stepfun_test.py:579-586: logits [0,0,100,0,0] on t=[1..6] -> linspace(3,4,11)
  t = np.array([1, 2, 3, 4, 5, 6], f32)
  lg = np.array([0, 0, 100, 0, 0], f32)
  out['kat_linspace/t'] = t
  out['kat_linspace/logits'] = lg
  out['kat_linspace/out'] = stepfun.sample_intervals(None, t, lg, 10, single_jitter=False,
                                                     domain=(1., 6.))
  # ---- weighted percentile / sorted_interp standalone -----------------------
  tt = np.sort(rng.uniform(0, 1, (5, 33)).astype(f32), -1)
  ww = rng.uniform(0, 1, (5, 32)).astype(f32)
  ww /= ww.sum(-1, keepdims=True)
  out['wp/t'] = tt
  out['wp/w'] = ww
  out['wp/out'] = stepfun.weighted_percentile(tt, ww, [5, 50, 95])
  # contract standalone incl. points inside/outside/at 0 (coord.py:21-27)
  x = np.concatenate([rng.normal(size=(20, 3)) * 3, rng.normal(size=(10, 3)) * 0.2,
                      np.zeros((1, 3))]).astype(f32)
  out['contract/x'] = x
  out['contract/z'] = coord.contract(x)

  np.savez_compressed(os.path.join(HERE, 'ref_leaves.npz'), **out)
  tot = sum(v.nbytes for v in out.values())
  print(f'wrote ref_leaves.npz: {len(out)} arrays, {tot/1e6:.2f} MB raw')


if __name__ == '__main__':
  main()

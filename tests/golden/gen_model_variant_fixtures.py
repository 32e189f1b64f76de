"""Golden forward passes of the reference's own Model.__call__ (MipNeRF360/internal/models.py:74-330, executed under the numpy
stand-ins exactly as tests/golden/gen_model_fixtures.py does) for OPTION VARIANTS the five main cases do not exercise: near-plane
annealing, one jitter draw per sample, cylinder ray shape, four sampling levels, non-default sampler / head / encoding knobs, a model
without a view layer, a deeper view MLP, a grey non-opaque background.  Training-mode forward only (jittered; the draws are recorded):
per-level sdist, weights, density, rgb and renderings.  Writes tests/golden/ref_model_variants.npz (data only).

    python tests/golden/gen_model_variant_fixtures.py        # ~1 min
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_model_fixtures as G

f32 = np.float32
M = lambda **kw: dict(G.BASE_MODEL, num_levels=2, num_prop_samples=32, num_nerf_samples=48, **kw)
CASES = {
    'near_anneal': dict(Model=M(near_anneal_rate=0.5), NerfMLP=G.SMALL_NERF, PropMLP=G.SMALL_PROP),
    'per_sample_jitter': dict(Model=M(single_jitter=False), NerfMLP=G.SMALL_NERF, PropMLP=G.SMALL_PROP),
    'cylinder': dict(Model=M(ray_shape='cylinder'), NerfMLP=G.SMALL_NERF, PropMLP=G.SMALL_PROP),
    'levels4': dict(Model=dict(G.BASE_MODEL, num_levels=4, num_prop_samples=32, num_nerf_samples=16), NerfMLP=G.SMALL_NERF, PropMLP=G.SMALL_PROP),
    'sampler_knobs': dict(Model=M(anneal_slope=3, dilation_multiplier=0.3, dilation_bias=0.01, resample_padding=0.01),
                          NerfMLP=G.SMALL_NERF, PropMLP=G.SMALL_PROP),
    'head_knobs': dict(Model=M(num_glo_features=4),
                       NerfMLP=dict(G.SMALL_NERF, deg_view=2, density_bias=0.0, rgb_padding=0.01, skip_layer=2, max_deg_point=8),
                       # (a 128-wide PropMLP: the HIP path does not build a skip concat on a trunk narrower than one MFMA tile)
                       PropMLP=dict(G.SMALL_PROP, net_width=128, skip_layer=2, max_deg_point=8, density_bias=0.0)),
    'no_viewdirs': dict(Model=M(use_viewdirs=False, num_glo_features=4), NerfMLP=G.SMALL_NERF, PropMLP=G.SMALL_PROP),
    'view_depth3': dict(Model=M(), NerfMLP=dict(G.SMALL_NERF, net_depth_viewdirs=3), PropMLP=G.SMALL_PROP),
    'grey_background': dict(Model=dict(M(bg_intensity_range=(0.5, 0.5)), opaque_background=False), NerfMLP=G.SMALL_NERF, PropMLP=G.SMALL_PROP),
    'log_raydist_contract': dict(Model=M(raydist_fn='@jnp.log'), NerfMLP=dict(G.SMALL_NERF, warp_fn='@coord.contract'),
                                 PropMLP=dict(G.SMALL_PROP, warp_fn='@coord.contract')),
}
TRAIN_FRAC = 0.37


def main():
  if not os.path.isdir(G.REF):
    raise SystemExit('needs the reference checkout at ' + G.REF)
  import _jax_standin
  import _flax_standin as F
  jax = F.install({})
  jnp = jax.numpy
  sys.path.insert(0, G.REF)
  from internal import configs, coord, geopoly, models, utils
  _gen_basis = geopoly.generate_basis
  geopoly.generate_basis = lambda *a, **k: np.asarray(_gen_basis(*a, **k)).astype(f32)      # (as gen_model_fixtures: models.py:393-396)
  def log(x):      # (coord.py:94 looks the inverse up by fn.__name__: the stand-in's own log is an anonymous wrapper)
    return jnp.log(x)
  resolve = {'@jnp.reciprocal': jnp.reciprocal, '@coord.contract': coord.contract, '@jnp.log': log}
  out = {}
  for case, spec in CASES.items():
    bind = {k: {kk: resolve.get(vv, vv) if isinstance(vv, str) else vv for kk, vv in spec[k].items()} for k in ('Model', 'NerfMLP', 'PropMLP')}
    F.set_bindings(bind)
    out[f'{case}/spec'] = np.array(json.dumps(spec))
    rng = np.random.default_rng(sum(map(ord, case)))
    config = configs.Config()
    model, variables = models.construct_model(jax.random.PRNGKey(17), utils.dummy_rays(), config)
    flat = G.flatten(variables['params'])
    for k in flat:
      if k.endswith('/kernel'):
        b = flat[k].view(np.uint32).astype(np.uint64)
        b = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint32) << 16
        flat[k] = b.view(f32).reshape(flat[k].shape)
      if k.endswith('/bias'):
        flat[k] = (rng.normal(size=flat[k].shape) * 0.1).astype(f32)
      if k.endswith('/embedding'):
        flat[k] = (rng.normal(size=flat[k].shape) * 0.5).astype(f32)
    variables = {'params': G.unflatten(flat)}
    for k, v in flat.items():
      if k.endswith('/kernel'):
        out[f'{case}/params_bf16/{k}'] = (v.view(np.uint32) >> 16).astype(np.uint16)
      else:
        out[f'{case}/params/{k}'] = v
    shp = (2, 4, 4)
    o = (rng.normal(size=shp + (3,)) * 0.5).astype(f32)
    d = rng.normal(size=shp + (3,))
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.8, 1.2, shp + (1,))).astype(f32)
    v = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(f32)
    contract = case == 'log_raydist_contract'
    rays = utils.Rays(
        pix_coords=rng.uniform(size=shp + (2,)).astype(f32), origins=o, directions=d, viewdirs=v,
        radii=rng.uniform(5e-4, 2e-3, shp + (1,)).astype(f32), lossmult=np.ones(shp + (1,), f32), static_mask=np.ones(shp + (1,), f32),
        near=np.full(shp + (1,), 0.2 if contract else 0.1, f32), far=np.full(shp + (1,), 50.0 if contract else 1.2, f32),
        embed_idx=(rng.integers(0, 16, (2, 1, 1, 1)) * np.ones(shp + (1,))).astype(np.int32), cam_idx=np.zeros(shp + (1,), np.int32))
    for name in ('pix_coords', 'origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'static_mask', 'near', 'far', 'embed_idx', 'cam_idx'):
      out[f'{case}/rays/{name}'] = getattr(rays, name)
    key = jax.random.PRNGKey(4321)
    renderings, history = model.apply(variables, key, rays, train_frac=TRAIN_FRAC, compute_extras=False, zero_glo=False, zero_tra=False)
    L = model.num_levels
    draws = list(key.draws)
    assert len(draws) == L, (case, len(draws))
    for lvl in range(L):
      out[f'{case}/l{lvl}_u01'] = np.asarray(draws[lvl], f32).reshape(32, -1)
      out[f'{case}/l{lvl}_rend_rgb'] = np.asarray(renderings[lvl]['rgb'], f32).reshape(-1, 3)
      for k in ('sdist', 'weights', 'density', 'rgb'):
        a = np.asarray(history[lvl][k], f32)
        out[f'{case}/l{lvl}_{k}'] = a.reshape((-1,) + a.shape[3:])
    print(case, L, [out[f'{case}/l{l}_u01'].shape for l in range(L)], float(np.abs(out[f'{case}/l{L-1}_rend_rgb']).mean()))
  out['train_frac'] = np.float64(TRAIN_FRAC)
  np.savez_compressed(os.path.join(HERE, 'ref_model_variants.npz'), **out)


if __name__ == '__main__':
  main()

"""Golden vectors for the MODEL / TRAIN-STEP half of the path, recorded by EXECUTING THE
REFERENCE'S OWN `MipNeRF360/internal/models.py` and `train_utils.py` (unmodified, imported from
/root/reference) under the numpy stand-ins in `_jax_standin.py` + `_flax_standin.py`.

Runs only in the build container (refuses without /root/reference).  Only DATA is committed.

    python tests/golden/gen_model_fixtures.py      # rewrites tests/golden/ref_model.npz

Reference code executed (file:line under /root/reference/MipNeRF360/internal):
  models.py:46-330   Model.__call__ (level loop, GLO / transient embeds, dual compositing, extras)
  models.py:333-357  construct_model (parameter tree / names via Module.init)
  models.py:360-550  MLP.__call__ (trunk + skip, density head, bottleneck, view branch, rgb head,
                     NeRF-W transient branch), :651-675 ImplicitMask
  train_utils.py:72-111 compute_data_loss (both modes), :114-147 compute_robustnerf_loss,
  :150-183 compute_nerfw_loss, :186-225 compute_hanerf_loss, :228-248 interlevel / distortion,
  :251-348 robustnerf_mask, :351-369 clip_gradients, :372-480 create_train_step.train_step
  (loss assembly, weight_l2s, grad stats, clip, nan_to_num, apply_gradients, update stats, psnr)
What the stand-in supplies instead of the reference's third-party code is listed in
`_flax_standin.py`; in particular autodiff is replaced by (i) a synthetic seeded gradient tree that
flows through the reference's stats / clip / update code, and (ii) float64 central differences of
the reference's own `loss_fn` along seeded directions, with every `stop_gradient` value replayed
from the unperturbed float32 run -- the directional derivatives any correct backward must match.
optax.adam is restated, so `opt_update_*` and `new_params` are NOT pins of Adam.

Direction / gradient convention (tests rebuild them, nothing is stored): leaves in sorted order of
'/'.join(path); `np.random.default_rng(seed)`; one `standard_normal(shape)` float64 draw per leaf.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/MipNeRF360'
f32 = np.float32
HIST_KEYS = ('density', 'rgb', 'sdist', 'weights', 'density_transient', 'rgb_transient', 'uncertainty')

BASE_MODEL = {'num_embeddings': 16, 'opaque_background': True}
# sizes the HIP path accepts (128-column MFMA tiles; a 64-wide PropMLP exercises the zero-padded layout)
SMALL_NERF = {'net_depth': 8, 'net_width': 128, 'bottleneck_width': 128, 'net_width_viewdirs': 128}
SMALL_PROP = {'net_depth': 4, 'net_width': 64, 'disable_rgb': True}

CASES = {
    # cfg2's shape (64 + 128), plain mse loss, jittered
    'base': dict(
        Config={'data_loss_type': 'mse', 'distortion_loss_mult': 0.01, 'randomized': True, 'patch_size': 4},
        Model=dict(BASE_MODEL, num_levels=2, num_prop_samples=64, num_nerf_samples=128),
        NerfMLP=SMALL_NERF, PropMLP=SMALL_PROP, n_patch=2, P=4, near=0.1, far=1.2, hist_step=1),
    # cfg3: HuGS static masks, charb, GLO, the reference-default (64,64,32) three levels
    'withmask': dict(
        Config={'transient_type': 'withmask', 'distortion_loss_mult': 0.001, 'randomized': True, 'patch_size': 4,
                'withmask_transient_weight': 0.25, 'grad_max_norm': 0.002, 'grad_max_val': 0.001},
        Model=dict(BASE_MODEL, num_levels=3, num_prop_samples=64, num_nerf_samples=32, num_glo_features=48,
                   opaque_background=False),
        NerfMLP=SMALL_NERF, PropMLP=SMALL_PROP, n_patch=2, P=4, near=(0.5, 1.0), far=(1.5, 3.0), hist_step=1),
    # cfg4: RobustNeRF 0.8 + contract + reciprocal + GLO 4, 16x16 patches
    'robust': dict(
        Config={'transient_type': 'robustnerf', 'robustnerf_inlier_quantile': 0.8, 'data_loss_type': 'mse',
                'distortion_loss_mult': 0.001, 'randomized': True, 'patch_size': 16},
        Model=dict(BASE_MODEL, num_levels=2, num_prop_samples=64, num_nerf_samples=128, num_glo_features=4,
                   raydist_fn='@jnp.reciprocal', opaque_background=False),
        NerfMLP=dict(SMALL_NERF, warp_fn='@coord.contract'), PropMLP=dict(SMALL_PROP, warp_fn='@coord.contract'),
        n_patch=2, P=16, near=(0.05, 0.3), far=1e6, hist_step=16, inlier=0.35),
    'nerfw': dict(
        Config={'transient_type': 'nerfw', 'randomized': True, 'patch_size': 4, 'distortion_loss_mult': 0.001},
        Model=dict(BASE_MODEL, num_levels=2, num_prop_samples=64, num_nerf_samples=64, num_glo_features=8,
                   num_transient_features=6),
        # (rgb_premultiplier / rgb_bias, models.py:380-381,514-516,534-536: both rgb heads of this case)
        NerfMLP=dict(SMALL_NERF, net_width_transient=128, rgb_premultiplier=1.5, rgb_bias=-0.3), PropMLP=SMALL_PROP,
        n_patch=2, P=4, near=0.1, far=1.2, hist_step=1),
    'hanerf': dict(
        Config={'transient_type': 'hanerf', 'randomized': True, 'patch_size': 4, 'distortion_loss_mult': 0.001,
                'max_steps': 1000},
        Model=dict(BASE_MODEL, num_levels=2, num_prop_samples=64, num_nerf_samples=64, num_glo_features=8,
                   num_transient_features=6),
        NerfMLP=SMALL_NERF, PropMLP=SMALL_PROP,      # ImplicitMask is not gin-configurable: always 4 x 256
        n_patch=2, P=4, near=0.1, far=1.2, hist_step=1),
}
TRAIN_FRAC = 0.37
N_DIRS = 10
SYN_GRAD_SEED = 4242
FINETUNE = False     # gen_model_finetune_fixtures.py sets it: the finetune stage's optimizer + train_step instead


def flatten(tree, prefix=''):
  out = {}
  for k, v in tree.items():
    if hasattr(v, 'items'):
      out.update(flatten(v, prefix + k + '/'))
    else:
      out[prefix + k] = v
  return out


def unflatten(flat):
  tree = {}
  for k, v in flat.items():
    parts = k.split('/')
    d = tree
    for p in parts[:-1]:
      d = d.setdefault(p, {})
    d[parts[-1]] = v
  return tree


def seeded_tree(flat_params, seed, scale=1.0):
  rng = np.random.default_rng(seed)
  return {k: rng.standard_normal(flat_params[k].shape) * scale for k in sorted(flat_params)}


def main():
  if not os.path.isdir(REF):
    raise SystemExit('gen_model_fixtures.py needs the reference checkout at ' + REF)
  sys.path.insert(0, HERE)
  import _jax_standin
  import _flax_standin as F
  jax = F.install({})
  jnp = jax.numpy
  sys.path.insert(0, REF)
  from internal import configs, coord, geopoly, math as rmath, models, train_utils, utils
  # models.py:393-396 stores jnp.array(generate_basis(...)): a float32 constant under jax's disabled x64.  The
  # MLPs are re-constructed on every Model.__call__, so without this the float64 finite-difference runs would
  # see an unrounded float64 basis, i.e. a (slightly) different function from the float32 one.
  _gen_basis = geopoly.generate_basis
  geopoly.generate_basis = lambda *a, **k: np.asarray(_gen_basis(*a, **k)).astype(f32)
  resolve = {'@jnp.reciprocal': jnp.reciprocal, '@coord.contract': coord.contract}

  # log the inverse-CDF interval index out of the reference's own mask expression (math.py:111)
  idx_log = []
  orig_sorted_interp = rmath.sorted_interp

  def logging_sorted_interp(x, xp, fp):
    mask = x[..., None, :] >= xp[..., :, None]                 # math.py:111, verbatim semantics
    idx_log.append((np.sum(mask, axis=-2) - 1).astype(np.int32))
    return orig_sorted_interp(x, xp, fp)

  rmath.sorted_interp = logging_sorted_interp

  out = {}
  for case, spec in CASES.items():
    bind = {k: {kk: resolve.get(vv, vv) if isinstance(vv, str) else vv for kk, vv in spec[k].items()}
            for k in ('Config', 'Model', 'NerfMLP', 'PropMLP', 'ImplicitMask') if k in spec}
    F.set_bindings(bind)
    out[f'{case}/spec'] = np.array(json.dumps({k: spec[k] for k in spec}))
    rng = np.random.default_rng(abs(hash(case)) % 1000 if False else sum(map(ord, case)))
    config = configs.Config()
    model, variables = models.construct_model(jax.random.PRNGKey(17), utils.dummy_rays(), config)
    # biases are zero-initialised in flax; give them values so that they are pinned too
    flat = flatten(variables['params'])
    for k in flat:
      if k.endswith('/kernel'):
        # kernels are stored as bfloat16 bit patterns (half the fixture): make them exactly representable
        b = flat[k].view(np.uint32).astype(np.uint64)
        b = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint32) << 16
        flat[k] = b.view(f32).reshape(flat[k].shape)
      if k.endswith('/bias'):
        flat[k] = (rng.normal(size=flat[k].shape) * 0.1).astype(f32)
      if k.endswith('/embedding'):
        flat[k] = (rng.normal(size=flat[k].shape) * 0.5).astype(f32)
    variables = {'params': unflatten(flat)}
    for k, v in flat.items():
      if k.endswith('/kernel'):
        assert np.all((v.view(np.uint32) & 0xFFFF) == 0)
        out[f'{case}/params_bf16/{k}'] = (v.view(np.uint32) >> 16).astype(np.uint16)
      else:
        out[f'{case}/params/{k}'] = v

    n_patch, P = spec['n_patch'], spec['P']
    shp = (n_patch, P, P)
    o = (rng.normal(size=shp + (3,)) * 0.5).astype(f32)
    d = rng.normal(size=shp + (3,))
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True) * rng.uniform(0.8, 1.2, shp + (1,))).astype(f32)
    v = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(f32)
    rng_or_const = lambda s: (np.full(shp + (1,), s, f32) if np.isscalar(s)
                              else rng.uniform(s[0], s[1], shp + (1,)).astype(f32))
    sm = (rng.uniform(size=shp + (1,)) < 0.7).astype(f32) * rng.uniform(0.5, 1., shp + (1,)).astype(f32)
    rays = utils.Rays(
        pix_coords=rng.uniform(size=shp + (2,)).astype(f32), origins=o, directions=d, viewdirs=v,
        radii=rng.uniform(5e-4, 2e-3, shp + (1,)).astype(f32),
        lossmult=rng.uniform(0.5, 2.0, shp + (1,)).astype(f32), static_mask=sm,
        near=rng_or_const(spec['near']), far=rng_or_const(spec['far']),
        embed_idx=(rng.integers(0, 16, (n_patch, 1, 1, 1)) * np.ones(shp + (1,))).astype(np.int32),
        cam_idx=np.zeros(shp + (1,), np.int32))
    rgb = rng.uniform(size=shp + (3,)).astype(f32)
    batch = utils.Batch(rays=rays, rgb=rgb)
    for name in ('pix_coords', 'origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'static_mask',
                 'near', 'far', 'embed_idx', 'cam_idx'):
      out[f'{case}/rays/{name}'] = getattr(rays, name)
    out[f'{case}/rgb'] = rgb
    L = model.num_levels
    thr = None
    if 'inlier' in spec:
      thr = np.full((L, 1), spec['inlier'], f32)
      out[f'{case}/inlier_thresholds'] = thr
    out[f'{case}/train_frac'] = np.float64(TRAIN_FRAC)

    # ---- the reference's train_step, executed (pmap stand-in: one device, no leading axis) ----
    make_optimizer = train_utils.create_finetune_optimizer if FINETUNE else train_utils.create_optimizer
    state, lr_fn = make_optimizer(config, variables)
    train_step = train_utils.create_train_step(model, config, FINETUNE)
    gsyn = seeded_tree(flat, SYN_GRAD_SEED, scale=3e-3)
    F.HOOK['grad'] = lambda params: {'params': unflatten({k: g.astype(f32) for k, g in gsyn.items()})}
    key0 = jax.random.PRNGKey(1234)
    idx_log.clear()
    new_state, stats, _ = train_step(key0, state, batch, TRAIN_FRAC, thr)
    draws = list(key0.draws)
    assert len(draws) == L and len(idx_log) == L, (len(draws), len(idx_log))
    for lvl in range(L):
      out[f'{case}/l{lvl}_u01'] = draws[lvl].reshape(-1)
      out[f'{case}/l{lvl}_idx'] = idx_log[lvl].reshape(-1, idx_log[lvl].shape[-1])
    for k, val in flatten({k: v for k, v in stats.items()}).items():
      out[f'{case}/stats/{k}'] = np.asarray(val)
    out[f'{case}/lr0'] = np.float64(lr_fn(0))
    newflat = flatten(new_state.params['params'])
    for k in flat:
      out[f'{case}/update_head/{k}'] = (newflat[k] - flat[k]).reshape(-1)[:32]
    clipped = flatten(train_utils.clip_gradients(F.HOOK['grad'](None), config)['params'])
    for k in flat:
      out[f'{case}/clip_head/{k}'] = clipped[k].reshape(-1)[:32]
      out[f'{case}/clip_norm/{k}'] = np.float64(np.sqrt(np.sum(clipped[k].astype(np.float64)**2)))

    # ---- forward of the same step, recorded (same draws: fresh key, same seed) ---------------
    key1 = jax.random.PRNGKey(1234)
    kk = jax.random.split(key1)[1]
    renderings, history = model.apply(variables, kk, rays, train_frac=TRAIN_FRAC, compute_extras=False,
                                      zero_glo=False, zero_tra=False)
    assert all(np.array_equal(a, b) for a, b in zip(key1.draws, draws))
    hs = spec['hist_step']
    for lvl in range(L):
      for k, val in renderings[lvl].items():
        out[f'{case}/train/l{lvl}_rend_{k}'] = np.asarray(val, f32).reshape((-1,) + np.shape(val)[3:])
      for k in HIST_KEYS:
        if k in history[lvl]:
          a = np.asarray(history[lvl][k], f32)
          a = a.reshape((-1,) + a.shape[3:])
          out[f'{case}/train/l{lvl}_{k}'] = a if k == 'sdist' else a[::hs]     # sdist: every ray
    out[f'{case}/hist_step'] = np.int64(hs)

    # ---- eval forward (rng None, compute_extras) -----------------------------------------------
    sub = jax.tree_util.tree_map(lambda x: x.reshape((-1, x.shape[-1]))[::max(1, hs // 2)], rays)
    erend, _ = model.apply(variables, None, sub, train_frac=1.0, compute_extras=True,
                           zero_glo=bool(config.enable_render_zero_glo),
                           zero_tra=bool(config.enable_render_zero_tra))
    out[f'{case}/eval/ray_step'] = np.int64(max(1, hs // 2))
    for k, val in erend[-1].items():
      if not k.startswith('ray_'):
        out[f'{case}/eval/{k}'] = np.asarray(val, f32)

    # ---- float64 central differences of the reference's loss_fn, stop_gradient replayed --------
    loss_fn = F.HOOK['loss_fn']                 # the closure train_step built (same key, batch, thr)
    F.Tape.mode, F.Tape.vals = 'record', []
    base_loss, _ = loss_fn(variables)           # float32 run: fills the tape
    F.Tape.mode = 'replay'
    out[f'{case}/loss'] = np.float64(base_loss)
    assert abs(float(base_loss) - float(stats['loss'])) <= 1e-6 * abs(float(base_loss)) + 1e-9
    _jax_standin.KEEP64[0] = True
    try:
      # a float64 twin of the closure: the same train_step body run on float64 copies of the batch (float32
      # inputs would keep numpy in float32 up to the first parameter), tape replayed
      to64 = lambda x: x.astype(np.float64) if x.dtype == np.float32 else x
      batch64 = jax.tree_util.tree_map(to64, batch)
      vars64 = jax.tree_util.tree_map(to64, variables)
      state64, _ = make_optimizer(config, vars64)
      # Model.__call__ runs first inside loss_fn, so the first L tape entries are the levels' sample positions:
      # keep the float32 run's (they are the fixture's `train/l*_sdist`), and rebuild every later constant
      # (interlevel c / w, robust errors / masks, HA-NeRF mask) in float64 on top of them.
      assert all(np.array_equal(F.Tape.vals[l].reshape(-1, F.Tape.vals[l].shape[-1]),
                                out[f'{case}/train/l{l}_sdist']) for l in range(L))
      for l in range(L):          # same values, float64 dtype: `1 - s` must not be evaluated in float32
        F.Tape.vals[l] = F.Tape.vals[l].astype(np.float64)
      F.Tape.pos, F.Tape.refresh_from = 0, L
      train_step(jax.random.PRNGKey(1234), state64, batch64, TRAIN_FRAC, thr)
      F.Tape.refresh_from = None
      loss_fn = F.HOOK['loss_fn']

      def loss64(flat64):
        F.Tape.pos = 0
        val, _ = loss_fn({'params': unflatten(flat64)})
        assert F.Tape.pos == len(F.Tape.vals)
        return float(val)
      flat64 = {k: x.astype(np.float64) for k, x in flat.items()}
      out[f'{case}/loss64'] = np.float64(loss64(flat64))
      for i in range(N_DIRS):
        vdir = seeded_tree(flat, 1000 + i)
        ests = []
        for h in (1e-8, 2e-8):   # small enough that ReLU-kink crossings do not matter, float64 keeps 8 digits
          lp = loss64({k: flat64[k] + h * vdir[k] for k in flat64})
          lm = loss64({k: flat64[k] - h * vdir[k] for k in flat64})
          ests.append((lp - lm) / (2 * h))
        out[f'{case}/fd/dir{i}'] = np.float64(ests[0])
        out[f'{case}/fd/dir{i}_h2'] = np.float64(ests[1])     # agreement of the two = FD quality
    finally:
      _jax_standin.KEEP64[0] = False
      F.Tape.mode = 'off'
    print(case, 'loss', float(base_loss), 'fd', [float(out[f'{case}/fd/dir{i}']) for i in range(N_DIRS)],
          [float(out[f'{case}/fd/dir{i}_h2']) for i in range(N_DIRS)])

  # ---- unit vectors: robustnerf_mask across regimes, compute_data_loss corners ------------------
  rng = np.random.default_rng(99)
  F.set_bindings({})
  errs = (rng.uniform(size=(3, 16, 16, 3)) ** 2).astype(f32)
  errs[1] *= 0.3
  out['unit/robust/errors'] = errs
  for ti, thr_v in enumerate((0.05, 0.2, 0.45, 1.5)):
    for f in (3, 4, 5):
      for inner in (8, 5):
        config = configs.Config(patch_size=16, robustnerf_smoothed_filter_size=f,
                                robustnerf_inner_patch_size=inner)
        mask, st = train_utils.robustnerf_mask(errs, np.array([thr_v], f32), config)
        tag = f'unit/robust/t{ti}_f{f}_i{inner}'
        out[tag + '/thr'] = np.float64(thr_v)
        out[tag + '/mask_img'] = np.asarray(mask, f32)
        for k, val in st.items():
          out[tag + '/' + k] = np.float64(val)
  rend = [{'rgb': rng.uniform(size=(2, 4, 4, 3)).astype(f32)} for _ in range(3)]
  gt = rng.uniform(size=(2, 4, 4, 3)).astype(f32)
  urays = utils.dummy_rays().replace(
      lossmult=rng.uniform(0.2, 3., (2, 4, 4, 1)).astype(f32),
      static_mask=rng.uniform(size=(2, 4, 4, 1)).astype(f32))
  ub = utils.Batch(rays=urays, rgb=gt)
  out['unit/data/gt'] = gt
  out['unit/data/lossmult'] = urays.lossmult
  out['unit/data/static_mask'] = urays.static_mask
  for i, r in enumerate(rend):
    out[f'unit/data/rend{i}'] = r['rgb']
  for tag, kw, use_mask in (('mse', dict(data_loss_type='mse'), False), ('charb', {}, False),
                            ('nomulti', dict(disable_multiscale_loss=True), False),
                            ('mask', dict(withmask_transient_weight=0.0), True),
                            ('mask_w', dict(withmask_transient_weight=0.4, data_coarse_loss_mult=0.5), True)):
    config = configs.Config(**kw)
    losses, st = train_utils.compute_data_loss(ub, urays, rend, config, use_mask)
    out[f'unit/data/{tag}/data'] = np.float64(losses['data'])
    out[f'unit/data/{tag}/mses'] = np.asarray(st['mses'], np.float64)

  # ---- coord.construct_ray_warps for every raydist_fn it knows (coord.py:84-90) ---------------------------------
  sv = np.linspace(0, 1, 33).astype(f32)[None, :]
  nearv = np.array([[0.05], [0.3], [1.0]], f32)
  farv = np.array([[1.2], [40.0], [1e3]], f32)
  out['unit/raywarp/s'], out['unit/raywarp/near'], out['unit/raywarp/far'] = sv, nearv, farv

  def named(name, fn):
    fn.__name__ = name
    return fn
  for name, fn in (('none', None), ('reciprocal', named('reciprocal', lambda x: np.reciprocal(x))),
                   ('log', named('log', lambda x: np.log(x))), ('exp', named('exp', lambda x: np.exp(x))),
                   ('sqrt', named('sqrt', lambda x: np.sqrt(x))), ('square', named('square', lambda x: np.square(x))),
                   ('piecewise', 'piecewise')):
    fv = np.minimum(farv, 40.0) if name == 'exp' else farv          # exp(1e3) overflows float32
    t_to_s, s_to_t = coord.construct_ray_warps(fn, nearv, fv)
    out[f'unit/raywarp/{name}/t'] = np.asarray(s_to_t(sv), f32)
    out[f'unit/raywarp/{name}/s_back'] = np.asarray(t_to_s(s_to_t(sv)), f32)

  path = os.path.join(HERE, 'ref_model.npz')
  np.savez_compressed(path, **out)
  print(f'wrote {path}: {len(out)} arrays, {sum(np.asarray(v).nbytes for v in out.values())/1e6:.2f} MB raw, '
        f'{os.path.getsize(path)/1e6:.2f} MB on disk')


if __name__ == '__main__':
  main()

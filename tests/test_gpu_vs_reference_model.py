"""HIP path vs vectors recorded from the REFERENCE'S OWN models.py / train_utils.py (tests/golden/ref_model.npz,
generator tests/golden/gen_model_fixtures.py) -- directly, not through the oracle: forward of every level, the
sampler's interval indices at level > 0, losses / stats of train_step, directional derivatives of the loss,
clip + update on the reference's synthetic gradient tree.  fp32 MFMA mode (`compute_dtype='fp32'`).

Tolerances: sample positions 5e-5 (level 0: 5e-7), interval indices equal except at CDF knots (<= 0.2 %, never
more than one interval off), rendered colours 1e-4 (north_star), per-sample density / colour 1e-3 of the array max
(1 ulp of a level>0 sample position through the 2^11 IPE frequency), losses 1e-4, directional derivatives 1.5e-3 against the float32 oracle and 1e-2 against the reference's float64
finite differences (see the comment in the test)."""
import numpy as np
import pytest
import torch

from tests import ref_model_fixture as FX

pytestmark = pytest.mark.gpu
dev = 'cuda'


def _build(case, compute_dtype='fp32'):
  from nerf_hugs_amd.internal import configs, train_utils, utils
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, FX.gin_lines(case))
  config = configs.make_config()
  model, state, render_fn, train_step, lr_fn = train_utils.setup_model(config, 0, compute_dtype=compute_dtype)
  model.load_variables(state.flat, FX.param_tree(case))
  if FX.is_finetune(case):        # train.py:97-109: the stage after training, on the same model / parameters
    state, train_step, lr_fn = train_utils.setup_finetune_model(config, model, state)
  shp = FX.get(case, 'rays/origins').shape[:-1]
  T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
  rays = utils.Rays(**{f: T(FX.get(case, f'rays/{f}')) for f in FX.RAY_FIELDS})
  batch = utils.Batch(rays=rays, rgb=T(FX.get(case, 'rgb')))
  return config, model, state, train_step, batch, lr_fn


@pytest.mark.parametrize('case', FX.CASES + FX.TRAIN_VARIANTS + FX.FINETUNE_CASES)
def test_forward_vs_reference(case):
  from nerf_hugs_amd.internal import models as M
  config, model, state, train_step, batch, _ = _build(case)
  L = model.num_levels
  eng = model.engine(dev)
  eng.refresh_weights(state.flat)
  u01 = [u.to(dev) for u in FX.u01(case, L)]
  r = M.rays_to_dict(batch.rays, dev)
  N = r['origins'].shape[0]
  levels = eng.forward(state.flat, r, float(FX.get(case, 'train_frac')), u01, False)
  hs = int(FX.get(case, 'hist_step'))
  have = set(FX.keys(case, 'train/'))
  for l in range(L):
    S = levels[l]['S']
    np.testing.assert_allclose(levels[l]['sdist'].cpu().numpy(), FX.get(case, f'train/l{l}_sdist'), rtol=0,
                               atol=5e-7 if l == 0 else 5e-5, err_msg=f'{case} sdist l{l}')
    cmp = [('density', levels[l]['density'].reshape(N, S)), ('weights', levels[l]['weights'].reshape(N, S))]
    if levels[l]['rgb'] is not None:
      cmp.append(('rgb', levels[l]['rgb'].reshape(N, S, 3)))
    if levels[l].get('dens_t') is not None:
      cmp += [('density_transient', levels[l]['dens_t'].reshape(N, S)), ('rgb_transient', levels[l]['rgb_t'].reshape(N, S, 3)),
              ('uncertainty', levels[l]['unc'].reshape(N, S, 1))]
    for k, mine in cmp:
      ref = FX.get(case, f'train/l{l}_{k}')
      np.testing.assert_allclose(mine.float().cpu().numpy()[::hs].reshape(ref.shape), ref, rtol=0,
                                 atol=1e-3 * max(float(np.abs(ref).max()), 1e-6), err_msg=f'{case} l{l} {k}')
    np.testing.assert_allclose(levels[l]['rgb_out'].cpu().numpy(), FX.get(case, f'train/l{l}_rend_rgb'), rtol=0, atol=1e-4,
                               err_msg=f'{case} rendered rgb l{l}')
    for k in ('rgb_combined', 'rgb_static', 'rgb_transient', 'uncertainty'):
      if f'l{l}_rend_{k}' in have:
        ref = FX.get(case, f'train/l{l}_rend_{k}')
        np.testing.assert_allclose(levels[l][k].cpu().numpy().reshape(ref.shape), ref, rtol=0, atol=1e-4, err_msg=f'{case} {k}')
  if model.mask_spec is not None:
    m = eng.mask_forward(state.flat, r, N)['mask'][:N].cpu().numpy()
    np.testing.assert_allclose(m.reshape(-1), FX.get(case, f'train/l{L-1}_rend_implicit_mask').reshape(-1), rtol=0, atol=2e-5)


@pytest.mark.parametrize('case', FX.CASES)
def test_sampler_bins_and_indices_vs_reference(case):
  """Level by level, the HIP sampler fed with the REFERENCE's previous-level (sdist, weights): interval index of
  every sample against the reference's own inverse-CDF mask (math.py:111), sample positions, sortedness."""
  from nerf_hugs_amd.internal import stepfun
  from oracle import torch_ref as R
  cfg = FX.oracle_cfg(case)
  hs = int(FX.get(case, 'hist_step'))
  tf = float(FX.get(case, 'train_frac'))
  near = FX.get(case, 'rays/near').reshape(-1)
  far = FX.get(case, 'rays/far').reshape(-1)
  G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
  for l in range(cfg.num_levels):
    S = cfg.num_prop_samples if l < cfg.num_levels - 1 else cfg.num_nerf_samples
    rows = np.arange(near.shape[0]) if l == 0 else np.arange(near.shape[0])[::hs]
    if l == 0:
      t_prev = np.stack([np.zeros_like(near), np.ones_like(far)], -1)
      w_prev = np.ones_like(near)[:, None]
    else:
      t_prev, w_prev = FX.get(case, f'train/l{l-1}_sdist')[::hs], FX.get(case, f'train/l{l-1}_weights')
    dilation = cfg.dilation_bias + cfg.dilation_multiplier / (cfg.num_prop_samples ** l)
    anneal = (cfg.anneal_slope * tf) / ((cfg.anneal_slope - 1) * tf + 1)
    sd, td, idx, tin, win = stepfun.level_sample(
        G(t_prev), G(w_prev), l > 0, dilation if l > 0 else 0., (0., 1.), anneal, 0., S, G(FX.get(case, f'l{l}_u01')[rows]),
        cfg.raydist_fn, G(near[rows]), G(far[rows]), return_debug=True)
    ref_idx, ref_sd = FX.get(case, f'l{l}_idx')[rows], FX.get(case, f'train/l{l}_sdist')[rows]
    idx, sd = idx.cpu().numpy(), sd.cpu().numpy()
    assert np.all(np.diff(sd, axis=-1) >= 0)
    np.testing.assert_allclose(sd, ref_sd, rtol=0, atol=5e-7 if l == 0 else 2e-6)
    # reference order (numpy-pairwise sums, sequential cumsum -- what ships): EVERY interval index of every level equals
    # the one the reference's own math.sorted_interp mask picked
    assert np.array_equal(idx, ref_idx), f'{case} l{l}: {int((idx != ref_idx).sum())}/{idx.size} interval indices differ'


@pytest.mark.parametrize('case', FX.CASES + FX.TRAIN_VARIANTS + FX.FINETUNE_CASES)
def test_train_step_stats_and_derivatives_vs_reference(case):
  """One train_step on the reference's batch / weights / uniform draws: loss terms, mses, psnrs, robust stats,
  weight_l2s, and <grad, v> against the reference's float64 central differences along seeded directions."""
  config, model, state, train_step, batch, _ = _build(case)
  L = model.num_levels
  thr = FX.get(case, 'inlier_thresholds') if config.transient_type == 'robustnerf' else None
  theta0 = state.flat.clone()
  state, stats, _ = train_step([u.to(dev) for u in FX.u01(case, L)], state, batch, float(FX.get(case, 'train_frac')), thr)
  torch.cuda.synchronize()
  ref_loss = float(FX.get(case, 'stats/loss'))
  assert abs(float(stats['loss']) - ref_loss) <= 1e-4 * abs(ref_loss), (float(stats['loss']), ref_loss)
  assert set(stats['losses'].keys()) == set(FX.keys(case, 'stats/losses/'))
  for k in stats['losses']:
    ref = float(FX.get(case, f'stats/losses/{k}'))
    assert abs(float(stats['losses'][k]) - ref) <= 1e-4 * abs(ref) + 2e-6 * abs(ref_loss), (k, float(stats['losses'][k]), ref)
  np.testing.assert_allclose(np.asarray(stats['mses']), FX.get(case, 'stats/mses'), rtol=1e-4)
  np.testing.assert_allclose(np.asarray(stats['psnrs']), FX.get(case, 'stats/psnrs'), rtol=0, atol=1e-3)
  for k in [k for k in FX.keys(case, 'stats/') if k.startswith('robust_')]:
    np.testing.assert_allclose(np.asarray(stats[k]), FX.get(case, f'stats/{k}'), rtol=1e-4, atol=1e-6, err_msg=k)
  if config.transient_type == 'hanerf' and not FX.is_finetune(case):
    np.testing.assert_allclose(np.asarray(stats['implicit_mask']), FX.get(case, 'stats/implicit_mask'), rtol=1e-4)
  assert ('implicit_mask' in stats) == ('implicit_mask' in FX.keys(case, 'stats/'))
  # summarize_tree key sets (train_utils.py:61-69) and weight_l2s values
  for group in ('weight_l2s', 'grad_norms', 'grad_maxes', 'opt_update_norms', 'opt_update_maxes'):
    assert set(stats[group].keys()) == set(FX.keys(case, f'stats/{group}/')), group
  for k, v in stats['weight_l2s'].items():
    ref = float(FX.get(case, f'stats/weight_l2s/{k}'))
    assert abs(float(v) - ref) <= 1e-4 * ref + 1e-12, k
  # gradient (before clipping: the buffer the backward pass filled)
  grad = model.engine(dev).ws.get('grad', (model.layout.size + 64,))
  g = {'/'.join(lf['path']): model.layout.view(grad, lf['path']).double().cpu().numpy() for lf in model.layout.leaves}
  # The reference's number is the float64 derivative.  A float32 evaluation of the same network lands a handful of
  # the ~1e6 ReLU pre-activations (they differ by ~5e-4 through the 2^11 IPE frequency) on the other side of zero,
  # and with 32 rays one sample can carry percents of a leaf's gradient: the oracle evaluated in float32 shows the
  # SAME deviation from its own float64 value (tests/test_oracle_vs_reference_model.py pins that one to the
  # reference at 2e-5).  So: HIP vs the float32 oracle tightly, HIP vs the reference's float64 number loosely.
  from oracle import torch_ref as R
  cfg = FX.oracle_cfg(case)
  othr = None if thr is None else [torch.from_numpy(t.copy()) for t in thr]
  _, og, _, _ = R.loss_and_grad(cfg, FX.param_tree(case), FX.rays_flat(case),
                                torch.from_numpy(FX.get(case, 'rgb').reshape(-1, 3).copy()),
                                float(FX.get(case, 'train_frac')), FX.u01(case, L), othr, is_finetune=FX.is_finetune(case))
  if FX.is_finetune(case):
    # the parts of the tree the finetune loss does not reach: exactly zero, as in the reference's value_and_grad
    dead = {k for k in og if float(og[k].abs().max()) == 0.0}
    assert {k.split('/')[0] for k in dead} >= {'PropMLP_0', 'TransientEmbed_0'} and 'GloEmbed_0/embedding' not in dead
    for k in dead:
      assert float(np.abs(g[k]).max()) == 0.0, k
    # and only the reference's 'trainable' partition moved
    for lf in model.layout.leaves:
      d = float((model.layout.view(state.flat, lf['path']) - model.layout.view(theta0, lf['path'])).abs().max())
      assert (d > 0) == ('/'.join(lf['path']) == 'GloEmbed_0/embedding'), lf['path']
  if case in FX.TRAIN_VARIANTS + FX.FINETUNE_CASES:
    # (no finite differences recorded for the option variants: the gradient against the float32 oracle, whose losses the CPU suite
    #  holds to the reference for the same case, along the same seeded directions)
    ndir = 4
    proj = lambda tree, v: sum(float((np.asarray(tree[k], np.float64) * v[k]).sum()) for k in v)
    vs = [FX.seeded_tree(case, 1000 + i) for i in range(ndir)]
    o32s = [proj({k: og[k].double().numpy() for k in og}, v) for v in vs]
    scale = max(abs(x) for x in o32s)
    for v, o32 in zip(vs, o32s):
      assert abs(proj(g, v) - o32) <= 1.5e-3 * scale, (case, proj(g, v), o32)
    return
  scale = max(abs(float(FX.get(case, f'fd/dir{j}'))) for j in range(FX.N_DIRS))
  for i in range(FX.N_DIRS):
    v = FX.seeded_tree(case, 1000 + i)
    mine = sum(float((g[k] * v[k]).sum()) for k in v)
    o32 = sum(float((og[k].double().numpy() * v[k]).sum()) for k in v)
    fd = float(FX.get(case, f'fd/dir{i}'))
    assert abs(mine - o32) <= 1.5e-3 * scale, (case, i, mine, o32)
    assert abs(mine - fd) <= 1e-2 * scale, (case, i, mine, fd)


@pytest.mark.parametrize('case', ['base', 'withmask', 'robust', 'decay_clips_schedule', 'ft_nerfw', 'ft_hanerf'])
def test_clip_and_update_on_reference_gradient(case):
  """The reference's train_step was run with a seeded synthetic gradient tree through its own stats /
  clip_gradients / nan_to_num / apply_gradients code; the same tree through hugs_opt_stats + hugs_opt_adam.
  grad_norms / grad_maxes and clip are pinned by the reference; Adam itself is optax (restated on both sides)."""
  config, model, state, train_step, batch, lr_fn = _build(case)
  lay = model.layout
  grad = model.engine(dev).ws.get('grad', (lay.size + 64,))
  grad.zero_()
  syn = FX.seeded_tree(case, 4242, 3e-3)
  for lf in lay.leaves:
    lay.view(grad, lf['path']).copy_(torch.from_numpy(syn['/'.join(lf['path'])].astype(np.float32)))
  theta0 = {'/'.join(lf['path']): lay.view(state.flat, lf['path']).clone() for lf in lay.leaves}
  assert abs(lr_fn(0) - float(FX.get(case, 'lr0'))) <= 1e-6 * lr_fn(0)      # the reference's schedule runs in float32
  leaf_stats = train_step.optimizer_step(state, grad, 1.0).cpu().numpy()
  nleaf = len(lay.leaves)
  ls = leaf_stats[:nleaf * 4].reshape(nleaf, 4)
  for i, lf in enumerate(lay.leaves):
    name = '/'.join(lf['path'])
    assert abs(np.sqrt(ls[i, 0]) - float(FX.get(case, f'stats/grad_norms/{name}'))) <= 1e-5 * np.sqrt(ls[i, 0]), name
    assert abs(ls[i, 1] - float(FX.get(case, f'stats/grad_maxes/{name}'))) <= 1e-6 * ls[i, 1], name
    delta = (lay.view(state.flat, lf['path']) - theta0[name]).reshape(-1)[:32].cpu().numpy()
    ref = FX.get(case, f'update_head/{name}')
    # |update| ~ lr * g / (|g| + eps) with clipped g ~ eps: sensitive to the clip scale; 1 ulp of theta on top
    np.testing.assert_allclose(delta, ref, rtol=2e-3, atol=2e-8 + 1.2e-7 * float(theta0[name].abs().max()), err_msg=name)
  # per-module clip scale = clipped norm / value-clipped norm, recovered from the fixture
  mod_scale = leaf_stats[nleaf * 6:nleaf * 6 + len(lay.modules)]
  for mi, mod in enumerate(lay.modules):
    names = [k for k in syn if k.split('/')[0] == mod]
    vc = (lambda x: np.clip(x, -config.grad_max_val, config.grad_max_val)) if config.grad_max_val > 0 else (lambda x: x)
    raw = np.sqrt(sum(float((vc(syn[k].astype(np.float32)).astype(np.float64)**2).sum()) for k in names))
    clipped = np.sqrt(sum(float(FX.get(case, f'clip_norm/{k}'))**2 for k in names))
    assert abs(mod_scale[mi] - clipped / raw) <= 2e-5 * clipped / raw, (mod, mod_scale[mi], clipped / raw)


def test_robust_mask_kernel_vs_reference_unit_vectors():
  """hugs_robust_mask on the reference's robustnerf_mask unit vectors: thresholds from all-outlier to all-inlier,
  box filters 3 / 4 (even: asymmetric SAME padding) / 5, inner patches 8 / 5."""
  from nerf_hugs_amd import _lib
  z = FX.npz()
  errs = z['unit/robust/errors']                        # [3,16,16,3] per-channel errors = |pred - gt| with gt = 0
  pred = torch.from_numpy(errs.reshape(-1, 3).copy()).to(dev)
  gt = torch.zeros_like(pred)
  n_patch, P = errs.shape[0], 16
  N = n_patch * P * P
  tags = sorted({k.rsplit('/', 1)[0] for k in z.files if k.startswith('unit/robust/t')})
  for tag in tags:
    f, inner = int(tag.split('_f')[1].split('_')[0]), int(tag.split('_i')[1])
    thr = torch.tensor([float(z[tag + '/thr'])], device=dev)
    mask = torch.empty(N, device=dev); err = torch.empty(N, device=dev)
    part = torch.empty(n_patch * 4, device=dev); st = torch.empty(5, device=dev)
    _lib.call('hugs_robust_mask', n_patch, P, pred, gt, thr, 0.5, f, 0.5, inner, 0.4, mask, err, part, st)
    np.testing.assert_array_equal(mask.cpu().numpy().reshape(n_patch, P, P, 1), z[tag + '/mask_img'], err_msg=tag)
    st = st.cpu().numpy()
    for i, k in enumerate(('inlier_threshold', 'is_inlier_loss', 'has_inlier_neighbors', 'is_inlier_patch', 'mask')):
      assert abs(st[i] - float(z[f'{tag}/{k}'])) <= 2e-6, (tag, k, st[i], float(z[f'{tag}/{k}']))


def test_data_loss_kernel_vs_reference_unit_vectors():
  """hugs_data_loss on the reference's compute_data_loss unit vectors (per-ray lossmult, disable_multiscale_loss,
  static masks incl. the [...,1]-denominator quirk, coarse multipliers)."""
  from nerf_hugs_amd import _lib
  z = FX.npz()
  gt = torch.from_numpy(z['unit/data/gt'].reshape(-1, 3).copy()).to(dev)
  N, L = gt.shape[0], 3
  pred = torch.stack([torch.from_numpy(z[f'unit/data/rend{i}'].reshape(-1, 3).copy()) for i in range(L)]).to(dev).contiguous()
  lossmult = torch.from_numpy(z['unit/data/lossmult'].reshape(-1).copy()).to(dev)
  smask = torch.from_numpy(z['unit/data/static_mask'].reshape(-1).copy()).to(dev)
  for tag, charb, lm, mode, tw, coarse in (('mse', 0, lossmult, 0, 0., 0.), ('charb', 1, lossmult, 0, 0., 0.),
                                           ('nomulti', 1, None, 0, 0., 0.), ('mask', 1, smask, 1, 0., 0.),
                                           ('mask_w', 1, smask, 1, 0.4, 0.5)):
    coef = torch.tensor([coarse] * (L - 1) + [1.0], device=dev)
    d_pred = torch.empty(L, N, 3, device=dev); out = torch.zeros(2 * L, device=dev)
    _lib.call('hugs_data_loss', N, L, pred, gt, lm, mode, tw, charb, 0.001, coef, d_pred, out)
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[0::2], z[f'unit/data/{tag}/mses'], rtol=2e-5, err_msg=tag)
    data = coarse * o[1:2 * L - 2:2].sum() + o[2 * L - 1]
    assert abs(data - float(z[f'unit/data/{tag}/data'])) <= 2e-5 * abs(data), tag


from tests import ref_model_variants as FV


@pytest.mark.parametrize('case', FV.CASES)
def test_forward_vs_reference_option_variants(case):
  """HIP forward of every level against the reference's own Model.__call__ executed for option variants
  (tests/golden/gen_model_variant_fixtures.py): near-plane annealing, one jitter draw per sample, cylinder rays, four levels, sampler /
  head / encoding knobs, a model without a view layer, a deeper view MLP, a grey non-opaque background, log ray distance + contraction."""
  from nerf_hugs_amd.internal import configs, train_utils, utils, models as M
  from tests.test_oracle_vs_reference_model import _check_variant_levels
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, FV.gin_lines(case))
  config = configs.make_config()
  model, state, render_fn, train_step, lr_fn = train_utils.setup_model(config, 0, compute_dtype='fp32')
  model.load_variables(state.flat, FV.param_tree(case))
  T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
  rays = utils.Rays(**{f: T(FV.get(case, f'rays/{f}')) for f in FV.RAY_FIELDS})
  L = model.num_levels
  eng = model.engine(dev)
  eng.refresh_weights(state.flat)
  r = M.rays_to_dict(rays, dev)
  N = r['origins'].shape[0]
  levels = eng.forward(state.flat, r, float(FV.npz()['train_frac']), [u.to(dev) for u in FV.u01(case, L)], False)

  def mine(l, k):
    if k == 'rend_rgb':
      return levels[l]['rgb_out'].cpu().numpy()
    return levels[l][k].reshape(N, -1).float().cpu().numpy()
  _check_variant_levels(case, L, mine)

"""Round-4 parity statements that round 3 left loose ("weak" 1-3):

 * whole-step gradients WITHOUT masking rays: the oracle replays the HIP path's own ReLU decisions (pre * mask instead of
   relu(pre), masks = "stored activation > 0"), so a pre-activation of +-1e-7 that the two float32 implementations put on
   different sides of zero no longer moves a weight gradient by a whole sample -- every leaf must agree to 1e-4 of its max;
 * the benchmarked bf16 mode against the oracle evaluated with bf16-rounded GEMM operands (weights, layer inputs, and the
   gradients that flow back through them; fp32 accumulation): per-leaf relative L2 error instead of a cosine;
 * BASELINE config 1's shape (base / mse, L=2, S=(64,64), 1024 rays, the 8x1024 + 4x256 nets) through the HIP path;
 * a 3-step TRAINING TRAJECTORY (forward, loss, backward, clip, Adam, lr schedule, re-cast; three times) against the oracle's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_train_step import SMALL, _run_case


def _hip_masks_samples_feats(model, levels, N):
  """What the oracle replays: per level the ReLU masks in call order (trunk layers, then the view layer), the sample
  positions and the encoder output of the HIP step."""
  masks, ov, ofe = [], [], []
  for lv in levels:
    spec, S = lv['spec'], lv['S']
    W, F = spec.net_width, spec.F
    m = [(lv['acts'][i + 1][:, :W].float() > 0).cpu().reshape(N, S, W) for i in range(spec.net_depth)]
    if lv.get('hview') is not None:      # the view layer, then the further view layers of net_depth_viewdirs > 1
      for h in (lv.get('hviews') or [lv['hview']]):
        m.append((h.float() > 0).cpu().reshape(N, S, -1))
    masks.append(m)
    ov.append((lv['sdist'].cpu().clone(), lv['tdist'].cpu().clone()))
    ofe.append(lv['X0'][:, :F].float().cpu().reshape(N, S, F))
  return masks, ov, ofe


_LAST = {}


def _step_and_replay(gin, compute_dtype, n_patch=1, P=8, near=0.1, far=1.2, quant=False, inlier=None, replay_feats=True):
  from tests import hugs_testlib as H
  from oracle import torch_ref as R
  from nerf_hugs_amd.internal import models as M
  config, model, state, _, train_step, cfg, oparams = H.make_pair(gin, compute_dtype=compute_dtype)
  batch = H.synth_rays(n_patch, P, 5, near=near, far=far)
  N, L = n_patch * P * P, model.num_levels
  gen = torch.Generator(device='cuda').manual_seed(11)
  u01 = [torch.rand(N, generator=gen, device='cuda') for _ in range(L)]
  eng = model.engine('cuda')
  _LAST['engine'] = eng
  eng.refresh_weights(state.flat)
  levels = eng.forward(state.flat, M.rays_to_dict(batch.rays, 'cuda'), 0.37, u01, False)
  thr = None if inlier is None else np.full((L, 1), inlier, np.float32)
  state, stats, _ = train_step(u01, state, batch, 0.37, thr)     # (the same jitter: the step's forward rewrites the same buffers)
  torch.cuda.synchronize()
  masks, ov, ofe = _hip_masks_samples_feats(model, levels, N)
  othr = None if inlier is None else [torch.tensor([inlier]) for _ in range(L)]
  ostats, ograds, orend, _ = R.loss_and_grad(cfg, oparams, H.oracle_rays(batch), batch.rgb.reshape(-1, 3), 0.37,
                                             [u.cpu() for u in u01], othr, relu_masks=masks, override_samples=ov,
                                             override_feats=ofe if replay_feats else None, quant=quant)
  grad = eng.ws.get('grad', (model.layout.size + 64,))
  out = {}
  for lf in model.layout.leaves:
    name = '/'.join(lf['path'])
    g = model.layout.view(grad, lf['path']).cpu().double()
    og = ograds[name].double()
    out[name] = (float((g - og).abs().max() / og.abs().max().clamp(min=1e-30)), float((g - og).norm() / og.norm().clamp(min=1e-30)),
                 og.numel())
  return out, stats, ostats


@pytest.mark.parametrize('variant', ['base2', 'default3_charb_contract_glo', 'withmask_glo48', 'robustnerf', 'view_depth3_glo'])
def test_whole_step_gradient_every_leaf_with_replayed_relu_masks(variant):
  gin, kw = list(SMALL), {}
  if variant == 'default3_charb_contract_glo':
    gin = [g for g in SMALL if not g.startswith('Model.num_') and 'data_loss_type' not in g] + [
        "Model.num_levels = 3", "Model.num_prop_samples = 64", "Model.num_nerf_samples = 32",
        "Model.raydist_fn = @jnp.reciprocal", "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract",
        "Model.num_glo_features = 4", "Config.data_coarse_loss_mult = 0.1"]
    kw = dict(near=(0.05, 0.3), far=1e6)
  elif variant == 'withmask_glo48':
    gin = [g for g in SMALL if 'data_loss_type' not in g] + ["Config.transient_type = 'withmask'", "Model.num_glo_features = 48"]
    kw = dict(n_patch=2)
  elif variant == 'view_depth3_glo':      # round 5: NerfMLP.net_depth_viewdirs > 1 (models.py:508-512)
    gin = list(SMALL) + ["NerfMLP.net_depth_viewdirs = 3", "Model.num_glo_features = 4"]
  elif variant == 'robustnerf':
    gin = [g.replace('patch_size = 8', 'patch_size = 16') for g in SMALL] + [
        "Config.transient_type = 'robustnerf'", "Config.robustnerf_inlier_quantile = 0.8"]
    kw = dict(n_patch=2, P=16, inlier=0.3)
  errs, stats, ostats = _step_and_replay(gin, 'fp32', **kw)
  assert abs(float(stats['loss']) / float(ostats['loss']) - 1) < 1e-4
  worst = max(errs.items(), key=lambda kv: kv[1][0])
  for name, (emax, el2, n) in errs.items():
    # (measured, profiles/r04_parity_margins.txt: 3.3e-5 at worst) -- north_star's own 1e-4
    assert emax <= 1e-4, f'{variant}: {name}: max err {emax:.2e} of the leaf max (L2 {el2:.2e}); worst {worst}'


def test_bf16_full_width_gradients_vs_bf16_rounded_oracle():
  """bf16 mode (what bench.py times): the oracle with bf16-rounded weights / activations / back-flowing gradients and the
  HIP path's ReLU masks, samples and features.  What is left is fp32 accumulation order and the places where the product
  keeps MORE precision than the emulation does; per-leaf relative L2 error."""
  gin = [g for g in SMALL if 'net_width' not in g] + ["PropMLP.net_width = 256", "NerfMLP.net_width = 1024"]
  errs, stats, ostats = _step_and_replay(gin, 'bf16', quant=True)
  assert abs(float(stats['loss']) / float(ostats['loss']) - 1) < 2e-3
  rep = {k: v for k, v in errs.items()}
  big = [(k, v) for k, v in rep.items() if v[2] >= 1024]
  assert len(big) >= 12
  for name, (emax, el2, n) in big:
    assert el2 < 1.5e-2, f'{name}: relative L2 error {el2:.3e} (max {emax:.2e})'
  for name, (emax, el2, n) in rep.items():        # biases and heads too (1 .. 1024 numbers each): looser, a few samples decide them
    assert el2 < 2e-2, f'{name}: relative L2 error {el2:.3e}'


FULL_WIDTH = [g for g in SMALL if 'net_width' not in g] + ["PropMLP.net_width = 256", "NerfMLP.net_width = 1024"]


def test_whole_step_gradient_every_leaf_full_width_256_rays_fp32():
  """Round 5: the every-leaf, no-ray-masked 1e-4 statement at the BENCHMARKED width -- NerfMLP 8x1024 +
  PropMLP 4x256, 64 + 128 samples, 256 rays (32768 rows through the 1024-wide trunk), fp32 parity mode, the HIP step's ReLU
  decisions replayed by the oracle.  (The 128/64-wide variants above cover the option space; this one the size.)"""
  errs, stats, ostats = _step_and_replay(list(FULL_WIDTH), 'fp32', n_patch=4, P=8)
  assert abs(float(stats['loss']) / float(ostats['loss']) - 1) < 1e-4
  worst = max(errs.items(), key=lambda kv: kv[1][0])
  for name, (emax, el2, n) in errs.items():
    assert emax <= 1e-4, f'{name}: max err {emax:.2e} of the leaf max (L2 {el2:.2e}); worst {worst}'


def test_bf16_full_width_gradients_encoder_inside_the_comparison():
  """Round 5: as test_bf16_full_width_gradients_vs_bf16_rounded_oracle, but the oracle computes its OWN
  integrated positional encoding (float32 sin / exp of the reference's formulas, rounded to bf16 as a GEMM operand) instead of
  replaying the product's encoder output: the bf16 k_cast_ipe -- hardware sin / exp, fp32 moments, one rounding -- is inside the
  comparison.  A feature the two sides round to neighbouring bf16 values (2^-8 relative) moves the first layer's
  pre-activations; the ReLU decisions stay the product's.  Bounds from profiles/r05_parity_margins.txt."""
  errs, stats, ostats = _step_and_replay(list(FULL_WIDTH), 'bf16', quant=True, replay_feats=False)
  assert abs(float(stats["loss"]) / float(ostats["loss"]) - 1) < 2e-3      # (measured 2.7e-4)
  big = [(k, v) for k, v in errs.items() if v[2] >= 1024]
  assert len(big) >= 12
  for name, (emax, el2, n) in big:
    assert el2 < 1.5e-2, f"{name}: relative L2 error {el2:.3e} (max {emax:.2e})"      # (measured 8.4e-3 at worst)


def test_train_step_config1_shape_full_width_1024_rays():
  """BASELINE.json configs[0]: base / mse, L=2, 64 proposal + 64 fine samples, 1024 rays, NerfMLP 8x1024 + PropMLP 4x256,
  fp32 parity mode -- the whole step against the oracle (forward 1e-4, statistical gradient check, optimizer update)."""
  gin = [g for g in SMALL if 'net_width' not in g and 'num_nerf_samples' not in g] + [
      "PropMLP.net_width = 256", "NerfMLP.net_width = 1024", "Model.num_nerf_samples = 64", "Config.distortion_loss_mult = 0."]
  _run_case(gin, n_patch=16, P=8, tol_grad=1e-1)


def test_three_step_training_trajectory_vs_oracle():
  """Adam moments, bias correction, lr schedule, clip and the weight re-cast are exercised ACROSS steps: three full train
  steps on both sides from the same weights, same rays, same jitter.  The loss of every step and the parameters after the
  third step (as total update since initialisation) against the oracle's."""
  from tests import hugs_testlib as H
  from oracle import torch_ref as R
  config, model, state, _, train_step, cfg, oparams = H.make_pair(list(SMALL), compute_dtype='fp32')
  batches = [H.synth_rays(1, 8, 5 + i) for i in range(3)]
  N, L = 64, model.num_levels
  gen = torch.Generator(device='cuda').manual_seed(21)
  u01s = [[torch.rand(N, generator=gen, device='cuda') for _ in range(L)] for _ in range(3)]
  fracs = [0.1, 0.35, 0.8]
  theta0 = state.flat.clone()
  names = [n for n, _ in R.flat_leaves(oparams['params'])]
  p = {n: t.clone() for n, t in R.flat_leaves(oparams['params'])}
  m = {n: torch.zeros_like(p[n]) for n in names}
  v = {n: torch.zeros_like(p[n]) for n in names}
  p0 = {n: t.clone() for n, t in p.items()}

  def tree(flatd):
    out = {}
    for n, t in flatd.items():
      d = out
      ks = n.split('/')
      for k in ks[:-1]:
        d = d.setdefault(k, {})
      d[ks[-1]] = t
    return {'params': out}

  hip_losses, orc_losses = [], []
  for i in range(3):
    state, stats, _ = train_step(u01s[i], state, batches[i], fracs[i], None)
    hip_losses.append(float(stats['loss']))
    ostats, ograds, _, _ = R.loss_and_grad(cfg, tree(p), H.oracle_rays(batches[i]), batches[i].rgb.reshape(-1, 3), fracs[i],
                                           [u.cpu() for u in u01s[i]])
    orc_losses.append(float(ostats['loss']))
    p, m, v = R.adam_update(cfg, p, R.clip_gradients(cfg, ograds), m, v, i)
  for i in range(3):
    assert abs(hip_losses[i] / orc_losses[i] - 1) < 1e-4, (i, hip_losses, orc_losses)
  assert hip_losses[2] != hip_losses[0]
  worst = 0.
  for lf in model.layout.leaves:
    name = '/'.join(lf['path'])
    d_hip = (model.layout.view(state.flat, lf['path']) - model.layout.view(theta0, lf['path'])).cpu().double()
    d_orc = (p[name] - p0[name]).double()
    # Adam's first updates are ~lr * sign(g): an entry whose gradient is within rounding of zero may take the other sign
    # on one side -- a population statement (99.9 % of every leaf's entries within 2 % of the largest update), not a max
    sc = float(d_orc.abs().max())
    frac_bad = float(((d_hip - d_orc).abs() > 2e-2 * sc).double().mean())
    assert frac_bad < 1e-3 or d_orc.numel() <= 8, (name, frac_bad)
    worst = max(worst, frac_bad)


def test_bf16_three_levels_two_fused_propmlp_backwards_keep_their_own_relu_masks():
  """with the reference-default three levels both proposal levels run the fused PropMLP backward
  (hugs_mlp256_tail_bwd: bf16, 4 x 256, M >= 2048, equal M).  Its device-pointer table was cached under a key without the level,
  so level 0 ran with level 1's mask-bit buffers.  Here: 3 levels, PropMLP 4 x 256, 64 rays x 64 samples = 4096 rows per proposal
  level; (a) every PropMLP leaf against the bf16-rounded oracle replaying the HIP step's own masks, (b) against the per-layer
  backward (HUGS_MLPFUSE_CHAIN3 off), which never used the table."""
  from nerf_hugs_amd.internal import engine as E
  gin = [g for g in SMALL if 'net_width' not in g and not g.startswith('Model.num_')] + [
      "PropMLP.net_width = 256", "NerfMLP.net_width = 256", "Model.num_levels = 3", "Model.num_prop_samples = 64",
      "Model.num_nerf_samples = 32"]
  errs, stats, ostats = _step_and_replay(list(gin), 'bf16', quant=True)
  assert abs(float(stats['loss']) / float(ostats['loss']) - 1) < 2e-3
  tabs = [k for k in _LAST['engine'].ws.bufs if k[0] == 'mlp_tail_bwd']
  assert len(tabs) == 2, tabs                                   # both proposal levels took the fused backward, each with its own table
  prop = {k: v for k, v in errs.items() if k.startswith('PropMLP') and v[2] >= 1024}
  assert len(prop) >= 4
  for name, (emax, el2, n) in prop.items():
    assert el2 < 2e-2, f'{name}: relative L2 error {el2:.3e} (max {emax:.2e})'
  old = E._MLP_CHAIN3
  try:
    E._MLP_CHAIN3 = False
    errs0, _, _ = _step_and_replay(list(gin), 'bf16', quant=True)
  finally:
    E._MLP_CHAIN3 = old
  for name in prop:
    assert abs(errs[name][1] - errs0[name][1]) < 1e-2, (name, errs[name], errs0[name])

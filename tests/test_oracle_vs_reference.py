"""Pins the oracle (oracle/stepfun_ref.c + oracle/torch_ref.py) against vectors produced
by executing the reference's own leaf modules (tests/golden/gen_fixtures.py), and against
the known-answer tests of the reference's suite restated here (file:line cited per test)."""
import math

import numpy as np
import pytest
import torch

from oracle import cstepfun as C
from oracle import torch_ref as R

CASES = [('cfg2_det', 2), ('cfg2_jit', 2), ('def3_warp', 3), ('def3_nobg', 3), ('cyl', 2)]
T = lambda a: torch.from_numpy(np.asarray(a))


def _u(golden, tag, lvl, S):
  key = f'{tag}/l{lvl}_u01'
  ub, mj = R.sample_u_base(S, key in golden.files)
  if key in golden.files:
    return ub[None, :] + (golden[key] * np.float32(mj)).astype(np.float32)
  return np.broadcast_to(ub, (golden[f'{tag}/l{lvl}_sdist'].shape[0], S)).copy()


@pytest.mark.parametrize('tag,L', CASES)
def test_dilate_sorted_bins_bit_exact(golden, tag, L):
  # stepfun.py:99-128; sorted t-bins must be bit-exact, weights to float rounding.
  for l in range(1, L):
    td, wd = C.max_dilate_weights(golden[f'{tag}/l{l}_in_sdist'], golden[f'{tag}/l{l}_in_weights'],
                                  float(golden[f'{tag}/l{l}_dilation']), 0., 1.)
    assert np.array_equal(td, golden[f'{tag}/l{l}_dil_t_full'])
    np.testing.assert_allclose(wd, golden[f'{tag}/l{l}_dil_w_full'], rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize('tag,L', CASES)
def test_sample_intervals_vs_reference(golden, tag, L):
  # stepfun.py:214-263 fed with the reference's own logits and u.
  for l in range(L):
    sd_ref = golden[f'{tag}/l{l}_sdist']
    S = sd_ref.shape[1] - 1
    t_in = golden[f'{tag}/l{l}_in_sdist'] if l == 0 else golden[f'{tag}/l{l}_dil_t_full'][:, 1:-1]
    sd, idx = C.sample_intervals(_u(golden, tag, l, S), t_in, golden[f'{tag}/l{l}_logits'], 0., 1.)
    # level>0: (u-cw0)/(cw1-cw0) amplifies the 1-ulp summation-order difference of cw
    # (oracle: wave order; fixture: numpy pairwise) by 1/bin-weight -> 2e-5 abs on s in [0,1].
    np.testing.assert_allclose(sd, sd_ref, rtol=0, atol=3e-7 if l == 0 else 2e-5)
    assert np.all(np.diff(sd, axis=-1) >= 0)


@pytest.mark.parametrize('tag,L', CASES)
def test_level_sample_composition(golden, tag, L):
  # models.py:155-212: dilate -> trim -> anneal logits -> sample -> s_to_t, as one call.
  tf = float(golden[f'{tag}/train_frac'])
  anneal = (10 * tf) / (9 * tf + 1)
  rd = 1 if tag == 'def3_warp' else 0
  for l in range(L):
    S = golden[f'{tag}/l{l}_sdist'].shape[1] - 1
    key = f'{tag}/l{l}_u01'
    ub, mj = R.sample_u_base(S, key in golden.files)
    jit = (golden[key][:, 0] * np.float32(mj)).astype(np.float32) if key in golden.files else None
    sd, td, idx = C.level_sample(golden[f'{tag}/l{l}_in_sdist'], golden[f'{tag}/l{l}_in_weights'], l > 0,
                                 float(golden[f'{tag}/l{l}_dilation']), 0., 1., anneal, 0.0, ub, jit, rd,
                                 golden[f'{tag}/near'], golden[f'{tag}/far'])
    np.testing.assert_allclose(sd, golden[f'{tag}/l{l}_sdist'], rtol=0, atol=5e-7 if l == 0 else 2e-5)
    np.testing.assert_allclose(td, golden[f'{tag}/l{l}_tdist'], rtol=1e-4 if rd else 3e-5, atol=1e-7)


def test_kat_linspace(golden):
  # tests/stepfun_test.py:579-586: logits [0,0,100,0,0] on t=[1..6], 10 samples -> linspace(3,4,11)
  ub, _ = R.sample_u_base(10, False)
  out, idx = C.sample_intervals(ub[None], golden['kat_linspace/t'][None], golden['kat_linspace/logits'][None], 1., 6.)
  np.testing.assert_allclose(out[0], np.linspace(3, 4, 11), atol=1e-4)
  np.testing.assert_allclose(out, golden['kat_linspace/out'][None], atol=3e-7)
  assert np.all(idx == 2)


def test_sample_intervals_raises_like_reference():
  # stepfun.py:239-240
  with pytest.raises(ValueError):
    C.sample_intervals(np.zeros((1, 1), np.float32), np.array([[0., 1.]], np.float32), np.zeros((1, 1), np.float32), 0., 1.)


def test_searchsorted_vs_numpy():
  # tests/stepfun_test.py:108-124 (in-range == np.searchsorted) and :79-106 (out of range)
  rng = np.random.default_rng(0)
  a = np.sort(rng.uniform(size=(10, 50)).astype(np.float32), -1)
  v = rng.uniform(a[:, :1], a[:, -1:], size=(10, 30)).astype(np.float32)
  lo, hi = C.searchsorted(a, v)
  for r in range(10):
    ref = np.searchsorted(a[r], v[r], side='right')
    assert np.array_equal(hi[r], ref) and np.array_equal(lo[r], ref - 1)
  lo, hi = C.searchsorted(a, a[:, :1] - 1 + 0 * v)
  assert np.all(lo == 0) and np.all(hi == 0)
  lo, hi = C.searchsorted(a, a[:, -1:] + 1 + 0 * v)
  assert np.all(lo == 49) and np.all(hi == 49)
  tlo, thi = R.searchsorted(T(a), T(v))
  lo, hi = C.searchsorted(a, v)
  assert np.array_equal(tlo.numpy(), lo) and np.array_equal(thi.numpy(), hi)


def test_dilate_brute_force():
  # tests/stepfun_test.py:275-300: dilated value == max over the dilated window (exact)
  rng = np.random.default_rng(1)
  n, dil = 32, 0.02
  t = np.sort(rng.uniform(size=(4, n + 1)).astype(np.float32), -1)
  w = rng.uniform(size=(4, n)).astype(np.float32)
  w /= w.sum(-1, keepdims=True)
  td, wd = C.max_dilate_weights(t, w, dil, -np.inf, np.inf)
  p = w / np.maximum(np.float32(np.finfo(np.float32).eps)**2, np.diff(t))
  for r in range(4):
    for i in range(3 * n):
      tm = td[r, i]
      m = (t[r, :-1] - np.float32(dil) <= tm) & (t[r, 1:] + np.float32(dil) > tm)
      pd_ref = p[r][m].max() if m.any() else 0
      wref = pd_ref * (td[r, i + 1] - td[r, i])
      # renormalised by the same sum -> compare ratios
      if wd[r].sum() > 0 and wref > 0:
        assert abs(wd[r, i] / wref - wd[r].max() / (max(
            (p[r][(t[r, :-1] - np.float32(dil) <= td[r, k]) & (t[r, 1:] + np.float32(dil) > td[r, k])].max(initial=0)
             * (td[r, k + 1] - td[r, k])) for k in range(3 * n)))) < 1e-4


@pytest.mark.parametrize('tag,L', CASES)
def test_float_leaves_vs_reference(golden, tag, L):
  basis = T(golden['basis_ico2'].T.astype(np.float32).copy())
  o, d, radii = T(golden[f'{tag}/o']), T(golden[f'{tag}/d']), T(golden[f'{tag}/radii'])
  for l in range(L):
    td = T(golden[f'{tag}/l{l}_tdist'])
    means, covs = R.cast_rays(td, o, d, radii, 'cylinder' if tag == 'cyl' else 'cone')
    np.testing.assert_allclose(means, golden[f'{tag}/l{l}_means'], rtol=2e-6, atol=1e-6)
    cref = golden[f'{tag}/l{l}_covs']
    np.testing.assert_allclose(covs, cref, rtol=1e-5, atol=1e-6 * np.abs(cref).max())
    if tag == 'def3_warp':
      m2, c2 = R.contract_track_linearize(T(golden[f'{tag}/l{l}_means']), T(cref))
      np.testing.assert_allclose(m2, golden[f'{tag}/l{l}_wmeans'], rtol=2e-6, atol=1e-6)
      wref = golden[f'{tag}/l{l}_wcovs']
      # far=1e6 samples: cov ~ 3e10, J ~ 1e-6 -> J cov J^T cancels ~7 digits in float32; the
      # closed-form Jacobian itself is checked in float64 in test_contract_jacobian_matches_autograd.
      np.testing.assert_allclose(c2, wref, rtol=2e-3, atol=1e-3 * np.abs(wref).max())
      means, covs = T(golden[f'{tag}/l{l}_wmeans']), T(wref)
    lm, lv = R.lift_and_diagonalize(means, covs, basis)
    np.testing.assert_allclose(lm, golden[f'{tag}/l{l}_lift_mean'], rtol=1e-5, atol=2e-6)
    vref = golden[f'{tag}/l{l}_lift_var']
    np.testing.assert_allclose(lv, vref, rtol=1e-4, atol=1e-6 * max(np.abs(vref).max(), 1e-6))
    ipe = R.integrated_pos_enc(T(golden[f'{tag}/l{l}_lift_mean']), T(vref), 0, 12)
    np.testing.assert_allclose(ipe[:2, ::3], golden[f'{tag}/l{l}_ipe_sub'], rtol=1e-5, atol=2e-6)
    w, a, tr = R.compute_alpha_weights(T(golden[f'{tag}/l{l}_density']), td, d, tag not in ('def3_nobg', 'cyl'))
    np.testing.assert_allclose(w, golden[f'{tag}/l{l}_weights'], rtol=1e-5, atol=3e-7)
    np.testing.assert_allclose(a, golden[f'{tag}/l{l}_alpha'], rtol=1e-5, atol=3e-7)
    np.testing.assert_allclose(tr, golden[f'{tag}/l{l}_trans'], rtol=1e-5, atol=3e-7)
    rend = R.volumetric_rendering(T(golden[f'{tag}/l{l}_rgb']), T(golden[f'{tag}/l{l}_weights']), td, 1.0,
                                  T(golden[f'{tag}/far']), True)
    for k, v in rend.items():
      np.testing.assert_allclose(v, golden[f'{tag}/l{l}_rend_{k}'], rtol=2e-5, atol=2e-6, err_msg=k)
  c, w = T(golden[f'{tag}/l{L-1}_sdist']), T(golden[f'{tag}/l{L-1}_weights'])
  for l in range(L - 1):
    cp, wp = T(golden[f'{tag}/l{l}_sdist']), T(golden[f'{tag}/l{l}_weights'])
    lo, hi = R.searchsorted(cp, c)
    assert np.array_equal(lo.numpy(), golden[f'{tag}/l{l}_idx_lo'])
    assert np.array_equal(hi.numpy(), golden[f'{tag}/l{l}_idx_hi'])
    inner, outer = R.inner_outer(c, cp, wp)
    np.testing.assert_allclose(outer, golden[f'{tag}/l{l}_w_outer'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(inner, golden[f'{tag}/l{l}_w_inner'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(R.lossfun_outer(c, w, cp, wp), golden[f'{tag}/l{l}_lossfun_outer'], rtol=1e-3, atol=1e-6)
  np.testing.assert_allclose(R.lossfun_distortion(c, w), golden[f'{tag}/lossfun_distortion'], rtol=1e-5, atol=1e-7)
  np.testing.assert_allclose(R.pos_enc(T(golden[f'{tag}/viewdirs']), 0, 4), golden[f'{tag}/dir_enc'], rtol=1e-5, atol=1e-6)


def test_basis_lr_percentile_contract(golden):
  # geopoly_test.py:76-99 golden basis comes through the reference's generate_basis itself.
  np.testing.assert_allclose(R.generate_basis('icosahedron', 2), golden['basis_ico2'], atol=1e-12)
  np.testing.assert_allclose(R.generate_basis('octahedron', 1), golden['basis_octa1'], atol=1e-12)
  assert R.generate_basis('icosahedron', 2).shape == (21, 3)
  for s, v in zip(golden['lr_steps'], golden['lr_vals']):
    assert abs(R.learning_rate_decay(float(s), 2e-3, 2e-5, 250000, 512, 0.01) / v - 1) < 1e-5
  np.testing.assert_allclose(R.weighted_percentile(T(golden['wp/t']), T(golden['wp/w']), [5, 50, 95]),
                             golden['wp/out'], rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(R.contract(T(golden['contract/x'])), golden['contract/z'], rtol=1e-6, atol=1e-7)


def test_contract_jacobian_matches_autograd():
  # coord_test.py:142-176 spirit: the closed-form Jacobian == autodiff Jacobian.
  x = torch.randn(50, 3, dtype=torch.float64) * 2
  cov = torch.randn(50, 3, 3, dtype=torch.float64)
  cov = cov @ cov.transpose(-1, -2)
  _, c2 = R.contract_track_linearize(x, cov)
  for i in range(50):
    J = torch.autograd.functional.jacobian(R.contract, x[i])
    np.testing.assert_allclose(c2[i], J @ cov[i] @ J.T, rtol=1e-9, atol=1e-12)


def test_ipe_var0_is_posenc():
  # coord_test.py:129-140
  x = torch.rand(20, 3) * 2 - 1
  a = R.integrated_pos_enc(x, torch.zeros_like(x), 0, 5)
  b = R.pos_enc(x, 0, 5, append_identity=False)
  np.testing.assert_allclose(a, b, atol=1e-4)


def test_alpha_weights_delta_density_and_finite_grads():
  # render_test.py:443-463 (delta density -> one-hot) and :408-441 (finite grads over scales)
  td = torch.linspace(0, 1, 33)[None]
  dens = torch.zeros(1, 32); dens[0, 10] = 1e10
  w, _, _ = R.compute_alpha_weights(dens, td, torch.tensor([[0., 0., 1.]]))
  assert w[0, 10] == 1 and w.sum() == 1
  for ls in [-100., -10., 0., 10.]:
    dens = (torch.rand(4, 32) * math.exp(ls)).requires_grad_(True)
    w, _, _ = R.compute_alpha_weights(dens, td.expand(4, -1), torch.randn(4, 3))
    g, = torch.autograd.grad(w.sum() + (w**2).sum(), dens)
    assert torch.isfinite(w).all() and torch.isfinite(g).all()


def test_param_count_matches_published():
  # scripts/generate_tables.ipynb:145 (9,007,493) and :164 (9,012,005 with 1000x4 GLO upstream)
  cfg = R.kubric_cfg()
  n = sum(fi * fo + fo for which in ('nerf', 'prop') for fi, fo in R.mlp_layer_dims(cfg, which))
  assert n == 9007493
  cfg = R.kubric_cfg(num_glo_features=4)
  n = sum(fi * fo + fo for which in ('nerf', 'prop') for fi, fo in R.mlp_layer_dims(cfg, which))
  assert n + 1000 * 4 == 9012005


def test_psnr_golden():
  # image.py:28-30
  assert abs(float(R.mse_to_psnr(torch.tensor(0.01))) - 20.0) < 1e-5


def test_inner_outer_match_interval_loops():
  """stepfun_test.py:697-737 restated (own random numbers): inner / outer measures against the O(n^2) definition --
  outer sums the bins that touch [t0_i, t0_{i+1}], inner the bins that lie inside it -- and lossfun_outer(t,w,t,w)=0."""
  rng = np.random.default_rng(0)
  for _ in range(10):
    d0, d1 = rng.integers(10, 20, 2)
    t0 = np.sort(rng.uniform(size=d0 + 1))
    t1 = np.sort(rng.uniform(size=d1 + 1))
    w0 = np.exp(rng.normal(size=d0))
    inner, outer = R.inner_outer(torch.tensor(t1), torch.tensor(t0), torch.tensor(w0))
    ref_in = [sum(w0[j] for j in range(d0) if t0[j] >= t1[i] and t0[j + 1] < t1[i + 1]) for i in range(d1)]
    ref_out = [sum(w0[j] for j in range(d0) if t0[j + 1] >= t1[i] and t0[j] <= t1[i + 1]) for i in range(d1)]
    np.testing.assert_allclose(inner.numpy(), ref_in, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(outer.numpy(), ref_out, rtol=1e-5, atol=1e-5)
    assert float(R.lossfun_outer(torch.tensor(t0), torch.tensor(w0), torch.tensor(t0), torch.tensor(w0)).max()) < 1e-10


def test_level_sample_per_sample_jitter_vs_reference():
  """Model.single_jitter = False (stepfun.py:203-209, one uniform draw per sample): the reference's sample_intervals executed on two
  chained levels (tests/golden/gen_persample_jitter_fixture.py) against the C oracle's per-sample entry (orc_level_sample_batch_pj)."""
  import os
  z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_persample_jitter.npz'))
  n, S0, S1, dilation, anneal = z['meta']
  n, S0, S1 = int(n), int(S0), int(S1)
  near, far = np.zeros(n, np.float32), np.ones(n, np.float32)
  ub, mj = R.sample_u_base(S0, True)
  sd0, _, _ = C.level_sample(np.tile(np.array([[0., 1.]], np.float32), (n, 1)), np.ones((n, 1), np.float32), False, 0., 0., 1., 1.0, 0.0, ub,
                             (z['l0_u01'] * np.float32(mj)).astype(np.float32), 0, near, far)
  np.testing.assert_allclose(sd0, z['l0_sdist'], rtol=0, atol=5e-7)
  ub, mj = R.sample_u_base(S1, True)
  sd1, _, _ = C.level_sample(z['l0_sdist'], z['l0_weights'], True, float(dilation), 0., 1., float(anneal), 0.0, ub,
                             (z['l1_u01'] * np.float32(mj)).astype(np.float32), 0, near, far)
  np.testing.assert_allclose(sd1, z['l1_sdist'], rtol=0, atol=2e-5)
  assert np.all(np.diff(sd1, axis=-1) >= 0)
  # the per-sample draws matter: with the first column only (the single-jitter form) the result differs
  sdx, _, _ = C.level_sample(z['l0_sdist'], z['l0_weights'], True, float(dilation), 0., 1., float(anneal), 0.0, ub,
                             (z['l1_u01'][:, 0] * np.float32(mj)).astype(np.float32), 0, near, far)
  assert float(np.abs(sdx - z['l1_sdist']).max()) > 1e-3

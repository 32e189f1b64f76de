"""GPU parity of the jax-compatible PRNG through the C ABI: bit-exact against
oracle/threefry_ref.py, the reference's datasets_test golden reproduced on the device, and the train step's key
consumption (train_utils.py:408, models.py:196,230, stepfun.py:207-209)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _u32(t):
  return t.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize('n', [0, 1, 2, 5, 64, 1023, 65536, 1000001])
def test_bits_and_uniform_bit_exact(n):
  from nerf_hugs_amd.internal import random as hr
  from oracle import threefry_ref as T
  for seed in (0, 20200823, 2 ** 40 + 17):
    key = hr.PRNGKey(seed)
    assert _u32(key).tolist() == T.prng_key(seed).tolist()
    okey = T.prng_key(seed)
    assert (_u32(hr.bits(key, (n,))) == T.random_bits(okey, (n,))).all()
    for lo, hi in ((0., 1.), (0., 0.0078), (-1., 1.), (2.5, 2.75)):
      a = hr.uniform(key, (n,), lo, hi).cpu().numpy()
      b = T.uniform(okey, (n,), lo, hi)
      assert a.dtype == np.float32 and (a.view(np.uint32) == b.view(np.uint32)).all(), (seed, lo, hi)
      if n:
        # (with lo != 0 the final rounding of f*(hi-lo)+lo may land on hi: jax's behaviour too)
        assert a.min() >= np.float32(lo) and (a.max() < np.float32(hi) if lo == 0 else a.max() <= np.float32(hi))


@pytest.mark.parametrize('n', [1, 2, 7, 1000, 100001])
def test_normal_vs_oracle(n):
  """random.normal on the device (XLA's single-precision erf_inv polynomial) against the float64 evaluation of the same
  definition (oracle/threefry_ref.py normal, which reproduces the reference's datasets_test golden poses): a few ulp."""
  from nerf_hugs_amd.internal import random as hr
  from oracle import threefry_ref as T
  key, okey = hr.PRNGKey(1234 + n), T.prng_key(1234 + n)
  got = hr.normal(key, (n,)).cpu().numpy()
  ref = T.normal(okey, (n,))
  np.testing.assert_allclose(got, ref, rtol=4e-6, atol=4e-7)
  if n >= 100000:
    assert abs(got.mean()) < 0.02 and abs(got.std() - 1.0) < 0.02 and np.abs(got).max() > 3.5


def test_split_chain_and_shapes():
  from nerf_hugs_amd.internal import random as hr
  from oracle import threefry_ref as T
  rng, orng = hr.PRNGKey(0), T.prng_key(0)
  assert _u32(hr.split(rng)).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
  for i in range(6):
    ks, oks = hr.split(rng, 2 + i), T.split(orng, 2 + i)
    assert (_u32(ks) == oks).all()
    rng, orng = ks[1], oks[1]
  assert hr.uniform(rng, (3, 5, 2)).shape == (3, 5, 2)
  with pytest.raises(TypeError):
    hr.split(torch.zeros(3, dtype=torch.int32, device='cuda'))


def test_reference_dataset_golden_on_device():
  """datasets_test.py:54-104 with every number produced by the HIP path: pixels by hugs_prng_uniform, rays by
  hugs_pixels_to_rays (poses come from random.normal, which only the init path uses: taken from the oracle)."""
  from nerf_hugs_amd.internal import camera_utils as cu, random as hr
  from tests import test_oracle_threefry as O
  rng = hr.PRNGKey(0)
  key, rng = hr.split(rng)
  images = hr.uniform(key, (2, 3, 4, 3))
  np.testing.assert_allclose(images[0].cpu().numpy().ravel(), O.RGB_GT, atol=1e-7, rtol=0)
  _, c2w, pixtocam = O.dummy_dataset()
  x, y = cu.pixel_coordinates(4, 3)
  t = lambda a: torch.from_numpy(np.asarray(a, np.float32)).cuda()
  o, d, v, r = cu.pixels_to_rays(x, y, t(pixtocam), t(c2w[0]))
  np.testing.assert_allclose(o.cpu().numpy().ravel(), np.tile(O.ORIGIN_GT, 12), atol=1e-4, rtol=1e-4)
  np.testing.assert_allclose(d.cpu().numpy().ravel(), O.DIRS_GT, atol=1e-4, rtol=1e-4)


def test_permutation_is_jax_shuffle():
  from nerf_hugs_amd.internal import random as hr
  from oracle import threefry_ref as T
  for n in (1, 7, 4096, 100000):
    p = hr.permutation(hr.PRNGKey(0), n).cpu().numpy()
    assert sorted(p.tolist()) == list(range(n))
    key, x = T.prng_key(0), np.arange(n)
    for _ in range(int(np.ceil(3 * np.log(max(1, n)) / np.log(2 ** 32 - 1)))):
      key, sub = T.split(key)
      x = x[np.argsort(T.random_bits(sub, (n,)), kind='stable')]
    assert (p == x).all()


def test_large_draw_statistics():
  from nerf_hugs_amd.internal import random as hr
  u = hr.uniform(hr.PRNGKey(1), (1 << 24,))
  assert abs(float(u.double().mean()) - .5) < 5e-4 and abs(float(u.double().var()) - 1 / 12) < 5e-4
  assert float(u.min()) >= 0 and float(u.max()) < 1
  b = hr.bits(hr.PRNGKey(1), (1 << 24,))
  ones = sum(int(((b >> s) & 1).sum()) for s in (0, 7, 19, 30))
  assert abs(ones / (4 * (1 << 24)) - .5) < 5e-4


GIN = ["Config.patch_size = 8", "Model.num_levels = 3", "PropMLP.net_depth = 2", "PropMLP.net_width = 128",
       "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 4", "NerfMLP.net_width = 128", "NerfMLP.bottleneck_width = 128",
       "Model.num_prop_samples = 16", "Model.num_nerf_samples = 8"]


@pytest.mark.parametrize('single_jitter', [True, False])
def test_train_step_consumes_keys_like_the_reference(single_jitter):
  from nerf_hugs_amd.internal import configs, random as hr, stepfun, train_utils
  from oracle import threefry_ref as T
  from tests import hugs_testlib as H
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, GIN + [f"Model.single_jitter = {single_jitter}"])
  config = configs.make_config()
  model, state, _, train_step, _ = train_utils.setup_model(config, 0, compute_dtype='fp32')
  batch = H.synth_rays(2, 8, 5)
  N = 128
  rng, orng = hr.PRNGKey(20200823), T.prng_key(20200823)
  # the jitter the model will draw == the oracle's jax.random.uniform with the reference's split order
  orng_next, okey = T.split(orng)
  jit, _ = model.level_jitter(hr.split(rng)[1], N)
  k = okey
  for l, S in enumerate((16, 16, 8)):
    sk, k = T.split(k)
    want = T.uniform(sk, (N, 1 if single_jitter else S), 0., stepfun.sample_u(S, True)[1])
    assert (jit[l].cpu().numpy().view(np.uint32) == want.view(np.uint32)).all(), l
    _, k = T.split(k)
  # the step returns the advanced key, is reproducible from a key, and differs from the deterministic step
  s1, st1, r1 = train_step(rng, state, batch, 0.5, None)
  assert (_u32(r1) == orng_next).all()
  l1 = float(st1['loss'])
  _, state2, _, train_step2, _ = train_utils.setup_model(config, 0, compute_dtype='fp32')
  s2, st2, r2 = train_step2(hr.PRNGKey(20200823), state2, batch, 0.5, None)
  assert float(st2['loss']) == l1 and torch.equal(s1.flat, s2.flat)
  _, state3, _, train_step3, _ = train_utils.setup_model(config, 0, compute_dtype='fp32')
  _, st3, _ = train_step3(hr.PRNGKey(1), state3, batch, 0.5, None)
  assert float(st3['loss']) != l1
  # Model.apply with a key: level-0 sample positions follow u = linspace + jitter exactly
  rend, hist = model.apply(s1.flat, hr.PRNGKey(3), batch.rays, 0.5, False)
  rend2, hist2 = model.apply(s1.flat, hr.PRNGKey(3), batch.rays, 0.5, False)
  assert torch.equal(hist[0]['sdist'], hist2[0]['sdist'])
  rend3, hist3 = model.apply(s1.flat, None, batch.rays, 0.5, False)
  assert not torch.equal(hist[0]['sdist'].clone(), hist3[0]['sdist'])


def test_step_jitter_is_the_chain_of_splits_and_uniforms():
  """hugs_prng_step_jitter (one launch per training step) == train_utils.py:408 split + per level models.py:196 split ->
  stepfun.py:207-209 uniform -> models.py:230 split, bit for bit (odd and even sizes, up to 8 levels)."""
  from nerf_hugs_amd.internal import random as hr
  for seed, sizes, maxvals in ((0, [1024, 1024], [0.0154, 0.0077]), (20200823, [1, 7, 131072 * 3 + 1], [1.0, 0.5, 0.003]),
                               (5, [64] * 8, [0.1 * (i + 1) for i in range(8)]), (9, [], [])):
    key = hr.PRNGKey(seed, 'cuda')
    outs, rng = hr.step_jitter(key, sizes, maxvals)
    want_rng, k = hr.split(key)
    for n, mv, o in zip(sizes, maxvals, outs):
      kl, k = hr.split(k)
      assert torch.equal(o, hr.uniform(kl, (n,), maxval=mv)), (seed, n)
      _, k = hr.split(k)
    assert torch.equal(rng, want_rng)


def test_train_step_consumes_the_stream_as_before():
  """Model.step_jitter (fused) hands the sampler the same draws and returns the same advanced key as split + level_jitter."""
  from nerf_hugs_amd.internal import configs, models
  from nerf_hugs_amd.internal import random as hr
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, ["Model.num_levels = 3", "Model.single_jitter = False"])
  m = models.Model(configs.make_config())
  key = hr.PRNGKey(11, 'cuda')
  fused, rng = m.step_jitter(key, 96)
  want_rng, k = hr.split(key)
  chain, _ = m.level_jitter(k, 96)
  assert torch.equal(rng, want_rng) and len(fused) == 3 and fused.scaled
  for a, b in zip(fused, chain):
    assert a.shape == b.shape and torch.equal(a, b)
  configs.clear_config()


def test_fold_in_and_flax_init_stream_vs_oracle():
  """random.fold_in and Model.init_flax (flax's `model.init(rng, ...)` stream, models.py:348-356: he_uniform Dense kernels,
  nn.Embed's normal tables, keys folded from the parameter path) bit for bit against the restatement in
  oracle/threefry_ref.py, for the three published folding rules.  The restatement itself is UNPINNED (no flax / jax here)."""
  from oracle import threefry_ref as T
  from nerf_hugs_amd.internal import configs, models, random as hr
  key = hr.PRNGKey(20200823)
  okey = T.prng_key(20200823)
  for d in (0, 1, 7, 0x9E3779B9, 0xFFFFFFFF):
    assert np.array_equal(hr.fold_in(key, d).cpu().numpy().view(np.uint32), T.fold_in(okey, d)), d
  # fold_in(key, d) is split's first counter pair generalised: split(key)[0] uses counters (0, 2), fold_in(key, 2) uses (0, 2)
  assert np.array_equal(T.fold_in(okey, 2), np.array([T.split(okey)[0][0], T.split(okey)[1][0]], np.uint32))
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, ["NerfMLP.net_width = 128", "PropMLP.net_width = 128", "NerfMLP.net_depth = 6", "PropMLP.net_depth = 2",
                                                 "Model.num_glo_features = 4", "Model.num_embeddings = 12", "PropMLP.disable_rgb = True"])
  model = models.Model(configs.make_config())
  for variant in ('lazy', 'lazy_sep', 'legacy'):
    flat = model.init(key, flax_rng=variant)
    seen = set()
    for lf in model.layout.leaves:
      got = model.layout.view(flat, lf['path']).cpu().numpy()
      if lf['path'][-1] == 'bias':
        assert not got.any()
        continue
      k = T.flax_param_key(okey, lf['path'][:-1], 1, variant)
      ref = T.he_uniform(k, lf['shape']) if lf['path'][-1] == 'kernel' else T.embed_init(k, lf['shape'])
      if lf['path'][-1] == 'kernel':
        assert np.array_equal(got, ref), (variant, lf['path'])
        lim = np.sqrt(6.0 / lf['shape'][0])
        assert np.abs(got).max() <= lim * (1 + 1e-6) and got.std() > 0.5 * lim      # U(-lim, lim): std = lim / sqrt(3)
      else:
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7)      # (erf_inv: the device polynomial vs scipy, as test_normal_vs_oracle)
      seen.add(tuple(k))
    assert len(seen) == sum(1 for lf in model.layout.leaves if lf['path'][-1] != 'bias'), 'every parameter has its own key'
  # construct_model(key, ...) takes the flax stream, construct_model(int, ...) torch's generator (the distributions only)
  _, v1 = models.construct_model(key, None, configs.make_config())
  assert torch.equal(v1, model.init(key, flax_rng='lazy'))
  configs.clear_config()

"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/hugs.h declares, the
product refuses to compute without a GPU (no fallback), the gin-subset reader parses every reference gin file
form, host helpers match the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  txt = open(os.path.join(ROOT, 'include', 'hugs.h')).read()
  return sorted(set(re.findall(r'\b(hugs_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
  from nerf_hugs_amd import _lib
  lib = _lib.lib().cdll
  names = _declared()
  assert len(names) >= 30
  for n in names:
    assert hasattr(lib, n), f'{n} declared in include/hugs.h but not exported'
  assert lib.hugs_version() >= 10000
  # every prototype the Python binding uses is declared in the header
  for n in _lib._PROTOS:
    assert n in names, n


def test_no_cpu_fallback():
  from nerf_hugs_amd import _lib
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  with pytest.raises(_lib.HugsError):
    _lib.call('hugs_sum', 4, torch.zeros(4), 1.0, torch.zeros(1))
  from nerf_hugs_amd.internal import configs, models
  configs.clear_config()
  m = models.Model(configs.make_config())
  with pytest.raises(_lib.HugsError):
    m.engine('cuda')


def test_product_never_imports_oracle():
  pkg = os.path.join(ROOT, 'nerf-hugs_amd')
  for dp, _, fs in os.walk(pkg):
    for f in fs:
      if f.endswith(('.py', '.hip', '.h', '.sh')):
        txt = open(os.path.join(dp, f)).read()
        assert 'import oracle' not in txt and 'from oracle' not in txt and 'liborc_' not in txt, os.path.join(dp, f)


GIN_FORMS = """
Config.dataset_loader = 'distractor'
Config.near = 0.2
Config.far = 1e6   # trailing comment
Config.transient_type = 'robustnerf'
Config.enable_render_zero_glo = True
# comment line
Model.raydist_fn = @jnp.reciprocal
Model.num_glo_features = 4
PropMLP.warp_fn = @coord.contract
NerfMLP.net_width = 1024
"""


def test_gin_subset_reader(tmp_path):
  from nerf_hugs_amd.internal import configs, models
  p = tmp_path / 'x.gin'
  p.write_text(GIN_FORMS)
  configs.clear_config()
  cfg = configs.load_config([str(p)], ["Config.data_dir = '/a/b'", "Config.patch_size = 16"], save_config=False)
  assert cfg.dataset_loader == 'distractor' and cfg.far == 1e6 and cfg.transient_type == 'robustnerf'
  assert cfg.data_dir == '/a/b' and cfg.patch_size == 16 and cfg.enable_render_zero_glo is True
  m = models.Model(cfg)
  assert m.raydist == 'reciprocal' and m.num_glo_features == 4 and m.nerf_spec.net_width == 1024
  assert m.prop_spec.warp_fn is not None and m.nerf_spec.warp_fn is None
  with pytest.raises(ValueError):
    configs.parse_binding('Model.raydist_fn = @jnp.cosh')
  configs.clear_config()
  assert configs.make_config().data_loss_type == 'charb'          # configs.py:85 default
  # attrs mirrored from models.py:46-72
  m = models.Model(configs.make_config())
  assert (m.num_levels, m.num_prop_samples, m.num_nerf_samples, m.num_embeddings) == (3, 64, 32, 3500)


def test_param_layout_matches_published_counts():
  # scripts/generate_tables.ipynb:145 (9,007,493 params for NerfMLP 8x1024 + PropMLP 4x256)
  from nerf_hugs_amd.internal import configs, models
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, ["PropMLP.net_depth = 4", "PropMLP.net_width = 256",
                                                 "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 8",
                                                 "NerfMLP.net_width = 1024"])
  m = models.Model(configs.make_config())
  assert m.layout.num_params() == 9007493
  names = ['/'.join(l['path']) for l in m.layout.leaves]
  assert 'NerfMLP_0/Dense_11/kernel' in names and 'PropMLP_0/Dense_4/bias' in names
  assert m.layout.by_path[('NerfMLP_0', 'Dense_5', 'kernel')]['shape'] == (1528, 1024)   # skip concat
  flat = m.init(0, 'cpu')
  tree = m.variables(flat)['params']
  k = tree['NerfMLP_0']['Dense_0']['kernel']
  assert k.shape == (504, 1024) and float(k.abs().max()) <= (6 / 504)**0.5 + 1e-6
  assert float(tree['NerfMLP_0']['Dense_0']['bias'].abs().max()) == 0
  configs.clear_config()


def test_errors_like_reference():
  from nerf_hugs_amd.internal import configs, models
  configs.clear_config()
  with pytest.raises(ValueError):
    models.Model(configs.make_config(), ray_shape='sphere')           # render.py:124
  with pytest.raises(ValueError):
    models.Model(configs.make_config(transient_type='bogus'))         # models.py:96-101
  with pytest.raises(AssertionError):                                  # models.py:98-99
    models.Model(configs.make_config(transient_type='nerfw'), num_transient_features=0)
  m = models.Model(configs.make_config(transient_type='nerfw'), num_transient_features=16)
  assert [l['kind'] for l in m.nerf_spec.layers[-7:]] == ['tview', 'ttrunk', 'ttrunk', 'ttrunk', 'tdensity', 'trgb', 'tuncert']
  assert m.prop_spec.num_tra == 0 and 'TransientEmbed_0' in m.layout.modules
  mb = models.Model(configs.make_config(), bg_intensity_range=(0., 1.))       # models.py:246-261: a range -> random draws in training,
  assert mb.bg_random and mb.bg_intensity == 0.5 and mb.has_noise()            # the midpoint when rendering without a key
  with pytest.raises(NotImplementedError):
    models.Model(configs.make_config(), stop_level_grad=False)


def test_host_helpers_match_oracle():
  from nerf_hugs_amd.internal import geopoly, math as hmath, stepfun
  from oracle import torch_ref as R
  for shape, v in [('icosahedron', 2), ('octahedron', 1), ('octahedron', 4)]:
    np.testing.assert_allclose(geopoly.generate_basis(shape, v), R.generate_basis(shape, v), atol=1e-12)
  for s in [0, 10, 512, 125000, 250000]:
    assert abs(hmath.learning_rate_decay(s, 2e-3, 2e-5, 250000, 512, 0.01) /
               R.learning_rate_decay(s, 2e-3, 2e-5, 250000, 512, 0.01) - 1) < 1e-12
  for S in (32, 64, 128):
    for rnd in (False, True):
      a, ma = stepfun.sample_u(S, rnd)
      b, mb = R.sample_u_base(S, rnd)
      assert np.array_equal(a, b) and ma == mb


def test_utils_shard_unshard_and_dummy_rays():
  from nerf_hugs_amd.internal import utils
  r = utils.dummy_rays()
  assert r.origins.shape == (1, 3) and r.embed_idx.dtype == torch.int32     # tests/utils_test.py
  x = torch.arange(24.).reshape(8, 3)
  s = utils.shard(x)
  assert s.shape == (1, 8, 3)
  assert torch.equal(utils.unshard(s, 2), x[:-2])


def test_checkpoint_roundtrip(tmp_path):
  """flax-msgpack state dict round trip (train.py:121,232-236 call pattern) on CPU buffers."""
  from nerf_hugs_amd.internal import checkpoints, configs, models, train_utils
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, ["PropMLP.net_depth = 4", "PropMLP.net_width = 128", "PropMLP.disable_rgb = True",
                                                 "NerfMLP.net_depth = 8", "NerfMLP.net_width = 128", "Model.num_glo_features = 4"])
  cfg = configs.make_config()
  model = models.Model(cfg)
  flat = model.init(3, 'cpu')
  state, _ = train_utils.create_optimizer(cfg, flat, model)
  state.m.normal_(); state.v.uniform_(); state.step = 1234
  p = checkpoints.save_checkpoint(str(tmp_path), state, state.step)
  assert os.path.basename(p) == 'checkpoint_1234'
  d = checkpoints.from_bytes(open(p, 'rb').read())
  assert set(d.keys()) == {'step', 'params', 'opt_state'}
  k = d['params']['params']['NerfMLP_0']['Dense_5']['kernel']
  assert k.shape == (128 + 504, 128) and k.dtype == np.float32        # flax [in,out], skip-concat layer
  assert d['opt_state']['0']['mu']['params']['GloEmbed_0']['embedding'].shape == (3500, 4)   # same tree as TrainState.params
  assert set(d['opt_state']['0'].keys()) == {'count', 'mu', 'nu'} and set(d['opt_state']['1'].keys()) == {'count'}
  state2, _ = train_utils.create_optimizer(cfg, model.init(9, 'cpu'), model)
  state2 = checkpoints.restore_checkpoint(str(tmp_path), state2)
  assert state2.step == 1234
  lay = model.layout
  for lf in lay.leaves:        # logical (unpadded) leaves round-trip exactly; padding rows stay zero
    for a, b in ((state.flat, state2.flat), (state.m, state2.m), (state.v, state2.v)):
      assert torch.equal(lay.view(a, lf['path']), lay.view(b, lf['path']))
  assert checkpoints.restore_checkpoint(str(tmp_path / 'none'), state2) is state2
  configs.clear_config()


def test_every_reference_gin_file_parses():
  """All 19 gin files of the reference load through the gin-subset reader and build a Model (or raise the
  documented NotImplementedError).  Runs only where the reference checkout
  exists (build container); the GPU box has no /root/reference."""
  import glob
  from nerf_hugs_amd.internal import configs, models
  files = sorted(glob.glob('/root/reference/MipNeRF360/configs/*.gin'))
  if not files:
    pytest.skip('reference checkout not present')
  built, refused = 0, []
  for f in files:
    configs.clear_config()
    cfg = configs.load_config([f], ["Config.checkpoint_dir = None"], save_config=False)
    try:
      m = models.Model(cfg)
      assert m.layout.num_params() > 0
      built += 1
    except NotImplementedError:
      refused.append(os.path.basename(f))
  configs.clear_config()
  assert built == len(files) and refused == [], refused      # (debug.gin's 64-wide PropMLP is zero-padded to a tile)


def test_ctypes_prototypes_match_the_header():
  """Every signature string in _lib._PROTOS has the argument kinds include/hugs.h declares (i/f/q/p + stream)."""
  import re
  from nerf_hugs_amd import _lib
  h = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'hugs.h')).read()
  h = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
  checked = 0
  for name, proto in _lib._PROTOS.items():
    m = re.search(r'\b(?:int|long long)\s+' + name + r'\s*\((.*?)\)\s*;', h, re.S)
    assert m, f'{name} is bound but not declared in include/hugs.h'
    args = [a.strip() for a in m.group(1).split(',') if a.strip() and a.strip() != 'void']
    sig = ''.join('p' if '*' in a else 'f' if a.startswith('float') else 'q' if a.startswith('long long') else 'i'
                  for a in args)
    if proto and proto[-1] == 's':
      sig = sig[:-1] + 's'
    assert sig == proto, f'{name}: header says {sig}, _lib says {proto}'
    checked += 1
  assert checked >= 40


def test_geopoly_basis_reference_golden():
  """tests/geopoly_test.py:76-99: the icosahedron-2 basis the MLPs lift with, as a set of directions (each golden row
  matches exactly one basis column up to 1e-4)."""
  from nerf_hugs_amd.internal import geopoly
  golden = np.array([
      [0.85065081, 0.00000000, 0.52573111], [0.80901699, 0.50000000, 0.30901699], [0.52573111, 0.85065081, 0.00000000],
      [1.00000000, 0.00000000, 0.00000000], [0.80901699, 0.50000000, -0.30901699], [0.85065081, 0.00000000, -0.52573111],
      [0.30901699, 0.80901699, -0.50000000], [0.00000000, 0.52573111, -0.85065081], [0.50000000, 0.30901699, -0.80901699],
      [0.00000000, 1.00000000, 0.00000000], [-0.52573111, 0.85065081, 0.00000000], [-0.30901699, 0.80901699, -0.50000000],
      [0.00000000, 0.52573111, 0.85065081], [-0.30901699, 0.80901699, 0.50000000], [0.30901699, 0.80901699, 0.50000000],
      [0.50000000, 0.30901699, 0.80901699], [0.50000000, -0.30901699, 0.80901699], [0.00000000, 0.00000000, 1.00000000],
      [-0.50000000, 0.30901699, 0.80901699], [-0.80901699, 0.50000000, 0.30901699], [-0.80901699, 0.50000000, -0.30901699]])

  def same_basis(x, y, tol=1e-4):      # geopoly_test.py:22-31
    match = np.minimum(((x[:, None, :] - y[None]) ** 2).sum(-1), ((x[:, None, :] + y[None]) ** 2).sum(-1)) <= tol
    return bool(np.all(match.sum(0) == 1) and np.all(match.sum(1) == 1))

  basis = np.asarray(geopoly.generate_basis('icosahedron', 2))
  assert basis.shape == (21, 3) and same_basis(basis, golden)
  np.testing.assert_allclose(np.linalg.norm(basis, axis=-1), 1, atol=1e-12)
  # symmetric directions were removed: no column is the negation of another (geopoly.py:113-121)
  assert (np.abs(basis @ basis.T + 1) < 1e-6).sum() == 0


def test_allreduce_leftover_ranges_cover_every_leaf():
  """The bucketed gradient all-reduce (train_utils.py:457-459 is ONE pmean): whatever the buckets leave out is found
  leaf by leaf -- a tiny leaf (1-float density bias, 3-float rgb bias) sitting between two bucketed ranges must be
  reduced, only alignment padding may stay out."""
  from nerf_hugs_amd.internal import configs, models, train_utils
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, ["Model.num_glo_features = 4"])
  m = models.Model(configs.make_config())
  lay = m.layout
  total = lay.size + train_utils.STAT_TAIL
  span = lambda lf: (lf['off'], lf['off'] + int(np.prod(lf['pshape'])))

  def check(covered):
    todo = train_utils.uncovered_ranges(lay, covered, total)
    flags = np.zeros(total, bool)
    for lo, hi in covered + todo:
      flags[lo:hi] = True
    for lf in lay.leaves:
      lo, hi = span(lf)
      assert flags[lo:hi].all(), f'{"/".join(lf["path"])} is never all-reduced'
    assert flags[lay.size:].all(), 'the stat tail is never all-reduced'
    for (a, b), (c, d) in zip(todo, todo[1:]):
      assert b <= c
    return todo

  assert check([]) == [(0, total)]
  # buckets = every NerfMLP kernel; all the (small) biases between them, the PropMLP, the GLO table and the tail are left over
  kernels = [span(lf) for lf in lay.leaves if lf['path'][0] == 'NerfMLP_0' and lf['path'][-1] == 'kernel']
  todo = check(kernels)
  dens_bias = lay.by_path[('NerfMLP_0', f'Dense_{m.nerf_spec.net_depth}', 'bias')]
  assert int(np.prod(dens_bias['pshape'])) == 1
  assert any(lo <= dens_bias['off'] < hi for lo, hi in todo)
  # a covered 1-float leaf (padded to 4) between two uncovered neighbours is NOT swallowed into a merged range (it would be
  # summed over the ranks twice): nothing that is reduced here may overlap a bucket
  i = next(k for k, lf in enumerate(lay.leaves) if lf is dens_bias)
  todo = check([span(dens_bias)])
  assert not any(lo < span(dens_bias)[1] and span(dens_bias)[0] < hi for lo, hi in todo)
  assert any(hi == span(lay.leaves[i - 1])[1] for lo, hi in todo) and any(lo == span(lay.leaves[i + 1])[0] for lo, hi in todo)
  # a bucket that cuts a leaf in half does not count as covering it -- and reducing the whole leaf on top of the half bucket
  # would double-count: that is refused loudly
  k0 = kernels[0]
  with pytest.raises(RuntimeError):
    check([(k0[0], (k0[0] + k0[1]) // 2)] + kernels[1:])
  configs.clear_config()


def test_stat_tail_slots_do_not_overlap_for_any_level_count():
  from nerf_hugs_amd.internal import train_utils
  for L in range(1, 8):
    o = train_utils._tail_slots(L)
    spans = sorted([(o['data'], 2 * L), (o['interlevel'], L - 1), (o['distortion'], 1), (o['robust'], 5 * L), (o['hanerf'], 2),
                    (o['nerfw'], 2)])
    end = 0
    for lo, n in spans:
      assert lo >= end
      end = lo + n
    assert end <= train_utils.STAT_TAIL
  with pytest.raises(NotImplementedError):
    train_utils._tail_slots(8)


def test_nerfacto_yml_configs():
  """nerfacto/configs/*.yml -> NerfactoConfig: every shipped nerfacto yml parses (the NeRF-W ones raise the NameError the
  reference's own model dies with), and the restated BASELINE config-5 dictionary equals the yml.  Needs the reference
  checkout (build container only)."""
  import glob
  import yaml
  ref = '/root/reference/nerfacto/configs'
  if not os.path.isdir(ref):
    pytest.skip('reference checkout not present')
  from nerf_hugs_amd.nerfacto import configs as C
  seen = 0
  for p in sorted(glob.glob(os.path.join(ref, '*nerfacto*.yml'))):
    if 'nerfw' in p:
      with pytest.raises(NameError):
        C.load_yml(p)
      continue
    c = C.load_yml(p)
    seen += 1
    assert c.num_proposal_iterations == len(c.num_proposal_samples_per_ray)
    if 'hanerf' in p:
      assert c.transient_type == 'hanerf' and c.use_transient_embedding
  assert seen >= 10
  kw = C.yml_to_kwargs(yaml.safe_load(open(os.path.join(ref, 'phototourism_nerfacto_base.yml'))))
  kw.pop('use_transient_embedding')
  assert kw == C.PHOTOTOURISM_NERFACTO_BASE


def test_stop_level_grad_false_is_refused_at_construction():
  """Model.stop_level_grad = False (models.py:55,208-209: gradients through the resampling step) is not built: it raises when
  the model is configured, not somewhere inside a step (INTEGRATION.md lists the refused options)."""
  from nerf_hugs_amd.internal import configs, models
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, ["Model.stop_level_grad = False"])
  with pytest.raises(NotImplementedError, match='stop_level_grad'):
    models.Model(configs.make_config())
  configs.clear_config()

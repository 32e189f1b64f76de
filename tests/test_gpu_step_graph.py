"""The train step replayed from a captured hipGraph (train_utils.create_train_step, HUGS_STEP_GRAPH) against the same steps
enqueued launch by launch: the same kernels in the same order on one stream, so parameters, Adam moments, the advanced
jax key and every stat must be BIT-identical, with the per-step scalars (annealing factor, learning rate, bias corrections)
changing from step to step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_train_step import SMALL, HANERF, NERFW


def _run(mode, gin, nsteps, rng_kind, n_patch=2, P=8):
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import train_utils, random as hr
  old = train_utils._STEP_GRAPH
  train_utils._STEP_GRAPH = mode
  try:
    config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(gin, compute_dtype='bf16')
    batches = [H.synth_rays(n_patch, P, 5 + (i % 3)) for i in range(nsteps)]      # fresh tensors every step: inputs are staged
    key = hr.PRNGKey(123) if rng_kind == 'key' else None
    out = []
    for i in range(nsteps):
      state, stats, key = train_step(key, state, batches[i], 0.1 + 0.07 * i, None)
      out.append((float(stats['loss']), float(stats['psnr']), float(stats['grad_norms']['NerfMLP_0']), float(stats['opt_update_maxes']['PropMLP_0'])))
    torch.cuda.synchronize()
    return (state.flat.clone(), state.m.clone(), state.v.clone(), None if key is None else key.clone(), out, state.step,
            train_step.graph_active(), model.layout)
  finally:
    train_utils._STEP_GRAPH = old


@pytest.mark.parametrize('rng_kind', ['key', 'none'])
@pytest.mark.parametrize('variant', ['base', 'withmask_glo', 'robustnerf', 'no_viewdirs_glo'])
def test_graph_replay_is_bit_identical_to_the_eager_step(rng_kind, variant):
  gin = list(SMALL)
  kw = {}
  if variant == 'no_viewdirs_glo':      # no view layer: nothing in the forward pass waits for the ray encodings of the weight-cast lane
    gin += ["Model.use_viewdirs = False", "Model.num_glo_features = 4"]
  if variant == 'withmask_glo':
    gin += ["Config.transient_type = 'withmask'", "Model.num_glo_features = 4", "Config.data_loss_type = 'charb'"]
  if variant == 'robustnerf':      # thresholds fed back on the device from step to step (train.py:145-148 through the host)
    gin = [g_.replace('patch_size = 8', 'patch_size = 16') for g_ in gin] + ["Config.transient_type = 'robustnerf'", "Config.robustnerf_inlier_quantile = 0.8"]
    kw = dict(n_patch=1, P=16)
  e = _run('0', gin, 7, rng_kind, **kw)
  g = _run('1', gin, 7, rng_kind, **kw)
  assert not e[6] and g[6], 'the graph path did not engage'
  assert e[5] == g[5] == 7
  for a, b, name in zip(e[:3], g[:3], ('params', 'adam m', 'adam v')):
    if variant == 'withmask_glo':
      # the GLO rows are a float-atomic scatter-add (hugs_embed_scatter_add): equal to rounding only within a step, and the
      # rounding reaches every parameter through the next step's forward pass -- two EAGER runs differ by as much
      sc = float(a.abs().max())
      assert float((a - b).abs().max()) <= 2e-3 * sc, (name, float((a - b).abs().max()), sc)
    else:
      assert torch.equal(a, b), name
  if rng_kind == 'key':
    assert torch.equal(e[3], g[3]), 'jax key after 7 steps'
  if variant in ('base', 'robustnerf', 'no_viewdirs_glo'):
    assert e[4] == g[4], (e[4], g[4])
  else:
    np.testing.assert_allclose(np.array(e[4]), np.array(g[4]), rtol=2e-3)


@pytest.mark.parametrize('variant', ['hanerf', 'nerfw'])
def test_graph_replay_of_the_transient_variants(variant):
  """HA-NeRF (its mask-size weight decays with the step: a device-resident scalar of the captured step) and NeRF-W replayed from the
  graph against the eager steps.  Their embedding rows are float-atomic scatter-adds (hugs_embed_scatter_add, k_glo_bwd), so two
  runs agree to rounding, not bit for bit -- the bound is the one the GLO variant above uses."""
  gin = list(HANERF if variant == 'hanerf' else NERFW)
  e = _run('0', gin, 6, 'key')
  g = _run('1', gin, 6, 'key')
  assert not e[6] and g[6], 'the graph path did not engage'
  assert e[5] == g[5] == 6
  for a, b, name in zip(e[:3], g[:3], ('params', 'adam m', 'adam v')):
    sc = float(a.abs().max())
    assert float((a - b).abs().max()) <= 2e-3 * sc, (name, float((a - b).abs().max()), sc)
  assert torch.equal(e[3], g[3]), 'jax key after 6 steps'
  np.testing.assert_allclose(np.array(e[4]), np.array(g[4]), rtol=2e-3)


def _run_finetune(mode, gin, nsteps):
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import train_utils, random as hr
  old = train_utils._STEP_GRAPH
  train_utils._STEP_GRAPH = mode
  try:
    config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(gin + ["Config.finetune_enable = True"], compute_dtype='bf16')
    key = hr.PRNGKey(7)
    for i in range(2):                     # the training stage first: every gradient slot and both Adam moments are non-trivial
      state, _, key = train_step(key, state, H.synth_rays(2, 8, 11 + i), 0.5, None)
    fstate, ftrain, _ = train_utils.setup_finetune_model(config, model, state)
    theta0 = fstate.flat.clone()
    out = []
    for i in range(nsteps):
      fstate, stats, key = ftrain(key, fstate, H.synth_rays(2, 8, 5 + (i % 3)), 1.0, None)
      out.append((float(stats['loss']), float(stats['psnr']), float(stats['grad_norms']['NerfMLP_0']), float(stats['opt_update_maxes']['GloEmbed_0'])))
    torch.cuda.synchronize()
    return fstate.flat.clone(), fstate.m.clone(), fstate.v.clone(), key.clone(), out, fstate.step, ftrain.graph_active(), model.layout, theta0
  finally:
    train_utils._STEP_GRAPH = old


@pytest.mark.parametrize('variant', ['glo', 'nerfw'])
def test_graph_replay_of_the_finetune_stage(variant):
  """train.py:97-109: after training, create_train_step(model, config, True) + the embedding-only optimizer.  The finetune step is
  captured like the training step; against the eager enqueue it agrees to the rounding of the GLO scatter-add, and in both only the
  GLO table moves."""
  gin = list(NERFW) if variant == 'nerfw' else list(SMALL) + ["Model.num_glo_features = 4"]
  e = _run_finetune('0', gin, 6)
  g = _run_finetune('1', gin, 6)
  assert not e[6] and g[6], 'the graph path did not engage'
  assert e[5] == g[5] == 6
  for a, b, name in zip(e[:3], g[:3], ('params', 'adam m', 'adam v')):
    sc = float(a.abs().max())
    assert float((a - b).abs().max()) <= 2e-3 * sc, (name, float((a - b).abs().max()), sc)
  assert torch.equal(e[3], g[3])
  np.testing.assert_allclose(np.array(e[4]), np.array(g[4]), rtol=2e-3)
  for r in (e, g):
    lay = r[7]
    for lf in lay.leaves:
      d = float((lay.view(r[0], lf['path']) - lay.view(r[8], lf['path'])).abs().max())
      assert (d > 0) == ('/'.join(lf['path']) == 'GloEmbed_0/embedding'), lf['path']


def test_graph_is_not_used_where_the_step_cannot_be_captured():
  """RobustNeRF thresholds fed from the host, explicit jitter draws, a torch.Generator: eager enqueue, same results as ever."""
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import train_utils
  old = train_utils._STEP_GRAPH
  train_utils._STEP_GRAPH = '1'
  try:
    config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(list(SMALL), compute_dtype='bf16')
    batch = H.synth_rays(1, 8, 5)
    gen = torch.Generator(device='cuda').manual_seed(3)
    for _ in range(4):
      state, stats, gen = train_step(gen, state, batch, 0.5, None)
    assert not train_step.graph_active() and np.isfinite(float(stats['loss']))
  finally:
    train_utils._STEP_GRAPH = old


def test_stage_step_pub_and_hanerf_loss_dyn_through_the_c_abi():
  """hugs_stage_step_pub: <= 16 copies + <= 8 scalars + the two publish addresses in one launch; hugs_hanerf_loss_dyn == hugs_hanerf_loss
  with the mask-size weight read from device memory."""
  import ctypes
  from nerf_hugs_amd import _lib as L
  dev = 'cuda'
  g = torch.Generator(device=dev).manual_seed(0)
  srcs = [torch.randn(n, generator=g, device=dev) for n in (3, 1024, 70001)] + [torch.randint(0, 1 << 30, (2,), generator=g, device=dev, dtype=torch.int32)]
  dsts = [torch.zeros_like(t) for t in srcs]
  a = np.array([t.data_ptr() for t in srcs], np.uint64); b = np.array([t.data_ptr() for t in dsts], np.uint64)
  w = np.array([t.numel() for t in srcs], np.int32)
  dyn = torch.zeros(8, device=dev)
  ptrs = torch.zeros(2, dtype=torch.int64, device=dev)
  sc = np.array([0.5, 1e-3, 0.25, 0.125, 7.0, 0, 0, 0], np.float32)
  L.call('hugs_stage_step_pub', len(srcs), a.ctypes.data, b.ctypes.data, w.ctypes.data, dyn, 5, sc.ctypes.data, ptrs, 0x1234560, dsts[0])
  torch.cuda.synchronize()
  for s_, d_ in zip(srcs, dsts):
    assert torch.equal(s_, d_)
  assert dyn.cpu().tolist() == [0.5, float(np.float32(1e-3)), 0.25, 0.125, 7.0, 0.0, 0.0, 0.0]
  assert ptrs.cpu().tolist() == [0x1234560, dsts[0].data_ptr()]
  with pytest.raises(ValueError):      # (argument errors surface as ValueError, as the reference's do)
    L.call('hugs_stage_step_pub', 0, None, None, None, dyn, 9, sc.ctypes.data, ptrs, 0, None)      # > 8 scalars
  # HA-NeRF loss, by-value vs device-resident weight
  N, Lv = 256, 2
  pred, gt, mask = torch.rand(Lv, N, 3, generator=g, device=dev), torch.rand(N, 3, generator=g, device=dev), torch.rand(N, generator=g, device=dev)
  coef = torch.tensor([0.1, 1.0], device=dev)
  outs = []
  for form in (0, 1):
    dp, dm, st = torch.zeros(Lv, N, 3, device=dev), torch.zeros(N, device=dev), torch.zeros(2 * Lv + 2, device=dev)
    if form == 0:
      L.call('hugs_hanerf_loss', N, Lv, pred, gt, mask, 1, 1e-3, coef, 0.0371, dp, dm, st)
    else:
      L.call('hugs_hanerf_loss_dyn', N, Lv, pred, gt, mask, 1, 1e-3, coef, torch.tensor([0.0371], device=dev), dp, dm, st)
    outs.append((dp, dm, st))
  for x, y in zip(*outs):
    assert torch.equal(x, y)


def _mixed_sequence(mode):
  """Steps at two batch sizes, renderings between them (Model.apply re-casts weights from whatever buffer it is handed), a parameter
  restore in the middle (load into the SAME flat buffer: its version count moves) and a cloned-parameter rendering."""
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import train_utils, random as hr
  old = train_utils._STEP_GRAPH
  train_utils._STEP_GRAPH = mode
  try:
    gin = list(SMALL) + ["Model.num_glo_features = 4"]
    config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(gin, compute_dtype='bf16')
    key = hr.PRNGKey(7)
    big = [H.synth_rays(2, 8, 20 + i) for i in range(4)]
    small = [H.synth_rays(1, 8, 40 + i) for i in range(4)]
    probe = H.synth_rays(1, 8, 99)
    snap = None
    outs = []
    for i in range(10):
      b = (big if i % 3 else small)[i % 4]
      state, stats, key = train_step(key, state, b, 0.05 * i, None)
      if i == 3:
        snap = (state.flat.clone(), state.m.clone(), state.v.clone(), state.step)
      if i in (2, 5, 8):
        rend, _ = model.apply(state.flat, None, probe.rays, 1.0, False)
        outs.append(rend[-1]['rgb'].clone())
      if i == 5:
        rend, _ = model.apply(state.flat.clone(), None, probe.rays, 1.0, False)      # another buffer: another cast table
        outs.append(rend[-1]['rgb'].clone())
      if i == 6:      # restore the step-3 snapshot in place
        state.flat.copy_(snap[0]); state.m.copy_(snap[1]); state.v.copy_(snap[2]); state.step = snap[3]
      outs.append(torch.tensor(float(stats['loss'])))
    torch.cuda.synchronize()
    return state.flat.clone(), state.m.clone(), key.clone(), outs, train_step.graph_active()
  finally:
    train_utils._STEP_GRAPH = old


def test_mixed_sequence_graph_vs_eager():
  e = _mixed_sequence('0')
  g = _mixed_sequence('1')
  assert not e[4] and g[4]
  # (GLO rows: float-atomic scatter-adds -- agreement to rounding, as in the GLO variant above)
  for a, b, name in ((e[0], g[0], 'params'), (e[1], g[1], 'adam m')):
    sc = float(a.abs().max())
    assert float((a - b).abs().max()) <= 2e-3 * sc, (name, float((a - b).abs().max()), sc)
  assert torch.equal(e[2], g[2]), 'jax key'
  assert len(e[3]) == len(g[3])
  for i, (a, b) in enumerate(zip(e[3], g[3])):
    assert float((a.float() - b.float()).abs().max()) <= 2e-3 * max(1.0, float(a.float().abs().max())), i


def test_captured_step_takes_a_key_that_lives_on_the_cpu():
  """the replay path stages its inputs by raw address; a jax key restored on the CPU (a checkpointed rng, PRNGKey(seed,
  device='cpu')) handed to a step that is already captured must be moved to the device first, not dereferenced as a host pointer.  The
  step with the CPU copy of a key must give the result of the step with the device key."""
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import train_utils, random as hr
  old = train_utils._STEP_GRAPH
  train_utils._STEP_GRAPH = '1'
  try:
    outs = []
    for on_cpu in (False, True):
      config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(list(SMALL), compute_dtype='bf16')
      batch = H.synth_rays(2, 8, 5)
      key = hr.PRNGKey(7)
      for i in range(4):      # two eager steps, the capture, one replay
        state, stats, key = train_step(key, state, batch, 0.2, None)
      assert train_step.graph_active()
      fresh = hr.PRNGKey(99)
      k_in = fresh.cpu() if on_cpu else fresh
      state, stats, key = train_step(k_in, state, batch, 0.3, None)
      torch.cuda.synchronize()
      outs.append((state.flat.clone(), key.clone(), float(stats['loss'])))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1].cpu(), outs[1][1].cpu()) and outs[0][2] == outs[1][2]
  finally:
    train_utils._STEP_GRAPH = old

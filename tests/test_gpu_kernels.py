"""Per-kernel GPU parity through the C ABI: bit-exact for the sampler (sample indices, sorted t-bins, sdist,
tdist), float tolerances stated per test elsewhere.  Includes the committed golden vectors produced by the
reference's own source and size-independent properties at the BASELINE batch size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = 'cuda'
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)


def _L():
  from nerf_hugs_amd import _lib
  return _lib


def test_portable_explog_and_ieee_ops_bit_exact():
  from oracle import cstepfun as C
  L = _L()
  x = torch.cat([torch.linspace(-104, 89, 200001), torch.rand(100000) * 1e-30, torch.rand(100000), torch.rand(1000) * 1e-42]).float()
  ye = torch.empty_like(x, device=dev); yl = torch.empty_like(x, device=dev)
  L.call('hugs_test_explog', x.to(dev), x.numel(), ye, yl)
  assert np.array_equal(bits(ye.cpu().numpy()), bits(C.expf(x.numpy())))
  assert np.array_equal(bits(yl.cpu().numpy()), bits(C.logf(x.numpy())))
  rng = np.random.default_rng(0)
  n = 1 << 18
  for lo, hi in [(-5, 5), (-80, 0), (-100, -60)]:
    a = (np.exp(rng.uniform(lo, hi, n)) * rng.choice([-1, 1], n)).astype(np.float32)
    b = (np.exp(rng.uniform(-20, 20, n)) * rng.choice([-1, 1], n)).astype(np.float32)
    o = torch.empty(4 * n, device=dev)
    L.call('hugs_test_arith', G(a), G(b), n, o)
    o = o.cpu().numpy().reshape(4, n)
    with np.errstate(all='ignore'):
      for k, ref in enumerate([a / b, a * b, a + b, a - b]):
        assert np.array_equal(bits(o[k]), bits(ref))


def _run_level(N, n_prev, ns, dil, raydist, jitter, seed, wpow=1, zeros=False, anneal=0.7, order=1):
  from oracle import cstepfun as C, torch_ref as R
  from nerf_hugs_amd.internal import stepfun
  rng = np.random.default_rng(seed)
  t = np.sort(rng.uniform(0, 1, (N, n_prev + 1)).astype(np.float32), -1) if n_prev > 1 else np.tile(np.array([[0., 1.]], np.float32), (N, 1))
  w = rng.uniform(0, 1, (N, n_prev)).astype(np.float32)**wpow
  if zeros:
    w[rng.uniform(size=w.shape) < 0.1] = 0
  w /= np.maximum(w.sum(-1, keepdims=True), 1e-9)
  u01 = rng.random(N, dtype=np.float32) if jitter else None
  ub, mj = R.sample_u_base(ns, jitter)
  jit = (u01 * np.float32(mj)).astype(np.float32) if jitter else None
  near = rng.uniform(0.05, 0.3, N).astype(np.float32)
  far = np.full(N, {0: 1.2, 1: 1e6, 3: 5.0}.get(raydist, 300.0), np.float32)
  sd_o, td_o, idx_o = C.level_sample(t, w, dil is not None, dil or 0., 0., 1., anneal, 0., ub, jit, raydist, near, far, sum_order=order)
  sd, td, idx, tin, win = stepfun.level_sample(G(t), G(w), dil is not None, dil or 0., (0., 1.), anneal, 0., ns,
                                               None if u01 is None else G(u01), [None, 'reciprocal', 'log', 'exp', 'sqrt', 'square', 'piecewise'][raydist],
                                               G(near), G(far), return_debug=True, sum_order=order)
  assert np.array_equal(idx.cpu().numpy(), idx_o), 'interval indices'
  assert np.array_equal(bits(sd.cpu().numpy()), bits(sd_o)), 'sdist'
  assert np.array_equal(bits(td.cpu().numpy()), bits(td_o)), 'tdist'
  if dil is not None:
    tdil, wdil = C.max_dilate_weights(t, w, dil, 0., 1., sum_order=order)
    assert np.array_equal(bits(tin.cpu().numpy()), bits(tdil[:, 1:-1])), 'sorted dilated t-bins'
    assert np.array_equal(bits(win.cpu().numpy()), bits(wdil[:, 1:-1])), 'dilated weights'
  s = sd.cpu().numpy()
  assert np.all(np.diff(s, axis=-1) >= 0) and s.min() >= 0 and s.max() <= 1


@pytest.mark.parametrize('case', [
    dict(N=1000, n_prev=1, ns=64, dil=None, raydist=0, jitter=True),
    dict(N=1000, n_prev=64, ns=128, dil=0.0103125, raydist=0, jitter=True),
    dict(N=1000, n_prev=64, ns=64, dil=0.0103125, raydist=1, jitter=False),
    dict(N=999, n_prev=64, ns=32, dil=0.00262, raydist=1, jitter=True, wpow=4),
    dict(N=513, n_prev=85, ns=256, dil=0.003, raydist=0, jitter=True, zeros=True),
    dict(N=3, n_prev=2, ns=2, dil=0.1, raydist=0, jitter=False),
    dict(N=257, n_prev=190, ns=128, dil=None, raydist=0, jitter=True, wpow=6, zeros=True, anneal=1.0),
    # blender/llff gins: 128 proposal samples -> 382 dilated bins: 8-per-lane canonical order (capacity 512)
    dict(N=300, n_prev=128, ns=32, dil=0.0064, raydist=0, jitter=True, wpow=3, zeros=True),
    dict(N=64, n_prev=128, ns=128, dil=0.0064, raydist=1, jitter=False),
    dict(N=33, n_prev=400, ns=300, dil=None, raydist=0, jitter=True, wpow=5),
    # 256 samples per level (the per-ray kernels' limit) -> 766 dilated bins: 16-per-lane order (capacity 1024)
    dict(N=40, n_prev=256, ns=256, dil=0.0031, raydist=1, jitter=True, wpow=3, zeros=True),
    dict(N=17, n_prev=256, ns=64, dil=0.0031, raydist=0, jitter=False),
    dict(N=9, n_prev=1000, ns=1024, dil=None, raydist=0, jitter=True, wpow=4),
    # numpy's pairwise sum keeps halving until a block is <= 128: 1015 -> 511 -> 263 -> 135 needs a FOURTH split
    dict(N=7, n_prev=1015, ns=64, dil=None, raydist=0, jitter=True, wpow=3),
    dict(N=7, n_prev=1023, ns=64, dil=None, raydist=0, jitter=False, wpow=2),
    dict(N=5, n_prev=339, ns=32, dil=0.002, raydist=0, jitter=True, wpow=3),       # 3*339 - 2 = 1015 dilated bins
    dict(N=5, n_prev=341, ns=32, dil=0.002, raydist=1, jitter=True),               # 1021: the largest dilated level
    # the other raydist_fn curves of coord.py:84-90 (log, exp, sqrt, square)
    dict(N=200, n_prev=64, ns=64, dil=0.0103125, raydist=2, jitter=True),
    dict(N=200, n_prev=64, ns=64, dil=0.0103125, raydist=3, jitter=False),
    dict(N=200, n_prev=64, ns=128, dil=0.0103125, raydist=4, jitter=True),
    dict(N=200, n_prev=64, ns=32, dil=0.0103125, raydist=5, jitter=True),
    dict(N=200, n_prev=64, ns=64, dil=0.0103125, raydist=6, jitter=True),       # 'piecewise' (coord.py:81-84)
])
@pytest.mark.parametrize('order', [1, 0], ids=['reference_order', 'wave_order'])
def test_level_sample_bit_exact_vs_oracle(case, order):
  """Both summation orders of hugs_level_sample_fwd (1 = reference order: numpy-pairwise sums + sequential cumsum, the
  one that ships; 0 = wave order) against the C oracle in the same order: sample indices, sorted t-bins, sdist, tdist
  bit for bit."""
  _run_level(seed=3, order=order, **case)


def test_level_sample_errors_and_empty():
  from nerf_hugs_amd.internal import stepfun
  t = torch.tensor([[0., 1.]], device=dev); w = torch.ones(1, 1, device=dev); z = torch.zeros(1, device=dev)
  with pytest.raises(ValueError):                     # stepfun.py:239-240
    stepfun.level_sample(t, w, False, 0., (0., 1.), 1., 0., 1, None, None, z, z + 1)
  from nerf_hugs_amd import _lib
  with pytest.raises(_lib.HugsError):                  # 3*342 bins > capacity 1024
    stepfun.level_sample(torch.rand(2, 343, device=dev).sort(-1).values, torch.rand(2, 342, device=dev), True, 0.01, (0., 1.), 1., 0.,
                         8, None, None, torch.zeros(2, device=dev), torch.ones(2, device=dev))
  sd, td = stepfun.level_sample(t[:0], w[:0], False, 0., (0., 1.), 1., 0., 8, None, None, z[:0], z[:0])
  assert sd.shape == (0, 9)


def test_level_sample_golden_reference_vectors(golden):
  """The reference's own sample_intervals outputs (tests/golden): level 0 within float rounding; KAT of
  tests/stepfun_test.py:579-586 (linspace(3,4,11))."""
  from nerf_hugs_amd.internal import stepfun
  for tag in ['cfg2_det', 'cfg2_jit', 'def3_nobg']:
    key = f'{tag}/l0_u01'
    u01 = G(golden[key][:, 0]) if key in golden.files else None
    ref = golden[f'{tag}/l0_sdist']
    sd, td = stepfun.level_sample(G(golden[f'{tag}/l0_in_sdist']), G(golden[f'{tag}/l0_in_weights']), False, 0., (0., 1.), 0.5, 0.,
                                  ref.shape[1] - 1, u01, None, G(golden[f'{tag}/near'][:, 0]), G(golden[f'{tag}/far'][:, 0]))
    np.testing.assert_allclose(sd.cpu().numpy(), ref, atol=5e-7)
    np.testing.assert_allclose(td.cpu().numpy(), golden[f'{tag}/l0_tdist'], rtol=3e-6, atol=1e-7)
  t = G(golden['kat_linspace/t'][None]); lg = golden['kat_linspace/logits'][None]
  w = np.exp(lg - lg.max()); w /= w.sum()
  sd, _ = stepfun.level_sample(t, G(w.astype(np.float32)), False, 0., (1., 6.), 1., 0., 10, None, None,
                               torch.zeros(1, device=dev), torch.ones(1, device=dev))
  np.testing.assert_allclose(sd.cpu().numpy()[0], np.linspace(3, 4, 11), atol=1e-4)


@pytest.mark.parametrize('warp', [0, 1])
def test_cast_ipe_vs_oracle_and_golden(golden, warp):
  from oracle import torch_ref as R
  L = _L()
  N, S = 64, 128
  g = torch.Generator().manual_seed(1)
  basis = torch.tensor(R.generate_basis('icosahedron', 2).T.copy(), dtype=torch.float32)
  o = torch.randn(N, 3, generator=g) * 0.5
  d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1) * (0.8 + 0.4 * torch.rand(N, 1, generator=g))
  radii = 5e-4 + 1.5e-3 * torch.rand(N, 1, generator=g)
  td = torch.sort(torch.rand(N, S + 1, generator=g) * 3 + 0.1, -1).values
  means, covs = R.cast_rays(td, o, d, radii)
  if warp:
    means, covs = R.contract_track_linearize(means, covs)
  lm, lv = R.lift_and_diagonalize(means, covs, basis)
  ref = R.integrated_pos_enc(lm, lv, 0, 12).reshape(N * S, 504)
  out = torch.empty(N * S, 512, device=dev)
  args = (N, S, td.to(dev), o.to(dev), d.to(dev), radii.reshape(-1).to(dev), basis.to(dev), 21, 0, warp, 12)
  L.call('hugs_cast_ipe_fwd', *args, 0, 512, out)
  # |feature| <= 1; high degrees multiply the mean by 2^11, so 1e-7 relative on the mean is 2e-4 absolute phase
  assert float((out.cpu()[:, :504] - ref).abs().max()) < 2e-3
  assert float((out.cpu()[:, :504] - ref).abs().mean()) < 2e-5
  assert float(out[:, 504:].abs().max()) == 0
  outb = torch.empty(N * S, 512, device=dev, dtype=torch.bfloat16)
  L.call('hugs_cast_ipe_fwd', *args, 1, 512, outb)
  assert float((outb.float().cpu()[:, :504] - ref).abs().max()) < 6e-3
  with pytest.raises(ValueError):
    L.call('hugs_cast_ipe_fwd', *args[:8], 2, *args[9:], 0, 512, out)       # bad ray_shape, render.py:124
  if not warp:   # the reference's own vectors (cone, no warp), low degrees exactly, all degrees loosely
    tag = 'cfg2_det'
    td = G(golden[f'{tag}/l1_tdist']); n, s = td.shape[0], td.shape[1] - 1
    o2 = torch.empty(n * s, 512, device=dev)
    L.call('hugs_cast_ipe_fwd', n, s, td, G(golden[f'{tag}/o']), G(golden[f'{tag}/d']), G(golden[f'{tag}/radii'][:, 0]),
           basis.to(dev), 21, 0, 0, 12, 0, 512, o2)
    got = o2.cpu().reshape(n, s, 512)[:2, ::3, :504].numpy()
    np.testing.assert_allclose(got, golden[f'{tag}/l1_ipe_sub'], atol=2e-3)
    np.testing.assert_allclose(got[..., :63], golden[f'{tag}/l1_ipe_sub'][..., :63], atol=2e-5)


def test_level_sample_per_sample_jitter_vs_reference_and_oracle():
  """One jitter draw per sample (Model.single_jitter = False, stepfun.py:203-209): the HIP sampler with jitter_stride = S against the
  reference-executed fixture (tests/golden/ref_persample_jitter.npz) and, bit for bit, against the C oracle's per-sample entry."""
  import os
  from oracle import cstepfun as C, torch_ref as R
  L = _L()
  z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_persample_jitter.npz'))
  n, S0, S1, dilation, anneal = z['meta']
  n, S0, S1 = int(n), int(S0), int(S1)
  near, far = torch.zeros(n, device=dev), torch.ones(n, device=dev)
  ub, mj = R.sample_u_base(S1, True)
  jit = (z['l1_u01'] * np.float32(mj)).astype(np.float32)
  sd, td = torch.empty(n, S1 + 1, device=dev), torch.empty(n, S1 + 1, device=dev)
  idx = torch.empty(n, S1, device=dev, dtype=torch.int32)
  L.call('hugs_level_sample_fwd', n, G(z['l0_sdist']), G(z['l0_weights']), S0, 1, float(dilation), 0.0, 1.0, float(anneal), 0.0, G(ub), G(jit), S1,
         S1, 0, 1, near, far, sd, td, idx, None, None)
  np.testing.assert_allclose(sd.cpu().numpy(), z['l1_sdist'], rtol=0, atol=2e-5)
  osd, otd, oidx = C.level_sample(z['l0_sdist'], z['l0_weights'], True, float(dilation), 0., 1., float(anneal), 0.0, ub, jit, 0,
                                  np.zeros(n, np.float32), np.ones(n, np.float32), sum_order=1)
  assert np.array_equal(sd.cpu().numpy(), osd) and np.array_equal(idx.cpu().numpy(), oidx)


def test_cast_ipe_general_basis_shapes_bf16_and_fp32():
  """Basis / degree combinations off the bf16 fast path (it takes feature counts per half that are multiples of 4: 21 x 12): an
  octahedron basis (3 directions) with 5 degrees = 15 features per half, 32-column rows -- both builds against the oracle, padding zero."""
  from oracle import torch_ref as R
  L = _L()
  N, S, nb, deg = 32, 64, 3, 5
  g = torch.Generator().manual_seed(2)
  basis = torch.tensor(R.generate_basis('octahedron', 1).T.copy(), dtype=torch.float32)
  assert basis.shape == (3, nb)
  o = torch.randn(N, 3, generator=g) * 0.5
  d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
  radii = 5e-4 + 1.5e-3 * torch.rand(N, 1, generator=g)
  td = torch.sort(torch.rand(N, S + 1, generator=g) * 3 + 0.1, -1).values
  means, covs = R.cast_rays(td, o, d, radii)
  lm, lv = R.lift_and_diagonalize(means, covs, basis)
  ref = R.integrated_pos_enc(lm, lv, 0, deg).reshape(N * S, 2 * nb * deg)
  args = (N, S, td.to(dev), o.to(dev), d.to(dev), radii.reshape(-1).to(dev), basis.to(dev), nb, 0, 0, deg)
  out = torch.full((N * S, 32), 7.0, device=dev)
  L.call('hugs_cast_ipe_fwd', *args, 0, 32, out)
  assert float((out.cpu()[:, :30] - ref).abs().max()) < 2e-5 and float(out[:, 30:].abs().max()) == 0
  outb = torch.full((N * S, 32), 7.0, device=dev, dtype=torch.bfloat16)
  L.call('hugs_cast_ipe_fwd', *args, 1, 32, outb)
  assert float((outb.float().cpu()[:, :30] - ref).abs().max()) < 5e-3 and float(outb[:, 30:].float().abs().max()) == 0
  # ... and one ON the fast path with a ragged last workgroup (64 samples per workgroup; 33 rays x 3 samples = 99)
  N2, S2 = 33, 3
  td2 = torch.sort(torch.rand(N2, S2 + 1, generator=g) * 3 + 0.1, -1).values
  o2, d2, r2 = torch.randn(N2, 3, generator=g) * 0.5, torch.nn.functional.normalize(torch.randn(N2, 3, generator=g), dim=-1), torch.full((N2, 1), 1e-3)
  b21 = torch.tensor(R.generate_basis('icosahedron', 2).T.copy(), dtype=torch.float32)
  m2, c2 = R.cast_rays(td2, o2, d2, r2)
  lm2, lv2 = R.lift_and_diagonalize(m2, c2, b21)
  ref2 = R.integrated_pos_enc(lm2, lv2, 0, 12).reshape(N2 * S2, 504)
  ob = torch.full((N2 * S2 + 5, 512), 7.0, device=dev, dtype=torch.bfloat16)      # (+5 guard rows)
  L.call('hugs_cast_ipe_fwd', N2, S2, td2.to(dev), o2.to(dev), d2.to(dev), r2.reshape(-1).to(dev), b21.to(dev), 21, 0, 0, 12, 1, 512, ob)
  assert float((ob[:N2 * S2].float().cpu()[:, :504] - ref2).abs().max()) < 6e-3
  assert float(ob[:N2 * S2, 504:].float().abs().max()) == 0 and float((ob[N2 * S2:].float() - 7.0).abs().max()) == 0


def test_dir_enc_vs_golden(golden):
  L = _L()
  v = G(golden['cfg2_det/viewdirs'])
  out = torch.empty(v.shape[0], 27, device=dev)
  L.call('hugs_dir_enc_fwd', v.shape[0], 4, v, out)
  np.testing.assert_allclose(out.cpu().numpy(), golden['cfg2_det/dir_enc'], atol=2e-6)


@pytest.mark.parametrize('dtype', [0, 1])
def test_gemm_nt_tn_vs_fp64(dtype):
  L = _L()
  tdt = torch.bfloat16 if dtype else torch.float32
  tol = 2e-2 if dtype else 3e-5
  g = torch.Generator(device=dev).manual_seed(0)
  rn = lambda *s: torch.randn(*s, device=dev, generator=g)
  for (M, N, K1, K2, relu, mask, r1, rb) in [(256, 128, 64, 0, 1, False, False, False), (1024, 1024, 1024, 512, 1, False, False, False),
                                               (640, 256, 512, 0, 0, True, True, False), (384, 128, 256, 0, 1, False, False, True),
                                               # K = 128: four K-stages, the shortest pipeline of the 256x256 / 256x128 ring kernels
                                               (512, 256, 128, 0, 1, False, False, False), (768, 512, 64, 64, 0, True, False, False),
                                               (512, 128, 128, 0, 1, False, False, False), (66048, 256, 128, 0, 1, False, False, False)]:
    A1 = rn(M, K1).to(tdt); A2 = rn(M, K2).to(tdt) if K2 else None
    Bt = (rn(N, K1 + K2) / (K1 + K2)**0.5).to(tdt)
    bias = rn(N); rbt = rn(M // 64, N) if rb else None
    mk = rn(M, N).to(tdt) if mask else None
    rr = rn(M) if r1 else None; rc = rn(N) if r1 else None
    out = torch.empty(M, N, device=dev, dtype=tdt)
    L.call('hugs_gemm_nt', dtype, M, N, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, bias, rbt, 64, N, relu, mk, N, rr, rc, out, N)
    A = torch.cat([A1, A2], 1) if K2 else A1
    ref = A.double() @ Bt.double().T + bias.double()
    if rb: ref += rbt.double().repeat_interleave(64, 0)
    if r1: ref += rr.double()[:, None] * rc.double()[None]
    if relu: ref = ref.clamp(min=0)
    if mask: ref = ref * (mk.double() > 0)
    assert float((out.double() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
  for (Mr, Kc, N, ns) in [(1024, 128, 128, 1), (4096, 512, 256, 4), (8192, 1024, 1024, 8),
                          (16384, 256, 256, 8)]:      # one 256x256 tile, 2048 rows per split: the ring kernel's single-tile form
    X = rn(Mr, Kc).to(tdt); Gm = rn(Mr, N).to(tdt)
    dW = torch.empty(Kc, N, device=dev); db = torch.empty(N, device=dev)
    ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(Kc, N, ns) // 4, device=dev)
    L.call('hugs_gemm_tn', dtype, Mr, Kc, N, ns, X, Kc, Gm, N, dW, db, ws)
    ref = X.double().T @ Gm.double()
    assert float((dW.double() - ref).abs().max()) < 2e-6 * Mr**0.5 * 16
    assert float((db.double() - Gm.double().sum(0)).abs().max()) < 1e-3
  with pytest.raises(L.HugsError):
    L.call('hugs_gemm_nt', dtype, 100, 128, 64, 0, A1, 64, None, 0, Bt, 64, None, None, 1, 0, 0, None, 0, None, None, out, 128)


def test_composite_fwd_extras_and_golden(golden):
  from oracle import torch_ref as R
  L = _L()
  for tag, opaque in [('cfg2_jit', 1), ('def3_nobg', 0)]:
    lv = 1
    dens, td, rgb = golden[f'{tag}/l{lv}_density'], golden[f'{tag}/l{lv}_tdist'], golden[f'{tag}/l{lv}_rgb']
    N, S = dens.shape
    w = torch.empty(N, S, device=dev); ro = torch.empty(N, 3, device=dev); ex = torch.empty(N, 5, device=dev)
    L.call('hugs_composite_fwd', N, S, G(dens), G(rgb), G(td), G(golden[f'{tag}/d']), opaque, 1.0, G(golden[f'{tag}/far'][:, 0]), w, ro, ex)
    np.testing.assert_allclose(w.cpu().numpy(), golden[f'{tag}/l{lv}_weights'], rtol=2e-5, atol=3e-7)
    np.testing.assert_allclose(ro.cpu().numpy(), golden[f'{tag}/l{lv}_rend_rgb'], rtol=2e-5, atol=2e-6)
    e = ex.cpu().numpy()
    for i, k in enumerate(['acc', 'distance_mean', 'distance_median', 'distance_percentile_5', 'distance_percentile_95']):
      np.testing.assert_allclose(e[:, i], golden[f'{tag}/l{lv}_rend_{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
  # delta density -> one-hot weights (tests/render_test.py:443-463)
  td = torch.linspace(0, 1, 33, device=dev)[None].contiguous(); dens = torch.zeros(1, 32, device=dev); dens[0, 10] = 1e10
  w = torch.empty(1, 32, device=dev); ro = torch.empty(1, 3, device=dev)
  L.call('hugs_composite_fwd', 1, 32, dens, None, td, torch.tensor([[0., 0., 1.]], device=dev), 0, 1.0, None, w, ro, None)
  assert float(w[0, 10]) == 1 and float(w.sum()) == 1


def test_composite_bwd_finite_and_vs_autograd():
  from oracle import torch_ref as R
  L = _L()
  g = torch.Generator().manual_seed(0)
  # (more than 256 samples per level run 8 / 16 samples per lane: 320, 700, 1024)
  for S, opaque, ls in [(128, 1, 0.), (64, 0, 3.), (32, 1, -10.), (200, 0, 10.), (32, 0, -100.), (320, 1, 1.), (700, 0, 2.), (1024, 1, 0.)]:
    N = 37
    td = torch.sort(torch.rand(N, S + 1, generator=g) * 2 + 0.1, -1).values
    dens = (torch.rand(N, S, generator=g) * np.exp(ls)).double().requires_grad_(True)
    rgb = torch.rand(N, S, 3, generator=g).double().requires_grad_(True)
    d = torch.randn(N, 3, generator=g)
    dro = torch.randn(N, 3, generator=g); dwe = torch.randn(N, S, generator=g)
    w, _, _ = R.compute_alpha_weights(dens, td.double(), d.double(), bool(opaque))
    rend = R.volumetric_rendering(rgb, w, td.double(), 1.0, None, False)
    (gd, gr) = torch.autograd.grad((rend['rgb'] * dro.double()).sum() + (w * dwe.double()).sum(), [dens, rgb])
    dd = torch.empty(N, S, device=dev); dr = torch.empty(N, S, 3, device=dev)
    L.call('hugs_composite_bwd', N, S, dens.detach().float().to(dev), rgb.detach().float().to(dev), td.to(dev), d.to(dev), opaque, 1.0,
           dro.to(dev), dwe.to(dev), dd, dr)
    assert torch.isfinite(dd).all() and torch.isfinite(dr).all()            # tests/render_test.py:408-441
    sc = max(float(gd.abs().max()), 1e-30)
    assert float((dd.cpu().double() - gd).abs().max()) < 2e-4 * sc, (S, opaque, ls)
    assert float((dr.cpu().double() - gr).abs().max()) < 1e-5


def test_losses_vs_oracle_autograd(golden):
  from oracle import torch_ref as R
  L = _L()
  tag = 'def3_nobg'
  c, w = golden[f'{tag}/l2_sdist'], golden[f'{tag}/l2_weights']
  N, S = w.shape
  for lv in range(2):
    cp = golden[f'{tag}/l{lv}_sdist']; wp = torch.tensor(golden[f'{tag}/l{lv}_weights']).double().requires_grad_(True)
    loss = R.lossfun_outer(torch.tensor(c).double(), torch.tensor(w).double(), torch.tensor(cp).double(), wp)
    g, = torch.autograd.grad(loss.mean(), wp)
    lr = torch.empty(N, device=dev); dwe = torch.empty(N, cp.shape[1] - 1, device=dev)
    L.call('hugs_interlevel', N, S, cp.shape[1] - 1, G(c), G(w), G(cp), G(golden[f'{tag}/l{lv}_weights']), 1.0 / (N * S), lr, dwe)
    np.testing.assert_allclose(lr.cpu().numpy(), golden[f'{tag}/l{lv}_lossfun_outer'].sum(-1), rtol=2e-3, atol=1e-7)
    assert float((dwe.cpu().double() - g).abs().max()) < 1e-4 * float(g.abs().max()) + 1e-9
  wt = torch.tensor(w).double().requires_grad_(True)
  ld = R.lossfun_distortion(torch.tensor(c).double(), wt)
  g, = torch.autograd.grad(ld.mean(), wt)
  lr = torch.empty(N, device=dev); dw = torch.empty(N, S, device=dev)
  L.call('hugs_distortion', N, S, G(c), G(w), 1.0 / N, lr, dw)
  np.testing.assert_allclose(lr.cpu().numpy(), golden[f'{tag}/lossfun_distortion'], rtol=2e-5)
  assert float((dw.cpu().double() - g).abs().max()) < 1e-5 * float(g.abs().max())


def test_more_than_256_samples_per_level():
  """The per-ray kernels of the Mip-NeRF 360 path beyond 256 samples per level (8 / 16 samples per lane: capacity 512 /
  1024): compositing forward incl. extras, interlevel and distortion losses with gradients, against the float64 oracle."""
  from oracle import torch_ref as R
  L = _L()
  g = torch.Generator().manual_seed(3)
  N = 19
  for S, Sp in ((384, 300), (1000, 512), (128, 700)):
    td = torch.sort(torch.rand(N, S + 1, generator=g) * 2 + 0.1, -1).values
    dens = torch.rand(N, S, generator=g) * 20
    rgb = torch.rand(N, S, 3, generator=g)
    d = torch.randn(N, 3, generator=g)
    far = td[:, -1] + 1.0
    w = torch.empty(N, S, device=dev); ro = torch.empty(N, 3, device=dev); ex = torch.empty(N, 5, device=dev)
    L.call('hugs_composite_fwd', N, S, dens.to(dev), rgb.to(dev), td.to(dev), d.to(dev), 0, 1.0, far.to(dev), w, ro, ex)
    wo, _, _ = R.compute_alpha_weights(dens.double(), td.double(), d.double(), False)
    rend = R.volumetric_rendering(rgb.double(), wo, td.double(), 1.0, far.double()[:, None], True)
    np.testing.assert_allclose(w.cpu().numpy(), wo.numpy(), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(ro.cpu().numpy(), rend['rgb'].numpy(), rtol=0, atol=2e-5)
    for i, k in enumerate(['acc', 'distance_mean', 'distance_median', 'distance_percentile_5', 'distance_percentile_95']):
      np.testing.assert_allclose(ex.cpu().numpy()[:, i], rend[k].numpy().reshape(-1), rtol=2e-4, atol=2e-5, err_msg=f'{S} {k}')
    # losses: final level (S) against a proposal level (Sp)
    c = torch.sort(torch.rand(N, S + 1, generator=g), -1).values; c[:, 0], c[:, -1] = 0., 1.
    wf = torch.rand(N, S, generator=g); wf = wf / wf.sum(-1, keepdim=True)
    cp = torch.sort(torch.rand(N, Sp + 1, generator=g), -1).values; cp[:, 0], cp[:, -1] = 0., 1.
    wp = (torch.rand(N, Sp, generator=g) * (1.5 / Sp)).double().requires_grad_(True)
    lo = R.lossfun_outer(c.double(), wf.double(), cp.double(), wp)
    gp, = torch.autograd.grad(lo.mean(), wp)
    lr = torch.empty(N, device=dev); dwe = torch.empty(N, Sp, device=dev)
    L.call('hugs_interlevel', N, S, Sp, c.to(dev), wf.to(dev), cp.to(dev), wp.detach().float().to(dev), 1.0 / (N * S), lr, dwe)
    np.testing.assert_allclose(lr.cpu().numpy(), lo.detach().sum(-1).numpy(), rtol=5e-3, atol=1e-7)
    assert float((dwe.cpu().double() - gp).abs().max()) < 2e-3 * float(gp.abs().max()) + 1e-9
    wt = wf.double().requires_grad_(True)
    ld = R.lossfun_distortion(c.double(), wt)
    gd, = torch.autograd.grad(ld.mean(), wt)
    dw = torch.empty(N, S, device=dev)
    L.call('hugs_distortion', N, S, c.to(dev), wf.to(dev), 1.0 / N, lr, dw)
    np.testing.assert_allclose(lr.cpu().numpy(), ld.detach().numpy(), rtol=5e-5)
    assert float((dw.cpu().double() - gd).abs().max()) < 2e-5 * float(gd.abs().max())
  with pytest.raises(L.HugsError):
    L.call('hugs_composite_fwd', 1, 1025, dens.to(dev), None, td.to(dev), d.to(dev), 0, 1.0, None, w, ro, None)


def test_properties_at_baseline_batch():
  """BASELINE size (1024 rays, 64 -> 128): sortedness, domain, weights sum to 1 with an opaque background,
  idempotent determinism (same inputs -> same bits)."""
  from nerf_hugs_amd.internal import stepfun
  L = _L()
  N = 1024
  g = torch.Generator(device=dev).manual_seed(0)
  near = torch.full((N,), 0.1, device=dev); far = torch.full((N,), 1.2, device=dev)
  t0 = torch.tensor([[0., 1.]], device=dev).repeat(N, 1); w0 = torch.ones(N, 1, device=dev)
  u = torch.rand(N, generator=g, device=dev)
  sd0, td0 = stepfun.level_sample(t0, w0, False, 0., (0., 1.), 0.9, 0., 64, u, None, near, far)
  dens = torch.rand(N, 64, generator=g, device=dev) * 30
  d = torch.randn(N, 3, generator=g, device=dev)
  w = torch.empty(N, 64, device=dev); ro = torch.empty(N, 3, device=dev)
  L.call('hugs_composite_fwd', N, 64, dens, None, td0, d, 1, 1.0, None, w, ro, None)
  assert float((w.sum(-1) - 1).abs().max()) < 1e-5
  a = stepfun.level_sample(sd0, w, True, 0.0103125, (0., 1.), 0.9, 0., 128, u, None, near, far)
  b = stepfun.level_sample(sd0, w, True, 0.0103125, (0., 1.), 0.9, 0., 128, u, None, near, far)
  assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
  s = a[0]
  assert bool((s[:, 1:] >= s[:, :-1]).all()) and float(s.min()) >= 0 and float(s.max()) <= 1
  assert bool((a[1][:, 1:] >= a[1][:, :-1]).all()) and float(a[1].min()) >= 0.1 - 1e-6 and float(a[1].max()) <= 1.2 + 1e-6


def test_torch_library_custom_ops():
  """`torch.ops.hugs.*` (nerf_hugs_amd/ops.py): the torch.library registration of the hot-path kernels gives the same
  numbers as the direct C-ABI calls and refuses CPU tensors (no CPU fallback)."""
  import nerf_hugs_amd.ops as O
  from nerf_hugs_amd import _lib as L
  for name in O.OPS:
    assert hasattr(torch.ops.hugs, name), name
  g = torch.Generator(device=dev).manual_seed(0)
  M, K, N = 512, 256, 256
  x = torch.randn(M, K, device=dev, generator=g).bfloat16()
  wt = (torch.randn(N, K, device=dev, generator=g) / 16).bfloat16()
  bias = torch.randn(N, device=dev, generator=g)
  out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  torch.ops.hugs.gemm_nt(x, wt, bias, True, out)
  ref = torch.relu(x.float() @ wt.float().t() + bias)
  assert float((out.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
  gm = torch.randn(M, N, device=dev, generator=g).bfloat16()
  dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
  torch.ops.hugs.gemm_nt_masked(gm, wt.t().contiguous(), x, dx)      # wn [K,N] = wt^T
  refx = (gm.float() @ wt.float()) * (x.float() > 0)
  assert float((dx.float() - refx).abs().max()) < 2e-2 * float(refx.abs().max())
  dw, db = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
  ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(K, N, 2) // 4, device=dev)
  torch.ops.hugs.gemm_tn(x, gm, 2, dw, db, ws)
  refw = x.float().t() @ gm.float()
  assert float((dw - refw).abs().max()) < 1e-3 * float(refw.abs().max())
  np.testing.assert_allclose(db.cpu().numpy(), gm.float().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)
  with pytest.raises(Exception):
    torch.ops.hugs.gemm_nt(x.cpu(), wt.cpu(), bias.cpu(), True, out.cpu())


@pytest.mark.parametrize('shape', [(256, 256, 128, 0), (512, 256, 192, 0), (768, 512, 320, 0), (256, 768, 1024, 0), (1280, 256, 128, 512),
                                   (66560, 256, 192, 0), (67072, 512, 256, 64), (66560, 256, 128, 0), (33280, 1024, 128, 0), (66560, 256, 64, 64)])
def test_gemm_ring_kernels_over_stage_counts(shape):
  """The 256x256 ring kernels (one tile per workgroup, and the persistent form once there are more tiles than CUs) over
  stage counts 4, 6, 10, 32, 20 and two-segment K: bias + relu forward against float64, and the masked dX form."""
  L = _L()
  M, N, K1, K2 = shape
  g = torch.Generator(device=dev).manual_seed(M + N + K1)
  rn = lambda *s: torch.randn(*s, device=dev, generator=g)
  A1 = rn(M, K1).bfloat16(); A2 = rn(M, K2).bfloat16() if K2 else None
  Bt = (rn(N, K1 + K2) / (K1 + K2)**0.5).bfloat16()
  bias = rn(N)
  out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  L.call('hugs_gemm_nt', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, bias, None, 1, 0, 1, None, 0, None, None, out, N)
  A = torch.cat([A1, A2], 1) if K2 else A1
  ref = (A.double() @ Bt.double().T + bias.double()).clamp(min=0)
  assert float((out.double() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))
  mk = rn(M, N).bfloat16()
  L.call('hugs_gemm_nt', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, None, None, 1, 0, 0, mk, N, None, None, out, N)
  ref = (A.double() @ Bt.double().T) * (mk.double() > 0)
  assert float((out.double() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))
  if (K1 + K2) >= 128 and (K1 + K2) % 64 == 0 and M % 256 == 0 and N % 256 == 0:
    # the 1-bit mask forms (forward writes the bits, the dX form reads them, with the rank-1 term of the last trunk layer)
    bits = torch.empty(M * N // 32, dtype=torch.int32, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    L.call('hugs_gemm_nt_bits', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, bias, 1, None, None, y, N, bits, None)
    r1r, r1c = rn(M), rn(N)
    L.call('hugs_gemm_nt_bits', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, None, 0, r1r, r1c, out, N, None, bits)
    ref = (A.double() @ Bt.double().T + r1r.double()[:, None] * r1c.double()[None]) * (y.double() > 0)
    assert float((out.double() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_gemm_tn_batch_vs_matmul(dtype):
  """hugs_gemm_tn_batch (round 4: the weight gradients of several layers in one launch, reduction rows cut into #CUs / tiles
  uneven pieces): every item against the float64 product of the same 16-bit operands, bias gradients included, for piece
  counts that do not divide the row units, and bit-identical between two runs (fixed partition, fixed summation order)."""
  import ctypes
  from nerf_hugs_amd import _lib
  from nerf_hugs_amd.internal import engine as E
  td = torch.bfloat16 if dtype == 'bf16' else torch.float16
  dt = 1 if dtype == 'bf16' else 2
  g = torch.Generator(device='cuda').manual_seed(5)
  shapes = [(4096, 256, 256, True), (4096, 512, 256, False), (8192 + 64 * 3, 256, 512, True), (4096, 1024, 256, True)]
  Xs, Gs, dWs, dbs = [], [], [], []
  for (M, Kc, N, hb) in shapes:
    ldx, ldg = Kc + 64, N            # a leading dimension wider than the panel (the skip layer reads a window of [W | Fp])
    Xs.append((torch.randn(M, ldx, generator=g, device='cuda') * 0.5).to(td))
    Gs.append((torch.randn(M, ldg, generator=g, device='cuda') * 0.5).to(td))
    dWs.append(torch.empty(Kc, N, device='cuda'))
    dbs.append(torch.empty(N, device='cuda') if hb else None)
  arr = np.zeros(len(shapes), E._TN_ITEM)
  for k, (M, Kc, N, hb) in enumerate(shapes):
    arr[k] = (Xs[k].data_ptr(), Gs[k].data_ptr(), dWs[k].data_ptr(), dbs[k].data_ptr() if hb else 0, Xs[k].shape[1], Gs[k].shape[1], M, Kc, N, 0)
  lib = _lib.lib().cdll
  auto = int(lib.hugs_gemm_tn_batch_nsplit(len(shapes), ctypes.c_void_p(arr.ctypes.data)))
  assert auto == 256 // (1 + 2 + 2 + 4) or auto >= 1
  for ns in (auto, 7, 1, 8):
    ws = torch.empty(int(lib.hugs_gemm_tn_batch_ws_bytes(len(shapes), arr.ctypes.data, ns)) // 4, device='cuda')
    outs = []
    for rep in range(2):
      for t in dWs + [d for d in dbs if d is not None]:
        t.fill_(float('nan'))
      _lib.call('hugs_gemm_tn_batch', dt, len(shapes), arr.ctypes.data, ns, ws)
      torch.cuda.synchronize()
      outs.append([t.clone() for t in dWs] + [d.clone() for d in dbs if d is not None])
    for a, b in zip(*outs):
      assert torch.equal(a, b), 'two runs differ'
    for k, (M, Kc, N, hb) in enumerate(shapes):
      ref = Xs[k][:, :Kc].double().T @ Gs[k].double()
      err = float((dWs[k].double() - ref).abs().max() / ref.abs().max())
      assert err < 2e-5, (ns, k, err)
      if hb:
        rb = Gs[k].double().sum(0)
        assert float((dbs[k].double() - rb).abs().max() / rb.abs().max()) < 2e-5, (ns, k)
  with pytest.raises(_lib.HugsError):      # fewer than 512 rows per piece
    _lib.call('hugs_gemm_tn_batch', dt, len(shapes), arr.ctypes.data, 9, ws)


@pytest.mark.parametrize('M,nl', [(256, 1), (1280, 3), (66560, 3), (512, 7)])
def test_mlp256_tail_fused_forward_vs_layer_by_layer(M, nl):
  """hugs_mlp256_tail_fwd (round 4: layers 1.. of a 256-wide trunk + the density head in one launch, activations LDS-resident)
  against the float64 chain on the same bf16 operands (every layer's output re-rounded to bf16 as the stored activation is),
  the stored Y_l, the density head, and the 1-bit masks -- consumed by the dX form of hugs_gemm_nt_bits exactly as the
  backward pass consumes them."""
  L = _L()
  g = torch.Generator(device=dev).manual_seed(M + nl)
  rn = lambda *s: torch.randn(*s, device=dev, generator=g)
  Y0 = rn(M, 256).clamp_(min=0).bfloat16()
  Wt = [(rn(256, 256) * (2.0 / 256)**0.5).bfloat16() for _ in range(nl)]      # [out, in]
  bias = [rn(256) * 0.1 for _ in range(nl)]
  wd, bd = rn(256) * 0.1, rn(1)
  Y = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(nl)]
  bits = [torch.zeros(M * 256 // 32, dtype=torch.int32, device=dev) for _ in range(nl)]
  raw, dens = torch.empty(M, device=dev), torch.empty(M, device=dev)
  ptrs = lambda ts: np.ascontiguousarray([t.data_ptr() for t in ts], np.uint64)
  a_w, a_b, a_y, a_bits = ptrs(Wt), ptrs(bias), ptrs(Y), ptrs(bits)
  L.call('hugs_mlp256_tail_fwd', 1, M, nl, Y0, a_w.ctypes.data, a_b.ctypes.data, a_y.ctypes.data, a_bits.ctypes.data, wd, bd, -1.0, raw, dens)
  torch.cuda.synchronize()
  x = Y0.double()
  for l in range(nl):
    ref = (x @ Wt[l].double().T + bias[l].double()).clamp(min=0)
    got = Y[l].double()
    assert float((got - ref).abs().max()) < 1e-2 * max(1.0, float(ref.abs().max())), l      # bf16 rounding of the output
    # the masks: dX = (G W) * (Y_l > 0) through the bit-mask GEMM against the same product masked by the stored activation
    G = rn(M, 256).bfloat16()
    out = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
    L.call('hugs_gemm_nt_bits', 1, M, 256, 256, 0, G, 256, None, 0, Wt[l], 256, None, 0, None, None, out, 256, None, bits[l])
    refm = (G.double() @ Wt[l].double().T) * (Y[l].double() > 0)
    assert float((out.double() - refm).abs().max()) < 2e-2 * max(1.0, float(refm.abs().max())), ('mask bits', l)
    x = Y[l].double()      # the next layer reads the bf16-rounded activation
  r = Y[-1].double() @ wd.double() + bd.double()
  np.testing.assert_allclose(raw.cpu().numpy(), r.cpu().numpy(), rtol=2e-5, atol=2e-5)
  np.testing.assert_allclose(dens.cpu().numpy(), torch.nn.functional.softplus(r - 1.0).cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('M', [256, 1280, 66560])
def test_mlp256_tail_fused_backward_vs_layer_by_layer(M):
  """hugs_mlp256_tail_bwd (round 5: the PropMLP's dX chain -- rank-1 head gradient + three masked dX products -- in one launch with
  register-resident weights) against (i) the float64 chain on the same bf16 operands with every G_l re-rounded to bf16 as the
  stored gradient is and (ii) the launches it replaces (hugs_rank1_mask + hugs_gemm_nt_bits x 3), with the masks the fused FORWARD
  kernel wrote."""
  L = _L()
  g = torch.Generator(device=dev).manual_seed(M + 7)
  rn = lambda *s: torch.randn(*s, device=dev, generator=g)
  Y0f = rn(M, 256)
  Y0 = Y0f.clamp(min=0).bfloat16()
  bits0 = torch.zeros(M * 256 // 32, dtype=torch.int32, device=dev)
  # Y0 and its mask bits the way the step makes them: a relu GEMM that writes bits (identity-like product of a random input)
  X = rn(M, 256).bfloat16()
  W0 = (rn(256, 256) * (2.0 / 256)**0.5).bfloat16()
  L.call('hugs_gemm_nt_bits', 1, M, 256, 256, 0, X, 256, None, 0, W0, 256, torch.zeros(256, device=dev), 1, None, None, Y0, 256, bits0, None)
  Wt = [(rn(256, 256) * (2.0 / 256)**0.5).bfloat16() for _ in range(3)]      # [out, in]
  Wn = [w.t().contiguous() for w in Wt]                                       # [in, out]: the dX operand copies
  bias = [rn(256) * 0.1 for _ in range(3)]
  wd, bd = rn(256) * 0.1, rn(1)
  Y = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(3)]
  bits = [torch.zeros(M * 256 // 32, dtype=torch.int32, device=dev) for _ in range(3)]
  raw, dens = torch.empty(M, device=dev), torch.empty(M, device=dev)
  ptrs = lambda ts: np.ascontiguousarray([t.data_ptr() for t in ts], np.uint64)
  a_w, a_b, a_y, a_bits = ptrs(Wt), ptrs(bias), ptrs(Y), ptrs(bits)
  L.call('hugs_mlp256_tail_fwd', 1, M, 3, Y0, a_w.ctypes.data, a_b.ctypes.data, a_y.ctypes.data, a_bits.ctypes.data, wd, bd, -1.0, raw, dens)
  d_raw = rn(M) * 0.3
  G = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(4)]
  a_wn, a_allbits, a_g = ptrs(Wn), ptrs([bits0] + bits), ptrs(G)
  L.call('hugs_mlp256_tail_bwd', 1, M, 3, d_raw, wd, a_wn.ctypes.data, a_allbits.ctypes.data, a_g.ctypes.data)
  torch.cuda.synchronize()
  acts = [Y0] + Y
  # (i) float64 chain
  ref = (d_raw.double()[:, None] * wd.double()[None]).float().bfloat16().double() * (acts[3].double() > 0)
  assert float((G[3].double() - ref).abs().max()) <= 1e-2 * max(1e-6, float(ref.abs().max()))
  for l in (3, 2, 1):
    x = G[l].double()      # the next product reads the bf16-rounded gradient
    ref = (x @ Wt[l - 1].double()) * (acts[l - 1].double() > 0)
    assert float((G[l - 1].double() - ref).abs().max()) <= 1e-2 * max(1e-6, float(ref.abs().max())), l
  # (ii) the launches it replaces
  Gu = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(4)]
  L.call('hugs_rank1_mask', 1, M, 256, d_raw, wd, acts[3], 256, Gu[3], 256)
  allbits = [bits0] + bits
  for l in (3, 2, 1):
    L.call('hugs_gemm_nt_bits', 1, M, 256, 256, 0, Gu[l], 256, None, 0, Wn[l - 1], 256, None, 0, None, None, Gu[l - 1], 256, None, allbits[l - 1])
  torch.cuda.synchronize()
  assert torch.equal(G[3], Gu[3])
  for l in (2, 1, 0):      # (different accumulation order: equal to a bf16 ulp of the largest entry)
    assert float((G[l].double() - Gu[l].double()).abs().max()) <= 8e-3 * float(Gu[l].double().abs().max()), l
  assert float(G[0].double().abs().max()) > 0

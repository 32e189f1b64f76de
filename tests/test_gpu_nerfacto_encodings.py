"""GPU parity of the nerfacto encodings (hash grid, SH-4) through the C ABI against oracle/hashgrid_ref.py.
PARITY UNPINNED with respect to tiny-cuda-nn itself (not importable); the oracle restates the published algorithm and
is cross-checked by properties here (partition of unity, grid-vertex interpolation, SH orthonormality)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grid(**kw):
  from nerf_hugs_amd.nerfacto import encodings as E
  return E.HashGrid(**kw)


@pytest.mark.parametrize('kw', [dict(n_levels=16, features_per_level=2, log2_hashmap_size=19, base_resolution=16, max_resolution=2048),
                                dict(n_levels=5, features_per_level=2, log2_hashmap_size=17, base_resolution=16, max_resolution=128),
                                dict(n_levels=8, features_per_level=4, log2_hashmap_size=14, base_resolution=16, max_resolution=1024)])
def test_hashgrid_forward_backward_vs_oracle(kw):
  from oracle import hashgrid_ref as H
  g = _grid(**kw)
  F = kw['features_per_level']
  offs, ress, scales = H.level_table(kw['n_levels'], kw['base_resolution'],
                                     np.exp((np.log(kw['max_resolution']) - np.log(kw['base_resolution'])) / (kw['n_levels'] - 1)),
                                     kw['log2_hashmap_size'])
  assert np.array_equal(offs, g.offsets) and np.array_equal(ress, g.resolutions) and np.array_equal(scales, g.scales)
  rng = np.random.default_rng(0)
  n = 3000
  x = rng.uniform(size=(n, 3)).astype(np.float32)
  x[:4] = [[0, 0, 0], [1, 1, 1], [0, 1, 0.5], [0.999999, 0, 1]]            # domain corners / faces
  g.table.copy_(torch.from_numpy(rng.normal(size=(g.n_entries, F)).astype(np.float32)))
  table = g.table.cpu().numpy()
  want = H.hashgrid_forward(x, table, offs, ress, scales, F)
  xt = torch.from_numpy(x).cuda()
  got = g.forward(xt, dtype=torch.float32).cpu().numpy()
  assert np.abs(got - want).max() < 2e-5 * max(1, np.abs(want).max())
  got16 = g.forward(xt, dtype=torch.bfloat16).float().cpu().numpy()
  assert np.abs(got16 - want).max() < 1e-2 * max(1, np.abs(want).max())
  # backward: d table = scatter of trilinear weights * d_out
  d_out = rng.normal(size=want.shape).astype(np.float32)
  gw = H.hashgrid_backward(x, d_out, g.n_entries, offs, ress, scales, F)
  d_table = torch.zeros_like(g.table)
  g.backward(xt, torch.from_numpy(d_out).cuda(), d_table)
  err = np.abs(d_table.cpu().numpy() - gw).max()
  assert err < 1e-4 * max(1, np.abs(gw).max()), err
  # <forward(table), d_out> is linear in the table: its gradient is exactly the backward
  t2 = rng.normal(size=table.shape).astype(np.float32)
  lhs = float((H.hashgrid_forward(x, t2, offs, ress, scales, F) * d_out).sum())
  assert abs(lhs - float((gw * t2).sum())) < 1e-6 * abs(lhs) + 1e-6


def test_hashgrid_properties():
  g = _grid(n_levels=6, features_per_level=2, log2_hashmap_size=15, base_resolution=8, max_resolution=256)
  rng = np.random.default_rng(1)
  x = torch.from_numpy(rng.uniform(size=(2000, 3)).astype(np.float32)).cuda()
  # partition of unity: a constant table interpolates to that constant on every level
  g.table.fill_(0.75)
  out = g.forward(x, dtype=torch.float32)
  assert float((out - 0.75).abs().max()) < 1e-6
  # at a vertex of a DENSE level the lookup returns that vertex's entry exactly
  res0, sc0 = int(g.resolutions[0]), float(g.scales[0])
  g.table.copy_(torch.randn(g.table.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0)))
  v = torch.tensor([[2, 3, 1], [0, 0, 0], [res0 - 2, 1, 4]], dtype=torch.float32, device='cuda')
  xv = (v - 0.5 + 1e-4) / sc0               # pos = x*scale + 0.5 -> cell = v, weight ~ 1e-4
  out = g.forward(xv, dtype=torch.float32)[:, :2]
  idx = (v[:, 0] + v[:, 1] * res0 + v[:, 2] * res0 * res0).long()
  assert float((out - g.table[idx]).abs().max()) < 2e-3
  # errors
  from nerf_hugs_amd.nerfacto import encodings as E
  with pytest.raises(ValueError):
    E.HashGrid(features_per_level=3)


def test_sh4_vs_oracle_and_orthonormal():
  from nerf_hugs_amd.nerfacto import encodings as E
  from oracle import hashgrid_ref as H
  rng = np.random.default_rng(2)
  v = rng.normal(size=(200000, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
  d01 = ((v + 1) / 2).astype(np.float32)
  got = E.spherical_harmonics4(torch.from_numpy(d01).cuda(), dtype=torch.float32).cpu().numpy().astype(np.float64)
  assert np.abs(got - H.sh4(d01)).max() < 2e-6
  gram = got.T @ got / len(v) * 4 * np.pi                                   # Monte-Carlo inner products on the sphere
  assert np.abs(gram - np.eye(16)).max() < 2e-2
  out = torch.zeros(5, 32, device='cuda', dtype=torch.bfloat16)
  E.spherical_harmonics4(torch.from_numpy(d01[:5]).cuda(), out=out, col0=16)
  assert float(out[:, :16].abs().max()) == 0 and float((out[:, 16:].float().cpu() - torch.from_numpy(got[:5]).float()).abs().max()) < 1e-2


@pytest.mark.parametrize('F', [2, 4])
def test_hashgrid_backward_many_ray_ordered_samples(F):
  """>= 65536 samples in ray order (runs of samples inside one coarse cell): the size at which level 0 is accumulated in LDS
  by persistent workgroups and the other levels use the quad-cooperative atomics; against the numpy oracle."""
  from oracle import hashgrid_ref as H
  kw = dict(n_levels=4, features_per_level=F, log2_hashmap_size=12, base_resolution=16, max_resolution=96)
  g = _grid(**kw)
  offs, ress, scales = H.level_table(4, 16, np.exp((np.log(96) - np.log(16)) / 3), 12)
  rng = np.random.default_rng(3)
  nr, S = 560, 128                                  # 71680 samples
  o = rng.uniform(0.3, 0.7, (nr, 1, 3)); d = rng.normal(size=(nr, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  x = np.clip(o + d * np.linspace(0, 0.5, S)[None, :, None], 0, 1).reshape(-1, 3).astype(np.float32)
  d_out = rng.normal(size=(x.shape[0], 4 * F)).astype(np.float32)
  gw = H.hashgrid_backward(x, d_out, g.n_entries, offs, ress, scales, F)
  d_table = torch.zeros_like(g.table)
  g.backward(torch.from_numpy(x).cuda(), torch.from_numpy(d_out).cuda(), d_table)
  err = np.abs(d_table.cpu().numpy() - gw).max()
  assert err < 3e-4 * max(1, np.abs(gw).max()), err
  # level 0 alone (the LDS path) and the rest separately
  n0 = int(offs[1])
  assert np.abs(d_table.cpu().numpy()[:n0] - gw[:n0]).max() < 3e-4 * np.abs(gw[:n0]).max()


def test_hashgrid_table_gradient_binned_equals_the_atomic_scatter():
  """hugs_hashgrid_bwd_ws (round 6, csrc/hugs_hashgrid_binned.inc): the table gradient as a segmented reduction by table slot -- bins
  of 16384 entries summed in LDS by one workgroup each -- gives the atomic scatter's gradient up to fp32 summation order.  (Measured not
  faster, so the model calls it only under HUGS_HG_BINNED=1; the entry point, its fall-backs and its arithmetic stay under test.)
  131 072 samples on a 6-level grid with dense and hashed levels, a third of the rows without gradient, fp32 and 16-bit gradient rows."""
  from nerf_hugs_amd import _lib as L
  dev = 'cuda'
  g = torch.Generator(device=dev).manual_seed(3)
  n, F = 131072, 2
  res = [16, 24, 40, 64, 128, 256]
  ent = [min((r + 1) ** 3, 1 << 17) for r in res]
  ent = [(e + 7) // 8 * 8 for e in ent]
  off = np.concatenate([[0], np.cumsum(ent)]).astype(np.int64)
  resa = np.asarray(res, np.int32); sc = np.asarray([float(r) for r in res], np.float32)
  x = torch.rand(n, 3, generator=g, device=dev)
  for dt, tdt in ((0, torch.float32), (1, torch.bfloat16), (2, torch.float16)):
    dX = torch.randn(n, len(res) * F, generator=g, device=dev)
    dX[torch.rand(n, generator=g, device=dev) < 0.33] = 0
    dX = dX.to(tdt).contiguous()
    ta = torch.zeros(int(off[-1]) * F, device=dev); tb = torch.zeros_like(ta)
    L.call('hugs_hashgrid_bwd', n, len(res), F, off.ctypes.data, resa.ctypes.data, sc.ctypes.data, x, dX, dt, dX.stride(0), ta)
    nbytes = int(L.lib().cdll.hugs_hashgrid_bwd_ws_bytes(n, len(res), F))
    assert nbytes == 3 * 1024 * 4 + 12 * 8 * n * len(res)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    L.call('hugs_hashgrid_bwd_ws', n, len(res), F, off.ctypes.data, resa.ctypes.data, sc.ctypes.data, x, dX, dt, dX.stride(0), tb, ws, nbytes)
    torch.cuda.synchronize()
    assert int((ta != 0).sum()) > 100000
    assert torch.equal(ta != 0, tb != 0)
    assert float((ta.double() - tb.double()).abs().max()) <= 2e-5 * float(ta.abs().max())
    # a workspace that is too small falls back to the scatter (same numbers, nothing written out of bounds)
    tc = torch.zeros_like(ta)
    L.call('hugs_hashgrid_bwd_ws', n, len(res), F, off.ctypes.data, resa.ctypes.data, sc.ctypes.data, x, dX, dt, dX.stride(0), tc, ws, 4096)
    assert float((ta.double() - tc.double()).abs().max()) <= 2e-5 * float(ta.abs().max())

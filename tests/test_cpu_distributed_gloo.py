"""world_size-2 and -4 gloo tests (CPU) of the data-parallel host logic: batch sharding by whole patches, the
gradient/stat all-reduce mean, and render_image's chunk split + pad + gather across ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _fake_rays(h, w):
  from nerf_hugs_amd.internal import utils
  g = torch.Generator().manual_seed(0)
  f = lambda c: torch.rand(h, w, c, generator=g)
  return utils.Rays(pix_coords=f(2), origins=f(3), directions=f(3), viewdirs=f(3), radii=f(1), lossmult=f(1),
                    static_mask=f(1), near=f(1), far=f(1), embed_idx=torch.zeros(h, w, 1, dtype=torch.int32),
                    cam_idx=torch.zeros(h, w, 1, dtype=torch.int32))


def _fake_render_fn(rng, rays):
  # [ndev=1, n, c] in -> per-level dicts with the leading device axis of the all-gathered result
  from nerf_hugs_amd.internal import parallel
  o = rays.origins[0]
  rgb = parallel.all_gather_cat(o * 2 + 1)[None]
  acc = parallel.all_gather_cat(o.sum(-1))[None]
  lvl = {'rgb': rgb, 'acc': acc, 'ray_sdist': torch.zeros(1, 4, 3)}
  return [dict(lvl), dict(lvl)], None


def _worker(rank, world, port, tmp):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from nerf_hugs_amd.internal import configs, models, parallel, utils
  # 1. gradient mean
  g = torch.full((1000,), float(rank + 1))
  parallel.allreduce_mean_(g)
  assert torch.allclose(g, torch.full((1000,), (world + 1) / 2))
  # 2. batch sharding keeps whole patches
  rays = _fake_rays(4, 4)
  batch = utils.Batch(rays=rays.map(lambda x: x.reshape(4, 2, 2, -1)), rgb=torch.arange(4 * 2 * 2 * 3.).reshape(4, 2, 2, 3))
  sb = parallel.shard_batch(batch, rank, world)
  per = 4 // world
  assert sb.rgb.shape == (per, 2, 2, 3) and torch.equal(sb.rgb, batch.rgb[rank * per:(rank + 1) * per])
  try:      # 3 patches over 2 or 4 devices: train.py:53-56 'Batch size must be divisible by the number of devices.'
    parallel.shard_batch(utils.Batch(rays=batch.rays.map(lambda x: x[:3]), rgb=batch.rgb[:3]), rank, world)
    raise AssertionError('expected ValueError')
  except ValueError as e:
    assert 'divisible' in str(e)
  # 2b. the train step's gradient exchange (train_utils.step_core): asynchronous bucket all-reduces issued while the
  # backward pass runs + ONE all-reduce per range no bucket covered (uncovered_ranges) == one all-reduce of the whole buffer
  from nerf_hugs_amd.internal import train_utils
  from nerf_hugs_amd.internal.engine import MLPSpec, ParamLayout
  lay = ParamLayout([MLPSpec('NerfMLP_0', False, 4, net_depth=3, net_width=128, bottleneck_width=128),
                     MLPSpec('PropMLP_0', True, 0, net_depth=2, net_width=128, disable_rgb=True)], 16, 4)
  total = lay.size + train_utils.STAT_TAIL
  gsrc = torch.Generator().manual_seed(100 + rank)
  grad = torch.randn(total, generator=gsrc)
  ref = grad.clone()
  dist.all_reduce(ref, op=dist.ReduceOp.SUM)
  works, ranges = [], []
  nerf = [lf for lf in lay.leaves if lf['path'][0] == 'NerfMLP_0']
  for lo_lf, hi_lf in ((nerf[6], nerf[-1]), (nerf[4], nerf[5]), (nerf[0], nerf[3])):      # heads first, then trunk layers downwards
    lo, hi = lo_lf['off'], hi_lf['off'] + int(np.prod(hi_lf['pshape']))
    ranges.append((lo, hi))
    works.append(dist.all_reduce(grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
  for lo, hi in train_utils.uncovered_ranges(lay, ranges, total):
    works.append(dist.all_reduce(grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
  for w_ in works:
    w_.wait()
  # (alignment padding between leaves is never written by the step and stays out of the exchange: compare leaves + tail)
  # (and a ring all-reduce over more than two ranks sums a chunk in an order that depends on its position in the buffer: equal
  #  to rounding, bit-equal for two ranks)
  same = torch.equal if world == 2 else (lambda x, y: torch.allclose(x, y, rtol=1e-5, atol=1e-6))
  for lf in lay.leaves:
    a, b = lf['off'], lf['off'] + int(np.prod(lf['pshape']))
    assert same(grad[a:b], ref[a:b]), lf['path']
  assert same(grad[lay.size:], ref[lay.size:])
  # 3. render_image: 7x5 = 35 rays, chunks of 16 -> 16,16,3(+1 pad)
  configs.clear_config()
  cfg = configs.make_config(render_chunk_size=16, vis_num_rays=2)
  rays = _fake_rays(7, 5)
  # the ray_* subsample draws jax's permutation on the GPU; this CPU test of the host logic substitutes the oracle's
  from oracle import threefry_ref as T

  def cpu_permutation(key, n):
    k, x = T.prng_key(0), np.arange(n)
    for _ in range(int(np.ceil(3 * np.log(max(1, n)) / np.log(2 ** 32 - 1)))):
      k, sub = T.split(k)
      x = x[np.argsort(T.random_bits(sub, (n,)), kind='stable')]
    return torch.from_numpy(x)
  models.hrandom.permutation = cpu_permutation
  models.hrandom.PRNGKey = lambda seed, device=None: None
  out = models.render_image(_fake_render_fn, rays, None, cfg, verbose=False)
  assert out['rgb'].shape == (7, 5, 3)
  assert torch.allclose(out['rgb'], rays.origins * 2 + 1)
  assert torch.allclose(out['acc'], rays.origins.sum(-1))
  if rank == 0:
    open(os.path.join(tmp, 'ok'), 'w').write('1')
  dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_gloo_ranks(tmp_path, world):
  port = _free_port()
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  assert (tmp_path / 'ok').exists()

"""Stand-in for the `tinycudann` package -- BUILD-CONTAINER FIXTURE TOOLING, not product code, not the reference.

tiny-cuda-nn is a third-party CUDA dependency of the reference's nerfacto path (requirements_torch.txt, un-versioned
git URL; absent from /root/reference and from this image).  The reference's `nerfacto/models/nerfacto.py` only needs
`tcnn.Encoding` when its configs say `enable_tcnn_mlp: False` (every shipped yml does): a multiresolution hash grid
(3-D for the fields, 2-D for HA-NeRF's ImplicitMask) and a degree-4 spherical-harmonics encoding; all MLPs are then
torch `nn.Linear`.  This module provides `Encoding` on top of oracle/hashgrid_ref.py (the published Instant-NGP
algorithm, PARITY UNPINNED against tiny-cuda-nn itself) so that tests/golden/gen_nerfacto_model_fixtures.py can import
and EXECUTE the reference's own `Model` / `Loss` classes: what gets pinned is the reference's field / model / loss
WIRING around the encodings, not tiny-cuda-nn's arithmetic.  `Network` / `NetworkWithInputEncoding` raise."""
import sys
import types

import numpy as np
import torch

from oracle import hashgrid_ref as HG


class _GridFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, params, x, spec):
    ctx.spec, ctx.x = spec, x.detach().numpy().astype(np.float32)
    table = params.detach().numpy().reshape(spec['n_entries'], spec['F'])
    out = HG.hashgrid_forward(ctx.x, table, spec['offsets'], spec['resolutions'], spec['scales'], spec['F'])
    return torch.from_numpy(out).to(params.dtype)

  @staticmethod
  def backward(ctx, g):
    s = ctx.spec
    gt = HG.hashgrid_backward(ctx.x, g.numpy(), s['n_entries'], s['offsets'], s['resolutions'], s['scales'], s['F'])
    return torch.from_numpy(gt.reshape(-1)).to(g.dtype), None, None


class Encoding(torch.nn.Module):
  """tcnn.Encoding(n_input_dims, encoding_config, dtype): `params` is the flat parameter vector (level-major
  [entries, features] tables), `n_output_dims` as in tiny-cuda-nn."""

  def __init__(self, n_input_dims, encoding_config, dtype=None, seed=1337):
    super().__init__()
    c = dict(encoding_config)
    self.otype = c['otype']
    if self.otype == 'HashGrid':
      offs, ress, scales = HG.level_table(c['n_levels'], c['base_resolution'], c['per_level_scale'], c['log2_hashmap_size'],
                                          dims=n_input_dims)
      self.spec = dict(offsets=offs, resolutions=ress, scales=scales, F=c['n_features_per_level'], n_entries=int(offs[-1]))
      self.n_output_dims = c['n_levels'] * c['n_features_per_level']
      g = torch.Generator().manual_seed(seed)
      self.params = torch.nn.Parameter((torch.rand(self.spec['n_entries'] * self.spec['F'], generator=g) * 2 - 1) * 1e-4)
    elif self.otype == 'SphericalHarmonics':
      if c['degree'] != 4 or n_input_dims != 3:
        raise NotImplementedError('stand-in: SphericalHarmonics degree 4 of 3-D inputs only')
      self.n_output_dims = 16
    else:
      raise NotImplementedError(f'stand-in: encoding {self.otype!r}')
    self.n_input_dims = n_input_dims

  def forward(self, x):
    if self.otype == 'HashGrid':
      return _GridFn.apply(self.params, x, self.spec)
    return torch.from_numpy(HG.sh4(x.detach().numpy())).to(x.dtype)


def _no(*a, **k):
  raise NotImplementedError('stand-in: tcnn.Network / NetworkWithInputEncoding (enable_tcnn_mlp: True) are not provided')


def install():
  m = types.ModuleType('tinycudann')
  m.Encoding, m.Network, m.NetworkWithInputEncoding = Encoding, _no, _no
  sys.modules['tinycudann'] = m
  return m

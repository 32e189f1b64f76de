"""tiny-cuda-nn's `HashGrid` and `SphericalHarmonics` encodings as nerfacto configures them
(nerfacto/models/nerfacto.py:693-700,714-733,761-770,921-947), on csrc/hugs_hashgrid.hip.

PARITY UNPINNED: tiny-cuda-nn is a third-party CUDA package that cannot be imported here; the level layout, the
coherent prime hash and the trilinear lookup follow the published algorithm (oracle/hashgrid_ref.py)."""
import numpy as np
import torch

from .. import _lib as L


class HashGrid:
  """n_levels x features_per_level multiresolution hash encoding of points in [0,1]^3 (dims=2: of image-plane
  coordinates in [0,1]^2, the nerfacto HA-NeRF ImplicitMask's grid -- level table only, the model calls the 2-D kernels).

  `table` is the fp32 parameter tensor [total entries, features] (tiny-cuda-nn initialises U(-1e-4, 1e-4));
  `forward(x01)` returns [N, n_levels*features] in bf16 or fp32, `backward(x01, d_out, d_table)` accumulates
  d loss / d table into `d_table` (the positions get no gradient: nerfacto does not optimise cameras here)."""

  def __init__(self, n_levels=16, features_per_level=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=None,
               max_resolution=2048, device='cuda', seed=0, dims=3):
    if per_level_scale is None:       # nerfacto.py:713 growth_factor
      per_level_scale = float(np.exp((np.log(max_resolution) - np.log(base_resolution)) / (n_levels - 1))) if n_levels > 1 else 1.0
    if features_per_level not in (2, 4):
      raise ValueError('features_per_level must be 2 or 4')
    if dims not in (2, 3):
      raise ValueError('hash grid of 2-D (image plane) or 3-D points')
    self.n_levels, self.features, self.dims = n_levels, features_per_level, dims
    offs, ress, scales, off = [0], [], [], 0
    l2 = np.float32(np.log2(np.float32(per_level_scale)))
    for l in range(n_levels):
      scale = np.float32(np.exp2(np.float32(l) * l2) * np.float32(base_resolution) - np.float32(1.0))
      res = int(np.ceil(scale)) + 1
      n = min(res ** dims, 2 ** 31 - 1)
      n = min((n + 7) // 8 * 8, 1 << log2_hashmap_size)
      off += n
      offs.append(off); ress.append(res); scales.append(scale)
    self.offsets = np.ascontiguousarray(offs, np.int64)
    self.resolutions = np.ascontiguousarray(ress, np.int32)
    self.scales = np.ascontiguousarray(scales, np.float32)
    self.n_entries = int(off)
    self.n_output_dims = n_levels * features_per_level
    g = torch.Generator().manual_seed(seed)
    self.table = ((torch.rand(self.n_entries, features_per_level, generator=g) * 2 - 1) * 1e-4).to(device)

  def _tables(self):
    return self.offsets.ctypes.data, self.resolutions.ctypes.data, self.scales.ctypes.data

  def forward(self, x01, out=None, dtype=torch.bfloat16, table=None):
    x01 = x01.reshape(-1, 3).to(torch.float32).contiguous()
    n = x01.shape[0]
    if out is None:
      out = torch.empty((n, self.n_output_dims), dtype=dtype, device=x01.device)
    o, r, s = self._tables()
    L.call('hugs_hashgrid_fwd', n, self.n_levels, self.features, o, r, s, x01, self.table if table is None else table,
           int(out.dtype == torch.bfloat16), out.stride(0), out)
    return out

  def backward(self, x01, d_out, d_table):
    x01 = x01.reshape(-1, 3).to(torch.float32).contiguous()
    o, r, s = self._tables()
    L.call('hugs_hashgrid_bwd', x01.shape[0], self.n_levels, self.features, o, r, s, x01, d_out,
           int(d_out.dtype == torch.bfloat16), d_out.stride(0), d_table)
    return d_table


def spherical_harmonics4(dirs01, out=None, col0=0, dtype=torch.bfloat16):
  """tcnn SphericalHarmonics(degree=4) of directions mapped to [0,1] ((viewdirs + 1) / 2, nerfacto.py:857)."""
  d = dirs01.reshape(-1, 3).to(torch.float32).contiguous()
  if out is None:
    out = torch.empty((d.shape[0], 16), dtype=dtype, device=d.device)
  L.call('hugs_sh4_fwd', d.shape[0], d, int(out.dtype == torch.bfloat16), out.stride(0), col0, out)
  return out

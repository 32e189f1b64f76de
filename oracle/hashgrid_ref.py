"""CPU oracle for the nerfacto encodings.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

The reference's nerfacto path (nerfacto/models/nerfacto.py:693-733,761-770,921-947) gets its multiresolution hash
grid and its spherical-harmonics direction encoding from tiny-cuda-nn, a third-party CUDA dependency that is neither
vendored in /root/reference nor installable here (requirements_torch.txt pins no version).  This file restates the
published algorithm (Mueller et al., "Instant Neural Graphics Primitives", SIGGRAPH 2022, and tiny-cuda-nn's
`HashGrid` / `SphericalHarmonics` encodings as documented there):

  level l:  scale_l = 2^(l * log2(per_level_scale)) * base_resolution - 1,  resolution_l = ceil(scale_l) + 1
            entries_l = min(round_up(resolution_l^3, 8), 2^log2_hashmap_size);  levels are concatenated
  lookup :  pos = fma(x, scale_l, 0.5);  cell = floor(pos);  w = pos - cell;  trilinear weights over the 8 corners
            (bilinear over 4 for the 2-D grid: resolution_l^2 dense entries, hash over the first two primes);
            corner index = dense (x + y*res + z*res^2) while res^d stays within the level's entries, otherwise
            the coherent prime hash (x*1) ^ (y*2654435761) ^ (z*805459861), both modulo entries_l (uint32 arithmetic)
  output :  [N, n_levels * features_per_level], level-major
  SH     :  real spherical harmonics up to degree 4 (16 coefficients) of d = 2*x - 1 (inputs arrive in [0,1])

The pure-torch parts of the nerfacto path (samplers, losses, contraction, trunc_exp) ARE executable here and are
pinned by vectors recorded from the reference; the two encodings are not, and every parity statement about them
says so.
"""
import numpy as np

PRIMES = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))


def level_table(n_levels, base_resolution, per_level_scale, log2_hashmap_size, dims=3):
  """-> (offsets [L+1] in entries, resolutions [L], scales [L] float32).  dims: 3 (fields) or 2 (the nerfacto HA-NeRF
  ImplicitMask's image-plane grid, nerfacto.py:1036-1047)."""
  offs, ress, scales, off = [0], [], [], 0
  l2 = np.float32(np.log2(np.float32(per_level_scale)))
  for l in range(n_levels):
    scale = np.float32(np.exp2(np.float32(l) * l2) * np.float32(base_resolution) - np.float32(1.0))
    res = int(np.ceil(scale)) + 1
    n = min(res ** dims, 2 ** 31 - 1)
    n = (n + 7) // 8 * 8
    n = min(n, 1 << log2_hashmap_size)
    off += n
    offs.append(off); ress.append(res); scales.append(scale)
  return np.array(offs, np.int64), np.array(ress, np.int64), np.array(scales, np.float32)


def _index(cs, res, entries):
  """cs: list of D integer coordinate arrays."""
  D = len(cs)
  res = np.uint64(res)
  if int(res) ** D <= entries:       # dense: the stride never exceeds the level's size
    idx, stride = np.zeros(cs[0].shape, np.uint64), np.uint64(1)
    for c in cs:
      idx = idx + c.astype(np.uint64) * stride
      stride = stride * res
  else:
    with np.errstate(over='ignore'):
      idx = cs[0].astype(np.uint32) * PRIMES[0]
      for k in range(1, D):
        idx = idx ^ (cs[k].astype(np.uint32) * PRIMES[k])
  return (idx.astype(np.uint64) % np.uint64(entries)).astype(np.int64)


def _fma(x, sc):
  """fmaf(scale, x, 0.5) in binary32: the product of two binary32 numbers is exact in binary64, one rounding at the end."""
  return (x.astype(np.float64) * np.float64(sc) + 0.5).astype(np.float32)


def hashgrid_forward(x, table, offsets, resolutions, scales, F):
  """x [N,D] float32 in [0,1] (D = 3 or 2); table [sum entries, F] -> [N, L*F] (float64 accumulation of float32 operands)."""
  x = np.asarray(x, np.float32)
  D = x.shape[1]
  out = np.zeros((x.shape[0], len(resolutions) * F), np.float64)
  for l, (res, sc) in enumerate(zip(resolutions, scales)):
    pos = _fma(x, sc)
    cell = np.floor(pos)
    w = (pos - cell).astype(np.float64)
    c = cell.astype(np.int64)
    entries = int(offsets[l + 1] - offsets[l])
    for corner in range(1 << D):
      d = [(corner >> k) & 1 for k in range(D)]
      wt = np.ones(x.shape[0])
      for k in range(D):
        wt = wt * (w[:, k] if d[k] else 1 - w[:, k])
      idx = _index([c[:, k] + d[k] for k in range(D)], res, entries) + int(offsets[l])
      out[:, l * F:(l + 1) * F] += wt[:, None] * table[idx].astype(np.float64)
  return out


def hashgrid_backward(x, d_out, n_entries, offsets, resolutions, scales, F):
  """d loss / d table [n_entries, F] (float64)."""
  x = np.asarray(x, np.float32)
  D = x.shape[1]
  g = np.zeros((n_entries, F), np.float64)
  for l, (res, sc) in enumerate(zip(resolutions, scales)):
    pos = _fma(x, sc)
    cell = np.floor(pos)
    w = (pos - cell).astype(np.float64)
    c = cell.astype(np.int64)
    entries = int(offsets[l + 1] - offsets[l])
    for corner in range(1 << D):
      d = [(corner >> k) & 1 for k in range(D)]
      wt = np.ones(x.shape[0])
      for k in range(D):
        wt = wt * (w[:, k] if d[k] else 1 - w[:, k])
      idx = _index([c[:, k] + d[k] for k in range(D)], res, entries)
      v = wt[:, None] * np.asarray(d_out, np.float64)[:, l * F:(l + 1) * F]
      for f in range(F):       # (bincount into the level's own slice: np.add.at is ~100x slower at BASELINE config 5's sizes)
        g[int(offsets[l]):int(offsets[l + 1]), f] += np.bincount(idx, weights=v[:, f], minlength=entries)
  return g


def sh4(x01):
  """Real spherical harmonics, degree 4 (16 values), of d = 2 x - 1."""
  d = np.asarray(x01, np.float64) * 2 - 1
  x, y, z = d[:, 0], d[:, 1], d[:, 2]
  xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
  o = np.empty((d.shape[0], 16))
  o[:, 0] = 0.28209479177387814
  o[:, 1] = -0.48860251190291987 * y
  o[:, 2] = 0.48860251190291987 * z
  o[:, 3] = -0.48860251190291987 * x
  o[:, 4] = 1.0925484305920792 * xy
  o[:, 5] = -1.0925484305920792 * yz
  o[:, 6] = 0.94617469575755997 * z2 - 0.31539156525251999
  o[:, 7] = -1.0925484305920792 * xz
  o[:, 8] = 0.54627421529603959 * x2 - 0.54627421529603959 * y2
  o[:, 9] = 0.59004358992664352 * y * (-3 * x2 + y2)
  o[:, 10] = 2.8906114426405538 * xy * z
  o[:, 11] = 0.45704579946446572 * y * (1 - 5 * z2)
  o[:, 12] = 0.3731763325901154 * z * (5 * z2 - 3)
  o[:, 13] = 0.45704579946446572 * x * (1 - 5 * z2)
  o[:, 14] = 1.4453057213202769 * z * (x2 - y2)
  o[:, 15] = 0.59004358992664352 * x * (-x2 + 3 * y2)
  return o

"""`jax.random` as the reference uses it, on the device (csrc/hugs_prng.hip).

Keys are int32 CUDA tensors of shape [2] holding jax's two uint32 key words; they never leave the GPU, so splitting
and drawing cost one small launch each and no synchronisation.  Streams are bit-identical to jax's default
(non-partitionable) threefry generator -- see oracle/threefry_ref.py for what pins that.

  PRNGKey(seed)            train.py:46            key = (seed >> 32, seed & 0xffffffff)
  split(key, num=2)        train_utils.py:408     [num, 2] keys
  uniform(key, shape, ...) stepfun.py:207-209     float32 in [minval, maxval)
  normal(key, shape)       models.py:458-460      float32 N(0, 1): sqrt(2) erf_inv(uniform(-1, 1)) like jax
  bits(key, shape)                                raw uint32 draws (as int32 bit patterns)
  permutation(key, n)      models.py:644          jax's sort-by-random-keys shuffle (ceil(3 ln n / ln 2^32) rounds)
"""
import math

import numpy as np
import torch

from .. import _lib as L


def is_key(x):
  return torch.is_tensor(x) and x.dtype == torch.int32 and tuple(x.shape) == (2,)


def PRNGKey(seed, device='cuda'):
  seed = int(seed) & 0xFFFFFFFFFFFFFFFF
  words = np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], np.uint32).view(np.int32)
  return torch.from_numpy(words.copy()).to(device)


def _check(key):
  if not is_key(key):
    raise TypeError('PRNG keys are int32 tensors of shape [2] (random.PRNGKey)')
  if not key.is_cuda:
    raise L.HugsError('PRNG key is not on the GPU (no CPU fallback)')


def bits(key, shape):
  _check(key)
  shape = tuple(shape)
  out = torch.empty(shape, dtype=torch.int32, device=key.device)
  L.call('hugs_prng_bits', key, out.numel(), out)
  return out


def split(key, num=2):
  return bits(key, (num, 2))


def uniform(key, shape=(), minval=0., maxval=1.):
  _check(key)
  out = torch.empty(tuple(shape), dtype=torch.float32, device=key.device)
  L.call('hugs_prng_uniform', key, out.numel(), minval, maxval, out)
  return out


def normal(key, shape=()):
  """jax.random.normal(key, shape): float32 standard normal draws (models.py:458-460,478-481; flax's initialisers)."""
  _check(key)
  out = torch.empty(tuple(shape), dtype=torch.float32, device=key.device)
  L.call('hugs_prng_normal', key, out.numel(), out)
  return out


def fold_in(key, data):
  """jax.random.fold_in(key, data) for a uint32 `data`."""
  _check(key)
  out = torch.empty(2, dtype=torch.int32, device=key.device)
  d = int(data) & 0xFFFFFFFF
  L.call('hugs_prng_fold_in', key, d - (1 << 32) if d >= (1 << 31) else d, out)      # (the same 32 bits as a C int)
  return out


def flax_param_key(root, path, counter, variant='lazy'):
  """The key flax hands a parameter's initialiser: module `path` (tuple of scope names) and the scope's make_rng counter
  (1 for the first `self.param` of the scope: nn.Dense's kernel, nn.Embed's embedding) folded into the `params` rng.
  flax.core.scope is un-vendored and unversioned in the reference (requirements_jax.txt: `flax`), and the folding changed
  between releases, hence the variants (restated from the published source; PARITY UNPINNED):
    'lazy'      flax >= 0.6.x LazyRng: ONE fold_in of uint32(sha1(path strings + counter bytes)[:4])   (default)
    'lazy_sep'  the same with config.flax_fix_rng_separator: a 0x00 byte in front of every component
    'legacy'    older flax: fold_in(sha1(name)[:4]) per scope name on the way down, then fold_in(counter)"""
  import hashlib
  h32 = lambda b: int.from_bytes(hashlib.sha1(b).digest()[:4], 'big')
  ib = lambda x: x.to_bytes((x.bit_length() + 7) // 8, 'big')
  if variant == 'legacy':
    k = root
    for name in path:
      k = fold_in(k, h32(name.encode('utf-8')))
    return fold_in(k, counter)
  if variant not in ('lazy', 'lazy_sep'):
    raise ValueError(f'flax rng variant {variant!r}')
  sep = b'\x00' if variant == 'lazy_sep' else b''
  data = b''.join(sep + (x.encode('utf-8') if isinstance(x, str) else ib(x)) for x in tuple(path) + (counter,))
  return fold_in(root, h32(data))


def step_jitter_max_levels():
  """The level cap of the fused launch, read from the library (csrc/hugs_prng.hip HUGS_STEP_JITTER_MAX_LEVELS)."""
  return int(L.lib().cdll.hugs_prng_step_jitter_max_levels())


def step_jitter(key, sizes, maxvals):
  """One training step's whole consumption of the stream in ONE launch (csrc/hugs_prng.hip k_step_jitter), bit-identical to

      rng, k = split(key)                                    # train_utils.py:408
      for each level l:  kl, k = split(k); u_l = uniform(kl, (sizes[l],), maxval=maxvals[l]); _, k = split(k)     # models.py:196,230

  Returns ([u_0, u_1, ...], rng)."""
  _check(key)
  outs = [torch.empty(int(n), dtype=torch.float32, device=key.device) for n in sizes]
  new_key = torch.empty(2, dtype=torch.int32, device=key.device)
  n_arr = np.ascontiguousarray(sizes, np.int64)
  m_arr = np.ascontiguousarray(maxvals, np.float32)
  p_arr = np.ascontiguousarray([o.data_ptr() for o in outs], np.uint64)
  L.call('hugs_prng_step_jitter', key, len(outs), n_arr.ctypes.data, m_arr.ctypes.data, p_arr.ctypes.data, new_key)
  return outs, new_key


def permutation(key, n):
  """jax.random.permutation(key, n) (jax/_src/random.py _shuffle): repeated stable sort by fresh 32-bit keys."""
  x = torch.arange(n, device=key.device)
  rounds = int(math.ceil(3 * math.log(max(1, n)) / math.log(2 ** 32 - 1)))
  for _ in range(rounds):
    key, sub = split(key)
    k = bits(sub, (n,)).long() & 0xFFFFFFFF      # unsigned order
    x = x[torch.sort(k, stable=True).indices]
  return x

"""CPU oracle for the JAX-compatible PRNG.  TEST INFRASTRUCTURE ONLY.

The reference draws every random number through `jax.random` (train.py:79 PRNGKey(20200823), train_utils.py:415
and models.py:131-141 `random.split`, stepfun.py:203-209 `random.uniform`).  jax is a third-party dependency that
is not vendored in /root/reference and not installed here (requirements_jax.txt pins no version), so this file
restates the published algorithm of its default generator:

  * Threefry-2x32, 20 rounds (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), key schedule
    constant 0x1BD11BDA, rotations (13,15,26,6 | 17,29,16,24);
  * jax/_src/prng.py (non-partitionable threefry, the default through jax 0.4): `threefry_seed` key = (hi32, lo32)
    of the seed; `threefry_2x32(key, counts)` pads an odd count array with one zero, uses the FIRST HALF of the
    counts as word 0 and the SECOND HALF as word 1, and concatenates the two output words; `split(key, n)` =
    that on iota(2n) reshaped [n,2]; `random_bits(key, 32, shape)` = that on iota(size);
  * jax/_src/random.py: uniform = bitcast((bits >> 9) | 0x3F800000) - 1, scaled to [minval, maxval) and clamped at
    minval; normal = sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1)).

Pinned (tests/test_oracle_threefry.py) by the Random123 known-answer vectors for threefry2x32-20 and by the golden
numbers the reference's own tests/datasets_test.py:66-104 holds, which are functions of PRNGKey(0) -> split ->
uniform / normal.
"""
import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
M32 = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
  return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & M32


def threefry2x32(key, x0, x1):
  """key (k0, k1) ints; x0, x1 uint32 arrays of equal shape -> (y0, y1) uint32 arrays."""
  k0, k1 = np.uint64(key[0]), np.uint64(key[1])
  ks = (k0, k1, k0 ^ k1 ^ np.uint64(0x1BD11BDA))
  a = (np.asarray(x0, np.uint64) + ks[0]) & M32
  b = (np.asarray(x1, np.uint64) + ks[1]) & M32
  for i in range(5):
    for r in _ROT[i % 2]:
      a = (a + b) & M32
      b = _rotl(b, r) ^ a
    a = (a + ks[(i + 1) % 3]) & M32
    b = (b + ks[(i + 2) % 3] + np.uint64(i + 1)) & M32
  return a.astype(np.uint32), b.astype(np.uint32)


def prng_key(seed):
  seed = int(seed)
  return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], np.uint32)


def _blocks(key, n):
  """jax's threefry_2x32(key, iota(n)): first half -> word 0, second half -> word 1, outputs concatenated."""
  c = np.arange(n + (n & 1), dtype=np.uint32)
  c[n:] = 0
  h = c.size // 2
  y0, y1 = threefry2x32(key, c[:h], c[h:])
  return np.concatenate([y0, y1])[:n]


def split(key, num=2):
  return _blocks(key, 2 * num).reshape(num, 2)


def random_bits(key, shape):
  n = int(np.prod(shape, dtype=np.int64))
  return _blocks(key, n).reshape(shape)


def uniform(key, shape=(), minval=0., maxval=1.):
  bits = random_bits(key, shape)
  f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.)
  lo, hi = np.float32(minval), np.float32(maxval)
  return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)


def normal(key, shape=()):
  from scipy.special import erfinv
  lo = np.nextafter(np.float32(-1.), np.float32(0.))
  u = uniform(key, shape, lo, 1.)
  return (np.float32(np.sqrt(2)) * erfinv(u.astype(np.float64)).astype(np.float32)).astype(np.float32)


def fold_in(key, data):
  """jax.random.fold_in(key, data): threefry_2x32(key, threefry_seed(uint32 data)) -- the seed words are (0, data)."""
  y0, y1 = threefry2x32(key, np.array([0], np.uint32), np.array([int(data) & 0xFFFFFFFF], np.uint32))
  return np.array([y0[0], y1[0]], np.uint32)


def flax_param_key(root, path, counter, variant='lazy'):
  """The initialiser key of a flax parameter (restated from flax/core/scope.py as published; flax is absent here: UNPINNED).
  'lazy' (LazyRng, flax >= 0.6): fold_in(root, uint32(sha1(b''.join(path names as utf-8 + counter big-endian bytes))[:4]));
  'lazy_sep': the same with a 0x00 byte before every component (config.flax_fix_rng_separator);
  'legacy': fold_in(sha1(name)[:4]) per scope on the way down (Scope.push / _fold_in_str), then fold_in(counter)."""
  import hashlib
  h32 = lambda b: int.from_bytes(hashlib.sha1(b).digest()[:4], 'big')
  if variant == 'legacy':
    k = root
    for name in path:
      k = fold_in(k, h32(name.encode()))
    return fold_in(k, counter)
  sep = b'\x00' if variant == 'lazy_sep' else b''
  parts = [sep + (x.encode() if isinstance(x, str) else x.to_bytes((x.bit_length() + 7) // 8, 'big')) for x in tuple(path) + (counter,)]
  return fold_in(root, h32(b''.join(parts)))


def he_uniform(key, shape):
  """jax.nn.initializers.he_uniform()(key, shape, float32) = variance_scaling(2.0, 'fan_in', 'uniform'):
  random.uniform(key, shape, float32, -1) * sqrt(3 * variance), variance = float32(2 / fan_in), fan_in = shape[-2]."""
  variance = np.float32(2.0 / shape[-2])
  return (uniform(key, shape, -1.0, 1.0) * np.sqrt(np.float32(3.0) * variance)).astype(np.float32)


def embed_init(key, shape):
  """flax nn.Embed's default_embed_init = variance_scaling(1.0, 'fan_in', 'normal', out_axis=0): fan_in = features."""
  return (normal(key, shape) * np.sqrt(np.float32(1.0 / shape[-1]))).astype(np.float32)

"""Numpy-backed stand-in for the tiny slice of `jax` that the reference's leaf
modules (MipNeRF360/internal/{math,stepfun,render,coord,geopoly}.py) touch.

TEST INFRASTRUCTURE ONLY.  It exists so that `gen_fixtures.py` can *import and
execute the reference's own source* in the build container (jax/flax are not
installed and there is no network) and record input/output vectors.  Nothing in
the product or in the GPU-side tests imports this file.

Semantics pinned here (they are the stand-in's, not XLA's -- see DESIGN.md
"parity pinning"): float32 everywhere, `cumsum` sequential left-to-right
(numpy), `softmax` = exp(x-max)/sum with numpy's pairwise float32 sum, `sort`
numpy, `jax.linearize` by float64 central differences.
"""
import sys
import types
import numpy as np

f32 = np.float32


class _Key:
  """Stand-in PRNG key: hands out pre-drawn U[0,1) float32 numbers so the
  fixture can store exactly what the reference consumed."""

  def __init__(self, seed):
    self.rng = np.random.default_rng(seed)
    self.draws = []

  def uniform01(self, shape):
    u = self.rng.random(shape, dtype=np.float32)
    self.draws.append(u)
    return u


def _small_int(a):
  # jax: int32 (op) float32 -> float32, and 2**int32 stays exact.  numpy would promote
  # int32*float32 to float64, so the stand-in hands out float32 "integers" (exact up to
  # 2**24; the index consumers below cast back to int64).
  return np.asarray(a).astype(np.float32)


KEEP64 = [False]  # set while the reference's source is evaluated in float64 (central differences)


def build():
  jnp = types.ModuleType('jax.numpy')
  for name in dir(np):
    if not name.startswith('_'):
      setattr(jnp, name, getattr(np, name))

  keep64 = KEEP64

  def _f(x):
    x = np.asarray(x)
    if keep64[0]:
      return x
    return x.astype(f32) if x.dtype == np.float64 else x

  jnp.float32 = np.float32
  jnp.matmul = lambda a, b, precision=None: np.matmul(_f(a), _f(b))
  jnp.zeros = lambda shape, dtype=f32: np.zeros(shape, dtype)
  jnp.ones = lambda shape, dtype=f32: np.ones(shape, dtype)
  jnp.eye = lambda n, dtype=f32: np.eye(n, dtype=dtype)
  jnp.full = lambda shape, v, dtype=f32: np.full(shape, v, dtype)
  jnp.array = lambda x, dtype=None: _f(np.array(x, dtype=dtype))
  jnp.linspace = lambda a, b, n: np.linspace(a, b, n, dtype=np.float64).astype(f32)
  jnp.arange = lambda *a: _small_int(np.arange(*a))
  jnp.concatenate = lambda xs, axis=0: np.concatenate([_f(x) for x in xs], axis=axis)
  jnp.nan_to_num = lambda x, nan=0.0, posinf=None, neginf=None: np.nan_to_num(
      x, nan=nan, posinf=posinf, neginf=neginf)
  jnp.take_along_axis = lambda a, i, axis: np.take_along_axis(a, i.astype(np.int64), axis=axis)
  jnp.interp = lambda x, xp, fp: _f(np.interp(x, xp, fp))
  jnp.where = lambda c, a, b: _f(np.where(c, a, b))

  def vectorize(fn, signature=None):
    v = np.vectorize(fn, signature=signature)
    return lambda *a: _f(v(*a))

  jnp.vectorize = vectorize

  nn = types.ModuleType('jax.nn')

  def softmax(x, axis=-1):
    x = _f(x)
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)

  nn.softmax = softmax

  random = types.ModuleType('jax.random')

  def uniform(key, shape, minval=0., maxval=1.):
    u = key.uniform01(tuple(shape))
    return (f32(minval) + u * (f32(maxval) - f32(minval))).astype(f32)

  random.uniform = uniform

  lax = types.ModuleType('jax.lax')

  class Precision:
    HIGHEST = 'highest'

  lax.Precision = Precision
  lax.stop_gradient = lambda x: x

  jax = types.ModuleType('jax')
  jax.numpy = jnp
  jax.nn = nn
  jax.random = random
  jax.lax = lax

  def vmap(fn, in_axes=0, out_axes=0):
    def wrapped(*args):
      axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
      n = args[0].shape[axes[0]]
      outs = [fn(*[np.take(a, i, axis=ax) for a, ax in zip(args, axes)]) for i in range(n)]
      return _f(np.stack(outs, axis=out_axes))
    return wrapped

  jax.vmap = vmap

  def custom_jvp(fn):
    fn.defjvp = lambda g: g
    return fn

  jax.custom_jvp = custom_jvp

  def linearize(fn, x):
    x64 = np.asarray(x, np.float64)
    y = fn(x64 if keep64[0] else np.asarray(x, f32))
    # step relative to max(1, |x|): contract's Jacobian at |x| = 1e6 has a radial entry ~1/|x|^2 next to
    # tangential entries ~1/|x|; an absolute 1e-6 step loses the radial one in float64 rounding.
    h = 1e-4 * np.maximum(1.0, np.linalg.norm(x64, axis=-1, keepdims=True))

    def lin(v):
      v64 = np.asarray(v, np.float64)
      # float64 evaluation of the same source: the reference functions are dtype-generic, eps clamps
      # (finfo float32) stay as written.  The tangent is normalised so that huge covariances (far=1e6)
      # stay in the linear regime of fn.
      scale = np.maximum(np.linalg.norm(v64, axis=-1, keepdims=True), 1e-30)
      prev, keep64[0] = keep64[0], True
      try:
        jv = (fn(x64 + h * v64 / scale) - fn(x64 - h * v64 / scale)) / (2 * h) * scale
      finally:
        keep64[0] = prev
      return jv if prev else jv.astype(f32)

    return y, lin

  jax.linearize = linearize
  return jax


def install():
  jax = build()
  sys.modules['jax'] = jax
  sys.modules['jax.numpy'] = jax.numpy
  sys.modules['jax.nn'] = jax.nn
  sys.modules['jax.random'] = jax.random
  sys.modules['jax.lax'] = jax.lax
  return jax

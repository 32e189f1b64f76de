"""ctypes loader for the C oracle (oracle/stepfun_ref.c).  TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int32)


def build():
  subprocess.check_call(['make', '-s', '-C', _HERE, 'liborc_stepfun.so'])


def lib():
  global _LIB
  if _LIB is None:
    path = os.path.join(_HERE, 'liborc_stepfun.so')
    if not os.path.exists(path):
      build()
    _LIB = ctypes.CDLL(path)
    _LIB.orc_wave_sum.restype = ctypes.c_float
  return _LIB


def _p(a):
  return a.ctypes.data_as(f32p)


def _c(a):
  return np.ascontiguousarray(a, np.float32)


# order of the three order-sensitive sums (stepfun_ref.c orc_set_sum_order): 1 = reference order (numpy pairwise sums,
# sequential cumsum -- the product's default), 0 = wave order
SUM_ORDER = 0 if os.environ.get('HUGS_SAMPLER_ORDER', 'reference') == 'wave' else 1


def set_sum_order(order):
  lib().orc_set_sum_order(ctypes.c_int(int(order)))


def expf(x):
  x = _c(x); y = np.empty_like(x)
  lib().orc_expf_vec(_p(x), ctypes.c_int(x.size), _p(y))
  return y


def logf(x):
  x = _c(x); y = np.empty_like(x)
  lib().orc_logf_vec(_p(x), ctypes.c_int(x.size), _p(y))
  return y


def wave_sum(x):
  x = _c(x); lib().orc_set_sum_order(ctypes.c_int(0))
  return np.float32(lib().orc_wave_sum(_p(x), ctypes.c_int(x.size)))


def wave_cumsum(x):
  x = _c(x); y = np.empty_like(x); lib().orc_set_sum_order(ctypes.c_int(0))
  lib().orc_wave_cumsum(_p(x), ctypes.c_int(x.size), _p(y))
  return y


def max_dilate_weights(t, w, dilation, lo, hi, sum_order=None):
  """stepfun.py:112-128 with renormalize=True. t[N,n+1], w[N,n] -> [N,3n+1],[N,3n]."""
  t = _c(t); w = _c(w)
  set_sum_order(SUM_ORDER if sum_order is None else sum_order)
  N, n = w.shape
  td = np.empty((N, 3 * n + 1), np.float32); wd = np.empty((N, 3 * n), np.float32)
  for r in range(N):
    rc = lib().orc_max_dilate_weights(_p(t[r]), _p(w[r]), n, ctypes.c_float(dilation),
                                      ctypes.c_float(lo), ctypes.c_float(hi), _p(td[r]), _p(wd[r]))
    if rc: raise ValueError(f'orc_max_dilate_weights rc={rc}')
  return td, wd


def sample_intervals(u, t, logits, lo, hi, sum_order=None):
  """stepfun.py:214-263 with explicit u[N,ns]. Returns (sdist[N,ns+1], idx[N,ns])."""
  u = _c(u); t = _c(t); logits = _c(logits)
  set_sum_order(SUM_ORDER if sum_order is None else sum_order)
  N, ns = u.shape
  n = logits.shape[1]
  if ns <= 1:
    raise ValueError(f'num_samples must be > 1, is {ns}.')
  out = np.empty((N, ns + 1), np.float32); idx = np.empty((N, ns), np.int32)
  for r in range(N):
    rc = lib().orc_sample_intervals(_p(u[r]), ns, _p(t[r]), _p(logits[r]), n, ctypes.c_float(lo),
                                    ctypes.c_float(hi), _p(out[r]), idx[r].ctypes.data_as(i32p))
    if rc: raise ValueError(f'orc_sample_intervals rc={rc}')
  return out, idx


def level_sample(t_prev, w_prev, do_dilate, dilation, lo, hi, anneal, pad, u_base, jitter,
                 raydist, near, far, sum_order=None):
  """One sampling level (models.py:155-212). Returns sdist, tdist, idx."""
  set_sum_order(SUM_ORDER if sum_order is None else sum_order)
  t_prev = _c(t_prev); w_prev = _c(w_prev); u_base = _c(u_base)
  near = _c(near).reshape(-1); far = _c(far).reshape(-1)
  N, n_prev = w_prev.shape
  ns = u_base.shape[0]
  sd = np.empty((N, ns + 1), np.float32); td = np.empty((N, ns + 1), np.float32)
  idx = np.empty((N, ns), np.int32)
  if jitter is not None and np.asarray(jitter).ndim == 2 and np.asarray(jitter).shape[1] == ns and ns > 1:
    # one draw per sample (Model.single_jitter = False, stepfun.py:203-209)
    jit = _c(jitter)
    rc = lib().orc_level_sample_batch_pj(
        N, _p(t_prev), _p(w_prev), n_prev, int(do_dilate), ctypes.c_float(dilation),
        ctypes.c_float(lo), ctypes.c_float(hi), ctypes.c_float(anneal), ctypes.c_float(pad),
        _p(u_base), _p(jit), ns, int(raydist), _p(near), _p(far), _p(sd), _p(td), idx.ctypes.data_as(i32p))
    if rc: raise ValueError(f'orc_level_sample rc={rc}')
    return sd, td, idx
  jit = None if jitter is None else _c(jitter).reshape(-1)
  rc = lib().orc_level_sample_batch(
      N, _p(t_prev), _p(w_prev), n_prev, int(do_dilate), ctypes.c_float(dilation),
      ctypes.c_float(lo), ctypes.c_float(hi), ctypes.c_float(anneal), ctypes.c_float(pad),
      _p(u_base), _p(jit) if jit is not None else None, ns, int(raydist), _p(near), _p(far),
      _p(sd), _p(td), idx.ctypes.data_as(i32p))
  if rc: raise ValueError(f'orc_level_sample rc={rc}')
  return sd, td, idx


def searchsorted(a, v):
  a = _c(a); v = _c(v)
  N, na = a.shape; nv = v.shape[1]
  lo = np.empty((N, nv), np.int32); hi = np.empty((N, nv), np.int32)
  for r in range(N):
    lib().orc_searchsorted(_p(a[r]), na, _p(v[r]), nv, lo[r].ctypes.data_as(i32p),
                           hi[r].ctypes.data_as(i32p))
  return lo, hi

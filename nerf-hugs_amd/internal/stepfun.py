"""Step-function sampling entry points (reference MipNeRF360/internal/stepfun.py), HIP underneath.

`sample_intervals` / `max_dilate_weights` keep the reference's argument meaning; the PRNG stays
outside the kernels: callers pass the uniform draws (`u01`) the reference would have taken from
`jax.random.uniform` (stepfun.py:203-209)."""
import os

import numpy as np
import torch

from .. import _lib

EPS = float(np.finfo(np.float32).eps)


class Jitter(list):
  """Per-level jitter draws that are already scaled to [0, max_jitter) (the jax.random path).  `mlp_keys`: the key each
  level's MLP call receives (models.py:230), kept when the draws come from Model.level_jitter -- the density / bottleneck
  noise draws are split off it (models.py:435,458-460,478-481)."""
  scaled = True
  mlp_keys = None
  bg_rgbs = None      # per level [N, 3] background draws when Model.bg_intensity_range is a real range (models.py:256-261)


def sample_u(num_samples, randomized, deterministic_center=True):
  """Fixed part of the inverse-CDF abscissae (stepfun.py:191-209) and the jitter scale.
  float64 host linspace rounded to float32 (jnp.linspace's own rounding is unpinned)."""
  if not randomized:
    if deterministic_center:
      pad = 1 / (2 * num_samples)
      u = np.linspace(pad, 1. - pad - EPS, num_samples)
    else:
      u = np.linspace(0, 1. - EPS, num_samples)
    return u.astype(np.float32), 0.0
  # binary32 scalar arithmetic, as `jnp.finfo(jnp.float32).eps` makes it in the reference (stepfun.py:203-205)
  eps, one = np.float32(EPS), np.float32(1)
  u_max = eps + (one - eps) / np.float32(num_samples)
  max_jitter = (one - u_max) / np.float32(num_samples - 1) - eps
  return np.linspace(0, float(1 - u_max), num_samples).astype(np.float32), float(max_jitter)


_UB_CACHE = {}


def _u_base_on_device(num_samples, randomized, dev):
  """sample_u's abscissae as a device tensor, uploaded once: a pageable host->device copy inside the step is a
  synchronisation point (the host waits for everything queued before it), which kept the CPU from running ahead."""
  key = (num_samples, bool(randomized), str(dev))
  hit = _UB_CACHE.get(key)
  if hit is None:
    ub, mj = sample_u(num_samples, randomized)
    hit = (torch.from_numpy(ub).to(dev), mj)
    _UB_CACHE[key] = hit
  return hit


RAYDIST = {None: 0, 'reciprocal': 1, 'log': 2, 'exp': 3, 'sqrt': 4, 'square': 5, 'piecewise': 6}      # kernel codes (coord.py:78-90)


# Order of the sampler's three order-sensitive float sums (include/hugs.h hugs_level_sample_fwd): 1 = reference order
# (numpy-pairwise jnp.sum, sequential jnp.cumsum: the order the reference-executed fixtures pin) -- what ships;
# 0 = wave order (lane-blocked tree), kept as an A/B switch (HUGS_SAMPLER_ORDER=wave).
SUM_ORDER = 0 if os.environ.get('HUGS_SAMPLER_ORDER', 'reference') == 'wave' else 1


def level_sample(t_prev, w_prev, do_dilate, dilation, domain, anneal, resample_padding, num_samples, u01,
                 raydist, near, far, return_debug=False, jitter=None, sum_order=None):
  """One hierarchical-sampling level (models.py:155-212): [dilate] -> logits -> sample_intervals -> s_to_t.

  t_prev [N, n+1], w_prev [N, n] float32 cuda; u01: None (rng=None) or [N] / [N, num_samples] U[0,1) draws;
  jitter: instead of u01, draws already scaled to [0, max_jitter) (random.uniform(key, ..., maxval=max_jitter)).
  Returns sdist, tdist ([N, num_samples+1]) (+ idx, t_in, w_in when return_debug)."""
  if num_samples <= 1:
    raise ValueError(f'num_samples must be > 1, is {num_samples}.')
  N, n_prev = w_prev.shape
  dev = t_prev.device
  ub, mj = _u_base_on_device(num_samples, u01 is not None or jitter is not None, dev)
  stride = 1
  if jitter is None and u01 is not None:
    jitter = u01.to(torch.float32) * np.float32(mj)
  if jitter is not None:
    jitter = jitter.to(torch.float32).contiguous()
    stride = 1 if jitter.dim() == 1 or jitter.shape[-1] == 1 else num_samples
  sdist = torch.empty(N, num_samples + 1, device=dev)
  tdist = torch.empty_like(sdist)
  idx = t_in = w_in = None
  if return_debug:
    n_in = 3 * n_prev - 2 if do_dilate else n_prev
    idx = torch.empty(N, num_samples, dtype=torch.int32, device=dev)
    t_in = torch.empty(N, n_in + 1, device=dev)
    w_in = torch.empty(N, n_in, device=dev)
  rd = RAYDIST.get(raydist)
  if rd is None:
    raise NotImplementedError(f"raydist_fn {raydist!r}: coord.py:78-90 knows None, 'piecewise' and jnp.reciprocal / log / exp / "
                              "sqrt / square")
  if torch.is_tensor(anneal):      # a captured train step: the annealing factor is a device scalar (train_utils._GraphStep)
    if return_debug:
      raise ValueError('level_sample: the debug outputs are not available with a device-resident anneal')
    _lib.call('hugs_level_sample_fwd_dyn', N, t_prev.contiguous(), w_prev.contiguous(), n_prev, int(do_dilate), dilation,
              domain[0], domain[1], anneal, resample_padding, ub, jitter, stride, num_samples, rd,
              SUM_ORDER if sum_order is None else int(sum_order), near.reshape(-1).contiguous(), far.reshape(-1).contiguous(), sdist, tdist)
    return sdist, tdist
  _lib.call('hugs_level_sample_fwd', N, t_prev.contiguous(), w_prev.contiguous(), n_prev, int(do_dilate), dilation,
            domain[0], domain[1], anneal, resample_padding, ub, jitter, stride, num_samples, rd,
            SUM_ORDER if sum_order is None else int(sum_order), near.reshape(-1).contiguous(), far.reshape(-1).contiguous(), sdist, tdist, idx, t_in, w_in)
  if return_debug:
    return sdist, tdist, idx, t_in, w_in
  return sdist, tdist


def sample_intervals(u01, t, w_logits_weights, num_samples, single_jitter=False, domain=(-float('inf'), float('inf'))):
  """stepfun.py:214-263 on weights (logits = log w; anneal 1, no dilation); returns sdist only."""
  near = torch.zeros(t.shape[0], device=t.device)
  far = torch.ones(t.shape[0], device=t.device)
  lo = max(domain[0], -3.0e38)
  hi = min(domain[1], 3.0e38)
  return level_sample(t, w_logits_weights, False, 0., (lo, hi), 1.0, 0.0, num_samples, u01, None, near, far)[0]

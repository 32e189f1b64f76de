"""Golden vectors for Model.single_jitter = False (one jitter draw per SAMPLE): the reference's own stepfun.sample_intervals
(MipNeRF360/internal/stepfun.py:164-263, `d = num_samples` at :203-209) executed under the numpy jax stand-in, two chained levels
(uniform first histogram; dilated + annealed second one, models.py:155-212).  Writes tests/golden/ref_persample_jitter.npz (data only).

    python tests/golden/gen_persample_jitter_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/MipNeRF360'
f32 = np.float32


def main():
  if not os.path.isdir(REF):
    raise SystemExit('needs the reference checkout at ' + REF)
  sys.path.insert(0, HERE)
  import _jax_standin
  _jax_standin.install()
  sys.path.insert(0, REF)
  from internal import stepfun
  rng = np.random.default_rng(7)
  n, S0, S1 = 12, 32, 24
  out = {}
  key = _jax_standin._Key(11)
  sdist = np.concatenate([np.zeros((n, 1), f32), np.ones((n, 1), f32)], -1)
  weights = np.ones((n, 1), f32)
  with np.errstate(divide='ignore'):
    logits = np.where(sdist[..., 1:] > sdist[..., :-1], f32(1.0) * np.log(weights), -np.inf).astype(f32)
  nd = len(key.draws)
  s0 = stepfun.sample_intervals(key, sdist, logits, S0, single_jitter=False, domain=(0., 1.))
  out['l0_u01'] = key.draws[nd]
  out['l0_sdist'] = np.asarray(s0, f32)
  # a bumpy histogram on the level-0 intervals, dilated and annealed as models.py:161-193 does for level 1
  w0 = (rng.uniform(0.02, 1.0, (n, S0)) ** 3).astype(f32)
  w0 = (w0 / w0.sum(-1, keepdims=True)).astype(f32)
  out['l0_weights'] = w0
  dilation = 0.0025 + 0.5 / S0
  sd, wd = stepfun.max_dilate_weights(np.asarray(s0, f32), w0, dilation, domain=(0., 1.), renormalize=True)
  sd, wd = sd[..., 1:-1], wd[..., 1:-1]
  anneal = f32((10 * 0.37) / (9 * 0.37 + 1))
  with np.errstate(divide='ignore'):
    lg = np.where(sd[..., 1:] > sd[..., :-1], anneal * np.log(wd + f32(0.0)), -np.inf).astype(f32)
  nd = len(key.draws)
  s1 = stepfun.sample_intervals(key, sd, lg, S1, single_jitter=False, domain=(0., 1.))
  out['l1_u01'] = key.draws[nd]
  out['l1_sdist'] = np.asarray(s1, f32)
  out['meta'] = np.array([n, S0, S1, dilation, float(anneal)], np.float64)
  assert out['l0_u01'].shape == (n, S0) and out['l1_u01'].shape == (n, S1), (out['l0_u01'].shape, out['l1_u01'].shape)
  np.savez_compressed(os.path.join(HERE, 'ref_persample_jitter.npz'), **out)
  print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
  main()

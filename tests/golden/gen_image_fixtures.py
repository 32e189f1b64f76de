"""Golden vectors for the eval-side image helpers, recorded by EXECUTING THE REFERENCE'S OWN
MipNeRF360/internal/image.py (linear_to_srgb :48, srgb_to_linear :59, downsample :70, color_correct :82,
mse_to_psnr :28) under the numpy-backed jax stand-in.  `dm_pix` (absent) is only touched by MetricHarness, which
is not exercised; an empty placeholder module satisfies the import.  Build container only; data only is committed.

    python tests/golden/gen_image_fixtures.py      # rewrites tests/golden/ref_image.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/MipNeRF360'


def main():
  if not os.path.isdir(REF):
    raise SystemExit('needs the reference checkout at ' + REF)
  sys.path.insert(0, HERE)
  import _jax_standin
  _jax_standin.install()
  sys.modules['dm_pix'] = types.ModuleType('dm_pix')
  sys.modules['dm_pix'].ssim = None
  sys.path.insert(0, REF)
  from internal import image
  rng = np.random.default_rng(82)
  out = {}
  x = np.concatenate([np.linspace(-0.2, 1.4, 400), [0.0031308, 0.04045, 0., 1.]])
  out['curve/x'] = x
  out['curve/linear_to_srgb'] = np.asarray(image.linear_to_srgb(x, xnp=np), np.float64)
  out['curve/srgb_to_linear'] = np.asarray(image.srgb_to_linear(x, xnp=np), np.float64)
  img = rng.uniform(size=(12, 8, 3))
  out['down/img'] = img
  out['down/by2'] = np.asarray(image.downsample(img, 2), np.float64)
  out['down/by4'] = np.asarray(image.downsample(img, 4), np.float64)
  for i in range(3):
    im0 = rng.uniform(0.02, 0.98, (24, 20, 3))
    ccm = np.eye(3) + rng.normal(size=(3, 3)) * 0.08
    im1 = np.clip(im0.reshape(-1, 3) @ ccm, 0, 1).reshape(im0.shape) + rng.normal() * 0.1 * im0 ** 2 + rng.normal() * 0.05
    im1 = np.clip(im1 + rng.normal(size=im1.shape) * (0.01 if i == 2 else 0.), 0, 1)
    out[f'cc{i}/img'], out[f'cc{i}/ref'] = im0, im1
    out[f'cc{i}/out'] = np.asarray(image.color_correct(im0, im1), np.float64)
  mse = np.exp(np.linspace(-10, 0, 16))
  out['psnr/mse'] = mse
  out['psnr/psnr'] = np.asarray(image.mse_to_psnr(mse), np.float64)
  path = os.path.join(HERE, 'ref_image.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, len(out), 'arrays', os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()

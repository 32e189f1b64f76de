"""Pins oracle/image_ref.py and the host-side colour helpers: vectors recorded from the reference's image.py
(tests/golden/ref_image.npz), the golden tables its tests/image_test.py:91-130 holds, and an independent
direct-window SSIM."""
import os

import numpy as np
import torch

from oracle import image_ref as I

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_image.npz'))

# tests/image_test.py:95-107 (expected outputs held by the reference's test)
LINEAR_GT = np.array([
    0.00000000, 0.00122856, 0.00245712, 0.00372513, 0.00526076, 0.00711347, 0.00929964, 0.01183453, 0.01473243,
    0.01800687, 0.02167065, 0.02573599, 0.03021459, 0.03511761, 0.04045585, 0.04623971, 0.05247922, 0.05918410,
    0.06636375, 0.07402734, 0.08218378, 0.09084171, 0.10000957, 0.10969563, 0.11990791, 0.13065430, 0.14194246,
    0.15377994, 0.16617411, 0.17913227, 0.19266140, 0.20676863, 0.22146071, 0.23674440, 0.25262633, 0.26911288,
    0.28621066, 0.30392596, 0.32226467, 0.34123330, 0.36083785, 0.38108405, 0.40197787, 0.42352500, 0.44573134,
    0.46860245, 0.49214387, 0.51636110, 0.54125960, 0.56684470, 0.59312177, 0.62009590, 0.64777250, 0.67615650,
    0.70525320, 0.73506740, 0.76560410, 0.79686830, 0.82886493, 0.86159873, 0.89507430, 0.92929670, 0.96427040,
    1.00000000])
# tests/image_test.py:114-128, first / middle / last entries
PSNR_GT = {0: 43.429447, 31: 22.059400, 62: 0.68935364, 63: 0.}


def test_srgb_and_psnr_reference_goldens():
  from nerf_hugs_amd.internal import image
  s = np.linspace(0, 1, 64)
  np.testing.assert_allclose(I.srgb_to_linear(s), LINEAR_GT, atol=1e-5, rtol=1e-5)
  np.testing.assert_allclose(image.srgb_to_linear(torch.from_numpy(s)).numpy(), LINEAR_GT, atol=1e-5, rtol=1e-5)
  mse = np.exp(np.linspace(-10, 0, 64))
  p = image.mse_to_psnr(torch.from_numpy(mse)).numpy()
  for i, v in PSNR_GT.items():
    assert abs(p[i] - v) < 1e-4
  x = torch.linspace(-1, 3, 10000, dtype=torch.float64)     # image_test.py:78-83 round trips
  np.testing.assert_allclose(image.linear_to_srgb(image.srgb_to_linear(x)).numpy(), x.numpy(), atol=1e-5, rtol=1e-5)
  np.testing.assert_allclose(image.srgb_to_linear(image.linear_to_srgb(x)).numpy(), x.numpy(), atol=1e-5, rtol=1e-5)
  for v in (-0.9, 0., 0.9):
    assert abs(image.dssim_to_ssim(image.ssim_to_dssim(v)) - v) < 1e-12
  for v in (10., 20., 30.):
    assert abs(float(image.mse_to_psnr(image.psnr_to_mse(torch.tensor(v, dtype=torch.float64)))) - v) < 1e-9


def test_vectors_recorded_from_reference_image_py():
  from nerf_hugs_amd.internal import image
  x = G['curve/x']
  np.testing.assert_allclose(I.linear_to_srgb(x), G['curve/linear_to_srgb'], rtol=1e-12, atol=1e-15)
  np.testing.assert_allclose(I.srgb_to_linear(x), G['curve/srgb_to_linear'], rtol=1e-12, atol=1e-15)
  xt = torch.from_numpy(x)
  np.testing.assert_allclose(image.linear_to_srgb(xt).numpy(), G['curve/linear_to_srgb'], rtol=1e-12, atol=1e-15)
  np.testing.assert_allclose(image.srgb_to_linear(xt).numpy(), G['curve/srgb_to_linear'], rtol=1e-12, atol=1e-15)
  img = torch.from_numpy(G['down/img'])
  np.testing.assert_allclose(image.downsample(img, 2).numpy(), G['down/by2'], rtol=1e-12)
  np.testing.assert_allclose(image.downsample(img, 4).numpy(), G['down/by4'], rtol=1e-12)
  try:
    image.downsample(img, 5)
    raise AssertionError('expected ValueError')
  except ValueError:
    pass
  for i in range(3):   # the recorded outputs went through the stand-in's float32 matmul: 1e-5, the reference test's own bar
    out = image.color_correct(G[f'cc{i}/img'], G[f'cc{i}/ref'])
    np.testing.assert_allclose(out, G[f'cc{i}/out'], atol=2e-5, rtol=1e-5)
  np.testing.assert_allclose(image.mse_to_psnr(torch.from_numpy(G['psnr/mse'])).numpy(), G['psnr/psnr'], rtol=1e-6)


def test_color_correction_undoes_a_ccm_warp():
  """image_test.py:32-58 (structure of the reference's test, own random numbers)."""
  from nerf_hugs_amd.internal import image
  rng = np.random.default_rng(0)
  for _ in range(4):
    im0 = rng.uniform(0.1, 0.9, (64, 64, 3))
    ccm = np.eye(3) + rng.normal(size=(3, 3)) * rng.normal() / 10
    im1 = np.clip((im0.reshape(-1, 3) @ ccm).reshape(im0.shape) + rng.normal() / 10 * im0 ** 2 + rng.normal() / 10, 0, 1)
    np.testing.assert_allclose(image.color_correct(im0, im1), im1, atol=1e-5, rtol=1e-5)


def test_ssim_oracle_against_direct_window_evaluation():
  rng = np.random.default_rng(1)
  a = rng.uniform(size=(19, 23, 3))
  b = np.clip(a + rng.normal(size=a.shape) * 0.1, 0, 1)
  w = I.gaussian_window()
  w2 = np.outer(w, w)
  m = I.ssim_map(a, b)
  assert m.shape == (9, 13, 3)
  for (y, x, c) in [(0, 0, 0), (8, 12, 2), (4, 7, 1)]:
    pa, pb = a[y:y + 11, x:x + 11, c], b[y:y + 11, x:x + 11, c]
    mu0, mu1 = (w2 * pa).sum(), (w2 * pb).sum()
    s00, s11, s01 = (w2 * pa * pa).sum() - mu0 ** 2, (w2 * pb * pb).sum() - mu1 ** 2, (w2 * pa * pb).sum() - mu0 * mu1
    want = (2 * mu0 * mu1 + 1e-4) * (2 * s01 + 9e-4) / ((mu0 ** 2 + mu1 ** 2 + 1e-4) * (s00 + s11 + 9e-4))
    assert abs(m[y, x, c] - want) < 1e-12
  assert abs(I.ssim(a, a) - 1) < 1e-12 and I.ssim(a, b) < 0.99

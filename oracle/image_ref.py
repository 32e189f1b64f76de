"""CPU oracle for the eval metrics.  TEST INFRASTRUCTURE ONLY.

`ssim` restates dm_pix.ssim (dm_pix is a third-party dependency of the reference -- requirements_jax.txt, no
version pinned, not vendored, not installed here): the SSIM of Wang et al. 2004 with an 11-tap Gaussian window
(sigma 1.5), 'valid' separable filtering per channel, k1 = .01, k2 = .03, variances clamped at eps^2 and the
covariance clamped to sqrt(var0 var1), averaged over the map.  PARITY UNPINNED for ssim: the reference holds no
golden SSIM value (tests/image_test.py only round-trips ssim<->dssim); it is cross-checked against an independent
direct-window evaluation in tests/test_oracle_image.py.  The sRGB curves, PSNR and colour correction are pinned by
tests/golden/ref_image.npz (recorded from the reference's image.py) and by the golden tables of the reference's
tests/image_test.py:91-130.
"""
import numpy as np


def gaussian_window(size=11, sigma=1.5):
  z = (np.arange(size) - size // 2) / sigma
  w = np.exp(-.5 * z * z)
  return w / w.sum()


def _valid_filter(x, w):
  """x [H,W,C] -> 'valid' separable correlation with w along H then W."""
  n = len(w)
  h = sum(w[k] * x[k:x.shape[0] - n + 1 + k] for k in range(n))
  return sum(w[k] * h[:, k:h.shape[1] - n + 1 + k] for k in range(n))


def ssim_map(a, b, max_val=1., size=11, sigma=1.5, k1=.01, k2=.03):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  w = gaussian_window(size, sigma)
  mu0, mu1 = _valid_filter(a, w), _valid_filter(b, w)
  mu00, mu11, mu01 = mu0 * mu0, mu1 * mu1, mu0 * mu1
  eps2 = float(np.finfo(np.float32).eps) ** 2
  s00 = np.maximum(eps2, _valid_filter(a * a, w) - mu00)
  s11 = np.maximum(eps2, _valid_filter(b * b, w) - mu11)
  s01 = _valid_filter(a * b, w) - mu01
  s01 = np.sign(s01) * np.minimum(np.sqrt(s00 * s11), np.abs(s01))
  c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
  return ((2 * mu01 + c1) * (2 * s01 + c2)) / ((mu00 + mu11 + c1) * (s00 + s11 + c2))


def ssim(a, b, **kw):
  return float(ssim_map(a, b, **kw).mean())


def psnr(a, b):
  return float(-10. / np.log(10.) * np.log(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).mean()))


def linear_to_srgb(x):
  eps = np.finfo(np.float32).eps
  return np.where(x <= 0.0031308, 323 / 25 * x, (211 * np.maximum(eps, x) ** (5 / 12) - 11) / 200)


def srgb_to_linear(x):
  eps = np.finfo(np.float32).eps
  return np.where(x <= 0.04045, 25 / 323 * x, np.maximum(eps, (200 * x + 11) / 211) ** (12 / 5))

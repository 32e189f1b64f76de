"""Image metrics and colour helpers (reference MipNeRF360/internal/image.py).

`MetricHarness` (image.py:127-141) runs on the device: the squared error and dm_pix's SSIM are HIP kernels
(csrc/hugs_image.hip).  sRGB transfer curves (:48-67) and `downsample` (:70-79) are elementwise tensor code;
`color_correct` (:82-124) keeps the reference's float64 host least squares (`np.linalg.lstsq`, which the
reference itself prefers over the accelerator for stability)."""
import math

import numpy as np
import torch

from .. import _lib as L


def mse_to_psnr(mse):
  """Compute PSNR given an MSE (we assume the maximum pixel value is 1)."""
  return -10. / math.log(10.) * torch.log(mse)


def psnr_to_mse(psnr):
  return torch.exp(-0.1 * math.log(10.) * psnr)


def ssim_to_dssim(ssim):
  return (1 - ssim) / 2


def dssim_to_ssim(dssim):
  return 1 - 2 * dssim


_EPS = float(np.finfo(np.float32).eps)


def linear_to_srgb(linear, eps=None):
  """Assumes `linear` is in [0, 1], see https://en.wikipedia.org/wiki/SRGB (image.py:48-56)."""
  eps = _EPS if eps is None else eps
  srgb0 = 323 / 25 * linear
  srgb1 = (211 * torch.clamp(linear, min=eps) ** (5 / 12) - 11) / 200
  return torch.where(linear <= 0.0031308, srgb0, srgb1)


def srgb_to_linear(srgb, eps=None):
  """image.py:59-67."""
  eps = _EPS if eps is None else eps
  linear0 = 25 / 323 * srgb
  linear1 = torch.clamp((200 * srgb + 11) / 211, min=eps) ** (12 / 5)
  return torch.where(srgb <= 0.04045, linear0, linear1)


def downsample(img, factor):
  """Area downsample img (factor must evenly divide img height and width) (image.py:70-79)."""
  sh = img.shape
  if not (sh[0] % factor == 0 and sh[1] % factor == 0):
    raise ValueError(f'Downsampling factor {factor} does not evenly divide image shape {tuple(sh[:2])}')
  img = img.reshape((sh[0] // factor, factor, sh[1] // factor, factor) + tuple(sh[2:]))
  return img.mean((1, 3))


def color_correct(img, ref, num_iters=5, eps=0.5 / 255):
  """Warp `img` to match the colors in `ref` (image.py:82-124): per channel a quadratic + linear + bias fit on the
  unsaturated pixels, re-solved num_iters times.  numpy float64 on the host, like eval.py:131-136 feeds it."""
  img = np.asarray(img.detach().cpu() if torch.is_tensor(img) else img, np.float64)
  ref = np.asarray(ref.detach().cpu() if torch.is_tensor(ref) else ref, np.float64)
  if img.shape[-1] != ref.shape[-1]:
    raise ValueError(f'img\'s {img.shape[-1]} and ref\'s {ref.shape[-1]} channels must match')
  nc = img.shape[-1]
  x = img.reshape(-1, nc)
  r = ref.reshape(-1, nc)
  ok = lambda z: (z >= eps) & (z <= 1 - eps)
  mask0 = ok(x)
  for _ in range(num_iters):
    cols = [x[:, c:c + 1] * x[:, c:] for c in range(nc)] + [x, np.ones_like(x[:, :1])]
    a = np.concatenate(cols, -1)
    warp = []
    for c in range(nc):
      m = mask0[:, c] & ok(x[:, c]) & ok(r[:, c])
      w = np.linalg.lstsq(np.where(m[:, None], a, 0), np.where(m, r[:, c], 0), rcond=-1)[0]
      assert np.all(np.isfinite(w))
      warp.append(w)
    x = np.clip(a @ np.stack(warp, -1), 0, 1)
  return x.reshape(img.shape)


def _dev_f32(x, device):
  t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
  return t.to(device=device, dtype=torch.float32).contiguous()


def mse(a, b):
  """mean((a - b)^2) as a 0-d device tensor."""
  if not (torch.is_tensor(a) and a.is_cuda):
    raise L.HugsError('image.mse: inputs must be on the GPU (no CPU fallback)')
  if a.shape != b.shape:
    raise ValueError(f'shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}')
  ws = torch.empty(1024, dtype=torch.float32, device=a.device)
  out = torch.empty(1, dtype=torch.float32, device=a.device)
  L.call('hugs_mse', a.numel(), a, b, ws, out)
  return out[0]


def ssim(a, b, max_val=1.0, filter_sigma=1.5, k1=0.01, k2=0.03):
  """dm_pix.ssim(a, b) for [H, W, C] images (filter_size 11), as a 0-d device tensor."""
  if not (torch.is_tensor(a) and a.is_cuda):
    raise L.HugsError('image.ssim: inputs must be on the GPU (no CPU fallback)')
  if a.shape != b.shape or a.dim() != 3:
    raise ValueError(f'ssim takes two [H, W, C] images, got {tuple(a.shape)} and {tuple(b.shape)}')
  h, w, c = a.shape
  nbytes = L.lib().cdll.hugs_ssim_ws_bytes(h, w, c)
  ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=a.device)
  out = torch.empty(1, dtype=torch.float32, device=a.device)
  L.call('hugs_ssim', h, w, c, a, b, max_val, filter_sigma, k1, k2, ws, out)
  return out[0]


class MetricHarness:
  """A helper class for evaluating several error metrics (image.py:127-141)."""

  def __init__(self, device='cuda'):
    self.device = torch.device(device)

  def __call__(self, rgb_pred, rgb_gt, name_fn=lambda s: s):
    a, b = _dev_f32(rgb_pred, self.device), _dev_f32(rgb_gt, self.device)
    return {name_fn('psnr'): float(mse_to_psnr(mse(a, b))), name_fn('ssim'): float(ssim(a, b))}

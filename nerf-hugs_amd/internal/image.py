"""PSNR helpers (reference MipNeRF360/internal/image.py:28-35)."""
import math

import torch


def mse_to_psnr(mse):
  """Compute PSNR given an MSE (we assume the maximum pixel value is 1)."""
  return -10. / math.log(10.) * torch.log(mse)


def psnr_to_mse(psnr):
  return torch.exp(-0.1 * math.log(10.) * psnr)

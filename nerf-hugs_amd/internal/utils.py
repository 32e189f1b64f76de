"""Rays / Batch containers and shard helpers (reference MipNeRF360/internal/utils.py:29-128).

The reference's flax.struct pytrees become plain dataclasses of torch tensors with the same field
names and trailing dims; `shard`/`unshard` keep the reference's [ndev, n/ndev, ...] view semantics
(one process per GPU here, so ndev == 1 inside a process)."""
import dataclasses
from typing import Optional

import torch

_FIELDS = ('pix_coords', 'origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'static_mask', 'near',
           'far', 'embed_idx', 'cam_idx')


@dataclasses.dataclass
class Rays:
  """All tensors must have the same num_dims and first n-1 dims must match (utils.py:44-57)."""
  pix_coords: torch.Tensor
  origins: torch.Tensor
  directions: torch.Tensor
  viewdirs: torch.Tensor
  radii: torch.Tensor
  lossmult: torch.Tensor
  static_mask: torch.Tensor
  near: torch.Tensor
  far: torch.Tensor
  embed_idx: torch.Tensor
  cam_idx: torch.Tensor

  def map(self, fn):
    return Rays(**{k: fn(getattr(self, k)) for k in _FIELDS})

  def to(self, device):
    return self.map(lambda x: x.to(device))

  def flat(self):
    return self.map(lambda x: x.reshape(-1, x.shape[-1]).contiguous())


@dataclasses.dataclass
class Pixels:
  """Integer pixel coordinates + per-ray metadata, the input of camera_utils.cast_ray_batch (utils.py:31-41)."""
  pix_x_int: torch.Tensor
  pix_y_int: torch.Tensor
  lossmult: torch.Tensor
  static_mask: torch.Tensor
  near: torch.Tensor
  far: torch.Tensor
  embed_idx: torch.Tensor
  cam_idx: torch.Tensor


@dataclasses.dataclass
class Batch:
  """Data batch for NeRF training or testing (utils.py:77-81)."""
  rays: Rays
  rgb: Optional[torch.Tensor] = None


def dummy_rays(device='cpu') -> Rays:
  """utils.py:61-74."""
  z = lambda n, dt=torch.float32: torch.zeros((1, n), dtype=dt, device=device)
  return Rays(pix_coords=z(2), origins=z(3), directions=z(3), viewdirs=z(3), radii=z(1), lossmult=z(1),
              static_mask=z(1), near=z(1), far=z(1), embed_idx=z(1, torch.int32), cam_idx=z(1, torch.int32))


def tree_map(fn, x):
  if isinstance(x, Rays):
    return x.map(fn)
  if isinstance(x, Batch):
    return Batch(rays=tree_map(fn, x.rays), rgb=None if x.rgb is None else fn(x.rgb))
  if isinstance(x, dict):
    return {k: tree_map(fn, v) for k, v in x.items()}
  if isinstance(x, (list, tuple)):
    return type(x)(tree_map(fn, v) for v in x)
  return fn(x)


def local_device_count():
  return 1  # one process per GPU


def shard(xs):
  """Split data into shards for multiple devices along the first dimension (utils.py:117-120)."""
  n = local_device_count()
  return tree_map(lambda x: x.reshape((n, -1) + tuple(x.shape[1:])), xs)


def unshard(x, padding=0):
  """Collect the sharded tensor to the shape before sharding (utils.py:123-128)."""
  y = x.reshape([x.shape[0] * x.shape[1]] + list(x.shape[2:]))
  if padding > 0:
    y = y[:-padding]
  return y


def save_img_u8(img, pth):
  """Save an image (probably RGB) in [0, 1] to disk as a uint8 PNG (utils.py:157-162)."""
  import numpy as np
  from PIL import Image
  img = img.detach().cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
  arr = (np.clip(np.nan_to_num(img), 0., 1.) * 255.).astype(np.uint8)
  with open(pth, 'wb') as f:
    Image.fromarray(arr).save(f, 'PNG')


def save_img_f32(depthmap, pth):
  """Save an image (probably a depthmap) to disk as a float32 TIFF (utils.py:165-168)."""
  import numpy as np
  from PIL import Image
  img = depthmap.detach().cpu().numpy() if torch.is_tensor(depthmap) else np.asarray(depthmap)
  with open(pth, 'wb') as f:
    Image.fromarray(np.nan_to_num(img).astype(np.float32)).save(f, 'TIFF')

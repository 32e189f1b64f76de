"""Device-side mirror of the reference's ray generator (MipNeRF360/internal/camera_utils.py).

Same names and argument meaning as the reference (`pixels_to_rays` :503, `cast_ray_batch` :611,
`pixel_coordinates` :403, `ProjectionType` :497); every array is a CUDA tensor and the arithmetic runs in
`hugs_pixels_to_rays` (csrc/hugs_camera.hip).  Where the reference broadcasts per-pixel [.., 3, 3] camera
stacks, this takes the camera tables once ([ncams, 3, 3] / [ncams, 3, 4]) plus a per-pixel `cam_idx`, so a
batch moves 4 bytes per ray instead of 84.
"""
import enum

import torch

from .. import _lib as L
from . import utils


class ProjectionType(enum.Enum):
  PERSPECTIVE = 'perspective'
  FISHEYE = 'fisheye'


_DIST_KEYS = ('k1', 'k2', 'k3', 'k4', 'p1', 'p2')


def distortion_tensor(distortion_params, device):
  """dict(k1.., p1, p2) (camera_utils.py:462-472 keyword defaults = 0) -> [1, 6] fp32 tensor, or None."""
  if distortion_params is None:
    return None
  if isinstance(distortion_params, torch.Tensor):
    return distortion_params.to(device=device, dtype=torch.float32).reshape(-1, 6).contiguous()
  unknown = set(distortion_params) - set(_DIST_KEYS)
  if unknown:
    raise TypeError(f'unexpected distortion parameters {sorted(unknown)}')
  return torch.tensor([[float(distortion_params.get(k, 0.)) for k in _DIST_KEYS]], dtype=torch.float32, device=device)


def pixel_coordinates(width, height, device='cuda'):
  """camera_utils.py:403-407: (x, y) int32 grids of shape [height, width]."""
  x = torch.arange(width, dtype=torch.int32, device=device)
  y = torch.arange(height, dtype=torch.int32, device=device)
  return x[None, :].expand(height, width).contiguous(), y[:, None].expand(height, width).contiguous()


def pixels_to_rays(pix_x_int, pix_y_int, pixtocams, camtoworlds, distortion_params=None, pixtocam_ndc=None,
                   camtype=ProjectionType.PERSPECTIVE, cam_idx=None, widths=None, heights=None, validate=True):
  """Returns (origins, directions, viewdirs, radii[, pix_coords]) of shape SH + [3|1|2] (camera_utils.py:503-607).

  pixtocams [3,3] or [ncams,3,3]; camtoworlds [3,4] or [ncams,3,4]; cam_idx int32 tensor of shape SH (required when
  ncams > 1).  pix_coords is returned when widths/heights ([ncams] int32) are given.  validate=False skips the
  cam_idx range check (a device->host sync); the kernel then reads camera 0 for an out-of-range index."""
  dev = pix_x_int.device
  sh = tuple(pix_x_int.shape)
  n = pix_x_int.numel()
  px = pix_x_int.to(torch.int32).contiguous()
  py = pix_y_int.to(torch.int32).expand(sh).contiguous()
  p2c = pixtocams.to(device=dev, dtype=torch.float32).reshape(-1, 3, 3).contiguous()
  c2w = camtoworlds.to(device=dev, dtype=torch.float32)[..., :3, :4].reshape(-1, 3, 4).contiguous()
  ncams = p2c.shape[0]
  if c2w.shape[0] != ncams:
    raise ValueError(f'{ncams} pixtocams but {c2w.shape[0]} camtoworlds')
  ci = None
  if cam_idx is not None:
    ci = cam_idx.to(torch.int32).expand(sh).contiguous()
    if validate and n and (int(ci.min()) < 0 or int(ci.max()) >= ncams):
      raise IndexError(f'cam_idx out of range for {ncams} cameras')
  dist = distortion_tensor(distortion_params, dev)
  if dist is not None and dist.shape[0] not in (1, ncams):
    raise ValueError('distortion table must have 1 or ncams rows')
  ndc = None if pixtocam_ndc is None else pixtocam_ndc.to(device=dev, dtype=torch.float32).reshape(3, 3).contiguous()
  ct = ProjectionType(camtype) if not isinstance(camtype, ProjectionType) else camtype
  f = lambda c: torch.empty(sh + (c,), dtype=torch.float32, device=dev)
  o, d, v, r = f(3), f(3), f(3), f(1)
  pc = f(2) if widths is not None else None
  w = None if widths is None else widths.to(device=dev, dtype=torch.int32).contiguous()
  h = None if heights is None else heights.to(device=dev, dtype=torch.int32).contiguous()
  L.call('hugs_pixels_to_rays', n, px, py, ci, ncams, p2c, c2w, dist, 0 if dist is None or dist.shape[0] == 1 else 1,
         ndc, 1 if ct == ProjectionType.FISHEYE else 0, w, h, o, d, v, r, pc)
  return (o, d, v, r) if pc is None else (o, d, v, r, pc)


def cast_ray_batch(cameras, pixels, heights, widths, distortion_params, camtype=ProjectionType.PERSPECTIVE):
  """camera_utils.py:611-672.  cameras = (pixtocams, camtoworlds, pixtocam_ndc); `pixels` carries pix_x_int,
  pix_y_int, cam_idx [..., 1] and the per-ray metadata, which is passed through unchanged."""
  pixtocams, camtoworlds, pixtocam_ndc = cameras
  cam_idx = pixels.cam_idx[..., 0]
  o, d, v, r, pc = pixels_to_rays(pixels.pix_x_int, pixels.pix_y_int, pixtocams, camtoworlds, distortion_params,
                                  pixtocam_ndc, camtype, cam_idx=cam_idx, widths=widths, heights=heights)
  return utils.Rays(pix_coords=pc, origins=o, directions=d, viewdirs=v, radii=r, lossmult=pixels.lossmult,
                    static_mask=pixels.static_mask, near=pixels.near, far=pixels.far, embed_idx=pixels.embed_idx,
                    cam_idx=pixels.cam_idx)

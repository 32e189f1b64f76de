"""CPU oracle for device-side ray generation.  TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's host ray generator, pinned by tests/test_oracle_cameras.py to
tests/golden/ref_cameras.npz (vectors recorded from the reference's own camera_utils.py with xnp=np).
Follows /root/reference/MipNeRF360/internal/camera_utils.py:
  :32-100  convert_to_ndc          -> to_ndc
  :410-458 residual and Jacobian,
  :462-495 10-step Newton undistort -> undistort
  :503-607 pixels_to_rays           -> pixels_to_rays
  :611-672 cast_ray_batch           -> pix_coords
and datasets.py:494-529 (_next_train patch sampling) -> sample_patches.
"""
import numpy as np


def undistort(xd, yd, k, iters=10, eps=1e-9):
  """k = (k1, k2, k3, k4, p1, p2).  Newton on the 2x2 system; steps with |det| <= eps are dropped."""
  k1, k2, k3, k4, p1, p2 = [xd.dtype.type(v) for v in k]
  x, y = xd.copy(), yd.copy()
  for _ in range(iters):
    r = x * x + y * y
    d = 1 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
    fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
    fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
    d_r = k1 + r * (2 * k2 + r * (3 * k3 + r * 4 * k4))
    d_x, d_y = 2 * x * d_r, 2 * y * d_r
    fx_x = d + d_x * x + 2 * p1 * y + 6 * p2 * x
    fx_y = d_y * x + 2 * p1 * x + 2 * p2 * y
    fy_x = d_x * y + 2 * p2 * y + 2 * p1 * x
    fy_y = d + d_y * y + 2 * p2 * x + 6 * p1 * y
    den = fy_x * fx_y - fx_x * fy_y
    ok = np.abs(den) > eps
    safe = np.where(ok, den, 1)
    x = x + np.where(ok, (fx * fy_y - fy * fx_y) / safe, 0)
    y = y + np.where(ok, (fy * fx_x - fx * fy_x) / safe, 0)
  return x, y


def to_ndc(o, d, pixtocam, near=1.):
  t = -(near + o[..., 2]) / d[..., 2]
  o = o + t[..., None] * d
  xm = o.dtype.type(pixtocam.dtype.type(1.) / pixtocam[0, 2])
  ym = o.dtype.type(pixtocam.dtype.type(1.) / pixtocam[1, 2])
  o_ndc = np.stack([xm * o[..., 0] / o[..., 2], ym * o[..., 1] / o[..., 2], -np.ones_like(t)], -1)
  inf_ndc = np.stack([xm * d[..., 0] / d[..., 2], ym * d[..., 1] / d[..., 2], np.ones_like(t)], -1)
  return o_ndc, inf_ndc - o_ndc


def pixels_to_rays(pix_x, pix_y, pixtocams, camtoworlds, dist=None, ndc=None, fisheye=False, dtype=np.float64):
  """pix_x/pix_y int [...]; pixtocams [...,3,3]; camtoworlds [...,3,4] (already indexed per pixel)."""
  f = dtype
  p2c, c2w = pixtocams.astype(f), camtoworlds.astype(f)
  outs = []
  for ox, oy in ((0, 0), (1, 0), (0, 1)):
    p = np.stack([pix_x.astype(f) + ox + f(.5), pix_y.astype(f) + oy + f(.5), np.ones(pix_x.shape, f)], -1)
    c = np.einsum('...ij,...j->...i', p2c, p)
    if dist is not None:
      x, y = undistort(c[..., 0], c[..., 1], dist)
      c = np.stack([x, y, np.ones_like(x)], -1)
    if fisheye:
      th = np.minimum(f(np.pi), np.sqrt(c[..., 0] ** 2 + c[..., 1] ** 2))
      s = np.sin(th) / th
      c = np.stack([c[..., 0] * s, c[..., 1] * s, np.cos(th)], -1)
    c = c * np.array([1, -1, -1], f)
    outs.append(np.einsum('...ij,...j->...i', c2w[..., :3, :3], c))
  d, dx, dy = outs
  o = np.broadcast_to(c2w[..., :3, 3], d.shape)
  v = d / np.linalg.norm(d, axis=-1, keepdims=True)
  if ndc is None:
    nx, ny = np.linalg.norm(dx - d, axis=-1), np.linalg.norm(dy - d, axis=-1)
  else:
    # (the reference takes 1/pixtocam[i,2] in the matrix's own binary32, then promotes: keep `ndc` as given)
    ox_, _ = to_ndc(o, dx, ndc)
    oy_, _ = to_ndc(o, dy, ndc)
    o, d = to_ndc(o, d, ndc)
    nx, ny = np.linalg.norm(ox_ - o, axis=-1), np.linalg.norm(oy_ - o, axis=-1)
  radii = (f(.5) * (nx + ny))[..., None] * 2 / np.sqrt(f(12))
  return o, d, v, radii


def pix_coords(pix_x, pix_y, widths, heights, cam_idx):
  w, h = widths[cam_idx], heights[cam_idx]
  return np.stack([(pix_x.astype(np.float32) + .5) / w, (pix_y.astype(np.float32) + .5) / h], -1)


def sample_patches(rs, n_examples, heights, widths, batch_size, patch_size=1, dilation=1, images_per_batch=1,
                   half_image=False):
  """The reference's np.random call sequence (datasets.py:494-529): per image one randint for the camera,
  then the patch x origins, then the y origins.  Returns cam_idx [I], pix_x / pix_y [I, P, ps, ps]."""
  p = (batch_size // images_per_batch) // patch_size ** 2
  upper = (patch_size - 1) * dilation
  dx, dy = np.meshgrid(np.arange(patch_size), np.arange(patch_size), indexing='xy')
  cams, xs, ys = [], [], []
  for _ in range(images_per_batch):
    c = rs.randint(0, n_examples)
    h, w = heights[c], widths[c]
    if half_image:
      w = w // 2
    x = rs.randint(0, w - upper, (p, 1, 1))
    y = rs.randint(0, h - upper, (p, 1, 1))
    cams.append(c)
    xs.append(x + dx * dilation)
    ys.append(y + dy * dilation)
  return np.array(cams), np.stack(xs), np.stack(ys)

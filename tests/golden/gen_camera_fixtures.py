"""Golden vectors for device-side ray generation, made by EXECUTING THE
REFERENCE'S OWN `camera_utils.py` with `xnp=np` (its host path, datasets.py:483).

Runs only in the build container (needs /root/reference).  `camera_utils.py` imports
`internal.configs` / `internal.utils` (absl, flax, gin: not installed); neither is touched by the
functions recorded here except for the `utils.Rays` / `utils.Pixels` containers, so two empty
placeholder modules carrying plain dataclasses with the same field names are pre-registered in
sys.modules.  Only DATA is committed.

    python tests/golden/gen_camera_fixtures.py      # rewrites tests/golden/ref_cameras.npz

Reference entry points exercised (under /root/reference/MipNeRF360/internal):
  camera_utils.py:32 convert_to_ndc, :403 pixel_coordinates, :462 _radial_and_tangential_undistort,
  :503 pixels_to_rays, :611 cast_ray_batch
"""
import dataclasses
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/MipNeRF360'


def _placeholders():
  utils = types.ModuleType('internal.utils')

  @dataclasses.dataclass
  class Pixels:
    pix_x_int: object
    pix_y_int: object
    lossmult: object
    static_mask: object
    near: object
    far: object
    embed_idx: object
    cam_idx: object

  @dataclasses.dataclass
  class Rays:
    pix_coords: object
    origins: object
    directions: object
    viewdirs: object
    radii: object
    lossmult: object
    static_mask: object
    near: object
    far: object
    embed_idx: object
    cam_idx: object

  utils.Pixels, utils.Rays = Pixels, Rays
  cfg = types.ModuleType('internal.configs')
  cfg.Config = object
  sys.modules['internal.utils'] = utils
  sys.modules['internal.configs'] = cfg
  return utils


def main():
  if not os.path.isdir(REF):
    raise SystemExit('needs the reference checkout at ' + REF)
  sys.path.insert(0, HERE)
  import _jax_standin
  _jax_standin.install()
  sys.path.insert(0, REF)
  utils = _placeholders()
  from internal import camera_utils as cu

  rng = np.random.default_rng(503)
  out = {}

  def cams(n, w, h, focal_rng=(0.9, 1.6)):
    p2c, c2w = [], []
    for _ in range(n):
      f = rng.uniform(*focal_rng) * w
      k = np.array([[f, 0, w / 2 + rng.uniform(-3, 3)], [0, f * rng.uniform(0.98, 1.02), h / 2 + rng.uniform(-3, 3)],
                    [0, 0, 1.]])
      p2c.append(np.linalg.inv(k))
      q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
      if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
      c2w.append(np.concatenate([q, rng.normal(size=(3, 1))], 1))
    return np.stack(p2c).astype(np.float32), np.stack(c2w).astype(np.float32)

  def record(tag, n, w, h, dist=None, ndc=False, camtype='perspective'):
    ncam = 5
    p2c, c2w = cams(ncam, w, h)
    px = rng.integers(0, w, (n,)).astype(np.int32)
    py = rng.integers(0, h, (n,)).astype(np.int32)
    ci = rng.integers(0, ncam, (n,)).astype(np.int32)
    if ndc:   # forward-facing: rotation near identity so that dz < 0
      for i in range(ncam):
        a = rng.normal(size=3) * 0.05
        r = np.eye(3) + np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        q, _ = np.linalg.qr(r)
        q = q * np.sign(np.diag(q))[None, :]
        c2w[i, :, :3] = q
      ndc_m = p2c[0].copy()
    else:
      ndc_m = None
    ct = cu.ProjectionType.FISHEYE if camtype == 'fisheye' else cu.ProjectionType.PERSPECTIVE
    o, d, v, r = cu.pixels_to_rays(px, py, p2c[ci], c2w[ci], distortion_params=dist, pixtocam_ndc=ndc_m, camtype=ct,
                                   xnp=np)
    out[tag + '/pix_x'], out[tag + '/pix_y'], out[tag + '/cam_idx'] = px, py, ci
    out[tag + '/pixtocams'], out[tag + '/camtoworlds'] = p2c, c2w
    out[tag + '/wh'] = np.array([w, h], np.int32)
    if dist is not None:
      out[tag + '/dist'] = np.array([dist.get(k, 0.) for k in ('k1', 'k2', 'k3', 'k4', 'p1', 'p2')], np.float64)
    if ndc_m is not None:
      out[tag + '/ndc'] = ndc_m
    out[tag + '/camtype'] = np.array(1 if camtype == 'fisheye' else 0, np.int32)
    for k, a in (('origins', o), ('directions', d), ('viewdirs', v), ('radii', r)):
      out[tag + '/' + k] = np.asarray(a, np.float64)

  record('persp', 64, 800, 600)
  record('dist', 64, 1297, 840, dist=dict(k1=-0.12, k2=0.03, p1=1e-3, p2=-5e-4))
  record('dist_k34', 32, 640, 480, dist=dict(k1=0.08, k2=-0.02, k3=0.004, k4=-0.001, p1=0., p2=0.))
  record('fisheye', 64, 1000, 1000, dist=dict(k1=-0.02, k2=0.003, k3=0., k4=0.), camtype='fisheye')
  record('ndc', 64, 1008, 756, ndc=True)

  # cast_ray_batch: stacked cameras indexed by cam_idx, pix_coords normalisation by per-image size
  p2c, c2w = cams(3, 320, 240)
  heights = np.array([240, 240, 200], np.int32)
  widths = np.array([320, 300, 320], np.int32)
  px = rng.integers(0, 300, (4, 2, 2)).astype(np.int32)
  py = rng.integers(0, 200, (4, 2, 2)).astype(np.int32)
  ci = rng.integers(0, 3, (4, 1, 1)).astype(np.int32)
  ci_b = np.broadcast_to(ci, px.shape)[..., None]
  one = np.ones(px.shape + (1,), np.float32)
  pixels = utils.Pixels(px, py, one, one, one * 0.1, one * 5., ci_b, ci_b)
  rays = cu.cast_ray_batch((p2c, c2w, None), pixels, heights, widths, None, cu.ProjectionType.PERSPECTIVE, xnp=np)
  out['crb/pix_x'], out['crb/pix_y'], out['crb/cam_idx'] = px, py, np.ascontiguousarray(ci_b)
  out['crb/pixtocams'], out['crb/camtoworlds'] = p2c, c2w
  out['crb/heights'], out['crb/widths'] = heights, widths
  for k in ('pix_coords', 'origins', 'directions', 'viewdirs', 'radii'):
    out['crb/' + k] = np.asarray(getattr(rays, k), np.float64)

  gx, gy = cu.pixel_coordinates(7, 5)
  out['pixel_coordinates_7x5/x'], out['pixel_coordinates_7x5/y'] = gx.astype(np.int32), gy.astype(np.int32)

  path = os.path.join(HERE, 'ref_cameras.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, len(out), 'arrays', os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()

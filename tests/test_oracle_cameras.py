"""The camera oracle (oracle/camera_ref.py) against vectors recorded from the reference's camera_utils.py."""
import os

import numpy as np
import pytest

from oracle import camera_ref as C

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_cameras.npz'))


def _case(tag):
  return {k.split('/', 1)[1]: G[k] for k in G.files if k.startswith(tag + '/')}


@pytest.mark.parametrize('tag', ['persp', 'dist', 'dist_k34', 'fisheye', 'ndc'])
def test_pixels_to_rays_matches_reference(tag):
  c = _case(tag)
  ci = c['cam_idx']
  o, d, v, r = C.pixels_to_rays(c['pix_x'], c['pix_y'], c['pixtocams'][ci], c['camtoworlds'][ci],
                                dist=c.get('dist'), ndc=c.get('ndc'), fisheye=bool(c['camtype']))
  for name, a in (('origins', o), ('directions', d), ('viewdirs', v), ('radii', r)):
    np.testing.assert_allclose(a, c[name], rtol=1e-9, atol=1e-12, err_msg=tag + ' ' + name)


def test_fp32_oracle_within_tolerance():
  # the device kernel works in binary32; the reference's host path promotes to float64
  for tag in ['persp', 'dist', 'fisheye', 'ndc']:
    c = _case(tag)
    ci = c['cam_idx']
    out = C.pixels_to_rays(c['pix_x'], c['pix_y'], c['pixtocams'][ci], c['camtoworlds'][ci], dist=c.get('dist'),
                           ndc=c.get('ndc'), fisheye=bool(c['camtype']), dtype=np.float32)
    for name, a in zip(('origins', 'directions', 'viewdirs', 'radii'), out):
      scale = np.abs(c[name]).max()
      assert np.abs(a - c[name]).max() <= (2e-3 if name == 'radii' else 1e-5) * scale, (tag, name)


def test_cast_ray_batch_pix_coords():
  c = _case('crb')
  ci = c['cam_idx'][..., 0]
  np.testing.assert_allclose(C.pix_coords(c['pix_x'], c['pix_y'], c['widths'], c['heights'], ci), c['pix_coords'],
                             rtol=1e-7)
  o, d, v, r = C.pixels_to_rays(c['pix_x'], c['pix_y'], c['pixtocams'][ci], c['camtoworlds'][ci])
  np.testing.assert_allclose(d, c['directions'], rtol=1e-9, atol=1e-12)
  np.testing.assert_allclose(r, c['radii'], rtol=1e-9)


def test_pixel_coordinates_layout():
  x, y = np.meshgrid(np.arange(7), np.arange(5), indexing='xy')
  assert (x == G['pixel_coordinates_7x5/x']).all() and (y == G['pixel_coordinates_7x5/y']).all()


def test_sample_patches_follows_numpy_stream():
  heights, widths = np.array([60, 50]), np.array([80, 70])
  a = C.sample_patches(np.random.RandomState(7), 2, heights, widths, 64, patch_size=4, dilation=2, images_per_batch=2)
  rs = np.random.RandomState(7)
  for i in range(2):
    c = rs.randint(0, 2)
    x = rs.randint(0, widths[c] - 6, (2, 1, 1))
    y = rs.randint(0, heights[c] - 6, (2, 1, 1))
    assert a[0][i] == c and (a[1][i][:, 0, 0] == x[:, 0, 0]).all() and (a[2][i][:, 0, 0] == y[:, 0, 0]).all()
    assert (a[1][i][:, 0, :] - x[:, 0, :] == np.arange(4) * 2).all()
    assert (a[2][i][:, :, 0] - y[:, :, 0] == np.arange(4) * 2).all()

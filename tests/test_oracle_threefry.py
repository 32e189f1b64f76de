"""Pins oracle/threefry_ref.py: Random123 known-answer vectors for Threefry-2x32-20, and the golden numbers of the
reference's own MipNeRF360/tests/datasets_test.py:66-104, which are jax.random draws from PRNGKey(0) (uniform ->
image pixels; split + normal -> camera poses) pushed through the reference's ray generator."""
import numpy as np

from oracle import camera_ref as C
from oracle import threefry_ref as T

# tests/datasets_test.py:70-104 (data held by the reference's test: expected outputs, not code)
RGB_GT = np.array([
    0.5289556, 0.28869557, 0.24527192, 0.12083626, 0.8904066, 0.6259936, 0.57573485, 0.09355974, 0.8017353, 0.538651,
    0.4998169, 0.42061496, 0.5591258, 0.00577283, 0.6804651, 0.9139203, 0.00444758, 0.96962905, 0.52956843, 0.38282406,
    0.28777933, 0.6640035, 0.39736128, 0.99495006, 0.13100398, 0.7597165, 0.8532667, 0.67468107, 0.6804743, 0.26873016,
    0.60699487, 0.5722265, 0.44482303, 0.6511061, 0.54807067, 0.09894073])
ORIGIN_GT = np.array([-0.20050469, -0.6451472, -0.8818224])
DIRS_GT = np.array([
    0.24370372, 0.89296186, -0.5227117, 0.05601424, 0.8468699, -0.57417226, -0.13167524, 0.8007779, -0.62563276,
    -0.31936473, 0.75468594, -0.67709327, 0.17780769, 0.96766925, -0.34928587, -0.0098818, 0.9215773, -0.4007464,
    -0.19757128, 0.87548524, -0.4522069, -0.38526076, 0.82939327, -0.5036674, 0.11191163, 1.0423766, -0.17586003,
    -0.07577785, 0.9962846, -0.22732055, -0.26346734, 0.95019263, -0.2787811, -0.45115682, 0.90410066, -0.3302416])


def test_random123_known_answers():
  kat = [((0, 0), (0, 0), (0x6b200159, 0x99ba4efe)),
         ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
         ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))]
  for key, ctr, want in kat:
    y0, y1 = T.threefry2x32(key, np.array([ctr[0]], np.uint32), np.array([ctr[1]], np.uint32))
    assert (int(y0[0]), int(y1[0])) == want


def dummy_dataset():
  """The DummyDataset of datasets_test.py:27-50 rebuilt from the oracle PRNG."""
  rng = T.prng_key(0)
  key, rng = T.split(rng)
  images = T.uniform(key, (2, 3, 4, 3))
  key, rng = T.split(rng)
  c2w = []
  for k in T.split(key, 2):
    look, up, pos = T.normal(k, (3, 3)).astype(np.float64)
    nrm = lambda v: v / np.linalg.norm(v)
    v2 = nrm(look); v0 = nrm(np.cross(up, v2)); v1 = nrm(np.cross(v2, v0))
    c2w.append(np.stack([v0, v1, v2, pos], 1))
  pixtocam = np.linalg.inv(np.array([[5., 0, 2.], [0, 5., 1.5], [0, 0, 1.]]))
  return images, np.stack(c2w), pixtocam


def test_reference_dataset_golden():
  images, c2w, pixtocam = dummy_dataset()
  np.testing.assert_allclose(images[0].ravel(), RGB_GT, atol=1e-7, rtol=0)          # uniform: exact to print precision
  x, y = np.meshgrid(np.arange(4), np.arange(3), indexing='xy')
  o, d, v, r = C.pixels_to_rays(x, y, pixtocam[None, None], c2w[0][None, None])
  np.testing.assert_allclose(o.ravel(), np.tile(ORIGIN_GT, 12), atol=1e-4, rtol=1e-4)   # the reference's own tolerance
  np.testing.assert_allclose(d.ravel(), DIRS_GT, atol=1e-4, rtol=1e-4)


def test_split_layout():
  k = T.split(T.prng_key(0))
  assert k.tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
  # odd draw counts pad the counters with one zero
  a = T.random_bits(T.prng_key(7), (5,))
  y0, y1 = T.threefry2x32((0, 7), np.array([0, 1, 2], np.uint32), np.array([3, 4, 0], np.uint32))
  assert a.tolist() == [int(y0[0]), int(y0[1]), int(y0[2]), int(y1[0]), int(y1[1])]

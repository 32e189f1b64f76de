"""Tiny on-disk Kubric / Phototourism / Distractor scenes + what the REFERENCE'S OWN loaders make of them.

Writes tests/golden/loader_scenes/{kubric_mini, brandenburg_gate, distractor_mini}/ (a few KB of PNG / JSON / TSV /
COLMAP .bin files, committed as data) and tests/golden/ref_loaders.npz: the arrays that
MipNeRF360/internal/datasets.py `Kubric / Phototourism / Distractor._load_renderings` (:984-1394, executed unmodified
from /root/reference) fill from those files -- images (alpha on white), static masks (resized), per-image near / far,
camera-to-world poses after recentring / PCA / scaling, inverse intrinsics, distortion parameters, embedding indices.
Build container only.  Stand-ins (absent third parties): `cv2.resize` = bilinear with half-pixel centres via
scipy.ndimage.map_coordinates (an implementation independent of the product's); `pycolmap.SceneManager` = a parser
of the .bin files this script wrote itself (COLMAP's published layout) -- the file PARSING is therefore not pinned,
everything the reference does after it is.

    python tests/golden/gen_loader_fixtures.py
"""
import json
import os
import struct
import sys
import types
import collections

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/MipNeRF360'
OUT = os.path.join(HERE, 'loader_scenes')


def save_png(path, arr):
  os.makedirs(os.path.dirname(path), exist_ok=True)
  Image.fromarray(arr).save(path, 'PNG')


def qvec_from_R(R):
  w = np.sqrt(max(0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
  x = (R[2, 1] - R[1, 2]) / (4 * w); y = (R[0, 2] - R[2, 0]) / (4 * w); z = (R[1, 0] - R[0, 1]) / (4 * w)
  return np.array([w, x, y, z])


def rand_rot(rng, amp):
  a = rng.normal(size=3) * amp
  th = np.linalg.norm(a)
  k = a / th
  K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
  return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def write_colmap(d, cams, imgs, pts):
  os.makedirs(d, exist_ok=True)
  with open(os.path.join(d, 'cameras.bin'), 'wb') as f:
    f.write(struct.pack('<Q', len(cams)))
    for cid, (model, w, h, params) in cams.items():
      f.write(struct.pack('<iiQQ', cid, model, w, h) + struct.pack('<' + 'd' * len(params), *params))
  with open(os.path.join(d, 'images.bin'), 'wb') as f:
    f.write(struct.pack('<Q', len(imgs)))
    for iid, (q, t, cid, name) in imgs.items():
      f.write(struct.pack('<i', iid) + struct.pack('<7d', *q, *t) + struct.pack('<i', cid) + name.encode() + b'\x00')
      f.write(struct.pack('<Q', 2) + struct.pack('<ddq', 1.5, 2.5, 7) + struct.pack('<ddq', 3.0, 1.0, -1))
  with open(os.path.join(d, 'points3D.bin'), 'wb') as f:
    f.write(struct.pack('<Q', len(pts)))
    for i, p in enumerate(pts):
      f.write(struct.pack('<Q', i + 1) + struct.pack('<3d', *p) + bytes([10, 20, 30]) + struct.pack('<d', 0.5))
      f.write(struct.pack('<Q', 1) + struct.pack('<ii', 1, 0))


def make_scenes():
  rng = np.random.default_rng(5)
  # ---- Kubric -------------------------------------------------------------------------------------------
  k = os.path.join(OUT, 'kubric_mini')
  os.makedirs(os.path.join(k, 'freeze-test'), exist_ok=True)
  json.dump(dict(center=[0.1, -0.2, 0.05], scale=0.35, near=0.4, far=2.5), open(os.path.join(k, 'scene_gt.json'), 'w'))
  json.dump(dict(train_ids=[0, 1, 2]), open(os.path.join(k, 'dataset.json'), 'w'))
  json.dump(dict(val_ids=[0, 1]), open(os.path.join(k, 'freeze-test/dataset.json'), 'w'))
  for sub, ids, rgbdir, camdir, mdir in ((k, [0, 1, 2], 'rgb/1x', 'camera-gt', 'static_masks'),
                                         (os.path.join(k, 'freeze-test'), [0, 1], 'static-rgb/1x', 'camera-gt', 'static_masks')):
    for i in ids:
      H, W = 10, 12
      rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
      if i == 1:
        rgba = rgba[..., :3]                       # an RGB image (no alpha)
      save_png(os.path.join(sub, rgbdir, f'{i}.png'), rgba)
      os.makedirs(os.path.join(sub, camdir), exist_ok=True)
      json.dump(dict(orientation=rand_rot(rng, 0.8).tolist(), position=(rng.normal(size=3) * 2).tolist(),
                     focal_length=float(14 + i), principal_point=[5.5 + 0.1 * i, 4.5], skew=0.01 * i,
                     pixel_aspect_ratio=1.0 + 0.02 * i, radial_distortion=[0.01, -0.002 * i, 0.0005],
                     tangential_distortion=[0.001, -0.0007 * i]), open(os.path.join(sub, camdir, f'{i}.json'), 'w'))
      if i != 2:                                   # image 2 has no mask file -> ones
        mshape = (H, W) if i == 0 else (5, 6)       # image 1's mask needs a resize
        m = (rng.uniform(size=mshape) < 0.6).astype(np.uint8) * 255
        save_png(os.path.join(sub, mdir, f'{i}.png'), np.stack([m] * 3, -1))      # the datasets' masks are RGB PNGs
  # ---- COLMAP scenes ---------------------------------------------------------------------------------------
  def colmap_scene(root, sparse, names, models):
    cams, imgs = {}, collections.OrderedDict()
    pts = rng.normal(size=(200, 3)) * 1.5 + np.array([0, 0, 6.0])
    for j, (name, model) in enumerate(zip(names, models)):
      W, H = 12 + 2 * (j % 2), 10
      params = {0: [15.0, W / 2, H / 2], 1: [15.0, 15.5, W / 2, H / 2], 2: [15.0, W / 2, H / 2, 0.02],
                3: [15.0, W / 2, H / 2, 0.02, -0.001], 4: [15.0, 15.2, W / 2, H / 2, 0.02, -0.001, 0.0005, -0.0003],
                5: [15.0, 15.2, W / 2, H / 2, 0.01, 0.002, -0.001, 0.0002]}[model]
      cams[j + 1] = (model, W, H, params)
      R = rand_rot(rng, 0.25)
      c = rng.normal(size=3) * 0.8                   # camera centre
      imgs[10 + j] = (qvec_from_R(R), -R @ c, j + 1, name)
    write_colmap(os.path.join(root, sparse), cams, imgs, pts)
    return cams
  p = os.path.join(OUT, 'brandenburg_gate')
  names = [f'img{j}.png' for j in range(5)]
  cams = colmap_scene(p, 'dense/sparse', names, [2, 0, 1, 3, 4])
  with open(os.path.join(p, 'brandenburg.tsv'), 'w') as f:
    f.write('filename\tid\tsplit\tdataset\n')
    for j, n in enumerate([names[3], names[0], names[4], names[1], names[2]]):
      f.write(f'{n}\t{j}\t{"test" if j == 3 else "train"}\tbrandenburg\n')
  for j, n in enumerate(names):
    _, W, H, _ = cams[j + 1]
    save_png(os.path.join(p, 'dense/images', n), rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
    if j != 1:
      save_png(os.path.join(p, 'dense/static_masks', n), np.stack([(rng.uniform(size=(H // 2, W // 2) if j == 0 else (H, W)) < 0.5).astype(np.uint8) * 255] * 3, -1))
  d = os.path.join(OUT, 'distractor_mini')
  names = [f'{j:03d}.png' for j in range(5)]
  cams = colmap_scene(d, '0/sparse/0', names, [4, 4, 5, 2, 1])
  json.dump(dict(train=[names[1], names[2], names[4]], test=[names[0], names[3]]), open(os.path.join(d, '0/data_split.json'), 'w'))
  for j, n in enumerate(names):
    _, W, H, _ = cams[j + 1]
    save_png(os.path.join(d, '0/images', n), rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
    if j != 4:
      save_png(os.path.join(d, '0/static_masks', n), np.stack([(rng.uniform(size=(H, W)) < 0.5).astype(np.uint8) * 255] * 3, -1))


def main():
  if not os.path.isdir(REF):
    raise SystemExit('needs the reference checkout at ' + REF)
  make_scenes()
  sys.path.insert(0, HERE)
  import _flax_standin as F
  F.install({})
  for name in ('internal.camera_utils', 'internal.datasets'):
    sys.modules.pop(name, None)
  import scipy.ndimage

  def cv2_resize(img, dsize):                        # INTER_LINEAR: half-pixel centres, border replicated
    w, h = dsize
    H, W = img.shape[:2]
    yy = (np.arange(h) + 0.5) * (H / h) - 0.5
    xx = (np.arange(w) + 0.5) * (W / w) - 0.5
    g = np.meshgrid(yy, xx, indexing='ij')
    f = lambda a: scipy.ndimage.map_coordinates(a, g, order=1, mode='nearest')
    return f(img) if img.ndim == 2 else np.stack([f(img[..., c]) for c in range(img.shape[-1])], -1)

  sys.modules['cv2'].resize = cv2_resize
  # pycolmap.SceneManager stand-in over the files written above
  sys.path.insert(0, os.path.dirname(HERE) + '/..')
  pyc = sys.modules['pycolmap']

  class SceneManager:
    def __init__(self, d):
      self.d = d

    def load_cameras(self):
      self.cameras = {}
      with open(os.path.join(self.d, 'cameras.bin'), 'rb') as f:
        n = struct.unpack('<Q', f.read(8))[0]
        for _ in range(n):
          cid, model, w, h = struct.unpack('<iiQQ', f.read(24))
          npar = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8, 5: 8}[model]
          p = struct.unpack('<' + 'd' * npar, f.read(8 * npar))
          c = types.SimpleNamespace(camera_type=model)
          if model in (0, 2, 3):
            c.fx = c.fy = p[0]; c.cx, c.cy = p[1], p[2]; rest = p[3:]
          else:
            c.fx, c.fy, c.cx, c.cy = p[:4]; rest = p[4:]
          if model in (2, 3):
            c.k1 = rest[0]; c.k2 = rest[1] if model == 3 else 0.
          if model == 4:
            c.k1, c.k2, c.p1, c.p2 = rest
          if model == 5:
            c.k1, c.k2, c.k3, c.k4 = rest
          self.cameras[cid] = c

    def load_images(self):
      self.images = collections.OrderedDict()
      with open(os.path.join(self.d, 'images.bin'), 'rb') as f:
        n = struct.unpack('<Q', f.read(8))[0]
        for _ in range(n):
          iid = struct.unpack('<i', f.read(4))[0]
          q = np.array(struct.unpack('<4d', f.read(32))); t = np.array(struct.unpack('<3d', f.read(24)))
          cid = struct.unpack('<i', f.read(4))[0]
          name = b''
          while (c := f.read(1)) != b'\x00':
            name += c
          m = struct.unpack('<Q', f.read(8))[0]; f.read(24 * m)
          w, x, y, z = q
          R = np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                        [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                        [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])
          self.images[iid] = types.SimpleNamespace(R=(lambda R=R: R), tvec=t, camera_id=cid, name=name.decode())

    def load_points3D(self):
      pts = []
      with open(os.path.join(self.d, 'points3D.bin'), 'rb') as f:
        n = struct.unpack('<Q', f.read(8))[0]
        for _ in range(n):
          f.read(8); pts.append(struct.unpack('<3d', f.read(24))); f.read(11)
          tl = struct.unpack('<Q', f.read(8))[0]; f.read(8 * tl)
      self.points3D = np.array(pts)

  pyc.SceneManager = SceneManager
  sys.path.insert(0, REF)
  from internal import configs, datasets, utils
  out = {}

  def run(cls, tag, data_dir, split, **cfg):
    ds = object.__new__(cls)                         # no producer thread: only the file decoding is exercised
    ds.split, ds.data_dir = split, data_dir
    config = configs.Config(**cfg)
    ds.near, ds.far = config.near, config.far
    ds._load_renderings(config)
    for k in ('images', 'static_masks', 'nears', 'fars'):
      for i, a in enumerate(getattr(ds, k)):
        out[f'{tag}/{k}/{i}'] = np.asarray(a, np.float64)
    for k in ('heights', 'widths', 'focals', 'embed_idxs', 'camtoworlds', 'pixtocams'):
      out[f'{tag}/{k}'] = np.asarray(getattr(ds, k), np.float64)
    keys = ('k1', 'k2', 'k3', 'k4', 'p1', 'p2')
    out[f'{tag}/dist'] = np.array([[np.nan] * 6 if d is None else [float(d.get(k, 0.)) for k in keys] for d in ds.distortion_params])
    out[f'{tag}/fisheye'] = np.array([int(c == datasets.camera_utils.ProjectionType.FISHEYE) for c in ds.camtypes])

  S = utils.DataSplit
  run(datasets.Kubric, 'kubric_train', os.path.join(OUT, 'kubric_mini'), S.TRAIN)
  run(datasets.Kubric, 'kubric_test', os.path.join(OUT, 'kubric_mini'), S.TEST)
  run(datasets.Phototourism, 'photo_train', os.path.join(OUT, 'brandenburg_gate'), S.TRAIN)
  run(datasets.Phototourism, 'photo_test_f2', os.path.join(OUT, 'brandenburg_gate'), S.TEST, factor=2)
  run(datasets.Distractor, 'distractor_train', os.path.join(OUT, 'distractor_mini'), S.TRAIN, far=1e6)
  run(datasets.Distractor, 'distractor_test', os.path.join(OUT, 'distractor_mini'), S.TEST, far=1e6)
  np.savez_compressed(os.path.join(HERE, 'ref_loaders.npz'), **out)
  print('wrote ref_loaders.npz', len(out), 'arrays')


if __name__ == '__main__':
  main()

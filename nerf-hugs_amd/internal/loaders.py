"""File-format loaders of the three HuGS datasets (MipNeRF360/internal/datasets.py: Kubric :984-1113,
Phototourism :1123-1256, Distractor :1259-1394) on top of the device-resident `Dataset`: the host side decodes the
files exactly as the reference does -- alpha composited on white, static masks read from `static_mask_dir_name`
and resized to the image, per-image near / far (scene_gt.json for Kubric, percentiles of the SfM points in the
camera frame for Phototourism / Distractor), pose recentring / PCA alignment / scaling, train-test splits, embedding
indices -- and `_upload()` moves everything into HBM once.  Batches are then assembled by HIP kernels.

cv2 / pycolmap are absent from this image: `resize_linear` restates cv2.resize's INTER_LINEAR (half-pixel centres,
edge clamp) and `colmap.process` the COLMAP reader."""
import json
import os
from pathlib import Path

import numpy as np
from PIL import Image

from . import colmap
from .camera_utils import ProjectionType
from .datasets import Dataset


def load_img(pth):
  """utils.py:131-135."""
  with open(pth, 'rb') as f:
    return np.array(Image.open(f), dtype=np.float32)


def resize_linear(img, width, height):
  """cv2.resize(img, (width, height)) with the default INTER_LINEAR on a float image: sample position
  (dst + 0.5) * scale - 0.5, clamped to the border, separable linear weights."""
  img = np.asarray(img)
  h, w = img.shape[:2]

  def axis(n_out, n_in):
    x = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
    x0 = np.floor(x).astype(np.int64)
    fr = x - x0
    lo, hi = np.clip(x0, 0, n_in - 1), np.clip(x0 + 1, 0, n_in - 1)
    return lo, hi, fr
  ylo, yhi, fy = axis(height, h)
  xlo, xhi, fx = axis(width, w)
  fy = fy.reshape((-1, 1) + (1,) * (img.ndim - 2))
  fx = fx.reshape((1, -1) + (1,) * (img.ndim - 2))
  top = img[ylo][:, xlo] * (1 - fx) + img[ylo][:, xhi] * fx
  bot = img[yhi][:, xlo] * (1 - fx) + img[yhi][:, xhi] * fx
  return top * (1 - fy) + bot * fy


def pad_poses(p):
  bottom = np.broadcast_to([0, 0, 0, 1.], p[..., :1, :4].shape)
  return np.concatenate([p[..., :3, :4], bottom], axis=-2)


def unpad_poses(p):
  return p[..., :3, :4]


def _normalize(x):
  return x / np.linalg.norm(x)


def viewmatrix(lookdir, up, position):
  vec2 = _normalize(lookdir)
  vec0 = _normalize(np.cross(up, vec2))
  vec1 = _normalize(np.cross(vec2, vec0))
  return np.stack([vec0, vec1, vec2, position], axis=1)


def recenter_poses(poses):
  """camera_utils.py:112-126."""
  c2w = viewmatrix(poses[:, :3, 2].mean(0), poses[:, :3, 1].mean(0), poses[:, :3, 3].mean(0))
  transform = np.linalg.inv(pad_poses(c2w))
  return unpad_poses(transform @ pad_poses(poses)), transform


def transform_poses_pca(poses):
  """camera_utils.py:191-230."""
  t = poses[:, :3, 3]
  t_mean = t.mean(axis=0)
  t = t - t_mean
  eigval, eigvec = np.linalg.eig(t.T @ t)
  inds = np.argsort(eigval)[::-1]
  eigvec = eigvec[:, inds]
  rot = eigvec.T
  if np.linalg.det(rot) < 0:
    rot = np.diag(np.array([1, 1, -1])) @ rot
  transform = np.concatenate([rot, rot @ -t_mean[:, None]], -1)
  poses_recentered = unpad_poses(transform @ pad_poses(poses))
  transform = np.concatenate([transform, np.eye(4)[3:]], axis=0)
  if poses_recentered.mean(axis=0)[2, 1] < 0:      # flip so that y points down in world space
    poses_recentered = np.diag(np.array([1, -1, -1])) @ poses_recentered
    transform = np.diag(np.array([1, -1, -1, 1])) @ transform
  scale_factor = 1. / np.max(np.abs(poses_recentered[:, :3, 3]))
  poses_recentered[:, :3, 3] *= scale_factor
  transform = np.diag(np.array([scale_factor] * 3 + [1])) @ transform
  return poses_recentered, transform


def _static_mask(mask_dir, stem, image, width, height):
  """datasets.py:1079-1088 / 1223-1232 / 1361-1367: the mask file if it exists (resized to the image), else ones."""
  path = os.path.join(mask_dir, f'{stem}.png')
  if os.path.exists(path):
    m = load_img(path) / 255.
    if m.shape[0] != height or m.shape[1] != width:
      m = resize_linear(m, width, height)
    return m
  return np.ones_like(image[..., :3])


class Kubric(Dataset):
  """datasets.py:984-1113."""

  def _load_renderings(self, config):
    factor = config.factor if config.factor > 0 else 1
    with open(os.path.join(self.data_dir, 'scene_gt.json')) as f:
      sj = json.load(f)
    center, scale = np.array(sj['center']), sj['scale']
    self.scale_factor = scale
    near, far = sj['near'], sj['far'] * 1.2            # "original far is not enough"
    with open(os.path.join(self.data_dir, 'dataset.json')) as f:
      train_names = [str(i) for i in json.load(f)['train_ids']]
    with open(os.path.join(self.data_dir, 'freeze-test/dataset.json')) as f:
      val_names = [str(i) for i in json.load(f)['val_ids']]
    if _is_train(self.split):
      image_dir = os.path.join(self.data_dir, f'rgb/{factor}x')
      mask_dir = os.path.join(self.data_dir, config.static_mask_dir_name)
      camera_dir = os.path.join(self.data_dir, 'camera-gt')
      names, embed_offset = train_names, 0
    else:
      image_dir = os.path.join(self.data_dir, f'freeze-test/static-rgb/{factor}x')
      mask_dir = os.path.join(self.data_dir, f'freeze-test/{config.static_mask_dir_name}')
      camera_dir = os.path.join(self.data_dir, 'freeze-test/camera-gt')
      names, embed_offset = val_names, len(train_names)
    if not os.path.exists(mask_dir):
      print(f'{mask_dir} does not exist. ')
    images, masks, nears, fars, heights, widths, focals, embed, dist, ctypes, c2ws, p2cs = ([] for _ in range(12))
    for i, name in enumerate(names):
      with open(os.path.join(camera_dir, f'{name}.json')) as f:
        cj = json.load(f)
      orientation, position = np.asarray(cj['orientation']), np.asarray(cj['position'])
      fl, pp = cj['focal_length'], np.asarray(cj['principal_point'])
      skew, par = cj['skew'], cj['pixel_aspect_ratio']
      rd, td = np.asarray(cj['radial_distortion']), np.asarray(cj['tangential_distortion'])
      sx, sy = fl, fl * par
      p2c = np.array([[1 / sx, -skew / sx, -pp[0] / sx], [0, 1 / sy, -pp[1] / sy], [0, 0, 1]], dtype=np.float32)
      if factor > 1:
        p2c = p2c @ np.diag([factor, factor, 1.])
      c2w = np.concatenate([orientation.T, position.reshape(3, 1)], axis=1) @ np.diag([1, -1, -1, 1])
      c2w[:3, 3] -= center
      c2w[:3, 3] *= scale
      image = load_img(os.path.join(image_dir, f'{name}.png')) / 255.
      if image.shape[-1] == 4:
        image = image[..., :3] * image[..., -1:] + (1. - image[..., -1:])      # white background
      h, w = image.shape[:2]
      m = _static_mask(mask_dir, name, image, w, h)
      images.append(image)
      masks.append(m[..., :1].reshape(h, w, 1))
      nears.append(np.ones((h, w, 1), np.float32) * near)
      fars.append(np.ones((h, w, 1), np.float32) * far)
      heights.append(h); widths.append(w); embed.append(embed_offset + i); focals.append(fl / factor)
      dist.append(dict(k1=rd[0], k2=rd[1], k3=rd[2], p1=td[0], p2=td[1]))
      ctypes.append(ProjectionType.PERSPECTIVE)
      c2ws.append(c2w); p2cs.append(p2c)
    self.images, self.static_masks, self.nears, self.fars = images, masks, nears, fars
    self.heights, self.widths = np.array(heights), np.array(widths)
    self.focals = np.array(focals, dtype=np.float32)
    self.embed_idxs = np.array(embed)
    self.distortion_params, self.camtypes = dist, ctypes
    self.camtoworlds, self.pixtocams = np.stack(c2ws, 0), np.stack(p2cs, 0)


PHOTOTOURISM_BOUND_DICT = {'brandenburg_gate': 24, 'sacre_coeur': 11, 'taj_mahal': 16, 'trevi_fountain': 35}


def _is_train(split):
  return getattr(split, 'value', split) == 'train'


class _ColmapScene(Dataset):
  """Shared part of Phototourism / Distractor: COLMAP model, split lists, per-image loop."""

  def _select(self, pose_data, train_names, test_names, factor):
    names, poses, p2c, dist, ctypes, pts3d = pose_data
    all_names = train_names + test_names
    sel = train_names if _is_train(self.split) else test_names
    idx = [names.index(n) for n in all_names]
    poses = np.stack([poses[i] for i in idx], 0)
    p2c = np.stack([p2c[i] for i in idx], 0)
    dist, ctypes = [dist[i] for i in idx], [ctypes[i] for i in idx]
    p2c = (p2c @ np.diag([factor, factor, 1.])).astype(np.float32)
    return all_names, sel, poses, p2c, dist, ctypes, pts3d

  def _finish(self, rec):
    self.images, self.static_masks, self.nears, self.fars = rec['images'], rec['masks'], rec['nears'], rec['fars']
    self.focals = np.array(rec['focals'])
    self.heights, self.widths = np.array(rec['heights']), np.array(rec['widths'])
    self.embed_idxs = np.array(rec['embed'])
    self.camtoworlds, self.pixtocams = np.stack(rec['c2w'], 0), np.stack(rec['p2c'], 0)
    self.distortion_params, self.camtypes = rec['dist'], rec['ctypes']


class Phototourism(_ColmapScene):
  """datasets.py:1123-1256."""

  def _load_renderings(self, config):
    factor = config.factor if config.factor > 0 else 1
    pose_data = colmap.process(os.path.join(self.data_dir, 'dense/sparse'))
    split_file = sorted(Path(self.data_dir).glob('*.tsv'))[0]
    rows = [l.rstrip('\n').split('\t') for l in open(split_file)]
    col = {k: i for i, k in enumerate(rows[0])}
    train = [r[col['filename']] for r in rows[1:] if r[col['split']] == 'train']
    test = [r[col['filename']] for r in rows[1:] if r[col['split']] == 'test']
    names, sel, poses, p2c, dist, ctypes, pts3d = self._select(pose_data, train, test, factor)
    focals = 1. / p2c[:, 0, 0]
    poses, transform = recenter_poses(poses)
    pts3d = np.concatenate([pts3d, np.ones_like(pts3d[..., :1])], axis=-1) @ transform.T
    ct = np.eye(4)
    ct[:3, 3] = -pts3d[:, :3].mean(0)                      # the point cloud's centre becomes the origin
    poses = unpad_poses(ct @ pad_poses(poses))
    pts3d = pts3d @ ct.T
    transform = ct @ transform
    sf = 2 / PHOTOTOURISM_BOUND_DICT[Path(self.data_dir).name]
    nt = np.diag([sf, sf, sf, 1])
    poses[..., :3, 3] *= sf
    pts3d = pts3d @ nt.T
    self.colmap_to_world_transform, self.poses, self.pts3d = nt @ transform, poses, pts3d
    image_dir = os.path.join(self.data_dir, 'dense/images')
    mask_dir = os.path.join(self.data_dir, f'dense/{config.static_mask_dir_name}')
    if not os.path.exists(mask_dir):
      print(f'{mask_dir} does not exist. ')
    rec = {k: [] for k in ('images', 'masks', 'nears', 'fars', 'focals', 'heights', 'widths', 'embed', 'c2w', 'p2c', 'dist', 'ctypes')}
    for name in sel:
      i = names.index(name)
      image = load_img(os.path.join(image_dir, name)) / 255.
      h, w = image.shape[:2]
      mpath = os.path.join(mask_dir, f"{name.split('.')[0]}.png")
      m = load_img(mpath) / 255. if os.path.exists(mpath) else np.ones_like(image)
      if factor > 1:
        h, w = h // factor, w // factor
        image = resize_linear(image, w, h)
      if m.shape[0] != h or m.shape[1] != w:
        m = resize_linear(m, w, h)
      pose = pad_poses(poses[i]) @ np.diag([1, -1, -1, 1])           # back to the COLMAP camera frame
      pc = (pts3d @ np.linalg.inv(pose).T)[:, :3]
      pc = pc[pc[:, 2] > 0]
      near, far = np.percentile(pc[:, 2], 0.1), np.percentile(pc[:, 2], 99.9)
      rec['images'].append(image.reshape(h, w, 3)); rec['masks'].append(m[..., :1].reshape(h, w, 1))
      rec['nears'].append(np.ones((h, w, 1), np.float32) * near); rec['fars'].append(np.ones((h, w, 1), np.float32) * far)
      rec['focals'].append(focals[i]); rec['heights'].append(h); rec['widths'].append(w); rec['embed'].append(i)
      rec['c2w'].append(poses[i]); rec['p2c'].append(p2c[i]); rec['dist'].append(dist[i]); rec['ctypes'].append(ctypes[i])
    self._finish(rec)


class Distractor(_ColmapScene):
  """datasets.py:1259-1394 (the RobustNeRF scenes)."""

  def _load_renderings(self, config):
    suffix, factor = ('', 1) if config.factor <= 0 else (f'_{config.factor}', config.factor)
    pose_data = colmap.process(os.path.join(self.data_dir, '0/sparse/0'))
    with open(os.path.join(self.data_dir, '0/data_split.json')) as f:
      sd = json.load(f)
    names, sel, poses, p2c, dist, ctypes, pts3d = self._select(pose_data, sd['train'], sd['test'], factor)
    focals = 1. / p2c[:, 0, 0]
    poses, transform = transform_poses_pca(poses)
    pts3d = np.concatenate([pts3d, np.ones_like(pts3d[..., :1])], axis=-1) @ transform.T
    ct = np.eye(4)
    ct[:3, 3] = -pts3d[:, :3].mean(0)
    poses = unpad_poses(ct @ pad_poses(poses))
    pts3d = pts3d @ ct.T
    transform = ct @ transform
    sf = 1. / np.max(np.abs(poses[:, :3, 3]))                  # cameras back into the unit cube
    poses[:, :3, 3] *= sf
    pts3d[:, :3] *= sf
    self.colmap_to_world_transform = np.diag(np.array([sf] * 3 + [1])) @ transform
    self.poses, self.pts3d = poses, pts3d
    image_dir = os.path.join(self.data_dir, f'0/images{suffix}')
    mask_dir = os.path.join(self.data_dir, f'0/{config.static_mask_dir_name}')
    if not os.path.exists(mask_dir):
      print(f'{mask_dir} does not exist. ')
    rec = {k: [] for k in ('images', 'masks', 'nears', 'fars', 'focals', 'heights', 'widths', 'embed', 'c2w', 'p2c', 'dist', 'ctypes')}
    for name in sel:
      i = names.index(name)
      image = load_img(os.path.join(image_dir, name)) / 255.
      h, w = image.shape[:2]
      m = _static_mask(mask_dir, name.split('.')[0], image, w, h)
      pose = pad_poses(poses[i]) @ np.diag([1, -1, -1, 1])
      pc = (pts3d @ np.linalg.inv(pose).T)[:, :3]
      pc = pc[pc[:, 2] >= 0]
      uv = (pc @ np.linalg.inv(p2c[i]).T) / np.maximum(pc[:, 2:], np.finfo(pc.dtype).eps)
      inside = (uv[:, 0] <= w) * (uv[:, 0] >= 0) * (uv[:, 1] <= h) * (uv[:, 1] >= 0)
      near = np.percentile(pc[inside][:, 2], 0.1) * 0.8
      rec['images'].append(image.reshape(h, w, 3)); rec['masks'].append(m[..., :1].reshape(h, w, 1))
      rec['nears'].append(np.ones((h, w, 1), np.float32) * near); rec['fars'].append(np.ones((h, w, 1), np.float32) * self.far)
      rec['focals'].append(focals[i]); rec['heights'].append(h); rec['widths'].append(w); rec['embed'].append(i)
      rec['c2w'].append(poses[i]); rec['p2c'].append(p2c[i]); rec['dist'].append(dist[i]); rec['ctypes'].append(ctypes[i])
    self._finish(rec)


def load_dataset(split, is_training, sample_from_half_image, batch_size, patch_size, patch_dilation, image_num_per_batch,
                 train_dir, config, **kw):
  """datasets.py:45-77 for the loaders built here."""
  table = {'kubric': Kubric, 'phototourism': Phototourism, 'distractor': Distractor}
  if config.dataset_loader not in table:
    raise NotImplementedError(f'dataset_loader {config.dataset_loader!r}: built here are {sorted(table)} '
                              '(blender / llff / tat / dtu are not HuGS datasets; ArrayDataset takes decoded arrays)')
  return table[config.dataset_loader](split, is_training, sample_from_half_image, batch_size, patch_size, patch_dilation,
                                      image_num_per_batch, train_dir, config, **kw)

"""COLMAP sparse-model reader (cameras.bin / images.bin / points3D.bin) + the NeRF post-processing the reference
applies on top of it (`NeRFSceneManager.process`, MipNeRF360/internal/datasets.py:80-185).

The reference delegates the file parsing to pycolmap's SceneManager (un-vendored third party: absent from
/root/reference and from this image), so the binary layouts below follow COLMAP's published format
(src/colmap/scene/reconstruction_io: little endian; cameras: id i32, model i32, width u64, height u64, params f64[];
images: id i32, qvec f64[4] (w,x,y,z), tvec f64[3], camera_id i32, name cstring, n u64, n x (x f64, y f64, point3D_id
i64); points3D: id u64, xyz f64[3], rgb u8[3], error f64, track u64, track x (image_id i32, point2D_idx i32)).
Host-side IO: numpy only, runs once per dataset."""
import os
import struct

import numpy as np

from .camera_utils import ProjectionType

# model id -> (name, number of params)
CAMERA_MODELS = {0: ('SIMPLE_PINHOLE', 3), 1: ('PINHOLE', 4), 2: ('SIMPLE_RADIAL', 4), 3: ('RADIAL', 5), 4: ('OPENCV', 8),
                 5: ('OPENCV_FISHEYE', 8), 6: ('FULL_OPENCV', 12), 7: ('FOV', 5), 8: ('SIMPLE_RADIAL_FISHEYE', 4),
                 9: ('RADIAL_FISHEYE', 5), 10: ('THIN_PRISM_FISHEYE', 12)}


def _rd(f, fmt):
  return struct.unpack('<' + fmt, f.read(struct.calcsize('<' + fmt)))


def read_cameras_binary(path):
  cams = {}
  with open(path, 'rb') as f:
    (n,) = _rd(f, 'Q')
    for _ in range(n):
      cid, model, w, h = _rd(f, 'iiQQ')
      if model not in CAMERA_MODELS:
        raise ValueError(f'unknown COLMAP camera model {model}')
      params = np.array(_rd(f, 'd' * CAMERA_MODELS[model][1]))
      cams[cid] = dict(model=model, width=w, height=h, params=params)
  return cams


def read_images_binary(path):
  imgs = {}
  with open(path, 'rb') as f:
    (n,) = _rd(f, 'Q')
    for _ in range(n):
      (iid,) = _rd(f, 'i')
      q = np.array(_rd(f, 'dddd'))
      t = np.array(_rd(f, 'ddd'))
      (cid,) = _rd(f, 'i')
      name = b''
      while True:
        c = f.read(1)
        if c in (b'\x00', b''):
          break
        name += c
      (m,) = _rd(f, 'Q')
      f.seek(24 * m, os.SEEK_CUR)
      imgs[iid] = dict(qvec=q, tvec=t, camera_id=cid, name=name.decode())
  return imgs


def read_points3D_binary(path):
  pts = []
  with open(path, 'rb') as f:
    (n,) = _rd(f, 'Q')
    for _ in range(n):
      _rd(f, 'Q')
      pts.append(_rd(f, 'ddd'))
      f.seek(3 + 8, os.SEEK_CUR)
      (tl,) = _rd(f, 'Q')
      f.seek(8 * tl, os.SEEK_CUR)
  return np.array(pts, np.float64).reshape(-1, 3)


def qvec_to_rotmat(q):
  w, x, y, z = q
  return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                   [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                   [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def process(colmap_dir):
  """datasets.py:87-185.  Returns (names, poses [N,3,4] camera-to-world in the NeRF frame, pixtocams [N,3,3],
  distortion_params (per image dict or None), camtypes, pts3d [P,3])."""
  cams = read_cameras_binary(os.path.join(colmap_dir, 'cameras.bin'))
  imgs = read_images_binary(os.path.join(colmap_dir, 'images.bin'))
  pts3d = read_points3D_binary(os.path.join(colmap_dir, 'points3D.bin'))
  bottom = np.array([0, 0, 0, 1.]).reshape(1, 4)
  w2c, p2c, dist, ctypes, names = [], [], [], [], []
  for k in imgs:
    im = imgs[k]
    w2c.append(np.concatenate([np.concatenate([qvec_to_rotmat(im['qvec']), im['tvec'].reshape(3, 1)], 1), bottom], 0))
    cam = cams[im['camera_id']]
    m, p = cam['model'], cam['params']
    if m in (0, 2, 3):               # one focal length
      fx = fy = p[0]; cx, cy = p[1], p[2]; rest = p[3:]
    else:
      fx, fy, cx, cy = p[:4]; rest = p[4:]
    p2c.append(np.linalg.inv(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.]])))
    params, ct = None, ProjectionType.PERSPECTIVE
    if m == 2:
      params = dict(k1=rest[0], k2=0., k3=0., p1=0., p2=0.)
    elif m == 3:
      params = dict(k1=rest[0], k2=rest[1], k3=0., p1=0., p2=0.)
    elif m == 4:
      params = dict(k1=rest[0], k2=rest[1], k3=0., p1=rest[2], p2=rest[3])
    elif m == 5:
      params = dict(k1=rest[0], k2=rest[1], k3=rest[2], k4=rest[3])
      ct = ProjectionType.FISHEYE
    elif m not in (0, 1):
      raise NotImplementedError(f'COLMAP camera model {CAMERA_MODELS[m][0]} (the reference handles models 0-5)')
    dist.append(params)
    ctypes.append(ct)
    names.append(im['name'])
  poses = np.linalg.inv(np.stack(w2c, 0))[:, :3, :4]
  poses = poses @ np.diag([1, -1, -1, 1])      # COLMAP (right, down, fwd) -> NeRF (right, up, back)
  return names, poses, np.stack(p2c, 0), dist, ctypes, pts3d

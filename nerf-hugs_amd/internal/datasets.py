"""Device-resident batch provider: the reference's `Dataset` (MipNeRF360/internal/datasets.py:225-548) with the
images, static masks and camera tables kept in HBM and every batch assembled by HIP kernels
(`hugs_expand_patches`, `hugs_gather_pixels`, `hugs_pixels_to_rays`) on the training stream.

What is kept from the reference: the constructor signature, `_load_renderings` as the subclass hook and the
attributes it must fill (datasets.py:319-332), the batch layout ([num_patches, patch, patch, C], images
concatenated on axis 0), `generate_ray_batch`, `peek`, `size`, and -- so that a run is reproducible against the
reference -- the exact `np.random` call sequence of `_next_train` (:507-519: camera, patch x origins, patch y
origins, per image).  What is gone: the producer thread and `Queue(3)` (:289): kernels are asynchronous, so
`__next__` just enqueues ~6 launches.  The HuGS file-format loaders (Kubric, Phototourism, Distractor) live in
loaders.py; the upstream multinerf ones (Blender, LLFF, T&T, DTU) are not HuGS datasets and are not built --
`ArrayDataset` takes arrays a loader has already decoded.
"""
import numpy as np
import torch

from .. import _lib as L
from . import camera_utils
from . import utils


def _const(a):
  a = np.asarray(a)
  return a.size > 0 and bool((a == a.flat[0]).all())


class Dataset:

  def __init__(self, split, is_training, sample_from_half_image, batch_size, patch_size, patch_dilation,
               image_num_per_batch, data_dir, config, device='cuda', world_size=1, random_state=None):
    self._patch_size = max(int(patch_size), 1)
    self._batch_size = batch_size // world_size
    self._image_num_per_batch = image_num_per_batch // world_size
    self._patch_dilation = patch_dilation
    if self._image_num_per_batch * self._patch_size ** 2 > self._batch_size:
      raise ValueError(f'Image size {self._image_num_per_batch} * Patch size {self._patch_size}^2 too large for '
                       f'per-process batch size {self._batch_size}')
    if self._image_num_per_batch < 1:
      raise ValueError('image_num_per_batch must be at least the number of processes')
    self._test_camera_idx = 0
    self.split = split
    self.is_training = is_training
    self.sample_from_half_image = sample_from_half_image
    self.data_dir = data_dir
    self.near, self.far = config.near, config.far
    if getattr(config, 'render_path', False):
      raise NotImplementedError()   # as datasets.py:335
    self.device = torch.device(device)
    self._rs = np.random if random_state is None else random_state   # the reference draws from the global stream
    self.distortion_params = None
    self.pixtocam_ndc = None
    self.camtypes = None
    self.images = self.static_masks = self.heights = self.widths = self.nears = self.fars = None
    self.embed_idxs = self.camtoworlds = self.pixtocams = None
    self._load_renderings(config)
    self._n_examples = int(np.asarray(self.camtoworlds).shape[0])
    self._upload()
    self._next_fn = self._next_train if is_training else self._next_test
    self._peeked = None

  def _load_renderings(self, config):
    raise NotImplementedError

  # ---- HBM layout: one flat pixel buffer per attribute + per-image offsets ---------------------------------
  def _flat(self, arrs, channels, allow_u8=False):
    arrs = [np.asarray(a) for a in arrs]
    u8 = allow_u8 and all(a.dtype == np.uint8 for a in arrs)
    flat = np.concatenate([a.reshape(-1, channels) for a in arrs], 0)
    flat = flat if u8 else flat.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(flat)).to(self.device), u8

  def _per_image_or_pixel(self, arrs, default):
    """near / far / static mask: [ncams] table when every image is constant, else a per-pixel buffer."""
    n = self._n_examples
    if arrs is None:
      return torch.full((n, 1), float(default), dtype=torch.float32, device=self.device), 0
    if all(np.ndim(a) == 0 or _const(a) for a in arrs):
      t = np.array([[float(np.asarray(a).flat[0])] for a in arrs], np.float32)
      return torch.from_numpy(t).to(self.device), 0
    return self._flat(arrs, 1)[0], 1

  def _upload(self):
    n, dev = self._n_examples, self.device
    self.heights = np.asarray(self.heights).astype(np.int64)
    self.widths = np.asarray(self.widths).astype(np.int64)
    if len(self.heights) != n or len(self.widths) != n:
      raise ValueError('heights / widths must have one entry per camera')
    off = np.concatenate([[0], np.cumsum(self.heights * self.widths)])
    i32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int32)).to(dev)
    self._offsets = torch.from_numpy(off[:-1].astype(np.int64)).to(dev)
    self._widths, self._heights = i32(self.widths), i32(self.heights)
    self._images = None
    if self.images is not None:
      for a, h, w in zip(self.images, self.heights, self.widths):
        if tuple(np.shape(a)[:2]) != (h, w):
          raise ValueError(f'image of shape {np.shape(a)} in a {h}x{w} slot')
      self._images, self._images_u8 = self._flat(self.images, 3, allow_u8=True)
    self._masks, self._masks_pp = self._per_image_or_pixel(self.static_masks, 1.)
    self._nears, self._nears_pp = self._per_image_or_pixel(self.nears, self.near)
    self._fars, self._fars_pp = self._per_image_or_pixel(self.fars, self.far)
    self._embed = i32(np.arange(n) if self.embed_idxs is None else self.embed_idxs)
    p2c = np.asarray(self.pixtocams, np.float32)
    self._p2c = torch.from_numpy(np.array(np.broadcast_to(p2c, (n, 3, 3)) if p2c.ndim == 2 else p2c)).to(dev)
    self._c2w = torch.from_numpy(np.ascontiguousarray(np.asarray(self.camtoworlds, np.float32)[:, :3, :4])).to(dev)
    self._ndc = None if self.pixtocam_ndc is None else torch.from_numpy(np.asarray(self.pixtocam_ndc, np.float32)).to(dev)
    dp = self.distortion_params
    if dp is None or isinstance(dp, dict):
      dp = [dp] * n
    if all(d is None for d in dp):
      self._dist = None
    else:   # an all-zero row leaves the Newton iteration at its starting point: identical to "no distortion"
      rows = [[0.] * 6 if d is None else [float(d.get(k, 0.)) for k in camera_utils._DIST_KEYS] for d in dp]
      self._dist = torch.tensor(rows, dtype=torch.float32, device=dev)
    ct = self.camtypes
    if ct is None:
      ct = [camera_utils.ProjectionType.PERSPECTIVE] * n
    self._camtypes = [camera_utils.ProjectionType(c) if not isinstance(c, camera_utils.ProjectionType) else c for c in ct]
    self.cameras = (self._p2c, self._c2w, self._ndc)

  @property
  def size(self):
    return self._n_examples

  def __iter__(self):
    return self

  def __next__(self):
    if self._peeked is not None:
      b, self._peeked = self._peeked, None
      return b
    return self._next_fn()

  def peek(self):
    """datasets.py:410-421: the next batch without consuming it."""
    if self._peeked is None:
      self._peeked = self._next_fn()
    return self._peeked

  # ---- batch assembly -----------------------------------------------------------------------------------------
  def _gather(self, n, px, py, ci, table, per_pixel, channels=1, u8=False):
    out = torch.empty((n, channels), dtype=torch.float32, device=self.device)
    L.call('hugs_gather_pixels', n, channels, px, py, ci, self._offsets, self._widths, per_pixel, 1 if u8 else 0,
           table, out)
    return out

  def _make_ray_batch(self, px, py, ci, shape, lossmult=None):
    """datasets.py:447-492 for flat int32 device tensors px, py, ci; outputs reshaped to shape + [C]."""
    n = px.numel()
    kinds = {self._camtypes[c] for c in np.unique(ci.cpu().numpy())} if len(set(self._camtypes)) > 1 \
        else {self._camtypes[0]}
    if len(kinds) != 1:
      raise NotImplementedError('one batch mixing perspective and fisheye cameras')
    o, d, v, r, pc = camera_utils.pixels_to_rays(px, py, self._p2c, self._c2w, self._dist, self._ndc, kinds.pop(),
                                                 cam_idx=ci, widths=self._widths, heights=self._heights,
                                                 validate=False)
    rs = lambda a: a.reshape(tuple(shape) + (a.shape[-1],))
    ci64 = ci.long()
    rays = utils.Rays(
        pix_coords=rs(pc), origins=rs(o), directions=rs(d), viewdirs=rs(v), radii=rs(r),
        lossmult=rs(torch.ones((n, 1), dtype=torch.float32, device=self.device) if lossmult is None else lossmult),
        static_mask=rs(self._gather(n, px, py, ci, self._masks, self._masks_pp)),
        near=rs(self._gather(n, px, py, ci, self._nears, self._nears_pp)),
        far=rs(self._gather(n, px, py, ci, self._fars, self._fars_pp)),
        embed_idx=rs(self._embed[ci64][:, None]), cam_idx=rs(ci[:, None]))
    rgb = None
    if self._images is not None:
      rgb = rs(self._gather(n, px, py, ci, self._images, 1, channels=3, u8=self._images_u8))
    return utils.Batch(rays=rays, rgb=rgb)

  def _next_train(self):
    ps, dil = self._patch_size, self._patch_dilation
    p = (self._batch_size // self._image_num_per_batch) // ps ** 2
    upper = (ps - 1) * dil
    npatch = self._image_num_per_batch * p
    pinned, ev = self._staging(npatch)
    host = pinned.numpy()
    for i in range(self._image_num_per_batch):   # the reference's draw order, datasets.py:507-519
      cam = self._rs.randint(0, self._n_examples)
      h, w = int(self.heights[cam]), int(self.widths[cam])
      if self.sample_from_half_image:
        w = w // 2
      host[0, i * p:(i + 1) * p] = self._rs.randint(0, w - upper, (p, 1, 1))[:, 0, 0]
      host[1, i * p:(i + 1) * p] = self._rs.randint(0, h - upper, (p, 1, 1))[:, 0, 0]
      host[2, i * p:(i + 1) * p] = cam
    org = pinned.to(self.device, non_blocking=True)     # pinned source: a truly asynchronous copy, no host stall
    ev.record()
    n = npatch * ps * ps
    pix = torch.empty((3, n), dtype=torch.int32, device=self.device)
    L.call('hugs_expand_patches', npatch, ps, dil, org[0], org[1], org[2], pix[0], pix[1], pix[2])
    return self._make_ray_batch(pix[0], pix[1], pix[2], (npatch, ps, ps))

  def _staging(self, npatch):
    """A small ring of pinned host buffers for the patch origins; a slot is reused only after the copy that last read
    it has completed (the event is normally long done: the ring is 8 deep)."""
    ring = getattr(self, '_pin_ring', None)
    if ring is None or ring[0][0].shape[1] != npatch:
      ring = [(torch.empty((3, npatch), dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(8)]
      self._pin_ring, self._pin_next = ring, 0
    buf, ev = ring[self._pin_next]
    self._pin_next = (self._pin_next + 1) % len(ring)
    ev.synchronize()
    return buf, ev

  def generate_ray_batch(self, cam_idx):
    """datasets.py:531-541: every pixel of one image, [H, W, C]."""
    h, w = int(self.heights[cam_idx]), int(self.widths[cam_idx])
    x, y = camera_utils.pixel_coordinates(w, h, self.device)
    ci = torch.full((h * w,), int(cam_idx), dtype=torch.int32, device=self.device)
    return self._make_ray_batch(x.reshape(-1), y.reshape(-1), ci, (h, w))

  def _next_test(self):
    cam = self._test_camera_idx
    self._test_camera_idx = (cam + 1) % self._n_examples
    return self.generate_ray_batch(cam)


class ArrayDataset(Dataset):
  """A Dataset over arrays that are already decoded: images [H,W,3] float32 in [0,1] or uint8, pixtocams
  [3,3] | [N,3,3], camtoworlds [N,3,4]; optional static_masks [H,W,1] (1 = static), nears / fars (scalars or
  [H,W,1]), embed_idxs, distortion_params (dict or per-camera list), camtypes, pixtocam_ndc."""

  def __init__(self, config, images, pixtocams, camtoworlds, heights=None, widths=None, static_masks=None, nears=None,
               fars=None, embed_idxs=None, distortion_params=None, camtypes=None, pixtocam_ndc=None, split='train',
               is_training=True, sample_from_half_image=False, **kw):
    self._src = dict(images=images, pixtocams=pixtocams, camtoworlds=camtoworlds, heights=heights, widths=widths,
                     static_masks=static_masks, nears=nears, fars=fars, embed_idxs=embed_idxs,
                     distortion_params=distortion_params, camtypes=camtypes, pixtocam_ndc=pixtocam_ndc)
    super().__init__(split, is_training, sample_from_half_image, config.batch_size, config.patch_size,
                     getattr(config, 'patch_dilation', 1), getattr(config, 'image_num_per_batch', 1), None, config, **kw)

  def _load_renderings(self, config):
    s = self._src
    for k, v in s.items():
      setattr(self, k, v)
    if self.heights is None:
      self.heights = [np.shape(a)[0] for a in self.images]
      self.widths = [np.shape(a)[1] for a in self.images]
    del self._src

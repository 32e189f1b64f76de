"""Synthetic inputs with structure (no dataset ships; there is no network): an analytic scene whose pixel colours are a function of the
ray -- a textured, shaded unit sphere on a white background seen by pinhole cameras on a ring of radius 3 -- so that a network trained
on it LEARNS (bench.py's operand distribution, the PSNR-equivalence tests).  `scene_batch` draws whole P x P patches."""
import numpy as np
import torch

from . import utils


def scene_batch(rng, n_patch, P, device):
  o = np.zeros((n_patch, P, P, 3), np.float32); d = np.zeros_like(o)
  for i in range(n_patch):
    th = rng.uniform(0, 2 * np.pi); c = np.array([3 * np.cos(th), 3 * np.sin(th), rng.uniform(-0.5, 0.5)], np.float32)
    f = -c / np.linalg.norm(c); r = np.cross(f, [0, 0, 1]); r /= np.linalg.norm(r); u = np.cross(r, f)
    x0, y0 = rng.uniform(-0.35, 0.35, 2)
    px = x0 + (np.arange(P) - P / 2) * 0.004; py = y0 + (np.arange(P) - P / 2) * 0.004
    X, Y = np.meshgrid(px, py)
    o[i] = c; d[i] = f[None, None] + X[..., None] * r + Y[..., None] * u
  v = d / np.linalg.norm(d, axis=-1, keepdims=True)
  b = (o * v).sum(-1); cc = (o * o).sum(-1) - 1.0; disc = b * b - cc
  hit = disc > 0
  t = -b - np.sqrt(np.maximum(disc, 0))
  p = o + v * t[..., None]
  tex = 0.5 + 0.5 * np.stack([np.sin(6 * p[..., 0]), np.sin(6 * p[..., 1] + 1), np.sin(6 * p[..., 2] + 2)], -1)
  shade = np.clip((p * np.array([0.5, 0.3, 0.8])).sum(-1, keepdims=True) * 0.5 + 0.6, 0.2, 1.0)
  rgb = np.where(hit[..., None], tex * shade, 1.0).astype(np.float32)
  f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(device)
  shp = (n_patch, P, P)
  rays = utils.Rays(pix_coords=f32(np.zeros(shp + (2,))), origins=f32(o), directions=f32(d), viewdirs=f32(v),
                    radii=f32(np.full(shp + (1,), 0.004 * 2 / np.sqrt(12))), lossmult=f32(np.ones(shp + (1,))),
                    static_mask=f32(np.ones(shp + (1,))), near=f32(np.full(shp + (1,), 1.5)), far=f32(np.full(shp + (1,), 4.5)),
                    embed_idx=torch.zeros(shp + (1,), dtype=torch.int32, device=device),
                    cam_idx=torch.zeros(shp + (1,), dtype=torch.int32, device=device))
  return utils.Batch(rays=rays, rgb=f32(rgb))

"""Function-level entry points of reference MipNeRF360/internal/render.py on the HIP kernels (same argument
meaning, torch cuda tensors in/out).  The fused training path does not go through these wrappers; they exist so
that code written against the reference's leaf API keeps working."""
import torch

from .. import _lib


def _flat(x, c):
  return x.reshape(-1, c).to(torch.float32).contiguous()


def compute_alpha_weights(density, tdist, dirs, opaque_background=False):
  """render.py:130-151.  Returns weights (alpha/trans are internal to the kernel)."""
  S = density.shape[-1]
  d, t, dr = _flat(density, S), _flat(tdist, S + 1), _flat(dirs, 3)
  N = d.shape[0]
  w = torch.empty(N, S, device=d.device)
  rgb = torch.empty(N, 3, device=d.device)
  _lib.call('hugs_composite_fwd', N, S, d, None, t, dr, int(opaque_background), 0.0, None, w, rgb, None)
  return w.reshape(density.shape)


def volumetric_rendering(rgbs, density, tdist, dirs, bg_rgbs, t_far, compute_extras, opaque_background=False):
  """render.py:185-244 fused with compute_alpha_weights (the kernel composites from density).
  bg_rgbs: scalar.  Returns the reference's rendering dict."""
  S = density.shape[-1]
  lead = density.shape[:-1]
  d, t, dr = _flat(density, S), _flat(tdist, S + 1), _flat(dirs, 3)
  c = rgbs.reshape(-1, S, 3).to(torch.float32).contiguous()
  N = d.shape[0]
  w = torch.empty(N, S, device=d.device)
  rgb = torch.empty(N, 3, device=d.device)
  ex = torch.empty(N, 5, device=d.device) if compute_extras else None
  far = _flat(t_far, 1).reshape(-1) if t_far is not None else None
  _lib.call('hugs_composite_fwd', N, S, d, c, t, dr, int(opaque_background), float(bg_rgbs), far, w, rgb, ex)
  out = {'rgb': rgb.reshape(lead + (3,))}
  if compute_extras:
    for i, k in enumerate(['acc', 'distance_mean', 'distance_median', 'distance_percentile_5', 'distance_percentile_95']):
      out[k] = ex[:, i].reshape(lead)
  return out, w.reshape(lead + (S,))


def cast_rays_ipe(tdist, origins, directions, radii, ray_shape, basis, max_deg, warp_contract=False, bf16=False):
  """render.py:103-127 cast_rays + coord.py:39-60,129-133,107-126 in one kernel: returns the IPE features
  [..., S, 2*nb*max_deg] (the Gaussians themselves are never materialised)."""
  if ray_shape not in ('cone', 'cylinder'):
    raise ValueError('ray_shape must be \'cone\' or \'cylinder\'')
  S = tdist.shape[-1] - 1
  t, o, d, r = _flat(tdist, S + 1), _flat(origins, 3), _flat(directions, 3), _flat(radii, 1).reshape(-1)
  N = t.shape[0]
  nb = basis.shape[1]
  F = 2 * nb * max_deg
  Fp = (F + 63) // 64 * 64
  out = torch.empty(N * S, Fp, device=t.device, dtype=torch.bfloat16 if bf16 else torch.float32)
  _lib.call('hugs_cast_ipe_fwd', N, S, t, o, d, r, basis.to(torch.float32).contiguous(), nb, 0 if ray_shape == 'cone' else 1,
            int(warp_contract), max_deg, int(bf16), Fp, out)
  return out[:, :F].reshape(tdist.shape[:-1] + (S, F))

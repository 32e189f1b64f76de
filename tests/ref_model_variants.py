"""Reader for tests/golden/ref_model_variants.npz (the reference's own Model.__call__ executed for option variants; generator
tests/golden/gen_model_variant_fixtures.py).  Data only: nothing here touches /root/reference."""
import json
import os

import numpy as np
import torch

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_model_variants.npz')
_NPZ = None
CASES = ('near_anneal', 'per_sample_jitter', 'cylinder', 'levels4', 'sampler_knobs', 'head_knobs', 'no_viewdirs', 'view_depth3',
         'grey_background', 'log_raydist_contract')
RAY_FIELDS = ('pix_coords', 'origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'static_mask', 'near', 'far', 'embed_idx', 'cam_idx')


def npz():
  global _NPZ
  if _NPZ is None:
    _NPZ = np.load(_PATH)
  return _NPZ


def spec(case):
  return json.loads(str(npz()[f'{case}/spec']))


def get(case, key):
  return npz()[f'{case}/{key}']


def param_tree(case, dtype=torch.float32):
  tree = {}
  pre, preb = f'{case}/params/', f'{case}/params_bf16/'
  for k in npz().files:
    if k.startswith(pre):
      name, v = k[len(pre):], npz()[k]
    elif k.startswith(preb):
      name, v = k[len(preb):], (npz()[k].astype(np.uint32) << 16).view(np.float32)
    else:
      continue
    d = tree
    parts = name.split('/')
    for p in parts[:-1]:
      d = d.setdefault(p, {})
    d[parts[-1]] = torch.from_numpy(v.copy()).to(dtype)
  return {'params': tree}


def rays_flat(case, dtype=torch.float32):
  out = {}
  for f in RAY_FIELDS:
    a = get(case, f'rays/{f}')
    t = torch.from_numpy(a.reshape(-1, a.shape[-1]).copy())
    out[f] = t if f in ('embed_idx', 'cam_idx') else t.to(dtype)
  return out


def u01(case, L):
  """per level: [N] (one draw per ray) or [N, S] (Model.single_jitter = False)"""
  out = []
  for l in range(L):
    a = get(case, f'l{l}_u01')
    out.append(torch.from_numpy((a[:, 0] if a.shape[1] == 1 else a).copy()))
  return out


def gin_lines(case):
  s = spec(case)
  lines = ["Config.randomized = True"]
  for scope in ('Model', 'NerfMLP', 'PropMLP'):
    for k, v in s.get(scope, {}).items():
      if isinstance(v, list):
        v = tuple(v)
      lines.append(f'{scope}.{k} = {v if isinstance(v, str) and v.startswith("@") else repr(v)}')
  return lines


def oracle_cfg(case):
  from oracle import torch_ref as R
  s = spec(case)
  m, n, p = s['Model'], s['NerfMLP'], s['PropMLP']
  rd = {'@jnp.reciprocal': 'reciprocal', '@jnp.log': 'log'}.get(m.get('raydist_fn'))
  bg = m.get('bg_intensity_range', (1., 1.))
  cfg = R.ModelCfg(
      num_prop_samples=m['num_prop_samples'], num_nerf_samples=m['num_nerf_samples'], num_levels=m['num_levels'], raydist_fn=rd,
      num_glo_features=m.get('num_glo_features', 0), num_embeddings=m['num_embeddings'], opaque_background=m['opaque_background'],
      warp=n.get('warp_fn') == '@coord.contract', nerf_depth=n['net_depth'], nerf_width=n['net_width'],
      bottleneck_width=n['bottleneck_width'], width_viewdirs=n.get('net_width_viewdirs', 128), prop_depth=p['net_depth'],
      prop_width=p['net_width'], prop_disable_rgb=p['disable_rgb'], ray_shape=m.get('ray_shape', 'cone'),
      single_jitter=m.get('single_jitter', True), anneal_slope=float(m.get('anneal_slope', 10.)),
      dilation_multiplier=m.get('dilation_multiplier', 0.5), dilation_bias=m.get('dilation_bias', 0.0025),
      resample_padding=m.get('resample_padding', 0.0), bg_intensity=0.5 * (bg[0] + bg[1]),
      skip_layer=n.get('skip_layer', 4), deg_view=n.get('deg_view', 4), density_bias=n.get('density_bias', -1.),
      rgb_padding=n.get('rgb_padding', 0.001), max_deg_point=n.get('max_deg_point', 12))
  cfg.use_viewdirs = m.get('use_viewdirs', True)
  cfg.depth_viewdirs = n.get('net_depth_viewdirs', 1)
  cfg.near_anneal_rate, cfg.near_anneal_init = m.get('near_anneal_rate'), m.get('near_anneal_init', 0.95)
  return cfg

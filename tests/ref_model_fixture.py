"""Reader for tests/golden/ref_model.npz (the reference's own models.py / train_utils.py executed
under the numpy stand-ins; generator tests/golden/gen_model_fixtures.py).  Shared by the CPU test of the
oracle and the GPU test of the HIP path.  Data only: nothing here touches /root/reference."""
import json
import os

import numpy as np
import torch

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
_PATH = os.path.join(_HERE, 'ref_model.npz')
_NPZ = None
CASES = ('base', 'withmask', 'robust', 'nerfw', 'hanerf')
# round 5: loss / optimizer option variants of the reference's train_step (tests/golden/gen_model_train_variant_fixtures.py ->
# ref_model_train_variants.npz: same key layout, no finite differences); one key space with the main file
TRAIN_VARIANTS = ('coarse_charb', 'decay_clips_schedule')
# the finetune stage executed (tests/golden/gen_model_finetune_fixtures.py -> ref_model_finetune.npz: create_finetune_optimizer +
# create_train_step(model, config, True) on a NeRF-W and a HA-NeRF model); same key layout, no finite differences
FINETUNE_CASES = ('ft_nerfw', 'ft_hanerf')


class _Merged:
  def __init__(self, paths):
    zs = [np.load(p) for p in paths]
    self._of = {k: z for z in zs for k in z.files}
    self.files = list(self._of)

  def __getitem__(self, k):
    return self._of[k][k]
RAY_FIELDS = ('pix_coords', 'origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'static_mask', 'near', 'far',
              'embed_idx', 'cam_idx')


def npz():
  global _NPZ
  if _NPZ is None:
    _NPZ = _Merged([_PATH, os.path.join(_HERE, 'ref_model_train_variants.npz'), os.path.join(_HERE, 'ref_model_finetune.npz')])
  return _NPZ


def spec(case):
  return json.loads(str(npz()[f'{case}/spec']))


def get(case, key):
  return npz()[f'{case}/{key}']


def keys(case, prefix):
  p = f'{case}/{prefix}'
  return [k[len(p):] for k in npz().files if k.startswith(p)]


def flat_params(case):
  """{'NerfMLP_0/Dense_0/kernel': float32 array, ...}; kernels are stored as bfloat16 bit patterns (they were
  rounded to bfloat16-representable values before the reference ran on them)."""
  out = {k: npz()[f'{case}/params/{k}'] for k in keys(case, 'params/')}
  for k in keys(case, 'params_bf16/'):
    out[k] = (npz()[f'{case}/params_bf16/{k}'].astype(np.uint32) << 16).view(np.float32)
  return out


def param_tree(case, dtype=torch.float32):
  tree = {}
  for k, v in flat_params(case).items():
    d = tree
    parts = k.split('/')
    for p in parts[:-1]:
      d = d.setdefault(p, {})
    d[parts[-1]] = torch.from_numpy(v.copy()).to(dtype)
  return {'params': tree}


N_DIRS = 10      # seeded parameter directions with a float64 finite-difference derivative of the reference's loss_fn per case


def seeded_tree(case, seed, scale=1.0):
  """The generator's direction / synthetic-gradient convention: leaves in sorted key order, one
  standard_normal(shape) float64 draw each from default_rng(seed)."""
  fp = flat_params(case)
  rng = np.random.default_rng(seed)
  return {k: rng.standard_normal(fp[k].shape) * scale for k in sorted(fp)}


def rays_flat(case, dtype=torch.float32):
  """dict of [N,c] tensors (oracle.torch_ref convention)."""
  out = {}
  for f in RAY_FIELDS:
    a = get(case, f'rays/{f}')
    t = torch.from_numpy(a.reshape(-1, a.shape[-1]).copy())
    out[f] = t if f in ('embed_idx', 'cam_idx') else t.to(dtype)
  return out


def u01(case, L):
  return [torch.from_numpy(get(case, f'l{l}_u01').copy()) for l in range(L)]


def is_finetune(case):
  return bool(spec(case)['Config'].get('finetune_enable', False))


def oracle_cfg(case, finetune_optimizer=False):
  """spec -> oracle.torch_ref.ModelCfg (same knob names the reference's gin bindings use).  finetune_optimizer: the schedule /
  Adam fields carry the spec's finetune_* values (create_finetune_optimizer, train_utils.py:518-537)."""
  from oracle import torch_ref as R
  s = spec(case)
  c, m, n, p = s['Config'], s['Model'], s['NerfMLP'], s['PropMLP']
  kw = dict(
      num_prop_samples=m['num_prop_samples'], num_nerf_samples=m['num_nerf_samples'], num_levels=m['num_levels'],
      raydist_fn='reciprocal' if m.get('raydist_fn') == '@jnp.reciprocal' else None,
      num_glo_features=m.get('num_glo_features', 0), num_transient_features=m.get('num_transient_features', 0),
      num_embeddings=m['num_embeddings'], opaque_background=m['opaque_background'],
      warp=n.get('warp_fn') == '@coord.contract',
      nerf_depth=n['net_depth'], nerf_width=n['net_width'], bottleneck_width=n['bottleneck_width'],
      width_viewdirs=n.get('net_width_viewdirs', 128), prop_depth=p['net_depth'], prop_width=p['net_width'],
      prop_disable_rgb=p['disable_rgb'])
  for k in ('data_loss_type', 'distortion_loss_mult', 'transient_type', 'patch_size', 'withmask_transient_weight',
            'grad_max_norm', 'grad_max_val', 'robustnerf_inlier_quantile', 'max_steps', 'charb_padding', 'data_coarse_loss_mult',
            'interlevel_loss_mult', 'disable_multiscale_loss', 'lr_init', 'lr_final', 'lr_delay_steps', 'lr_delay_mult', 'adam_eps'):
    if k in c:
      kw[k] = c[k]
  cfg = R.ModelCfg(**kw)
  if 'weight_decay_mults' in c:
    cfg.weight_decay_mults = dict(c['weight_decay_mults'])
  for k in ('adam_beta1', 'adam_beta2'):
    if k in c:
      setattr(cfg, k, c[k])
  if 'net_width_transient' in n:
    cfg.transient_width = n['net_width_transient']
  cfg.rgb_premultiplier, cfg.rgb_bias = n.get('rgb_premultiplier', 1.), n.get('rgb_bias', 0.)
  if finetune_optimizer:
    for dst, src in (('lr_init', 'finetune_lr_init'), ('lr_final', 'finetune_lr_final'), ('max_steps', 'finetune_max_steps'),
                     ('lr_delay_steps', 'finetune_lr_delay_steps'), ('lr_delay_mult', 'finetune_lr_delay_mult'),
                     ('adam_beta1', 'finetune_adam_beta1'), ('adam_beta2', 'finetune_adam_beta2'), ('adam_eps', 'finetune_adam_eps')):
      setattr(cfg, dst, c[src])
  return cfg


def gin_lines(case):
  """spec -> gin binding lines for the product's configs.parse_config_files_and_bindings."""
  s = spec(case)
  lines = []
  for scope in ('Config', 'Model', 'NerfMLP', 'PropMLP', 'ImplicitMask'):
    for k, v in s.get(scope, {}).items():
      lines.append(f'{scope}.{k} = {v if isinstance(v, str) and v.startswith("@") else repr(v)}')
  return lines

"""Config + a gin-subset reader (gin / absl are not assumed to be installed).

Mirrors reference MipNeRF360/internal/configs.py:45-204: the `Config` dataclass fields and
defaults, `load_config`, and the `--gin_configs/--gin_bindings` inputs.  The reader accepts the
forms the reference's 19 gin files and shell scripts use: `Name.attr = <python literal>`,
`Name.attr = @module.fn` (e.g. @jnp.reciprocal, @coord.contract), comments and blank lines.
Bindings for `Model`, `NerfMLP`, `PropMLP` are kept in a registry that `models.Model` consults,
like gin's configurables (models.py:46, :553-560).
"""
import ast
import dataclasses
import os
from typing import Any, Callable, Optional, Tuple

_INT32_MAX = 2**31 - 1


@dataclasses.dataclass
class Config:
  """Configuration flags for everything (field-for-field with configs.py:47-184)."""
  dataset_loader: str = 'llff'
  batch_size: int = 16384
  patch_size: int = 1
  patch_dilation: int = 1
  image_num_per_batch: int = 64
  factor: int = 0
  load_alphabetical: bool = True
  forward_facing: bool = False
  render_path: bool = False
  llffhold: int = 8
  llff_use_all_images_for_training: bool = False
  gc_every: int = 10000
  disable_multiscale_loss: bool = False
  randomized: bool = True
  near: float = 2.
  far: float = 6.
  checkpoint_dir: Optional[str] = None
  render_dir: Optional[str] = None
  data_dir: Optional[str] = None
  vocab_tree_path: Optional[str] = None
  render_chunk_size: int = 16384
  num_showcase_images: int = 5
  deterministic_showcase: bool = True
  vis_num_rays: int = 16
  vis_decimate: int = 0
  transient_type: Optional[str] = None
  max_steps: int = 250000
  early_exit_steps: Optional[int] = None
  checkpoint_every: int = 25000
  print_every: int = 100
  train_render_every: int = 5000
  data_loss_type: str = 'charb'
  charb_padding: float = 0.001
  data_loss_mult: float = 1.0
  data_coarse_loss_mult: float = 0.
  interlevel_loss_mult: float = 1.0
  weight_decay_mults: Any = dataclasses.field(default_factory=dict)
  lr_init: float = 0.002
  lr_final: float = 0.00002
  lr_delay_steps: int = 512
  lr_delay_mult: float = 0.01
  adam_beta1: float = 0.9
  adam_beta2: float = 0.999
  adam_eps: float = 1e-6
  grad_max_norm: float = 0.001
  grad_max_val: float = 0.
  distortion_loss_mult: float = 0.01
  enable_render_zero_glo: bool = False
  enable_render_zero_tra: bool = False
  robustnerf_inlier_quantile: float = 0.5
  robustnerf_inlier_quantile_static: float = 0.95
  robustnerf_smoothed_filter_size: int = 3
  robustnerf_smoothed_inlier_quantile: float = 0.5
  robustnerf_inner_patch_size: int = 8
  robustnerf_inner_patch_inlier_quantile: float = 0.4
  nerfw_beta_loss_mult: float = 1.0
  nerfw_beta_loss_bias: float = 3.0
  nerfw_density_loss_mult: float = 0.01
  hanerf_mask_size_loss_mult_min: float = 6.0e-3
  hanerf_mask_size_loss_mult_max: float = 5.0e-2
  hanerf_mask_size_loss_mult_k: float = 1.0e-3
  withmask_transient_weight: float = 0
  static_mask_dir_name: str = 'static_masks'
  finetune_enable: bool = False
  finetune_max_steps: int = 5000
  finetune_batch_size: int = 16384
  finetune_patch_size: int = 1
  finetune_patch_dilation: int = 1
  finetune_image_num_per_batch: int = 64
  finetune_lr_init: float = 0.005
  finetune_lr_final: float = 0.0005
  finetune_lr_delay_steps: int = 500
  finetune_lr_delay_mult: float = 0.01
  finetune_adam_beta1: float = 0.9
  finetune_adam_beta2: float = 0.999
  finetune_adam_eps: float = 1e-8
  eval_only_once: bool = True
  eval_save_output: bool = True
  eval_save_ray_data: bool = False
  eval_render_interval: int = 1
  eval_dataset_limit: int = _INT32_MAX
  eval_quantize_metrics: bool = True
  eval_crop_borders: int = 0
  render_video_fps: int = 60
  render_video_crf: int = 18
  render_path_frames: int = 120
  z_variation: float = 0.
  z_phase: float = 0.
  render_dist_percentile: float = 0.5
  render_dist_curve_fn: Any = 'jnp.log'
  render_path_file: Optional[str] = None
  render_job_id: int = 0
  render_num_jobs: int = 1
  render_resolution: Optional[Tuple[int, int]] = None
  render_focal: Optional[float] = None
  render_camtype: Optional[str] = None
  render_embed_idx: Optional[int] = None
  render_spherical: bool = False
  render_save_async: bool = True
  render_spline_keyframes: Optional[str] = None
  render_spline_n_interp: int = 30
  render_spline_degree: int = 5
  render_spline_smoothness: float = .03


class Ref:
  """`@module.fn` reference in a gin file (e.g. @jnp.reciprocal)."""

  def __init__(self, name):
    self.name = name
    self.__name__ = name.split('.')[-1]

  def __repr__(self):
    return '@' + self.name

  def __eq__(self, other):
    return isinstance(other, Ref) and other.name == self.name

  def __hash__(self):
    return hash(self.name)


# scope -> {attr: value}; cleared by clear_config()
_BINDINGS = {}
_KNOWN_REFS = {
    'jnp.reciprocal', 'jnp.log', 'jnp.log1p', 'jnp.exp', 'jnp.sqrt', 'jnp.square', 'jax.nn.relu',
    'jax.nn.softplus', 'jax.nn.silu', 'math.safe_exp', 'coord.contract'
}


def clear_config():
  _BINDINGS.clear()


def _parse_value(text):
  text = text.strip()
  if text.startswith('@'):
    name = text[1:].rstrip('()')
    if name not in _KNOWN_REFS:
      raise ValueError(f'unknown gin reference @{name}')
    return Ref(name)
  return ast.literal_eval(text)


def parse_binding(line):
  line = line.split('#', 1)[0].strip() if not ("'" in line or '"' in line) else _strip_comment(line)
  if not line:
    return None
  if '=' not in line:
    raise ValueError(f'unsupported gin statement: {line!r}')
  lhs, rhs = line.split('=', 1)
  lhs = lhs.strip()
  if '.' not in lhs:
    raise ValueError(f'unsupported gin binding target: {lhs!r}')
  scope, attr = lhs.rsplit('.', 1)
  scope = scope.split('/')[-1]            # drop gin scopes such as train/Config
  return scope, attr, _parse_value(rhs)


def _strip_comment(line):
  out, quote = [], None
  for ch in line:
    if quote:
      if ch == quote:
        quote = None
    elif ch in '\'"':
      quote = ch
    elif ch == '#':
      break
    out.append(ch)
  return ''.join(out).strip()


def parse_config_files_and_bindings(gin_configs=None, gin_bindings=None, skip_unknown=True):
  """Counterpart of gin.parse_config_files_and_bindings (configs.py:197-198)."""
  for path in gin_configs or []:
    with open(path) as f:
      for line in f:
        b = parse_binding(line)
        if b:
          _BINDINGS.setdefault(b[0], {})[b[1]] = b[2]
  for line in gin_bindings or []:
    for part in line.split('\n'):
      b = parse_binding(part)
      if b:
        _BINDINGS.setdefault(b[0], {})[b[1]] = b[2]


def bindings(scope):
  return dict(_BINDINGS.get(scope, {}))


def config_str():
  lines = []
  for scope in sorted(_BINDINGS):
    for k, v in sorted(_BINDINGS[scope].items()):
      lines.append(f'{scope}.{k} = {v!r}')
    lines.append('')
  return '\n'.join(lines)


def make_config(**overrides):
  """Config() with the currently parsed `Config.*` bindings applied (gin.configurable behaviour)."""
  fields = {f.name for f in dataclasses.fields(Config)}
  kw = {}
  for k, v in bindings('Config').items():
    if k not in fields:
      raise ValueError(f'Config has no field {k!r}')
    kw[k] = v
  kw.update(overrides)
  return Config(**kw)


def load_config(gin_configs=None, gin_bindings=None, save_config=True):
  """Load the config, and optionally checkpoint it (configs.py:195-204)."""
  parse_config_files_and_bindings(gin_configs, gin_bindings, skip_unknown=True)
  config = make_config()
  if save_config and config.checkpoint_dir:
    os.makedirs(config.checkpoint_dir, exist_ok=True)
    with open(os.path.join(config.checkpoint_dir, 'config.gin'), 'w') as f:
      f.write(config_str())
  return config

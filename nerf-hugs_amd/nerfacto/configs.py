"""Configuration entry for the nerfacto path: the reference's yml files (nerfacto/configs/*.yml: a `base:` section read by
train.py and a `model:` section that becomes models/nerfacto.py ModelConfig) -> NerfactoConfig."""
from .model import NerfactoConfig

# nerfacto/configs/phototourism_nerfacto_base.yml, `model:` section + the `base:` keys the model / optimizer read
# (BASELINE.json configs[4]: "Nerfacto + hash-grid encoding, Phototourism trevi_fountain, 16384 rays").  Restated as
# data; `load_yml` reads the file itself.
PHOTOTOURISM_NERFACTO_BASE = dict(
    hidden_dim=256, geo_feat_dim=64, hidden_dim_color=256, base_res=16, max_res=8192, log2_hashmap_size=21, features_per_level=2,
    use_appearance_embedding=True, appearance_embedding_dim=48, eval_embedding='original', opaque_background=True,
    num_nerf_samples_per_ray=128, num_proposal_samples_per_ray=(512, 256), num_proposal_iterations=2,
    proposal_net_args_list=[dict(base_res=16, hidden_dim=64, log2_hashmap_size=17, features_per_level=2, num_levels=5, max_res=512),
                            dict(base_res=16, hidden_dim=64, log2_hashmap_size=17, features_per_level=2, num_levels=7, max_res=2048)],
    proposal_initial_sampler='uniform', proposal_histogram_padding=0.005, proposal_weights_anneal_max_num_iters=10000,
    rgb_loss_type='mse', distortion_loss_mult=0.001,
    bound=2.0, enable_scene_contraction=False, patch_size=16, lr_init=1e-2, lr_final=1e-3, lr_decay_mult=1e-8, warmup_steps=500,
    num_steps=25000, opt_betas=(0.9, 0.999), opt_eps=1e-15)

_BASE_KEYS = ('bound', 'enable_scene_contraction', 'patch_size', 'lr_init', 'lr_final', 'lr_decay_mult', 'warmup_steps', 'num_steps',
              'opt_betas', 'opt_eps')
_IGNORED_MODEL_KEYS = ('enable_tcnn_mlp',)


def yml_to_kwargs(doc):
  """{'base': {...}, 'model': {...}} (a parsed reference yml) -> NerfactoConfig keyword arguments."""
  base, model = doc.get('base', {}), dict(doc.get('model', {}))
  if base.get('model_type', 'nerfacto') != 'nerfacto':
    raise ValueError(f"model_type {base.get('model_type')!r}: this path builds the nerfacto model")
  if model.get('enable_tcnn_mlp', False):
    raise NotImplementedError('enable_tcnn_mlp: True (tiny-cuda-nn fused MLPs): the shipped ymls select the nn.Linear form')
  # (round 5: use_same_proposal_network and density_activation = 'softplus' are built -- NerfactoConfig checks their values)
  kw = {k: v for k, v in model.items() if k not in _IGNORED_MODEL_KEYS}
  for k in ('num_proposal_samples_per_ray',):
    if k in kw:
      kw[k] = tuple(kw[k])
  for k in _BASE_KEYS:
    if k in base:
      kw[k] = tuple(base[k]) if isinstance(base[k], list) else base[k]
  if 'bound' in kw:
    kw['bound'] = float(kw['bound'])
  if kw.get('enable_scene_contraction'):
    kw['bound'] = 2.0          # datasets/base.py:88-89: with scene contraction the dataset's bound is 2 whatever the yml says
  return kw


def load_yml(path):
  import yaml
  with open(path) as f:
    return NerfactoConfig(**yml_to_kwargs(yaml.safe_load(f)))

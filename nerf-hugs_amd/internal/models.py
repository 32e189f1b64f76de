"""Model surface of the hot path (reference MipNeRF360/internal/models.py): `Model`, `construct_model`,
`render_image`, with the reference's call signatures and result dictionaries.  Arithmetic lives in
csrc/*.hip (see engine.py for the launch sequence)."""
import math

import numpy as np
import torch

from . import configs
from . import engine as _engine
from . import random as hrandom
from . import stepfun
from . import utils

_MODEL_DEFAULTS = dict(
    num_prop_samples=64, num_nerf_samples=32, num_levels=3, bg_intensity_range=(1., 1.), anneal_slope=10,
    stop_level_grad=True, use_viewdirs=True, raydist_fn=None, ray_shape='cone', disable_integration=False,
    single_jitter=True, dilation_multiplier=0.5, dilation_bias=0.0025, num_glo_features=0, num_transient_features=0,
    num_embeddings=3500, near_anneal_rate=None, near_anneal_init=0.95, resample_padding=0.0, use_gpu_resampling=False,
    opaque_background=False, beta_min=0.03)


class Model:
  """A mip-NeRF 360 model containing all MLPs (models.py:46-330).  Attributes are the gin-configurable
  fields of the reference; `Model.*`, `NerfMLP.*`, `PropMLP.*` gin bindings are applied at construction."""

  def __init__(self, config=None, compute_dtype=None, **overrides):
    self.config = config
    attrs = dict(_MODEL_DEFAULTS)
    for k, v in {**configs.bindings('Model'), **overrides}.items():
      if k not in attrs:
        raise ValueError(f'Model has no attribute {k!r}')
      attrs[k] = v
    for k, v in attrs.items():
      setattr(self, k, v)
    tt = None if config is None else config.transient_type
    if tt in [None, 'withmask', 'robustnerf']:
      assert self.num_transient_features == 0
    elif tt in ['nerfw', 'hanerf']:
      assert self.num_transient_features > 0
    else:
      raise ValueError()
    if self.ray_shape not in ('cone', 'cylinder'):
      raise ValueError('ray_shape must be \'cone\' or \'cylinder\'')
    # use_gpu_resampling only picks between two XLA formulations of the same inverse-CDF lookup (stepfun.py:153-161,
    # math.py:101-127); there is one HIP formulation, so the flag is accepted and has no effect.
    if not self.stop_level_grad:
      raise NotImplementedError('stop_level_grad=False (gradients through the resampling step) is not built')
    # models.py:246-259: a fixed intensity, or -- with a range -- a uniform draw per ray and channel in training (rng given) and
    # the midpoint when rendering deterministically
    self.bg_random = float(self.bg_intensity_range[0]) != float(self.bg_intensity_range[1])
    self.bg_intensity = 0.5 * (float(self.bg_intensity_range[0]) + float(self.bg_intensity_range[1]))
    rd = self.raydist_fn
    name = None if rd is None else getattr(rd, 'name', rd)
    self.raydist = None if name is None else (name[4:] if str(name).startswith('jnp.') else name)
    if self.raydist not in stepfun.RAYDIST:        # coord.py:84-90 knows reciprocal / log / exp / sqrt / square (+ 'piecewise')
      raise NotImplementedError(f"raydist_fn {rd!r}: coord.py:78-90 knows None, 'piecewise' and @jnp.reciprocal / log / exp / sqrt / square")
    # models.py:104-105: NerfMLP(disable_transient=(transient_type != 'nerfw')), PropMLP(disable_transient=True)
    self.nerf_spec = _engine.MLPSpec('NerfMLP_0', False, self.num_glo_features,
                                     self.num_transient_features if tt == 'nerfw' else 0, use_viewdirs=self.use_viewdirs,
                                     **configs.bindings('NerfMLP'))
    self.prop_spec = _engine.MLPSpec('PropMLP_0', True, self.num_glo_features, use_viewdirs=self.use_viewdirs,
                                     **configs.bindings('PropMLP'))
    self.specs = [self.nerf_spec, self.prop_spec]
    self.mask_spec = None
    if tt == 'hanerf':      # models.py:107: ImplicitMask() is constructed after the two MLPs
      # ImplicitMask is not @gin.configurable in the reference (models.py:651): `ImplicitMask.*` bindings are unknown
      # names that gin skips, so the mask MLP always has its class defaults (4 x 256, deg_coord 10)
      self.mask_spec = _engine.MaskSpec(self.num_transient_features)
    self.layout = _engine.ParamLayout(self.specs + ([self.mask_spec] if self.mask_spec else []), self.num_embeddings,
                                      self.num_glo_features, self.num_transient_features)
    self.compute_dtype = compute_dtype or 'bf16'
    self._engine = None

  # -- engine / params ---------------------------------------------------------------------------------
  def engine(self, device='cuda'):
    if self._engine is None:
      self._engine = _engine.Engine(self, device, self.compute_dtype)
    return self._engine

  def init(self, seed, device='cuda', flax_rng=None):
    """Random-init parameters: he_uniform kernels, zero biases (models.py:372,432-433), N(0,1/G) GLO rows
    (flax nn.Embed default).  Returns the flat fp32 buffer.  `seed`: an int (torch's generator: the distributions, not the
    stream, of the reference) or a jax key from internal/random.py -- then the reference's own stream, `model.init(rng, ...)`
    of models.py:348-356, through init_flax."""
    if hrandom.is_key(seed):
      return self.init_flax(seed, flax_rng or __import__('os').environ.get('HUGS_FLAX_RNG', 'lazy'))
    g = torch.Generator().manual_seed(int(seed))
    flat = torch.zeros(self.layout.size, dtype=torch.float32)
    for lf in self.layout.leaves:
      v = self.layout.view(flat, lf['path'])
      if lf['path'][-1] == 'kernel':
        lim = math.sqrt(6.0 / lf['shape'][0])
        v.copy_(((torch.rand(lf['shape'], generator=g, dtype=torch.float64) * 2 - 1) * lim).float())
      elif lf['path'][-1] == 'embedding':
        v.copy_((torch.randn(lf['shape'], generator=g, dtype=torch.float64) / math.sqrt(lf['shape'][1])).float())
    return flat.to(device)

  def init_flax(self, key, variant='lazy'):
    """flax's initialisation stream on the device: every Dense kernel = jax.nn.initializers.he_uniform()(k, [in, out]) =
    random.uniform(k, shape, minval=-1) * sqrt(3 * 2 / in) (variance_scaling(2, 'fan_in', 'uniform')), every nn.Embed table =
    random.normal(k, shape) * sqrt(1 / features) (variance_scaling(1, 'fan_in', 'normal', out_axis=0)), biases zero, with
    k = random.flax_param_key(rng, module path, 1): a function of the parameter's PATH, not of the creation order.
    jax's non-partitionable threefry (the default before jax 0.5) as everywhere in internal/random.py.  PARITY UNPINNED: flax
    and jax are absent from the reference tree and this image; the folding rule is selectable (random.flax_param_key)."""
    flat = torch.zeros(self.layout.size, dtype=torch.float32, device=key.device)
    for lf in self.layout.leaves:
      v = self.layout.view(flat, lf['path'])
      if lf['path'][-1] == 'kernel':
        k = hrandom.flax_param_key(key, lf['path'][:-1], 1, variant)
        scale = np.sqrt(np.float32(3.0) * np.float32(2.0 / lf['shape'][0]))      # jnp: float32 variance, float32 sqrt
        v.copy_(hrandom.uniform(k, lf['shape'], -1.0, 1.0) * float(np.float32(scale)))
      elif lf['path'][-1] == 'embedding':
        k = hrandom.flax_param_key(key, lf['path'][:-1], 1, variant)
        v.copy_(hrandom.normal(k, lf['shape']) * float(np.sqrt(np.float32(1.0 / lf['shape'][1]))))
    return flat

  def variables(self, flat):
    """flax-style view tree {'params': {...}} sharing memory with `flat` (reference TrainState.params)."""
    return Variables(self.layout.tree(flat), flat)

  def load_variables(self, flat, variables):
    """Copy a flax-style nested dict (reference TrainState.params layout) into the flat buffer."""
    for lf in self.layout.leaves:
      d = variables['params'] if 'params' in variables else variables
      for k in lf['path']:
        d = d[k]
      self.layout.view(flat, lf['path']).copy_((torch.from_numpy(np.array(d)) if not torch.is_tensor(d) else d).to(flat.device))
    return flat

  def level_jitter(self, rng, N):
    """The reference's consumption of `rng` inside Model.__call__: per level one split for the sampler key
    (models.py:196) -- spent by random.uniform(key, [N, d], maxval=max_jitter) (stepfun.py:207-209) -- and one for
    the MLP key (models.py:230; density / bottleneck noise are 0 in every shipped gin, the split still advances).
    Returns (stepfun.Jitter, rng after the last level)."""
    out = stepfun.Jitter()
    out.mlp_keys, out.bg_rgbs = [], []
    for l in range(self.num_levels):
      S = self.num_prop_samples if l < self.num_levels - 1 else self.num_nerf_samples
      key, rng = hrandom.split(rng)
      out.append(hrandom.uniform(key, (N, 1 if self.single_jitter else S), maxval=stepfun.sample_u(S, True)[1]))
      key, rng = hrandom.split(rng)
      out.mlp_keys.append(key)
      if self.bg_random:                         # models.py:256-261: key, rng = random_split(rng); uniform(key, [N, 3], lo, hi)
        key, rng = hrandom.split(rng)
        out.bg_rgbs.append(hrandom.uniform(key, (N, 3), float(self.bg_intensity_range[0]), float(self.bg_intensity_range[1])))
    return out, rng

  def has_noise(self):
    """density_noise / bottleneck_noise > 0 on either MLP (models.py:378-381) or a random background (models.py:246-261): draws that
    hang off the per-level keys, which only level_jitter (not the fused chain kernel) keeps."""
    return self.bg_random or any(sp.density_noise > 0 or sp.bottleneck_noise > 0 for sp in (self.prop_spec, self.nerf_spec) if sp is not None)

  def step_jitter(self, rng, N):
    """train_step's `rng, key = random.split(rng)` (train_utils.py:408) followed by level_jitter(key, N), as one launch.
    Returns (stepfun.Jitter, the advanced rng)."""
    Ss = [self.num_prop_samples if l < self.num_levels - 1 else self.num_nerf_samples for l in range(self.num_levels)]
    d = [1 if self.single_jitter else S for S in Ss]
    outs, rng = hrandom.step_jitter(rng, [N * dl for dl in d], [stepfun.sample_u(S, True)[1] for S in Ss])
    return stepfun.Jitter([o.view(N, dl) for o, dl in zip(outs, d)]), rng

  # -- forward -----------------------------------------------------------------------------------------
  def apply(self, variables, rng, rays, train_frac, compute_extras, zero_glo=False, zero_tra=False,
            refresh_weights=True):
    """Model.__call__ (models.py:74-330).  `variables`: flat buffer or the tree from variables().
    `rng`: None (deterministic), a jax-style key (internal/random.py: the reference's exact split / uniform stream)
    or a torch.Generator on the GPU (philox draws).  rays: utils.Rays with any leading
    shape.  Returns (renderings, ray_history) as lists of dicts, one per level."""
    flat = variables if torch.is_tensor(variables) else variables.flat
    eng = self.engine(flat.device)
    lead = rays.origins.shape[:-1]
    r = rays_to_dict(rays, flat.device)
    N = r['origins'].shape[0]
    u01 = None
    if hrandom.is_key(rng):
      u01, _ = self.level_jitter(rng, N)
    elif rng is not None:
      shape = (N,) if self.single_jitter else None
      u01 = [torch.rand(shape if shape else (N, (self.num_prop_samples if l < self.num_levels - 1 else self.num_nerf_samples)),
                        generator=rng, device=flat.device) for l in range(self.num_levels)]
    # Ragged chunks (the reference renders any ray count, e.g. the last chunk of an odd-sized image): the GEMM
    # tiles need rays*samples to be a multiple of 128 on every level, so the batch is padded with copies of its
    # last ray and the padding is cut off the outputs.
    align = max(128 // math.gcd(128, S_) for S_ in (self.num_prop_samples, self.num_nerf_samples))
    Np = max(align, (N + align - 1) // align * align)
    if Np != N:
      if N == 0:
        raise ValueError('Model.apply needs at least one ray')
      pad = lambda x: torch.cat([x, x[-1:].expand((Np - N,) + tuple(x.shape[1:]))]).contiguous()
      r = {k: pad(v) for k, v in r.items()}
      if u01 is not None:
        padded = [pad(u) for u in u01]
        if isinstance(u01, stepfun.Jitter):
          # the draws were made for the N real rays (level_jitter above); the noise draws inside the engine are sized by
          # n_real too, so a padded batch consumes the reference's stream unchanged
          pj = stepfun.Jitter(padded)
          pj.mlp_keys = u01.mlp_keys
          pj.bg_rgbs = None if not u01.bg_rgbs else [pad(b) for b in u01.bg_rgbs]
          u01 = pj
        else:
          u01 = padded
    take = lambda buf, *tail: buf.reshape((Np,) + tail)[:N].clone().reshape(lead + tail)
    if refresh_weights:
      eng.refresh_weights(flat)
    levels = eng.forward(flat, r, float(train_frac), u01, compute_extras, zero_glo, zero_tra, n_real=N if Np != N else None)
    implicit_mask = None
    if self.mask_spec is not None:
      implicit_mask = eng.mask_forward(flat, r, Np, zero_tra)['mask'][:N].clone().reshape(lead + (1,))
    renderings, history = [], []
    n = 0 if self.config is None else min(self.config.vis_num_rays, N)
    for lv in levels:
      S = lv['S']
      rend = {'rgb': take(lv['rgb_out'], 3)}
      rgb_s = lv['rgb'].reshape(Np, S, 3) if lv['rgb'] is not None else torch.zeros(Np, S, 3, device=flat.device)
      if compute_extras:
        e = lv['extras']
        rend['acc'] = take(e[:, 0].contiguous())
        rend['distance_mean'] = take(e[:, 1].contiguous())
        rend['distance_median'] = take(e[:, 2].contiguous())
        rend['distance_percentile_5'] = take(e[:, 3].contiguous())
        rend['distance_percentile_95'] = take(e[:, 4].contiguous())
        rend['ray_sdist'] = lv['sdist'][:n].clone()
        rend['ray_weights'] = lv['weights'][:n].clone()
        rend['ray_rgbs'] = rgb_s[:n].clone()
      hist = dict(density=take(lv['density'], S), rgb=take(rgb_s, S, 3), sdist=take(lv['sdist'], S + 1),
                  weights=take(lv['weights'], S))
      if lv.get('dens_t') is not None:       # NeRF-W (models.py:285-307, 545-548)
        for k in ('rgb_combined', 'rgb_static', 'rgb_transient'):
          rend[k] = take(lv[k], 3)
        rend['uncertainty'] = take(lv['uncertainty'], 1)
        hist.update(density_transient=take(lv['dens_t'], S), rgb_transient=take(lv['rgb_t'], S, 3),
                    uncertainty=take(lv['unc'], S, 1))
      renderings.append(rend)
      history.append(hist)
    if implicit_mask is not None:
      renderings[-1]['implicit_mask'] = implicit_mask        # models.py:327-328
    if compute_extras:
      # proposal levels show the final average colour (models.py:314-325)
      final_rgb = (renderings[-1]['ray_rgbs'] * renderings[-1]['ray_weights'][..., None]).sum(-2)
      for rr in renderings[:-1]:
        rr['ray_rgbs'] = final_rgb[:, None, :].expand(rr['ray_rgbs'].shape).clone()
    return renderings, history

  __call__ = apply


class Variables(dict):
  """Nested parameter dict that remembers the flat buffer it views."""

  def __init__(self, tree, flat):
    super().__init__(tree)
    self.flat = flat


def rays_to_dict(rays, device):
  f = lambda x, dt=torch.float32: x.reshape(-1, x.shape[-1]).to(device=device, dtype=dt).contiguous()
  return dict(pix_coords=f(rays.pix_coords), origins=f(rays.origins), directions=f(rays.directions), viewdirs=f(rays.viewdirs),
              radii=f(rays.radii).reshape(-1), lossmult=f(rays.lossmult).reshape(-1),
              static_mask=f(rays.static_mask).reshape(-1), near=f(rays.near).reshape(-1), far=f(rays.far).reshape(-1),
              embed_idx=f(rays.embed_idx, torch.int32).reshape(-1))


def construct_model(rng, rays, config, compute_dtype=None, device='cuda'):
  """Construct a mip-NeRF 360 model (models.py:333-357).  `rng`: an int seed, or a jax key (internal/random.PRNGKey) for
  flax's own initialisation stream (Model.init_flax).  Returns (model, variables) with variables = the flat fp32 parameter
  buffer on `device` (use model.variables(flat) for the flax-style tree)."""
  model = Model(config=config, compute_dtype=compute_dtype)
  return model, model.init(rng if rng is not None else 0, device)


def render_image(render_fn, rays, rng, config, verbose=True):
  """Render all the pixels of an image in test mode (models.py:568-649).

  render_fn(rng, chunk_rays) -> (renderings, ray_history) with a leading device axis of size 1 per process
  (world_size > 1: each rank renders its slice of every chunk and the slices are all-gathered)."""
  import torch.distributed as dist
  height, width = rays.origins.shape[:2]
  num_rays = height * width
  rays = rays.map(lambda r: r.reshape((num_rays, -1)))
  world = dist.get_world_size() if dist.is_initialized() else 1
  rank = dist.get_rank() if dist.is_initialized() else 0
  chunks = []
  idx0s = range(0, num_rays, config.render_chunk_size)
  for i_chunk, idx0 in enumerate(idx0s):
    if verbose and i_chunk % max(1, len(idx0s) // 10) == 0:
      print(f'Rendering chunk {i_chunk}/{len(idx0s)-1}')
    chunk_rays = rays.map(lambda r: r[idx0:idx0 + config.render_chunk_size])
    actual = chunk_rays.origins.shape[0]
    rem = actual % world
    padding = (world - rem) if rem != 0 else 0
    if padding:
      chunk_rays = chunk_rays.map(lambda r: torch.cat([r, r[-1:].expand(padding, -1)], 0))   # mode='edge'
    per = chunk_rays.origins.shape[0] // world
    local = chunk_rays.map(lambda r: r[rank * per:(rank + 1) * per])
    chunk_renderings, _ = render_fn(rng, utils.shard(local))
    # v[0] of the reference's all-gathered [ndev, n/ndev, ...] leaves == the full chunk
    chunk_renderings = [{k: utils.unshard(v, 0 if k.startswith('ray_') else padding) for k, v in r.items()}
                        for r in chunk_renderings]
    chunk_rendering = dict(chunk_renderings[-1])
    for k in chunk_renderings[0]:
      if k.startswith('ray_'):
        chunk_rendering[k] = [r[k] for r in chunk_renderings]
    chunks.append(chunk_rendering)
  rendering = {}
  for k in chunks[0]:
    if k.startswith('ray_'):
      rendering[k] = [torch.cat([c[k][l] for c in chunks]) for l in range(len(chunks[0][k]))]
    else:
      z = torch.cat([c[k] for c in chunks])
      rendering[k] = z.reshape((height, width) + tuple(z.shape[1:]))
  keys = [k for k in rendering if k.startswith('ray_')]
  if keys:
    n = rendering[keys[0]][0].shape[0]
    # models.py:644: random.permutation(random.PRNGKey(0), num_rays)[:vis_num_rays], jax's own shuffle
    dev = rendering[keys[0]][0].device
    ray_idx = hrandom.permutation(hrandom.PRNGKey(0, dev), n)[:config.vis_num_rays]
    for k in keys:
      rendering[k] = [r[ray_idx.to(r.device)] for r in rendering[k]]
  return rendering

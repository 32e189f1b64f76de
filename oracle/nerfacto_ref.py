"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (PyTorch, autograd for gradients) of the reference's NERFACTO training path for its base /
withmask configurations (BASELINE config 5).  Paths under /root/reference/nerfacto/.

Pinned: the pure-torch half -- `sample` / `sample_intervals` (utils/ray_utils.py:112-223), `density_to_weight`
(:226-249, INCLUDING its `euclidean_bins[..., 1:] - euclidean_bins[..., :1]` deltas: distance from the FIRST bin edge,
not interval widths), `render_features` / `render_depth` (:300-345), `lossfun_outer` / `interlevel_loss` /
`lossfun_distortion` (utils/loss_utils.py:7-86, EPS = 1e-7), `spatial_distortion_norm2`, `trunc_exp`
(models/custom_functions.py) -- by vectors recorded from importing those modules (tests/golden/gen_nerfacto_fixtures.py
-> ref_nerfacto.npz).  PARITY UNPINNED: the hash grid and spherical harmonics (tiny-cuda-nn, un-vendored:
oracle/hashgrid_ref.py) and therefore the fields' outputs; the field / model wiring (models/nerfacto.py:286-414,
:818-876, :971-988; the `enable_tcnn_mlp: False` torch-Linear form that configs/phototourism_nerfacto_base.yml uses)
is restated from the cited lines.
"""
import math

import numpy as np
import torch

from . import hashgrid_ref as HG

EPS_LOSS = 1.0e-7      # loss_utils.py:5


class Cfg:
  """models/nerfacto.py:18-114 ModelConfig + the Model.__init__ arguments, defaults of the dataclass."""

  def __init__(self, **kw):
    self.bound = 2.0
    self.enable_scene_contraction = False
    self.num_levels, self.base_res, self.max_res, self.log2_hashmap_size, self.features_per_level = 16, 16, 2048, 19, 2
    self.hidden_dim, self.geo_feat_dim, self.hidden_dim_color = 64, 15, 64
    self.num_layers, self.num_layers_color = 2, 3
    self.use_appearance_embedding, self.appearance_embedding_dim, self.num_embedding = False, 32, 3500
    self.eval_embedding = 'average'
    self.num_proposal_samples_per_ray, self.num_nerf_samples_per_ray = (256, 96), 48
    self.num_proposal_iterations = 2
    self.proposal_net_args_list = [dict(hidden_dim=16, log2_hashmap_size=17, num_levels=5, max_res=128),
                                   dict(hidden_dim=16, log2_hashmap_size=17, num_levels=5, max_res=256)]
    self.proposal_initial_sampler = 'uniform'
    self.proposal_histogram_padding = 0.01
    self.use_proposal_weight_anneal, self.proposal_weights_anneal_slope = True, 10.0
    self.proposal_weights_anneal_max_num_iters = 1000
    self.use_single_jitter, self.opaque_background = True, False
    self.rgb_loss_type, self.rgb_charb_loss_padding, self.rgb_loss_mult = 'mse', 0.001, 1.0
    self.interlevel_loss_mult, self.distortion_loss_mult = 1.0, 0.002
    self.transient_type, self.withmask_transient_weight = None, 0.
    self.robustnerf_inlier_quantile, self.robustnerf_smoothed_filter_size = 0.8, 3
    self.robustnerf_smoothed_inlier_quantile, self.robustnerf_inner_patch_size = 0.5, 8
    self.robustnerf_inner_patch_inlier_quantile, self.patch_size = 0.4, 16
    self.rgb_bias = 0.
    # nerfacto.py:36 density_activation ('trunc_exp' | 'softplus' = F.softplus(raw + density_bias), density_bias = -1: :660,890) and
    # :66 use_same_proposal_network (network 0 evaluates every proposal level, :334)
    self.density_activation, self.density_bias, self.use_same_proposal_network = 'trunc_exp', -1., False
    self.use_transient_embedding, self.transient_embedding_dim = False, 16
    self.num_levels_implicit, self.base_res_implicit, self.max_res_implicit = 8, 16, 1024
    self.log2_hashmap_size_implicit, self.features_per_level_implicit, self.hidden_dim_implicit = 17, 2, 128
    self.hanerf_mask_size_loss_mult_min, self.hanerf_mask_size_loss_mult_max, self.hanerf_mask_size_loss_mult_k = 6e-3, 5e-2, 1e-3
    for k, v in kw.items():
      if not hasattr(self, k):
        raise AttributeError(k)
      setattr(self, k, v)

  def prop_args(self, i):
    a = dict(num_levels=8, base_res=16, max_res=1024, log2_hashmap_size=18, features_per_level=2, hidden_dim=64, num_layers=2)
    a.update(self.proposal_net_args_list[min(i, len(self.proposal_net_args_list) - 1)])
    return a


# ---- custom_functions.py ------------------------------------------------------------------------------------------
def spatial_distortion_norm2(x):
  """custom_functions.py:17-24."""
  eps = torch.finfo(x.dtype).eps
  m2 = torch.sum(x**2, dim=-1, keepdim=True).clamp_min(eps)
  return torch.where(m2 <= 1, x, ((2 * torch.sqrt(m2) - 1) / m2) * x)


class _TruncExp(torch.autograd.Function):
  """custom_functions.py:38-52: exp forward, gradient exp(clamp(x, -15, 15))."""

  @staticmethod
  def forward(ctx, x):
    ctx.save_for_backward(x)
    return torch.exp(x)

  @staticmethod
  def backward(ctx, g):
    (x,) = ctx.saved_tensors
    return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def density_activation(cfg, raw):
  """nerfacto.py:702-710 / 910-918."""
  if cfg.density_activation == 'trunc_exp':
    return trunc_exp(raw)
  if cfg.density_activation == 'softplus':
    return torch.nn.functional.softplus(raw + cfg.density_bias)
  raise NotImplementedError()


# ---- ray_utils.py -----------------------------------------------------------------------------------------------------
def spacing_fns(name):
  """nerfacto.py:231-241."""
  if name == 'piecewise':
    return (lambda x: torch.where(x < 1, x / 2, 1 - 1 / (2 * x))), (lambda x: torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x)))
  if name == 'uniform':
    return (lambda x: x), (lambda x: x)
  if name == 'reciprocal':
    return torch.reciprocal, torch.reciprocal
  raise ValueError(f'Sampler does not support {name}. ')


def sample(spacing_bins, weights, anneal, padding, num_samples, u01, single_jitter, deterministic_center=True):
  """ray_utils.py:112-205.  u01: None (perturb=False) or the U[0,1) draws [N,1] / [N,num_samples] torch.rand returned."""
  dtype = spacing_bins.dtype
  eps = torch.finfo(dtype).eps
  logit = torch.where(spacing_bins[..., 1:] > spacing_bins[..., :-1], anneal * torch.log(weights + padding),
                      torch.full_like(weights, -math.inf))
  logit = logit.clone()
  logit[(logit <= -math.inf).all(dim=-1)] = 1
  pdf = torch.softmax(logit, dim=-1)
  cdf = torch.cumsum(pdf[..., :-1], dim=-1).clamp_max(1)
  cdf = torch.cat([torch.zeros_like(pdf[..., :1]), cdf, torch.ones_like(pdf[..., :1])], dim=-1)
  if u01 is not None:
    u_max = eps + (1. - eps) / num_samples
    max_jitter = (1. - u_max) / (num_samples - 1) - eps
    u = torch.linspace(0, 1. - u_max, num_samples, dtype=dtype) + u01 * max_jitter
  else:
    pad = 1 / (2 * num_samples)
    u = torch.linspace(pad, 1. - pad - eps, num_samples, dtype=dtype) if deterministic_center else \
        torch.linspace(0, 1. - eps, num_samples, dtype=dtype)
    u = u.expand(cdf.shape[:-1] + (num_samples,))
  u = u.contiguous()
  inds = torch.searchsorted(cdf.contiguous(), u, side='right')
  below = torch.clamp(inds - 1, 0, spacing_bins.shape[-1] - 1)
  above = torch.clamp(inds, 0, spacing_bins.shape[-1] - 1)
  c0, b0 = torch.gather(cdf, -1, below), torch.gather(spacing_bins, -1, below)
  c1, b1 = torch.gather(cdf, -1, above), torch.gather(spacing_bins, -1, above)
  t = torch.clip(torch.nan_to_num((u - c0) / (c1 - c0), 0), 0, 1)
  return b0 + t * (b1 - b0)


def sample_intervals(spacing_bins, weights, anneal, padding, num_samples, u01, single_jitter, domain):
  """ray_utils.py:208-231."""
  centers = sample(spacing_bins, weights, anneal, padding, num_samples, u01, single_jitter, True)
  mid = (centers[..., 1:] + centers[..., :-1]) / 2
  first = (2 * centers[..., :1] - mid[..., :1]).clamp_min(domain[0])
  last = (2 * centers[..., -1:] - mid[..., -1:]).clamp_max(domain[1])
  return torch.cat([first, mid, last], dim=-1)


def density_to_weight(densities, euclidean_bins, directions, opaque_background=False):
  """ray_utils.py:234-257 -- with the reference's deltas: every edge minus the FIRST edge."""
  deltas = (euclidean_bins[..., 1:] - euclidean_bins[..., :1]) * torch.linalg.norm(directions[..., None, :], dim=-1)
  dd = densities * deltas
  if opaque_background:
    dd = torch.cat([dd[..., :-1], torch.full_like(dd[..., -1:], math.inf)], dim=-1)
  alphas = 1 - torch.exp(-dd)
  trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], dim=-1)], dim=-1))
  return torch.nan_to_num(alphas * trans), alphas, trans


def render_features(weights, features, bg, require_detach=False):
  """ray_utils.py:300-314."""
  w = weights[..., None].detach() if require_detach else weights[..., None]
  feats = torch.sum(w * features, dim=-2)
  if bg is not None:
    feats = feats + bg * (1. - torch.sum(w, dim=-2)).clamp_min(0)
  return feats


def render_depth(weights, euclidean_bins):
  """ray_utils.py:340-347."""
  steps = (euclidean_bins[..., 1:] + euclidean_bins[..., :-1]) / 2
  acc = torch.sum(weights, dim=-1)
  acc = torch.where(acc > 0, acc, torch.full_like(acc, torch.finfo(acc.dtype).eps))
  depth = torch.sum(weights * steps, dim=-1) / acc
  return torch.clip(depth, 0.0, steps.max())


# ---- loss_utils.py ------------------------------------------------------------------------------------------------------
def outer(t0s, t0e, t1s, t1e, y1):
  """loss_utils.py:7-31."""
  cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
  lo = torch.clamp(torch.searchsorted(t1s.contiguous(), t0s.contiguous(), side='right') - 1, 0, y1.shape[-1] - 1)
  hi = torch.clamp(torch.searchsorted(t1e.contiguous(), t0e.contiguous(), side='right'), 0, y1.shape[-1] - 1)
  return torch.take_along_dim(cy1[..., 1:], hi, dim=-1) - torch.take_along_dim(cy1[..., :-1], lo, dim=-1)


def lossfun_outer(t, w, t_env, w_env):
  """loss_utils.py:34-47."""
  w_outer = outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
  return torch.clip(w - w_outer, min=0)**2 / (w + EPS_LOSS)


def interlevel_loss(weights_list, bins_list):
  """loss_utils.py:50-62."""
  c, w = bins_list[-1].detach(), weights_list[-1].detach()
  return sum(torch.mean(lossfun_outer(c, w, b, ww)) for b, ww in zip(bins_list[:-1], weights_list[:-1]))


def lossfun_distortion(t, w):
  """loss_utils.py:66-77."""
  ut = (t[..., 1:] + t[..., :-1]) / 2
  dut = torch.abs(ut[..., :, None] - ut[..., None, :])
  return torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1) + torch.sum(w**2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3


# ---- fields (nerfacto.py:643-1008, enable_tcnn_mlp=False form) ----------------------------------------------------------
def mask_grid_spec(cfg):
  """ImplicitMask's 2-D grid (nerfacto.py:1036-1047)."""
  growth = np.exp((np.log(cfg.max_res_implicit) - np.log(cfg.base_res_implicit)) / (cfg.num_levels_implicit - 1))
  offs, ress, scales = HG.level_table(cfg.num_levels_implicit, cfg.base_res_implicit, growth, cfg.log2_hashmap_size_implicit, dims=2)
  return dict(offsets=offs, resolutions=ress, scales=scales, F=cfg.features_per_level_implicit, n_entries=int(offs[-1]),
              out_dim=cfg.num_levels_implicit * cfg.features_per_level_implicit)


def implicit_mask(cfg, P, coords, tra):
  """ImplicitMask.forward (nerfacto.py:1080-1091, the nn.Linear form :1062-1075): grid(coords) | embedding -> (Linear+relu) x 2
  -> Linear(1) -> sigmoid."""
  x = _HashGridFn.apply(P['table'], coords, mask_grid_spec(cfg))
  h = torch.cat([x, tra], dim=-1)
  h = torch.relu(h @ P['m0'] + P['mb0'])
  h = torch.relu(h @ P['m1'] + P['mb1'])
  return torch.sigmoid(h @ P['m2'] + P['mb2'])


def grid_spec(num_levels, base_res, max_res, log2_hashmap_size, features_per_level):
  growth = float(np.exp((np.log(max_res) - np.log(base_res)) / (num_levels - 1))) if num_levels > 1 else 1.0
  offs, ress, scales = HG.level_table(num_levels, base_res, growth, log2_hashmap_size)
  return dict(offsets=offs, resolutions=ress, scales=scales, F=features_per_level, n_entries=int(offs[-1]),
              out_dim=num_levels * features_per_level)


class _HashGridFn(torch.autograd.Function):
  """hashgrid_ref forward / table-gradient as an autograd node (positions get no gradient)."""

  @staticmethod
  def forward(ctx, table, x01, spec):
    ctx.spec, ctx.x = spec, x01.detach().numpy().astype(np.float32)
    out = HG.hashgrid_forward(ctx.x, table.detach().numpy(), spec['offsets'], spec['resolutions'], spec['scales'], spec['F'])
    return torch.from_numpy(out).to(table.dtype)

  @staticmethod
  def backward(ctx, g):
    s = ctx.spec
    gt = HG.hashgrid_backward(ctx.x, g.numpy(), s['n_entries'], s['offsets'], s['resolutions'], s['scales'], s['F'])
    return torch.from_numpy(gt).to(g.dtype), None, None


def normalize_positions(cfg, positions):
  """nerfacto.py:822-829 / 975-982: (contract ->) [0,1]^3, selector, zeroed outside."""
  if cfg.enable_scene_contraction:
    p = (spatial_distortion_norm2(positions) + 2.0) / 4.0
  else:
    p = (positions + cfg.bound) / (2 * cfg.bound)
  sel = ((p >= 0.0) & (p <= 1.0)).all(dim=-1)
  return p * sel[..., None], sel


def prop_density(cfg, P, i, positions):
  """HashMLPDensityField.density (nerfacto.py:971-988): grid -> Linear -> relu -> Linear(1) -> trunc_exp, x selector."""
  a = cfg.prop_args(i)
  spec = grid_spec(a['num_levels'], a['base_res'], a['max_res'], a['log2_hashmap_size'], a['features_per_level'])
  p, sel = normalize_positions(cfg, positions)
  x = _HashGridFn.apply(P['table'], p, spec)
  x = torch.relu(x @ P['w0'] + P['b0'])
  raw = x @ P['w1'] + P['b1']
  return density_activation(cfg, raw) * sel[..., None]


def field_forward(cfg, P, positions, viewdirs, app):
  """NerfactoField.forward (nerfacto.py:818-876), no transient branch."""
  spec = grid_spec(cfg.num_levels, cfg.base_res, cfg.max_res, cfg.log2_hashmap_size, cfg.features_per_level)
  p, sel = normalize_positions(cfg, positions)
  x = _HashGridFn.apply(P['table'], p, spec)
  x = torch.relu(x @ P['w0'] + P['b0']) @ P['w1'] + P['b1']
  raw, geo = x[..., :1], x[..., 1:]
  density = density_activation(cfg, raw) * sel[..., None]
  d = torch.from_numpy(HG.sh4(((viewdirs + 1.0) / 2.0).detach().numpy())).to(positions.dtype)
  h = torch.cat([d, geo] + ([app] if app is not None else []), dim=-1)
  h = torch.relu(h @ P['c0'] + P['cb0'])
  h = torch.relu(h @ P['c1'] + P['cb1'])
  rgb = torch.sigmoid(h @ P['c2'] + P['cb2'] + cfg.rgb_bias)
  return rgb, density


def init_params(cfg, seed=0, dtype=torch.float32):
  """kaiming_uniform_ weights (nerfacto.py:789-791; a = 0 -> bound sqrt(6 / fan_in)), nn.Linear default biases
  U(+-1/sqrt(fan_in)), tcnn-style U(+-1e-4) tables, N(0,1) embeddings (nn.Embedding)."""
  g = torch.Generator().manual_seed(seed)

  def lin(fi, fo):
    w = (torch.rand(fi, fo, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / fi)
    b = (torch.rand(fo, generator=g, dtype=torch.float64) * 2 - 1) / math.sqrt(fi)
    return w.to(dtype), b.to(dtype)

  def table(spec):
    return ((torch.rand(spec['n_entries'], spec['F'], generator=g, dtype=torch.float64) * 2 - 1) * 1e-4).to(dtype)
  P = {}
  for i in range(1 if cfg.use_same_proposal_network else cfg.num_proposal_iterations):      # nerfacto.py:191-214
    a = cfg.prop_args(i)
    spec = grid_spec(a['num_levels'], a['base_res'], a['max_res'], a['log2_hashmap_size'], a['features_per_level'])
    w0, b0 = lin(spec['out_dim'], a['hidden_dim']); w1, b1 = lin(a['hidden_dim'], 1)
    P[f'prop{i}'] = dict(table=table(spec), w0=w0, b0=b0, w1=w1, b1=b1)
  spec = grid_spec(cfg.num_levels, cfg.base_res, cfg.max_res, cfg.log2_hashmap_size, cfg.features_per_level)
  app = cfg.appearance_embedding_dim if cfg.use_appearance_embedding else 0
  w0, b0 = lin(spec['out_dim'], cfg.hidden_dim); w1, b1 = lin(cfg.hidden_dim, 1 + cfg.geo_feat_dim)
  c0, cb0 = lin(16 + cfg.geo_feat_dim + app, cfg.hidden_dim_color)
  c1, cb1 = lin(cfg.hidden_dim_color, cfg.hidden_dim_color); c2, cb2 = lin(cfg.hidden_dim_color, 3)
  P['field'] = dict(table=table(spec), w0=w0, b0=b0, w1=w1, b1=b1, c0=c0, cb0=cb0, c1=c1, cb1=cb1, c2=c2, cb2=cb2)
  if app:
    P['appearance'] = torch.randn(cfg.num_embedding, app, generator=g, dtype=torch.float64).to(dtype)
  if cfg.transient_type == 'hanerf':
    T, H, ms = cfg.transient_embedding_dim, cfg.hidden_dim_implicit, mask_grid_spec(cfg)
    P['transient'] = torch.randn(cfg.num_embedding, T, generator=g, dtype=torch.float64).to(dtype)
    m0, mb0 = lin(ms['out_dim'] + T, H); m1, mb1 = lin(H, H); m2, mb2 = lin(H, 1)
    P['mask'] = dict(table=table(ms), m0=m0, mb0=mb0, m1=m1, mb1=mb1, m2=m2, mb2=mb2)
  return P


def anneal_of(cfg, curr_step):
  """nerfacto.py:289-297."""
  if not cfg.use_proposal_weight_anneal:
    return 1.0
  f = float(np.clip(curr_step / cfg.proposal_weights_anneal_max_num_iters, 0, 1))
  s = cfg.proposal_weights_anneal_slope
  return (s * f) / ((s - 1) * f + 1)


def forward_rays(cfg, P, rays, curr_step, u01, training=True):
  """Model.forward_rays (nerfacto.py:286-414) in training mode.  rays: origin, direction, viewdir [N,3], near, far [N,1],
  embed_idx [N,1] int, bg_rgb [N,3].  u01: None or one [N,1] draw tensor per level."""
  fwd, inv = spacing_fns(cfg.proposal_initial_sampler)
  s_near, s_far = fwd(rays['near']), fwd(rays['far'])
  s_to_t = lambda s: inv(s * s_far + (1 - s) * s_near)
  anneal = anneal_of(cfg, curr_step)
  bins = torch.cat([torch.zeros_like(rays['near']), torch.ones_like(rays['far'])], dim=-1)
  weights = torch.ones_like(rays['near'])
  out, wl, bl = {}, [], []
  L = cfg.num_proposal_iterations
  for lvl in range(L + 1):
    is_prop = lvl < L
    ns = cfg.num_proposal_samples_per_ray[lvl] if is_prop else cfg.num_nerf_samples_per_ray
    with torch.no_grad():
      bins = sample_intervals(bins, weights, anneal, cfg.proposal_histogram_padding, ns,
                              None if u01 is None else u01[lvl], cfg.use_single_jitter, (0., 1.))
    ebins = s_to_t(bins)
    t_mid = (ebins[..., 1:] + ebins[..., :-1]) / 2
    pos = rays['origin'][:, None, :] + rays['direction'][:, None, :] * t_mid[..., None]
    N, S = t_mid.shape
    if is_prop:
      net = 0 if cfg.use_same_proposal_network else lvl                                       # nerfacto.py:334
      dens = prop_density(cfg, P[f'prop{net}'], net, pos.reshape(-1, 3)).reshape(N, S)
      rgb = None
    else:
      vd = rays['viewdir'][:, None, :].expand_as(pos).reshape(-1, 3)
      app = None
      if cfg.use_appearance_embedding:
        if training or cfg.eval_embedding == 'original':       # Model.get_embedding nerfacto.py:266-284
          emb = P['appearance'][rays['embed_idx'][:, 0].long()]
        elif cfg.eval_embedding == 'average':
          emb = torch.ones(N, P['appearance'].shape[-1], dtype=P['appearance'].dtype) * P['appearance'].mean(dim=0)
        else:
          emb = torch.zeros(N, P['appearance'].shape[-1], dtype=P['appearance'].dtype)
        app = emb[:, None, :].expand(N, S, -1).reshape(N * S, -1)
      rgb, dens = field_forward(cfg, P['field'], pos.reshape(-1, 3), vd, app)
      rgb, dens = rgb.reshape(N, S, 3), dens.reshape(N, S)
    weights, _, _ = density_to_weight(dens, ebins, rays['direction'], cfg.opaque_background)
    wl.append(weights); bl.append(bins)
    sfx = f'_prop_{lvl}' if is_prop else ''
    if rgb is not None:
      out['rgb'] = render_features(weights, rgb, rays['bg_rgb'])
    out[f'depth{sfx}'] = render_depth(weights, ebins)
    out[f'accumulation{sfx}'] = torch.sum(weights, dim=-1)
    out[f'density{sfx}'], out[f'ebins{sfx}'] = dens, ebins
  if cfg.transient_type == 'hanerf':          # nerfacto.py:403-408
    T = P['transient']
    if training or cfg.eval_embedding == 'original':
      tra = T[rays['embed_idx'][:, 0].long()]
    elif cfg.eval_embedding == 'average':
      tra = torch.ones(rays['embed_idx'].shape[0], T.shape[-1], dtype=T.dtype) * T.mean(dim=0)
    else:
      tra = torch.zeros(rays['embed_idx'].shape[0], T.shape[-1], dtype=T.dtype)
    out['implicit_mask'] = implicit_mask(cfg, P['mask'], rays['coord'], tra)
  out['weights_list'], out['spacing_bins_list'] = wl, bl
  return out


def get_robustnerf_mask(cfg, errors, curr_threshold):
  """utils/loss_utils.py:88-150.  errors [n, P, P, 3]; curr_threshold: extra_infos['inlier_threshold'] (1.0 before the
  first step).  Returns (mask [n,P,P,1], info dict incl. the NEXT threshold)."""
  eps = 1e-3
  err = errors.mean(-1, keepdim=True)
  P = err.shape[1]
  assert cfg.robustnerf_inner_patch_size <= min(err.shape[1:3]), 'patch_size must be larger than robustnerf_inner_patch_size.'
  info = {'inlier_threshold': torch.quantile(err, cfg.robustnerf_inlier_quantile)}
  inl = (err < curr_threshold).to(err.dtype)
  info['is_inlier_loss'] = inl.mean()
  f = cfg.robustnerf_smoothed_filter_size
  # F.conv2d(padding='same'): f-1 zeros in total per axis, (f-1)//2 in front, the rest behind
  lo_, hi_ = (f - 1) // 2, f - 1 - (f - 1) // 2
  nb = torch.nn.functional.conv2d(torch.nn.functional.pad(inl.permute(0, 3, 1, 2), (lo_, hi_, lo_, hi_)),
                                  torch.ones(1, 1, f, f, dtype=err.dtype) / (f * f)).permute(0, 2, 3, 1)
  nb = (nb > 1. - cfg.robustnerf_smoothed_inlier_quantile).to(err.dtype)
  info['has_inlier_neighbors'] = nb.mean()
  ip = cfg.robustnerf_inner_patch_size
  h0, w0 = (err.shape[1] - ip) // 2, (err.shape[2] - ip) // 2
  inner = torch.zeros_like(err)
  inner[:, h0:h0 + ip, w0:w0 + ip, :] = 1
  patch = (inl.mean(dim=(1, 2), keepdim=True) > 1. - cfg.robustnerf_inner_patch_inlier_quantile).to(err.dtype) * inner
  info['is_inlier_patch'] = patch.mean()
  mask = (patch + nb + inl > eps).to(err.dtype)
  info['robust_mask'] = mask.mean()
  return mask, info


def loss_fn(cfg, out, gt_rgb, static_mask=None, inlier_threshold=1.0, curr_step=0, is_finetune=False):
  """Loss.forward (nerfacto.py:598-640) for transient_type None / 'withmask' / 'robustnerf' (rays in whole
  cfg.patch_size^2 patches, patch-major) / 'hanerf'; is_finetune selects compute_data_loss whatever the type (:606)."""
  resid_sq = (out['rgb'] - gt_rgb)**2
  dl = resid_sq if cfg.rgb_loss_type == 'mse' else torch.sqrt(resid_sq + cfg.rgb_charb_loss_padding**2)
  info = {}
  tt = None if is_finetune else cfg.transient_type
  extra = 0.
  if tt == 'hanerf':                             # compute_hanerf_loss nerfacto.py:560-596
    msm = max(cfg.hanerf_mask_size_loss_mult_min, cfg.hanerf_mask_size_loss_mult_max * np.exp(-curr_step * cfg.hanerf_mask_size_loss_mult_k))
    m = out['implicit_mask']
    info['implicit_mask'] = m.mean().detach()
    extra = msm * (m**2).mean()
    info['mask_size_loss'] = extra.detach()
    rgb_loss = cfg.rgb_loss_mult * ((1 - m) * dl).mean()
    info['mse'] = resid_sq.mean().detach()
  elif tt == 'robustnerf':       # compute_robustnerf_loss nerfacto.py:492-527
    P = cfg.patch_size
    mask, rinfo = get_robustnerf_mask(cfg, resid_sq.detach().reshape(-1, P, P, 3), inlier_threshold)
    info.update(rinfo)
    lm = mask.reshape(-1, 1).expand_as(resid_sq)
    den = lm.sum().clamp_min(torch.finfo(lm.dtype).eps)
    rgb_loss = cfg.rgb_loss_mult * ((lm * dl).sum() / den)
    info['mse'] = ((lm * resid_sq).sum() / den).detach()
  elif tt == 'withmask':
    sm = (static_mask >= 0.5).to(gt_rgb.dtype)
    lm = (sm + (1 - sm) * cfg.withmask_transient_weight).expand_as(resid_sq)
    den = lm.sum().clamp_min(torch.finfo(lm.dtype).eps)
    rgb_loss = cfg.rgb_loss_mult * ((lm * dl).sum() / den)
    info['mse'] = ((lm * resid_sq).sum() / den).detach()
  else:
    rgb_loss = cfg.rgb_loss_mult * dl.mean()
    info['mse'] = resid_sq.mean().detach()
  loss = rgb_loss + extra
  info['rgb_loss'] = rgb_loss.detach()
  if cfg.interlevel_loss_mult > 0:
    il = cfg.interlevel_loss_mult * interlevel_loss(out['weights_list'], out['spacing_bins_list'])
    loss = loss + il; info['interlevel_loss'] = il.detach()
  if cfg.distortion_loss_mult > 0:
    dl_ = cfg.distortion_loss_mult * torch.mean(lossfun_distortion(out['spacing_bins_list'][-1], out['weights_list'][-1]))
    loss = loss + dl_; info['distortion_loss'] = dl_.detach()
  return loss, info


def lr_factor(step, lr_init, lr_final, lr_delay_mult, warmup_steps, max_steps):
  """utils/lr_scheduler_utils.py:6-27 (the LambdaLR factor)."""
  if step < warmup_steps:
    return lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / warmup_steps, 0), 1))
  t = min(max((step - warmup_steps) / (max_steps - warmup_steps), 0), 1)
  return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t) / lr_init

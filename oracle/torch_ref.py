"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (PyTorch, dtype-generic: float32 for parity, float64 for gradient
checks) of the floating-point half of the reference's Mip-NeRF 360 per-ray path;
the bit-exact sampling half lives in stepfun_ref.c.  Every function cites the
reference lines it follows (paths under /root/reference/MipNeRF360/).  Gradients
come from torch.autograd, i.e. they are independent of the hand-written HIP
backward kernels they check.

Pinned twice: (i) against tests/golden/ref_leaves.npz -- the
reference's own leaf modules internal/{math,stepfun,render,coord,geopoly}.py
executed under a numpy stand-in for jax -- by tests/test_oracle_vs_reference.py;
(ii) against tests/golden/ref_model.npz -- the reference's internal/models.py and
internal/train_utils.py themselves, executed unmodified under bookkeeping
stand-ins for flax.linen / gin / optax (tests/golden/gen_model_fixtures.py):
Model.__call__ outputs at every level, every inverse-CDF index, loss terms,
clip_gradients, train_step stats and float64 finite differences of the
reference's own loss_fn -- by tests/test_oracle_vs_reference_model.py.
What stays "parity unpinned": third-party arithmetic that is not in
/root/reference or this image (optax.adam, flax initialisers, XLA's float32
summation order); Adam below follows optax's published formula.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import math

import numpy as np
import torch

from . import cstepfun

EPS = float(np.finfo(np.float32).eps)


# ----------------------------------------------------------------------------
# geopoly.py:21-124 -- geodesic basis (init-time constant, float64 numpy)
# ----------------------------------------------------------------------------
def _sq_dist(m0, m1=None):
  m1 = m0 if m1 is None else m1
  return np.maximum(0, (m0**2).sum(0)[:, None] + (m1**2).sum(0)[None, :] - 2 * m0.T @ m1)


def generate_basis(base_shape='icosahedron', v=2, remove_symmetries=True, eps=1e-4):
  """geopoly.py:78-124. Returns [n,3]; the model uses the transpose [3,n] (models.py:393-396)."""
  if base_shape == 'icosahedron':
    a = (np.sqrt(5) + 1) / 2
    verts = np.array([(-1, 0, a), (1, 0, a), (-1, 0, -a), (1, 0, -a), (0, a, 1), (0, a, -1),
                      (0, -a, 1), (0, -a, -1), (a, 1, 0), (-a, 1, 0), (a, -1, 0),
                      (-a, -1, 0)]) / np.sqrt(a + 2)
    faces = np.array([(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1),
                      (8, 3, 10), (5, 3, 8), (5, 2, 3), (2, 7, 3), (7, 10, 3), (7, 6, 10),
                      (7, 11, 6), (11, 0, 6), (0, 1, 6), (6, 1, 10), (9, 0, 11), (9, 11, 2),
                      (9, 2, 5), (7, 2, 11)])
  elif base_shape == 'octahedron':
    import itertools
    verts = np.array([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0)], float)
    corners = np.array(list(itertools.product([-1, 1], repeat=3)))
    pairs = np.argwhere(_sq_dist(corners.T, verts.T) == 2)
    faces = np.sort(np.reshape(pairs[:, 1], [3, -1]).T, 1)
  else:
    raise ValueError(f'base_shape {base_shape} not supported')
  # tesselate (geopoly.py:45-75)
  wts = np.array([(i, j, v - i - j) for i in range(v + 1) for j in range(v + 1 - i)]) / v
  allv = []
  for f in faces:
    nv = wts @ verts[f, :]
    allv.append(nv / np.sqrt((nv**2).sum(1, keepdims=True)))
  allv = np.concatenate(allv, 0)
  assign = np.array([np.min(np.argwhere(d <= eps)) for d in _sq_dist(allv.T)])
  verts = allv[np.unique(assign), :]
  if remove_symmetries:
    match = _sq_dist(verts.T, -verts.T) < eps
    verts = verts[np.any(np.triu(match), 1), :]
  return verts[:, ::-1].copy()


# ----------------------------------------------------------------------------
# math.py
# ----------------------------------------------------------------------------
def safe_sin(x):
  """math.py:26-38: sin(where(|x| < 100pi, x, x mod 100pi))."""
  t = 100 * math.pi
  return torch.sin(torch.where(x.abs() < t, x, torch.remainder(x, t)))


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
  """math.py:57-98 (float64 host arithmetic)."""
  if lr_delay_steps > 0:
    delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(
        0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
  else:
    delay = 1.
  t = min(max(step / max_steps, 0), 1)
  return delay * math.exp(t * (math.log(lr_final) - math.log(lr_init)) + math.log(lr_init))


def mse_to_psnr(mse):
  """image.py:28-30."""
  return -10. / math.log(10.) * torch.log(mse)


# ----------------------------------------------------------------------------
# coord.py
# ----------------------------------------------------------------------------
def contract(x):
  """coord.py:21-27."""
  n2 = torch.clamp((x**2).sum(-1, keepdim=True), min=EPS)
  return torch.where(n2 <= 1, x, ((2 * torch.sqrt(n2) - 1) / n2) * x)


def contract_track_linearize(mean, cov):
  """coord.py:39-60 with fn=contract: cov' = J cov J^T, J the Jacobian of contract at mean.

  The reference obtains J via jax.linearize; for contract J is closed-form:
  |x|<=1: I;  else  s I + (ds/dn2 * 2) x x^T with s=(2n-1)/n^2, n=|x|.
  """
  n2 = torch.clamp((mean**2).sum(-1, keepdim=True), min=EPS)
  n = torch.sqrt(n2)
  s = (2 * n - 1) / n2
  # d s / d n2 = (1/n)/n2 - (2n-1)/n2^2
  ds = (1 / n) / n2 - (2 * n - 1) / (n2 * n2)
  eye = torch.eye(3, dtype=mean.dtype)
  J = s[..., None] * eye + (2 * ds)[..., None] * mean[..., :, None] * mean[..., None, :]
  inside = (n2 <= 1)[..., None]
  J = torch.where(inside, eye.expand_as(J), J)
  return contract(mean), J @ cov @ J.transpose(-1, -2)


RAYDIST = {None: 0, 'reciprocal': 1, 'log': 2, 'exp': 3, 'sqrt': 4, 'square': 5, 'piecewise': 6}


def s_to_t(s, near, far, raydist):
  """coord.py:63-99: fn_inv(s * fn(far) + (1 - s) * fn(near)) with the (fn, fn_inv) pairs of :84-90."""
  fwd, inv = {None: (lambda x: x, lambda x: x), 'reciprocal': (torch.reciprocal, torch.reciprocal),
              'log': (torch.log, torch.exp), 'exp': (torch.exp, torch.log), 'sqrt': (torch.sqrt, torch.square),
              'square': (torch.square, torch.sqrt),
              'piecewise': (lambda x: torch.where(x < 1, .5 * x, 1 - .5 / x), lambda x: torch.where(x < .5, 2 * x, .5 / (1 - x)))}[raydist]
  return inv(s * fwd(far) + (1 - s) * fwd(near))


def lift_and_diagonalize(mean, cov, basis):
  """coord.py:129-133, basis [3,21]."""
  return mean @ basis, (basis * (cov @ basis)).sum(-2)


def integrated_pos_enc(mean, var, min_deg, max_deg):
  """coord.py:102-126: [sin block (scale-major), sin(x+pi/2) block] * exp(-var/2)."""
  scales = torch.tensor([2.0**i for i in range(min_deg, max_deg)], dtype=mean.dtype)
  shape = mean.shape[:-1] + (-1,)
  sm = (mean[..., None, :] * scales[:, None]).reshape(shape)
  sv = (var[..., None, :] * scales[:, None]**2).reshape(shape)
  x = torch.cat([sm, sm + 0.5 * math.pi], -1)
  v = torch.cat([sv, sv], -1)
  return torch.exp(-0.5 * v) * safe_sin(x)


def pos_enc(x, min_deg, max_deg, append_identity=True):
  """coord.py:136-147 (plain sin, no safe_sin)."""
  scales = torch.tensor([2.0**i for i in range(min_deg, max_deg)], dtype=x.dtype)
  sx = (x[..., None, :] * scales[:, None]).reshape(x.shape[:-1] + (-1,))
  four = torch.sin(torch.cat([sx, sx + 0.5 * math.pi], -1))
  return torch.cat([x, four], -1) if append_identity else four


# ----------------------------------------------------------------------------
# render.py
# ----------------------------------------------------------------------------
def cast_rays(tdist, origins, directions, radii, ray_shape='cone'):
  """render.py:103-127 -> :44-78 (stable cone) / :81-100 (cylinder) -> :21-41 (diag=False)."""
  t0, t1 = tdist[..., :-1], tdist[..., 1:]
  d = directions
  if ray_shape == 'cone':
    mu = (t0 + t1) / 2
    hw = (t1 - t0) / 2
    denom = torch.clamp(3 * mu**2 + hw**2, min=EPS)
    t_mean = mu + (2 * mu * hw**2) / denom
    t_var = (hw**2) / 3 - (4 / 15) * hw**4 * (12 * mu**2 - hw**2) / denom**2
    r_var = (mu**2) / 4 + (5 / 12) * hw**2 - (4 / 15) * (hw**4) / denom
    r_var = r_var * radii**2
  elif ray_shape == 'cylinder':
    t_mean = (t0 + t1) / 2
    r_var = (radii**2 / 4).expand_as(t_mean)
    t_var = (t1 - t0)**2 / 12
  else:
    raise ValueError('ray_shape must be \'cone\' or \'cylinder\'')
  mean = d[..., None, :] * t_mean[..., None]
  d_mag_sq = torch.clamp((d**2).sum(-1, keepdim=True), min=1e-10)
  d_outer = d[..., :, None] * d[..., None, :]
  eye = torch.eye(3, dtype=d.dtype)
  null_outer = eye - d[..., :, None] * (d / d_mag_sq)[..., None, :]
  cov = t_var[..., None, None] * d_outer[..., None, :, :] + \
      r_var[..., None, None] * null_outer[..., None, :, :]
  return mean + origins[..., None, :], cov


def compute_alpha_weights(density, tdist, dirs, opaque_background=False):
  """render.py:130-151."""
  t_delta = tdist[..., 1:] - tdist[..., :-1]
  delta = t_delta * torch.linalg.norm(dirs[..., None, :], dim=-1)
  dd = density * delta
  if opaque_background:
    dd = torch.cat([dd[..., :-1], torch.full_like(dd[..., -1:], float('inf'))], -1)
  alpha = 1 - torch.exp(-dd)
  trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], -1)], -1))
  return alpha * trans, alpha, trans


class _MaxZero(torch.autograd.Function):
  """jnp.maximum(0, x) including JAX's tie rule (gradient 1/2 at x == 0)."""

  @staticmethod
  def forward(ctx, x):
    ctx.save_for_backward(x)
    return torch.clamp(x, min=0)

  @staticmethod
  def backward(ctx, g):
    x, = ctx.saved_tensors
    return g * torch.where(x > 0, 1.0, torch.where(x == 0, 0.5, 0.0)).to(g.dtype)


def integrate_weights(w):
  """stepfun.py:131-150."""
  cw = torch.clamp(torch.cumsum(w[..., :-1], -1), max=1)
  z = torch.zeros_like(w[..., :1])
  return torch.cat([z, cw, z + 1], -1)


def _interp(x, xp, fp):
  """np.interp semantics per row (xp increasing): x[...,q], xp/fp[...,m]."""
  idx = torch.searchsorted(xp.contiguous(), x.contiguous(), right=True)  # count xp <= x
  i0 = torch.clamp(idx - 1, 0, xp.shape[-1] - 1)
  i1 = torch.clamp(idx, 0, xp.shape[-1] - 1)
  x0, x1 = torch.gather(xp, -1, i0), torch.gather(xp, -1, i1)
  f0, f1 = torch.gather(fp, -1, i0), torch.gather(fp, -1, i1)
  den = x1 - x0
  w = torch.where(den > 0, (x - x0) / torch.where(den > 0, den, torch.ones_like(den)),
                  torch.zeros_like(den))
  out = f0 + w * (f1 - f0)
  out = torch.where(x <= xp[..., :1], fp[..., :1].expand_as(out), out)
  out = torch.where(x >= xp[..., -1:], fp[..., -1:].expand_as(out), out)
  return out


def weighted_percentile(t, w, ps):
  """stepfun.py:298-308 (jnp.interp of ps/100 into the integrated weights)."""
  cw = integrate_weights(w)
  q = torch.tensor(ps, dtype=t.dtype) / 100
  return _interp(q.expand(cw.shape[:-1] + (len(ps),)), cw, t)


def volumetric_rendering(rgbs, weights, tdist, bg_rgbs, t_far, compute_extras):
  """render.py:185-244 (extras=None)."""
  out = {}
  acc = weights.sum(-1)
  bg_w = _MaxZero.apply(1 - acc[..., None])
  out['rgb'] = (weights[..., None] * rgbs).sum(-2) + bg_w * bg_rgbs
  if compute_extras:
    out['acc'] = acc
    t_mids = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    e = (weights * torch.log(t_mids)).sum(-1) / torch.clamp(acc, min=EPS)
    dm = torch.nan_to_num(torch.exp(e), nan=float('inf'))
    out['distance_mean'] = torch.minimum(torch.maximum(dm, tdist[..., 0]), tdist[..., -1])
    t_aug = torch.cat([tdist, t_far], -1)
    w_aug = torch.cat([weights, bg_w], -1)
    pct = weighted_percentile(t_aug, w_aug, [5, 50, 95])
    out['distance_percentile_5'] = pct[..., 0]
    out['distance_median'] = pct[..., 1]
    out['distance_percentile_95'] = pct[..., 2]
  return out


# ----------------------------------------------------------------------------
# stepfun.py losses
# ----------------------------------------------------------------------------
def searchsorted(a, v):
  """stepfun.py:30-53 (compare-matrix definition)."""
  i = torch.arange(a.shape[-1])
  ge = v[..., None, :] >= a[..., :, None]
  lo = torch.where(ge, i[:, None], i[:1, None]).max(-2).values
  hi = torch.where(~ge, i[:, None], i[-1:, None]).min(-2).values
  return lo, hi


def inner_outer(t0, t1, y1):
  """stepfun.py:64-77."""
  cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, -1)], -1)
  lo, hi = searchsorted(t1, t0)
  cy_lo, cy_hi = torch.gather(cy1, -1, lo), torch.gather(cy1, -1, hi)
  outer = cy_hi[..., 1:] - cy_lo[..., :-1]
  inner = torch.where(hi[..., :-1] <= lo[..., 1:], cy_lo[..., 1:] - cy_hi[..., :-1],
                      torch.zeros_like(outer))
  return inner, outer


def lossfun_outer(t, w, t_env, w_env):
  """stepfun.py:80-86."""
  _, w_outer = inner_outer(t, t_env, w_env)
  return torch.clamp(w - w_outer, min=0)**2 / (w + EPS)


def lossfun_distortion(t, w):
  """stepfun.py:266-276."""
  ut = (t[..., 1:] + t[..., :-1]) / 2
  dut = (ut[..., :, None] - ut[..., None, :]).abs()
  inter = (w * (w[..., None, :] * dut).sum(-1)).sum(-1)
  intra = (w**2 * (t[..., 1:] - t[..., :-1])).sum(-1) / 3
  return inter + intra


# ----------------------------------------------------------------------------
# models.py -- parameters, MLP, level loop
# ----------------------------------------------------------------------------
class ModelCfg:
  """The gin-visible knobs of Model / NerfMLP / PropMLP / Config used on the path
  (models.py:46-72, :359-391; configs.py:47-136)."""

  def __init__(self, **kw):
    self.num_prop_samples = 64
    self.num_nerf_samples = 32
    self.num_levels = 3
    self.bg_intensity = 1.0
    self.anneal_slope = 10.
    self.raydist_fn = None           # None | 'reciprocal' | 'log' | 'exp' | 'sqrt' | 'square' (coord.py:84-90)
    self.ray_shape = 'cone'
    self.single_jitter = True
    self.dilation_multiplier = 0.5
    self.dilation_bias = 0.0025
    self.num_glo_features = 0
    self.num_transient_features = 0   # HA-NeRF: TransientEmbed width (models.py:64)
    self.mask_depth, self.mask_width, self.mask_deg_coord = 4, 256, 10   # ImplicitMask (models.py:651-656)
    self.num_embeddings = 3500
    self.resample_padding = 0.0
    self.opaque_background = False
    self.warp = False                # MLP.warp_fn = @coord.contract
    self.nerf_depth, self.nerf_width = 8, 256
    self.prop_depth, self.prop_width = 8, 256
    self.prop_disable_rgb = False
    self.bottleneck_width = 256
    self.width_viewdirs = 128
    self.depth_viewdirs = 1                                  # models.py:365 net_depth_viewdirs (skip_layer_dir = 4 never triggers for <= 4)
    self.skip_layer = 4
    self.max_deg_point = 12
    self.basis_shape = 'icosahedron'
    self.basis_subdivisions = 2
    self.deg_view = 4
    self.density_bias = -1.
    self.rgb_padding = 0.001
    self.rgb_premultiplier, self.rgb_bias = 1., 0.          # models.py:380-381
    self.disable_integration = False                         # models.py:59
    self.use_viewdirs = True                                 # models.py:56
    # Config
    self.data_loss_type = 'charb'
    self.charb_padding = 0.001
    self.data_loss_mult = 1.0
    self.data_coarse_loss_mult = 0.
    self.interlevel_loss_mult = 1.0
    self.distortion_loss_mult = 0.01
    self.transient_type = None
    self.transient_depth, self.transient_width, self.beta_min = 4, 128, 0.03   # NeRF-W (models.py:71,367-368)
    self.nerfw_beta_loss_mult, self.nerfw_beta_loss_bias, self.nerfw_density_loss_mult = 1.0, 3.0, 0.01
    self.hanerf_mask_size_loss_mult_min = 6.0e-3
    self.hanerf_mask_size_loss_mult_max = 5.0e-2
    self.hanerf_mask_size_loss_mult_k = 1.0e-3
    self.withmask_transient_weight = 0.
    self.disable_multiscale_loss = False
    self.patch_size = 1
    self.robustnerf_inlier_quantile = 0.5
    self.robustnerf_smoothed_filter_size = 3
    self.robustnerf_smoothed_inlier_quantile = 0.5
    self.robustnerf_inner_patch_size = 8
    self.robustnerf_inner_patch_inlier_quantile = 0.4
    self.weight_decay_mults = {}      # {summarize_tree key: multiplier} (train_utils.py:444-447)
    self.grad_max_norm = 0.001
    self.grad_max_val = 0.
    self.lr_init, self.lr_final = 0.002, 0.00002
    self.lr_delay_steps, self.lr_delay_mult = 512, 0.01
    self.max_steps = 250000
    self.adam_beta1, self.adam_beta2, self.adam_eps = 0.9, 0.999, 1e-6
    for k, v in kw.items():
      if not hasattr(self, k):
        raise AttributeError(k)
      setattr(self, k, v)


def kubric_cfg(**kw):
  """configs/kubric_1024_base.gin:1-16."""
  base = dict(patch_size=16, data_loss_type='mse', distortion_loss_mult=0., opaque_background=True,
              prop_depth=4, prop_width=256, prop_disable_rgb=True, nerf_depth=8, nerf_width=1024)
  base.update(kw)
  return ModelCfg(**base)


def mlp_layer_dims(cfg, which):
  """Dense layer (fan_in, fan_out) list in flax creation order (models.py:432-519)."""
  depth, width = (cfg.nerf_depth, cfg.nerf_width) if which == 'nerf' else (cfg.prop_depth, cfg.prop_width)
  F = 2 * generate_basis(cfg.basis_shape, cfg.basis_subdivisions).shape[0] * cfg.max_deg_point
  dims, k = [], F
  for i in range(depth):
    dims.append((k, width))
    k = width + F if (i % cfg.skip_layer == 0 and i > 0) else width
  dims.append((k, 1))                                   # raw density
  disable_rgb = cfg.prop_disable_rgb if which == 'prop' else False
  if not disable_rgb:
    dims.append((k, cfg.bottleneck_width))              # bottleneck
    kv = cfg.bottleneck_width + 3 + 3 * 2 * cfg.deg_view + (cfg.num_glo_features if which == 'nerf' else 0)
    dims.append((kv, cfg.width_viewdirs))
    for _ in range(getattr(cfg, 'depth_viewdirs', 1) - 1):       # models.py:508-512: further Dense(net_width_viewdirs) + relu layers
      dims.append((cfg.width_viewdirs, cfg.width_viewdirs))
    dims.append((cfg.width_viewdirs, 3))
    if which == 'nerf' and cfg.transient_type == 'nerfw':        # models.py:521-539, created after the rgb head
      k = cfg.bottleneck_width + cfg.num_transient_features
      for _ in range(cfg.transient_depth):
        dims.append((k, cfg.transient_width))
        k = cfg.transient_width
      dims += [(k, 1), (k, 3), (k, 1)]                            # density_transient, rgb_transient, uncertainty
  return dims


def init_params(cfg, seed=20200823, dtype=torch.float32):
  """he_uniform kernels (limit sqrt(6/fan_in)), zero biases (models.py:372,432-433);
  GLO embedding ~ N(0, 1/G) (flax nn.Embed default)."""
  g = torch.Generator().manual_seed(seed)
  params = {}
  for name, which in (('NerfMLP_0', 'nerf'), ('PropMLP_0', 'prop')):
    mod = {}
    for i, (fi, fo) in enumerate(mlp_layer_dims(cfg, which)):
      lim = math.sqrt(6.0 / fi)
      k = (torch.rand(fi, fo, generator=g, dtype=torch.float64) * 2 - 1) * lim
      mod[f'Dense_{i}'] = {'kernel': k.to(dtype), 'bias': torch.zeros(fo, dtype=dtype)}
    params[name] = mod
  if cfg.transient_type == 'hanerf':
    mod, fi = {}, 2 + 4 * cfg.mask_deg_coord + cfg.num_transient_features
    for i in range(cfg.mask_depth + 1):
      fo = cfg.mask_width if i < cfg.mask_depth else 1
      k = (torch.rand(fi, fo, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / fi)
      mod[f'Dense_{i}'] = {'kernel': k.to(dtype), 'bias': torch.zeros(fo, dtype=dtype)}
      fi = fo
    params['ImplicitMask_0'] = mod
  if cfg.num_glo_features > 0:
    e = torch.randn(cfg.num_embeddings, cfg.num_glo_features, generator=g, dtype=torch.float64)
    params['GloEmbed_0'] = {'embedding': (e / math.sqrt(cfg.num_glo_features)).to(dtype)}
  if cfg.num_transient_features > 0:
    e = torch.randn(cfg.num_embeddings, cfg.num_transient_features, generator=g, dtype=torch.float64)
    params['TransientEmbed_0'] = {'embedding': (e / math.sqrt(cfg.num_transient_features)).to(dtype)}
  return {'params': params}


def implicit_mask_forward(cfg, mod, pix_coords, tra_vec, taps=None):
  """models.py:651-674 ImplicitMask: sigmoid(Dense(1)((Dense+relu)^depth([pos_enc(pix_coords,0,deg,True) | tra_vec])))."""
  x = torch.cat([pos_enc(pix_coords, 0, cfg.mask_deg_coord, True), tra_vec], -1)
  for i in range(cfg.mask_depth):
    pre = x @ mod[f'Dense_{i}']['kernel'] + mod[f'Dense_{i}']['bias']
    if taps is not None:
      taps.append(pre.detach())
    x = torch.relu(pre)
  L = mod[f'Dense_{cfg.mask_depth}']
  return torch.sigmoid(x @ L['kernel'] + L['bias'])


class _QuantSTE(torch.autograd.Function):
  """Round to bf16 (kept as float32 values) in the forward AND the backward pass: what a 16-bit operand copy of a tensor
  is to the product's bf16 mode (weights, activations and the gradients that flow back through them)."""

  @staticmethod
  def forward(ctx, x):
    return x.to(torch.bfloat16).to(x.dtype)

  @staticmethod
  def backward(ctx, g):
    return g.to(torch.bfloat16).to(g.dtype)


def _relu(pre, masks):
  """relu, or -- tests that replay the implementation under test's own ReLU decisions -- pre * mask with the next mask of
  the list (consumed in call order: trunk layers, then the view layer)."""
  if masks is None:
    return torch.relu(pre)
  m = masks.pop(0)
  return pre * m.reshape(pre.shape).to(pre.dtype)


def mlp_forward(cfg, mod, which, feats, viewdirs, glo_vec, taps=None, tra_vec=None, noise=None, relu_masks=None, quant=False):
  """models.py:406-550 (no transient branch). feats [...,S,504].  taps: optional list that receives the
  relu pre-activations (tests use it to find samples sitting on a ReLU kink).  relu_masks: optional list of 0/1 masks that
  replace the ReLU decisions (see _relu).  quant: emulate the product's bf16 mode -- GEMM operands (weights, layer inputs)
  rounded to bf16 with fp32 accumulation, gradients rounded on the way back."""
  depth = cfg.nerf_depth if which == 'nerf' else cfg.prop_depth
  q = _QuantSTE.apply if quant else (lambda t: t)
  feats = q(feats)
  masks = None if relu_masks is None else list(relu_masks)
  x, inputs = feats, feats
  for i in range(depth):
    L = mod[f'Dense_{i}']
    pre = x @ q(L['kernel']) + L['bias']
    if taps is not None:
      taps.append(pre.detach())
    x = q(_relu(pre, masks))
    if i % cfg.skip_layer == 0 and i > 0:
      x = torch.cat([x, inputs], -1)
  L = mod[f'Dense_{depth}']
  raw_density = (x @ L['kernel'] + L['bias'])[..., 0]
  if noise is not None and noise.get('density') is not None:      # models.py:458-460 (the draws are the caller's: the key chain
    raw_density = raw_density + noise['density']                  # of models.py:230,435 restated by the test with oracle/threefry_ref)
  density = torch.logaddexp(raw_density + cfg.density_bias, torch.zeros_like(raw_density))
  if which == 'prop' and cfg.prop_disable_rgb:
    return density, torch.zeros(feats.shape[:-1] + (3,), dtype=feats.dtype)
  if not getattr(cfg, 'use_viewdirs', True):
    # models.py:233 viewdirs=None -> :486-516: no bottleneck / view layer, the rgb layer reads the trunk output
    L = mod[f'Dense_{depth + 1}']
    rgb = torch.sigmoid(cfg.rgb_premultiplier * (x @ L['kernel'] + L['bias']) + cfg.rgb_bias)
    return density, rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
  L = mod[f'Dense_{depth + 1}']
  bott = q(x @ q(L['kernel']) + L['bias'])
  if noise is not None and noise.get('bottleneck') is not None:   # models.py:478-481
    bott = bott + noise['bottleneck']
  parts = [bott, pos_enc(viewdirs, 0, cfg.deg_view, True)[..., None, :].expand(bott.shape[:-1] + (-1,))]
  if glo_vec is not None:
    parts.append(glo_vec[..., None, :].expand(bott.shape[:-1] + (-1,)))
  x = torch.cat(parts, -1)
  L = mod[f'Dense_{depth + 2}']
  if quant:      # the product keeps the view-direction / GLO rows of this kernel in fp32 (a per-ray bias): only the bottleneck rows are 16-bit
    Bw = bott.shape[-1]
    pre = bott @ q(L['kernel'][:Bw]) + x[..., Bw:] @ L['kernel'][Bw:] + L['bias']
  else:
    pre = x @ L['kernel'] + L['bias']
  if taps is not None:
    taps.append(pre.detach())
  x = q(_relu(pre, masks))
  Dv = getattr(cfg, 'depth_viewdirs', 1)
  assert 1 <= Dv <= 4, 'models.py:511: the skip concat of the view MLP (i % skip_layer_dir == 0 and i > 0) starts at depth 5'
  for i in range(1, Dv):                                           # models.py:508-512, i >= 1
    L = mod[f'Dense_{depth + 2 + i}']
    pre = x @ q(L['kernel']) + L['bias']
    if taps is not None:
      taps.append(pre.detach())
    x = q(_relu(pre, masks))
  L = mod[f'Dense_{depth + 2 + Dv}']
  rgb = torch.sigmoid(cfg.rgb_premultiplier * (x @ L['kernel'] + L['bias']) + cfg.rgb_bias)      # models.py:514-516
  rgb = rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
  if tra_vec is None or which != 'nerf' or cfg.transient_type != 'nerfw':
    return density, rgb
  # models.py:521-539 (skip_layer_transient = 4 never triggers for depth 4)
  x = torch.cat([bott, tra_vec[..., None, :].expand(bott.shape[:-1] + (-1,))], -1)
  j = depth + 3 + Dv
  for i in range(cfg.transient_depth):
    pre = x @ mod[f'Dense_{j + i}']['kernel'] + mod[f'Dense_{j + i}']['bias']
    if taps is not None:
      taps.append(pre.detach())
    x = torch.relu(pre)
  j += cfg.transient_depth
  sp = lambda z: torch.logaddexp(z, torch.zeros_like(z))
  dens_t = sp((x @ mod[f'Dense_{j}']['kernel'] + mod[f'Dense_{j}']['bias'])[..., 0] + cfg.density_bias)
  rgb_t = torch.sigmoid(cfg.rgb_premultiplier * (x @ mod[f'Dense_{j + 1}']['kernel'] + mod[f'Dense_{j + 1}']['bias']) + cfg.rgb_bias)   # :534-536
  rgb_t = rgb_t * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
  unc = sp(x @ mod[f'Dense_{j + 2}']['kernel'] + mod[f'Dense_{j + 2}']['bias'])
  return density, rgb, dict(density_transient=dens_t, rgb_transient=rgb_t, uncertainty=unc)


def sample_u_base(num_samples, randomized):
  """stepfun.py:191-209 with deterministic_center=True: the fixed part of u (float64 host
  linspace rounded to float32) and the jitter scale."""
  eps = EPS
  if not randomized:
    pad = 1 / (2 * num_samples)
    return np.linspace(pad, 1. - pad - eps, num_samples).astype(np.float32), 0.0
  # `eps = jnp.finfo(jnp.float32).eps` is a binary32 scalar: u_max / max_jitter are binary32 arithmetic
  eps, one = np.float32(eps), np.float32(1)
  u_max = eps + (one - eps) / np.float32(num_samples)
  max_jitter = (one - u_max) / np.float32(num_samples - 1) - eps
  return np.linspace(0, float(1 - u_max), num_samples).astype(np.float32), float(max_jitter)


def model_forward(cfg, variables, rays, train_frac, u01, compute_extras, zero_glo=False, taps=None,
                  override_samples=None, override_feats=None, zero_tra=False, mask_taps=None, noise=None, bg_rgbs=None,
                  relu_masks=None, quant=False):
  """Model.__call__ (models.py:74-330).  rays: dict of [N,c] tensors.  u01: None
  (rng=None) or list[num_levels] of [N] float32 uniform draws (single_jitter)."""
  P = variables['params']
  dt = rays['origins'].dtype
  N = rays['origins'].shape[0]
  # models.py:393-396: jnp.array(generate_basis(...)).T is a float32 constant whatever the compute dtype
  basis = torch.tensor(generate_basis(cfg.basis_shape, cfg.basis_subdivisions).T.astype(np.float32)).to(dt)  # [3,nb]
  glo = None
  if cfg.num_glo_features > 0:
    glo = (torch.zeros(N, cfg.num_glo_features, dtype=dt) if zero_glo else
           P['GloEmbed_0']['embedding'][rays['embed_idx'][:, 0].long()])
  near, far = rays['near'], rays['far']
  # models.py:138-149: `near_anneal_rate` anneals the near bound in over the first part of training
  rate = getattr(cfg, 'near_anneal_rate', None)
  init_s_near = 0. if rate is None else float(np.clip(1 - train_frac / rate, 0, getattr(cfg, 'near_anneal_init', 0.95)))
  init_s_far = 1.
  sdist = torch.cat([torch.full_like(near, init_s_near), torch.full_like(far, init_s_far)], -1)
  weights = torch.ones_like(near)
  prod = 1
  renderings, history = [], []
  for lvl in range(cfg.num_levels):
    is_prop = lvl < cfg.num_levels - 1
    S = cfg.num_prop_samples if is_prop else cfg.num_nerf_samples
    dilation = cfg.dilation_bias + cfg.dilation_multiplier * (init_s_far - init_s_near) / prod      # models.py:161-162
    prod *= S
    anneal = (cfg.anneal_slope * train_frac) / ((cfg.anneal_slope - 1) * train_frac + 1) \
        if cfg.anneal_slope > 0 else 1.
    ub, mj = sample_u_base(S, u01 is not None)
    jit = None if u01 is None else (u01[lvl].detach().numpy().astype(np.float32) * np.float32(mj))
    sd, td, _ = cstepfun.level_sample(
        sdist.detach().numpy(), weights.detach().numpy(), lvl > 0, dilation, init_s_near, init_s_far, anneal,      # domain: models.py:176,203
        cfg.resample_padding, ub, jit, RAYDIST[cfg.raydist_fn],
        near.numpy(), far.numpy())
    sdist = torch.from_numpy(sd).to(dt)          # stop_gradient (models.py:208-209)
    tdist = torch.from_numpy(td).to(dt)
    if override_samples is not None:   # tests: feed the sampler output of the implementation under test
      sdist, tdist = override_samples[lvl][0].to(dt), override_samples[lvl][1].to(dt)
    means, covs = cast_rays(tdist, rays['origins'], rays['directions'], rays['radii'], cfg.ray_shape)
    if getattr(cfg, 'disable_integration', False):      # models.py:223-226
      covs = torch.zeros_like(covs)
    if cfg.warp:
      means, covs = contract_track_linearize(means, covs)
    lm, lv = lift_and_diagonalize(means, covs, basis)
    feats = integrated_pos_enc(lm, lv, getattr(cfg, 'min_deg_point', 0), cfg.max_deg_point)
    if override_feats is not None:     # tests: identical MLP inputs on both sides
      feats = override_feats[lvl].to(dt)
    which = 'prop' if is_prop else 'nerf'
    lvl_taps = None if taps is None else []
    tra = None
    if not is_prop and cfg.transient_type == 'nerfw':
      tra = (torch.zeros(N, cfg.num_transient_features, dtype=dt) if zero_tra else
             P['TransientEmbed_0']['embedding'][rays['embed_idx'][:, 0].long()])
    res = mlp_forward(cfg, P['PropMLP_0' if is_prop else 'NerfMLP_0'], which, feats,
                      rays['viewdirs'], None if is_prop else glo, lvl_taps, tra, None if noise is None else noise[lvl],
                      relu_masks=None if relu_masks is None else relu_masks[lvl], quant=quant)
    density, rgb = res[0], res[1]
    if taps is not None:
      taps.append(lvl_taps)
    weights = compute_alpha_weights(density, tdist, rays['directions'], cfg.opaque_background)[0]
    bg_l = cfg.bg_intensity if bg_rgbs is None else bg_rgbs[lvl]        # models.py:246-261 (a [N, 3] draw per level)
    rend = volumetric_rendering(rgb, weights, tdist, bg_l, far, compute_extras)
    hist = dict(density=density, rgb=rgb, sdist=sdist, tdist=tdist, weights=weights)
    if len(res) == 3:                      # models.py:285-307
      tr = res[2]
      w1, w2, wc = compute_dual_alpha_weights(density, tr['density_transient'], tdist, rays['directions'],
                                              cfg.opaque_background)
      bgw = _MaxZero.apply(1 - wc.sum(-1))[..., None]
      rend['rgb_static'] = (w1[..., None] * rgb).sum(-2)
      rend['rgb_transient'] = (w2[..., None] * tr['rgb_transient']).sum(-2)
      rend['rgb_combined'] = rend['rgb_static'] + rend['rgb_transient'] + bgw * cfg.bg_intensity
      wt = compute_alpha_weights(tr['density_transient'], tdist, rays['directions'], cfg.opaque_background)[0]
      rend['uncertainty'] = (wt[..., None] * tr['uncertainty']).sum(-2) + cfg.beta_min
      hist.update(tr)
    renderings.append(rend)
    history.append(hist)
  if cfg.transient_type == 'hanerf':     # models.py:120-129,327-328
    tra = (torch.zeros(N, cfg.num_transient_features, dtype=dt) if zero_tra else
           P['TransientEmbed_0']['embedding'][rays['embed_idx'][:, 0].long()])
    renderings[-1]['implicit_mask'] = implicit_mask_forward(cfg, P['ImplicitMask_0'], rays['pix_coords'].to(dt), tra,
                                                            mask_taps)
  return renderings, history


# ----------------------------------------------------------------------------
# train_utils.py -- losses, clip, Adam
# ----------------------------------------------------------------------------
def compute_data_loss(cfg, gt_rgb, rays, renderings, use_static_mask):
  """train_utils.py:72-111 incl. the static-mask normaliser quirk (denominator sums
  the [..,1] mask while the numerator sums 3 channels)."""
  losses, mses = [], []
  static_mask = (rays['static_mask'] >= 0.5).to(gt_rgb.dtype)
  for r in renderings:
    if use_static_mask:
      lossmult = static_mask + (1 - static_mask) * cfg.withmask_transient_weight  # [...,1]
    else:
      lossmult = rays['lossmult'].expand_as(gt_rgb)
      if cfg.disable_multiscale_loss:
        lossmult = torch.ones_like(lossmult)
    resid_sq = (r['rgb'] - gt_rgb)**2
    denom = torch.clamp(lossmult.sum(), min=EPS)
    mses.append((lossmult * resid_sq).sum() / denom)
    dl = resid_sq if cfg.data_loss_type == 'mse' else torch.sqrt(resid_sq + cfg.charb_padding**2)
    losses.append((lossmult * dl).sum() / denom)
  losses = torch.stack(losses)
  return cfg.data_coarse_loss_mult * losses[:-1].sum() + cfg.data_loss_mult * losses[-1], \
      {'mses': torch.stack(mses)}


def compute_dual_alpha_weights(density1, density2, tdist, dirs, opaque_background=False):
  """render.py:154-182."""
  delta = (tdist[..., 1:] - tdist[..., :-1]) * torch.linalg.norm(dirs[..., None, :], dim=-1)
  d1, d2, dd = density1 * delta, density2 * delta, (density1 + density2) * delta
  if opaque_background:
    inf = torch.full_like(dd[..., -1:], float('inf'))
    d1, d2, dd = [torch.cat([z[..., :-1], inf], -1) for z in (d1, d2, dd)]
  trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], -1)], -1))
  return (1 - torch.exp(-d1)) * trans, (1 - torch.exp(-d2)) * trans, (1 - torch.exp(-dd)) * trans


def compute_nerfw_loss(cfg, gt_rgb, renderings, history):
  """train_utils.py:150-183."""
  beta = renderings[-1]['uncertainty']
  losses, mses, out = [], [], {}
  for i, r in enumerate(renderings):
    resid_sq = (r['rgb_combined' if 'rgb_combined' in r else 'rgb'] - gt_rgb)**2
    dl = resid_sq if cfg.data_loss_type == 'mse' else torch.sqrt(resid_sq + cfg.charb_padding**2)
    if i == len(renderings) - 1:
      out['beta'] = cfg.nerfw_beta_loss_mult * torch.log(beta).mean() + cfg.nerfw_beta_loss_bias
      dl = dl / (2 * beta**2)
      out['density'] = cfg.nerfw_density_loss_mult * history[-1]['density_transient'].mean()
    losses.append(dl.mean())
    mses.append(resid_sq.mean())
  losses = torch.stack(losses)
  out['data'] = cfg.data_coarse_loss_mult * losses[:-1].sum() + cfg.data_loss_mult * losses[-1]
  return out, {'mses': torch.stack(mses)}


def compute_hanerf_loss(cfg, gt_rgb, renderings, train_frac):
  """train_utils.py:186-225."""
  mult = max(cfg.hanerf_mask_size_loss_mult_min, cfg.hanerf_mask_size_loss_mult_max *
             math.exp(-train_frac * cfg.max_steps * cfg.hanerf_mask_size_loss_mult_k))
  m = renderings[-1]['implicit_mask']
  losses, mses, out = [], [], {}
  for i, r in enumerate(renderings):
    resid_sq = (r['rgb'] - gt_rgb)**2
    dl = resid_sq if cfg.data_loss_type == 'mse' else torch.sqrt(resid_sq + cfg.charb_padding**2)
    if i == len(renderings) - 1:
      dl = (1. - m) * dl
      out['mask_size'] = mult * (m**2).mean()
    else:
      dl = (1. - m.detach()) * dl
    losses.append(dl.mean())
    mses.append(resid_sq.mean())
  losses = torch.stack(losses)
  out['data'] = cfg.data_coarse_loss_mult * losses[:-1].sum() + cfg.data_loss_mult * losses[-1]
  return out, {'mses': torch.stack(mses), 'implicit_mask': m.mean().detach()[None]}


def robustnerf_mask(cfg, errors, thr):
  """train_utils.py:251-319. errors [n,P,P,3]; thr scalar (current inlier threshold)."""
  err = errors.mean(-1, keepdim=True)
  stats = {'inlier_threshold': torch.quantile(err.flatten(), cfg.robustnerf_inlier_quantile)}
  assert cfg.robustnerf_inner_patch_size <= cfg.patch_size, \
      'patch_size must be larger than robustnerf_inner_patch_size.'
  inl = (err < thr).to(err.dtype)
  stats['is_inlier_loss'] = inl.mean()
  f = cfg.robustnerf_smoothed_filter_size
  win = torch.ones(1, 1, f, f, dtype=err.dtype) / (f * f)
  # lax.conv 'SAME': f-1 zero rows/cols in total, (f-1)//2 before and the rest after (asymmetric for even f)
  lo_, hi_ = (f - 1) // 2, f - 1 - (f - 1) // 2
  nb = torch.nn.functional.conv2d(torch.nn.functional.pad(inl.permute(0, 3, 1, 2), (lo_, hi_, lo_, hi_)),
                                  win).permute(0, 2, 3, 1)
  nb = (nb > 1 - cfg.robustnerf_smoothed_inlier_quantile).to(err.dtype)
  stats['has_inlier_neighbors'] = nb.mean()
  ip, op = cfg.robustnerf_inner_patch_size, cfg.patch_size
  lo = (op - ip) // 2
  inner = torch.zeros(1, op, op, 1, dtype=err.dtype)
  inner[:, lo:lo + ip, lo:lo + ip, :] = 1
  patch = (inl.mean(dim=(1, 2), keepdim=True) > 1 - cfg.robustnerf_inner_patch_inlier_quantile).to(err.dtype) * inner
  stats['is_inlier_patch'] = patch.mean()
  mask = ((patch + nb + inl) > 1e-3).to(err.dtype)
  stats['mask'] = mask.mean()
  return mask, stats


def compute_robustnerf_loss(cfg, gt_rgb, renderings, inlier_thresholds):
  """train_utils.py:114-147. gt_rgb [n,P,P,3]; renderings' rgb same shape."""
  losses, stats = [], {'mses': []}
  for i, r in enumerate(renderings):
    resid_sq = (r['rgb'] - gt_rgb)**2
    dl = resid_sq if cfg.data_loss_type == 'mse' else torch.sqrt(resid_sq + cfg.charb_padding**2)
    mask, rs = robustnerf_mask(cfg, torch.sqrt(resid_sq).detach(), inlier_thresholds[i][0])
    for k, v in rs.items():
      stats.setdefault('robust_' + k, []).append(v)
    lossmult = mask.detach().expand_as(dl)
    denom = torch.clamp(lossmult.sum(), min=EPS)
    stats['mses'].append((lossmult * resid_sq).sum() / denom)
    losses.append((lossmult * dl).sum() / denom)
  losses = torch.stack(losses)
  stats = {k: torch.stack(v) for k, v in stats.items()}
  return cfg.data_coarse_loss_mult * losses[:-1].sum() + cfg.data_loss_mult * losses[-1], stats


def interlevel_loss(cfg, history):
  """train_utils.py:228-239."""
  c, w = history[-1]['sdist'].detach(), history[-1]['weights'].detach()
  tot = 0.
  for h in history[:-1]:
    tot = tot + lossfun_outer(c, w, h['sdist'], h['weights']).mean()
  return cfg.interlevel_loss_mult * tot


def distortion_loss(cfg, history):
  """train_utils.py:242-248."""
  return cfg.distortion_loss_mult * lossfun_distortion(history[-1]['sdist'], history[-1]['weights']).mean()


def flat_leaves(tree, prefix=()):
  out = []
  for k in sorted(tree.keys(), key=_natkey):
    v = tree[k]
    if isinstance(v, dict):
      out += flat_leaves(v, prefix + (k,))
    else:
      out.append(('/'.join(prefix + (k,)), v))
  return out


def _natkey(s):
  import re
  return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s)]


def loss_and_grad(cfg, variables, rays, gt_rgb, train_frac, u01, inlier_thresholds=None, bg_rgbs=None, is_finetune=False,
                  **forward_kw):
  """The value_and_grad half of train_step (train_utils.py:404-455).  rays/gt flat [N,c];
  robustnerf reshapes to [n,P,P,c] patches.  is_finetune (create_train_step's third argument): the plain data loss whatever
  transient_type is (:422-423), no interlevel / distortion / weight-decay term (:435-447); the model itself is unchanged."""
  leaves = flat_leaves(variables['params'])
  req = [v.clone().requires_grad_(True) for _, v in leaves]
  P = {}
  for (name, _), v in zip(leaves, req):
    d = P
    ks = name.split('/')
    for k in ks[:-1]:
      d = d.setdefault(k, {})
    d[ks[-1]] = v
  renderings, history = model_forward(cfg, {'params': P}, rays, train_frac, u01, False, bg_rgbs=bg_rgbs, **forward_kw)
  losses, stats = {}, {}
  if is_finetune or cfg.transient_type is None:
    losses['data'], st = compute_data_loss(cfg, gt_rgb, rays, renderings, False)
  elif cfg.transient_type == 'withmask':
    losses['data'], st = compute_data_loss(cfg, gt_rgb, rays, renderings, True)
  elif cfg.transient_type == 'robustnerf':
    ps = cfg.patch_size
    rs = [{'rgb': r['rgb'].reshape(-1, ps, ps, 3)} for r in renderings]
    losses['data'], st = compute_robustnerf_loss(cfg, gt_rgb.reshape(-1, ps, ps, 3), rs, inlier_thresholds)
  elif cfg.transient_type == 'hanerf':
    ls, st = compute_hanerf_loss(cfg, gt_rgb, renderings, train_frac)
    losses.update(ls)
  elif cfg.transient_type == 'nerfw':
    ls, st = compute_nerfw_loss(cfg, gt_rgb, renderings, history)
    losses.update(ls)
  else:
    raise ValueError()
  stats.update(st)
  if not is_finetune and cfg.interlevel_loss_mult > 0:
    losses['interlevel'] = interlevel_loss(cfg, history)
  if not is_finetune and cfg.distortion_loss_mult > 0:
    losses['distortion'] = distortion_loss(cfg, history)
  if not is_finetune and cfg.weight_decay_mults:        # train_utils.py:442-447: m * ||theta_group||^2 over summarize_tree groups
    losses['weight'] = sum(m * sum((v**2).sum() for (name, _), v in zip(leaves, req)
                                   if name == k or name.startswith(k + '/'))
                           for k, m in cfg.weight_decay_mults.items())
  loss = sum(losses.values())
  grads = torch.autograd.grad(loss, req, allow_unused=True)
  grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, req)]
  stats['loss'] = loss.detach()
  stats['losses'] = {k: v.detach() for k, v in losses.items()}
  return stats, {n: g for (n, _), g in zip(leaves, grads)}, renderings, history


def clip_gradients(cfg, grads):
  """train_utils.py:351-369: per top-level module value clip, then norm clip. grads: {leafname: g}."""
  out = {}
  mods = sorted({n.split('/')[0] for n in grads})
  for m in mods:
    names = [n for n in grads if n.split('/')[0] == m]
    gs = {n: grads[n] for n in names}
    if cfg.grad_max_val > 0:
      gs = {n: g.clamp(-cfg.grad_max_val, cfg.grad_max_val) for n, g in gs.items()}
    if cfg.grad_max_norm > 0:
      nrm = torch.sqrt(sum((g.double()**2).sum() for g in gs.values())).to(next(iter(gs.values())).dtype)
      mult = torch.clamp(cfg.grad_max_norm / (EPS + nrm), max=1.0)
      gs = {n: mult * g for n, g in gs.items()}
    out.update(gs)
  return out


def finetune_trainable(name):
  """create_finetune_optimizer's partition (train_utils.py:539-543): Adam on the leaves that have 'embedding' as one of their path
  keys, a zero update on every other leaf."""
  return 'embedding' in name.split('/')


def adam_update(cfg, params, grads, m, v, count):
  """optax.adam (scale_by_adam + scale_by_schedule(-lr)); count = 0-based update index.
  Documented formula: mhat = m/(1-b1^t), vhat = v/(1-b2^t), t=count+1;
  theta -= lr(count) * mhat / (sqrt(vhat) + eps)."""
  b1, b2, eps = cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps
  lr = learning_rate_decay(count, cfg.lr_init, cfg.lr_final, cfg.max_steps, cfg.lr_delay_steps,
                           cfg.lr_delay_mult)
  t = count + 1
  newp, newm, newv = {}, {}, {}
  for n in params:
    g = torch.nan_to_num(grads[n])       # train_utils.py:466
    newm[n] = b1 * m[n] + (1 - b1) * g
    newv[n] = b2 * v[n] + (1 - b2) * g * g
    mhat = newm[n] / (1 - b1**t)
    vhat = newv[n] / (1 - b2**t)
    newp[n] = params[n] - lr * mhat / (torch.sqrt(vhat) + eps)
  return newp, newm, newv

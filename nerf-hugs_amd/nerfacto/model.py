"""The reference's NERFACTO training path (nerfacto/models/nerfacto.py `Model` + `Loss`, nerfacto/train.py:183-215) on HIP
kernels: proposal sampling (csrc/hugs_nerfacto.hip), multiresolution hash
grids + SH-4 (csrc/hugs_hashgrid.hip), the fields' Linear layers on the shared MFMA GEMMs (csrc/hugs_gemm.hip, every
width zero-padded to the 128-column tile), density -> weights -> colour, interlevel / distortion / rgb losses, a
hand-scheduled backward pass and Adam.  SURVEY 8f row 3, BASELINE config 5.

Field form: the `enable_tcnn_mlp: False` one (torch Linear layers, what configs/phototourism_nerfacto_base.yml
selects); the reference runs them under fp16 autocast + GradScaler (`enable_amp: True`) = `compute_dtype='fp16'` here
(half MFMA operands / activations / table copies, fp32 accumulation and master parameters, the dynamic loss scale on the
device); `'bf16'` is the same structure with bf16 operands and no loss scale, `'fp32'` the parity mode.  The colour MLP's
last layer (hidden -> 3) + sigmoid is one pass over the hidden activation (the rgb head kernels of csrc/hugs_heads.hip)
rather than a GEMM padded to 128 columns.  PARITY UNPINNED for the encodings themselves (tiny-cuda-nn);
the sampler / weights / losses are pinned by vectors from the reference's pure-torch utils, and the field / model / loss
WIRING by vectors recorded from executing the reference's own `Model` / `Loss` classes over a stand-in `tinycudann`
(tests/golden/gen_nerfacto_model_fixtures.py -> tests/test_gpu_nerfacto_reference.py).
Built: transient_type None / 'withmask' / 'robustnerf' / 'hanerf' (ImplicitMask: 2-D hash grid of the ray's image
coordinate | transient embedding -> Linear+relu x 2 -> sigmoid, compute_hanerf_loss), eval-mode rendering with
eval_embedding average / zero / original, the finetune stage (plain data loss, listed parameter groups only), the
reference's proposal-update gating INCLUDING what it does to the optimizer (parameters whose .grad stays None are
skipped by torch.optim.Adam: no moment decay, no step count).
'nerfw' raises: the reference's own nerfacto NeRF-W branch cannot execute (nerfacto.py:394-401 format an undefined name
`output_type`: NameError on the first forward; recorded in ref_nerfacto_model.npz).
Multi-GPU: one process per GPU with an all-reduce of the flat gradient (the reference wraps the model in
nn.DataParallel: ONE loss over the gathered batch).  Per-rank normalisers therefore differ from the reference's for
'withmask' (mask sum) and 'robustnerf' (the inlier quantile is taken per rank, and each rank feeds back its own)."""
import math
import os

import numpy as np
import torch

from .. import _lib as L
from ..internal.engine import Workspace
from .encodings import HashGrid

PAD = 128


def _rup(x, m=PAD):
  return (x + m - 1) // m * m


# Round 5 (ADVICE r3 / VERDICT r4 item 7): the gradient that enters the hash grids -- the fused kernels' last product -- leaves in
# fp32.  As a 16-bit store the half mode flushed every scaled gradient below 6e-8 to zero there (71 / 79 / 33 % of the prop0 /
# prop1 / field grid-input gradients at initialisation, profiles/r03_cfg5_fp16_gradient_underflow.txt), and all-zero runs issue no
# table atomics: part of the fp16 mode's speed was gradient information it dropped.  HUGS_NF_GRID_GRAD_F32=0 restores the 16-bit
# store (what tiny-cuda-nn's half pipeline does; bench.py --config cfg5 reports both).
_GRID_GRAD_F32 = os.environ.get('HUGS_NF_GRID_GRAD_F32', '1') != '0'

class NerfactoConfig:
  """models/nerfacto.py:18-114 ModelConfig + Model.__init__'s bound / enable_scene_contraction (dataclass defaults)."""

  def __init__(self, **kw):
    self.bound, self.enable_scene_contraction = 2.0, False
    self.num_levels, self.base_res, self.max_res, self.log2_hashmap_size, self.features_per_level = 16, 16, 2048, 19, 2
    self.hidden_dim, self.geo_feat_dim, self.hidden_dim_color = 64, 15, 64
    self.use_appearance_embedding, self.appearance_embedding_dim, self.num_embedding = False, 32, 3500
    self.eval_embedding = 'average'          # 'average' | 'zero' | 'original' (Model.get_embedding, nerfacto.py:266-284)
    self.num_proposal_samples_per_ray, self.num_nerf_samples_per_ray, self.num_proposal_iterations = (256, 96), 48, 2
    self.proposal_net_args_list = [dict(hidden_dim=16, log2_hashmap_size=17, num_levels=5, max_res=128),
                                   dict(hidden_dim=16, log2_hashmap_size=17, num_levels=5, max_res=256)]
    self.proposal_update_every, self.proposal_warmup = 5, 5000
    self.proposal_initial_sampler, self.proposal_histogram_padding = 'uniform', 0.01
    self.use_proposal_weight_anneal, self.proposal_weights_anneal_slope = True, 10.0
    self.proposal_weights_anneal_max_num_iters = 1000
    self.use_single_jitter, self.opaque_background = True, False
    self.rgb_loss_type, self.rgb_charb_loss_padding, self.rgb_loss_mult = 'mse', 0.001, 1.0
    self.interlevel_loss_mult, self.distortion_loss_mult = 1.0, 0.002
    self.transient_type, self.withmask_transient_weight = None, 0.
    self.robustnerf_inlier_quantile, self.robustnerf_smoothed_filter_size = 0.8, 3
    self.robustnerf_smoothed_inlier_quantile, self.robustnerf_inner_patch_size = 0.5, 8
    self.robustnerf_inner_patch_inlier_quantile, self.patch_size = 0.4, 16      # patch_size: train.py's config.patch_size
    self.rgb_bias = 0.
    # round 5: nerfacto.py:36 density_activation ('trunc_exp' | 'softplus': F.softplus(raw + density_bias), density_bias = -1 is the
    # fields' constructor default, nerfacto.py:660,890) and :66 use_same_proposal_network (ONE proposal network for every level)
    self.density_activation, self.density_bias, self.use_same_proposal_network = 'trunc_exp', -1., False
    # HA-NeRF (nerfacto.py:42-53, 94-102): transient embedding + ImplicitMask
    self.use_transient_embedding, self.transient_embedding_dim = False, 16
    self.num_levels_implicit, self.base_res_implicit, self.max_res_implicit = 8, 16, 1024
    self.log2_hashmap_size_implicit, self.features_per_level_implicit, self.hidden_dim_implicit = 17, 2, 128
    self.hanerf_mask_size_loss_mult_min, self.hanerf_mask_size_loss_mult_max, self.hanerf_mask_size_loss_mult_k = 6e-3, 5e-2, 1e-3
    # train.py / yml optimiser settings
    self.lr_init, self.lr_final, self.lr_decay_mult, self.warmup_steps, self.num_steps = 1e-2, 1e-3, 1e-8, 500, 25000
    self.opt_betas, self.opt_eps = (0.9, 0.999), 1e-15
    for k, v in kw.items():
      if not hasattr(self, k):
        raise ValueError(f'ModelConfig has no field {k!r}')
      setattr(self, k, v)
    if self.transient_type == 'nerfw':
      # the reference constructs this model and dies on its first forward: nerfacto.py:394-401 format `output_type`,
      # a name that is never bound anywhere in the file
      raise NameError("name 'output_type' is not defined (the reference's nerfacto NeRF-W branch, models/nerfacto.py:394, "
                      "cannot execute; nothing to reproduce)")
    if self.density_activation not in ('trunc_exp', 'softplus'):
      raise NotImplementedError()                                                             # nerfacto.py:709-710
    if self.use_same_proposal_network:
      assert len(self.proposal_net_args_list) == 1, 'Only one proposal network is allowed.'     # nerfacto.py:192
    if self.transient_type not in (None, 'withmask', 'robustnerf', 'hanerf'):
      raise NotImplementedError(f"nerfacto transient_type {self.transient_type!r}")
    if self.transient_type == 'hanerf':       # nerfacto.py:139-143
      assert self.transient_embedding_dim > 0 and self.use_transient_embedding
    else:
      assert not self.use_transient_embedding
    if self.transient_type == 'robustnerf':
      assert self.robustnerf_inner_patch_size <= self.patch_size, 'patch_size must be larger than robustnerf_inner_patch_size.'
    if self.eval_embedding not in ('average', 'zero', 'original'):
      raise NotImplementedError(f'{self.eval_embedding} is not supported.')                   # nerfacto.py:283
    if self.proposal_initial_sampler not in ('uniform', 'piecewise', 'reciprocal'):
      raise ValueError(f'Sampler does not support {self.proposal_initial_sampler}. ')       # nerfacto.py:241
    if self.enable_scene_contraction and self.bound != 2.0:
      raise AssertionError(f'When using scene contraction, bound should be set to 2, but got {self.bound}')

  def prop_args(self, i):
    a = dict(num_levels=8, base_res=16, max_res=1024, log2_hashmap_size=18, features_per_level=2, hidden_dim=64)
    a.update(self.proposal_net_args_list[min(i, len(self.proposal_net_args_list) - 1)])
    return a


class _Layout:
  """Flat fp32 parameter buffer: name -> (offset, stored shape, logical shape)."""

  def __init__(self):
    self.items, self.size = {}, 0

  def add(self, name, pshape, shape=None):
    n = int(np.prod(pshape))
    self.items[name] = (self.size, tuple(pshape), tuple(shape or pshape))
    self.size += _rup(n, 4)

  def view(self, flat, name, padded=True):
    off, pshape, shape = self.items[name]
    v = flat[off:off + int(np.prod(pshape))].view(*pshape)
    return v if padded else v[tuple(slice(0, s) for s in shape)]


class NerfactoModel:

  def __init__(self, cfg, device='cuda', compute_dtype='bf16', seed=0, grid_grad_f32=None):
    if not torch.cuda.is_available():
      raise L.HugsError('no GPU visible: the hugs path has no CPU fallback')
    L.lib()
    self.cfg, self.device = cfg, torch.device(device)
    if compute_dtype not in ('fp32', 'bf16', 'fp16'):
      raise ValueError(f"compute_dtype {compute_dtype!r}: 'fp32' (parity), 'bf16' or 'fp16' (the reference's enable_amp)")
    self.dt = {'fp32': 0, 'bf16': 1, 'fp16': 2}[compute_dtype]
    self.tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[self.dt]
    # fp16 mode = the reference's `enable_amp: True` (phototourism_nerfacto_base.yml:3): half MFMA operands and activations,
    # half copies of the hash tables for the forward gathers (tiny-cuda-nn's parameter precision), fp32 master parameters and
    # accumulation, and torch.cuda.amp.GradScaler's dynamic loss scale (train.py:168,210-213) kept on the device
    self.amp = self.dt == 2
    self.grid_grad_f32 = _GRID_GRAD_F32 if grid_grad_f32 is None else bool(grid_grad_f32)      # (see _GRID_GRAD_F32 above)
    self.ws = Workspace(self.device)
    self.L = cfg.num_proposal_iterations
    self.lay = _Layout()
    self.grids, self.nets, self.fused = {}, {}, {}
    self.dact, self.dbias = int(cfg.density_activation == 'softplus'), float(cfg.density_bias)
    # level -> the proposal network it evaluates (nerfacto.py:334: network 0 for every level with use_same_proposal_network)
    self.net_of = (lambda lvl: 'prop0') if cfg.use_same_proposal_network else (lambda lvl: f'prop{lvl}')
    for i in range(1 if cfg.use_same_proposal_network else self.L):
      a = cfg.prop_args(i)
      g = HashGrid(a['num_levels'], a['features_per_level'], a['log2_hashmap_size'], a['base_res'], None, a['max_res'], device='cpu')
      self.grids[f'prop{i}'] = g
      self._add_net(f'prop{i}', g, [(g.n_output_dims, a['hidden_dim']), (a['hidden_dim'], 1)])
      # proposal nets up to 32 -> 64 -> 1 run as ONE fused kernel per direction (csrc/hugs_nerfacto.hip k_nf_prop_*: the
      # reference's tcnn fully fused MLP); wider ones fall back to the padded GEMMs
      self.fused[f'prop{i}'] = (g.n_output_dims <= 32 and a['hidden_dim'] <= 64 and os.environ.get('HUGS_NF_FUSED_PROP', '1') != '0')
    g = HashGrid(cfg.num_levels, cfg.features_per_level, cfg.log2_hashmap_size, cfg.base_res, None, cfg.max_res, device='cpu')
    self.grids['field'] = g
    self.napp = cfg.appearance_embedding_dim if cfg.use_appearance_embedding else 0
    self._add_net('field', g, [(g.n_output_dims, cfg.hidden_dim), (cfg.hidden_dim, 1 + cfg.geo_feat_dim)])
    hin = 16 + cfg.geo_feat_dim + self.napp
    # the colour MLP's last layer (hidden -> 3) + sigmoid is ONE pass over the hidden activation (csrc/hugs_heads.hip k_rgb_fwd /
    # k_rgb_bwd, the Mip-NeRF 360 rgb head's kernels) instead of a GEMM padded from 3 to 128 output columns: its weight is
    # stored [hidden_padded, 3]
    self.rgb_head = _rup(cfg.hidden_dim_color) in (128, 256) and os.environ.get('HUGS_NF_RGB_HEAD', '1') != '0'
    for j, (fi, fo) in enumerate([(hin, cfg.hidden_dim_color), (cfg.hidden_dim_color, cfg.hidden_dim_color), (cfg.hidden_dim_color, 3)]):
      if j == 2 and self.rgb_head:
        self.lay.add('field/c2', (_rup(fi), 3), (fi, 3))
        self.lay.add('field/cb2', (3,), (3,))
        continue
      self.lay.add(f'field/c{j}', (_rup(fi), _rup(fo)), (fi, fo))
      self.lay.add(f'field/cb{j}', (_rup(fo),), (fo,))
    if self.napp:
      self.lay.add('appearance', (cfg.num_embedding, self.napp))
    self.ntra, self.mask_grid = 0, None
    if cfg.transient_type == 'hanerf':
      # TransientEmbed + ImplicitMask (nerfacto.py:159-166, 218-231, 1010-1091): 2-D grid | embedding -> (Linear+relu) x 2 ->
      # Linear(1) -> sigmoid; the head is a vector (dot product per ray)
      self.ntra = cfg.transient_embedding_dim
      self.lay.add('transient', (cfg.num_embedding, self.ntra))
      g2 = HashGrid(cfg.num_levels_implicit, cfg.features_per_level_implicit, cfg.log2_hashmap_size_implicit, cfg.base_res_implicit,
                    None, cfg.max_res_implicit, device='cpu', dims=2)
      self.mask_grid = g2
      self.lay.add('mask/table', (g2.n_entries, g2.features))
      Hm, kin = cfg.hidden_dim_implicit, g2.n_output_dims + self.ntra
      self.lay.add('mask/m0', (_rup(kin), _rup(Hm)), (kin, Hm)); self.lay.add('mask/mb0', (_rup(Hm),), (Hm,))
      self.lay.add('mask/m1', (_rup(Hm), _rup(Hm)), (Hm, Hm)); self.lay.add('mask/mb1', (_rup(Hm),), (Hm,))
      self.lay.add('mask/m2', (_rup(Hm), 1), (Hm, 1)); self.lay.add('mask/mb2', (1,), (1,))
    # optimizer groups (Model.get_params_dict nerfacto.py:250-264): name -> [lo, hi) of the flat buffer; the proposal
    # networks come first in the layout, so 'proposal' is one contiguous range
    self.groups = {'proposal': (0, self.lay.items['field/table'][0])}
    ends = {'field': 'appearance' if self.napp else ('transient' if self.ntra else None)}
    self.groups['field'] = (self.lay.items['field/table'][0], self.lay.items[ends['field']][0] if ends['field'] else self.lay.size)
    if self.napp:
      self.groups['appearance_embedding'] = (self.lay.items['appearance'][0], self.lay.items['transient'][0] if self.ntra else self.lay.size)
    if self.ntra:
      self.groups['transient_embedding'] = (self.lay.items['transient'][0], self.lay.items['mask/table'][0])
      self.groups['implicit_mask'] = (self.lay.items['mask/table'][0], self.lay.size)
    self.trainable = None         # None: every group (train stage); else the finetune stage's list (train.py:136)
    self.counts = {}              # per-group number of Adam updates (torch keeps `step` per parameter)
    self.flat = torch.zeros(self.lay.size, dtype=torch.float32, device=self.device)
    self.m = torch.zeros_like(self.flat)
    self.v = torch.zeros_like(self.flat)
    self.grad = torch.zeros_like(self.flat)
    self.step = 0
    self.wt, self.wn = {}, {}
    if self.amp:
      # GradScaler defaults (torch/amp/grad_scaler.py): init_scale 2^16, growth 2, backoff 0.5, growth_interval 2000
      self.amp_state = torch.tensor([65536.0, 0.0, 0.0], device=self.device)       # scale, growth tracker, found_inf
      self.amp_opts = dict(growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)
      self.amp_counts = torch.zeros(32, device=self.device)                         # Adam updates taken, per group
      self.amp_bc = torch.zeros(64, device=self.device)
      self.group_order = [g for g, _ in sorted(self.groups.items(), key=lambda kv: kv[1][0])]
      self.flat_h = torch.zeros(self.lay.size, dtype=torch.float16, device=self.device)     # half copy (tables are read from it)
    self._init(seed)
    self.refresh_weights()

  def _add_net(self, name, grid, dims):
    self.lay.add(f'{name}/table', (grid.n_entries, grid.features))
    for j, (fi, fo) in enumerate(dims):
      self.lay.add(f'{name}/w{j}', (_rup(fi), _rup(fo)), (fi, fo))
      self.lay.add(f'{name}/b{j}', (_rup(fo),), (fo,))

  # ---- parameters ---------------------------------------------------------------------------------------------------
  def _init(self, seed):
    """kaiming_uniform_ weights (nerfacto.py:789-791), nn.Linear default biases, U(+-1e-4) tables (tiny-cuda-nn), N(0,1)
    embedding rows (nn.Embedding)."""
    g = torch.Generator().manual_seed(int(seed))
    for name, (off, pshape, shape) in self.lay.items.items():
      leaf = name.split('/')[-1]
      v = self.lay.view(self.flat, name, padded=False)
      if leaf == 'table':
        v.copy_(((torch.rand(shape, generator=g) * 2 - 1) * 1e-4).to(self.device))
      elif name in ('appearance', 'transient'):
        v.copy_(torch.randn(shape, generator=g).to(self.device))
      elif (leaf[0] in 'wc' and not leaf.startswith('cb')) or (leaf[0] == 'm' and not leaf.startswith('mb')):
        v.copy_(((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / shape[0])).float().to(self.device))
      elif leaf.startswith('b') or leaf.startswith('cb') or leaf.startswith('mb'):
        fan_in = self.lay.items[name.replace('/cb', '/c').replace('/mb', '/m').replace('/b', '/w')][2][0]
        v.copy_(((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) / math.sqrt(fan_in)).float().to(self.device))

  def params(self):
    """{'prop0': {'table','w0','b0','w1','b1'}, ..., 'field': {...,'c0','cb0',...}, 'appearance': ...}: logical views."""
    out = {}
    for name in self.lay.items:
      parts = name.split('/')
      if len(parts) == 1:
        out[name] = self.lay.view(self.flat, name, False)
      else:
        out.setdefault(parts[0], {})[parts[1]] = self.lay.view(self.flat, name, False)
    return out

  def load_params(self, P):
    for name in self.lay.items:
      parts = name.split('/')
      src = P[name] if len(parts) == 1 else P[parts[0]][parts[1]]
      self.lay.view(self.flat, name, False).copy_(torch.as_tensor(src).detach().to(self.device, torch.float32))
    self.refresh_weights()

  def grads(self):
    out = {}
    for name in self.lay.items:
      parts = name.split('/')
      v = self.lay.view(self.grad, name, False)
      if len(parts) == 1:
        out[name] = v
      else:
        out.setdefault(parts[0], {})[parts[1]] = v
    return out

  def refresh_weights(self):
    if self.amp:
      self.flat_h.copy_(self.flat)             # one cast of the whole buffer; only the table ranges are read from it
    for name, (off, pshape, shape) in self.lay.items.items():
      leaf = name.split('/')[-1]
      if len(pshape) == 2 and leaf != 'table' and name not in ('appearance', 'transient', 'mask/m2') and not (name == 'field/c2' and self.rgb_head):
        K, N = pshape
        if name not in self.wt:
          self.wt[name] = torch.empty(N, K, dtype=self.tdt, device=self.device)
          self.wn[name] = torch.empty(K, N, dtype=self.tdt, device=self.device) if self.dt else None
        W = self.lay.view(self.flat, name)
        L.call('hugs_cast_weights', self.dt, K, N, W, self.wn[name], self.wt[name])
        if not self.dt:
          self.wn[name] = W
    self._w1x_stale = True

  def _refresh_w1x(self):
    """The base network's second layer in head-input row order (csrc/hugs_fieldfuse.hip: row 0 = raw density, rows 16 .. 16 + ngeo
    = the geo features): its outputs land in the colour network's input tile without a column shift.  Rebuilt on first use after
    a weight refresh."""
    g = self.cfg.geo_feat_dim
    if 'field/w1x' not in self.wt:
      self.wt['field/w1x'] = torch.zeros(128, self.wt['field/w1'].shape[1], dtype=self.tdt, device=self.device)
      self.b1x = torch.zeros(128, device=self.device)
    w1, b1 = self.wt['field/w1'], self.lay.view(self.flat, 'field/b1')
    self.wt['field/w1x'][0].copy_(w1[0]); self.wt['field/w1x'][16:16 + g].copy_(w1[1:1 + g])
    self.b1x[0:1].copy_(b1[0:1]); self.b1x[16:16 + g].copy_(b1[1:1 + g])
    if not hasattr(self, 'w1xn'):
      self.w1xn = torch.empty(self.wt['field/w1x'].shape[1], 128, dtype=self.tdt, device=self.device)
    self.w1xn.copy_(self.wt['field/w1x'].t())      # [k_out][n]: the backward's form (csrc/hugs_fieldfuse.hip k_field_bwd)
    self._w1x_stale = False

  def _field_fuse_ok(self):
    """The fused field kernels (csrc/hugs_fieldfuse.hip) take the phototourism yml's shape class: 16-bit operands, <= 32 hash
    features -> 256 -> 1 + geo, [SH16 | geo | appearance] = at most 128 columns -> 256 -> 256 -> 3 through the rgb head."""
    c = self.cfg
    if not self.dt or not self.rgb_head or os.environ.get('HUGS_NF_FIELD_FUSE', '1') == '0' or 'field/c0' not in self.lay.items:
      return False
    (K0, N0), (_, N1), (Kh, H) = (self.lay.items[k][1] for k in ('field/w0', 'field/w1', 'field/c0'))
    return (self.lay.items['field/w0'][2][0] <= 32 and N0 == 256 and N1 == 128 and Kh == 128 and H == 256 and
            self.lay.items['field/c1'][1] == (256, 256) and 16 + c.geo_feat_dim + self.napp <= 128 and
            # (the backward kernel's shape class, hugs_nf_field_bwd: a forward it cannot differentiate must not be taken)
            c.geo_feat_dim % 8 == 0 and self.napp % 4 == 0 and self.napp <= 64)

  # ---- GEMM helpers ---------------------------------------------------------------------------------------------------
  def _bits_ok(self, M, width, k):
    """1-bit relu masks (hugs_gemm_nt_bits: written by the forward layer in the dX kernel's own lane layout, M*width/8 bytes
    instead of a re-read of the 16-bit activation): 16-bit modes, whole 256 x 256 tiles, K >= 128."""
    return bool(self.dt and M % 256 == 0 and width % 256 == 0 and k % 64 == 0 and k >= 128 and os.environ.get('HUGS_NF_RELU_BITS', '1') != '0')

  def _nt(self, M, name, X, bias, relu, out, mask=None, transpose=False, bits=None):
    """out[M,N] = act(X W + b) (transpose=False; bits: relu mask bits written) or out[M,K] = (X W^T) (* mask > 0)
    (transpose=True, the dX form; bits: the mask as bits instead of the activation `mask`)."""
    K, N = self.lay.items[name][1]
    if not transpose:
      if bits is not None:
        L.call('hugs_gemm_nt_bits', self.dt, M, N, K, 0, X, K, None, 0, self.wt[name], K, bias, int(relu), None, None, out, N, bits, None)
      else:
        L.call('hugs_gemm_nt', self.dt, M, N, K, 0, X, K, None, 0, self.wt[name], K, bias, None, 1, 0, int(relu), None, 0, None, None, out, N)
    elif bits is not None:
      L.call('hugs_gemm_nt_bits', self.dt, M, K, N, 0, X, N, None, 0, self.wn[name], N, None, 0, None, None, out, K, None, bits)
    else:
      L.call('hugs_gemm_nt', self.dt, M, K, N, 0, X, N, None, 0, self.wn[name], N, None, None, 1, 0, 0, mask, K if mask is not None else 0,
             None, None, out, K)

  def _tn(self, M, name, X, G, bias_name, out=None, out_bias=None):
    """grad[name] = X^T G, grad[bias] = colsum(G) (out / out_bias: other destinations of the same shapes)."""
    K, N = self.lay.items[name][1]
    tiles, step, target = (K // 128) * (N // 128), (64 if self.dt else 16), 768
    if self.dt and K % 256 == 0 and N % 256 == 0 and M // (256 // max(1, (K // 256) * (N // 256))) >= 2048:
      tiles, target = (K // 256) * (N // 256), 256            # the 256x256 ring kernel, one workgroup per CU
    units = M // step
    ns = max(1, min(units, (target + tiles - 1) // tiles))
    while units % ns:
      ns -= 1
    nbytes = L.lib().cdll.hugs_gemm_tn_ws_bytes(K, N, ns)
    slab = self.ws.get(f'tn_slab_{torch.cuda.current_stream().cuda_stream}', (max(nbytes // 4, 1),))      # (one per stream: the levels' chains run concurrently)
    L.call('hugs_gemm_tn', self.dt, M, K, N, ns, X, K, G, N, self.lay.view(self.grad, name) if out is None else out,
           self.lay.view(self.grad, bias_name) if out_bias is None else out_bias, slab)

  # ---- forward ----------------------------------------------------------------------------------------------------------
  def _u_base(self, ns, randomized):
    from ..internal import stepfun
    u, mj = stepfun.sample_u(ns, randomized)
    key = ('ub', ns, randomized)
    if key not in self.ws.bufs:
      self.ws.bufs[key] = torch.from_numpy(u).to(self.device)
    return self.ws.bufs[key], mj

  def anneal(self, curr_step):
    c = self.cfg
    if not c.use_proposal_weight_anneal:
      return 1.0
    f = float(np.clip(curr_step / c.proposal_weights_anneal_max_num_iters, 0, 1))
    s = c.proposal_weights_anneal_slope
    return (s * f) / ((s - 1) * f + 1)

  def _grid_fwd(self, name, x01, X0):
    g = self.grids[name]
    o, r, s = g._tables()
    if self.amp:
      L.call('hugs_hashgrid_fwd_t', x01.shape[0], g.n_levels, g.features, o, r, s, x01, self.lay.view(self.flat_h, f'{name}/table'), 2,
             self.dt, X0.stride(0), X0)
    else:
      L.call('hugs_hashgrid_fwd', x01.shape[0], g.n_levels, g.features, o, r, s, x01, self.lay.view(self.flat, f'{name}/table'),
             self.dt, X0.stride(0), X0)

  def _grid_bwd(self, name, x01, dX0):
    g = self.grids[name]
    o, r, s = g._tables()
    # round 6: the table gradient as a segmented reduction by table slot (csrc/hugs_hashgrid_binned.inc; HUGS_HG_BINNED=1).  Built for
    # VERDICT r5 item 5 and measured: same gradient (1e-6), NOT faster -- field grid 1860 vs 1863 us, the proposal grids (few, small
    # levels = a handful of bins) 19-22 ms vs 0.6-0.8 (profiles/r06_cfg5_hashgrid_levels.txt) -- so the L2 atomic scatter stays the default.
    if os.environ.get('HUGS_HG_BINNED', '0') != '1':
      L.call('hugs_hashgrid_bwd', x01.shape[0], g.n_levels, g.features, o, r, s, x01, dX0, 0 if dX0.dtype == torch.float32 else self.dt,
             dX0.stride(0), self.lay.view(self.grad, f'{name}/table'))
      return
    # a workspace per grid (the three grids' backward kernels run on three streams), sized for the worst case
    n = x01.shape[0]
    ws = self._hg_ws.get((name, n)) if hasattr(self, '_hg_ws') else None
    if ws is None:
      if not hasattr(self, '_hg_ws'):
        self._hg_ws = {}
      nbytes = int(L.lib().cdll.hugs_hashgrid_bwd_ws_bytes(n, g.n_levels, g.features))
      ws = self._hg_ws[(name, n)] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x01.device)
    L.call('hugs_hashgrid_bwd_ws', n, g.n_levels, g.features, o, r, s, x01, dX0, 0 if dX0.dtype == torch.float32 else self.dt,
           dX0.stride(0), self.lay.view(self.grad, f'{name}/table'), ws, ws.numel())

  def forward(self, rays, curr_step, u01=None, training=True):
    """Model.forward_rays (nerfacto.py:286-414), training mode.  rays: dict of device tensors origin / direction / viewdir
    [N,3], near / far [N], embed_idx [N] int32, bg_rgb [N,3] (or None).  u01: None (perturb=False) or one [N] tensor of
    U[0,1) draws per level (single jitter).  Returns the per-level state the loss / backward use."""
    c, ws, dt = self.cfg, self.ws, self.dt
    N = rays['origin'].shape[0]
    spacing = {'uniform': 0, 'piecewise': 1, 'reciprocal': 2}[c.proposal_initial_sampler]
    anneal = self.anneal(curr_step)
    bins = ws.get('bins_init', (N, 2))
    bins[:, 0] = 0.; bins[:, 1] = 1.
    weights = ws.get('w_init', (N, 1))
    weights.fill_(1.)
    nb, levels = 1, []
    for lvl in range(self.L + 1):
      is_prop = lvl < self.L
      S = c.num_proposal_samples_per_ray[lvl] if is_prop else c.num_nerf_samples_per_ray
      M = N * S
      if M % 128:
        raise ValueError(f'{N} rays x {S} samples is not a multiple of the 128-row GEMM tile')
      name = self.net_of(lvl) if is_prop else 'field'
      ub, mj = self._u_base(S, u01 is not None)
      jit = None
      if u01 is not None:
        jit = ws.get(f'jit{lvl}', (N,))
        torch.mul(u01[lvl].reshape(-1), mj, out=jit)
      sb, eb = ws.get(f'sb{lvl}', (N, S + 1)), ws.get(f'eb{lvl}', (N, S + 1))
      L.call('hugs_nf_sample', N, nb, S, bins, weights, anneal, c.proposal_histogram_padding, ub, jit, 1, 0., 1., spacing,
             rays['near'], rays['far'], sb, eb)
      x01, sel = ws.get(f'x01_{lvl}', (M, 3)), ws.get(f'sel{lvl}', (M,))
      L.call('hugs_nf_positions', N, S, eb, rays['origin'], rays['direction'], int(c.enable_scene_contraction), c.bound, x01, sel)
      K0, N0 = self.lay.items[f'{name}/w0'][1]
      N1 = self.lay.items[f'{name}/w1'][1][1]
      dens = ws.get(f'dens{lvl}', (M,))
      if self.fused.get(name):
        in_dim, hid = self.lay.items[f'{name}/w0'][2]
        KP = 16 if in_dim <= 16 else 32
        X0 = ws.get(f'X0f_{lvl}', (M, KP), self.tdt)
        if (f'X0fz_{lvl}', M) not in ws.bufs:        # the padding columns are written once and stay zero
          X0.zero_(); ws.bufs[(f'X0fz_{lvl}', M)] = True
        self._grid_fwd(name, x01, X0)
        raw = ws.get(f'raw{lvl}', (M,))
        L.call('hugs_nf_prop_fwd', M, in_dim, hid, dt, X0, KP, self.lay.view(self.flat, f'{name}/w0'), N0,
               self.lay.view(self.flat, f'{name}/b0'), self.lay.view(self.flat, f'{name}/w1'), N1,
               self.lay.view(self.flat, f'{name}/b1'), sel, raw, dens, self.dact, self.dbias)
        st = dict(S=S, M=M, name=name, tag=f'L{lvl}', sbins=sb, ebins=eb, x01=x01, sel=sel, X0=X0, Y0=None, Y1=None, raw=raw, density=dens, rgb=None)
      elif not is_prop and M % 256 == 0 and self._field_fuse_ok():
        st = self._field_forward_fused(lvl, rays, N, S, M, x01, sel, dens, training)
        st.update(sbins=sb, ebins=eb)
      else:
        X0 = ws.get(f'X0_{lvl}', (M, K0), self.tdt)
        if (f'X0z_{lvl}', M) not in ws.bufs:        # the padding columns are written once and stay zero
          X0.zero_(); ws.bufs[(f'X0z_{lvl}', M)] = True
        self._grid_fwd(name, x01, X0)
        Y0, Y1 = ws.get(f'Y0_{lvl}', (M, N0), self.tdt), ws.get(f'Y1_{lvl}', (M, N1), self.tdt)
        # (the dX GEMM that consumes Y0's mask has K = N1: both shapes must suit the bit-mask kernels)
        bY0 = ws.get(f'bitsY0_{lvl}', (M * N0 // 32,), torch.int32) if training and self._bits_ok(M, N0, K0) and self._bits_ok(M, N0, N1) else None
        self._nt(M, f'{name}/w0', X0, self.lay.view(self.flat, f'{name}/b0'), True, Y0, bits=bY0)
        self._nt(M, f'{name}/w1', Y0, self.lay.view(self.flat, f'{name}/b1'), False, Y1)
        L.call('hugs_nf_density_act', M, dt, Y1, N1, 0, sel, dens, self.dact, self.dbias)
        st = dict(S=S, M=M, name=name, sbins=sb, ebins=eb, x01=x01, sel=sel, X0=X0, Y0=Y0, Y1=Y1, density=dens, rgb=None, bY0=bY0)
      rgb_out = None
      if st.get('fused_field'):
        rgb_out = ws.get('rgb_out', (N, 3))
      elif not is_prop:
        sh = ws.get('sh', (N, 16))
        vd01 = ws.get('vd01', (N, 3))
        torch.add(rays['viewdir'], 1.0, out=vd01); vd01.mul_(0.5)
        L.call('hugs_sh4_fwd', N, vd01, 0, 16, 0, sh)
        app = None
        if self.napp:
          app = ws.get('app', (N, self.napp))
          if training or c.eval_embedding == 'original':
            L.call('hugs_glo_gather', N, self.napp, self.lay.view(self.flat, 'appearance'), rays['embed_idx'], 0, app)
          elif c.eval_embedding == 'average':       # eval: every ray sees the mean embedding row (nerfacto.py:272-276)
            app.copy_(self.lay.view(self.flat, 'appearance').mean(dim=0, keepdim=True).expand(N, -1))
          else:
            app.zero_()
        Kh = self.lay.items['field/c0'][1][0]
        H = self.lay.items['field/c0'][1][1]
        Xh = ws.get('Xh', (M, Kh), self.tdt)
        L.call('hugs_nf_head_input', M, S, dt, sh, Y1, N1, c.geo_feat_dim, app, self.napp, Xh, Kh)
        H0, H1 = ws.get('H0', (M, H), self.tdt), ws.get('H1', (M, H), self.tdt)
        bH0 = ws.get('bitsH0', (M * H // 32,), torch.int32) if training and self._bits_ok(M, H, Kh) and self._bits_ok(M, H, H) else None
        self._nt(M, 'field/c0', Xh, self.lay.view(self.flat, 'field/cb0'), True, H0, bits=bH0)
        self._nt(M, 'field/c1', H0, self.lay.view(self.flat, 'field/cb1'), True, H1)
        st['bH0'] = bH0
        rgb = ws.get('rgb_s', (M, 3))
        Yc = None
        if self.rgb_head:
          beff = ws.get('cb2_eff', (4,))
          torch.add(self.lay.view(self.flat, 'field/cb2'), float(c.rgb_bias), out=beff[:3])      # sigmoid(raw + rgb_bias), nerfacto.py:711
          L.call('hugs_rgb_fwd', dt, M, H, H1, H, self.lay.view(self.flat, 'field/c2'), beff, 0.0, rgb)
        else:
          Yc = ws.get('Yc', (M, self.lay.items['field/c2'][1][1]), self.tdt)
          self._nt(M, 'field/c2', H1, self.lay.view(self.flat, 'field/cb2'), False, Yc)
          L.call('hugs_nf_rgb_act', M, dt, Yc, Yc.shape[1], c.rgb_bias, rgb)
        st.update(rgb=rgb, Xh=Xh, H0=H0, H1=H1, Yc=Yc, app=app)
        rgb_out = ws.get('rgb_out', (N, 3))
      w = ws.get(f'w{lvl}', (N, S))
      acc, depth = ws.get(f'acc{lvl}', (N,)), ws.get(f'depth{lvl}', (N,))
      L.call('hugs_nf_weights_fwd', N, S, dens, eb, rays['direction'], int(c.opaque_background), st['rgb'],
             rays.get('bg_rgb') if st['rgb'] is not None else None, w, rgb_out, acc, depth)
      st.update(weights=w, rgb_out=rgb_out, acc=acc, depth=depth)
      levels.append(st)
      bins, weights, nb = sb, w, S
    return levels

  def _field_forward_fused(self, lvl, rays, N, S, M, x01, sel, dens, training):
    """Base network + colour network of the field level in ONE launch (csrc/hugs_fieldfuse.hip k_field_fwd): the same state
    dict as the layer-by-layer path, with the 16-bit raw density column in place of the padded base output."""
    c, ws, dt = self.cfg, self.ws, self.dt
    K0, N0 = self.lay.items['field/w0'][1]
    Kh, H = self.lay.items['field/c0'][1]
    X0 = ws.get(f'X0_{lvl}', (M, K0), self.tdt)
    if (f'X0z_{lvl}', M) not in ws.bufs:        # the padding columns are written once and stay zero
      X0.zero_(); ws.bufs[(f'X0z_{lvl}', M)] = True
    self._grid_fwd('field', x01, X0)
    sh, vd01 = ws.get('sh', (N, 16)), ws.get('vd01', (N, 3))
    torch.add(rays['viewdir'], 1.0, out=vd01); vd01.mul_(0.5)
    L.call('hugs_sh4_fwd', N, vd01, 0, 16, 0, sh)
    app = None
    if self.napp:
      app = ws.get('app', (N, self.napp))
      if training or c.eval_embedding == 'original':
        L.call('hugs_glo_gather', N, self.napp, self.lay.view(self.flat, 'appearance'), rays['embed_idx'], 0, app)
      elif c.eval_embedding == 'average':       # eval: every ray sees the mean embedding row (nerfacto.py:272-276)
        app.copy_(self.lay.view(self.flat, 'appearance').mean(dim=0, keepdim=True).expand(N, -1))
      else:
        app.zero_()
    if getattr(self, '_w1x_stale', True):
      self._refresh_w1x()
    tmpl = ws.get('head_tmpl', (N, 128), self.tdt)
    L.call('hugs_nf_head_template', dt, N, sh, app, c.geo_feat_dim, self.napp, tmpl)
    Y0, raw16, Xh = ws.get(f'Y0_{lvl}', (M, N0), self.tdt), ws.get('raw16', (M, 1), self.tdt), ws.get('Xh', (M, Kh), self.tdt)
    H0, H1 = ws.get('H0', (M, H), self.tdt), ws.get('H1', (M, H), self.tdt)
    bY0 = ws.get(f'bitsY0_{lvl}', (M * N0 // 32,), torch.int32) if training and self._bits_ok(M, N0, 128) else None
    bH0 = ws.get('bitsH0', (M * H // 32,), torch.int32) if training and self._bits_ok(M, H, H) else None
    rgb = ws.get('rgb_s', (M, 3))
    beff = ws.get('cb2_eff', (4,))
    torch.add(self.lay.view(self.flat, 'field/cb2'), float(c.rgb_bias), out=beff[:3])      # sigmoid(raw + rgb_bias), nerfacto.py:711
    V = lambda n: self.lay.view(self.flat, n)
    L.call('hugs_nf_field_fwd', dt, M, S, X0, K0, self.wt['field/w0'], K0, self.wt['field/w1x'], self.wt['field/c0'], self.wt['field/c1'],
           V('field/b0'), self.b1x, V('field/cb0'), V('field/cb1'), V('field/c2'), beff, tmpl, c.geo_feat_dim, sel,
           Y0, raw16, Xh, H0, H1, bY0, bH0, dens, rgb, self.dact, self.dbias)
    return dict(S=S, M=M, name='field', x01=x01, sel=sel, X0=X0, Y0=Y0, Y1=raw16, density=dens, rgb=rgb, bY0=bY0, bH0=bH0,
                Xh=Xh, H0=H0, H1=H1, Yc=None, app=app, fused_field=True)

  # ---- HA-NeRF ImplicitMask (per ray; nerfacto.py:403-408, 1080-1091) -------------------------------------------------
  def _mask_forward(self, rays, N, training):
    c, ws, dt, g, T = self.cfg, self.ws, self.dt, self.mask_grid, self.ntra
    Np = _rup(N)
    tra = ws.get('tra', (N, T))
    if training or c.eval_embedding == 'original':          # Model.get_embedding nerfacto.py:266-284
      L.call('hugs_glo_gather', N, T, self.lay.view(self.flat, 'transient'), rays['embed_idx'], 0, tra)
    elif c.eval_embedding == 'average':
      tra.copy_(self.lay.view(self.flat, 'transient').mean(dim=0, keepdim=True).expand(N, -1))
    else:
      tra.zero_()
    K0, H = self.lay.items['mask/m0'][1]
    X0 = ws.get('mask/X0', (Np, K0), self.tdt)
    if Np != N:
      X0[N:].zero_()
    o, r, sc = g._tables()
    L.call('hugs_hashgrid2d_fwd', N, g.n_levels, g.features, o, r, sc, rays['coord'], self.lay.view(self.flat, 'mask/table'), tra, T,
           dt, K0, X0)
    Y0, Y1 = ws.get('mask/Y0', (Np, H), self.tdt), ws.get('mask/Y1', (Np, H), self.tdt)
    self._nt(Np, 'mask/m0', X0, self.lay.view(self.flat, 'mask/mb0'), True, Y0)
    self._nt(Np, 'mask/m1', Y0, self.lay.view(self.flat, 'mask/mb1'), True, Y1)
    mask = ws.get('mask/out', (N,))
    L.call('hugs_mask_head_fwd', dt, N, H, Y1, H, self.lay.view(self.flat, 'mask/m2').reshape(-1), self.lay.view(self.flat, 'mask/mb2'),
           mask)
    return dict(X0=X0, Y0=Y0, Y1=Y1, mask=mask, Np=Np)

  def _mask_backward(self, st, rays, N, d_mask):
    """Gradients of the ImplicitMask parameters (=) and of the transient embedding rows (+=) from d loss / d mask [N]."""
    ws, dt, g, T = self.ws, self.dt, self.mask_grid, self.ntra
    Np = st['Np']
    K0, H = self.lay.items['mask/m0'][1]
    d_raw = ws.get('mask/d_raw', (Np,))
    L.call('hugs_mask_head_bwd', dt, N, Np, H, st['Y1'], H, st['mask'], d_mask, d_raw, self.lay.view(self.grad, 'mask/m2').reshape(-1),
           self.lay.view(self.grad, 'mask/mb2'))
    G1 = ws.get('mask/G1', (Np, H), self.tdt)
    L.call('hugs_rank1_mask', dt, Np, H, d_raw, self.lay.view(self.flat, 'mask/m2').reshape(-1), st['Y1'], H, G1, H)
    self._tn(Np, 'mask/m1', st['Y0'], G1, 'mask/mb1')
    G0 = ws.get('mask/G0', (Np, H), self.tdt)
    self._nt(Np, 'mask/m1', G1, None, False, G0, mask=st['Y0'], transpose=True)
    self._tn(Np, 'mask/m0', st['X0'], G0, 'mask/mb0')
    dX0 = ws.get('mask/dX0', (Np, K0), self.tdt)
    self._nt(Np, 'mask/m0', G0, None, False, dX0, transpose=True)
    o, r, sc = g._tables()
    L.call('hugs_hashgrid2d_bwd', N, g.n_levels, g.features, o, r, sc, rays['coord'], dX0, dt, K0, self.lay.view(self.grad, 'mask/table'))
    L.call('hugs_embed_scatter_add', dt, N, T, dX0, K0, g.n_output_dims, rays['embed_idx'], self.lay.view(self.grad, 'transient'))

  # ---- loss + backward + Adam ---------------------------------------------------------------------------------------------
  @torch.no_grad()
  def render(self, batch, curr_step, chunk_size=None):
    """Model.forward in eval mode (nerfacto.py:419-428, train.py:244-256): perturb=False, rays in chunks of
    `chunk_size`, embeddings per cfg.eval_embedding.  Returns {'rgb' [N,3], 'accumulation' [N], 'depth' [N]} (the
    training-only weights / bins lists are not returned, as in the reference)."""
    N = batch['origin'].shape[0]
    cs = N if not chunk_size else int(chunk_size)
    cs = max(128, (cs + 127) // 128 * 128)          # whole GEMM tiles per chunk; the last chunk is padded with its last ray
    out = {k: [] for k in ('rgb', 'accumulation', 'depth') + (('implicit_mask',) if self.ntra else ())}
    for lo in range(0, N, cs):
      hi = min(N, lo + cs)
      n = hi - lo
      npad = (n + 127) // 128 * 128
      sub = {}
      for k, v in batch.items():
        if not torch.is_tensor(v) or v.shape[0] != N:
          continue
        sl = v[lo:hi]
        if npad != n:
          sl = torch.cat([sl, sl[-1:].expand(npad - n, *sl.shape[1:])], 0)
        sub[k] = sl.contiguous()
      fin = self.forward(sub, curr_step, None, training=False)[-1]
      out['rgb'].append(fin['rgb_out'][:n].clone()); out['accumulation'].append(fin['acc'][:n].clone())
      out['depth'].append(fin['depth'][:n].clone())
      if self.ntra:
        out['implicit_mask'].append(self._mask_forward(sub, npad, False)['mask'][:n].clone())
    return {k: torch.cat(v, 0) for k, v in out.items()}

  def proposal_update_enabled(self, curr_step):
    """nerfacto.py:299-303."""
    c = self.cfg
    iv = int(np.clip(np.interp(curr_step, [0, c.proposal_warmup], [0, c.proposal_update_every]), 1, c.proposal_update_every))
    return (curr_step % iv) == 0

  def train_step(self, batch, curr_step=None, u01=None, apply_update=True, world=1, inlier_threshold=None, is_finetune=False):
    """One optimisation step (train.py:196-215 + Loss.forward nerfacto.py:598-640).  batch: rays dict + 'rgb' [N,3]
    (+ 'static_mask' [N] for transient_type='withmask'; whole 16x16 patches, patch-major, for 'robustnerf', whose stats
    slots 10..14 hold next inlier_threshold, is_inlier_loss, has_inlier_neighbors, is_inlier_patch, robust_mask).
    'hanerf': batch also holds 'coord' [N,2] (the ray's image coordinate in [0,1]^2, nerfacto.py:404); stats slots 12 / 13 =
    mask_size_loss / mean mask.  is_finetune: Loss.forward's finetune branch (plain data loss whatever the transient
    type, nerfacto.py:606-609); which parameters move is begin_finetune()'s list.
    Returns a dict of host-lazy device scalars."""
    c, ws, dt = self.cfg, self.ws, self.dt
    self.step += 1
    step = self.step if curr_step is None else curr_step
    N = batch['origin'].shape[0]
    levels = self.forward(batch, step, u01)
    fin = levels[-1]
    Sf = fin['S']
    stats = ws.get('stats', (16,))
    stats.zero_()
    # rgb loss (mean, or the static-mask weighted mean of compute_withmask_loss nerfacto.py:467-490)
    d_pred = ws.get('d_pred', (1, N, 3))
    tt = None if is_finetune else c.transient_type
    mode, lm = (1, batch['static_mask']) if tt == 'withmask' else (0, None)
    mask_st, d_mask = None, None
    if tt == 'hanerf':
      mask_st = self._mask_forward(batch, N, True)
    if tt == 'robustnerf':
      # compute_robustnerf_loss (nerfacto.py:492-527): rays come in whole patch_size^2 patches (train.py:188-192); the
      # inlier threshold is last step's quantile (extra_infos, 1.0 before the first step), fed back on the device
      P = c.patch_size
      if N % (P * P):
        raise ValueError('robustnerf needs whole patches per device')
      thr = inlier_threshold if inlier_threshold is not None else getattr(self, '_robust_thr', None)
      if thr is None:
        thr = torch.ones(1, device=self.device)
      elif not torch.is_tensor(thr):
        thr = torch.full((1,), float(thr), device=self.device)
      rmask, rerr, rpart = ws.get('robust_mask', (1, N)), ws.get('robust_err', (N,)), ws.get('robust_part', (N // (P * P) * 4,))
      L.call('hugs_nf_robust_mask', N // (P * P), P, fin['rgb_out'], batch['rgb'], thr, c.robustnerf_inlier_quantile,
             c.robustnerf_smoothed_filter_size, c.robustnerf_smoothed_inlier_quantile, c.robustnerf_inner_patch_size,
             c.robustnerf_inner_patch_inlier_quantile, rmask[0], rerr, rpart, stats[10:15])
      self._robust_thr = stats[10:11].clone()
      mode, lm = 2, rmask
    # hugs_data_loss' static-mask mode normalises by the [N,1] mask sum (Mip-NeRF 360's train_utils.py quirk); nerfacto's
    # compute_withmask_loss broadcasts the mask to the 3 channels first (nerfacto.py:481-483): a factor 3 in the denominator
    chan = 3.0 if mode == 1 else 1.0     # (the kernel's other modes already count the 3 channels)
    coef = ws.bufs.setdefault(('coef1', chan), torch.full((1,), float(c.rgb_loss_mult) / chan, device=self.device))
    if tt == 'hanerf':
      # compute_hanerf_loss (nerfacto.py:560-596): mean((1 - mask) * loss) + mult(curr_step) * mean(mask^2)
      msm = max(c.hanerf_mask_size_loss_mult_min, c.hanerf_mask_size_loss_mult_max * math.exp(-step * c.hanerf_mask_size_loss_mult_k))
      d_mask, hst = ws.get('d_mask', (N,)), ws.get('hanerf_stats', (4,))
      L.call('hugs_hanerf_loss', N, 1, fin['rgb_out'].reshape(1, N, 3), batch['rgb'], mask_st['mask'], int(c.rgb_loss_type == 'charb'),
             c.rgb_charb_loss_padding, coef, msm, d_pred, d_mask, hst)
      stats[0:2].copy_(hst[0:2])
      torch.mul(hst[2:3], msm, out=stats[12:13])
      stats[13:14].copy_(hst[3:4])
    else:
      L.call('hugs_data_loss', N, 1, fin['rgb_out'].reshape(1, N, 3), batch['rgb'], lm, mode, c.withmask_transient_weight,
             int(c.rgb_loss_type == 'charb'), c.rgb_charb_loss_padding, coef, d_pred, stats[0:2])
      if chan != 1.0:
        stats[0:2].mul_(1.0 / chan)
    loss_ray = ws.get('loss_ray', (N,))
    d_w = [None] * (self.L + 1)
    prop_on = self.proposal_update_enabled(step)
    if c.interlevel_loss_mult > 0:
      for l in range(self.L):
        d_w[l] = ws.get(f'd_w{l}', (N, levels[l]['S']))
        L.call('hugs_nf_interlevel', N, Sf, levels[l]['S'], fin['sbins'], fin['weights'], levels[l]['sbins'], levels[l]['weights'],
               c.interlevel_loss_mult / (N * Sf), loss_ray, d_w[l])
        L.call('hugs_sum', N, loss_ray, c.interlevel_loss_mult / (N * Sf), stats[2 + l:3 + l])
    if c.distortion_loss_mult > 0:
      d_w[self.L] = ws.get('d_w_fin', (N, Sf))
      L.call('hugs_distortion', N, Sf, fin['sbins'], fin['weights'], c.distortion_loss_mult / N, loss_ray, d_w[self.L])
      L.call('hugs_sum', N, loss_ray, c.distortion_loss_mult / N, stats[8:9])
    # ---- backward ----
    if self.amp:
      # scaler.scale(loss).backward(): every gradient seed times the current scale (a device scalar: no host read); the
      # loss VALUES above are unscaled.  Adam divides it out again (apply_gradients)
      sc = self.amp_state[0:1]
      d_pred.mul_(sc)
      for d in d_w:
        if d is not None:
          d.mul_(sc)
      if d_mask is not None:
        d_mask.mul_(sc)
    self.grad.zero_()
    want = lambda group: self.trainable is None or group in self.trainable
    # The levels' backward passes only share the loss gradients: the field's chain (GEMMs: HBM-bound) and the proposal levels'
    # (fused MLP + table scatter: atomic-bound) run on separate streams and fill each other's idle units (a single stream kept
    # the GPU 100 % busy with ONE kernel at a time).  HUGS_NF_BWD_STREAMS=0: everything on the caller's stream.
    cur = torch.cuda.current_stream()
    multi = os.environ.get('HUGS_NF_BWD_STREAMS', '1') != '0'
    if multi and not hasattr(self, '_bwd_streams'):
      self._bwd_streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
    ev0 = torch.cuda.Event(); ev0.record(cur)
    done = []
    self._bwd_done = []
    self._prop_grad_written = set()
    same_net = bool(self.cfg.use_same_proposal_network)
    for l in range(self.L, -1, -1):
      st = levels[l]
      is_prop = l < self.L
      if is_prop and (d_w[l] is None or not prop_on or not want('proposal')):
        continue
      if not is_prop and not (want('field') or want('appearance_embedding')):
        continue
      if multi and is_prop:
        side = self._bwd_streams[0 if same_net else l % 2]      # (levels of one shared network: one stream, in order)
        with torch.cuda.stream(side):
          side.wait_event(ev0)
          self._backward_level(st, batch, N, None, d_w[l])
          e = torch.cuda.Event(); e.record(side); done.append(e)
      else:
        self._backward_level(st, batch, N, d_pred[0] if not is_prop else None, d_w[l])
    if mask_st is not None:
      if multi:
        side = self._bwd_streams[2]
        with torch.cuda.stream(side):
          side.wait_event(ev0)
          self._mask_backward(mask_st, batch, N, d_mask)
          e = torch.cuda.Event(); e.record(side); done.append(e)
      else:
        self._mask_backward(mask_st, batch, N, d_mask)
    for e in done + self._bwd_done:
      cur.wait_event(e)
    self._bwd_done = None
    if world > 1:
      import torch.distributed as dist
      dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
      self.grad.mul_(1.0 / world)
    self._prop_updated = bool(prop_on and any(d is not None for d in d_w[:self.L]))
    if apply_update:
      self.apply_gradients(self._prop_updated)
    return dict(stats=stats, levels=levels, mask=None if mask_st is None else mask_st['mask'])

  def _opt(self):
    if not hasattr(self, 'opt'):
      c = self.cfg
      self.opt = dict(lr_init=c.lr_init, lr_final=c.lr_final, lr_decay_mult=c.lr_decay_mult, warmup_steps=c.warmup_steps,
                      num_steps=c.num_steps, betas=tuple(c.opt_betas), eps=c.opt_eps)
    return self.opt

  def lr(self, step):
    """utils/lr_scheduler_utils.py:6-27 get_warmup_decay_scheduler: lr_init x the LambdaLR factor at `step`."""
    o = self._opt()
    if step < o['warmup_steps']:
      f = o['lr_decay_mult'] + (1 - o['lr_decay_mult']) * math.sin(0.5 * math.pi * min(max(step / o['warmup_steps'], 0), 1))
    else:
      t = min(max((step - o['warmup_steps']) / (o['num_steps'] - o['warmup_steps']), 0), 1)
      f = math.exp(math.log(o['lr_init']) * (1 - t) + math.log(o['lr_final']) * t) / o['lr_init']
    return o['lr_init'] * f

  def begin_finetune(self, params=('appearance_embedding',), lr_init=5e-3, lr_final=5e-4, lr_decay_mult=0.01, warmup_steps=500,
                     num_steps=5000, betas=(0.9, 0.999), eps=1e-8):
    """The reference's second training stage (train.py:126-165): a NEW Adam + scheduler over the listed parameter groups
    only (`finetune_params`, the yml default is [appearance_embedding]); everything else keeps its value."""
    for p in params:
      if p not in self.groups:
        raise KeyError(f'finetune_params: {p!r} is not a parameter group (have {sorted(self.groups)})')      # train.py:160 params_dict[key]
    self.trainable = tuple(params)
    self.opt = dict(lr_init=lr_init, lr_final=lr_final, lr_decay_mult=lr_decay_mult, warmup_steps=warmup_steps, num_steps=num_steps,
                    betas=tuple(betas), eps=eps)
    self.m.zero_(); self.v.zero_()
    self.counts, self._updates = {}, 0
    if self.amp:
      self.amp_counts.zero_()              # a new optimizer and a new GradScaler per stage (train.py:159-168)
      self.amp_state.copy_(torch.tensor([65536.0, 0.0, 0.0]))

  def apply_gradients(self, prop_updated=True):
    """optimizer.step() + scheduler.step() (train.py:213-215): the k-th scheduler step's factor applies to update k.
    torch.optim.Adam skips a parameter whose .grad is None completely -- no update, no moment decay, no increment of its
    own `step` -- and that is what the proposal networks are on steps without a proposal update (their forward runs
    under set_grad_enabled(False), nerfacto.py:338, and zero_grad() resets to None): the 'proposal' range is left out
    and keeps its own update count for the bias corrections."""
    o = self._opt()
    k_sched = getattr(self, '_updates', 0)
    lr, (b1, b2) = self.lr(k_sched), o['betas']
    if self.amp:
      return self._apply_gradients_amp(prop_updated, lr, b1, b2, o['eps'], k_sched)
    todo = []
    for gname, (lo, hi) in sorted(self.groups.items(), key=lambda kv: kv[1][0]):
      if hi <= lo or (self.trainable is not None and gname not in self.trainable) or (gname == 'proposal' and not prop_updated):
        continue
      k = self.counts.get(gname, 0)
      self.counts[gname] = k + 1
      if todo and todo[-1][1] == lo and todo[-1][2] == k:
        todo[-1][1] = hi                      # neighbouring groups with the same update count: one launch
      else:
        todo.append([lo, hi, k])
    for lo, hi, k in todo:
      L.call('hugs_nf_adam', hi - lo, self.flat[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], lr, b1, b2, o['eps'],
             1.0 - b1**(k + 1), 1.0 - b2**(k + 1))
    self._updates = k_sched + 1
    self.refresh_weights()

  def _apply_gradients_amp(self, prop_updated, lr, b1, b2, eps, k_sched):
    """scaler.step(optimizer); scaler.update(); scheduler.step() (train.py:211-214) without a host read: the inf check, the
    skip, the per-group update counts and the scale update all stay on the device (csrc/hugs_nerfacto.hip k_amp_*).  The
    scheduler advances whether or not the step was skipped, as the reference's does."""
    part = [(gi, gname) + self.groups[gname] for gi, gname in enumerate(self.group_order)
            if self.groups[gname][1] > self.groups[gname][0] and (self.trainable is None or gname in self.trainable)
            and not (gname == 'proposal' and not prop_updated)]
    for gi, gname, lo, hi in part:
      L.call('hugs_amp_check', hi - lo, self.grad[lo:hi], self.amp_state)
    L.call('hugs_amp_prepare', len(self.group_order), self.amp_counts, b1, b2, self.amp_bc)
    mask = 0
    for gi, gname, lo, hi in part:
      L.call('hugs_nf_adam_amp', hi - lo, self.flat[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], lr, b1, b2, eps,
             self.amp_state, self.amp_bc[2 * gi:2 * gi + 2])
      mask |= 1 << gi
    a = self.amp_opts
    L.call('hugs_amp_update', self.amp_state, self.amp_counts, mask, a['growth_factor'], a['backoff_factor'], float(a['growth_interval']))
    self._updates = k_sched + 1
    self.refresh_weights()

  def loss_scale(self):
    """GradScaler.get_scale() (fp16 mode; a host read)."""
    return float(self.amp_state[0]) if self.amp else 1.0

  def _field_backward_fused(self, st, rays, N, S, M, G1, d_dens):
    """Colour layer 1 gradient -> hash-feature gradient in ONE launch (csrc/hugs_fieldfuse.hip k_field_bwd), then the four
    weight-gradient GEMMs on G1 and the G operands it wrote.  The base network's second layer runs in head-input column order
    there (W1x): its weight / bias gradients come out in that order and are moved to the layout's columns."""
    c, ws, dt = self.cfg, self.ws, self.dt
    K0, N0 = self.lay.items['field/w0'][1]
    Kh, H = self.lay.items['field/c0'][1]
    N1, g = self.lay.items['field/w1'][1][1], c.geo_feat_dim
    G0, Gb, Gy0 = ws.get('G0', (M, H), self.tdt), ws.get('Gb_field', (M, N1), self.tdt), ws.get('Gy0_field', (M, N0), self.tdt)
    dx32 = int(self.grid_grad_f32)
    dX0 = ws.get('dX0_field', (M, K0), torch.float32 if dx32 else self.tdt)
    L.call('hugs_nf_field_bwd', dt, M, S, G1, self.wn['field/c1'], self.wn['field/c0'], self.w1xn, self.wn['field/w0'], st['bH0'], st['bY0'],
           d_dens, st['sel'], st['Y1'], g, self.napp, rays['embed_idx'] if self.napp else None, G0, Gb, Gy0, dX0, K0,
           self.lay.view(self.grad, 'appearance') if self.napp else None, dx32, self.dact, self.dbias)
    # the field grid's table gradient (atomic-bound, 1.8 ms) needs only dX0: on its own stream next to the four weight-gradient
    # GEMMs (HBM-bound) instead of behind them
    side = None
    if os.environ.get('HUGS_NF_BWD_STREAMS', '1') != '0' and os.environ.get('HUGS_NF_GRID_SIDE', '1') != '0':
      if not hasattr(self, '_grid_stream'):
        self._grid_stream = torch.cuda.Stream(device=self.device)
      side, cur = self._grid_stream, torch.cuda.current_stream()
      ev = torch.cuda.Event(); ev.record(cur)
      with torch.cuda.stream(side):
        side.wait_event(ev)
        self._grid_bwd('field', st['x01'], dX0)
        e = torch.cuda.Event(); e.record(side)
      if getattr(self, '_bwd_done', None) is not None:
        self._bwd_done.append(e)      # (train_step joins it with the levels' streams)
      else:
        cur.wait_event(e)
    self._tn(M, 'field/c1', st['H0'], G1, 'field/cb1')
    self._tn(M, 'field/c0', st['Xh'], G0, 'field/cb0')
    tw, tb = ws.get('gw1x', (N0, N1)), ws.get('gb1x', (N1,))
    self._tn(M, 'field/w1', st['Y0'], Gb, 'field/b1', out=tw, out_bias=tb)
    gw, gb = self.lay.view(self.grad, 'field/w1'), self.lay.view(self.grad, 'field/b1')
    gw[:, 0:1].copy_(tw[:, 0:1]); gw[:, 1:1 + g].copy_(tw[:, 16:16 + g]); gw[:, 1 + g:].zero_()
    gb[0:1].copy_(tb[0:1]); gb[1:1 + g].copy_(tb[16:16 + g]); gb[1 + g:].zero_()
    self._tn(M, 'field/w0', st['X0'], Gy0, 'field/b0')
    if side is None:
      self._grid_bwd('field', st['x01'], dX0)

  def _backward_level(self, st, rays, N, d_rgb_out, d_w_extra):
    c, ws, dt = self.cfg, self.ws, self.dt
    S, M, name = st['S'], st['M'], st['name']
    d_dens = ws.get(f"d_dens_{st.get('tag', name)}", (M,))
    d_rgb_s = ws.get('d_rgb_s', (M, 3)) if st['rgb'] is not None else None
    L.call('hugs_nf_weights_bwd', N, S, st['density'], st['ebins'], rays['direction'], int(c.opaque_background), st['rgb'],
           rays.get('bg_rgb') if st['rgb'] is not None else None, st['weights'], d_rgb_out, d_w_extra, d_dens, d_rgb_s)
    if st.get('raw') is not None:       # fused proposal net: d density -> (dX0, weight gradients) in one kernel
      in_dim, hid = self.lay.items[f'{name}/w0'][2]
      N0 = self.lay.items[f'{name}/w0'][1][1]
      N1 = self.lay.items[f'{name}/w1'][1][1]
      KP = st['X0'].shape[1]
      tag = st.get('tag', name)
      dx32 = int(bool(self.dt) and self.grid_grad_f32 and KP == 16)      # (the matrix-core kernel writes either form)
      dX0 = ws.get(f'dX0f_{tag}', (M, KP), torch.float32 if dx32 else self.tdt)
      slab = ws.get(f'prop_slab_{tag}', (L.lib().cdll.hugs_nf_prop_ws_bytes(in_dim) // 4,))      # (per level: the levels run concurrently)
      gv = lambda leaf: self.lay.view(self.grad, f'{name}/{leaf}')
      # use_same_proposal_network: the second level that reaches a shared network ADDS its weight gradients (the kernel writes =):
      # through a scratch copy of the four leaves (the table gradient is scatter-added either way)
      accumulate = name in getattr(self, '_prop_grad_written', ())
      tgt = {leaf: (ws.get(f'prop_gtmp_{leaf}', tuple(gv(leaf).shape)) if accumulate else gv(leaf)) for leaf in ('w0', 'b0', 'w1', 'b1')}
      L.call('hugs_nf_prop_bwd', M, in_dim, hid, dt, st['X0'], KP, self.lay.view(self.flat, f'{name}/w0'), N0,
             self.lay.view(self.flat, f'{name}/b0'), self.lay.view(self.flat, f'{name}/w1'), N1, st['raw'], st['sel'], d_dens,
             dX0, tgt['w0'], tgt['b0'], tgt['w1'], tgt['b1'], slab, dx32, self.dact, self.dbias)
      if accumulate:
        for leaf in ('w0', 'b0', 'w1', 'b1'):
          L.call('hugs_add_inplace', gv(leaf).numel(), tgt[leaf], gv(leaf))
      elif hasattr(self, '_prop_grad_written'):
        self._prop_grad_written.add(name)
      self._grid_bwd(name, st['x01'], dX0)
      return
    if self.cfg.use_same_proposal_network and name != 'field':
      raise NotImplementedError('use_same_proposal_network with proposal networks wider than the fused kernels (in <= 32, hidden <= 64)')
    N1 = self.lay.items[f'{name}/w1'][1][1]
    dXh = None
    if st['rgb'] is not None:
      H = st['H0'].shape[1]
      G1 = ws.get('G1', (M, H), self.tdt)
      if self.rgb_head:
        rws = ws.get('rgb_bwd_ws', (L.lib().cdll.hugs_rgb_bwd_ws_bytes() // 4,))
        L.call('hugs_rgb_bwd', dt, M, H, st['H1'], H, self.lay.view(self.flat, 'field/c2'), st['rgb'], d_rgb_s, 0.0, G1, H,
               self.lay.view(self.grad, 'field/c2'), self.lay.view(self.grad, 'field/cb2'), rws)
      else:
        Nc = st['Yc'].shape[1]
        Gc = ws.get('Gc', (M, Nc), self.tdt)
        L.call('hugs_nf_rgb_grad', M, dt, st['rgb'], d_rgb_s, Gc, Nc)
        self._tn(M, 'field/c2', st['H1'], Gc, 'field/cb2')
        self._nt(M, 'field/c2', Gc, None, False, G1, mask=st['H1'], transpose=True)
      if (st.get('fused_field') and S % 64 == 0 and st.get('bH0') is not None and st.get('bY0') is not None and
          os.environ.get('HUGS_NF_FIELD_FUSE_BWD', '1') != '0'):
        self._field_backward_fused(st, rays, N, S, M, G1, d_dens)      # (runs all four weight-gradient GEMMs behind its launch)
        return
      self._tn(M, 'field/c1', st['H0'], G1, 'field/cb1')
      G0 = ws.get('G0', (M, H), self.tdt)
      self._nt(M, 'field/c1', G1, None, False, G0, mask=st['H0'], transpose=True, bits=st.get('bH0'))
      self._tn(M, 'field/c0', st['Xh'], G0, 'field/cb0')
      Kh = st['Xh'].shape[1]
      dXh = ws.get('dXh', (M, Kh), self.tdt)
      self._nt(M, 'field/c0', G0, None, False, dXh, transpose=True)
      if self.napp:
        L.call('hugs_nf_app_bwd', N, S, dt, dXh, Kh, 16 + c.geo_feat_dim, self.napp, rays['embed_idx'],
               self.lay.view(self.grad, 'appearance'))
    Gb = ws.get(f'Gb_{name}', (M, N1), self.tdt)
    L.call('hugs_nf_base_grad', M, dt, st['Y1'], st['Y1'].shape[1], st['sel'], d_dens, dXh, 0 if dXh is None else dXh.shape[1], 16,
           c.geo_feat_dim if dXh is not None else 0, Gb, N1, self.dact, self.dbias)
    self._tn(M, f'{name}/w1', st['Y0'], Gb, f'{name}/b1')
    N0 = st['Y0'].shape[1]
    Gy0 = ws.get(f'Gy0_{name}', (M, N0), self.tdt)
    self._nt(M, f'{name}/w1', Gb, None, False, Gy0, mask=st['Y0'], transpose=True, bits=st.get('bY0'))
    self._tn(M, f'{name}/w0', st['X0'], Gy0, f'{name}/b0')
    K0 = st['X0'].shape[1]
    dX0 = ws.get(f'dX0_{name}', (M, K0), self.tdt)
    self._nt(M, f'{name}/w0', Gy0, None, False, dX0, transpose=True)
    self._grid_bwd(name, st['x01'], dX0)

"""Kernel orchestration for one Model: parameter layout on a flat fp32 buffer, compute-dtype weight
copies, the per-level forward (models.py:155-312 of the reference) and the hand-scheduled backward
that replaces jax.value_and_grad (train_utils.py:454).  Python only sequences C-ABI launches on the
current HIP stream; every tensor op on the path is a kernel from csrc/."""
import math

import numpy as np
import torch

from .. import _lib
from . import geopoly
from . import stepfun

CHUNK = 16384
_USE_BITS = __import__('os').environ.get('HUGS_RELU_BITS', '1') != '0'        # A/B knob: 1-bit relu masks for the dX GEMMs
# Trunk weight gradients (A/B switch): 0 = one hugs_gemm_tn per layer next to the dX chain (rounds 1-3), 1 (default) = ONE
# batched launch of all layers after the dX chain, 2 = two batched launches (upper half of the layers under the lower half's
# dX GEMMs, the rest after the chain)
_TN_BATCH = int(__import__('os').environ.get('HUGS_TN_BATCH', '1'))
_DW_AFTER_PROP = __import__('os').environ.get('HUGS_DW_AFTER_PROP', '1') == '1'
# G_last through the folded matrix P = W_bottleneck W_view[:Bw] (one K = 128 GEMM on the critical chain instead of two; A/B switch)
_MLP_FUSE_ROWS = int(__import__('os').environ.get('HUGS_MLP_FUSE_ROWS', '32768'))      # 0: never fuse the 256-wide trunk tail
_MLP_CHAIN3 = __import__('os').environ.get('HUGS_MLPFUSE_CHAIN3', '1') != '0'      # (the library reads the same switch)
_HEAD_FOLD = __import__('os').environ.get('HUGS_HEAD_FOLD', '1') == '1'
_COMPOSITE_RAW = __import__('os').environ.get('HUGS_COMPOSITE_RAW', '1') == '1'      # A/B: d_raw out of the compositing backward
_RGB_REDUCE_SIDE = __import__('os').environ.get('HUGS_RGB_REDUCE_SIDE', '1') == '1'      # A/B: the view head's dW reduction on the head stream
_SIDE_LATE = __import__('os').environ.get('HUGS_SIDE_LATE', '0') == '1'      # A/B: side-stream work released behind the G_last GEMM
_TN_ITEM = np.dtype([('X', np.uint64), ('G', np.uint64), ('dW', np.uint64), ('db', np.uint64), ('ldx', np.int32), ('ldg', np.int32),
                     ('Mrows', np.int32), ('Kc', np.int32), ('N', np.int32), ('reserved', np.int32)])      # include/hugs.h HugsTnItem
# HUGS_NT_CHAIN=0|1: the trunk layers of a wide MLP (bf16, 1-bit relu masks, whole 256-row bands per CU round) as ONE persistent launch that
# walks layer -> tile (hugs_gemm_nt_chain) instead of one launch per layer; outputs bit-identical to the per-layer launches
_NT_CHAIN = __import__('os').environ.get('HUGS_NT_CHAIN', '0') != '0'
_CHUNK_BYTES = int(float(__import__('os').environ.get('HUGS_FWD_CHUNK_MB', '1e9')) * 1e6)   # forward row-chunk size (A/B knob)


KEEP_EVENTS = None      # a list while a train step is being captured into a hipGraph: its events must outlive the capture


class _Event(torch.cuda.Event):
  """torch.cuda.Event that remembers the stream it was recorded on (wait_event below skips a stream's wait on itself)."""

  def record(self, stream=None):
    stream = torch.cuda.current_stream() if stream is None else stream
    self.sid = stream.cuda_stream
    super().record(stream)


def new_event():
  e = _Event()
  if KEEP_EVENTS is not None:
    KEEP_EVENTS.append(e)
  return e


def wait_event(stream, ev):
  """stream.wait_event(ev), except when `ev` was recorded on `stream` itself: stream order already holds there, and a
  hipGraph capture on ROCm 7.2 dies in hipStreamEndCapture when a FORKED stream waits on (or forks again from) an event
  of its own (scratch/graph_probe2.py: lanes whose events stay on the origin stream capture fine)."""
  if getattr(ev, 'sid', None) != stream.cuda_stream:
    stream.wait_event(ev)


def _round_up(x, m):
  return (x + m - 1) // m * m


class MLPSpec:
  """Static description of one MLP (models.py:359-391 attributes + gin bindings)."""

  def __init__(self, name, is_prop, num_glo, num_transient=0, use_viewdirs=True, **kw):
    self.name = name
    self.is_prop = is_prop
    self.use_viewdirs = bool(use_viewdirs)      # Model.use_viewdirs (models.py:56,233): False -> the rgb head sits on the trunk
    self.net_depth = 8
    self.net_width = 256
    self.bottleneck_width = 256
    self.net_depth_viewdirs = 1
    self.net_width_viewdirs = 128
    self.skip_layer_dir = 4
    self.min_deg_point = 0
    self.max_deg_point = 12
    self.skip_layer = 4
    self.num_rgb_channels = 3
    self.deg_view = 4
    self.density_bias = -1.
    self.rgb_premultiplier = 1.
    self.rgb_bias = 0.
    self.rgb_padding = 0.001
    self.disable_rgb = False
    self.warp_fn = None
    self.basis_shape = 'icosahedron'
    self.basis_subdivisions = 2
    self.weight_init = 'he_uniform'
    self.bottleneck_noise = 0.0
    self.density_noise = 0.
    self.net_depth_transient = 4      # models.py:367-375 (NeRF-W branch of the NerfMLP)
    self.net_width_transient = 128
    self.skip_layer_transient = 4
    for k, v in kw.items():
      if not hasattr(self, k):
        raise ValueError(f'{name} has no attribute {k!r}')
      setattr(self, k, v)
    self.num_glo = 0 if is_prop else num_glo
    self.num_tra = 0 if is_prop else num_transient      # > 0 <=> disable_transient=False (models.py:104)
    self._check()
    self.basis = geopoly.generate_basis(self.basis_shape, self.basis_subdivisions).T.astype(np.float32).copy()  # [3,nb]
    self.nb = self.basis.shape[1]
    self.F = 2 * self.nb * (self.max_deg_point - self.min_deg_point)
    self.Fp = _round_up(self.F, 128)      # whole 128-wide K tiles of the weight-gradient GEMM (504 -> 512 at the default degree 12)
    self.nd = 3 + 6 * self.deg_view
    # Dense layers in flax creation order: (name, fan_in, fan_in_padded, fan_out, kind)
    # Trunk widths that are not a multiple of the 128-column MFMA tile (debug.gin's 64-wide PropMLP) are padded with
    # zero output columns / zero bias: relu(0) = 0 feeds zero rows of the next kernel, and every gradient of the
    # padding is exactly 0, so it stays 0 under Adam.
    self.Wp = _round_up(self.net_width, 128)
    L, k, kp = [], self.F, self.Fp
    for i in range(self.net_depth):
      concat_in = i > 0 and (i - 1) % self.skip_layer == 0 and (i - 1) > 0
      L.append(dict(fan_in=k, kpad=(self.Wp + self.Fp) if concat_in else kp, fan_out=self.net_width, npad=self.Wp,
                    kind='trunk', concat=concat_in))
      k = self.net_width + self.F if (i % self.skip_layer == 0 and i > 0) else self.net_width
      kp = self.Wp
    L.append(dict(fan_in=k, kpad=kp, fan_out=1, kind='density'))
    if not self.disable_rgb and not self.use_viewdirs:
      # models.py:486-516 with viewdirs=None: no bottleneck, no view layer -- rgb = activation(Dense(3)(trunk output))
      L.append(dict(fan_in=k, kpad=kp, fan_out=self.num_rgb_channels, kind='rgb'))
    elif not self.disable_rgb:
      L.append(dict(fan_in=k, kpad=kp, fan_out=self.bottleneck_width, kind='bottleneck'))
      kv = self.bottleneck_width + self.nd + self.num_glo
      L.append(dict(fan_in=kv, kpad=kv, fan_out=self.net_width_viewdirs, kind='view'))
      for _ in range(1, self.net_depth_viewdirs):      # models.py:508-512: further Dense(net_width_viewdirs) + relu layers (round 5)
        L.append(dict(fan_in=self.net_width_viewdirs, kpad=self.net_width_viewdirs, fan_out=self.net_width_viewdirs, kind='vtrunk'))
      L.append(dict(fan_in=self.net_width_viewdirs, kpad=self.net_width_viewdirs, fan_out=self.num_rgb_channels, kind='rgb'))
    self.t0 = len(L)          # index of the first transient layer (flax creates them after the rgb head, models.py:521-539)
    if self.num_tra > 0:
      Ht, kt = self.net_width_transient, self.bottleneck_width + self.num_tra
      L.append(dict(fan_in=kt, kpad=kt, fan_out=Ht, kind='tview'))
      for _ in range(1, self.net_depth_transient):
        L.append(dict(fan_in=Ht, kpad=Ht, fan_out=Ht, kind='ttrunk'))
      L.append(dict(fan_in=Ht, kpad=Ht, fan_out=1, kind='tdensity'))
      L.append(dict(fan_in=Ht, kpad=Ht, fan_out=self.num_rgb_channels, kind='trgb'))
      L.append(dict(fan_in=Ht, kpad=Ht, fan_out=1, kind='tuncert'))
    for i, l in enumerate(L):
      l['name'] = f'Dense_{i}'
    self.layers = L

  def head_layers(self):
    """(bottleneck, view, [further view layers], rgb) of an MLP with the view branch."""
    nd, dv = self.net_depth, self.net_depth_viewdirs
    return self.layers[nd + 1], self.layers[nd + 2], self.layers[nd + 3:nd + 2 + dv], self.layers[nd + 2 + dv]

  def _check(self):
    d = self.net_depth
    if (d - 1) > 0 and (d - 1) % self.skip_layer == 0:
      raise NotImplementedError('a skip-concat after the last trunk layer is not built')
    if not self.use_viewdirs and self.num_tra > 0:
      raise NotImplementedError('the NeRF-W transient branch hangs off the bottleneck, which use_viewdirs=False does not create (models.py:521-524)')
    if not self.disable_rgb and self.use_viewdirs and (self.bottleneck_width % 128 or self.net_width_viewdirs != 128):
      raise NotImplementedError('bottleneck_width must be a multiple of 128 and net_width_viewdirs == 128 (MFMA tiles)')
    if self.net_width % 128 and self.net_depth > self.skip_layer + 1:
      raise NotImplementedError('a trunk width that is not a multiple of 128 together with a skip concat is not built')
    if self.num_rgb_channels != 3:
      raise NotImplementedError('num_rgb_channels != 3 is not built')
    if self.net_depth_viewdirs < 1 or self.net_depth_viewdirs > self.skip_layer_dir:
      # models.py:511: x = concat(x, inputs) after view layer i when i % skip_layer_dir == 0 and i > 0, i.e. from depth skip_layer_dir + 1
      raise NotImplementedError('1 <= net_depth_viewdirs <= skip_layer_dir (the view MLP\'s skip concat is not built)')
    if self.net_depth_viewdirs > 1 and (not self.use_viewdirs or self.disable_rgb):
      pass      # (no view MLP is created: models.py:486)
    if not 0 <= self.min_deg_point < self.max_deg_point:
      raise ValueError(f'min_deg_point {self.min_deg_point} / max_deg_point {self.max_deg_point}')
    if self.num_tra > 0 and (self.disable_rgb or self.net_width_transient != 128 or self.net_depth_transient < 2 or
                             self.net_depth_transient > self.skip_layer_transient):
      raise NotImplementedError('transient MLP: width 128, 2 <= depth <= skip_layer_transient, rgb branch enabled')
    if self.warp_fn is not None and getattr(self.warp_fn, 'name', self.warp_fn) != 'coord.contract':
      raise NotImplementedError(f'warp_fn {self.warp_fn!r}: only @coord.contract is built')


class MaskSpec:
  """HA-NeRF's per-ray ImplicitMask MLP (models.py:651-674): [pos_enc(pix_coords) | tra_vec] -> (Dense+relu) x depth
  -> sigmoid(Dense(1)).  The input is zero-padded to a multiple of 128 columns so the input-gradient GEMM (for the
  TransientEmbed rows) has a whole number of output tiles."""

  def __init__(self, num_transient, **kw):
    self.name = 'ImplicitMask_0'
    self.net_depth = 4
    self.net_width = 256
    self.deg_coord = 10
    self.weight_init = 'he_uniform'
    for k, v in kw.items():
      if not hasattr(self, k):
        raise ValueError(f'ImplicitMask has no attribute {k!r}')
      setattr(self, k, v)
    if self.net_width % 128 or self.net_depth < 1:
      raise NotImplementedError('ImplicitMask.net_width must be a multiple of 128')
    self.T = num_transient
    self.E = 2 + 4 * self.deg_coord
    self.fan_in = self.E + num_transient
    self.kpad = _round_up(self.fan_in, 128)
    L, k, kp = [], self.fan_in, self.kpad
    for i in range(self.net_depth):
      L.append(dict(fan_in=k, kpad=kp, fan_out=self.net_width, kind='trunk', concat=False, name=f'Dense_{i}'))
      k = kp = self.net_width
    L.append(dict(fan_in=k, kpad=k, fan_out=1, kind='maskhead', name=f'Dense_{self.net_depth}'))
    self.layers = L


class ParamLayout:
  """Flat fp32 parameter buffer: leaves in (module, layer, kernel|bias) order, kernels stored
  [fan_in_padded, fan_out] row-major (flax [in,out] + zero rows), chunk table for the optimizer."""

  def __init__(self, specs, num_embeddings, num_glo, num_transient=0):
    self.leaves = []   # dict(path, off, shape (logical), pshape (padded), module, leaf)
    off = 0
    self.modules = []
    for spec in specs:
      mid = len(self.modules)
      self.modules.append(spec.name)
      for l in spec.layers:
        for kind in ('kernel', 'bias'):
          shape = (l['fan_in'], l['fan_out']) if kind == 'kernel' else (l['fan_out'],)
          npad = l.get('npad', l['fan_out'])
          pshape = (l['kpad'], npad) if kind == 'kernel' else (npad,)
          n = int(np.prod(pshape))
          self.leaves.append(dict(path=(spec.name, l['name'], kind), off=off, shape=shape, pshape=pshape,
                                  module=mid, leaf=len(self.leaves), layer=l, spec=spec))
          off += _round_up(n, 4)
    if num_glo > 0:
      mid = len(self.modules)
      self.modules.append('GloEmbed_0')
      n = num_embeddings * num_glo
      self.leaves.append(dict(path=('GloEmbed_0', 'embedding'), off=off, shape=(num_embeddings, num_glo),
                              pshape=(num_embeddings, num_glo), module=mid, leaf=len(self.leaves), layer=None, spec=None))
      off += _round_up(n, 4)
    if num_transient > 0:
      mid = len(self.modules)
      self.modules.append('TransientEmbed_0')
      n = num_embeddings * num_transient
      self.leaves.append(dict(path=('TransientEmbed_0', 'embedding'), off=off, shape=(num_embeddings, num_transient),
                              pshape=(num_embeddings, num_transient), module=mid, leaf=len(self.leaves), layer=None,
                              spec=None))
      off += _round_up(n, 4)
    self.size = off
    chunks = []
    for lf in self.leaves:
      n = int(np.prod(lf['pshape']))
      for s in range(0, n, CHUNK):
        chunks.append((lf['off'] + s, min(CHUNK, n - s), lf['leaf'], lf['module']))
    self.chunks = np.array(chunks, dtype=np.int32)
    info = []
    for lf in self.leaves:
      idx = np.nonzero(self.chunks[:, 2] == lf['leaf'])[0]
      info.append((int(idx[0]), int(idx[-1]) + 1, lf['module'], 0))
    self.leaf_info = np.array(info, dtype=np.int32)
    self.by_path = {lf['path']: lf for lf in self.leaves}

  def view(self, flat, path, padded=False):
    lf = self.by_path[path]
    if not padded and len(lf['shape']) == 2 and lf['pshape'][1] != lf['shape'][1]:
      # zero-padded output columns: the logical kernel is a strided window of the stored one
      n = int(np.prod(lf['pshape']))
      return flat[lf['off']:lf['off'] + n].view(*lf['pshape'])[:lf['shape'][0], :lf['shape'][1]]
    shp = lf['pshape'] if padded else lf['shape']
    n = int(np.prod(shp))
    return flat[lf['off']:lf['off'] + n].view(*shp)

  def tree(self, flat):
    """flax-style nested dict of views: {'params': {'NerfMLP_0': {'Dense_0': {'kernel','bias'}}, ...}}."""
    out = {}
    for lf in self.leaves:
      d = out
      for k in lf['path'][:-1]:
        d = d.setdefault(k, {})
      d[lf['path'][-1]] = self.view(flat, lf['path'])
    return {'params': out}

  def num_params(self):
    return sum(int(np.prod(lf['shape'])) for lf in self.leaves)


class Workspace:
  """Named device buffers reused across steps (no allocation inside the timed loop)."""

  def __init__(self, device):
    self.device = device
    self.bufs = {}
    self._ptr_keys = []

  def put_table(self, key, value, keep=64):
    """Host-side pointer tables keyed by buffer ADDRESSES (('mlp_tail', tag, theta.data_ptr(), M), ...): a function of addresses only,
    read by the C entry point at call time and passed to the kernel by value -- safe to drop and rebuild.  Kept for the `keep` most
    recent keys so that Model.apply on a stream of clones does not grow the dict without bound."""
    self.bufs[key] = value
    self._ptr_keys.append(key)
    while len(self._ptr_keys) > keep:
      self.bufs.pop(self._ptr_keys.pop(0), None)
    return value

  def get(self, name, shape, dtype=torch.float32, zero=False):
    key = (name, tuple(shape), dtype)
    b = self.bufs.get(key)
    if b is None:
      b = torch.empty(shape, dtype=dtype, device=self.device)
      self.bufs[key] = b
    if zero:
      b.zero_()
    return b


class Engine:

  def __init__(self, model, device, compute_dtype='bf16'):
    if compute_dtype not in ('bf16', 'fp32'):
      raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
    if not torch.cuda.is_available():
      raise _lib.HugsError('no GPU visible: the hugs path has no CPU fallback')
    _lib.lib()
    self.model = model
    self.device = torch.device(device)
    self.dt = 1 if compute_dtype == 'bf16' else 0
    self.tdt = torch.bfloat16 if self.dt else torch.float32
    self.layout = model.layout
    self.ws = Workspace(self.device)
    self.wn, self.wt = {}, {}     # compute-dtype weight copies keyed by leaf path
    self._cast_src = None         # (TrainState.gen, data_ptr, torch version) the operand copies were last cast for, or None
    self.basis = {s.name: torch.from_numpy(s.basis).to(self.device) for s in model.specs}
    self.chunks = torch.from_numpy(self.layout.chunks).to(self.device)
    self.leaf_info = torch.from_numpy(self.layout.leaf_info).to(self.device)

  # ---- weights ------------------------------------------------------------------------------------
  def refresh_weights(self, theta, owner=None):
    """Cast the fp32 masters to the GEMM operand copies: Wn [Kp,N] (dX) and Wt [N,Kp] (forward).  owner: the
    TrainState whose buffer `theta` is (the train step's own refresh), None for any other caller."""
    self._cast_src = None if owner is None else (owner.gen, theta.data_ptr(), theta._version)
    self.weights_stale = False
    if not hasattr(self, '_cast_tables'):
      self._cast_tables = {}
    tab = self._cast_tables.get(theta.data_ptr())
    if tab is None:
      # one 40-byte record per GEMM operand pair (include/hugs.h hugs_cast_weights_batch): device ADDRESSES only, a pure
      # function of (buffer address, layout).  ONE table per master-buffer address, kept for the life of the engine: a captured
      # train step has the table's address baked into its hipGraph, so a refresh on another buffer (Model.apply on a clone or a
      # loaded checkpoint between steps) must neither free nor rewrite the table the graph reads.  The two eager
      # steps in front of a capture build the table of the state's buffer, so no H2D copy happens inside the capture.
      rec, blk = [], 0
      for lf in self.layout.leaves:
        if lf['path'][-1] != 'kernel' or lf['layer']['kind'] not in ('trunk', 'bottleneck', 'view', 'vtrunk', 'tview', 'ttrunk'):
          continue
        l = lf['layer']
        W = self.layout.view(theta, lf['path'], padded=True)
        K = l['kpad'] if l['kind'] not in ('view', 'tview') else lf['spec'].bottleneck_width   # GEMM part of the layer
        N = l.get('npad', l['fan_out'])
        key = lf['path']
        if key not in self.wt:
          self.wt[key] = torch.empty(N, K, dtype=self.tdt, device=self.device)
          self.wn[key] = torch.empty(K, N, dtype=self.tdt, device=self.device) if self.dt else None
        if not self.dt:
          self.wn[key] = W[:K]      # fp32: the master itself is the natural-layout operand
        nbx = (N + 31) // 32
        rec.append((W.data_ptr(), self.wn[key].data_ptr() if self.dt else 0, self.wt[key].data_ptr(), K, N, blk, nbx))
        blk += ((K + 31) // 32) * nbx
      raw = np.zeros((len(rec), 5), np.int64)
      for i, (pw, pn, pt, K, N, b0, nbx) in enumerate(rec):
        raw[i, :3] = (pw, pn, pt)
        raw[i, 3:].view(np.int32)[:4] = (K, N, b0, nbx)
      tab = (theta.data_ptr(), torch.from_numpy(raw).to(self.device), len(rec), blk)
      self._cast_tables[theta.data_ptr()] = tab
      # bounded: tables of train-state buffers (`owner`: their address may be baked into a captured step) stay for the life
      # of the engine, those of other buffers (Model.apply on clones / loaded checkpoints) are kept for the 32 most recent addresses
      if owner is None:
        self._cast_lru = getattr(self, '_cast_lru', [])
        self._cast_lru.append(theta.data_ptr())
        while len(self._cast_lru) > 32:
          old = self._cast_lru.pop(0)
          if old not in getattr(self, '_cast_pinned', ()) and old != theta.data_ptr():
            self._cast_tables.pop(old, None)
    if owner is not None:
      self._cast_pinned = getattr(self, '_cast_pinned', set())
      self._cast_pinned.add(theta.data_ptr())
    _lib.call('hugs_cast_weights_batch', self.dt, tab[2], tab[1], tab[3])
    if _HEAD_FOLD:
      # P [Wp, H] = W_b [Wp, Bw] W_v[:Bw] [Bw, H]: the trunk's output gradient is (G_view W_v[:Bw]^T) W_b^T = G_view P^T, one
      # K = H GEMM straight from the view layer's gradient instead of dBott (K = H) followed by a K = Bw one -- 34 instead of 77
      # GFLOP on the critical chain of the backward pass at cfg2 (dBott itself is still needed, for the bottleneck's weight
      # gradient: it moves to the head stream).  67 MFLOP per refresh.
      for spec in self.model.specs:
        if spec.disable_rgb or not spec.use_viewdirs:
          continue
        lb, lv = spec.layers[spec.net_depth + 1], spec.layers[spec.net_depth + 2]
        kb, kv = (spec.name, lb['name'], 'kernel'), (spec.name, lv['name'], 'kernel')
        Bw, H = spec.bottleneck_width, spec.net_width_viewdirs
        Wp = self.wn[kb].shape[0]
        if not hasattr(self, 'wfold'):
          self.wfold = {}
        if spec.name not in self.wfold:
          self.wfold[spec.name] = torch.empty(Wp, H, dtype=self.tdt, device=self.device)
        _lib.call('hugs_gemm_nt', self.dt, Wp, H, Bw, 0, self.wn[kb], Bw, None, 0, self.wt[kv], Bw, None, None, 1, 0, 0, None, 0, None,
                  None, self.wfold[spec.name], H)
    for spec in self.model.specs:
      if spec.num_tra > 0:      # dBottleneck = [G_view | G_transient0] [Wv[:Bw] | Wt0[:Bw]]^T in one two-segment GEMM
        lv, lt = spec.layers[spec.net_depth + 2], spec.layers[spec.t0]
        self.wcat = torch.cat([self.wn[(spec.name, lv['name'], 'kernel')], self.wn[(spec.name, lt['name'], 'kernel')]],
                              1).contiguous()

  def weights_current(self, state):
    """True when the operand copies were cast for this very TrainState (its process-unique generation number: a new
    state that inherits a freed buffer's address and version count is NOT current) and torch has not written to its
    buffer since (load_variables / restore_checkpoint bump the tensor version; Model.apply on any variables resets the
    record).  The step's own Adam kernel writes through the raw pointer: `optimizer_step` then sets `weights_stale` and the NEXT
    step (or the next `forward` / `backward_level` / `mask_forward` call) re-casts -- nothing refreshes at the end of a step."""
    return self._cast_src is not None and self._cast_src == (state.gen, state.flat.data_ptr(), state.flat._version)

  # ---- forward ------------------------------------------------------------------------------------
  def _mlp_forward(self, spec, theta, lvl, N, S, tdist, rays, glo, keep, tra=None, mlp_key=None, n_real=None, weights_ready=None):
    M = N * S
    Mr = M if n_real is None else n_real * S      # rows of real rays (Model.apply pads ragged batches at the END)
    lay, ws, dt = self.layout, self.ws, self.dt
    tag = f'{spec.name}/L{lvl}'
    X0 = ws.get(tag + '/X0', (M, spec.Fp), self.tdt)
    _lib.call('hugs_cast_ipe_fwd', N, S, tdist, rays['origins'], rays['directions'], rays['radii'], self.basis[spec.name],
              spec.nb, (0 if self.model.ray_shape == 'cone' else 1) | (4 if self.model.disable_integration else 0) | (spec.min_deg_point << 8),
              int(spec.warp_fn is not None), spec.max_deg_point - spec.min_deg_point,
              dt, spec.Fp, X0)
    if weights_ready is not None:      # the operand copies are cast on their own lane at the start of a train step
      wait_event(torch.cuda.current_stream(), weights_ready)
    acts = [X0]
    x = X0
    W = spec.Wp
    Ys = [ws.get(f'{tag}/Y{i}', (M, W), self.tdt) for i in range(spec.net_depth)]
    # Row chunks (OFF by default, HUGS_FWD_CHUNK_MB): a [M,W] activation of the NerfMLP (268 MB at cfg2) does not fit
    # the 256 MiB Infinity Cache; walking the trunk chunk by chunk keeps each chunk's activations cache-resident from
    # the layer that writes them to the layer that reads them.  Stand-alone the 8-layer chain gains 3-10 %
    # (scratch/chain_chunk.py); inside the train step it is a wash (8.68 vs 8.67 ms, same box), so it stays off.
    # 1-bit relu masks for the backward dX GEMMs (written by the forward epilogue in the 256x256 kernels' own lane
    # layout, hugs_gemm_nt_bits): bf16, whole 256-row / 256-column tiles only; otherwise the backward reads Y itself
    use_bits = bool(self.dt and keep and M % 256 == 0 and W % 256 == 0 and _USE_BITS)
    bits = [ws.get(f'{tag}/bits{i}', (M * W // 32,), torch.int32) if use_bits and spec.layers[i]['kpad'] >= 256 else None
            for i in range(spec.net_depth)]
    nchunk = 1
    while (M // nchunk) * W * (2 if self.dt else 4) > _CHUNK_BYTES and (M // (2 * nchunk)) % 256 == 0 and M // (2 * nchunk) >= 65536:
      nchunk *= 2
    mc = M // nchunk
    ld = spec.layers[spec.net_depth]
    wd = lay.view(theta, (spec.name, ld['name'], 'kernel'), padded=True).reshape(-1)
    # Round 4: layers 1.. of a 256-wide trunk + its density head as ONE launch with the activations LDS-resident
    # (hugs_mlp256_tail_fwd), where it is faster than the per-layer GEMMs: up to _MLP_FUSE_ROWS rows (stand-alone, 3 layers + head:
    # 24 vs 45 us at 16 384 rows -- the launch-bound regime of small per-GPU batches --, 81 vs 67 us at 65 536, 1.05 vs 0.85 ms at
    # 1 M rows, where the per-layer launches already run at the HBM rate and the fused kernel's phases do not overlap well enough)
    # Round 5: exactly three fused layers with their mask bits (the reference's PropMLP, depth 4) run the register-resident kernel
    # (k_mlp256_chain3_fwd: the 384 KB of weights loaded once per CU) at every size -- no row cap
    chain3 = _MLP_CHAIN3 and _MLP_FUSE_ROWS > 0 and spec.net_depth == 4 and all(b is not None for b in bits[1:])
    fuse_tail = (dt == 1 and nchunk == 1 and W == 256 and spec.net_width == 256 and spec.net_depth >= 2 and M % 256 == 0 and
                 spec.net_depth - 1 <= int(_lib.lib().cdll.hugs_mlp256_tail_max_layers()) and (M <= _MLP_FUSE_ROWS or chain3) and
                 not any(l['concat'] for l in spec.layers[1:spec.net_depth]))
    chained = False
    if _NT_CHAIN and dt == 1 and nchunk == 1 and not fuse_tail and spec.net_depth >= 2 and all(b is not None for b in bits):
      chained = self._trunk_chain(spec, theta, tag, M, W, X0, Ys, bits)
    for c in range(0 if chained else nchunk):
      rows = slice(c * mc, (c + 1) * mc)
      x = X0[rows]
      for i in range(1 if fuse_tail else spec.net_depth):
        l = spec.layers[i]
        path = (spec.name, l['name'], 'kernel')
        bias = lay.view(theta, (spec.name, l['name'], 'bias'), padded=True)
        Y = Ys[i][rows]
        bts = None if bits[i] is None else bits[i][c * (mc * W // 32):(c + 1) * (mc * W // 32)]
        if bts is not None:
          if l['concat']:
            _lib.call('hugs_gemm_nt_bits', dt, mc, W, W, spec.Fp, x, W, X0[rows], spec.Fp, self.wt[path], l['kpad'], bias, 1, None,
                      None, Y, W, bts, None)
          else:
            K = l['kpad']
            _lib.call('hugs_gemm_nt_bits', dt, mc, W, K, 0, x, K, None, 0, self.wt[path], K, bias, 1, None, None, Y, W, bts, None)
        elif l['concat']:
          _lib.call('hugs_gemm_nt', dt, mc, W, W, spec.Fp, x, W, X0[rows], spec.Fp, self.wt[path], l['kpad'], bias, None, 1, 0, 1,
                    None, 0, None, None, Y, W)
        else:
          K = l['kpad']
          _lib.call('hugs_gemm_nt', dt, mc, W, K, 0, x, K, None, 0, self.wt[path], K, bias, None, 1, 0, 1, None, 0, None,
                    None, Y, W)
        x = Y
    acts += Ys
    x = Ys[-1]
    raw = ws.get(tag + '/raw', (M,))
    density = ws.get(tag + '/density', (M,))
    bd = lay.view(theta, (spec.name, ld['name'], 'bias'))
    if fuse_tail:
      nl = spec.net_depth - 1
      key = ('mlp_tail', tag, theta.data_ptr(), M)
      tab = ws.bufs.get(key)
      if tab is None:      # host arrays of device pointers (a function of buffer addresses only)
        ptr = lambda ts: np.ascontiguousarray([0 if t is None else t.data_ptr() for t in ts], np.uint64)
        tab = ws.put_table(key, (ptr([self.wt[(spec.name, spec.layers[i]['name'], 'kernel')] for i in range(1, spec.net_depth)]),
                                 ptr([lay.view(theta, (spec.name, spec.layers[i]['name'], 'bias'), padded=True) for i in range(1, spec.net_depth)]),
                                 ptr(Ys[1:]), ptr(bits[1:])))
      _lib.call('hugs_mlp256_tail_fwd', dt, M, nl, Ys[0], tab[0].ctypes.data, tab[1].ctypes.data, tab[2].ctypes.data,
                tab[3].ctypes.data, wd, bd, spec.density_bias, raw, density)
    else:
      _lib.call('hugs_density_fwd', dt, M, W, x, W, wd, bd, spec.density_bias, raw, density)
    noise_key = None
    if mlp_key is not None and (spec.density_noise > 0 or spec.bottleneck_noise > 0):
      from . import random as hrandom
      density_key, noise_key = hrandom.split(mlp_key)          # models.py:435 density_key, rng = random_split(rng)
      if spec.density_noise > 0:                                  # models.py:458-460: raw_density += noise * normal(key, [N, S])
        # the reference draws [n_real, S] values: a padded batch must not draw for its padding (another size = another stream)
        _lib.call('hugs_noise_softplus', M, Mr, raw, hrandom.normal(density_key, (Mr,)), float(spec.density_noise),
                  float(spec.density_bias), density)      # raw += noise * normal ; density = softplus(raw + bias), as hugs_density_fwd
    out = dict(X0=X0, acts=acts, raw=raw, density=density, rgb=None, bits=bits if nchunk == 1 else [None] * len(bits))
    if not spec.disable_rgb and not spec.use_viewdirs:
      lr = spec.layers[spec.net_depth + 1]
      rgb = ws.get(tag + '/rgb', (M, 3))
      Wr, br = self._rgb_head(theta, spec, lr, tag + '/rgbhead', padded=True)
      _lib.call('hugs_rgb_fwd', dt, M, W, x, W, Wr, br, spec.rgb_padding, rgb)
      out.update(bott=None, hview=None, rgb=rgb)
    elif not spec.disable_rgb:
      lb, lv, lvx, lr = spec.head_layers()
      Bw, H = spec.bottleneck_width, spec.net_width_viewdirs
      bott = ws.get(tag + '/bott', (M, Bw), self.tdt)
      _lib.call('hugs_gemm_nt', dt, M, Bw, W, 0, x, W, None, 0, self.wt[(spec.name, lb['name'], 'kernel')], W,
                lay.view(theta, (spec.name, lb['name'], 'bias')), None, 1, 0, 0, None, 0, None, None, bott, Bw)
      if noise_key is not None and spec.bottleneck_noise > 0:   # models.py:478-481: bottleneck += noise * normal(key, [N, S, Bw])
        from . import random as hrandom
        kb, _ = hrandom.split(noise_key)
        _lib.call('hugs_axpy_op', dt, Mr * Bw, float(spec.bottleneck_noise), hrandom.normal(kb, (Mr, Bw)), bott)
      Wv = lay.view(theta, (spec.name, lv['name'], 'kernel'))
      pre = rays.get('_raybias')
      if rays.get('_ev_rays') is not None:      # (dir_enc / the embedding rows / the precomputed bias come from the weight-cast lane)
        wait_event(torch.cuda.current_stream(), rays['_ev_rays'])
      if pre is not None and pre[0] == spec.name and pre[1] == lvl and glo is rays.get('_glo'):
        rb = pre[2]      # (encode_rays: computed on the train step's weight-cast lane)
      else:
        rb = ws.get(tag + '/raybias', (N, H))
        _lib.call('hugs_raybias_fwd', N, H, spec.nd, spec.num_glo, rays['dir_enc'], glo, Wv[Bw:],
                  lay.view(theta, (spec.name, lv['name'], 'bias')), rb)
      hact = ws.get(tag + '/hview', (M, H), self.tdt)
      _lib.call('hugs_gemm_nt', dt, M, H, Bw, 0, bott, Bw, None, 0, self.wt[(spec.name, lv['name'], 'kernel')], Bw, None, rb,
                S, H, 1, None, 0, None, None, hact, H)
      hviews = [hact]
      for k_, lx in enumerate(lvx):      # net_depth_viewdirs > 1 (models.py:508-512): Dense(H) + relu on the view branch
        hn = ws.get(f'{tag}/hview{k_ + 1}', (M, H), self.tdt)
        _lib.call('hugs_gemm_nt', dt, M, H, H, 0, hviews[-1], H, None, 0, self.wt[(spec.name, lx['name'], 'kernel')], H,
                  lay.view(theta, (spec.name, lx['name'], 'bias')), None, 1, 0, 1, None, 0, None, None, hn, H)
        hviews.append(hn)
      hact = hviews[-1]
      rgb = ws.get(tag + '/rgb', (M, 3))
      Wr, br = self._rgb_head(theta, spec, lr, tag + '/rgbhead')
      _lib.call('hugs_rgb_fwd', dt, M, H, hact, H, Wr, br, spec.rgb_padding, rgb)
      out.update(bott=bott, hview=hact, hviews=hviews, rgb=rgb)
      if spec.num_tra > 0 and tra is not None:
        # models.py:521-539: x = [bottleneck | tra_vec] -> (Dense+relu) x depth_t -> density_t, rgb_t, uncertainty.
        # The tra_vec part of the first layer is constant along a ray: a per-ray bias, like the view layer's.
        Ht, dtn, t0 = spec.net_width_transient, spec.net_depth_transient, spec.t0
        lt = spec.layers[t0]
        Wt0 = lay.view(theta, (spec.name, lt['name'], 'kernel'))
        rbt = ws.get(tag + '/raybias_t', (N, Ht))
        _lib.call('hugs_raybias_fwd', N, Ht, 0, spec.num_tra, None, tra, Wt0[Bw:],
                  lay.view(theta, (spec.name, lt['name'], 'bias')), rbt)
        tacts = []
        x = ws.get(tag + '/T0', (M, Ht), self.tdt)
        _lib.call('hugs_gemm_nt', dt, M, Ht, Bw, 0, bott, Bw, None, 0, self.wt[(spec.name, lt['name'], 'kernel')], Bw, None,
                  rbt, S, Ht, 1, None, 0, None, None, x, Ht)
        tacts.append(x)
        for i in range(1, dtn):
          l = spec.layers[t0 + i]
          y = ws.get(f'{tag}/T{i}', (M, Ht), self.tdt)
          _lib.call('hugs_gemm_nt', dt, M, Ht, Ht, 0, x, Ht, None, 0, self.wt[(spec.name, l['name'], 'kernel')], Ht,
                    lay.view(theta, (spec.name, l['name'], 'bias')), None, 1, 0, 1, None, 0, None, None, y, Ht)
          tacts.append(y)
          x = y
        ld_, lr_, lu_ = spec.layers[t0 + dtn:t0 + dtn + 3]
        raw_t, dens_t = ws.get(tag + '/raw_t', (M,)), ws.get(tag + '/dens_t', (M,))
        _lib.call('hugs_density_fwd', dt, M, Ht, x, Ht, lay.view(theta, (spec.name, ld_['name'], 'kernel')).reshape(-1),
                  lay.view(theta, (spec.name, ld_['name'], 'bias')), spec.density_bias, raw_t, dens_t)
        rgb_t = ws.get(tag + '/rgb_t', (M, 3))
        Wrt, brt = self._rgb_head(theta, spec, lr_, 'tbwd/rgbhead_t')
        _lib.call('hugs_rgb_fwd', dt, M, Ht, x, Ht, Wrt, brt, spec.rgb_padding, rgb_t)
        raw_u, unc = ws.get(tag + '/raw_u', (M,)), ws.get(tag + '/unc', (M,))
        _lib.call('hugs_density_fwd', dt, M, Ht, x, Ht, lay.view(theta, (spec.name, lu_['name'], 'kernel')).reshape(-1),
                  lay.view(theta, (spec.name, lu_['name'], 'bias')), 0.0, raw_u, unc)      # softplus, no bias shift
        out.update(tacts=tacts, raw_t=raw_t, dens_t=dens_t, rgb_t=rgb_t, raw_u=raw_u, unc=unc, tra=tra)
    return out

  def encode_viewdirs(self, rays, N):
    """rays['dir_enc'] = pos_enc(viewdirs) (coord.py:136-147; models.py:494-497), once per ray batch."""
    if 'dir_enc' not in rays:
      mdl = self.model
      rays['dir_enc'] = self.ws.get('dir_enc', (N, mdl.nerf_spec.nd))
      _lib.call('hugs_dir_enc_fwd', N, mdl.nerf_spec.deg_view, rays['viewdirs'], rays['dir_enc'])

  def encode_rays(self, theta, rays, N):
    """Everything of a TRAIN step's forward pass that depends on the rays and the fp32 masters only, not on a sampling level: the
    view-direction encoding, the GLO / transient embedding rows of the rays' cameras and the NerfMLP view layer's per-ray bias
    (models.py:120-129, 494-505).  The train step runs it on the weight-cast lane -- the first MLP product waits for that lane, and
    nothing here is needed before it -- instead of in front of the first sampler launch (dir_enc, gathers) and between the bottleneck
    and view products (ray bias: 11 us of a 1.2 ms step at 128 rays).  forward() / _mlp_forward() use what they find in `rays`."""
    mdl, ws, lay = self.model, self.ws, self.layout
    self.encode_viewdirs(rays, N)
    if mdl.num_glo_features > 0:
      rays['_glo'] = ws.get('glo', (N, mdl.num_glo_features))
      _lib.call('hugs_glo_gather', N, mdl.num_glo_features, lay.view(theta, ('GloEmbed_0', 'embedding')), rays['embed_idx'], 0, rays['_glo'])
    spec = mdl.nerf_spec
    if spec.num_tra > 0:
      rays['_tra'] = ws.get('tra_vec', (N, spec.num_tra))
      _lib.call('hugs_glo_gather', N, spec.num_tra, lay.view(theta, ('TransientEmbed_0', 'embedding')), rays['embed_idx'], 0, rays['_tra'])
    if not spec.disable_rgb and spec.use_viewdirs:
      lb, lv, lvx, lr = spec.head_layers()
      Bw, H = spec.bottleneck_width, spec.net_width_viewdirs
      Wv = lay.view(theta, (spec.name, lv['name'], 'kernel'))
      rb = ws.get(f'{spec.name}/L{mdl.num_levels - 1}/raybias', (N, H))
      _lib.call('hugs_raybias_fwd', N, H, spec.nd, spec.num_glo, rays['dir_enc'], rays.get('_glo'), Wv[Bw:],
                lay.view(theta, (spec.name, lv['name'], 'bias')), rb)
      rays['_raybias'] = (spec.name, mdl.num_levels - 1, rb)

  def forward(self, theta, rays, train_frac, u01, compute_extras, zero_glo=False, zero_tra=False, n_real=None, anneal_dev=None,
              weights_ready=None):
    """Model.__call__ (models.py:74-330).  rays: dict of contiguous [N,c] cuda tensors (+ 'dir_enc').
    u01: None, a list[num_levels] of U[0,1) draws, or a stepfun.Jitter list of scaled draws.  Returns per-level dicts (device tensors; buffers are
    reused by the next call)."""
    mdl = self.model
    N = rays['origins'].shape[0]
    ws = self.ws
    if getattr(self, 'weights_stale', False) and weights_ready is None:
      # an optimizer step has moved the masters since the operand copies were cast (the train step re-casts at the START of the
      # next step): a caller that drives the engine directly gets fresh copies here
      self.refresh_weights(theta)
    glo = None
    if mdl.num_glo_features > 0:
      glo = rays.get('_glo') if not zero_glo else None       # (encode_rays: gathered on the train step's weight-cast lane)
      if glo is None:
        glo = ws.get('glo', (N, mdl.num_glo_features))
        emb = self.layout.view(theta, ('GloEmbed_0', 'embedding'))
        _lib.call('hugs_glo_gather', N, mdl.num_glo_features, emb, rays['embed_idx'], int(zero_glo), glo)
    tra = None
    if mdl.nerf_spec.num_tra > 0:          # NeRF-W: TransientEmbed rows of the rays' cameras (models.py:120-129)
      tra = rays.get('_tra') if not zero_tra else None
      if tra is None:
        tra = ws.get('tra_vec', (N, mdl.nerf_spec.num_tra))
        _lib.call('hugs_glo_gather', N, mdl.nerf_spec.num_tra, self.layout.view(theta, ('TransientEmbed_0', 'embedding')),
                  rays['embed_idx'], int(zero_tra), tra)
    self.encode_viewdirs(rays, N)
    if mdl.near_anneal_rate is None:
      init_s_near = 0.
    else:
      init_s_near = float(np.clip(1 - train_frac / mdl.near_anneal_rate, 0, mdl.near_anneal_init))
    init_s_far = 1.
    # level-0 input histogram: one bin [init_s_near, init_s_far] of weight 1 per ray -- constant between steps unless
    # near_anneal moves it: written when the values change (three launches less per step)
    sdist = ws.get('sdist_init', (N, 2))
    weights = ws.get('w_init', (N, 1))
    if ws.bufs.get(('init_hist', N)) != (init_s_near, init_s_far):
      sdist[:, 0] = init_s_near
      sdist[:, 1] = init_s_far
      weights.fill_(1.0)
      ws.bufs[('init_hist', N)] = (init_s_near, init_s_far)
    prod = 1
    levels = []
    for lvl in range(mdl.num_levels):
      is_prop = lvl < mdl.num_levels - 1
      S = mdl.num_prop_samples if is_prop else mdl.num_nerf_samples
      dilation = mdl.dilation_bias + mdl.dilation_multiplier * (init_s_far - init_s_near) / prod
      prod *= S
      use_dilation = mdl.dilation_bias > 0 or mdl.dilation_multiplier > 0
      if anneal_dev is not None:      # captured step: the value anneal_factor() gives for this step, in device memory
        anneal = anneal_dev
      else:
        anneal = self.anneal_factor(train_frac)
      draw = None if u01 is None else u01[lvl]
      scaled = getattr(u01, 'scaled', False)     # stepfun.Jitter: draws already in [0, max_jitter)
      sd, td = stepfun.level_sample(sdist, weights, lvl > 0 and use_dilation, dilation, (init_s_near, init_s_far), anneal,
                                    mdl.resample_padding, S, None if scaled else draw, mdl.raydist, rays['near'],
                                    rays['far'], jitter=draw if scaled else None)
      spec = mdl.prop_spec if is_prop else mdl.nerf_spec
      mkeys = getattr(u01, 'mlp_keys', None)
      out = self._mlp_forward(spec, theta, lvl, N, S, td, rays, glo, True, None if is_prop else tra,
                              mlp_key=None if mkeys is None else mkeys[lvl], n_real=n_real, weights_ready=weights_ready if lvl == 0 else None)
      w = ws.get(f'L{lvl}/weights', (N, S))
      rgb_all = ws.get('rgb_out_all', (mdl.num_levels, N, 3))
      rgb_out = rgb_all[lvl]
      extras = ws.get(f'L{lvl}/extras', (N, 5)) if compute_extras else None
      bgs = getattr(u01, 'bg_rgbs', None)
      bg_rgb = bgs[lvl] if (mdl.bg_random and bgs) else None
      _lib.call('hugs_composite_fwd', N, S, out['density'], out['rgb'], td, rays['directions'],
                int(mdl.opaque_background), 0.0 if bg_rgb is not None else mdl.bg_intensity, rays['far'].reshape(-1), w, rgb_out, extras)
      if bg_rgb is not None:
        # a per-ray, per-channel background (models.py:256-261): the kernel composites against black and the background term
        # bg_w * bg with bg_w = max(0, 1 - sum w) (render.py:219-221) is added here; its gradient reaches the weights through
        # d_w_extra in backward_level
        if out.get('dens_t') is not None:
          raise NotImplementedError('a random background together with the NeRF-W transient branch')
        bgw = ws.get(f'L{lvl}/bgw', (N,))
        # (proposal levels too: their colours are zero, the background term is not)
        _lib.call('hugs_bg_blend_fwd', N, S, w, bg_rgb.contiguous(), rgb_out, bgw)
        out.update(bg_rgb=bg_rgb, bgw=bgw)
      if out.get('dens_t') is not None:     # models.py:285-307
        nw = {k: ws.get(f'L{lvl}/{k}', (N, 3)) for k in ('rgb_combined', 'rgb_static', 'rgb_transient')}
        nw['uncertainty'] = ws.get(f'L{lvl}/uncertainty', (N,))
        _lib.call('hugs_dual_composite_fwd', N, S, out['density'], out['dens_t'], out['rgb'], out['rgb_t'], out['unc'], td,
                  rays['directions'], int(mdl.opaque_background), mdl.bg_intensity, mdl.beta_min, nw['rgb_combined'],
                  nw['rgb_static'], nw['rgb_transient'], nw['uncertainty'])
        out.update(nw)
      out.update(sdist=sd, tdist=td, weights=w, rgb_out=rgb_out, rgb_all=rgb_all, extras=extras, S=S, spec=spec, glo=glo)
      levels.append(out)
      sdist, weights = sd, w
    return levels

  def _trunk_chain(self, spec, theta, tag, M, W, X0, Ys, bits):
    """All trunk layers of `spec` in one launch (hugs_gemm_nt_chain; csrc/hugs_gemm_chain.inc).  False when the shape does not
    qualify (the caller then launches layer by layer): whole 256 x 256 tiles, a whole number >= 4 of tiles per CU, whole row bands
    per XCD round, leading dimensions in multiples of 512, the layers' biases within 32 KiB of LDS."""
    depth = spec.net_depth
    ncu = torch.cuda.get_device_properties(self.device).multi_processor_count & ~7
    ntiles = (M // 256) * (W // 256)
    lds_ = [spec.Fp, W] + [spec.layers[i]['kpad'] for i in range(depth)]
    if (M % 256 or W % 256 or depth > 8 or depth * W * 4 > 32768 or ncu < 8 or ntiles % ncu or ntiles // ncu < 4 or
        (ntiles // 8) % (W // 256) or (ncu // 8) % (W // 256) or any(v % 512 for v in lds_)):
      return False
    key = ('nt_chain', tag, theta.data_ptr(), M)
    ent = self.ws.bufs.get(key)
    if ent is None:
      tab = np.zeros((depth, 12), np.uint64)
      for i in range(depth):
        l = spec.layers[i]
        a1 = X0 if i == 0 else Ys[i - 1]
        k1 = spec.Fp if i == 0 else W
        tab[i] = [a1.data_ptr(), X0.data_ptr(), self.wt[(spec.name, l['name'], 'kernel')].data_ptr(),
                  self.layout.view(theta, (spec.name, l['name'], 'bias'), padded=True).data_ptr(), Ys[i].data_ptr(), bits[i].data_ptr(),
                  k1, spec.Fp, l['kpad'], k1, spec.Fp if l['concat'] else 0, 0]
        assert l['kpad'] == k1 + (spec.Fp if l['concat'] else 0)
      ent = self.ws.bufs[key] = (tab, self.ws.get(tag + '/chain_flags', (depth * (M // 256),), torch.int32))
    try:
      _lib.call('hugs_gemm_nt_chain', self.dt, M, W, depth, ent[0].ctypes.data, ent[1])
    except _lib.HugsError as e:      # rc -3: the library's own qualification rules refused a shape the check above let through
      if getattr(e, 'rc', 0) != -3:  # (nothing was launched: the caller falls back to the per-layer launches)
        raise
      return False
    return True

  def anneal_factor(self, train_frac):
    """models.py:185-190: the annealing exponent of the resampling logits at this point of training."""
    mdl = self.model
    if mdl.anneal_slope > 0:
      return (mdl.anneal_slope * train_frac) / ((mdl.anneal_slope - 1) * train_frac + 1)
    return 1.

  # ---- backward -----------------------------------------------------------------------------------
  def _tn(self, M, Kc, Nn, X, ldx, G, ldg, dW, db):
    """dW[Kc,Nn] = X^T G (+ db = colsum G): split the M reduction so the grid fills the chip once."""
    if self.dt and Kc % 256 == 0 and Nn % 256 == 0 and (Kc // 256) * (Nn // 256) >= 4:
      tiles, step, target = (Kc // 256) * (Nn // 256), 64, 256      # 256x256 tiles, one workgroup per CU
    else:
      tiles, step, target = (Kc // 128) * (Nn // 128), (64 if self.dt else 16), 768
    units = M // step
    ns = max(1, min(units, (target + tiles - 1) // tiles))
    while units % ns:
      ns -= 1
    nbytes = _lib.lib().cdll.hugs_gemm_tn_ws_bytes(Kc, Nn, ns)
    # one slab workspace per stream (calls on a stream are ordered)
    slab = self.ws.get(f'tn_slab/{torch.cuda.current_stream().cuda_stream}', (max(nbytes // 4, 1),))
    _lib.call('hugs_gemm_tn', self.dt, M, Kc, Nn, ns, X, ldx, G, ldg, dW, db, slab)

  def _tn_batch(self, items):
    """hugs_gemm_tn_batch: items = [(Mrows, Kc, N, X, ldx, G, ldg, dW, dbias or None)], one launch + one reduce."""
    key = tuple((it[0], it[1], it[2], it[3].data_ptr(), it[4], it[5].data_ptr(), it[6], it[7].data_ptr(),
                 0 if it[8] is None else it[8].data_ptr()) for it in items)
    cache = self.ws.bufs.setdefault('tn_batch_tables', {})
    ent = cache.get(key)
    if ent is None:
      arr = np.zeros(len(items), _TN_ITEM)
      for k, (Mr, Kc, Nn, X, ldx, G, ldg, dW, db) in enumerate(key):
        arr[k] = (X, G, dW, db, ldx, ldg, Mr, Kc, Nn, 0)
      ns = int(_lib.lib().cdll.hugs_gemm_tn_batch_nsplit(len(items), __import__('ctypes').c_void_p(arr.ctypes.data)))
      nbytes = int(_lib.lib().cdll.hugs_gemm_tn_batch_ws_bytes(len(items), arr.ctypes.data, ns))
      ent = cache[key] = (arr, ns, nbytes)
    arr, ns, nbytes = ent
    slab = self.ws.get(f'tn_slab/{torch.cuda.current_stream().cuda_stream}', (max(nbytes // 4, 1),))
    if _lib.PROFILE is not None:      # bench.py's in-step timing: (kind, total flops, items, pieces, tag)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      _lib.call('hugs_gemm_tn_batch', self.dt, len(items), arr.ctypes.data, ns, slab)
      e1.record()
      _lib.PROFILE.append(('hugs_gemm_tn_batch', ('tnb', sum(2.0 * it[0] * it[1] * it[2] for it in items), len(items), ns,
                                                  f'items{len(items)}_split{ns}'), e0, e1))
      return
    _lib.call('hugs_gemm_tn_batch', self.dt, len(items), arr.ctypes.data, ns, slab)

  def _rgb_head(self, theta, spec, layer, tag, padded=False):
    """(W, b) the rgb head kernels see: rgb = sigmoid(premultiplier (h W + b) + rgb_bias) (models.py:514-516, :534-536) is
    sigmoid(h (p W) + (p b + rgb_bias)); the kernels' weight / bias gradients are then those of (p W, p b + r): times p."""
    W = self.layout.view(theta, (spec.name, layer['name'], 'kernel'), padded)      # (padded: [Wp, 3] for a head on the padded trunk)
    b = self.layout.view(theta, (spec.name, layer['name'], 'bias'))
    p_, r_ = float(spec.rgb_premultiplier), float(spec.rgb_bias)
    if p_ == 1. and r_ == 0.:
      return W, b
    We, be = self.ws.get(tag + '/W_eff', tuple(W.shape)), self.ws.get(tag + '/b_eff', (4,))
    _lib.call('hugs_affine', W.numel(), W, p_, 0.0, We)
    _lib.call('hugs_affine', 3, b, p_, r_, be)
    return We, be

  def backward_level(self, theta, grad, lv, rays, N, d_rgb_out, d_w_extra, nerfw=None, leaf_done=None, lane=0, after_heads=None,
                     before_dw=None):
    """Backward of one level: compositing -> heads -> trunk.  Writes (=, not +=) the level's MLP gradients
    into `grad` (flat, same layout as theta); GLO embedding rows are scatter-added (caller zeroes them).
    leaf_done(lo, hi): optional callback, called on the stream that produced them as soon as the gradient
    range grad[lo:hi] is final (the heads once, then one trunk layer at a time): the data-parallel step starts that
    bucket's all-reduce there, underneath the rest of the backward pass."""
    spec, S, lay, ws, dt = lv['spec'], lv['S'], self.layout, self.ws, self.dt
    self._require_fresh_weights()
    M = N * S
    tag = f'{spec.name}/bwd'
    W = spec.Wp
    gview = lambda p, padded=False: lay.view(grad, p, padded)
    d_density = ws.get(tag + '/d_density', (M,))
    d_rgb_s = ws.get(tag + '/d_rgb_s', (M, 3)) if lv['rgb'] is not None else None
    bg_int = self.model.bg_intensity
    if lv.get('bg_rgb') is not None:
      bg_int = 0.0
      if d_rgb_out is not None:
        # d/dw_s of bg_w * bg = -(bg . d_rgb_out) where 1 - sum w > 0, the same for every sample of the ray
        dwt = ws.get(tag + '/d_w_total', (N, S))
        _lib.call('hugs_bg_blend_bwd', N, S, d_rgb_out, lv['bg_rgb'].contiguous(), lv['bgw'], d_w_extra, dwt)
        d_w_extra = dwt
    d_raw = ws.get(tag + '/d_raw', (M,))
    # (round 5: without the NeRF-W branch -- whose compositing backward ADDS into d_density afterwards -- the density head's
    #  pre-activation gradient d_raw leaves in the same pass: one launch less between the losses and G_last)
    raw_fused = nerfw is None and _COMPOSITE_RAW
    if raw_fused:
      _lib.call('hugs_composite_bwd_raw', N, S, lv['density'], lv['rgb'], lv['tdist'], rays['directions'],
                int(self.model.opaque_background), bg_int, d_rgb_out, d_w_extra, d_density, d_rgb_s, lv['raw'], spec.density_bias, d_raw)
    else:
      _lib.call('hugs_composite_bwd', N, S, lv['density'], lv['rgb'], lv['tdist'], rays['directions'],
                int(self.model.opaque_background), bg_int, d_rgb_out, d_w_extra, d_density, d_rgb_s)
    if nerfw is not None:
      # the loss saw rgb_combined and beta: their gradients reach sigma_s (added), c_s, sigma_t, c_t, u
      d_dt, d_ct, d_u = ws.get(tag + '/d_dens_t', (M,)), ws.get(tag + '/d_rgb_t', (M, 3)), ws.get(tag + '/d_unc', (M,))
      _lib.call('hugs_dual_composite_bwd', N, S, lv['density'], lv['dens_t'], lv['rgb'], lv['rgb_t'], lv['unc'], lv['tdist'],
                rays['directions'], int(self.model.opaque_background), self.model.bg_intensity, nerfw['d_rgb_combined'],
                nerfw['d_beta'], nerfw['dens_t_const'], d_density, d_rgb_s, d_dt, d_ct, d_u)
    acts = lv['acts']
    Ylast = acts[-1]
    ld = spec.layers[spec.net_depth]
    dws = ws.get(tag + '/dens_ws', (max(_lib.lib().cdll.hugs_density_bwd_ws_bytes(W) // 4, 1),))
    if spec.disable_rgb or not spec.use_viewdirs:
      _lib.call('hugs_density_bwd', dt, M, W, Ylast, W, None if raw_fused else d_density, lv['raw'], spec.density_bias, d_raw,
                gview((spec.name, ld['name'], 'kernel'), True).reshape(-1), gview((spec.name, ld['name'], 'bias')), dws)
    elif not raw_fused:
      # only d_raw here: the density head's weight gradient (a pass over the whole [M, W] activation) goes to the head
      # weight-gradient stream below instead of sitting in front of rgb_bwd -> dBott -> G_last
      _lib.call('hugs_density_bwd', dt, M, W, Ylast, W, d_density, lv['raw'], spec.density_bias, d_raw, None, None, None)
    wd = lay.view(theta, (spec.name, ld['name'], 'kernel'), padded=True).reshape(-1)
    Ga = ws.get(tag + '/Ga', (M, W), self.tdt)
    Gb = ws.get(tag + '/Gb', (M, W), self.tdt)
    # Round 5: the proposal MLP's dX chain (rank-1 head gradient + three masked dX GEMMs) as ONE launch with register-resident
    # weights (hugs_mlp256_tail_bwd) when the forward wrote all four layers' mask bits
    bits_all = lv.get('bits') or []
    fused_bwd = (_MLP_CHAIN3 and _MLP_FUSE_ROWS > 0 and dt == 1 and spec.disable_rgb and W == 256 and spec.net_width == 256 and
                 spec.net_depth == 4 and M % 256 == 0 and len(bits_all) == 4 and all(b is not None for b in bits_all) and
                 not any(l['concat'] for l in spec.layers[:4]) and _TN_BATCH > 0 and spec.Fp % 256 == 0 and M >= 2048)
    if fused_bwd:
      pass          # (G3 .. G0 are written by the fused launch in the trunk section below)
    elif spec.disable_rgb:
      _lib.call('hugs_rank1_mask', dt, M, W, d_raw, wd, Ylast, W, Ga, W)
    elif not spec.use_viewdirs:
      # G_last = (d_rgb W_rgb^T + d_raw (x) w_d) * (Ylast > 0): the rgb head's own backward (its G is already masked) + the
      # density head's masked rank-1 term
      lr = spec.layers[spec.net_depth + 1]
      rws = ws.get(tag + '/rgb_ws', (max(_lib.lib().cdll.hugs_rgb_bwd_ws_bytes() // 4, 1),))
      Wr, _ = self._rgb_head(theta, spec, lr, tag + '/rgbhead', padded=True)
      _lib.call('hugs_rgb_bwd', dt, M, W, Ylast, W, Wr, lv['rgb'], d_rgb_s, spec.rgb_padding, Ga, W,
                gview((spec.name, lr['name'], 'kernel'), True), gview((spec.name, lr['name'], 'bias')), rws)
      if spec.rgb_premultiplier != 1.:
        for g_ in (gview((spec.name, lr['name'], 'kernel'), True), gview((spec.name, lr['name'], 'bias'))):
          _lib.call('hugs_affine', g_.numel(), g_, float(spec.rgb_premultiplier), 0.0, g_)
      _lib.call('hugs_rank1_mask', dt, M, W, d_raw, wd, Ylast, W, Gb, W)
      _lib.call('hugs_add_op', dt, M * W, Gb, Ga)
    else:
      lb, lvw, lvx, lr = spec.head_layers()
      Bw, H = spec.bottleneck_width, spec.net_width_viewdirs
      Gv = ws.get(tag + '/Gview', (M, H), self.tdt)
      rws = ws.get(tag + '/rgb_ws', (max(_lib.lib().cdll.hugs_rgb_bwd_ws_bytes() // 4, 1),))
      Wr, _ = self._rgb_head(theta, spec, lr, tag + '/rgbhead')
      # (round 5: only G is on the way to the trunk; the head's 387-column weight-gradient reduction -- 5 us alone, 50+ next to the
      #  proposal level's persistent backward kernel -- goes to the head weight-gradient stream with the other head gradients)
      # (the final level only: the proposal levels share one workspace per MLP, and their next level's rgb_bwd would overwrite the
      #  partial sums before a reduce on another stream has read them)
      defer_rgb = _RGB_REDUCE_SIDE and H <= 256 and not spec.is_prop
      def rgb_dw():
        if defer_rgb:
          _lib.call('hugs_rgb_bwd_reduce', M, H, gview((spec.name, lr['name'], 'kernel')), gview((spec.name, lr['name'], 'bias')), rws)
        if spec.rgb_premultiplier != 1.:
          for g_ in (gview((spec.name, lr['name'], 'kernel')), gview((spec.name, lr['name'], 'bias'))):
            _lib.call('hugs_affine', g_.numel(), g_, float(spec.rgb_premultiplier), 0.0, g_)
      _lib.call('hugs_rgb_bwd', dt, M, H, lv['hview'], H, Wr, lv['rgb'],
                d_rgb_s, spec.rgb_padding, Gv, H, None if defer_rgb else gview((spec.name, lr['name'], 'kernel')),
                gview((spec.name, lr['name'], 'bias')), rws)
      if not defer_rgb:
        rgb_dw()
      # net_depth_viewdirs > 1: back through the further view layers, last to first -- dW_i = h_{i-1}^T G_i, db_i = colsum G_i,
      # G_{i-1} = (G_i W_i^T) * (h_{i-1} > 0); Gv ends up as the gradient at the FIRST view layer's pre-activation, as below expects
      hvs = lv.get('hviews') or [lv['hview']]
      for k_ in range(len(lvx) - 1, -1, -1):
        lx = lvx[k_]
        px = (spec.name, lx['name'], 'kernel')
        self._tn(M, H, H, hvs[k_], H, Gv, H, gview(px), gview((spec.name, lx['name'], 'bias')))
        Gp = ws.get(f'{tag}/Gview_{k_ & 1}', (M, H), self.tdt)
        _lib.call('hugs_gemm_nt', dt, M, H, H, 0, Gv, H, None, 0, self.wn[px], H, None, None, 1, 0, 0, hvs[k_], H, None, None, Gp, H)
        Gv = Gp
      gWv = gview((spec.name, lvw['name'], 'kernel'))
      Wv = lay.view(theta, (spec.name, lvw['name'], 'kernel'))
      d_rb = ws.get(tag + '/d_rb', (int(_lib.lib().cdll.hugs_raybias_bwd_ws_rows(N, spec.nd, spec.num_glo)), H))
      demb = gview(('GloEmbed_0', 'embedding')) if spec.num_glo > 0 else None
      # Only Gv -> dBott -> G_last is on the way to the trunk backward.  The head weight gradients (view-layer ray-bias
      # part + GLO rows, view dW, bottleneck dW: ~0.3 ms of reductions and skinny GEMMs) go to their own stream and run
      # underneath the trunk GEMMs.
      cur = torch.cuda.current_stream()
      hl = self._side_stream(lane + 3)

      def head_dw_1():
        if defer_rgb:
          rgb_dw()
        _lib.call('hugs_density_bwd', dt, M, W, Ylast, W, None, lv['raw'], spec.density_bias, d_raw,
                  gview((spec.name, ld['name'], 'kernel'), True).reshape(-1), gview((spec.name, ld['name'], 'bias')), dws)
        _lib.call('hugs_raybias_bwd', dt, N, S, H, spec.nd, spec.num_glo, Gv, H, rays['dir_enc'], lv['glo'], Wv[Bw:],
                  rays.get('embed_idx'), d_rb, gWv[Bw:], demb)
        # dWv[:Bw] = bott^T Gv ; db_v = colsum(Gv)
        self._tn(M, Bw, H, lv['bott'], Bw, Gv, H, gWv[:Bw], gview((spec.name, lvw['name'], 'bias')))
      if not _SIDE_LATE:
        ev_gv = new_event(); ev_gv.record(cur)
        with torch.cuda.stream(hl):
          wait_event(hl, ev_gv)
          head_dw_1()
      dB = ws.get(tag + '/dBott', (M, Bw), self.tdt)
      fold = _HEAD_FOLD and nerfw is None and hasattr(self, 'wfold') and spec.name in self.wfold

      def dbott():      # dBott = Gv Wv[:Bw]^T
        _lib.call('hugs_gemm_nt', dt, M, Bw, H, 0, Gv, H, None, 0, self.wn[(spec.name, lvw['name'], 'kernel')], H, None,
                  None, 1, 0, 0, None, 0, None, None, dB, Bw)
      if fold:
        pass            # (computed on the head stream, in front of the bottleneck's weight gradient: head_dw_2)
      elif nerfw is not None:
        G0t = self._transient_backward(theta, grad, lv, rays, N, d_dt, d_ct, d_u, side=hl)
        Ht = spec.net_width_transient
        # dBott = Gv Wv[:Bw]^T + G0t Wt0[:Bw]^T: one GEMM over the two K segments
        _lib.call('hugs_gemm_nt', dt, M, Bw, H, Ht, Gv, H, G0t, Ht, self.wcat, H + Ht, None, None, 1, 0, 0, None, 0, None,
                  None, dB, Bw)
      else:
        dbott()
      def head_dw_2():
        if fold:
          dbott()
        self._tn(M, W, Bw, Ylast, W, dB, Bw, gview((spec.name, lb['name'], 'kernel'), True), gview((spec.name, lb['name'], 'bias')))
        if leaf_done is not None:      # density / bottleneck / view / rgb (/ transient) layers: everything behind the trunk
          first = lay.by_path[(spec.name, spec.layers[spec.net_depth]['name'], 'kernel')]
          last = lay.by_path[(spec.name, spec.layers[-1]['name'], 'bias')]
          leaf_done(first['off'], last['off'] + int(np.prod(last['pshape'])))
      if not _SIDE_LATE:
        ev_db = new_event(); ev_db.record(cur)
        with torch.cuda.stream(hl):
          wait_event(hl, ev_db)
          head_dw_2()
          heads_done = new_event(); heads_done.record(hl)
      # G_last = (dBott Wb^T + d_raw (x) w_d) * (Ylast > 0)
      blast = lv['bits'][spec.net_depth - 1] if lv.get('bits') else None
      if fold:      # G_last = (Gv P^T + d_raw (x) w_d) * (Ylast > 0)
        P_ = self.wfold[spec.name]
        if blast is not None and H >= 128:
          _lib.call('hugs_gemm_nt_bits', dt, M, W, H, 0, Gv, H, None, 0, P_, H, None, 0, d_raw, wd, Ga, W, None, blast)
        else:
          _lib.call('hugs_gemm_nt', dt, M, W, H, 0, Gv, H, None, 0, P_, H, None, None, 1, 0, 0, Ylast, W, d_raw, wd, Ga, W)
      elif blast is not None and Bw >= 256:
        _lib.call('hugs_gemm_nt_bits', dt, M, W, Bw, 0, dB, Bw, None, 0, self.wn[(spec.name, lb['name'], 'kernel')], Bw, None, 0,
                  d_raw, wd, Ga, W, None, blast)
      else:
        _lib.call('hugs_gemm_nt', dt, M, W, Bw, 0, dB, Bw, None, 0, self.wn[(spec.name, lb['name'], 'kernel')], Bw, None, None,
                  1, 0, 0, Ylast, W, d_raw, wd, Ga, W)
      if _SIDE_LATE:      # (A/B: the head weight gradients and whatever `after_heads` launches start behind G_last)
        ev_gl = new_event(); ev_gl.record(cur)
        with torch.cuda.stream(hl):
          wait_event(hl, ev_gl)
          head_dw_1(); head_dw_2()
          heads_done = new_event(); heads_done.record(hl)
    if after_heads is not None:
      after_heads()
    if (spec.disable_rgb or not spec.use_viewdirs) and leaf_done is not None:
      first = lay.by_path[(spec.name, spec.layers[spec.net_depth]['name'], 'kernel')]
      last = lay.by_path[(spec.name, spec.layers[-1]['name'], 'bias')]
      leaf_done(first['off'], last['off'] + int(np.prod(last['pshape'])))
    G = Ga
    X0 = lv['X0']
    # Trunk backward on two HIP streams: the weight-gradient GEMM of layer i (side stream) and the dX GEMM that
    # produces G_{i-1} (main stream) only share the read of G_i, so they run concurrently and fill each other's
    # tile-epilogue / launch-boundary bubbles.  Four G buffers rotate so dX never overwrites a buffer a pending
    # dW still reads AND never has to wait for the dW (+ its slab reduction) of the layer just above: with three, every
    # layer boundary was a ~40 us bubble (dX_{i-1} waited for dW_i to release the buffer it writes).
    main = torch.cuda.current_stream()
    side = self._side_stream(lane)
    depth = spec.net_depth
    trunk = spec.layers[:depth]
    nitem = depth + sum(1 for l in trunk if l['concat'])
    if (_TN_BATCH > 0 and dt and W % 256 == 0 and spec.Fp % 256 == 0 and M >= 2048 and M % 64 == 0 and nitem <= 16 and
        all(l['concat'] or l['kpad'] % 256 == 0 for l in trunk)):
      # Round 4: the dX chain runs alone on the main stream (every G_i keeps its own buffer) and the weight gradients of the
      # layers leave as batched launches (hugs_gemm_tn_batch): next to each other the two kernels of a layer took 430 + 270 us
      # for 250 + 200 us of work alone (profiles/r03_step_timeline.txt), and one launch over all layers cuts the reduction
      # into #CUs / 128 tiles = 2 pieces instead of 16 per layer.
      # (two launches -- the upper half under the lower half's dX GEMMs -- measured: 6.80 ms against 6.61 for one launch and
      # 6.83 for the per-layer form, same box: what is lost is the concurrency itself.  The data-parallel step therefore keeps
      # ONE launch as well and releases the whole trunk's gradient range to its all-reduce behind it.)
      two = _TN_BATCH == 2 and not fused_bwd
      cuts = [depth // 2, 0] if (two and depth >= 2) else [0]
      Gs = [Ga, Gb] + [ws.get(f'{tag}/G{k}', (M, W), self.tdt) for k in range(2, depth)]
      if fused_bwd:
        # (the backward's `tag` carries no level, the mask buffers do: two proposal levels of equal M must not share a table --
        #  level 0 was masked with level 1's relu bits.  Keyed on the addresses the table is made of.)
        key = ('mlp_tail_bwd', tag, theta.data_ptr(), M, d_raw.data_ptr()) + tuple(b.data_ptr() for b in bits_all)
        tab = ws.bufs.get(key)
        if tab is None:      # host arrays of device pointers (a function of buffer addresses only)
          ptr = lambda ts: np.ascontiguousarray([t.data_ptr() for t in ts], np.uint64)
          tab = ws.put_table(key, (ptr([self.wn[(spec.name, trunk[i]['name'], 'kernel')] for i in (1, 2, 3)]), ptr(bits_all),
                                   ptr([Gs[3], Gs[2], Gs[1], Gs[0]])))      # G_l of layer l lives in Gs[depth - 1 - l]
        _lib.call('hugs_mlp256_tail_bwd', dt, M, 3, d_raw, wd, tab[0].ctypes.data, tab[1].ctypes.data, tab[2].ctypes.data)
      g_of, hi = {}, depth - 1
      done = []
      for i in range(depth - 1, -1, -1):
        l = trunk[i]
        path = (spec.name, l['name'], 'kernel')
        G = Gs[depth - 1 - i]
        g_of[i] = G
        if i in cuts:          # G_hi .. G_i are final: their weight gradients go out as one launch on the side stream
          ev = new_event(); ev.record(main)
          with torch.cuda.stream(side):
            wait_event(side, ev)
            if i == 0 and before_dw is not None and _DW_AFTER_PROP:
              # the batched launch holds every CU for its whole duration (one workgroup per CU, all registers): whatever another
              # stream still has queued would sit behind it -- at the reference-default shape the proposal levels' last dX GEMM
              # ran 7.0 ms next to it and their own weight-gradient launch came after (profiles/r04_ref360_timeline.txt)
              for e_ in before_dw():
                wait_event(side, e_)
            items = []
            for j in range(hi, i - 1, -1):
              lj = trunk[j]
              gW = gview((spec.name, lj['name'], 'kernel'), padded=True)
              gb = gview((spec.name, lj['name'], 'bias'), True)
              if lj['concat']:
                items.append((M, W, W, acts[j], W, g_of[j], W, gW[:W], gb))
                items.append((M, spec.Fp, W, X0, spec.Fp, g_of[j], W, gW[W:], None))
              else:
                items.append((M, lj['kpad'], W, acts[j], lj['kpad'], g_of[j], W, gW, gb))
            self._tn_batch(items)
            e = new_event(); e.record(side)
            done.append(e)
            if leaf_done is not None:
              lk = lay.by_path[(spec.name, trunk[i]['name'], 'kernel')]
              lb_ = lay.by_path[(spec.name, trunk[hi]['name'], 'bias')]
              leaf_done(lk['off'], lb_['off'] + int(np.prod(lb_['pshape'])))
          hi = i - 1
        if i > 0 and not fused_bwd:
          bprev = lv['bits'][i - 1] if lv.get('bits') else None
          Wn_ = self.wn[path][:W] if l['concat'] else self.wn[path]
          if bprev is not None:
            _lib.call('hugs_gemm_nt_bits', dt, M, W, W, 0, G, W, None, 0, Wn_, W, None, 0, None, None, Gs[depth - i], W, None, bprev)
          else:
            _lib.call('hugs_gemm_nt', dt, M, W, W, 0, G, W, None, 0, Wn_, W, None, None, 1, 0, 0, acts[i], W, None, None,
                      Gs[depth - i], W)
      for e in done:
        wait_event(main, e)
      if not spec.disable_rgb and spec.use_viewdirs:
        wait_event(main, heads_done)
      return
    Gc = ws.get(tag + '/Gc', (M, W), self.tdt)
    Gd = ws.get(tag + '/Gd', (M, W), self.tdt)
    ring = [Ga, Gb, Gc, Gd]
    gi = 0
    ev_g = new_event()
    ev_g.record(main)
    tn_done = {}
    for i in range(spec.net_depth - 1, -1, -1):
      l = spec.layers[i]
      path = (spec.name, l['name'], 'kernel')
      gW = gview(path, padded=True)
      gb = gview((spec.name, l['name'], 'bias'), True)
      xin = acts[i]          # acts[0] = X0, acts[i] = Y_{i-1}
      with torch.cuda.stream(side):
        wait_event(side, ev_g)                       # G_i is ready
        if l['concat']:
          self._tn(M, W, W, xin, W, G, W, gW[:W], gb)
          self._tn(M, spec.Fp, W, X0, spec.Fp, G, W, gW[W:], None)
        else:
          self._tn(M, l['kpad'], W, xin, l['kpad'], G, W, gW, gb)
        e = new_event()
        e.record(side)
        tn_done[gi] = e
        if leaf_done is not None:
          lk, lb_ = lay.by_path[path], lay.by_path[(spec.name, l['name'], 'bias')]
          leaf_done(lk['off'], lb_['off'] + int(np.prod(lb_['pshape'])))
      if i > 0:
        nxt = (gi + 1) % 4
        if nxt in tn_done:                           # the dW that last read this buffer must be finished
          wait_event(main, tn_done.pop(nxt))
        # G_{i-1} = (G_i W_i[:W]^T) * (Y_{i-1} > 0)
        bprev = lv['bits'][i - 1] if lv.get('bits') else None
        if bprev is not None:
          _lib.call('hugs_gemm_nt_bits', dt, M, W, W, 0, G, W, None, 0, self.wn[path][:W] if l['concat'] else self.wn[path], W,
                    None, 0, None, None, ring[nxt], W, None, bprev)
        else:
          _lib.call('hugs_gemm_nt', dt, M, W, W, 0, G, W, None, 0, self.wn[path][:W] if l['concat'] else self.wn[path], W, None,
                    None, 1, 0, 0, acts[i], W, None, None, ring[nxt], W)
        ev_g = new_event()
        ev_g.record(main)
        G, gi = ring[nxt], nxt
    for e in tn_done.values():
      wait_event(main, e)
    if not spec.disable_rgb and spec.use_viewdirs:
      wait_event(main, heads_done)

  def _transient_backward(self, theta, grad, lv, rays, N, d_dt, d_ct, d_u, side=None):
    """Backward of the NeRF-W transient branch (heads -> trunk -> per-ray tra_vec part).  Returns G at the first
    transient layer's pre-activation [M, Ht]; writes the branch's weight gradients (=) and scatter-adds into
    TransientEmbed_0."""
    spec, S, lay, ws, dt = lv['spec'], lv['S'], self.layout, self.ws, self.dt
    M, Ht, dtn, t0, Bw = N * S, spec.net_width_transient, spec.net_depth_transient, spec.t0, spec.bottleneck_width
    gview = lambda p, padded=False: lay.view(grad, p, padded)
    tacts = lv['tacts']
    x3 = tacts[-1]
    ld_, lr_, lu_ = spec.layers[t0 + dtn:t0 + dtn + 3]
    G = ws.get('tbwd/Ga', (M, Ht), self.tdt)
    rws = ws.get('rgb_ws', (max(_lib.lib().cdll.hugs_rgb_bwd_ws_bytes() // 4, 1),))
    Wrt, _ = self._rgb_head(theta, spec, lr_, 'tbwd/rgbhead_t')
    _lib.call('hugs_rgb_bwd', dt, M, Ht, x3, Ht, Wrt, lv['rgb_t'], d_ct,
              spec.rgb_padding, G, Ht, gview((spec.name, lr_['name'], 'kernel')), gview((spec.name, lr_['name'], 'bias')), rws)
    if spec.rgb_premultiplier != 1.:
      for g_ in (gview((spec.name, lr_['name'], 'kernel')), gview((spec.name, lr_['name'], 'bias'))):
        _lib.call('hugs_affine', g_.numel(), g_, float(spec.rgb_premultiplier), 0.0, g_)
    dws = ws.get('dens_ws_t', (max(_lib.lib().cdll.hugs_density_bwd_ws_bytes(Ht) // 4, 1),))
    d_raw_t, d_raw_u = ws.get('tbwd/d_raw_t', (M,)), ws.get('tbwd/d_raw_u', (M,))
    # Round 5: only rgb_t -> d_raw -> G_3 -> ... -> G_0 is on the way to dBott and the trunk backward.  The branch's weight gradients
    # (two head column sums, three 128 x 128 products + reductions, the per-ray tra_vec part with its embedding scatter, the first
    # layer's product: ~0.4 ms of small launches that sat BETWEEN the dX GEMMs) go to the head weight-gradient stream; every G_i
    # keeps its own buffer until they have read it.
    cur = torch.cuda.current_stream()
    side = side if side is not None else cur
    def on_side(fn):
      ev = new_event(); ev.record(cur)
      with torch.cuda.stream(side):
        wait_event(side, ev)
        fn()
    _lib.call('hugs_density_bwd', dt, M, Ht, x3, Ht, d_dt, lv['raw_t'], spec.density_bias, d_raw_t, None, None, None)
    _lib.call('hugs_density_bwd', dt, M, Ht, x3, Ht, d_u, lv['raw_u'], 0.0, d_raw_u, None, None, None)
    def head_dw():
      _lib.call('hugs_density_bwd', dt, M, Ht, x3, Ht, None, lv['raw_t'], spec.density_bias, d_raw_t,
                gview((spec.name, ld_['name'], 'kernel')).reshape(-1), gview((spec.name, ld_['name'], 'bias')), dws)
      _lib.call('hugs_density_bwd', dt, M, Ht, x3, Ht, None, lv['raw_u'], 0.0, d_raw_u,
                gview((spec.name, lu_['name'], 'kernel')).reshape(-1), gview((spec.name, lu_['name'], 'bias')), dws)
    on_side(head_dw)
    _lib.call('hugs_rank1_add2_mask', dt, M, Ht, d_raw_t, lay.view(theta, (spec.name, ld_['name'], 'kernel')).reshape(-1),
              d_raw_u, lay.view(theta, (spec.name, lu_['name'], 'kernel')).reshape(-1), x3, Ht, G, Ht)
    for i in range(dtn - 1, 0, -1):
      l = spec.layers[t0 + i]
      path = (spec.name, l['name'], 'kernel')
      Gi, xin = G, tacts[i - 1]
      on_side(lambda Gi=Gi, xin=xin, path=path, l=l: self._tn(M, Ht, Ht, xin, Ht, Gi, Ht, gview(path), gview((spec.name, l['name'], 'bias'))))
      Gn = ws.get(f'tbwd/G{i - 1}', (M, Ht), self.tdt)
      _lib.call('hugs_gemm_nt', dt, M, Ht, Ht, 0, G, Ht, None, 0, self.wn[path], Ht, None, None, 1, 0, 0, tacts[i - 1], Ht,
                None, None, Gn, Ht)
      G = Gn
    lt = spec.layers[t0]
    gWt0 = gview((spec.name, lt['name'], 'kernel'))
    Wt0 = lay.view(theta, (spec.name, lt['name'], 'kernel'))
    d_rb = ws.get('tbwd/d_rb', (int(_lib.lib().cdll.hugs_raybias_bwd_ws_rows(N, 0, spec.num_tra)), Ht))
    def first_dw(G0=G):
      _lib.call('hugs_raybias_bwd', dt, N, S, Ht, 0, spec.num_tra, G0, Ht, None, lv['tra'], Wt0[Bw:], rays.get('embed_idx'), d_rb,
                gWt0[Bw:], gview(('TransientEmbed_0', 'embedding')))
      self._tn(M, Bw, Ht, lv['bott'], Bw, G0, Ht, gWt0[:Bw], gview((spec.name, lt['name'], 'bias')))
    on_side(first_dw)
    return G

  # ---- HA-NeRF ImplicitMask (per ray) --------------------------------------------------------------
  def mask_forward(self, theta, rays, N, zero_tra=False):
    """renderings[-1]['implicit_mask'] (models.py:327-328).  Returns the state mask_backward needs."""
    spec, lay, ws, dt = self.model.mask_spec, self.layout, self.ws, self.dt
    if getattr(self, 'weights_stale', False):      # (a caller that drives the engine directly after an optimizer step)
      self.refresh_weights(theta)
    Np = _round_up(N, 128)
    tra = None
    if not zero_tra:
      tra = ws.get('tra_vec', (N, spec.T))
      _lib.call('hugs_glo_gather', N, spec.T, lay.view(theta, ('TransientEmbed_0', 'embedding')), rays['embed_idx'], 0, tra)
    X0 = ws.get('mask/X0', (Np, spec.kpad), self.tdt)
    if Np != N:
      X0[N:].zero_()
    _lib.call('hugs_mask_input_fwd', N, spec.T, spec.deg_coord, rays['pix_coords'], tra, spec.kpad, dt, X0)
    acts, x, W = [X0], X0, spec.net_width
    for i in range(spec.net_depth):
      l = spec.layers[i]
      Y = ws.get(f'mask/Y{i}', (Np, W), self.tdt)
      _lib.call('hugs_gemm_nt', dt, Np, W, l['kpad'], 0, x, l['kpad'], None, 0, self.wt[(spec.name, l['name'], 'kernel')],
                l['kpad'], lay.view(theta, (spec.name, l['name'], 'bias')), None, 1, 0, 1, None, 0, None, None, Y, W)
      acts.append(Y)
      x = Y
    lh = spec.layers[spec.net_depth]
    mask = ws.get('mask/out', (N,))
    _lib.call('hugs_mask_head_fwd', dt, N, W, x, W, lay.view(theta, (spec.name, lh['name'], 'kernel')).reshape(-1),
              lay.view(theta, (spec.name, lh['name'], 'bias')), mask)
    return dict(acts=acts, mask=mask, N=N, Np=Np, zero_tra=zero_tra)

  def mask_backward(self, theta, grad, st, rays, d_mask):
    """Gradients of ImplicitMask_0 (=) and TransientEmbed_0 (+=, caller zeroes) from d loss / d mask [N]."""
    spec, lay, ws, dt = self.model.mask_spec, self.layout, self.ws, self.dt
    N, Np, W, acts = st['N'], st['Np'], spec.net_width, st['acts']
    gview = lambda p, padded=False: lay.view(grad, p, padded)
    lh = spec.layers[spec.net_depth]
    d_raw = ws.get('mask/d_raw', (Np,))
    _lib.call('hugs_mask_head_bwd', dt, N, Np, W, acts[-1], W, st['mask'], d_mask, d_raw,
              gview((spec.name, lh['name'], 'kernel')).reshape(-1), gview((spec.name, lh['name'], 'bias')))
    Ga = ws.get('mask/Ga', (Np, W), self.tdt)
    Gb = ws.get('mask/Gb', (Np, W), self.tdt)
    _lib.call('hugs_rank1_mask', dt, Np, W, d_raw, lay.view(theta, (spec.name, lh['name'], 'kernel')).reshape(-1), acts[-1], W,
              Ga, W)
    G, other = Ga, Gb
    for i in range(spec.net_depth - 1, -1, -1):
      l = spec.layers[i]
      path = (spec.name, l['name'], 'kernel')
      self._tn(Np, l['kpad'], W, acts[i], l['kpad'], G, W, gview(path, padded=True), gview((spec.name, l['name'], 'bias')))
      if i > 0:
        _lib.call('hugs_gemm_nt', dt, Np, W, W, 0, G, W, None, 0, self.wn[path], W, None, None, 1, 0, 0, acts[i], W, None, None,
                  other, W)
        G, other = other, G
      elif not st['zero_tra']:
        # dX0 = G_0 W_0^T; its tra_vec columns go to the embedding rows of the rays' cameras
        dX0 = ws.get('mask/dX0', (Np, spec.kpad), self.tdt)
        _lib.call('hugs_gemm_nt', dt, Np, spec.kpad, W, 0, G, W, None, 0, self.wn[path], W, None, None, 1, 0, 0, None, 0, None,
                  None, dX0, spec.kpad)
        _lib.call('hugs_embed_scatter_add', dt, N, spec.T, dX0, spec.kpad, spec.E, rays['embed_idx'],
                  gview(('TransientEmbed_0', 'embedding')))

  def _require_fresh_weights(self):
    """backward_level reads the operand copies (wn, wfold) the forward of the SAME step used: an optimizer step in between means
    the caller mixed two parameter versions."""
    if getattr(self, 'weights_stale', False):
      raise _lib.HugsError('the operand copies of the weights are stale (an optimizer step has moved the masters since the last '
                           'forward): call Engine.forward / refresh_weights first')

  def _side_stream(self, lane=0):
    """HIP streams next to the caller's: lane 0 carries the weight-gradient GEMMs of the NerfMLP backward (and the
    HA-NeRF mask MLP), lane 1 the whole proposal-level backward, lane 2 its weight-gradient GEMMs."""
    cap = getattr(self, 'capture_lanes', None)      # a step being captured into a hipGraph: only these lanes fork
    if cap is not None and lane not in cap:
      return torch.cuda.current_stream()
    if not hasattr(self, '_side'):
      self._side = {}
    if lane not in self._side:
      import os
      # HUGS_SINGLE_STREAM=1 (measurement hook): everything stays on the compute stream
      # HUGS_SIDE_LANES=0,3 (measurement hook): only the listed lanes get a stream of their own
      lanes = os.environ.get('HUGS_SIDE_LANES')
      single = os.environ.get('HUGS_SINGLE_STREAM') == '1' or (lanes is not None and str(lane) not in lanes.split(','))
      self._side[lane] = torch.cuda.current_stream() if single else torch.cuda.Stream(device=self.device)
    return self._side[lane]

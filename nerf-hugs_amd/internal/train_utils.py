"""Training step and model creation (reference MipNeRF360/internal/train_utils.py): the same entry points
(`setup_model`, `create_train_step`, `create_optimizer`, `create_render_fn`, `setup_finetune_model`) and the
same `train_pstep(rngs, state, batch, train_frac, inlier_thresholds) -> (state, stats, rngs)` contract.

One process drives one GPU; data parallelism = one process per GPU with a single RCCL all-reduce of the flat
gradient buffer (+ the step's scalar stats in its tail) per step, replacing jax.lax.pmean
(train_utils.py:457-459).  Loss normalisers stay per device, as in the reference (they run before pmean)."""
import math
import weakref

import numpy as np
import torch
import torch.distributed as dist

from .. import _lib
from . import engine as _engine
from . import image
from . import math as hmath
from . import models
from . import random as hrandom
from . import utils

_STATE_GEN = __import__('itertools').count(1)
# bench.py --gpus N: when a list, every data-parallel step appends (event, event) recorded on the compute stream around the point
# where it waits for its gradient all-reduces: the elapsed time between the two is the part of the exchange that the backward
# pass did NOT hide (the stream has nothing else to do there)
AR_PROFILE = None
STAT_TAIL = 64  # floats appended to the gradient buffer for the per-step scalars that get pmean'ed (<= 7 levels)


def _tail_slots(L):
  """Offsets of the per-step scalars in the stat tail for L sampling levels: (data [2L: mse, data loss per level],
  interlevel [L-1], distortion [1], robust [5L], hanerf [2], nerfw [2])."""
  o = dict(data=0, interlevel=2 * L, distortion=3 * L - 1, robust=3 * L, hanerf=8 * L, nerfw=8 * L + 2)
  if 8 * L + 4 > STAT_TAIL:
    raise NotImplementedError(f'{L} sampling levels: the stat tail holds {STAT_TAIL} floats (<= 7 levels)')
  return o


class TrainState:
  """Counterpart of flax TrainState (train_utils.py:509-512): .step, .params plus Adam moments, all views of
  flat fp32 device buffers that the step updates in place (the reference donates its state too)."""

  def __init__(self, model, flat, hyper):
    self.model = model
    self.flat = flat
    self.m = torch.zeros_like(flat)
    self.v = torch.zeros_like(flat)
    self.step = 0
    self.gen = next(_STATE_GEN)  # process-unique id of this state (Engine.weights_current: allocator addresses get reused)
    self.hyper = hyper           # dict(lr_fn, b1, b2, eps, trainable (device int32 per leaf) or None)
    self.params = model.variables(flat)

  def state_dict(self):
    return {'step': self.step, 'params': self.flat, 'mu': self.m, 'nu': self.v}


class LazyStats(dict):
  """stats dict whose values materialise from one packed device->host copy on first access, so the step
  stays asynchronous (the reference returns device arrays that the host reads every print_every)."""

  def __init__(self, packed_dev, build, host=None, pool=None):
    super().__init__()
    if host is None:
      self._host = torch.empty(packed_dev.shape, dtype=packed_dev.dtype, pin_memory=True)
      self._host.copy_(packed_dev, non_blocking=True)
    else:
      # a captured step's last launch has been told to write the packed stats straight into this pinned slot (no copy on the
      # stream); the slot goes back to the step function's pool when this dict dies
      self._host = host
    self._pool = pool
    self._ev = torch.cuda.Event()
    self._ev.record()
    self._build = build
    self._done = False

  def __del__(self):
    pool, self._pool = getattr(self, '_pool', None), None
    if pool is not None:
      pool.append(self._host)

  def _mat(self):
    if not self._done:
      self._ev.synchronize()
      super().update(self._build(self._host.numpy().copy()))
      self._done = True

  def __getitem__(self, k):
    self._mat()
    return super().__getitem__(k)

  def __contains__(self, k):
    self._mat()
    return super().__contains__(k)

  def keys(self):
    self._mat()
    return super().keys()

  def items(self):
    self._mat()
    return super().items()

  def __iter__(self):
    self._mat()
    return super().__iter__()

  def get(self, k, d=None):
    self._mat()
    return super().get(k, d)


def _world():
  return dist.get_world_size() if dist.is_initialized() else 1


def create_optimizer(config, variables, model=None):
  """Adam(lr schedule) state (train_utils.py:487-512).  variables: flat buffer."""
  lr_fn = lambda step: hmath.learning_rate_decay(step, config.lr_init, config.lr_final, config.max_steps,
                                                 config.lr_delay_steps, config.lr_delay_mult)
  hyper = dict(lr_fn=lr_fn, b1=config.adam_beta1, b2=config.adam_beta2, eps=config.adam_eps, trainable=None)
  return TrainState(model, variables, hyper), lr_fn


def create_finetune_optimizer(config, variables, model=None):
  """Adam on leaves whose path contains 'embedding', everything else frozen (train_utils.py:515-552)."""
  lr_fn = lambda step: hmath.learning_rate_decay(step, config.finetune_lr_init, config.finetune_lr_final,
                                                 config.finetune_max_steps, config.finetune_lr_delay_steps,
                                                 config.finetune_lr_delay_mult)
  tr = torch.tensor([1 if 'embedding' in lf['path'] else 0 for lf in model.layout.leaves], dtype=torch.int32,
                    device=variables.device)
  hyper = dict(lr_fn=lr_fn, b1=config.finetune_adam_beta1, b2=config.finetune_adam_beta2, eps=config.finetune_adam_eps,
               trainable=tr, finetune=True)
  return TrainState(model, variables, hyper), lr_fn


def _summarize(layout, per_leaf, reduce, post=lambda x: x):
  """summarize_tree (train_utils.py:61-69) to depth 3 from per-leaf values."""
  groups = {}
  for lf, val in zip(layout.leaves, per_leaf):
    for d in range(1, len(lf['path']) + 1):
      groups.setdefault('/'.join(lf['path'][:d]), []).append(val)
  return {k: post(reduce(v)) for k, v in groups.items()}


_AR_BUCKETS = __import__('os').environ.get('HUGS_AR_BUCKETS', '1') != '0'
# HUGS_NT_DYNQ=1: the step's persistent NT GEMMs draw their tiles from per-XCD ticket counters (csrc/hugs_gemm_dq.inc) instead of the
# static walk.  Built in round 6 against the launches that share the chip with side-stream kernels; bit-identical; measured SLOWER
# (5.934 / 5.942 vs 5.846 / 5.829 ms same box, and the dX launches' average-vs-fastest gap unchanged: profiles/r06_tile_queue_ab.txt) -- off
_NT_DYNQ = __import__('os').environ.get('HUGS_NT_DYNQ', '0') == '1'
# Replay of the train step as a captured hipGraph: '0' never, '1' whenever the step is capturable, 'auto' (default): one
# process -- whenever capturable; data parallel -- when the per-GPU step is small enough to be host-bound (rays x samples per
# step <= HUGS_STEP_GRAPH_ROWS: the graphs give up the overlap of the bucketed all-reduces with the backward pass)
_STEP_GRAPH = __import__('os').environ.get('HUGS_STEP_GRAPH', 'auto')
# lanes (Engine._side_stream) that keep a stream of their own inside the captured step; the others run on the stream they are
# called from.  Lane 2 (the proposal level's weight-gradient stream) forks from lane 1, a forked stream: see engine.wait_event
_GRAPH_LANES = tuple(int(x) for x in __import__('os').environ.get('HUGS_STEP_GRAPH_LANES', '1,3,4').split(',') if x != '')
# HUGS_STEP_GRAPH_TRANSIENT=0: the HA-NeRF / NeRF-W steps stay eager (round 5: their per-step scalar is device-resident and they capture)
_GRAPH_TRANSIENT = __import__('os').environ.get('HUGS_STEP_GRAPH_TRANSIENT', '1') != '0'
_GRAPH_TYPES = (None, 'withmask', 'robustnerf') + (('hanerf', 'nerfw') if _GRAPH_TRANSIENT else ())
# HUGS_INTERLEVEL_ON_PROP=0|1|auto: the interlevel loss kernels on the proposal stream in front of the proposal levels' backward (1) or on
# the main stream in front of the whole backward pass (0, rounds 1-4).  auto: on the proposal stream for small steps (<= HUGS_STEP_GRAPH_ROWS
# rows, the launch-bound regime: 128 rays 1.209 -> 1.195 ms), on the main stream for large ones (there it delays the proposal levels'
# backward into the trunk's first dX GEMMs: cfg2 +0.3-0.9 % in two same-box A/Bs)
_IL_ON_PROP = __import__('os').environ.get('HUGS_INTERLEVEL_ON_PROP', 'auto')
_STEP_GRAPH_ROWS = int(__import__('os').environ.get('HUGS_STEP_GRAPH_ROWS', '100000'))


def uncovered_ranges(layout, covered, total):
  """[lo, hi) ranges of the flat gradient buffer (+ stat tail) that no bucket all-reduced: every LEAF that is not
  wholly inside one covered range is reduced in full (adjacent ones merged), plus the tail behind the last leaf.  Only
  the alignment padding between leaves (never written, always zero) may stay out."""
  cov = sorted(covered)
  todo = []
  for lf in layout.leaves:
    lo, hi = lf['off'], lf['off'] + int(np.prod(lf['pshape']))
    if not any(c0 <= lo and hi <= c1 for c0, c1 in cov):
      # merge neighbours (leaves are padded to 4 floats) -- but never across a covered leaf: a [1] / [3] bias of a covered
      # layer sitting in that 4-float gap would be all-reduced (SUM) a second time
      if todo and todo[-1][1] >= lo - 4 and not any(c0 < lo and c1 > todo[-1][1] for c0, c1 in cov):
        todo[-1][1] = hi
      else:
        todo.append([lo, hi])
  if todo and todo[-1][1] >= layout.size - 4:
    todo[-1][1] = total
  elif layout.size < total:
    todo.append([layout.size, total])
  for lo, hi in todo:      # what is reduced here must not overlap anything a bucket already reduced (a second SUM of a leaf)
    if any(c0 < hi and lo < c1 for c0, c1 in cov):
      raise RuntimeError(f'all-reduce ranges overlap: [{lo}, {hi}) against the buckets {cov}')
  return [tuple(t) for t in todo]


def create_train_step(model, config, is_finetune=False):
  """Creates the training function (train_utils.py:372-484).  Returned callable:
  train_pstep(rng, state, batch, train_frac, inlier_thresholds) -> (state, stats, rng)."""
  layout = model.layout
  L = model.num_levels
  tt = None if is_finetune else config.transient_type
  if tt not in (None, 'withmask', 'robustnerf', 'hanerf', 'nerfw'):
    raise ValueError()
  if tt == 'robustnerf':
    assert config.robustnerf_inner_patch_size <= config.patch_size, \
        'patch_size must be larger than robustnerf_inner_patch_size.'
  if config.data_loss_type not in ('mse', 'charb'):
    assert False
  TS = _tail_slots(L)      # stat-tail slots sized by the number of levels (any L the 64-float tail holds: <= 7)
  o_il, o_dist, o_rob, o_han, o_nw = TS['interlevel'], TS['distortion'], TS['robust'], TS['hanerf'], TS['nerfw']
  # train_utils.py:444-447: loss += sum_k m_k * ||theta_k||^2 over summarize_tree keys (a module, 'module/layer' or a
  # leaf path); the gradient 2 m_k theta is added to the leaves under each key.
  decay = []
  if not is_finetune and config.weight_decay_mults:
    known = set()
    for lf in layout.leaves:
      for d in range(1, len(lf['path']) + 1):
        known.add('/'.join(lf['path'][:d]))
    for key, mult in dict(config.weight_decay_mults).items():
      if key not in known:
        raise KeyError(f'weight_decay_mults: {key!r} is not a parameter group (have e.g. {sorted(known)[:4]})')
      for lf in layout.leaves:
        if '/'.join(lf['path'])[:len(key)] == key and ('/'.join(lf['path']) == key or '/'.join(lf['path'])[len(key)] == '/'):
          decay.append((lf['off'], int(np.prod(lf['pshape'])), float(mult)))
  cache = {}

  def optimizer_step(state, grad, gscale=1.0, dyn=None, pub=None):
    """The second half of the reference's train_step on a gradient buffer in the flat layout
    (train_utils.py:461-473): grad_norms / grad_maxes, clip_gradients per module (value clip, then norm clip with
    `eps + norm`), nan_to_num, Adam (optax.adam: bias-corrected moments, eps outside the sqrt, schedule at the
    0-based count), opt_update_* stats.  The compute-dtype operand copies are NOT re-cast here: the step marks them stale
    (`eng.weights_stale`) and the next step / `Engine.forward` / `mask_forward` re-casts first thing; `backward_level` raises on
    stale copies.  `gscale`: factor the kernels apply to the buffer first (1/world after a SUM all-reduce).  Returns the per-leaf
    stat buffer."""
    eng = model.engine(state.flat.device)
    ws = eng.ws
    nch, nleaf, nmod = layout.chunks.shape[0], len(layout.leaves), len(layout.modules)
    part1 = ws.get('opt_part1', (nch * 4,))
    # (the per-leaf stats land directly behind the stat tail in the buffer the host reads: one copy less per step)
    leaf_stats = ws.get('stats_packed', (STAT_TAIL + nleaf * 6 + 16,))[STAT_TAIL:]
    mod_scale = leaf_stats[nleaf * 6:nleaf * 6 + 16]
    _lib.call('hugs_opt_stats', nch, nleaf, nmod, eng.chunks, eng.leaf_info, state.flat, grad, gscale, config.grad_max_val,
              config.grad_max_norm, part1, leaf_stats[:nleaf * 4], mod_scale)
    h = state.hyper
    count = state.step                      # optax's 0-based update count
    lr = h['lr_fn'](count)
    t = count + 1
    part2 = ws.get('opt_part2', (nch * 2,))
    if dyn is not None and pub is not None:      # + its last launch publishes the step's results (step_finish)
      _lib.call('hugs_opt_adam_pub', nch, nleaf, eng.chunks, eng.leaf_info, state.flat, grad, state.m, state.v, mod_scale, h['trainable'],
                gscale, config.grad_max_val, dyn[1:4], h['b1'], h['b2'], h['eps'], part2, leaf_stats[nleaf * 4:nleaf * 6], pub.ctypes.data)
    elif dyn is not None:     # a step being captured: {lr, 1 - b1^t, 1 - b2^t} are read from dyn[1:4] (step_scalars below writes them)
      _lib.call('hugs_opt_adam_dyn', nch, nleaf, eng.chunks, eng.leaf_info, state.flat, grad, state.m, state.v, mod_scale, h['trainable'],
                gscale, config.grad_max_val, dyn[1:4], h['b1'], h['b2'], h['eps'], part2, leaf_stats[nleaf * 4:nleaf * 6])
    else:
      _lib.call('hugs_opt_adam', nch, nleaf, eng.chunks, eng.leaf_info, state.flat, grad, state.m, state.v, mod_scale, h['trainable'], gscale,
                config.grad_max_val, lr, h['b1'], h['b2'], h['eps'], 1.0 - h['b1']**t, 1.0 - h['b2']**t, part2,
                leaf_stats[nleaf * 4:nleaf * 6])
    # (the compute-dtype operand copies are re-cast at the START of the next step, on a lane of their own underneath its
    # jitter / sampler / encoder launches -- step_core -- instead of here at the end of the serial optimizer tail)
    eng._cast_src = None
    eng.weights_stale = True
    state.step += 1
    return leaf_stats

  def mask_size_mult(train_frac):
    """train_utils.py:190-193: HA-NeRF's mask-size weight decays from _max to _min with the step."""
    return max(config.hanerf_mask_size_loss_mult_min, config.hanerf_mask_size_loss_mult_max *
               math.exp(-float(train_frac) * config.max_steps * config.hanerf_mask_size_loss_mult_k))

  def step_scalars(state, train_frac):
    """The per-step scalars a captured step reads from device memory: (anneal, lr, 1 - b1^t, 1 - b2^t, HA-NeRF mask-size weight)."""
    h = state.hyper
    t = state.step + 1
    return (model.engine(state.flat.device).anneal_factor(float(train_frac)), h['lr_fn'](state.step), 1.0 - h['b1']**t, 1.0 - h['b2']**t,
            mask_size_mult(train_frac) if tt == 'hanerf' else 0.0)

  def graph_signature(rng, state, N, train_frac, inlier_thresholds):
    """None when this step cannot be replayed from captured hipGraphs, else the key of its graphs.  Capturable: the plain /
    static-mask / RobustNeRF losses, jitter from a jax key through the fused chain kernel (or none), no near-plane annealing (its histogram
    is rewritten from the host).  Data parallel: TWO graphs (forward + backward | clip + Adam) around ONE eager all-reduce of
    the gradient buffer -- the collective is not captured."""
    if _STEP_GRAPH == '0' or _lib.PROFILE is not None or tt not in _GRAPH_TYPES or inlier_thresholds is not None:
      return None      # (RobustNeRF: with the thresholds fed back on the device -- inlier_thresholds=None -- not handed over by the host)
    if model.near_anneal_rate is not None or model.has_noise():
      return None
    if (model.nerf_spec.num_tra > 0 or model.mask_spec is not None) and not _GRAPH_TRANSIENT:
      return None
    if hrandom.is_key(rng):
      if not (config.randomized and L <= hrandom.step_jitter_max_levels()):
        return None
      kind = 'key'
    elif rng is None:
      kind = 'none'
    else:
      return None           # a torch.Generator (its philox offset lives on the host) or explicit draws
    if _STEP_GRAPH == 'auto' and _world() > 1 and N * (model.num_prop_samples * (L - 1) + model.num_nerf_samples) > _STEP_GRAPH_ROWS:
      return None           # data parallel, large per-GPU batches: the eager step hides its bucketed all-reduces under the backward
    return (kind, N, state.gen, state.flat.data_ptr(), state.m.data_ptr(), is_finetune, _world())

  graphs = {}

  def _handed(ent, rng):
    """rng IS the key tensor the previous replay handed back, untouched: its value already sits in ent['key']."""
    h = ent.get('key_handed')
    return h is not None and h[0]() is rng and rng._version == h[1]

  def train_step(rng, state, batch, train_frac, inlier_thresholds):
    eng = model.engine(state.flat.device)
    dev = state.flat.device
    rays = models.rays_to_dict(batch.rays, dev)
    gt = batch.rgb[..., :3].reshape(-1, 3).to(device=dev, dtype=torch.float32).contiguous()
    N = gt.shape[0]
    for S_ in (model.num_prop_samples, model.num_nerf_samples):
      if (N * S_) % 128:
        raise ValueError(f'per-device batch of {N} rays x {S_} samples is not a multiple of the 128-row GEMM tile: '
                         'use a batch size that is a multiple of 4 (eval pads ragged chunks itself)')
    sig = graph_signature(rng, state, N, train_frac, inlier_thresholds)
    ent = None
    if sig is not None:
      ent = graphs.setdefault(sig, {'calls': 0})
      ent['calls'] += 1
    if ent is None or ent['calls'] <= 2:       # (two eager steps first: every workspace buffer and cache exists before the capture)
      rng = step_core(state, rays, gt, N, rng, train_frac, inlier_thresholds, None, True)
      packed = step_finish(state, None)
      return state, LazyStats(packed, stats_builder(state)), rng
    # ---- replay (capture on first use) of the step as ONE hipGraph on the current stream: ~0.05 ms of host time instead of
    # ~1.6 ms of Python / ctypes for ~90 launches.  Inputs are staged into the buffers the graph was captured on, the
    # per-step scalars enter through one small launch, the jax key advances in place in a buffer that is handed back.
    if 'graph' not in ent:
      ent['rays'] = {k: torch.empty_like(v) for k, v in rays.items()}
      ent['gt'] = torch.empty_like(gt)
      ent['dyn'] = torch.zeros(8, dtype=torch.float32, device=dev)
      ent['key'] = torch.zeros(2, dtype=torch.int32, device=dev) if sig[0] == 'key' else None
      ent['ptrs'] = torch.zeros(2, dtype=torch.int64, device=dev)     # {pinned stats slot, fresh key buffer} of the step in flight
      ent['host_pool'] = []
    names = list(rays)      # (the engine adds derived entries -- 'dir_enc' -- to the dict it is handed: they are not inputs)
    srcs = [rays[k] for k in names if rays[k].data_ptr() != ent['rays'][k].data_ptr()] + ([gt] if gt.data_ptr() != ent['gt'].data_ptr() else [])
    dsts = [ent['rays'][k] for k in names if rays[k].data_ptr() != ent['rays'][k].data_ptr()] + ([ent['gt']] if gt.data_ptr() != ent['gt'].data_ptr() else [])
    # the advanced key is handed back in a FRESH tensor, as the eager path does (ent['key'] is overwritten by every replay, and a
    # caller that keeps an earlier key -- a checkpoint of the rng, an eval stream -- must not see it change): the step's last launch
    # writes it there.  A caller that hands the previous step's key straight back (the train loop) finds it in ent['key'] already.
    key_new = torch.empty_like(ent['key']) if ent['key'] is not None else None
    if ent['key'] is not None and rng.data_ptr() != ent['key'].data_ptr() and not _handed(ent, rng):
      # (the staging launch dereferences raw addresses: a key restored on the CPU or living on another device goes through torch first)
      srcs.append((rng if rng.device == ent['key'].device else rng.to(ent['key'].device)).contiguous()); dsts.append(ent['key'])
    # ONE launch: every input copy (all of them 4-byte element types) + the step's four scalars + the two addresses the step's
    # last launch publishes to (round 5; it was two _foreach_copy_ launches, a key copy and hugs_set_floats: ~40 us of launch
    # latency in front of a 1.3 ms step at 128 rays)
    npk = STAT_TAIL + len(layout.leaves) * 6 + 16
    host = ent['host_pool'].pop() if ent['host_pool'] else torch.empty((npk,), dtype=torch.float32, pin_memory=True)
    one = len(srcs) <= 16 and all(s_.element_size() == 4 and s_.is_contiguous() and s_.numel() == d_.numel() and s_.device == d_.device
                                  for s_, d_ in zip(srcs, dsts))
    if not one:
      for s_, d_ in zip(srcs, dsts):
        d_.copy_(s_)
      srcs, dsts = [], []
    tb = ent.setdefault('stage_tab', (np.zeros(16, np.uint64), np.zeros(16, np.uint64), np.zeros(16, np.int32)))
    for i_, (s_, d_) in enumerate(zip(srcs, dsts)):
      tb[0][i_], tb[1][i_], tb[2][i_] = s_.data_ptr(), d_.data_ptr(), s_.numel()
    sc_ = ent.setdefault('scal', np.zeros(8, np.float32))
    scal_ = step_scalars(state, train_frac)
    sc_[:5] = scal_
    if tt == 'hanerf':
      cache['mask_size_mult'] = scal_[4]      # (the stats builder reports it: step_core only runs at capture time)
    _lib.call('hugs_stage_step_pub', len(srcs), tb[0].ctypes.data, tb[1].ctypes.data, tb[2].ctypes.data, ent['dyn'], 5, sc_.ctypes.data,
              ent['ptrs'], host.data_ptr(), key_new)
    if 'graph' not in ent:
      world = _world()
      g, g2 = torch.cuda.CUDAGraph(), (torch.cuda.CUDAGraph() if world > 1 else None)
      cap = torch.cuda.Stream(device=dev)
      cap.wait_stream(torch.cuda.current_stream())
      step0 = state.step
      eng.capture_lanes = _GRAPH_LANES
      _engine.KEEP_EVENTS = []
      ok = False
      try:
        with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
          ent['key_out'] = step_core(state, ent['rays'], ent['gt'], N, ent['key'], train_frac, None, ent['dyn'], False)
          if ent['key'] is None:
            ent['key_out'] = None
          if world == 1:
            packed = step_finish(state, ent['dyn'], ent)
        if world > 1:
          with torch.cuda.graph(g2, stream=cap, capture_error_mode='thread_local'):
            packed = step_finish(state, ent['dyn'], ent)
        ok = True
      finally:
        eng.capture_lanes = None
        ent['events'], _engine.KEEP_EVENTS = _engine.KEEP_EVENTS, None
        state.step = step0          # (the capture ran the host side of the step once without executing anything)
        if not ok:                  # a failed capture leaves no half-built entry behind: the next call starts over (eagerly)
          graphs.pop(sig, None)
      torch.cuda.current_stream().wait_stream(cap)
      ent['graph'], ent['graph_opt'], ent['packed'] = g, g2, packed
    ent['graph'].replay()
    if ent['graph_opt'] is not None:      # pmean(grad), pmean(stats) (train_utils.py:457-459): one SUM over the whole buffer + stat tail
      if AR_PROFILE is not None:
        e0_ = torch.cuda.Event(enable_timing=True); e0_.record()
      dist.all_reduce(eng.ws.get('grad', (layout.size + STAT_TAIL,)), op=dist.ReduceOp.SUM)
      if AR_PROFILE is not None:
        e1_ = torch.cuda.Event(enable_timing=True); e1_.record()
        AR_PROFILE.append((e0_, e1_))
      ent['graph_opt'].replay()
    state.step += 1
    eng._cast_src = None          # (the replayed Adam update has moved the masters; the next step re-casts first thing)
    eng.weights_stale = True
    if key_new is not None:
      ent['key_handed'] = (weakref.ref(key_new), key_new._version)
    return state, LazyStats(None, stats_builder(state), host=host, pool=ent['host_pool']), (key_new if key_new is not None else rng)

  def copy_many(pairs):
    """dst.copy_(src) for a few small fp32 buffers in ONE launch (hugs_stage_step without scalars) instead of one blit each."""
    pairs = [(s_, d_) for s_, d_ in pairs if s_.numel() > 0]
    if not all(s_.is_contiguous() and d_.is_contiguous() and s_.element_size() == 4 and s_.numel() == d_.numel() for s_, d_ in pairs):
      for s_, d_ in pairs:
        d_.copy_(s_)
      return
    a = np.array([s_.data_ptr() for s_, _ in pairs], np.uint64)
    b = np.array([d_.data_ptr() for _, d_ in pairs], np.uint64)
    w = np.array([s_.numel() for s_, _ in pairs], np.int32)
    _lib.call('hugs_stage_step', len(pairs), a.ctypes.data, b.ctypes.data, w.ctypes.data, None, 0, 0.0, 0.0, 0.0, 0.0)

  def step_core(state, rays, gt, N, rng, train_frac, inlier_thresholds, dyn, reduce):
    """The first part of the step: forward, losses, backward and -- when `reduce` -- the all-reduces of the gradient buffer
    (buckets issued underneath the backward pass + the rest).  dyn: None, or the device scalars of a captured step.  Returns
    the advanced rng."""
    # round 6: the persistent NT GEMMs of the step draw their tiles from per-XCD ticket counters (csrc/hugs_gemm_dq.inc): one 32-byte
    # slot per launch out of this buffer, zeroed first thing on the step's stream (inside a captured step: a memset node per replay)
    eng0 = model.engine(state.flat.device)
    if _NT_DYNQ and eng0.dt:
      qbuf = eng0.ws.get('nt_tile_queues', (64 * 8,), torch.int32)
      _lib.call('hugs_gemm_nt_queue_begin', qbuf, qbuf.numel() * 4)
      try:
        return _step_core(state, rays, gt, N, rng, train_frac, inlier_thresholds, dyn, reduce)
      finally:
        _lib.call('hugs_gemm_nt_queue_end', 0)
    return _step_core(state, rays, gt, N, rng, train_frac, inlier_thresholds, dyn, reduce)

  def _step_core(state, rays, gt, N, rng, train_frac, inlier_thresholds, dyn, reduce):
    eng = model.engine(state.flat.device)
    dev = state.flat.device
    ws = eng.ws
    world = _world()
    # operand copies of the weights (bf16 Wt / Wn, the folded head matrix): cast on a lane of their own; the first MLP GEMM waits
    ev_w = None
    if dyn is not None or not eng.weights_current(state):
      main_w, lane_w = torch.cuda.current_stream(), eng._side_stream(4)
      e0 = _engine.new_event(); e0.record(main_w)
      with torch.cuda.stream(lane_w):
        _engine.wait_event(lane_w, e0)
        eng.refresh_weights(state.flat, owner=state)
        ev_w = _engine.new_event(); ev_w.record(lane_w)
        # behind the casts on the same lane: whatever depends on the rays and the fp32 masters only (first used by the final level's view
        # layer, which waits for rays['_ev_rays'])
        eng.encode_rays(state.flat, rays, N)
        rays['_ev_rays'] = _engine.new_event(); rays['_ev_rays'].record(lane_w)
    u01 = None
    if isinstance(rng, (list, tuple)):             # explicit U[0,1) draws, one [N] (or [N,S]) tensor per level: the
      if len(rng) != L:                            # numbers jax.random.uniform handed the reference (fixtures, tests)
        raise ValueError(f'explicit jitter needs one tensor per level ({L}), got {len(rng)}')
      u01 = [u.to(device=dev, dtype=torch.float32).contiguous() for u in rng]
    elif hrandom.is_key(rng):                        # jax stream: rng, key = random.split(rng) (train_utils.py:408)
      if config.randomized and L <= hrandom.step_jitter_max_levels() and not model.has_noise():
        u01, rng = model.step_jitter(rng, N)         # that split + every level's split / uniform / split in one launch
      else:
        rng, key = hrandom.split(rng)
        if config.randomized:
          u01, _ = model.level_jitter(key, N)
    elif config.randomized and rng is not None:
      u01 = []
      for l in range(L):
        S = model.num_prop_samples if l < L - 1 else model.num_nerf_samples
        u01.append(torch.rand((N,) if model.single_jitter else (N, S), generator=rng, device=dev))
    mask_st, ev_mask_bwd = None, None
    if tt == 'hanerf':
      # the per-ray ImplicitMask MLP is independent of the level pipeline until the loss: it runs on the side stream
      # underneath the NerfMLP forward (and its backward underneath the level backward)
      main_s, side_s = torch.cuda.current_stream(), eng._side_stream()
      ev0 = _engine.new_event(); ev0.record(main_s)
      with torch.cuda.stream(side_s):
        _engine.wait_event(side_s, ev0)
        if ev_w is not None:
          _engine.wait_event(side_s, ev_w)
        mask_st = eng.mask_forward(state.flat, rays, N)
        ev_mask = _engine.new_event(); ev_mask.record(side_s)
    levels = eng.forward(state.flat, rays, float(train_frac), u01, False, False, anneal_dev=None if dyn is None else dyn[0:1],
                         weights_ready=ev_w)
    if rays.get('_ev_rays') is not None:
      # (a model without a view layer -- use_viewdirs = False, rgb disabled -- never waited for the ray encodings of the weight-cast lane:
      #  inside a capture that is a stream left unjoined, outside it a dangling dependency for the backward's reads of dir_enc / glo)
      _engine.wait_event(torch.cuda.current_stream(), rays['_ev_rays'])
    if mask_st is not None:
      _engine.wait_event(main_s, ev_mask)

    grad = ws.get('grad', (layout.size + STAT_TAIL,))
    tail = grad[layout.size:]
    # the stat tail is zeroed ONCE per step function: every slot a step uses is overwritten (=, never +=) by its loss kernels on
    # every step, the slots it does not use stay zero (their all-reduce sums zeros) -- one launch less per step
    # (the owner token lives on the shared workspace, not in this function's cache: two step functions used alternately on one engine
    #  -- train / finetune, two loss types -- must not leave each other's slots standing, which a SUM all-reduce would multiply by the
    #  world size on every step)
    if ws.bufs.get('tail_owner') != (id(cache), tail.data_ptr()):
      tail.zero_()
      ws.bufs['tail_owner'] = (id(cache), tail.data_ptr())
    # ---- losses -------------------------------------------------------------------------------------
    if 'coef' not in cache:
      cache['coef'] = torch.tensor([config.data_coarse_loss_mult] * (L - 1) + [config.data_loss_mult],
                                   dtype=torch.float32, device=dev)
    pred = levels[0]['rgb_all']
    d_pred = ws.get('d_pred', (L, N, 3))
    mode, lm = 0, (None if config.disable_multiscale_loss else rays['lossmult'])
    if tt == 'withmask':
      mode, lm = 1, rays['static_mask']
    elif tt == 'robustnerf':
      P = config.patch_size
      if N % (P * P):
        raise ValueError('robustnerf needs whole patches per device')
      if inlier_thresholds is None:
        # device-side feedback of the previous step's (rank-averaged) thresholds: what train.py:145-148 does through
        # the host, without the device->host->device round trip (a synchronisation point in every step)
        thr = cache.get('thr_dev')
        if thr is None:      # (ONE buffer for the life of the step function, updated in place: a captured step reads it by address)
          thr = cache['thr_dev'] = torch.ones((L, 1), dtype=torch.float32, device=dev)          # train.py:130
      else:
        thr = torch.as_tensor(np.asarray(inlier_thresholds, dtype=np.float32) if not torch.is_tensor(inlier_thresholds)
                              else inlier_thresholds).to(device=dev, dtype=torch.float32).reshape(L, -1)[:, :1].contiguous()
      mask = ws.get('robust_mask', (L, N))
      err = ws.get('robust_err', (N,))
      part = ws.get('robust_part', (N // (P * P) * 4,))
      for l in range(L):
        _lib.call('hugs_robust_mask', N // (P * P), P, pred[l], gt, thr[l], config.robustnerf_inlier_quantile,
                  config.robustnerf_smoothed_filter_size, config.robustnerf_smoothed_inlier_quantile,
                  config.robustnerf_inner_patch_size, config.robustnerf_inner_patch_inlier_quantile, mask[l], err, part,
                  tail[o_rob + 5 * l:o_rob + 5 * l + 5])
      mode, lm = 2, mask
    d_mask = None
    if tt == 'hanerf':
      msm = mask_size_mult(train_frac)
      cache['mask_size_mult'] = msm
      d_mask = ws.get('d_mask', (N,))
      hst = ws.get('hanerf_stats', (2 * L + 2,))
      if dyn is not None:      # a step being captured: the weight is read from dyn[4] (step_scalars)
        _lib.call('hugs_hanerf_loss_dyn', N, L, pred, gt, mask_st['mask'], int(config.data_loss_type == 'charb'),
                  config.charb_padding, cache['coef'], dyn[4:5], d_pred, d_mask, hst)
      else:
        _lib.call('hugs_hanerf_loss', N, L, pred, gt, mask_st['mask'], int(config.data_loss_type == 'charb'),
                  config.charb_padding, cache['coef'], msm, d_pred, d_mask, hst)
      copy_many([(hst[:2 * L], tail[0:2 * L]), (hst[2 * L:], tail[o_han:o_han + 2])])
    elif tt == 'nerfw':
      fin_ = levels[-1]
      Mf = N * fin_['S']
      pred_nw = ws.get('pred_nerfw', (L, N, 3))
      # the final level is scored on rgb_combined (train_utils.py:158-159)
      copy_many(([(pred[:L - 1], pred_nw[:L - 1])] if L > 1 else []) + [(fin_['rgb_combined'], pred_nw[L - 1])])
      nw = dict(d_rgb_combined=d_pred[L - 1], d_beta=ws.get('d_beta', (N,)),
                dens_t_const=config.nerfw_density_loss_mult / Mf)
      nst = ws.get('nerfw_stats', (2 * L + 1,))
      _lib.call('hugs_nerfw_loss', N, L, pred_nw, gt, fin_['uncertainty'], int(config.data_loss_type == 'charb'),
                config.charb_padding, cache['coef'], config.nerfw_beta_loss_mult, d_pred, nw['d_beta'], nst)
      copy_many([(nst[:2 * L], tail[0:2 * L]), (nst[2 * L:], tail[o_nw:o_nw + 1])])
      _lib.call('hugs_sum', Mf, fin_['dens_t'], 1.0 / Mf, tail[o_nw + 1:o_nw + 2])
    else:
      _lib.call('hugs_data_loss', N, L, pred, gt, lm, mode, config.withmask_transient_weight,
                int(config.data_loss_type == 'charb'), config.charb_padding, cache['coef'], d_pred, tail[0:2 * L])
    fin = levels[-1]
    Sf = fin['S']
    d_w = [None] * L
    loss_ray = ws.get('loss_ray', (N,))

    def interlevel_losses():
      # (round 5: launched on the proposal stream in front of the proposal levels' backward -- their d_w and one stat each are all
      #  these kernels produce; on the main stream they sat between the data loss and the final level's compositing backward)
      if not is_finetune and config.interlevel_loss_mult > 0:
        lr_ = ws.get('loss_ray_il', (N,))
        for l in range(L - 1):
          d_w[l] = ws.get(f'd_w{l}', (N, levels[l]['S']))
          _lib.call('hugs_interlevel', N, Sf, levels[l]['S'], fin['sdist'], fin['weights'], levels[l]['sdist'],
                    levels[l]['weights'], config.interlevel_loss_mult / (N * Sf), lr_, d_w[l])
          _lib.call('hugs_sum', N, lr_, 1.0 / (N * Sf), tail[o_il + l:o_il + l + 1])
    il_on_prop = _IL_ON_PROP == '1' or (_IL_ON_PROP == 'auto' and N * (model.num_prop_samples * (L - 1) + model.num_nerf_samples) <= _STEP_GRAPH_ROWS)
    if not il_on_prop:
      interlevel_losses()
    if not is_finetune and config.distortion_loss_mult > 0:
      d_w[L - 1] = ws.get(f'd_w{L-1}', (N, Sf))
      _lib.call('hugs_distortion', N, Sf, fin['sdist'], fin['weights'], config.distortion_loss_mult / N, loss_ray, d_w[L - 1])
      _lib.call('hugs_sum', N, loss_ray, 1.0 / N, tail[o_dist:o_dist + 1])
    # ---- backward -----------------------------------------------------------------------------------
    if model.num_glo_features > 0:
      layout.view(grad, ('GloEmbed_0', 'embedding')).zero_()
    if model.num_transient_features > 0:
      layout.view(grad, ('TransientEmbed_0', 'embedding')).zero_()
    if model.nerf_spec.num_tra > 0 and tt != 'nerfw':       # finetune stage of a nerfw model: the branch is not in the loss
      sp = model.nerf_spec
      lo = layout.by_path[('NerfMLP_0', sp.layers[sp.t0]['name'], 'kernel')]['off']
      last = layout.by_path[('NerfMLP_0', sp.layers[-1]['name'], 'bias')]
      grad[lo:last['off'] + int(np.prod(last['pshape']))].zero_()
    if mask_st is not None:
      ev1 = _engine.new_event(); ev1.record(main_s)          # loss gradients and the zeroed embedding rows are ready
      with torch.cuda.stream(side_s):
        _engine.wait_event(side_s, ev1)
        eng.mask_backward(state.flat, grad, mask_st, rays, d_mask)
        ev_mask_bwd = _engine.new_event(); ev_mask_bwd.record(side_s)
    elif model.mask_spec is not None:            # finetune stage of a hanerf model: the mask is not in the loss
      lo = layout.by_path[('ImplicitMask_0', 'Dense_0', 'kernel')]['off']
      last = [lf for lf in layout.leaves if lf['path'][0] == 'ImplicitMask_0'][-1]
      grad[lo:last['off'] + int(np.prod(last['pshape']))].zero_()
    prop_done = False
    prop_lo = layout.by_path[('PropMLP_0', 'Dense_0', 'kernel')]['off']
    last_prop = [lf for lf in layout.leaves if lf['path'][0] == 'PropMLP_0'][-1]
    prop_hi = last_prop['off'] + int(np.prod(last_prop['pshape']))
    # pmean(grad) in buckets (train_utils.py:457-459 is ONE pmean of the whole tree after the backward pass): the
    # NerfMLP holds 96 % of the bytes, and its gradient becomes final layer by layer -- heads first, then trunk layer
    # 7 down to 0, each ~4 MB.  Every bucket's all-reduce (SUM; the 1/world is folded into the clip/Adam kernels) is
    # issued on the stream that produced it the moment it is final, so RCCL runs underneath the remaining dX / dW
    # GEMMs and only the last bucket + the small rest (PropMLP, embeddings, stat tail) is exposed.
    # HUGS_AR_BUCKETS=0 (A/B switch for the first hardware scaling run): no buckets, ONE all-reduce of the whole buffer
    # after the backward pass -- the reference's own structure.
    ar_works, ar_ranges = [], []

    def bucket_done(lo, hi):
      ar_ranges.append((lo, hi))
      ar_works.append(dist.all_reduce(grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def level_backward(l, lane, after=None, before_dw=None):
      nonlocal prop_done
      coef = config.data_loss_mult if l == L - 1 else config.data_coarse_loss_mult
      is_prop = l < L - 1
      if is_prop and d_w[l] is None and coef == 0:
        return          # (always so in the finetune stage -- no interlevel term -- unless a coarse data loss reaches a proposal MLP that renders colour)
      tgt = grad
      if is_prop and prop_done:
        tgt = ws.get('grad_tmp', (layout.size + STAT_TAIL,))
      bucketed = bucket_done if (reduce and world > 1 and not is_prop and not is_finetune and _AR_BUCKETS) else None
      if tt == 'nerfw' and not is_prop:
        eng.backward_level(state.flat, tgt, levels[l], rays, N, None, d_w[l], nerfw=nw, leaf_done=bucketed, lane=lane, after_heads=after,
                           before_dw=before_dw)
      else:
        eng.backward_level(state.flat, tgt, levels[l], rays, N, d_pred[l] if coef != 0 else None, d_w[l], leaf_done=bucketed,
                           lane=lane, after_heads=after, before_dw=before_dw)
      if is_prop and prop_done:
        _lib.call('hugs_add_inplace', prop_hi - prop_lo, tgt[prop_lo:prop_hi], grad[prop_lo:prop_hi])
      if is_prop:
        prop_done = True

    # With stop_level_grad the proposal levels' backward depends on the losses only (interlevel d_w, coarse data loss),
    # not on the NerfMLP's: it is ~0.4 ms of small launch-latency-bound kernels, enqueued on its own pair of streams
    # underneath the NerfMLP trunk backward (whose GEMMs fill the chip; the proposal kernels fit in their tails).
    bwd_main = torch.cuda.current_stream()
    prop_stream = eng._side_stream(1)
    box = {}

    def launch_prop():
      ev_loss = _engine.new_event(); ev_loss.record(bwd_main)
      with torch.cuda.stream(prop_stream):
        _engine.wait_event(prop_stream, ev_loss)
        if il_on_prop:
          interlevel_losses()
        for l in range(L - 2, -1, -1):
          level_backward(l, 2)
        if not prop_done:
          grad[prop_lo:prop_hi].zero_()
        box['ev'] = _engine.new_event(); box['ev'].record(prop_stream)
    prop_events = lambda: [box['ev']] if 'ev' in box else []
    if _engine._SIDE_LATE:
      level_backward(L - 1, 0, after=launch_prop, before_dw=prop_events)
    else:
      launch_prop()
      level_backward(L - 1, 0, before_dw=prop_events)
    _engine.wait_event(bwd_main, box['ev'])
    if ev_mask_bwd is not None:
      _engine.wait_event(torch.cuda.current_stream(), ev_mask_bwd)
    # ---- pmean(grad), pmean(stats) ------------------------------------------------------------------
    if world > 1 and reduce:
      # whatever no bucket covered (PropMLP, embeddings, ImplicitMask, the stat tail; everything in the finetune stage)
      if AR_PROFILE is not None:
        e0_ = torch.cuda.Event(enable_timing=True); e0_.record()
      for lo, hi in uncovered_ranges(layout, ar_ranges, grad.numel()):
        ar_works.append(dist.all_reduce(grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
      for w in ar_works:
        w.wait()
      if AR_PROFILE is not None:
        e1_ = torch.cuda.Event(enable_timing=True); e1_.record()
        AR_PROFILE.append((e0_, e1_))
    return rng

  def step_finish(state, dyn, ent=None):
    """The second part: weight decay, stats / clip / Adam / re-cast, stat packing.  Returns the packed stats buffer.
    ent: the graph entry of a step being captured -- the optimizer's last launch then also publishes the results (packed stat
    tail, RobustNeRF threshold feedback, the stats to a pinned host slot, the advanced key: hugs_opt_adam_pub) instead of three
    copies behind the graph."""
    eng = model.engine(state.flat.device)
    ws = eng.ws
    world = _world()
    grad = ws.get('grad', (layout.size + STAT_TAIL,))
    tail = grad[layout.size:]
    gscale = 1.0 / world
    for off, n_, mult in decay:        # after pmean; the kernels below scale the buffer by gscale, hence the 1/gscale
      _lib.call('hugs_axpy', n_, 2.0 * mult / gscale, state.flat[off:off + n_], grad[off:off + n_])
    nleaf = len(layout.leaves)
    packed = ws.get('stats_packed', (STAT_TAIL + nleaf * 6 + 16,))
    if tt == 'robustnerf' and 'thr_dev' not in cache:
      cache['thr_dev'] = torch.ones((L, 1), dtype=torch.float32, device=packed.device)
    if ent is not None and dyn is not None:
      pub = ent['pub'] = np.zeros(12, np.uint64)
      pub[0], pub[1], pub[2], pub[3] = tail.data_ptr(), packed.data_ptr(), STAT_TAIL, packed.numel()
      pub[4] = int(np.float32(gscale).view(np.uint32))
      if tt == 'robustnerf':
        pub[5], pub[6], pub[7], pub[8] = cache['thr_dev'].data_ptr(), o_rob, 5, L
      if ent.get('key_out') is not None:
        pub[9], pub[10] = ent['key_out'].data_ptr(), ent['key'].data_ptr()
      pub[11] = ent['ptrs'].data_ptr()
      leaf_stats = optimizer_step(state, grad, gscale, dyn, pub)
      assert leaf_stats.data_ptr() == packed[STAT_TAIL:].data_ptr()
      return packed
    leaf_stats = optimizer_step(state, grad, gscale, dyn)
    # ---- stats (lazy) -------------------------------------------------------------------------------
    packed[:STAT_TAIL].copy_(tail)
    if world > 1:
      _lib.call('hugs_affine', STAT_TAIL, packed, gscale, 0.0, packed)
    assert leaf_stats.data_ptr() == packed[STAT_TAIL:].data_ptr()
    if tt == 'robustnerf':
      cache['thr_dev'].copy_(packed[o_rob:o_rob + 5 * L].reshape(L, 5)[:, :1])
    return packed

  def stats_builder(state):
    nleaf = len(layout.leaves)
    msm_now = cache.get('mask_size_mult', 0.0)

    def build(hst):
      tl = hst[:STAT_TAIL]
      ls = hst[STAT_TAIL:STAT_TAIL + nleaf * 4].reshape(nleaf, 4)
      lu = hst[STAT_TAIL + nleaf * 4:STAT_TAIL + nleaf * 6].reshape(nleaf, 2)
      T = lambda x: torch.tensor(np.asarray(x, dtype=np.float32))
      stats = {}
      mses = tl[0:2 * L:2]
      dls = tl[1:2 * L:2]
      losses = {'data': float(config.data_coarse_loss_mult * dls[:-1].sum() + config.data_loss_mult * dls[-1])}
      if not is_finetune and config.interlevel_loss_mult > 0:
        losses['interlevel'] = float(config.interlevel_loss_mult * tl[o_il:o_il + L - 1].sum())
      if not is_finetune and config.distortion_loss_mult > 0:
        losses['distortion'] = float(config.distortion_loss_mult * tl[o_dist])
      if decay:
        wl2 = _summarize(layout, ls[:, 2], sum)
        losses['weight'] = float(sum(float(m) * wl2[k] for k, m in dict(config.weight_decay_mults).items()))
      if tt == 'nerfw':
        losses['beta'] = float(config.nerfw_beta_loss_mult * tl[o_nw] + config.nerfw_beta_loss_bias)
        losses['density'] = float(config.nerfw_density_loss_mult * tl[o_nw + 1])
      if tt == 'hanerf':
        losses['mask_size'] = float(msm_now * tl[o_han])
        stats['implicit_mask'] = T([tl[o_han + 1]])
      stats['losses'] = {k: T(v) for k, v in losses.items()}
      stats['loss'] = T(sum(losses.values()))
      stats['mses'] = T(mses)
      stats['psnrs'] = image.mse_to_psnr(stats['mses'])
      stats['psnr'] = stats['psnrs'][-1]
      if tt == 'robustnerf':
        r = tl[o_rob:o_rob + 5 * L].reshape(L, 5)
        for i, k in enumerate(['inlier_threshold', 'is_inlier_loss', 'has_inlier_neighbors', 'is_inlier_patch', 'mask']):
          stats['robust_' + k] = T(r[:, i])
      stats['weight_l2s'] = {k: T(v) for k, v in _summarize(layout, ls[:, 2], sum).items()}
      stats['grad_norms'] = {k: T(v) for k, v in _summarize(layout, ls[:, 0], sum, math.sqrt).items()}
      stats['grad_maxes'] = {k: T(v) for k, v in _summarize(layout, ls[:, 1], max).items()}
      stats['opt_update_norms'] = {k: T(v) for k, v in _summarize(layout, lu[:, 0], sum, math.sqrt).items()}
      stats['opt_update_maxes'] = {k: T(v) for k, v in _summarize(layout, lu[:, 1], max).items()}
      return stats

    return build

  train_step.optimizer_step = optimizer_step
  train_step.graph_active = lambda: any('graph' in e for e in graphs.values())
  return train_step


def create_render_fn(model, config):
  """Creates the full-image render function (train_utils.py:555-576):
  render_eval_pfn(variables, train_frac, _, rays) -> (renderings, ray_history) with the reference's leading
  device axis (size = world size after the all-gather)."""

  def render_eval_fn(variables, train_frac, _, rays):
    lead = rays.origins.shape[:2]
    flat_rays = rays.map(lambda r: r.reshape((-1, r.shape[-1])))
    rend, hist = model.apply(variables, None, flat_rays, train_frac=train_frac, compute_extras=True,
                             zero_glo=config.enable_render_zero_glo, zero_tra=config.enable_render_zero_tra)
    world = _world()

    def gather(v):
      if world == 1:
        return v[None]
      out = [torch.empty_like(v) for _ in range(world)]
      dist.all_gather(out, v.contiguous())
      return torch.stack(out)

    return [{k: gather(v) for k, v in r.items()} for r in rend], [{k: gather(v) for k, v in h.items()} for h in hist]

  return render_eval_fn


def setup_model(config, rng, compute_dtype=None, device='cuda'):
  """Creates NeRF model, optimizer, and train/render functions (train_utils.py:579-596)."""
  model, variables = models.construct_model(rng, utils.dummy_rays(), config, compute_dtype=compute_dtype, device=device)
  state, lr_fn = create_optimizer(config, variables, model)
  render_eval_pfn = create_render_fn(model, config)
  train_pstep = create_train_step(model, config, False)
  return model, state, render_eval_pfn, train_pstep, lr_fn


def setup_finetune_model(config, model, state):
  """train_utils.py:599-608."""
  new_state, lr_fn = create_finetune_optimizer(config, state.flat, model)
  train_pstep = create_train_step(model, config, True)
  return new_state, train_pstep, lr_fn

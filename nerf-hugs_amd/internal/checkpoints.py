"""Checkpoint interop with the reference's `flax.training.checkpoints` files (train.py:121,232-236): a flax
msgpack state dict of `TrainState{step, params, opt_state}`.  flax is not installed here, so this
follows flax.serialization's documented wire format (msgpack with ExtType 1 = ndarray packed as
(shape, dtype.name, C-order bytes), ExtType 3 = numpy scalar).  Tested by a round trip AND against an independent
msgpack encoder / decoder written from the msgpack spec (tests/test_cpu_checkpoint_bytes.py); a file written by real
flax has still never been read (none is available offline): "parity unpinned" in that sense.

`step` and the optimizer counts are written as 0-d int32 ARRAYS (ext 1) -- what flax emits for the jax scalars a TrainState
holds -- not as numpy scalars (ext 3); the reader takes either, plus flax's chunked form of arrays above 2**30 bytes and
bfloat16 leaves.  tests/flax_wire.py restates flax.serialization's rules (to_state_dict of tuples / NamedTuples / the
TrainState dataclass, _ndarray_to_bytes, _chunk) independently of this file and tests/test_cpu_checkpoint_bytes.py feeds its
output to restore_checkpoint.

Layout written / read:
  {'step': i32, 'params': {'params': {module: {Dense_i: {'kernel','bias'}}, 'GloEmbed_0': {'embedding'}}},
   'opt_state': {'0': {'count': i32, 'mu': {'params': ...}, 'nu': {'params': ...}}, '1': {'count': i32}}}
(optax.adam = chain(scale_by_adam, scale_by_schedule); tuples serialise as dicts with '0','1',... keys; mu / nu
mirror TrainState.params, i.e. they carry the same outer 'params' level -- flax's restore_checkpoint(dir, state)
matches field names against the target state, so a file without that level does not load in the reference).
The reader accepts both forms.

Finetune stage (train.py:97-109 -> `<checkpoint_dir>/finetune`, train_utils.py:515-552: optax.multi_transform over
{'trainable': adam, 'frozen': set_to_zero}):
  'opt_state': {'inner_states': {'trainable': {'inner_state': {'0': {'count','mu','nu'}, '1': {'count'}}},
                                 'frozen': {'inner_state': {}}}}
where mu / nu hold arrays at the 'embedding' leaves and an EMPTY dict at every other leaf (optax's MaskedNode is a
field-less NamedTuple; flax.serialization.to_state_dict of a NamedTuple is the dict of its fields).  optax is
un-vendored and not installed: this restates its documented state classes, parity unpinned."""
import os
import re

import msgpack
import numpy as np
import torch

_EXT_NDARRAY, _EXT_NPSCALAR = 1, 3


def _pack_ext(x):
  if isinstance(x, torch.Tensor):
    x = x.detach().cpu().numpy()
  if isinstance(x, np.ndarray):
    return msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(x.shape), x.dtype.name, x.tobytes('C')), use_bin_type=True))
  if isinstance(x, np.generic):
    return msgpack.ExtType(_EXT_NPSCALAR, msgpack.packb(((), x.dtype.name, x.tobytes()), use_bin_type=True))
  raise TypeError(type(x))


def _unpack_ext(code, data):
  if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
    shape, dtype, buf = msgpack.unpackb(data, raw=True)      # (flax reads the triple raw: the dtype name arrives as bytes)
    dtype = dtype.decode() if isinstance(dtype, bytes) else dtype
    if dtype == 'bfloat16':      # flax maps the name to jax.numpy.bfloat16; here: widen to float32 (exact)
      u = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
      arr = u.view(np.float32).reshape(shape)
    else:
      arr = np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape)
    return arr if code == _EXT_NDARRAY else arr[()]
  return msgpack.ExtType(code, data)


def _unchunk(tree):
  """flax.serialization._unchunk_array_leaves_in_place: an array above 2**30 bytes is written as
  {'__msgpack_chunked_array__': True, 'shape': {'0': d0, ...}, 'chunks': {'0': flat piece, ...}}."""
  if isinstance(tree, dict):
    if '__msgpack_chunked_array__' in tree:
      order = lambda d: [d[str(i)] for i in range(len(d))]
      return np.concatenate([np.asarray(c).reshape(-1) for c in order(tree['chunks'])]).reshape(tuple(int(x) for x in order(tree['shape'])))
    return {k: _unchunk(v) for k, v in tree.items()}
  return tree


_MAX_CHUNK = 2 ** 30


def _chunk(tree, max_chunk):
  """flax.serialization._chunk_array_leaves_in_place (the writer side of _unchunk)."""
  if isinstance(tree, dict):
    return {k: _chunk(v, max_chunk) for k, v in tree.items()}
  if isinstance(tree, np.ndarray) and tree.size * tree.dtype.itemsize > max_chunk:
    n = max(1, int(max_chunk / tree.dtype.itemsize))
    flat = tree.reshape(-1)
    return {'__msgpack_chunked_array__': True, 'shape': {str(i): int(d) for i, d in enumerate(tree.shape)},
            'chunks': {str(i): flat[s_:s_ + n] for i, s_ in enumerate(range(0, flat.size, n))}}
  return tree


def to_bytes(tree, max_chunk=_MAX_CHUNK):
  return msgpack.packb(_chunk(tree, max_chunk), default=_pack_ext, strict_types=True, use_bin_type=True)


def from_bytes(b):
  return _unchunk(msgpack.unpackb(b, ext_hook=_unpack_ext, raw=False, strict_map_key=False))


def _tree_np(model, flat):
  t = model.variables(flat)
  conv = lambda d: {k: (conv(v) if isinstance(v, dict) else v.detach().cpu().numpy().copy()) for k, v in d.items()}
  return conv(dict(t))


def _mask_tree(tree, keep):
  """optax.masked over a parameter tree: leaves whose path fails `keep` become MaskedNode() == {} on the wire."""
  def walk(d, path):
    return {k: (walk(v, path + (k,)) if isinstance(v, dict) else (v if keep(path + (k,)) else {})) for k, v in d.items()}
  return walk(tree, ())


def state_dict(state):
  model = state.model
  if getattr(state, 'hyper', {}).get('finetune'):      # finetune stage: optax.multi_transform state
    keep = lambda path: 'embedding' in path
    adam = {'0': {'count': np.array(state.step, np.int32), 'mu': _mask_tree(_tree_np(model, state.m), keep),
                  'nu': _mask_tree(_tree_np(model, state.v), keep)},
            '1': {'count': np.array(state.step, np.int32)}}
    return {'step': np.array(state.step, np.int32), 'params': _tree_np(model, state.flat),
            'opt_state': {'inner_states': {'trainable': {'inner_state': adam}, 'frozen': {'inner_state': {}}}}}
  return {'step': np.array(state.step, np.int32), 'params': _tree_np(model, state.flat),
          'opt_state': {'0': {'count': np.array(state.step, np.int32), 'mu': _tree_np(model, state.m),
                              'nu': _tree_np(model, state.v)},
                        '1': {'count': np.array(state.step, np.int32)}}}


def save_checkpoint(ckpt_dir, state, step, keep=100):
  """flax.training.checkpoints.save_checkpoint(dir, state, step, keep) (train.py:232-236)."""
  os.makedirs(ckpt_dir, exist_ok=True)
  path = os.path.join(ckpt_dir, f'checkpoint_{int(step)}')
  with open(path + '.tmp', 'wb') as f:
    f.write(to_bytes(state_dict(state)))
  os.replace(path + '.tmp', path)
  olds = sorted((int(m.group(1)) for m in (re.fullmatch(r'checkpoint_(\d+)', n) for n in os.listdir(ckpt_dir)) if m))
  for s in olds[:-keep]:
    os.remove(os.path.join(ckpt_dir, f'checkpoint_{s}'))
  return path


def latest_checkpoint(ckpt_dir):
  if not ckpt_dir or not os.path.isdir(ckpt_dir):
    return None
  steps = [int(m.group(1)) for m in (re.fullmatch(r'checkpoint_(\d+)', n) for n in os.listdir(ckpt_dir)) if m]
  return os.path.join(ckpt_dir, f'checkpoint_{max(steps)}') if steps else None


def restore_checkpoint(ckpt_dir, state):
  """flax.training.checkpoints.restore_checkpoint(dir, state) (train.py:121): returns `state` unchanged when there
  is no checkpoint, otherwise loads step / params / Adam moments in place."""
  path = latest_checkpoint(ckpt_dir) if os.path.isdir(ckpt_dir or '') else ckpt_dir
  if not path or not os.path.exists(path):
    return state
  with open(path, 'rb') as f:
    d = from_bytes(f.read())
  model = state.model
  model.load_variables(state.flat, d['params'])
  opt = d['opt_state']
  if 'inner_states' in opt:        # finetune stage: moments exist at the trainable (embedding) leaves only
    adam = opt['inner_states']['trainable']['inner_state']['0']
    for buf, tree in ((state.m, adam['mu']), (state.v, adam['nu'])):
      tree = tree.get('params', tree)
      buf.zero_()                  # frozen leaves carry no moments (MaskedNode): a reused state must not keep stale ones
      for lf in model.layout.leaves:
        leaf = tree
        for k in lf['path']:
          if not isinstance(leaf, dict) or k not in leaf:
            raise ValueError(f'checkpoint {path}: optimizer state has no entry {"/".join(lf["path"])}')
          leaf = leaf[k]
        if not isinstance(leaf, dict):
          arr = np.array(leaf)
          if tuple(arr.shape) != tuple(lf['shape']):
            raise ValueError(f'checkpoint {path}: moment {"/".join(lf["path"])} has shape {arr.shape}, the model wants {lf["shape"]}')
          model.layout.view(buf, lf['path']).copy_(torch.from_numpy(arr).to(buf.device))
  else:
    adam = opt['0'] if '0' in opt else opt[0]
    model.load_variables(state.m, {'params': adam['mu'].get('params', adam['mu'])})
    model.load_variables(state.v, {'params': adam['nu'].get('params', adam['nu'])})
  state.step = int(np.asarray(d['step']))
  if getattr(model, '_engine', None) is not None:      # the compute-dtype weight copies follow the restored masters
    model._engine.refresh_weights(state.flat, owner=state)
  return state

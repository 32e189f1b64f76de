"""TEST-SIDE restatement of flax.serialization's wire rules (flax is not installed; restated from its published source,
flax/serialization.py -- `to_state_dict`, `_ndarray_to_bytes`, `_msgpack_ext_pack`, `_chunk`, `msgpack_serialize` -- and
optax's state classes as the NamedTuples they are).  NOT the product's encoder: it shares no code with
nerf_hugs_amd/internal/checkpoints.py and emits bytes through the hand-written msgpack encoder of
tests/test_cpu_checkpoint_bytes.py.  Parity unpinned in the strict sense (no flax-written file exists offline); this pins
the READER against every construct the published writer can emit for the reference's TrainState (train.py:121,232-236).

Rules restated:
  * to_state_dict: dict -> dict of its items; list / tuple -> {'0': .., '1': ..} (`_tuple_to_dict`: keys are str(index));
    NamedTuple -> {field: to_state_dict(value)} (`_namedtuple_state_dict`), so a field-less one (optax MaskedNode,
    EmptyState) becomes {}; a flax struct dataclass -> its pytree-node fields only (TrainState: step, params, opt_state --
    apply_fn / tx are static); arrays and scalars are leaves.
  * leaves: np.ndarray / jax.Array -> ExtType(1, msgpack((shape, dtype.name, arr.tobytes('C')))) -- a jax scalar (TrainState.step,
    the optimizer counts) is a 0-d ARRAY and so also goes out as ext 1; a numpy scalar type (np.int32(3)) -> ExtType(3, same
    triple of np.asarray(x)); Python ints / floats / bools stay native msgpack.
  * arrays above MAX_CHUNK_SIZE = 2**30 bytes -> {'__msgpack_chunked_array__': True, 'shape': {'0': d0, ..},
    'chunks': {'0': flat[0:n], '1': ..}} with n = max(1, MAX_CHUNK_SIZE // itemsize) elements per chunk (`_chunk`).
"""
import collections
import struct

import numpy as np

from tests.test_cpu_checkpoint_bytes import enc as _enc_basic

MAX_CHUNK_SIZE = 2 ** 30


# ---- optax state classes (NamedTuples, as published) ------------------------------------------------------------------
ScaleByAdamState = collections.namedtuple('ScaleByAdamState', ['count', 'mu', 'nu'])
ScaleByScheduleState = collections.namedtuple('ScaleByScheduleState', ['count'])
EmptyState = collections.namedtuple('EmptyState', [])
MaskedNode = collections.namedtuple('MaskedNode', [])
MaskedState = collections.namedtuple('MaskedState', ['inner_state'])
MultiTransformState = collections.namedtuple('MultiTransformState', ['inner_states'])   # optax: PartitionState in newer releases, same field


class TrainState:
  """flax.training.train_state.TrainState: pytree-node fields step / params / opt_state; apply_fn and tx are static."""
  node_fields = ('step', 'params', 'opt_state')

  def __init__(self, step, params, opt_state):
    self.step, self.params, self.opt_state = step, params, opt_state


class NpScalar:
  """marks a value that flax would see as a numpy scalar type (ext 3) rather than a 0-d array (ext 1)"""

  def __init__(self, v):
    self.v = v


def to_state_dict(x):
  if isinstance(x, TrainState):
    return {k: to_state_dict(getattr(x, k)) for k in x.node_fields}
  if isinstance(x, tuple) and hasattr(x, '_fields'):
    return {k: to_state_dict(getattr(x, k)) for k in x._fields}
  if isinstance(x, (list, tuple)):
    return {str(i): to_state_dict(v) for i, v in enumerate(x)}
  if isinstance(x, dict):
    return {k: to_state_dict(v) for k, v in x.items()}
  return x


def _ndarray_to_bytes(arr):
  return _enc_basic([list(arr.shape), arr.dtype.name, arr.tobytes('C')])


def _ext(code, payload):
  n = len(payload)
  if n < 256:
    return b'\xc7' + bytes([n, code]) + payload          # ext 8
  if n < 65536:
    return b'\xc8' + struct.pack('>H', n) + bytes([code]) + payload
  return b'\xc9' + struct.pack('>I', n) + bytes([code]) + payload


def _chunk(arr, max_chunk):
  chunksize = max(1, int(max_chunk / arr.dtype.itemsize))
  flat = arr.reshape(-1)
  return {'__msgpack_chunked_array__': True, 'shape': {str(i): int(d) for i, d in enumerate(arr.shape)},
          'chunks': {str(i): flat[s:s + chunksize] for i, s in enumerate(range(0, flat.size, chunksize))}}


def msgpack_serialize(state_dict, max_chunk=MAX_CHUNK_SIZE):
  def pack(o):
    if isinstance(o, dict):
      n = len(o)
      head = bytes([0x80 | n]) if n < 16 else (b'\xde' + struct.pack('>H', n) if n < 65536 else b'\xdf' + struct.pack('>I', n))
      return head + b''.join(_enc_basic(k) + pack(v) for k, v in o.items())
    if isinstance(o, np.ndarray):
      if o.size * o.dtype.itemsize > max_chunk:
        return pack(_chunk(o, max_chunk))
      return _ext(1, _ndarray_to_bytes(o))
    if isinstance(o, NpScalar):
      return _ext(3, _ndarray_to_bytes(np.asarray(o.v)))
    if isinstance(o, float):
      return b'\xcb' + struct.pack('>d', o)
    return _enc_basic(o)
  return pack(state_dict)


def adam_train_state(step, params, mu, nu, counts_as=np.array):
  """The reference's TrainState after `step` updates: tx = optax.adam(lr_fn) = chain(scale_by_adam, scale_by_schedule)
  (train_utils.py:510); counts / step are 0-d int32 arrays (jax scalars) unless counts_as says otherwise."""
  c = lambda: counts_as(step, np.int32) if counts_as is np.array else counts_as(step)
  return TrainState(c(), params, (ScaleByAdamState(c(), mu, nu), ScaleByScheduleState(c())))


def finetune_train_state(step, params, mu, nu, is_trainable):
  """train_utils.py:538-544: optax.multi_transform({'trainable': adam, 'frozen': set_to_zero()}, partitions): each inner
  transform is wrapped in optax.masked -> MaskedState(inner_state); masked-out leaves of mu / nu are MaskedNode()."""
  def mask(tree, path=()):
    return {k: (mask(v, path + (k,)) if isinstance(v, dict) else (v if is_trainable(path + (k,)) else MaskedNode())) for k, v in tree.items()}
  c = lambda: np.array(step, np.int32)
  adam = (ScaleByAdamState(c(), mask(mu), mask(nu)), ScaleByScheduleState(c()))
  return TrainState(c(), params, MultiTransformState({'trainable': MaskedState(adam), 'frozen': MaskedState(EmptyState())}))

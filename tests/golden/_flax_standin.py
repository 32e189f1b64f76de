"""Stand-ins for `flax`, `gin`, `optax`, `absl`, `dm_pix`, `cv2` and the rest of `jax` that
MipNeRF360/internal/{models,train_utils,configs,utils,image}.py import at module level.

TEST INFRASTRUCTURE ONLY (build container only).  Purpose: let `gen_model_fixtures.py` IMPORT AND
EXECUTE THE REFERENCE'S OWN `models.py` / `train_utils.py` -- `Model.__call__`, `MLP.__call__`,
`ImplicitMask`, `compute_*_loss`, `robustnerf_mask`, `interlevel_loss`, `distortion_loss`,
`clip_gradients` and the body of `train_step` -- on numpy arrays, and record input/output vectors.
Nothing in the product or in the GPU-side tests imports this file.

What is the reference's and what is the stand-in's:
  * every line of models.py / train_utils.py runs unmodified;
  * `flax.linen.Module/@compact/Dense/Embed` here only do bookkeeping: auto-naming (`Dense_3`,
    `NerfMLP_0`, per-class counters restarted on every call of a compact method, as flax does),
    explicit parameter trees, `Dense = x @ kernel + bias`, `Embed = embedding[idx]`;
  * `jax.value_and_grad` cannot be provided by numpy: the stand-in evaluates the loss with the
    reference's own `loss_fn` and takes the gradient tree from a hook the generator sets (the
    generator also calls the captured `loss_fn` itself for float64 central differences with
    `stop_gradient` values replayed from the unperturbed run);
  * `optax.adam` is restated from its documentation (bias-corrected moments, eps outside the sqrt,
    schedule evaluated at the pre-increment count) -- third-party, un-versioned: NOT a pin.
"""
import dataclasses
import sys
import types

import numpy as np

import _jax_standin

f32 = np.float32

# ---------------------------------------------------------------------------------------------
# gin
# ---------------------------------------------------------------------------------------------
BINDINGS = {}          # {'NerfMLP': {'net_width': 64, ...}, 'Config': {...}}


def set_bindings(b):
  BINDINGS.clear()
  for k, v in b.items():
    BINDINGS[k] = dict(v)


def _configurable(obj=None, **_kw):
  def deco(cls):
    if isinstance(cls, type):
      orig = cls.__init__
      name = cls.__name__

      def __init__(self, *a, **k):
        merged = dict(BINDINGS.get(name, {}))
        merged.update(k)
        orig(self, *a, **merged)

      cls.__init__ = __init__
    return cls
  if obj is None or isinstance(obj, str):
    return deco
  return deco(obj)


def _build_gin():
  gin = types.ModuleType('gin')
  gin.configurable = _configurable
  gin.config = types.ModuleType('gin.config')
  gin.config.external_configurable = lambda fn, module=None, name=None: fn
  gin.parse_config_files_and_bindings = lambda *a, **k: None
  gin.config_str = lambda: ''
  gin.add_config_file_search_path = lambda *a, **k: None
  return gin


# ---------------------------------------------------------------------------------------------
# pytrees (dict / list / tuple / struct dataclasses)
# ---------------------------------------------------------------------------------------------
def _is_struct(x):
  return dataclasses.is_dataclass(x) and not isinstance(x, type) and getattr(type(x), '_pytree', False)


def _node_fields(x):
  return [f.name for f in dataclasses.fields(x) if f.metadata.get('pytree_node', True)]


def tree_map(fn, tree, *rest):
  if isinstance(tree, dict):
    return type(tree)({k: tree_map(fn, v, *[r[k] for r in rest]) for k, v in tree.items()})
  if isinstance(tree, (list, tuple)) and not hasattr(tree, '_fields'):
    return type(tree)(tree_map(fn, v, *[r[i] for r in rest]) for i, v in enumerate(tree))
  if _is_struct(tree):
    upd = {n: tree_map(fn, getattr(tree, n), *[getattr(r, n) for r in rest]) for n in _node_fields(tree)}
    return dataclasses.replace(tree, **upd)
  if tree is None:
    return None
  return fn(tree, *rest)


def tree_leaves(tree):
  out = []
  tree_map(lambda x: out.append(x), tree)
  return out


def tree_reduce(fn, tree, initializer=None):
  acc = initializer
  for leaf in tree_leaves(tree):
    acc = leaf if acc is None else fn(acc, leaf)
  return acc


def struct_dataclass(cls):
  cls = dataclasses.dataclass(frozen=True)(cls)
  cls._pytree = True
  cls.replace = lambda self, **kw: dataclasses.replace(self, **kw)
  return cls


def struct_field(pytree_node=True, **kw):
  return dataclasses.field(metadata={'pytree_node': pytree_node}, **kw)


# ---------------------------------------------------------------------------------------------
# flax.linen
# ---------------------------------------------------------------------------------------------
_STACK = []            # modules whose compact method is executing
INIT = {'rng': None}   # numpy Generator while Module.init runs, else None


class Module:
  """Bookkeeping-only linen.Module: dataclass-style fields from class annotations, auto-named
  children, explicit parameter dict."""

  def __init_subclass__(cls, **kw):
    super().__init_subclass__(**kw)
    fields = []
    for klass in reversed(cls.__mro__):
      for n in getattr(klass, '__annotations__', {}):
        if n not in fields and not n.startswith('_'):
          fields.append(n)
    cls._fields = fields

  def __init__(self, *args, **kwargs):
    names = type(self)._fields
    if len(args) > len(names):
      raise TypeError('too many positional arguments')
    vals = dict(zip(names, args))
    for k, v in kwargs.items():
      if k in ('name', 'parent'):
        continue
      if k not in names:
        raise TypeError(f'{type(self).__name__} has no field {k}')
      vals[k] = v
    for n in names:
      if n in vals:
        object.__setattr__(self, n, vals[n])
      elif hasattr(type(self), n):             # instance attribute, so a function default is not bound
        object.__setattr__(self, n, type(self).__dict__[n] if n in type(self).__dict__ else
                           next(k.__dict__[n] for k in type(self).__mro__ if n in k.__dict__))
      else:
        raise TypeError(f'{type(self).__name__} missing field {n}')
    self._params = None
    self._setup_done = False
    self._counters = {}
    self._name = kwargs.get('name')
    if _STACK:                                  # constructed inside a parent's compact method
      parent = _STACK[-1]
      cname = type(self).__name__
      if self._name is None:
        i = parent._counters.get(cname, 0)
        parent._counters[cname] = i + 1
        self._name = f'{cname}_{i}'
      if INIT['rng'] is not None:
        self._params = parent._params.setdefault(self._name, {})
      else:
        self._params = parent._params[self._name]

  def setup(self):
    pass

  def _enter(self):
    if not self._setup_done:
      self._setup_done = True
      self.setup()
    self._counters = {}
    _STACK.append(self)

  def param(self, name, init_fn, *shape_args):
    if INIT['rng'] is not None and name not in self._params:
      self._params[name] = init_fn(INIT['rng'], *shape_args)
    return self._params[name]

  def apply(self, variables, /, *args, **kwargs):
    assert not _STACK
    self._params = variables['params']
    return self(*args, **kwargs)

  def init(self, init_rng, /, *args, **kwargs):
    assert not _STACK
    INIT['rng'] = init_rng.rng if hasattr(init_rng, 'rng') else np.random.default_rng(init_rng)
    self._params = {}
    try:
      self(*args, **kwargs)
    finally:
      INIT['rng'] = None
    return {'params': self._params}


def compact(fn):
  def wrapped(self, *a, **k):
    self._enter()
    try:
      return fn(self, *a, **k)
    finally:
      _STACK.pop()
  wrapped.__name__ = fn.__name__
  return wrapped


def he_uniform():
  # jax.nn.initializers.he_uniform: variance_scaling(2, 'fan_in', 'uniform') -> U(-sqrt(6/fan_in), +)
  def init(rng, shape, dtype=f32):
    lim = np.sqrt(6.0 / shape[0])
    return rng.uniform(-lim, lim, shape).astype(dtype)
  return init


class Dense(Module):
  features: int
  use_bias: bool = True
  kernel_init: object = None

  @compact
  def __call__(self, x):
    x = np.asarray(x)
    kinit = self.kernel_init or he_uniform()
    kernel = self.param('kernel', kinit, (x.shape[-1], self.features))
    bias = self.param('bias', lambda rng, shape: np.zeros(shape, f32), (self.features,))
    if kernel.shape != (x.shape[-1], self.features):
      raise ValueError(f'{self._name}: kernel {kernel.shape} vs input {x.shape}')
    # flax: lax.dot_general(x, kernel, contracting last/first) + bias
    return np.matmul(x, kernel) + bias


class Embed(Module):
  num_embeddings: int
  features: int

  @compact
  def __call__(self, inputs):
    inputs = np.asarray(inputs)
    if not np.issubdtype(inputs.dtype, np.integer):
      raise ValueError('Input type must be an integer or unsigned integer.')   # flax Embed
    # flax default_embed_init = variance_scaling(1.0, 'fan_in', 'normal', out_axis=0)
    emb = self.param('embedding',
                     lambda rng, shape: (rng.normal(size=shape) / np.sqrt(shape[1])).astype(f32),
                     (self.num_embeddings, self.features))
    return np.take(emb, inputs, axis=0)


def _relu(x):
  return np.maximum(x, 0)


def _softplus(x):
  return np.logaddexp(x, 0)          # jax.nn.softplus


def _sigmoid(x):
  return 1 / (1 + np.exp(-x))        # jax.nn.sigmoid = lax.logistic


def _build_flax():
  flax = types.ModuleType('flax')
  linen = types.ModuleType('flax.linen')
  linen.Module, linen.compact, linen.Dense, linen.Embed = Module, compact, Dense, Embed
  linen.relu, linen.softplus, linen.sigmoid = _relu, _softplus, _sigmoid
  flax.linen = linen
  struct = types.ModuleType('flax.struct')
  struct.dataclass, struct.field = struct_dataclass, struct_field
  flax.struct = struct
  core = types.ModuleType('flax.core')

  import collections.abc

  class FrozenDict(collections.abc.Mapping):
    def __init__(self, d=None):
      self._d = dict(d or {})

    def __class_getitem__(cls, item):
      return cls

    def __getitem__(self, k):
      return self._d[k]

    def __iter__(self):
      return iter(self._d)

    def __len__(self):
      return len(self._d)

    def __hash__(self):
      return id(self)

  core.FrozenDict = FrozenDict
  core.freeze = lambda x: x
  scope = types.ModuleType('flax.core.scope')
  scope.FrozenVariableDict = dict
  core.scope = scope
  flax.core = core
  training = types.ModuleType('flax.training')
  train_state = types.ModuleType('flax.training.train_state')

  @struct_dataclass
  class TrainState:
    step: object
    apply_fn: object = struct_field(pytree_node=False, default=None)
    params: object = None
    tx: object = struct_field(pytree_node=False, default=None)
    opt_state: object = None

    def apply_gradients(self, *, grads, **kw):
      updates, new_opt = self.tx.update(grads, self.opt_state, self.params)
      new_params = tree_map(lambda p, u: (p + u).astype(p.dtype), self.params, updates)
      return self.replace(step=self.step + 1, params=new_params, opt_state=new_opt)

    @classmethod
    def create(cls, *, apply_fn, params, tx, **kw):
      return cls(step=0, apply_fn=apply_fn, params=params, tx=tx, opt_state=tx.init(params))

  train_state.TrainState = TrainState
  training.train_state = train_state
  flax.training = training
  traverse_util = types.ModuleType('flax.traverse_util')

  def path_aware_map(fn, tree, path=()):
    if isinstance(tree, dict):
      return {k: path_aware_map(fn, v, path + (k,)) for k, v in tree.items()}
    return fn(path, tree)

  traverse_util.path_aware_map = path_aware_map
  flax.traverse_util = traverse_util
  mods = {'flax': flax, 'flax.linen': linen, 'flax.struct': struct, 'flax.core': core,
          'flax.core.scope': scope, 'flax.training': training,
          'flax.training.train_state': train_state, 'flax.traverse_util': traverse_util}
  return mods


# ---------------------------------------------------------------------------------------------
# optax (restated from documentation; third-party => NOT a pin)
# ---------------------------------------------------------------------------------------------
def _build_optax():
  optax = types.ModuleType('optax')

  class _Adam:
    def __init__(self, learning_rate, b1=0.9, b2=0.999, eps=1e-8):
      self.lr, self.b1, self.b2, self.eps = learning_rate, b1, b2, eps

    def init(self, params):
      z = lambda p: np.zeros_like(p)
      return {'count': 0, 'mu': tree_map(z, params), 'nu': tree_map(z, params)}

    def update(self, grads, state, params=None):
      b1, b2 = self.b1, self.b2
      mu = tree_map(lambda m, g: b1 * m + (1 - b1) * g, state['mu'], grads)
      nu = tree_map(lambda v, g: b2 * v + (1 - b2) * g * g, state['nu'], grads)
      c = state['count'] + 1
      lr = self.lr(state['count']) if callable(self.lr) else self.lr
      upd = tree_map(lambda m, v: (-lr * (m / (1 - b1**c)) / (np.sqrt(v / (1 - b2**c)) + self.eps)).astype(m.dtype),
                     mu, nu)
      return upd, {'count': c, 'mu': mu, 'nu': nu}

  class _Zero:
    # optax.set_to_zero: stateless, every update is zero
    def init(self, params):
      return {}

    def update(self, grads, state, params=None):
      return tree_map(lambda g: np.zeros_like(g), grads), state

  class _Multi:
    # optax.multi_transform(transforms, param_labels): each leaf is updated by the transform its label names.  Both
    # transforms in use here (adam, set_to_zero) act leaf by leaf, so running each on the whole tree and picking per
    # leaf equals running each on its own partition.
    def __init__(self, transforms, labels):
      self.tx, self.labels = dict(transforms), labels

    def init(self, params):
      return {name: t.init(params) for name, t in self.tx.items()}

    def update(self, grads, state, params=None):
      upd, new = {}, {}
      for name, t in self.tx.items():
        upd[name], new[name] = t.update(grads, state[name], params)
      names = list(self.tx)
      pick = lambda label, *us: us[names.index(label)]
      return tree_map(pick, self.labels, *[upd[n] for n in names]), new

  optax.adam = _Adam
  optax.set_to_zero = _Zero
  optax.multi_transform = _Multi
  return optax


# ---------------------------------------------------------------------------------------------
# the remaining corners of jax
# ---------------------------------------------------------------------------------------------
class Tape:
  """lax.stop_gradient record / replay (float64 central differences of the reference's loss).
  `refresh_from`: while replaying, entries at or beyond this position are re-recorded from the live
  value (used once to rebuild the downstream constants after the leading entries were replaced)."""
  mode = 'off'
  vals = []
  pos = 0
  refresh_from = None

  @classmethod
  def stop_gradient(cls, x):
    if cls.mode == 'record':
      cls.vals.append(tree_map(lambda a: np.array(a, copy=True), x))
    elif cls.mode == 'replay':
      i = cls.pos
      cls.pos += 1
      if cls.refresh_from is not None and i >= cls.refresh_from:
        cls.vals[i] = tree_map(lambda a: np.array(a, copy=True), x)
        return x
      return cls.vals[i]
    return x


HOOK = {'grad': None, 'loss_fn': None}


def _conv_same(lhs, rhs, strides, padding):
  """jax.lax.conv (NCHW x OIHW, cross-correlation, XLA 'SAME': pad_lo=(k-1)//2, pad_hi=k-1-pad_lo)."""
  assert padding == 'SAME' and tuple(strides) == (1, 1)
  n, c, h, w = lhs.shape
  o, i, kh, kw = rhs.shape
  assert i == c
  pt, pl = (kh - 1) // 2, (kw - 1) // 2
  x = np.zeros((n, c, h + kh - 1, w + kw - 1), lhs.dtype)
  x[:, :, pt:pt + h, pl:pl + w] = lhs
  out = np.zeros((n, o, h, w), np.result_type(lhs.dtype, rhs.dtype))
  for dy in range(kh):
    for dx in range(kw):
      out += np.einsum('nchw,oc->nohw', x[:, :, dy:dy + h, dx:dx + w], rhs[:, :, dy, dx])
  return out


def install(bindings=None):
  """Installs every stand-in module; returns the `jax` stand-in."""
  jax = _jax_standin.install()
  jnp = jax.numpy
  keep64 = lambda: _jax_standin.KEEP64[0]

  def _f(x):
    if keep64():
      return x
    if isinstance(x, np.ndarray) and x.dtype == np.float64:
      return x.astype(f32)
    if isinstance(x, np.float64):
      return f32(x)
    return x

  def _ax(a):
    return tuple(a) if isinstance(a, list) else a

  # x64 is disabled in the reference's runs: python-float math that numpy would carry as float64
  # (np.exp(python float), np.float64 scalars times float32 arrays) is float32 in jax.
  for name in ('exp', 'log', 'sqrt', 'maximum', 'minimum', 'sum', 'prod', 'clip', 'abs', 'square'):
    fn = getattr(np, name)
    setattr(jnp, name, (lambda fn: lambda *a, **k: _f(fn(*a, **k)))(fn))
  jnp.mean = lambda x, axis=None, keepdims=False: _f(np.mean(x, axis=_ax(axis), keepdims=keepdims))
  jnp.array = lambda x, dtype=None: _f(np.array(x, dtype=dtype))
  jnp.quantile = lambda x, q: _f(np.quantile(x, q))            # both default to linear interpolation
  jnp.ones_like = lambda x, dtype=None: np.ones_like(x, dtype=dtype)
  jnp.iinfo, jnp.finfo, jnp.int32 = np.iinfo, np.finfo, np.int32
  jnp.ndarray = np.ndarray
  jnp.zeros = lambda shape, dtype=f32: np.zeros(shape, dtype)

  jax.lax.stop_gradient = Tape.stop_gradient
  jax.lax.pmean = lambda x, axis_name=None: x
  jax.lax.all_gather = lambda x, axis_name=None: tree_map(lambda a: np.asarray(a)[None], x)
  jax.lax.conv = _conv_same

  tu = types.ModuleType('jax.tree_util')
  tu.tree_map, tu.tree_reduce, tu.tree_leaves = tree_map, tree_reduce, tree_leaves
  jax.tree_util = tu
  jax.local_device_count = lambda: 1
  jax.device_count = lambda: 1
  jax.process_count = lambda: 1
  jax.process_index = lambda: 0
  jax.jit = lambda fn, **k: fn
  jax.pmap = lambda fn, **k: fn                     # one device, no leading axis

  def value_and_grad(fn, has_aux=False):
    def run(params):
      HOOK['loss_fn'] = fn
      out = fn(params)
      return out, HOOK['grad'](params)
    return run

  jax.value_and_grad = value_and_grad

  ini = types.ModuleType('jax.nn.initializers')
  ini.he_uniform = he_uniform
  ini.he_normal = ini.glorot_normal = ini.glorot_uniform = he_uniform
  jax.nn.initializers = ini
  jax.nn.relu, jax.nn.softplus, jax.nn.sigmoid = _relu, _softplus, _sigmoid
  jax.nn.silu = lambda x: x * _sigmoid(x)

  # keys are functional, as jax's: the same key always yields the same numbers; split() derives
  # children; every uniform draw is also logged on the root key in call order
  class JKey:
    def __init__(self, path, log=None):
      self.path = tuple(path)
      self.draws = [] if log is None else log

    @property
    def rng(self):
      return np.random.default_rng(list(self.path))

    def uniform01(self, shape):
      u = self.rng.random(shape, dtype=np.float32)
      self.draws.append(u)
      return u

  jax.random.split = lambda key, num=2: [JKey(key.path + (i,), key.draws) for i in range(num)]
  jax.random.PRNGKey = lambda seed: JKey((seed,))
  jax.random.permutation = lambda key, n: key.rng.permutation(n)

  sys.modules.update({'jax.tree_util': tu, 'jax.nn.initializers': ini})
  sys.modules.update(_build_flax())
  gin = _build_gin()
  sys.modules['gin'] = gin
  sys.modules['gin.config'] = gin.config
  sys.modules['optax'] = _build_optax()
  absl = types.ModuleType('absl')
  absl.flags = types.ModuleType('absl.flags')
  for n in ('DEFINE_string', 'DEFINE_multi_string', 'DEFINE_integer', 'DEFINE_bool'):
    setattr(absl.flags, n, lambda *a, **k: None)
  absl.flags.FLAGS = types.SimpleNamespace()
  sys.modules['absl'] = absl
  sys.modules['absl.flags'] = absl.flags
  for name in ('dm_pix', 'cv2', 'pycolmap'):
    sys.modules.setdefault(name, types.ModuleType(name))
  sys.modules['dm_pix'].ssim = None
  # train_utils imports these two for type hints only; their own imports (cv2, pycolmap, PIL
  # readers) are irrelevant to the path, so empty placeholders stand in.
  for name in ('internal.camera_utils', 'internal.datasets'):
    sys.modules[name] = types.ModuleType(name)
  if bindings is not None:
    set_bindings(bindings)
  return jax

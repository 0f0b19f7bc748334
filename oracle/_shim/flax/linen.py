"""Eager NumPy stand-in for the slice of flax.linen (0.3.x) the reference's modules use (see ../README.md).

Semantics reproduced: modules are dataclasses; submodules created inside an @nn.compact method are auto-named
`<Class>_<n>` in construction order unless `name=` is given; submodules assigned in `setup()` are named after the
attribute (dict values: `<attr>_<key>`); parameters of a submodule are `parent_params[name]`; nn.Dense computes
`x @ kernel + bias` with kernel [in, out]; nn.Embed gathers rows of `embedding`; nn.vmap (params broadcast, rngs
not split) maps `__call__` over an axis.  Nothing is initialised here: parameters always come from `apply`."""
import dataclasses as _dc
import functools as _ft
from typing import Any, Optional

import numpy as _np
from jax import nn as _jnn
from jax import vmap as _jvmap
from jax.nn import initializers  # noqa: F401

relu, sigmoid, softplus = _jnn.relu, _jnn.sigmoid, _jnn.softplus
_stack = []


def compact(fn):
  fn._compact = True
  return fn


class Module:
  def __init_subclass__(cls, **kw):
    super().__init_subclass__(**kw)
    _dc.dataclass(cls, repr=False, eq=False)
    gen_init = cls.__init__

    def __init__(self, *a, name=None, parent=None, **k):
      d = self.__dict__
      d['_in_setup'] = False; d['_setup_done'] = False; d['_counters'] = {}; d['_depth'] = 0
      d['_bound'] = None; d['_rngs'] = None
      gen_init(self, *a, **k)
      if parent is None and _stack:
        parent = _stack[-1]
      d['parent'] = parent
      if name is None and parent is not None and not parent._in_setup:   # created inside a compact method
        n = parent._counters.get(type(self).__name__, 0)
        parent._counters[type(self).__name__] = n + 1
        name = f'{type(self).__name__}_{n}'
      d['name'] = name
    cls.__init__ = __init__
    for attr, fn in list(cls.__dict__.items()):
      if callable(fn) and not isinstance(fn, (staticmethod, classmethod, type)) and (attr == '__call__' or not attr.startswith('_')) \
          and attr != 'setup':
        setattr(cls, attr, Module._wrap(fn))

  @staticmethod
  def _wrap(fn):
    @_ft.wraps(fn)
    def wrapped(self, *a, **k):
      self._ensure_setup()
      if self._depth == 0:
        self._counters.clear()
      self.__dict__['_depth'] += 1
      _stack.append(self)
      try:
        return fn(self, *a, **k)
      finally:
        _stack.pop()
        self.__dict__['_depth'] -= 1
    return wrapped

  def __setattr__(self, key, value):
    if self.__dict__.get('_in_setup'):
      if isinstance(value, Module):
        value.__dict__['name'] = key; value.__dict__['parent'] = self
      elif isinstance(value, dict):
        for k, v in value.items():
          if isinstance(v, Module):
            v.__dict__['name'] = f'{key}_{k}'; v.__dict__['parent'] = self
    self.__dict__[key] = value

  def __getattr__(self, key):   # attributes defined by setup()
    if key.startswith('__') or self.__dict__.get('_setup_done') or self.__dict__.get('_in_setup'):
      raise AttributeError(key)
    self._ensure_setup()
    if key in self.__dict__:
      return self.__dict__[key]
    raise AttributeError(key)

  def setup(self):
    pass

  def _ensure_setup(self):
    if self._setup_done or self._in_setup:
      return
    self.__dict__['_in_setup'] = True
    _stack.append(self)
    try:
      self.setup()
    finally:
      _stack.pop()
      self.__dict__['_in_setup'] = False
      self.__dict__['_setup_done'] = True

  def _root(self):
    m = self
    while m.parent is not None:
      m = m.parent
    return m

  def _params(self):
    if self.parent is None:
      return self._bound['params']
    return self.parent._params()[self.name]

  def make_rng(self, name):
    return self._root()._rngs[name]

  def apply(self, variables, *args, rngs=None, method=None, **kwargs):
    self.__dict__['_bound'] = variables
    self.__dict__['_rngs'] = rngs or {}
    fn = method if method is not None else type(self).__call__
    if hasattr(fn, '__func__'):
      fn = fn.__func__
    return fn(self, *args, **kwargs)

  def init(self, *a, **k):
    raise NotImplementedError('parameter initialisation is outside the shim')


class Dense(Module):
  features: int
  use_bias: bool = True
  kernel_init: Any = None
  bias_init: Any = None
  dtype: Any = None
  precision: Any = None

  def __call__(self, x):
    p = self._params()
    k = _np.asarray(p['kernel'])
    assert k.shape[-1] == self.features, (self.name, k.shape, self.features)
    y = _np.asarray(x) @ k
    if self.use_bias:
      y = y + _np.asarray(p['bias'])
    return y


class Embed(Module):
  num_embeddings: int
  features: int
  embedding_init: Any = None
  dtype: Any = None

  def __call__(self, inputs):
    e = _np.asarray(self._params()['embedding'])
    assert e.shape == (self.num_embeddings, self.features), (e.shape, self.num_embeddings, self.features)
    return e[_np.asarray(inputs).astype(_np.int64)]


def vmap(target, variable_axes=None, split_rngs=None, in_axes=0, out_axes=0, **_kw):
  """nn.vmap of a Module class with broadcast params: maps __call__ over the given axes."""
  inner = target.__call__

  def call(self, *args):
    return _jvmap(lambda *a: inner(self, *a), in_axes=in_axes, out_axes=out_axes)(*args)
  return type('Vmap' + target.__name__, (target,), {'__call__': call, '__annotations__': {}})


def elu(x):
  return _np.where(x > 0, x, _np.expm1(x))


def tanh(x):
  return _np.tanh(x)


def softmax(x, axis=-1):
  e = _np.exp(x - _np.max(x, axis=axis, keepdims=True))
  return e / e.sum(axis=axis, keepdims=True)


def gelu(x):
  return 0.5 * x * (1 + _np.tanh(_np.sqrt(2 / _np.pi) * (x + 0.044715 * x ** 3)))


def leaky_relu(x, negative_slope=0.01):
  return _np.where(x >= 0, x, negative_slope * x)

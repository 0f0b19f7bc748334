"""NumPy stand-in for the parts of jax the reference's hot path touches (see ../README.md)."""
import functools as _ft

import numpy as _np

from . import lax, nn, numpy, random, tree_util  # noqa: F401


def jit(fun=None, **_kw):
  if fun is None:
    return lambda f: f
  return fun


def _slice(x, ax, i):
  if ax is None:
    return x
  if isinstance(x, dict):
    return {k: _slice(v, ax, i) for k, v in x.items()}
  return _np.take(x, i, axis=ax)


def _stack(outs, ax):
  first = outs[0]
  if isinstance(first, dict):
    return {k: _stack([o[k] for o in outs], ax) for k in first}
  if isinstance(first, (tuple, list)):
    return type(first)(_stack([o[j] for o in outs], ax) for j in range(len(first)))
  return _np.stack(outs, axis=ax)


def _leading(x, ax):
  if isinstance(x, dict):
    return _leading(next(iter(x.values())), ax)
  return _np.shape(x)[ax]


def vmap(fun, in_axes=0, out_axes=0):
  """Eager loop over the mapped axis (enough for the reference's per-point / per-sample functions)."""
  @_ft.wraps(fun)
  def mapped(*args):
    axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
    n = next(_leading(a, ax) for a, ax in zip(args, axes) if ax is not None)
    outs = [fun(*[_slice(a, ax, i) for a, ax in zip(args, axes)]) for i in range(n)]
    return _stack(outs, out_axes)
  return mapped


def jacfwd(fun, argnums=0, eps=1e-6):
  """Central finite differences in float64 (jax.jacfwd has no NumPy equivalent): d out / d args[argnums]."""
  def jac(*args):
    x = _np.asarray(args[argnums], dtype=_np.float64)
    cols = []
    for j in range(x.size):
      d = _np.zeros_like(x).reshape(-1)
      d[j] = eps
      d = d.reshape(x.shape)
      hi = _np.asarray(fun(*[x + d if k == argnums else a for k, a in enumerate(args)]))
      lo = _np.asarray(fun(*[x - d if k == argnums else a for k, a in enumerate(args)]))
      cols.append((hi - lo) / (2 * eps))
    return _np.stack(cols, axis=-1)
  return jac


def value_and_grad(*_a, **_k):
  raise NotImplementedError('reverse-mode autodiff is outside the NumPy shim')


grad = value_and_grad


class custom_jvp:   # noqa: N801
  """Forward evaluation only (the JVP rule is never used by the shim)."""

  def __init__(self, fun, nondiff_argnums=()):
    self.fun = fun
    _ft.update_wrapper(self, fun)

  def __call__(self, *a, **k):
    return self.fun(*a, **k)

  def defjvp(self, rule):
    return rule


tree_map = tree_util.tree_map
tree_multimap = tree_util.tree_map


def local_device_count():
  return 1


def device_count():
  return 1


def device_get(x):
  return x


def process_index():
  return 0


def process_count():
  return 1

"""jax.random -> caller-supplied arrays.  A "key" is a Key holding the uniforms / normals the call must return,
so the reference code paths that draw random numbers are driven with the SAME numbers the oracle gets."""
import numpy as _np


class Key:
  def __init__(self, uniform=None, normal=None, choice=None):
    self.u, self.n, self.c = uniform, normal, choice


def PRNGKey(seed):
  return Key()


def split(key, num=2):
  return [key for _ in range(num)]


def uniform(key, shape, dtype=None, minval=0.0, maxval=1.0):
  assert key.u is not None, 'this code path draws uniforms: pass random.Key(uniform=...)'
  u = _np.asarray(key.u, dtype=_np.float64)
  assert tuple(u.shape) == tuple(shape), (u.shape, shape)
  return minval + (maxval - minval) * u


def normal(key, shape, dtype=None):
  assert key.n is not None, 'this code path draws normals: pass random.Key(normal=...)'
  n = _np.asarray(key.n, dtype=_np.float64)
  assert tuple(n.shape) == tuple(shape), (n.shape, shape)
  return n


def choice(key, a, shape=(), replace=True, p=None):
  """random.choice(key, a, shape): key.c holds the INDICES into `a` the call must pick."""
  assert key.c is not None, 'this code path draws a choice: pass random.Key(choice=indices)'
  idx = _np.asarray(key.c)
  assert tuple(idx.shape) == tuple(shape), (idx.shape, shape)
  return _np.asarray(a)[idx]

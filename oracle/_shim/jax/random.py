"""jax.random -> caller-supplied arrays.  A "key" is a Key holding the uniforms / normals the call must return,
so the reference code paths that draw random numbers are driven with the SAME numbers the oracle gets."""
import numpy as _np


class Key:
  def __init__(self, uniform=None, normal=None):
    self.u, self.n = uniform, normal


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

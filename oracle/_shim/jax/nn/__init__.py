import numpy as _np

from . import initializers  # noqa: F401


def relu(x):
  return _np.maximum(x, 0)


def sigmoid(x):
  return 1.0 / (1.0 + _np.exp(-x))


def softplus(x):
  return _np.logaddexp(x, 0.0)

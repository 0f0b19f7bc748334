"""jax.numpy -> NumPy (float64)."""
import numpy as _np
from numpy import *  # noqa: F401,F403
from numpy import linalg, newaxis, pi, ndarray, float32, float64, int32, uint32, uint8  # noqa: F401


def __getattr__(name):
  return getattr(_np, name)

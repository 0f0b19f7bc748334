"""tensorflow.experimental.numpy: names only (nerfies.tf_camera's annotations); the TF camera is not run."""
import numpy as _np

ndarray = _np.ndarray
float32 = _np.float32


def __getattr__(name):
  raise NotImplementedError(f'tensorflow.experimental.numpy.{name} is outside the shim')

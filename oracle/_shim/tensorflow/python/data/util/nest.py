"""tensorflow.python.data.util.nest: imported by nerfies.datasets.core for its lazy tf.data path only."""


def __getattr__(name):
  raise NotImplementedError(f'nest.{name} is outside the shim')

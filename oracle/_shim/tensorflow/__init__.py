"""Only so that `nerfies.gpath` imports; file IO through tf.io.gfile is not provided."""


class _GFile:
  def __getattr__(self, name):
    raise NotImplementedError('tensorflow.io.gfile is outside the shim')


class io:   # noqa: N801
  gfile = _GFile()

"""Initialisers are never run: the parameters come from the caller."""


def _init(*_a, **_k):
  def init(*_args, **_kw):
    raise NotImplementedError('parameter initialisation is outside the shim')
  return init


glorot_uniform = xavier_uniform = uniform = zeros = ones = lecun_normal = normal = _init

import dataclasses as _dc


def dataclass(cls):
  cls = _dc.dataclass(cls)
  cls.replace = lambda self, **kw: _dc.replace(self, **kw)
  return cls


def field(pytree_node=True, **kw):
  return _dc.field(**kw)

"""tensorflow -> just enough for nerfies.gpath / nerfies.datasets to IMPORT and to read local files.

tf.io.gfile maps onto the local file system; TensorSpec / dtypes / tf.data names exist so that module-level tables
build; anything that would actually run a tf.data pipeline raises."""
import glob as _glob
import os as _os
import shutil as _shutil


class _GFile:
  @staticmethod
  def GFile(path, mode='r', *a, **k):   # noqa: N802
    return open(_os.fspath(path), mode)

  exists = staticmethod(lambda p: _os.path.exists(_os.fspath(p)))
  isdir = staticmethod(lambda p: _os.path.isdir(_os.fspath(p)))
  listdir = staticmethod(lambda p: sorted(_os.listdir(_os.fspath(p))))
  glob = staticmethod(lambda pattern: sorted(_glob.glob(pattern)))
  makedirs = staticmethod(lambda p: _os.makedirs(_os.fspath(p), exist_ok=True))
  mkdir = staticmethod(lambda p: _os.mkdir(_os.fspath(p)))
  rmtree = staticmethod(lambda p: _shutil.rmtree(_os.fspath(p)))


class io:   # noqa: N801
  gfile = _GFile()


float32, uint32, string = 'float32', 'uint32', 'string'


class TensorSpec:
  def __init__(self, shape=None, dtype=None):
    self.shape, self.dtype = shape, dtype


class _Outside:
  def __init__(self, what):
    self._what = what

  def __getattr__(self, name):
    raise NotImplementedError(f'{self._what}.{name} is outside the shim (no tf.data pipelines here)')


class data:   # noqa: N801
  class experimental:   # noqa: N801
    AUTOTUNE = -1
  Dataset = _Outside('tf.data.Dataset')


class config:   # noqa: N801
  class experimental:   # noqa: N801
    set_visible_devices = staticmethod(lambda *a, **k: None)

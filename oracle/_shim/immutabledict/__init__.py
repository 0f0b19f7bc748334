import collections.abc as _abc


class immutabledict(_abc.Mapping):   # noqa: N801
  def __init__(self, *a, **k):
    self._d = dict(*a, **k)

  def __getitem__(self, key):
    return self._d[key]

  def __iter__(self):
    return iter(self._d)

  def __len__(self):
    return len(self._d)

  def __hash__(self):
    return hash(tuple(sorted(self._d.items())))

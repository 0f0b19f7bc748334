"""Small host utilities the drivers use (nerfies/utils.py:283-293, 370-465): psnr, strided subsets, meters, timers,
a JSON-lines scalar log (tensorboard is not available in this image)."""
import collections
import contextlib
import json
import math
import os
import time

import numpy as np


def compute_psnr(mse):
  """utils.py:283-293."""
  return -10.0 * math.log10(float(mse))


def strided_subset(sequence, count):
  """At most ~count items at a regular stride (utils.py:370-375)."""
  if count:
    return sequence[::max(1, len(sequence) // count)]
  return sequence


class ValueMeter:
  """utils.py:392-415."""

  def __init__(self):
    self._values = []

  def reset(self):
    self._values.clear()

  def update(self, value):
    self._values.append(float(value))

  def reduce(self, reduction='mean'):
    if reduction == 'mean':
      return float(np.mean(self._values))
    if reduction == 'std':
      return float(np.std(self._values))
    if reduction == 'last':
      return self._values[-1]
    raise ValueError(f'Unknown reduction {reduction}')


class TimeTracker:
  """Host wall-clock sections averaged over steps (utils.py:418-465).  Device-side per-kernel times come from
  nrf_profile_enable / nrf_profile_read."""

  def __init__(self):
    self._meters = collections.defaultdict(ValueMeter)
    self._marked = {}

  @contextlib.contextmanager
  def record_time(self, key):
    t0 = time.time()
    yield
    self.update(key, time.time() - t0)

  def update(self, key, value):
    self._meters[key].update(value)

  def tic(self, *keys):
    for k in keys:
      self._marked[k] = time.time()

  def toc(self, *keys):
    for k in keys:
      self.update(k, time.time() - self._marked.pop(k))

  def reset(self):
    for m in self._meters.values():
      m.reset()

  def summary(self, reduction='mean'):
    out = {k: v.reduce(reduction) for k, v in self._meters.items() if v._values}
    if 'total' not in out:
      out['total'] = sum(out.values())
    out['steps_per_sec'] = 1.0 / max(out['total'], 1e-12)
    return out

  def summary_str(self, reduction='mean'):
    return ', '.join(f'{k}={v:.04f}' for k, v in self.summary(reduction).items())


class ScalarLog:
  """Stand-in for the tensorboard SummaryWriter calls of train.py / eval.py: one JSON object per line."""

  def __init__(self, directory, enabled=True):
    """enabled=False: a writer that drops everything (ranks other than 0: the reference writes summaries on process 0
    only, eval.py `jax.process_index() == 0`)."""
    self._fp = None
    if enabled:
      os.makedirs(directory, exist_ok=True)
      self._fp = open(os.path.join(directory, 'scalars.jsonl'), 'a')

  def scalar(self, tag, value, step):
    if self._fp is None:
      return
    self._fp.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')
    self._fp.flush()

  def text(self, tag, textdata, step):
    if self._fp is None:
      return
    self._fp.write(json.dumps({'tag': tag, 'text': textdata, 'step': int(step)}) + '\n')
    self._fp.flush()

  def close(self):
    if self._fp is not None:
      self._fp.close()

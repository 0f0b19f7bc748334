"""Scalar schedules (learning rate, warp_alpha, elastic / background loss weights).

Host-side mirror of nerfies/schedules.py (the reference's is host Python too: `train.py:219-233` evaluates the
schedules once per step and feeds the scalars to the jitted step).  Same class names, constructor arguments,
`from_tuple / from_dict / from_config` and `SCHEDULE_MAP` keys, so the gin presets' schedule tuples
(`configs/*.gin`, e.g. ('piecewise', [(50000, ('constant', 0.01)), ...])) evaluate to the same numbers;
tests/test_reference_vectors.py checks every type against the reference's own output.

Each schedule is a pure function step -> python float (what goes into nrf_step_scalars / nrf_adam_step)."""
import bisect
import collections.abc
import math
from typing import Any, Iterable, Tuple, Union


class Schedule:
  """step -> value.  Subclasses implement get()."""

  def get(self, step):
    raise NotImplementedError

  def __call__(self, step):
    return self.get(step)


class ConstantSchedule(Schedule):
  """nerfies/schedules.py:61-70."""

  def __init__(self, value):
    self.value = value

  def get(self, step):
    return float(self.value)


class LinearSchedule(Schedule):
  """Linear ramp from initial_value to final_value over num_steps, then flat (schedules.py:73-88)."""

  def __init__(self, initial_value, final_value, num_steps):
    self.initial_value, self.final_value, self.num_steps = initial_value, final_value, num_steps

  def get(self, step):
    if self.num_steps == 0:
      return float(self.final_value)
    t = min(step / self.num_steps, 1.0)
    return (1.0 - t) * self.initial_value + t * self.final_value


class ExponentialSchedule(Schedule):
  """Geometric decay initial -> max(final, eps) with exponent step/(num_steps-1); exactly final_value from
  num_steps on (schedules.py:91-115)."""

  def __init__(self, initial_value, final_value, num_steps, eps=1e-10):
    if initial_value <= final_value:
      raise ValueError('Final value must be less than initial value.')
    self.initial_value, self.final_value, self.num_steps, self.eps = initial_value, final_value, num_steps, eps

  def get(self, step):
    if step >= self.num_steps:
      return float(self.final_value)
    ratio = max(self.final_value, self.eps) / self.initial_value
    return self.initial_value * ratio ** (step / (self.num_steps - 1))


class CosineEasingSchedule(Schedule):
  """initial + (final - initial) * (1 - cos(pi t)) / 2, t = clip(step / num_steps, 0, 1) (schedules.py:118-133)."""

  def __init__(self, initial_value, final_value, num_steps):
    self.initial_value, self.final_value, self.num_steps = initial_value, final_value, num_steps

  def get(self, step):
    t = min(max(step / self.num_steps, 0.0), 1.0)
    return self.initial_value + (self.final_value - self.initial_value) * 0.5 * (1.0 + math.cos(math.pi * t + math.pi))


class StepSchedule(Schedule):
  """initial * decay_factor ** (step // decay_interval), final_value once max_decays is reached
  (schedules.py:136-160)."""

  def __init__(self, initial_value, decay_interval, decay_factor, max_decays, final_value=None):
    self.initial_value, self.decay_interval = initial_value, decay_interval
    self.decay_factor, self.max_decays = decay_factor, max_decays
    self.final_value = initial_value * decay_factor ** max_decays if final_value is None else final_value

  def get(self, step):
    phase = step // self.decay_interval
    if phase >= self.max_decays:
      return float(self.final_value)
    return self.initial_value * self.decay_factor ** phase


class PiecewiseSchedule(Schedule):
  """[(length, schedule), ...]: piece i is active for `length_i` steps and sees the step counted from its own
  start; the last piece runs forever (schedules.py:163-177)."""

  def __init__(self, schedules: Iterable[Tuple[int, Union[Schedule, Iterable[Any]]]]):
    schedules = list(schedules)
    self.schedules = [from_config(s) for _, s in schedules]
    self.milestones = []
    total = 0
    for length, _ in schedules[:-1]:
      total += length
      self.milestones.append(total)

  def get(self, step):
    idx = bisect.bisect_right(self.milestones, step)
    start = self.milestones[idx - 1] if idx >= 1 else 0
    return self.schedules[idx].get(step - start)


class DelayedSchedule(Schedule):
  """base(step) scaled by delay_mult + (1 - delay_mult) sin(pi/2 clip(step / delay_steps, 0, 1))
  (schedules.py:180-194)."""

  def __init__(self, base_schedule, delay_steps, delay_mult):
    self.base_schedule = from_config(base_schedule)
    self.delay_steps, self.delay_mult = delay_steps, delay_mult

  def get(self, step):
    t = min(max(step / self.delay_steps, 0.0), 1.0)
    rate = self.delay_mult + (1.0 - self.delay_mult) * math.sin(0.5 * math.pi * t)
    return rate * self.base_schedule(step)


SCHEDULE_MAP = {
    'constant': ConstantSchedule,
    'linear': LinearSchedule,
    'exponential': ExponentialSchedule,
    'cosine_easing': CosineEasingSchedule,
    'step': StepSchedule,
    'piecewise': PiecewiseSchedule,
    'delayed': DelayedSchedule,
}


def from_tuple(x):
  kind, *args = x
  return SCHEDULE_MAP[kind](*args)


def from_dict(d):
  d = dict(d)
  kind = d.pop('type')
  return SCHEDULE_MAP[kind](**d)


def from_config(schedule):
  """Schedule | (type, *args) | {'type': ..., **kwargs} -> Schedule (schedules.py:24-43)."""
  if isinstance(schedule, Schedule):
    return schedule
  if isinstance(schedule, (tuple, list)):
    return from_tuple(schedule)
  if isinstance(schedule, collections.abc.Mapping):
    return from_dict(schedule)
  raise ValueError(f'Unknown type {type(schedule)}.')

"""Checkpoint save / restore in the reference's on-disk format.

The reference checkpoints with `flax.training.checkpoints.{save,restore}_checkpoint(dir, TrainState, step)`
(train.py:231, 315-321; eval.py:296-305): one file `checkpoint_<step>` holding the msgpack encoding of the state
dict (flax.serialization).  flax is not installed here, so its published wire format is restated (msgpack is):

  * containers are msgpack maps keyed by field / dict key (strings);
  * an ndarray is ExtType(1, packb((shape, dtype.name, raw little-endian bytes)));
  * a NumPy scalar is ExtType(3, <same triple, shape ()>), python ints / floats are native msgpack;
  * the TrainState (model_utils.py:25-33) of an Adam optimizer (flax 0.3.x optim) is
      {'optimizer': {'target': {'model': <params>},
                     'state': {'step': int32, 'param_states': {'model': <per-leaf {'grad_ema','grad_sq_ema'}>}}},
       'warp_alpha': f32, 'time_alpha': f32}

so a checkpoint written by google/nerfies loads into this framework (parameters + Adam moments + step) and
vice versa.  PARITY UNPINNED for the container layout: no flax here and the reference ships no checkpoint
fixture; tests round-trip the format and check it byte-level against the description above.

File management follows flax.training.checkpoints: `<prefix><step>` files, atomic rename from a tmp file, newest
`keep` retained, refusal to write an older step unless overwrite=True, restore picks the numerically latest."""
import os
import re
from typing import Any, Dict, Optional

import msgpack
import numpy as np
import torch

from . import params as P

_EXT_NDARRAY, _EXT_NPSCALAR = 1, 3


def _pack_array(a: np.ndarray) -> bytes:
  a = np.asarray(a)
  if a.dtype.byteorder == '>':
    a = a.astype(a.dtype.newbyteorder('<'))
  return msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True)


def _default(o):
  if isinstance(o, torch.Tensor):
    o = o.detach().cpu().numpy()
  if isinstance(o, np.ndarray):
    return msgpack.ExtType(_EXT_NDARRAY, _pack_array(o))
  if isinstance(o, np.generic):
    return msgpack.ExtType(_EXT_NPSCALAR, _pack_array(np.asarray(o)))
  raise TypeError(f'cannot serialise {type(o)}')


def _ext_hook(code, data):
  if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
    shape, dtype, buf = msgpack.unpackb(data, raw=False)
    arr = np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape)
    return arr[()] if code == _EXT_NPSCALAR else arr.copy()
  return msgpack.ExtType(code, data)


def to_bytes(state_dict: Dict[str, Any]) -> bytes:
  return msgpack.packb(state_dict, default=_default, use_bin_type=True, strict_types=False)


def from_bytes(data: bytes) -> Dict[str, Any]:
  return msgpack.unpackb(data, ext_hook=_ext_hook, raw=False, strict_map_key=False)


# ---------------------------------------------------------------- TrainState <-> state dict
def _np_tree(flat: torch.Tensor, layout: P.ParamLayout):
  host = flat.detach().cpu()
  return _map_tree(lambda t: t.numpy().copy(), P.tree_from_flat(host, layout))


def _map_tree(fn, tree):
  return {k: _map_tree(fn, v) if isinstance(v, dict) else fn(v) for k, v in tree.items()}


def state_to_dict(state) -> Dict[str, Any]:
  """training.TrainState -> the nested dict flax.serialization.to_state_dict gives for the reference's state."""
  opt = state.optimizer
  layout = opt.target.layout
  m, v = _np_tree(opt.m, layout), _np_tree(opt.v, layout)

  def moments(mt, vt):
    return {k: moments(mt[k], vt[k]) if isinstance(mt[k], dict) else {'grad_ema': mt[k], 'grad_sq_ema': vt[k]}
            for k in mt}
  return {
      'optimizer': {'state': {'step': np.asarray(opt.step, np.int32), 'param_states': {'model': moments(m, v)}},
                    'target': {'model': _np_tree(opt.target.flat, layout)}},
      'warp_alpha': np.asarray(state.warp_alpha, np.float32),
      'time_alpha': np.asarray(state.time_alpha, np.float32),
  }


def state_from_dict(state, d: Dict[str, Any]):
  """Loads `d` (as produced by state_to_dict or by the reference) INTO `state` (device buffers are reused)."""
  opt = state.optimizer
  layout, dev = opt.target.layout, opt.target.flat.device
  try:
    target = d['optimizer']['target']
    target = target.get('model', target)
    P.flat_from_tree(target, layout, dev, out=opt.target.flat)
    st = d['optimizer']['state']
    ps = st['param_states']
    ps = ps.get('model', ps)
    is_leaf = lambda node: isinstance(node, dict) and 'grad_ema' in node

    def split(node, key):
      return {k: (v[key] if is_leaf(v) else split(v, key)) for k, v in node.items()}
    P.flat_from_tree(split(ps, 'grad_ema'), layout, dev, out=opt.m)
    P.flat_from_tree(split(ps, 'grad_sq_ema'), layout, dev, out=opt.v)
  except KeyError as e:
    raise KeyError(f'checkpoint does not match the model: missing {e}') from None
  opt.step = int(np.asarray(st['step']))
  state.warp_alpha = float(np.asarray(d.get('warp_alpha', 0.0)))
  state.time_alpha = float(np.asarray(d.get('time_alpha', 0.0)))
  return state


# ---------------------------------------------------------------- files
def _steps(ckpt_dir: str, prefix: str):
  out = []
  if os.path.isdir(ckpt_dir):
    for f in os.listdir(ckpt_dir):
      m = re.fullmatch(re.escape(prefix) + r'(\d+)', f)
      if m:
        out.append((int(m.group(1)), os.path.join(ckpt_dir, f)))
  return sorted(out)


def latest_checkpoint(ckpt_dir: str, prefix: str = 'checkpoint_') -> Optional[str]:
  found = _steps(ckpt_dir, prefix)
  return found[-1][1] if found else None


def save_checkpoint(ckpt_dir: str, target, step: int, prefix: str = 'checkpoint_', keep: int = 1,
                    overwrite: bool = False) -> str:
  """flax.training.checkpoints.save_checkpoint: writes `<ckpt_dir>/<prefix><step>`, keeps the newest `keep`."""
  os.makedirs(ckpt_dir, exist_ok=True)
  existing = _steps(ckpt_dir, prefix)
  if existing and existing[-1][0] >= step and not overwrite:
    raise ValueError(f'Trying to save an outdated checkpoint at step {step}: latest is {existing[-1][0]}')
  path = os.path.join(ckpt_dir, f'{prefix}{step}')
  tmp = os.path.join(ckpt_dir, f'{prefix}tmp')
  with open(tmp, 'wb') as fp:
    fp.write(to_bytes(target if isinstance(target, dict) else state_to_dict(target)))
  os.replace(tmp, path)
  found = _steps(ckpt_dir, prefix)
  if overwrite:                      # newer files than the one just written are stale
    for s, f in found:
      if s > step:
        os.remove(f)
    found = [(s, f) for s, f in found if s <= step]
  for _, f in found[:-keep] if keep > 0 else []:
    os.remove(f)
  return path


def restore_checkpoint(ckpt_dir: str, target, step: Optional[int] = None, prefix: str = 'checkpoint_'):
  """flax.training.checkpoints.restore_checkpoint: `ckpt_dir` may be a directory (latest or `step`) or a file.
  Returns `target` unchanged when nothing is found (a fresh run); target=None returns the raw state dict."""
  if os.path.isfile(ckpt_dir):
    path = ckpt_dir
  elif step is not None:
    path = os.path.join(ckpt_dir, f'{prefix}{step}')
    if not os.path.exists(path):
      raise ValueError(f'Matching checkpoint not found: {path}')
  else:
    path = latest_checkpoint(ckpt_dir, prefix)
    if path is None:
      return target
  with open(path, 'rb') as fp:
    d = from_bytes(fp.read())
  if target is None:
    return d
  return state_from_dict(target, d)

"""Flat fp32 parameter buffer <-> flax-style parameter tree.

The C-ABI keeps every leaf in ONE flat buffer (single-buffer all-reduce and Adam); the tree view
uses the flax paths of the reference, e.g.
params['nerf_mlps_coarse']['MLP_0']['hidden_4']['kernel'] with kernels stored [in, out].
"""
import math
from typing import Dict, List, Tuple

import torch


class ParamLayout:
  """Leaves of the flat buffer: (name, offset, shape)."""

  def __init__(self, entries: List[Tuple[str, int, Tuple[int, ...]]], total: int):
    self.entries = entries
    self.total = total
    self.by_name = {n: (o, s) for n, o, s in entries}

  def shape_of(self, name):
    return self.by_name[name][1]


def _leaf_shape(name, rows, cols):
  if name.endswith('/bias'):
    return (cols,)
  return (rows, cols)


def layout_from_infos(infos, total) -> ParamLayout:
  entries = []
  for t in infos:
    name = t.name.decode()
    entries.append((name, int(t.offset), _leaf_shape(name, int(t.rows), int(t.cols))))
  return ParamLayout(entries, int(total))


def tree_from_flat(flat: torch.Tensor, layout: ParamLayout) -> Dict:
  """Nested dict of VIEWS into `flat`."""
  tree: Dict = {}
  for name, off, shape in layout.entries:
    n = math.prod(shape)
    node = tree
    parts = name.split('/')
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    node[parts[-1]] = flat[off:off + n].view(*shape)
  return tree


def flat_from_tree(tree: Dict, layout: ParamLayout, device, out: torch.Tensor = None) -> torch.Tensor:
  flat = out if out is not None else torch.zeros(layout.total, dtype=torch.float32, device=device)
  for name, off, shape in layout.entries:
    node = tree
    for p in name.split('/'):
      node = node[p]
    t = torch.as_tensor(node, dtype=torch.float32).reshape(-1)
    assert t.numel() == math.prod(shape), (name, tuple(node.shape), shape)
    flat[off:off + t.numel()].copy_(t)
  return flat


class FlatParams:
  """The model parameters as one flat device buffer plus its layout; `.tree` gives flax-style views."""

  def __init__(self, flat: torch.Tensor, layout: ParamLayout):
    assert flat.dtype == torch.float32 and flat.numel() == layout.total
    self.flat = flat
    self.layout = layout

  @property
  def tree(self):
    return tree_from_flat(self.flat, self.layout)

  def __getitem__(self, key):
    return self.tree[key]

  def clone(self):
    return FlatParams(self.flat.clone(), self.layout)


def init_flat(layout: ParamLayout, seed: int, device) -> torch.Tensor:
  """Reference initialisation: glorot-uniform kernels, zero biases (modules.py:107-108, 127-140),
  embeddings U[0, 0.05) (glo.py:33); SE3 trunk xavier-uniform (= glorot, warping.py:237), heads U[0,1e-4)."""
  g = torch.Generator(device='cpu')
  g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
  flat = torch.zeros(layout.total, dtype=torch.float32)
  for name, off, shape in layout.entries:
    n = math.prod(shape)
    if (name.startswith('warp_field/branches_') and name.endswith('/kernel')) or name == 'warp_field/mlp/logit/kernel':
      flat[off:off + n] = torch.rand(n, generator=g) * 1e-4   # initializers.uniform(scale=1e-4), one-sided (warping.py:238-239)
    elif name.endswith('/kernel'):
      lim = math.sqrt(6.0 / (shape[0] + shape[1]))
      flat[off:off + n] = (torch.rand(n, generator=g) * 2 - 1) * lim
    elif name.endswith('/embedding'):
      flat[off:off + n] = torch.rand(n, generator=g) * 0.05
  return flat.to(device)

"""Host-side mirror of nerfies.evaluation.render_image (reference: nerfies/evaluation.py:28-101)."""
import math
from typing import Any, Callable, Dict, Optional

import torch
import torch.distributed as dist


def _tree_map(fn, tree):
  if isinstance(tree, dict):
    return {k: _tree_map(fn, v) for k, v in tree.items()}
  return fn(tree)


def render_image(state, rays_dict: Dict[str, Any], model_fn: Callable, device_count: int = 1, rng=0,
                 chunk: int = 8192, default_ret_key: Optional[str] = None):
  """Renders all pixels of an (H,W) ray image in chunks (evaluation.py:62-99).

  model_fn(key_0, key_1, params, chunk_rays_dict, warp_extra) -> {'coarse': {...}, 'fine': {...}}.
  With torch.distributed initialised each rank renders a contiguous slice of every chunk (image
  tiles are disjoint, so the reference's all_gather (eval.py:339) becomes one all_gather of the
  rendered slices); the last chunk is edge-padded to a multiple of the world size
  (evaluation.py:71-78)."""
  h, w = rays_dict['origins'].shape[:2]
  flat = _tree_map(lambda x: x.reshape(h * w, -1), rays_dict)
  num_rays = h * w
  world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
  rank = dist.get_rank() if world > 1 else 0
  del device_count
  ret_maps = []
  for batch_idx in range(int(math.ceil(num_rays / chunk))):
    i0 = batch_idx * chunk
    chunk_rays = _tree_map(lambda x: x[i0:i0 + chunk], flat)
    n = chunk_rays['origins'].shape[0]
    pad = (world - n % world) % world
    if pad:
      chunk_rays = _tree_map(lambda x: torch.cat([x, x[-1:].expand(pad, *x.shape[1:])], 0), chunk_rays)
    per = (n + pad) // world
    mine = _tree_map(lambda x: x[rank * per:(rank + 1) * per], chunk_rays)
    out = model_fn(rng, rng + 1, state.optimizer.target, mine, state.warp_extra)
    ret_key = default_ret_key or ('fine' if 'fine' in out else 'coarse')
    ret = out[ret_key]
    if world > 1:
      gathered = {}
      for k, v in ret.items():
        parts = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(parts, v.contiguous())
        gathered[k] = torch.cat(parts, 0)
      ret = gathered
    if pad:
      ret = {k: v[:-pad] for k, v in ret.items()}
    ret_maps.append(ret)
  return {k: torch.cat([r[k] for r in ret_maps], 0).reshape(h, w, *ret_maps[0][k].shape[1:]) for k in ret_maps[0]}

"""Host-side mirror of nerfies.training (reference: nerfies/training.py:35-271).

`train_step` keeps the reference signature and return triple.  One call = forward + loss +
backward inside the HIP library, one flat-buffer all-reduce over RCCL (the reference's
lax.pmean(grad), training.py:266, with the 1/n folded into Adam) and a fused Adam step
(flax.optim.Adam, training.py:268-269).  State is updated IN PLACE (device buffers are donated,
as `donate_argnums` does in train.py:254-262).
"""
import ctypes as C
import dataclasses
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist

from nerfies_amd import lib as L
from nerfies_amd import models
from nerfies_amd import params as P


@dataclasses.dataclass
class ScalarParams:
  """training.ScalarParams (training.py:35-43)."""
  learning_rate: float
  elastic_loss_weight: float = 0.0
  warp_reg_loss_weight: float = 0.0
  warp_reg_loss_alpha: float = -2.0
  warp_reg_loss_scale: float = 0.001
  background_loss_weight: float = 0.0
  background_noise_std: float = 0.001


class Optimizer:
  """flax.optim.Adam state over the flat buffer (defaults beta=(0.9,0.999), eps=1e-8)."""

  def __init__(self, target: P.FlatParams, beta1=0.9, beta2=0.999, eps=1e-8):
    self.target = target
    self.m = torch.zeros_like(target.flat)
    self.v = torch.zeros_like(target.flat)
    # gradient + the step statistics in ONE buffer: a single all-reduce per step carries both (training.py:266-267)
    n = target.flat.numel()
    self._gs = torch.zeros(n + L.NRF_NUM_STATS, dtype=torch.float32, device=target.flat.device)
    self.grad = self._gs[:n]
    self.stats = self._gs[n:]
    self.step = 0
    self.beta1, self.beta2, self.eps = beta1, beta2, eps

  def apply_gradient(self, grad: torch.Tensor, learning_rate: float, grad_scale: float = 1.0):
    lib = L.load_library()
    p = self.target.flat
    stream = C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
    L.check(lib.nrf_adam_step(C.c_void_p(p.data_ptr()), C.c_void_p(self.m.data_ptr()), C.c_void_p(self.v.data_ptr()),
                              C.c_void_p(grad.data_ptr()), p.numel(), float(learning_rate), self.beta1, self.beta2,
                              self.eps, int(self.step), float(grad_scale), stream), lib)
    self.step += 1
    return self


@dataclasses.dataclass
class TrainState:
  """model_utils.TrainState (model_utils.py:25-33)."""
  optimizer: Optimizer
  warp_alpha: float = 0.0
  time_alpha: float = 0.0

  @property
  def warp_extra(self):
    return {'alpha': self.warp_alpha, 'time_alpha': self.time_alpha}

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


def _world():
  if dist.is_available() and dist.is_initialized():
    return dist.get_world_size()
  return 1


def psum_gradients(grad: torch.Tensor, stats: torch.Tensor, fused: Optional[torch.Tensor] = None):
  """lax.pmean(grad) / lax.pmean(stats) (training.py:266-267) over the ray shards.  `fused`: the buffer both are views
  of (Optimizer._gs) -> ONE all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests); otherwise one each.
  The gradient is left as the SUM -- the 1/world factor is folded into the Adam kernel
  (`grad_scale`) -- and the stats are returned averaged.  Returns (grad, stats, world)."""
  n = _world()
  if n > 1:
    if fused is not None:
      dist.all_reduce(fused, op=dist.ReduceOp.SUM)
    else:
      dist.all_reduce(grad, op=dist.ReduceOp.SUM)
      dist.all_reduce(stats, op=dist.ReduceOp.SUM)
  # always a fresh tensor: `stats` is a view of the donated gradient buffer, which the next step overwrites
  stats = stats / n if n > 1 else stats.clone()
  return grad, stats, n


def train_step(model: models.NerfModel, rng_key, state: TrainState, batch: Dict[str, Any],
               scalar_params: ScalarParams, use_elastic_loss: bool = False, elastic_reduce_method: str = 'median',
               elastic_loss_type: str = 'log_svals', use_background_loss: bool = False,
               use_warp_reg_loss: bool = False, *, rngs: Optional[Dict[str, Any]] = None, bf16: bool = False):
  """One optimisation step (training.py:138-271).  `batch` holds this rank's ray shard
  ('rgb','origins','directions','metadata').  Returns (new_state, stats, rng_key).
  `rngs` (extra, parity runs): explicit uniforms {'coarse': (B,N_c), 'fine': (B,N_f)} instead of the streams derived
  from `rng_key` (the reference's threefry stream is not reproducible without JAX).  `bf16` (extra, no reference
  counterpart; BASELINE config D): the NeRF MLPs run forward / dgrad / wgrad on bfloat16 MFMA operands with a bfloat16
  activation stash; master weights, loss, compositing, the gradient all-reduce and Adam stay float32."""
  if use_elastic_loss and elastic_loss_type not in L.ELASTIC_TYPE:
    raise L.NrfError(f"elastic_loss_type {elastic_loss_type!r} is not built (one of {sorted(L.ELASTIC_TYPE)}; 'nr' produces "
                     'NaNs in the reference itself, training.py:58)')
  # random.split(rng_key, 4) (training.py:168): derive the per-step stream keys from an int key
  rng_key = int(rng_key)
  mix = lambda k, i: (k * 6364136223846793005 + 1442695040888963407 + i) & 0xFFFFFFFFFFFFFFFF
  next_key, fine_key, coarse_key, reg_key = mix(rng_key, 0), mix(rng_key, 1), mix(rng_key, 2), mix(rng_key, 3)
  opt = state.optimizer
  background = None
  if use_background_loss:   # training.compute_background_loss (training.py:117-135, 248-259)
    pts = torch.as_tensor(batch['background_points'], device=opt.target.flat.device).to(torch.float32).reshape(-1, 3)
    g = torch.Generator(device=pts.device).manual_seed(reg_key & 0x7FFFFFFFFFFFFFFF)
    ids_all = torch.as_tensor(list(model.warp_ids), device=pts.device, dtype=torch.int32)
    ids = ids_all[torch.randint(0, ids_all.numel(), (pts.shape[0],), generator=g, device=pts.device)]   # random.choice(warp_ids)
    noise = scalar_params.background_noise_std * torch.randn(pts.shape, generator=g, device=pts.device)
    background = {'points': pts + noise, 'warp_ids': ids, 'weight': scalar_params.background_loss_weight}
  grad, stats = model.loss_and_grad(opt.target, batch, warp_extra=state.warp_extra,
                                    rngs=rngs if rngs is not None else {'fine': fine_key, 'coarse': coarse_key},
                                    grad_out=opt.grad, stats_out=opt.stats, bf16=bf16,
                                    background=background,
                                    elastic={'weight': scalar_params.elastic_loss_weight, 'reduce_method': elastic_reduce_method,
                                             'loss_type': elastic_loss_type} if use_elastic_loss else None,
                                    warp_reg={'weight': scalar_params.warp_reg_loss_weight, 'alpha': scalar_params.warp_reg_loss_alpha,
                                              'scale': scalar_params.warp_reg_loss_scale} if use_warp_reg_loss else None)
  grad, stats, n = psum_gradients(grad, stats, fused=opt._gs)
  opt.apply_gradient(grad, learning_rate=scalar_params.learning_rate, grad_scale=1.0 / n)
  out = {
      'coarse': {'loss/rgb': stats[0], 'loss/total': stats[0], 'metric/psnr': stats[2]},
      'fine': {'loss/rgb': stats[1], 'loss/total': stats[1], 'metric/psnr': stats[3]},
  }
  if use_background_loss:
    out['background_loss'] = stats[5]
  if use_elastic_loss:   # coarse level only (training.py:246)
    out['coarse']['loss/elastic'] = stats[6]
    out['coarse']['residual/elastic'] = stats[7]
    out['coarse']['loss/total'] = stats[0] + scalar_params.elastic_loss_weight * stats[6]
    out['coarse']['metric/jacobian_det'] = stats[12]     # training.py:214-222
    out['coarse']['metric/jacobian_div'] = stats[13]
    out['coarse']['metric/jacobian_curl'] = stats[14]
  if use_warp_reg_loss:  # both levels (training.py:199-212)
    for lv, i in (('coarse', 0), ('fine', 1)):
      out[lv]['loss/warp_reg'] = stats[8 + i]
      out[lv]['residual/warp_reg'] = stats[10 + i]
      out[lv]['loss/total'] = out[lv]['loss/total'] + scalar_params.warp_reg_loss_weight * stats[8 + i]
  return state, out, next_key

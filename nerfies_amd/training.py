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

  def apply_gradient_dynamic(self, grad: torch.Tensor, dynamic: 'DynamicScalars'):
    """The same update with learning rate / bias corrections / grad scale read from the device (nrf_adam_step_dynamic): the
    launch can sit in a captured hipGraph.  The caller writes dynamic.write(..., adam_step=self.step) before each replay and
    bumps self.step."""
    lib = L.load_library()
    p = self.target.flat
    stream = C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
    L.check(lib.nrf_adam_step_dynamic(C.c_void_p(p.data_ptr()), C.c_void_p(self.m.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                      C.c_void_p(grad.data_ptr()), p.numel(), self.beta1, self.beta2, self.eps,
                                      C.c_void_p(dynamic.dev.data_ptr()), stream), lib)
    return self

  def apply_gradient(self, grad: torch.Tensor, learning_rate: float, grad_scale: float = 1.0):
    lib = L.load_library()
    p = self.target.flat
    stream = C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
    L.check(lib.nrf_adam_step(C.c_void_p(p.data_ptr()), C.c_void_p(self.m.data_ptr()), C.c_void_p(self.v.data_ptr()),
                              C.c_void_p(grad.data_ptr()), p.numel(), float(learning_rate), self.beta1, self.beta2,
                              self.eps, int(self.step), float(grad_scale), stream), lib)
    self.step += 1
    return self


class DynamicScalars:
  """nrf_dynamic_scalars in device memory: the scalars that change from step to step (schedules, rng key, Adam's step count).
  Kernels of a captured train step read them from here; `write` refreshes them with a one-thread kernel launched on the
  current stream (values travel as kernel arguments: asynchronous, no host buffer to keep alive)."""

  def __init__(self, device):
    self.dev = torch.zeros(C.sizeof(L.DynamicScalars) // 4, dtype=torch.int32, device=device)

  def write(self, *, warp_alpha=0.0, time_alpha=0.0, elastic_loss_weight=0.0, learning_rate=0.0, adam_step=0, beta1=0.9, beta2=0.999,
            grad_scale=1.0, rng_seed=0, rng_offset=0):
    t = float(adam_step) + 1.0
    v = L.DynamicScalars(float(warp_alpha), float(time_alpha), float(elastic_loss_weight), float(learning_rate),
                         1.0 - beta1 ** t, 1.0 - beta2 ** t, float(grad_scale), 0.0, int(rng_seed) & 0xFFFFFFFFFFFFFFFF,
                         int(rng_offset) & 0xFFFFFFFFFFFFFFFF)
    lib = L.load_library()
    stream = C.c_void_p(torch.cuda.current_stream(self.dev.device).cuda_stream)
    L.check(lib.nrf_dynamic_scalars_write(C.c_void_p(self.dev.data_ptr()), C.byref(v), stream), lib)


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
  # a one-rank communicator still goes through the collective (RCCL on the GPU): the call sequence of an N-GPU job is what a
  # one-GPU box can exercise (tests/test_gpu_rccl.py); without torch.distributed nothing is called
  if dist.is_available() and dist.is_initialized():
    if fused is not None:
      dist.all_reduce(fused, op=dist.ReduceOp.SUM)
    else:
      dist.all_reduce(grad, op=dist.ReduceOp.SUM)
      dist.all_reduce(stats, op=dist.ReduceOp.SUM)
  # always a fresh tensor: `stats` is a view of the donated gradient buffer, which the next step overwrites
  stats = stats / n if n > 1 else stats.clone()
  return grad, stats, n


def _step_keys(rng_key: int):
  """random.split(rng_key, 4) (training.py:168) for an integer key: (next key, fine, coarse, regulariser)."""
  rng_key = int(rng_key)
  mix = lambda k, i: (k * 6364136223846793005 + 1442695040888963407 + i) & 0xFFFFFFFFFFFFFFFF
  return mix(rng_key, 0), mix(rng_key, 1), mix(rng_key, 2), mix(rng_key, 3)


def _background_of(model, batch, scalar_params, device):
  """training.compute_background_loss's inputs (training.py:117-135, 248-259): the raw points, the model's warp ids to draw one
  per point from and the noise level -- the library draws ids and noise itself (Philox streams 4, 5 of the step's key), so a
  step launches no torch kernel (round 2 drew them with torch.randint / randn: 8 extra launches per step)."""
  ids = getattr(model, '_warp_id_choices', None)
  if ids is None or ids.device != device:
    ids = model._warp_id_choices = torch.as_tensor(list(model.warp_ids), device=device, dtype=torch.int32)
  pts = torch.as_tensor(batch['background_points'], device=device).to(torch.float32).reshape(-1, 3)
  return {'points': pts, 'id_choices': ids, 'noise_std': scalar_params.background_noise_std, 'weight': scalar_params.background_loss_weight}


def _stats_dict(stats, scalar_params, use_elastic_loss, use_background_loss, use_warp_reg_loss):
  """The stats tree training.train_step returns (training.py:214-262) from the library's flat stats vector."""
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
  return out


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
  next_key, fine_key, coarse_key, _ = _step_keys(rng_key)   # random.split(rng_key, 4) (training.py:168)
  opt = state.optimizer
  background = _background_of(model, batch, scalar_params, opt.target.flat.device) if use_background_loss else None
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
  out = _stats_dict(stats, scalar_params, use_elastic_loss, use_background_loss, use_warp_reg_loss)
  return state, out, next_key


class GraphedTrainStep:
  """training.train_step captured ONCE into a hipGraph (torch.cuda.CUDAGraph over the library's launches, the RCCL all-reduce
  and the Adam kernel on the capture stream) and replayed per step: the reference jits the whole step into one XLA
  executable (train.py:254-262); here the ~25 launches of a step are one graph launch, which is what a small per-GPU batch
  (the 128-ray share of a 1024-ray global batch on 8 GPUs) needs -- its kernels take ~1 ms in total and eager launches
  from Python do not keep up.

  Everything that changes between steps is read from DEVICE memory: the batch lives in static buffers (`load_batch`), the
  schedules' scalars, the rng key and Adam's step count in a DynamicScalars block refreshed before every replay.
  `gstep(rng_key, scalar_params=None, warp_alpha=None, time_alpha=None, batch=None)` -> stats (as train_step's)."""

  def __init__(self, model, state: TrainState, batch, scalar_params: ScalarParams, use_elastic_loss=False,
               elastic_reduce_method='median', elastic_loss_type='log_svals', use_background_loss=False, use_warp_reg_loss=False,
               bf16=False):
    if use_elastic_loss and elastic_loss_type not in L.ELASTIC_TYPE:
      raise L.NrfError(f'elastic_loss_type {elastic_loss_type!r} is not built')
    self.model, self.state, self.sp = model, state, scalar_params
    self.flags = dict(use_elastic_loss=use_elastic_loss, use_background_loss=use_background_loss, use_warp_reg_loss=use_warp_reg_loss)
    self.el = dict(reduce_method=elastic_reduce_method, loss_type=elastic_loss_type)
    self.bf16 = bf16
    opt = state.optimizer
    dev = opt.target.flat.device
    self.dyn = DynamicScalars(dev)
    self.batch = self._static_copy(batch, dev)
    self.stats_static = None
    # one eager step (uploads the descriptor tables, sizes the workspace), undone afterwards; then the capture
    keep = [t.clone() for t in (opt.target.flat, opt.m, opt.v)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
      self._write(0, scalar_params, state.warp_alpha, state.time_alpha)
      self._enqueue()
    torch.cuda.current_stream().wait_stream(s)
    for t, k in zip((opt.target.flat, opt.m, opt.v), keep):
      t.copy_(k)
    # RCCL's all-reduce is a stream operation and is captured with the rest.  gloo's is a host call (the CPU-side tests, two ranks
    # sharing one GPU): the step is then TWO graphs -- loss + gradient | Adam -- with the collective between them
    self.split = dist.is_available() and dist.is_initialized() and dist.get_backend() != 'nccl'
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      if self.split:
        self._enqueue_grad()
      else:
        self._enqueue()
    if self.split:
      self.graph_adam = torch.cuda.CUDAGraph()
      with torch.cuda.graph(self.graph_adam):
        opt.apply_gradient_dynamic(self._grad, self.dyn)

  @staticmethod
  def _static_copy(batch, dev):
    out = {}
    for k, v in batch.items():
      if k == 'metadata':
        out[k] = {kk: (models._f32(vv, dev).reshape(-1).clone() if kk == 'time' else models._ids(vv, dev).clone()) for kk, vv in (v or {}).items()}
      else:
        out[k] = models._f32(v, dev).clone()
    return out

  def load_batch(self, batch):
    """Copies a new batch (same shapes) into the static buffers the graph reads."""
    for k, v in batch.items():
      if k == 'metadata':
        for kk, vv in (v or {}).items():
          dst = self.batch['metadata'][kk]
          dst.copy_(torch.as_tensor(vv, device=dst.device).reshape(dst.shape))
      else:
        self.batch[k].copy_(torch.as_tensor(v, device=self.batch[k].device).reshape(self.batch[k].shape))

  def _write(self, rng_key, sp, warp_alpha, time_alpha):
    _, fine_key, coarse_key, _ = _step_keys(rng_key)
    opt = self.state.optimizer
    self.dyn.write(warp_alpha=warp_alpha, time_alpha=time_alpha, elastic_loss_weight=sp.elastic_loss_weight,
                   learning_rate=sp.learning_rate, adam_step=opt.step, beta1=opt.beta1, beta2=opt.beta2, grad_scale=1.0 / _world(),
                   rng_seed=models.rng_seed_of(coarse_key, fine_key), rng_offset=0)

  def _enqueue_grad(self):
    opt, sp, f = self.state.optimizer, self.sp, self.flags
    dev = opt.target.flat.device
    self._grad, self._stats = self.model.loss_and_grad(
        opt.target, self.batch, warp_extra=self.state.warp_extra, rngs={'fine': 0, 'coarse': 0}, grad_out=opt.grad, stats_out=opt.stats,
        bf16=self.bf16, dynamic=self.dyn.dev,
        background=_background_of(self.model, self.batch, sp, dev) if f['use_background_loss'] else None,
        elastic=dict(self.el, weight=sp.elastic_loss_weight) if f['use_elastic_loss'] else None,
        warp_reg={'weight': sp.warp_reg_loss_weight, 'alpha': sp.warp_reg_loss_alpha, 'scale': sp.warp_reg_loss_scale}
        if f['use_warp_reg_loss'] else None)

  def _enqueue(self):
    opt = self.state.optimizer
    self._enqueue_grad()
    grad, stats, _ = psum_gradients(self._grad, self._stats, fused=opt._gs)
    opt.apply_gradient_dynamic(grad, self.dyn)
    self.stats_static = stats

  def __call__(self, rng_key, scalar_params: Optional[ScalarParams] = None, warp_alpha=None, time_alpha=None, batch=None):
    sp = scalar_params or self.sp
    for name in ('background_loss_weight', 'background_noise_std', 'warp_reg_loss_weight', 'warp_reg_loss_alpha', 'warp_reg_loss_scale'):
      if getattr(sp, name) != getattr(self.sp, name):   # by-value arguments of the captured launches
        raise L.NrfError(f'GraphedTrainStep: ScalarParams.{name} is baked into the captured step; build a new GraphedTrainStep to change it')
    if batch is not None:
      self.load_batch(batch)
    if warp_alpha is not None:
      self.state.warp_alpha = warp_alpha
    if time_alpha is not None:
      self.state.time_alpha = time_alpha
    self._write(rng_key, sp, self.state.warp_alpha, self.state.time_alpha)
    self.graph.replay()
    if self.split:
      _, self.stats_static, _ = psum_gradients(self._grad, self._stats, fused=self.state.optimizer._gs)
      self.graph_adam.replay()
    self.state.optimizer.step += 1
    st = self.stats_static.clone()   # the static buffer is overwritten by the next replay
    return _stats_dict(st, sp, **self.flags)

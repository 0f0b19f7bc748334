"""The whole train step (nrf_train_step_loss_grad_ex -> all-reduce -> nrf_adam_step_dynamic) replayed from ONE hipGraph
(training.GraphedTrainStep; the reference jits the step into one XLA executable, train.py:254-262).  The captured launches
read everything that changes between steps from device memory (nrf_dynamic_scalars): the tests replay the same graph with
other rng keys, learning rates, warp_alpha and elastic weights and compare every replay with an eager train_step from the same
state.  Agreement is to float32 summation order, not bitwise: the per-ray sums and the embedding gradients are float atomics,
whose order differs between ANY two runs (two eager runs differ the same way)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _pair(cfg, B, warp_ids=(0, 1, 2, 3)):
  from nerfies_amd import models, training
  out = []
  for _ in range(2):   # same seed -> identical initial parameters
    model, fp = models.construct_nerf(11, cfg, B, [0, 1, 2, 3], [0, 1], list(warp_ids), 0.05, 1.0, device=DEV)
    out.append((model, training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=1.5)))
  assert torch.equal(out[0][1].optimizer.target.flat, out[1][1].optimizer.target.flat)
  return out


def _batch(B, seed, with_meta=False, nbg=0):
  g = torch.Generator().manual_seed(seed)
  o = (torch.rand(B, 3, generator=g) - 0.5).to(DEV)
  d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).to(DEV)
  b = {'origins': o, 'directions': d, 'rgb': torch.rand(B, 3, generator=g).to(DEV), 'metadata': {}}
  if with_meta:
    b['metadata'] = {'warp': torch.randint(0, 4, (B, 1), generator=g).to(DEV), 'camera': torch.randint(0, 2, (B, 1), generator=g).to(DEV)}
  if nbg:
    b['background_points'] = ((torch.rand(nbg, 3, generator=g) - 0.5) * 0.8).to(DEV)
  return b


def _close(a, b, layout, tol, what):
  worst = 0.0
  for name, off, shape in layout.entries:
    n = int(np.prod(shape))
    x, y = a[off:off + n], b[off:off + n]
    scale = y.abs().max().item()
    if scale > 0:
      worst = max(worst, (x - y).abs().max().item() / scale)
      assert (x - y).abs().max().item() <= tol * scale, (what, name, (x - y).abs().max().item(), scale)
  return worst


def test_graph_replay_equals_eager_config_a_shape():
  """Warp off, stratified sampling: 4 steps with a decaying learning rate from the same init, eager vs ONE captured graph."""
  from nerfies_amd import training

  class Cfg:
    num_coarse_samples, num_fine_samples, num_nerf_point_freqs = 64, 128, 8
    sigma_activation, use_stratified_sampling, use_viewdirs = 'softplus', True, True
  B = 128
  (me, se), (mg, sg) = _pair(Cfg, B)
  batches = [_batch(B, 50 + k) for k in range(4)]
  gstep = training.GraphedTrainStep(mg, sg, batches[0], training.ScalarParams(learning_rate=1e-3))
  assert torch.equal(sg.optimizer.target.flat, se.optimizer.target.flat) and sg.optimizer.step == 0   # the capture left no trace
  key = 7
  for k in range(4):
    sp = training.ScalarParams(learning_rate=1e-3 * 0.5 ** k)
    se, st_e, next_key = training.train_step(me, key, se, batches[k], sp)
    st_g = gstep(key, scalar_params=sp, batch=batches[k])
    key = next_key
    assert abs(st_e['fine']['loss/rgb'].item() - st_g['fine']['loss/rgb'].item()) < 1e-7
    assert abs(st_e['coarse']['metric/psnr'].item() - st_g['coarse']['metric/psnr'].item()) < 1e-4
    w = _close(sg.optimizer.grad.cpu(), se.optimizer.grad.cpu(), mg.layout, 2e-5, f'gradient of step {k}')
    # Adam turns rounding-level gradient entries into sign-like updates: parameters within lr per step, and tight in L2
    dp = (sg.optimizer.target.flat - se.optimizer.target.flat)
    assert dp.abs().max().item() <= 2e-3 and dp.norm().item() <= 2e-2 * (se.optimizer.target.flat - 0).norm().item() * 1e-3 + 1e-4
    # ... and bring the two replicas together again, so that every step's comparison is of ONE step from the same state
    for dst, src in ((sg.optimizer.target.flat, se.optimizer.target.flat), (sg.optimizer.m, se.optimizer.m), (sg.optimizer.v, se.optimizer.v)):
      dst.copy_(src)
  assert sg.optimizer.step == se.optimizer.step == 4
  print(f'[graphed step, config A shape] last-step gradient: worst leaf {w:.1e} of its max-abs vs eager')


def test_graph_replay_follows_the_schedules():
  """Warp + elastic + background (ids and noise drawn by the library): the SAME graph replayed with another warp_alpha, elastic
  weight, learning rate and rng key must equal an eager step with those values -- i.e. none of them is baked into the capture."""
  from nerfies_amd import training

  class Cfg:
    num_coarse_samples, num_fine_samples, num_nerf_point_freqs = 32, 32, 6
    sigma_activation, use_stratified_sampling, use_viewdirs = 'softplus', True, True
    use_warp, warp_field_type, num_warp_freqs, num_warp_features, use_camera_metadata = True, 'se3', 4, 8, True
  B, NBG = 96, 256
  (me, se), (mg, sg) = _pair(Cfg, B)
  batch = _batch(B, 60, with_meta=True, nbg=NBG)
  kw = dict(use_elastic_loss=True, elastic_reduce_method='weight', use_background_loss=True)
  sp0 = training.ScalarParams(learning_rate=1e-3, elastic_loss_weight=0.01, background_loss_weight=1.0)
  gstep = training.GraphedTrainStep(mg, sg, batch, sp0, **kw)
  seen = []
  for k, (alpha, el_w, lr, key) in enumerate([(0.5, 0.01, 1e-3, 3), (2.25, 0.004, 5e-4, 99), (4.0, 1e-5, 1e-4, 12345)]):
    sp = training.ScalarParams(learning_rate=lr, elastic_loss_weight=el_w, background_loss_weight=1.0)
    se = se.replace(warp_alpha=alpha)
    se, st_e, _ = training.train_step(me, key, se, batch, sp, **kw)
    st_g = gstep(key, scalar_params=sp, warp_alpha=alpha)
    for a, b in ((st_e['coarse']['loss/total'], st_g['coarse']['loss/total']), (st_e['coarse']['loss/elastic'], st_g['coarse']['loss/elastic']),
                 (st_e['background_loss'], st_g['background_loss']), (st_e['fine']['loss/rgb'], st_g['fine']['loss/rgb'])):
      assert abs(a.item() - b.item()) <= 1e-6 + 1e-5 * abs(a.item()), (k, a.item(), b.item())
    _close(sg.optimizer.grad.cpu(), se.optimizer.grad.cpu(), mg.layout, 5e-5, f'gradient of step {k}')
    seen.append(st_g['coarse']['loss/elastic'].item())
    # keep the two replicas together for the next comparison (Adam amplifies rounding-level gradient entries)
    for dst, src in ((sg.optimizer.target.flat, se.optimizer.target.flat), (sg.optimizer.m, se.optimizer.m), (sg.optimizer.v, se.optimizer.v)):
      dst.copy_(src)
  assert len(set(seen)) == 3   # the elastic term did move with alpha / the parameters
  with pytest.raises(Exception):   # what IS baked in must be refused, not silently ignored
    gstep(1, scalar_params=training.ScalarParams(learning_rate=1e-3, elastic_loss_weight=0.01, background_loss_weight=2.0))


def test_library_background_draw_matches_the_reference_distribution():
  """training.py:121-126 on the device: ids uniform over model.warp_ids, noise ~ N(0, noise_std^2) per coordinate, a different
  draw for every key, the same draw for the same key."""
  import ctypes as C
  from nerfies_amd import lib as L, training

  class Cfg:
    num_coarse_samples, num_fine_samples, num_nerf_point_freqs = 8, 8, 4
    sigma_activation, use_stratified_sampling, use_viewdirs = 'softplus', True, True
    use_warp, warp_field_type, num_warp_freqs, num_warp_features = True, 'se3', 4, 8
  B, NBG = 8, 8192
  ids = (3, 5, 6, 9, 10)
  (model, state), _ = _pair(Cfg, B, warp_ids=ids)
  batch = _batch(B, 70, nbg=NBG)
  batch['metadata'] = {'warp': torch.full((B, 1), 5, dtype=torch.int32, device=DEV)}
  sp = training.ScalarParams(learning_rate=0.0, background_loss_weight=1.0, background_noise_std=0.01)
  draws = []
  for key in (1, 1, 2):
    state, _, _ = training.train_step(model, key, state, batch, sp, use_background_loss=True)
    ws = model.workspace(B, True, DEV, NBG, False)
    torch.cuda.synchronize()
    off = C.c_int64(0)
    L.check(model.lib.nrf_debug_ws_offset(model.handle, b'bg_points', 2, C.byref(off)), model.lib)
    pts = ws[off.value:off.value + 3 * NBG].reshape(NBG, 3).clone()
    L.check(model.lib.nrf_debug_ws_offset(model.handle, b'bg_ids', 2, C.byref(off)), model.lib)
    got_ids = ws[off.value:off.value + NBG].view(torch.int32).clone()
    draws.append((pts, got_ids))
  (p1, i1), (p1b, i1b), (p2, i2) = draws
  assert torch.equal(p1, p1b) and torch.equal(i1, i1b) and not torch.equal(p1, p2) and not torch.equal(i1, i2)
  noise = (p1 - batch['background_points']).cpu().double().numpy()
  assert abs(noise.mean()) < 5e-4 and abs(noise.std() - 0.01) < 3e-4
  counts = np.array([(i1.cpu().numpy() == v).sum() for v in ids])
  assert counts.sum() == NBG and (np.abs(counts / NBG - 1 / len(ids)) < 0.03).all(), counts


def test_graph_replay_in_the_bf16_mode_with_warp_elastic_background():
  """train.py --graph together with --bf16: GraphedTrainStep(bf16=True) captures and replays the bf16 chain kernels,
  the bf16 SE3 forward / tangent / reverse kernels (per-launch hipFuncSetAttribute calls included) and the bf16 wgrad.  Three
  replays with other schedules' values against eager train_step(bf16=True) from the same state: the bf16 kernels are
  deterministic given their inputs, so the agreement is the float32 paths' atomics order, as in the fp32 test above."""
  from nerfies_amd import training

  class Cfg:
    num_coarse_samples, num_fine_samples, num_nerf_point_freqs = 32, 32, 6
    sigma_activation, use_stratified_sampling, use_viewdirs = 'softplus', True, True
    use_warp, warp_field_type, num_warp_freqs, num_warp_features, use_camera_metadata = True, 'se3', 4, 8, True
  B, NBG = 96, 256
  (me, se), (mg, sg) = _pair(Cfg, B)
  batch = _batch(B, 61, with_meta=True, nbg=NBG)
  kw = dict(use_elastic_loss=True, elastic_reduce_method='weight', use_background_loss=True)
  sp0 = training.ScalarParams(learning_rate=1e-3, elastic_loss_weight=0.01, background_loss_weight=1.0)
  gstep = training.GraphedTrainStep(mg, sg, batch, sp0, bf16=True, **kw)
  assert torch.equal(sg.optimizer.target.flat, se.optimizer.target.flat) and sg.optimizer.step == 0
  for k, (alpha, el_w, lr, key) in enumerate([(0.5, 0.01, 1e-3, 3), (2.25, 0.004, 5e-4, 99), (4.0, 1e-5, 1e-4, 12345)]):
    sp = training.ScalarParams(learning_rate=lr, elastic_loss_weight=el_w, background_loss_weight=1.0)
    se = se.replace(warp_alpha=alpha)
    se, st_e, _ = training.train_step(me, key, se, batch, sp, bf16=True, **kw)
    st_g = gstep(key, scalar_params=sp, warp_alpha=alpha)
    for a, b in ((st_e['coarse']['loss/total'], st_g['coarse']['loss/total']), (st_e['coarse']['loss/elastic'], st_g['coarse']['loss/elastic']),
                 (st_e['background_loss'], st_g['background_loss']), (st_e['fine']['loss/rgb'], st_g['fine']['loss/rgb'])):
      assert abs(a.item() - b.item()) <= 1e-6 + 1e-5 * abs(a.item()), (k, a.item(), b.item())
    _close(sg.optimizer.grad.cpu(), se.optimizer.grad.cpu(), mg.layout, 2e-4, f'bf16 gradient of step {k}')
    for dst, src in ((sg.optimizer.target.flat, se.optimizer.target.flat), (sg.optimizer.m, se.optimizer.m), (sg.optimizer.v, se.optimizer.v)):
      dst.copy_(src)
  assert sg.optimizer.step == se.optimizer.step == 3

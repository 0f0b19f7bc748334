"""The PRODUCT train_step on two ranks (training.py:266-267, train.py:254-262): two processes share cuda:0 (RCCL refuses
two ranks on one device, so the collective backend is gloo on the same CUDA tensors -- the code path is the one
`torchrun bench.py --gpus N` takes, only the transport differs), each runs nerfies_amd.training.train_step on its half
of the rays for three Adam steps, and the parameters must match a single-process run on the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.gpu
WORLD, B, STEPS, LR = 2, 96, 3, 1e-3
KW = dict(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=8, use_stratified_sampling=True, use_warp=True,
          use_camera_metadata=True)


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _inputs():
  from oracle import nerfies_oracle as O
  spec = O.ModelSpec(**KW)
  p = O.init_params(spec, seed=31, trained_like=True)
  b = O.synthetic_batch(B, seed=32)
  g = torch.Generator().manual_seed(33)
  uni = [(torch.rand(B, spec.num_coarse_samples, generator=g), torch.rand(B, spec.num_fine_samples, generator=g)) for _ in range(STEPS)]
  return spec, p, b, uni


def _run(rank, world):
  """STEPS train steps on rows [rank*B/world, (rank+1)*B/world) of the batch; returns (flat params, per-step stats)."""
  import helpers as H
  from nerfies_amd import training
  spec, p, b, uni = _inputs()
  per = B // world
  sl = slice(rank * per, (rank + 1) * per)
  model, fp = H.gpu_model(spec, p, per)
  gb = H.gpu_batch(b)
  gb = {k: (v[sl] if torch.is_tensor(v) else {kk: vv[sl] for kk, vv in v.items()}) for k, v in gb.items()}
  state = training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=4.0)
  sp = training.ScalarParams(learning_rate=LR, elastic_loss_weight=0.01)
  # step 0, before Adam: the all-reduced mean of the shard gradients (lax.pmean, training.py:266) IS the full-batch gradient
  import torch.distributed as dist
  grad0, _ = model.loss_and_grad(fp, gb, warp_extra=state.warp_extra, rngs={'coarse': uni[0][0][sl].to(H.DEV), 'fine': uni[0][1][sl].to(H.DEV)},
                                 elastic={'weight': 0.01, 'reduce_method': 'weight'})
  grad0 = grad0.clone()
  if world > 1:
    dist.all_reduce(grad0)
    grad0 /= world
  hist = []
  for k, (t_rand, u) in enumerate(uni):
    state, stats, _ = training.train_step(model, k, state, gb, sp, use_elastic_loss=True, elastic_reduce_method='weight',
                                          rngs={'coarse': t_rand[sl].to(H.DEV), 'fine': u[sl].to(H.DEV)})
    hist.append([stats['coarse']['loss/rgb'].item(), stats['fine']['loss/rgb'].item(), stats['coarse']['loss/elastic'].item()])
  torch.cuda.synchronize()
  return fp.flat.cpu(), np.array(hist), grad0.cpu()


def _worker(rank, port, tmp):
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.cuda.set_device(0)
  dist.init_process_group('gloo', rank=rank, world_size=WORLD)
  flat, hist, grad0 = _run(rank, WORLD)
  both = [torch.empty_like(flat) for _ in range(WORLD)]
  dist.all_gather(both, flat)
  assert torch.equal(both[0], both[1])   # replicas stay bit-identical: same all-reduced gradient, same Adam
  if rank == 0:
    torch.save({'flat': flat, 'hist': hist, 'grad0': grad0}, tmp)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_train_step_equals_full_batch(tmp_path):
  from nerfies_amd import params as P
  import helpers as H
  tmp = str(tmp_path / 'out.pt')
  mp.spawn(_worker, args=(_free_port(), tmp), nprocs=WORLD, join=True)
  got = torch.load(tmp, weights_only=False)
  want, hist, grad_full = _run(0, 1)
  spec, p, _, _ = _inputs()
  model, fp0 = H.gpu_model(spec, p, B)
  # the first-step gradient through the collective, leaf by leaf, against the full-batch gradient: only the float32
  # summation order differs (tiles over other row sets, atomics)
  worst = 0.0
  for name, off, shape in model.layout.entries:
    n = int(np.prod(shape))
    a, b = got['grad0'][off:off + n], grad_full[off:off + n]
    scale = b.abs().max().item()
    if scale > 0:
      worst = max(worst, (a - b).abs().max().item() / scale)
  print(f'[2 ranks vs 1] step-0 all-reduced gradient: worst leaf {worst:.2e} of its max-abs entry')
  assert worst < 1e-5
  init = fp0.flat.cpu()
  travel = (want - init).norm().item()
  diff = (got['flat'] - want).norm().item()
  print(f'[2 ranks vs 1] |dp| {diff:.3e} over a travel of {travel:.3e}; max entry {(got["flat"] - want).abs().max().item():.2e}')
  assert travel > 0.1 * LR * STEPS * np.sqrt(want.numel())     # Adam moved (most entries by ~lr per step)
  # float32 summation order differs (64-row tiles over different row sets, atomics in the per-ray sums) and Adam turns
  # rounding-level gradient entries into sign-like updates: measured 0.9 - 1.1 % of the travel after three steps
  assert diff < 3e-2 * travel
  # single entries: an entry whose gradient is at rounding level gets sign-like Adam updates (m / sqrt(v)), so two float32
  # summation orders (64-row tiles over different row sets) may move it in different directions -- bounded by the travel
  assert (got['flat'] - want).abs().max().item() < LR * STEPS
  # pmean of the statistics (training.py:267): the mean of the two shard MSEs is the full-batch MSE; the elastic loss
  # likewise (mean over rays)
  np.testing.assert_allclose(got['hist'], hist, rtol=2e-4, atol=1e-7)


# ---------------------------------------------------------------------------------------------
# the whole step from hipGraphs on two ranks (training.GraphedTrainStep; train.py --graph)
# ---------------------------------------------------------------------------------------------
def _run_keys(rank, world, graphed):
  """STEPS steps driven by integer rng keys (the library's Philox streams), eager or replayed from the captured step."""
  import helpers as H
  from nerfies_amd import training
  spec, p, b, _ = _inputs()
  per = B // world
  sl = slice(rank * per, (rank + 1) * per)
  model, fp = H.gpu_model(spec, p, per)
  gb = H.gpu_batch(b)
  gb = {k: (v[sl] if torch.is_tensor(v) else {kk: vv[sl] for kk, vv in v.items()}) for k, v in gb.items()}
  state = training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=4.0)
  sp = training.ScalarParams(learning_rate=LR, elastic_loss_weight=0.01)
  kw = dict(use_elastic_loss=True, elastic_reduce_method='weight')
  gstep = training.GraphedTrainStep(model, state, gb, sp, **kw) if graphed else None
  hist = []
  for k in range(STEPS):
    if graphed:
      stats = gstep(100 + k)
    else:
      state, stats, _ = training.train_step(model, 100 + k, state, gb, sp, **kw)
    hist.append([stats['coarse']['loss/rgb'].item(), stats['fine']['loss/rgb'].item(), stats['coarse']['loss/elastic'].item()])
  torch.cuda.synchronize()
  return fp.flat.cpu(), np.array(hist), (gstep.split if graphed else None)


def _graph_worker(rank, port, tmp):
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.cuda.set_device(0)
  dist.init_process_group('gloo', rank=rank, world_size=WORLD)
  eager, hist_e, _ = _run_keys(rank, WORLD, False)
  flat, hist_g, split = _run_keys(rank, WORLD, True)
  both = [torch.empty_like(flat) for _ in range(WORLD)]
  dist.all_gather(both, flat)
  assert torch.equal(both[0], both[1])   # replicas bit-identical after STEPS replays
  if rank == 0:
    torch.save({'flat': flat, 'eager': eager, 'hist_g': hist_g, 'hist_e': hist_e, 'split': split}, tmp)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_graphed_step_keeps_replicas_identical_and_equals_eager(tmp_path):
  """GraphedTrainStep with a process group: the gradient all-reduce and the 1/world factor inside the replayed step
  (training.py:264-269).  gloo's collective is a host call, so the step is two graphs around it (RCCL's is captured with the
  rest: tests/test_gpu_rccl.py); the replicas must stay bit-identical and follow the eager two-rank run."""
  import helpers as H
  tmp = str(tmp_path / 'graph.pt')
  mp.spawn(_graph_worker, args=(_free_port(), tmp), nprocs=WORLD, join=True)
  got = torch.load(tmp, weights_only=False)
  assert got['split'] is True
  spec, p, _, _ = _inputs()
  _, fp0 = H.gpu_model(spec, p, B)
  travel = (got['eager'] - fp0.flat.cpu()).norm().item()
  diff = (got['flat'] - got['eager']).norm().item()
  print(f'[2 ranks, graph vs eager] |dp| {diff:.3e} over a travel of {travel:.3e}')
  assert travel > 0.1 * LR * STEPS * np.sqrt(got['flat'].numel())
  assert diff < 3e-2 * travel                       # float32 atomics order only (same bound as the one-rank comparison above)
  np.testing.assert_allclose(got['hist_g'], got['hist_e'], rtol=2e-4, atol=1e-7)


# ---------------------------------------------------------------------------------------------
# eval: frame-tile parallel render_image (BASELINE configs[4] "8-GPU image-tile parallel"; evaluation.py:61-99, eval.py:339)
# ---------------------------------------------------------------------------------------------
FRAME_H, FRAME_W, FRAME_CHUNK = 37, 29, 256    # 1073 rays = 4 full chunks + a ragged fifth: 3 chunks on rank 0, 2 on rank 1


def _render_frames():
  import helpers as H
  from nerfies_amd import evaluation, training
  from oracle import nerfies_oracle as O
  # eval renders deterministically (eval.py:239 forces use_stratified_sampling off): with stratified draws a ray's uniforms are
  # indexed by its position in the launch, and the per-chunk split moves rays to other positions
  spec = O.ModelSpec(**dict(KW, use_stratified_sampling=False))
  p = O.init_params(spec, seed=31, trained_like=True)
  model, fp = H.gpu_model(spec, p, FRAME_CHUNK)
  g = torch.Generator().manual_seed(77)
  n = FRAME_H * FRAME_W
  d = torch.randn(n, 3, generator=g)
  rays = {'origins': (torch.rand(n, 3, generator=g) - 0.5).reshape(FRAME_H, FRAME_W, 3).to(H.DEV),
          'directions': (d / d.norm(dim=-1, keepdim=True)).reshape(FRAME_H, FRAME_W, 3).to(H.DEV),
          'metadata': {'warp': torch.full((FRAME_H, FRAME_W, 1), 2, dtype=torch.int32, device=H.DEV),
                       'camera': torch.full((FRAME_H, FRAME_W, 1), 1, dtype=torch.int32, device=H.DEV)}}
  state = training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=4.0)
  out = {}
  for mode in ('band', 'chunk'):
    fn = evaluation.GraphedChunkRenderer(model)        # hipGraph replay: fixed-size chunks
    img = evaluation.render_image(state, rays, fn, chunk=FRAME_CHUNK, tile_parallel=mode)
    out[mode] = {k: v.cpu() for k, v in img.items()}
    out[mode + '_captures'] = fn.captures
  torch.cuda.synchronize()
  return out


def _frame_worker(rank, port, tmp):
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.cuda.set_device(0)
  dist.init_process_group('gloo', rank=rank, world_size=WORLD)
  out = _render_frames()
  if rank == 1:      # any rank holds the whole frame after the gather (the reference keeps replica 0, evaluation.py:92)
    torch.save(out, tmp)
  dist.barrier()
  dist.destroy_process_group()


def test_band_parallel_frame_is_bit_identical_to_the_single_rank_frame(tmp_path):
  """render_image on two ranks -- a contiguous band of whole chunks per rank and ONE all_gather per frame (default), or the
  reference's per-chunk split -- against the single-process frame: rows are rendered independently of their position in a
  launch, so all three frames must agree bit for bit (rgb, depth, med_depth, acc), with the warp field on."""
  tmp = str(tmp_path / 'frame.pt')
  mp.spawn(_frame_worker, args=(_free_port(), tmp), nprocs=WORLD, join=True)
  got = torch.load(tmp, weights_only=False)
  want = _render_frames()          # no process group in this process: the plain chunk loop
  assert want['band_captures'] == 1 and got['band_captures'] == 1      # one graph serves the frame (tail padded to the chunk)
  for mode in ('band', 'chunk'):
    assert set(got[mode]) == set(want['band']) >= {'rgb', 'depth', 'acc'}
    for k, v in want['band'].items():
      assert v.shape[:2] == (FRAME_H, FRAME_W)
      assert torch.equal(got[mode][k], v), (mode, k, (got[mode][k] - v).abs().max().item())

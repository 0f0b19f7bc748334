"""The N>1 path on CPU: two gloo ranks exercising the host-side collective logic the GPU path uses
(training.psum_gradients, evaluation.render_image).  The per-shard gradients come from the CPU
oracle -- the HIP kernels need a GPU; their data-parallel equivalence is tested on the GPU in
tests/test_gpu_parity.py::test_data_parallel_gradient_equals_full_batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nerfies_oracle as O

WORLD = 2


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _spec():
  return O.ModelSpec(num_coarse_samples=8, num_fine_samples=8, num_nerf_point_freqs=4)


def _flat(tree):
  return torch.cat([t.reshape(-1) for _, t in O.tree_leaves_with_path(tree)])


def _shard(batch, r, n):
  B = batch['origins'].shape[0]
  per = B // n
  sl = slice(r * per, (r + 1) * per)
  out = {k: v[sl] for k, v in batch.items() if torch.is_tensor(v)}
  out['metadata'] = {k: v[sl] for k, v in batch['metadata'].items()}
  return out


def _fake_model_fn(key0, key1, params, rays, warp_extra):
  """A deterministic per-ray function standing in for NerfModel.apply (any rank computes the same
  value for the same ray)."""
  o, d = rays['origins'], rays['directions']
  rgb = torch.sigmoid(o * 3.0 + d)
  return {'fine': {'rgb': rgb, 'depth': (o * d).sum(-1), 'acc': o.norm(dim=-1)}}


class _FixedChunkFn:
  wants_fixed_chunks = True   # what evaluation.GraphedChunkRenderer declares

  def __init__(self):
    self.sizes = set()

  def __call__(self, key0, key1, params, rays, warp_extra):
    self.sizes.add(rays['origins'].shape[0])
    return _fake_model_fn(key0, key1, params, rays, warp_extra)


class _State:
  class optimizer:
    target = None
  warp_extra = {}


def _worker(rank, port, tmp):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=WORLD)
  torch.set_num_threads(1)
  from nerfies_amd import evaluation, training
  # ---- training: sum of shard gradients * 1/world == full-batch gradient (training.py:266) ----
  spec = _spec()
  params = O.init_params(spec, seed=3, trained_like=True)
  batch = O.synthetic_batch(8, seed=4)
  loss, stats, grads, _ = O.loss_and_grad(params, spec, _shard(batch, rank, WORLD))
  g = _flat(grads).float()
  st = torch.zeros(8)
  st[0], st[1] = stats['coarse']['loss/rgb'], stats['fine']['loss/rgb']
  if rank == 0:     # both call forms must give the same reduction: separate buffers ...
    pass
  fused = torch.cat([g, st])                      # ... and the Optimizer's fused [grad | stats] buffer (one all-reduce)
  g2, st2, n = training.psum_gradients(fused[:g.numel()], fused[g.numel():], fused=fused)
  g, st, n = training.psum_gradients(g, st)
  assert n == WORLD and torch.equal(g2, g) and torch.allclose(st2, st)
  # ---- eval: rank-sliced chunks + all_gather reassemble the full image (evaluation.py:62-99) ----
  rays = {'origins': torch.linspace(-1, 1, 5 * 7 * 3).reshape(5, 7, 3), 'directions': torch.ones(5, 7, 3) * 0.1}
  # the reference's order: every chunk cut `world` ways, one gather per chunk (evaluation.py:61-92)
  img = evaluation.render_image(_State, rays, _fake_model_fn, device_count=WORLD, chunk=9, tile_parallel='chunk')   # 35 px: ragged
  fixed = _FixedChunkFn()   # a graph-replaying renderer: every call must see the same per-rank slice length
  img_fixed = evaluation.render_image(_State, rays, fixed, device_count=WORLD, chunk=9, tile_parallel='chunk')
  assert fixed.sizes == {5}, fixed.sizes    # ceil(9 / 2) rays per rank in every chunk, the tail padded up to it
  for k in img:
    assert torch.equal(img[k], img_fixed[k])
  # the default: every rank renders a contiguous band of WHOLE chunks (35 px = 4 chunks of 9 -> 2 per rank), ONE gather per frame
  calls = []
  counting = lambda k0, k1, p, r, e: (calls.append(r['origins'].shape[0]), _fake_model_fn(k0, k1, p, r, e))[1]
  img_band = evaluation.render_image(_State, rays, counting, device_count=WORLD, chunk=9)
  assert calls == ([9, 9] if rank == 0 else [9, 8]), calls       # whole chunks, the ragged tail on the last rank
  fixed = _FixedChunkFn()
  img_band_fixed = evaluation.render_image(_State, rays, fixed, device_count=WORLD, chunk=9)
  assert fixed.sizes == {9}, fixed.sizes    # full-size launches only: the tail is edge-padded to the chunk
  for k in img:
    assert torch.equal(img[k], img_band[k]) and torch.equal(img[k], img_band_fixed[k]), k
  # an odd chunk count (35 px = 5 chunks of 8: 3 + 2): balanced bands of unequal length, assembled from each rank's leading rows
  calls = []
  img_odd = evaluation.render_image(_State, rays, counting, device_count=WORLD, chunk=8)
  assert calls == ([8, 8, 8] if rank == 0 else [8, 3]), calls
  for k in img:
    assert torch.allclose(img[k], img_odd[k], atol=1e-7, rtol=0), k
  # more ranks than chunks: the idle rank still joins the frame's one collective
  img_one = evaluation.render_image(_State, rays, _fake_model_fn, device_count=WORLD, chunk=64)
  for k in img:   # (torch's CPU sigmoid differs by an ulp between a 9-row and a 35-row call: vector body vs scalar tail)
    assert torch.allclose(img[k], img_one[k], atol=1e-7, rtol=0), k
  # ---- data: every rank takes its own 1/world slice of each global batch (core.py:110-121) ----
  from nerfies_amd import datasets
  n = 50
  table = datasets.RayTable({'origins': torch.arange(n * 3, dtype=torch.float32).reshape(n, 3),
                             'metadata/warp': torch.arange(n, dtype=torch.int32).reshape(n, 1)}, n)
  mine = [b['metadata']['warp'][:, 0] for b in table.batches(16, repeat=False)]
  gathered = []
  for b in mine:
    parts = [torch.zeros(16 // WORLD if len(b) == 16 // WORLD else len(b), dtype=torch.int32) for _ in range(WORLD)]
    dist.all_gather(parts, b)
    gathered.append(torch.cat(parts))
  try:
    next(table.batches(15))
    odd = 'accepted'
  except ValueError:
    odd = 'rejected'
  if rank == 0:
    torch.save({'grad_sum': g, 'stats': st, 'img': img, 'batches': gathered, 'odd': odd}, tmp)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_gradients_and_render(tmp_path):
  tmp = str(tmp_path / 'out.pt')
  mp.spawn(_worker, args=(_free_port(), tmp), nprocs=WORLD, join=True)
  got = torch.load(tmp)
  spec = _spec()
  params = O.init_params(spec, seed=3, trained_like=True)
  batch = O.synthetic_batch(8, seed=4)
  loss, stats, grads, _ = O.loss_and_grad(params, spec, batch)
  full = _flat(grads).float()
  mean = got['grad_sum'] / WORLD          # the factor nrf_adam_step applies as grad_scale
  np.testing.assert_allclose(mean.numpy(), full.numpy(), atol=1e-6 * max(full.abs().max().item(), 1.0))
  np.testing.assert_allclose(got['stats'][0].item(), stats['coarse']['loss/rgb'].item(), rtol=1e-5)
  np.testing.assert_allclose(got['stats'][1].item(), stats['fine']['loss/rgb'].item(), rtol=1e-5)
  # single-process render of the same image
  from nerfies_amd import evaluation
  rays = {'origins': torch.linspace(-1, 1, 5 * 7 * 3).reshape(5, 7, 3), 'directions': torch.ones(5, 7, 3) * 0.1}
  ref = evaluation.render_image(_State, rays, _fake_model_fn, device_count=1, chunk=9)
  assert set(ref) == set(got['img']) == {'rgb', 'depth', 'acc'}
  for k in ref:
    assert ref[k].shape == got['img'][k].shape
    np.testing.assert_allclose(got['img'][k].numpy(), ref[k].numpy(), atol=1e-7)
  assert ref['rgb'].shape == (5, 7, 3) and ref['depth'].shape == (5, 7)
  # the two ranks' shards of each batch, concatenated in rank order, are the global batches in table order
  assert got['odd'] == 'rejected'          # batch_size % device_count != 0 (train.py:155-156)
  assert torch.equal(torch.cat(got['batches']), torch.arange(50, dtype=torch.int32))
  assert [len(b) for b in got['batches']] == [16, 16, 16, 2]
